"""Host-side check of the device helpers that decide WHICH tile pair a wave works on (csrc/sym_kernels.h: unit_seek /
unit_next for the row-major and the blocked order, xcd_swizzle): the functions are plain integer arithmetic, so they are
compiled for the host (g++, `__device__` defined away) straight from the header and checked exhaustively -- seek and
next agree, every tile pair I <= J is visited exactly once, the XCD numbering is a bijection."""
import os
import subprocess

from conftest import ROOT

HARNESS = r'''
int main() {
  long bad = 0;
  std::vector<int> Ts;
  for (int T = 1; T <= 140; ++T) Ts.push_back(T);
  Ts.push_back(157); Ts.push_back(384); Ts.push_back(1563); Ts.push_back(2049);
  for (int T : Ts) {
    const long n_units = (long)T * (T + 1) / 2;
    for (int order = 0; order <= 1; ++order) {
      std::set<std::pair<int, int>> seen;
      int I = -1, J = -1;
      for (long u = 0; u < n_units; ++u) {
        int Is, Js;
        unit_seek(order, u, T, Is, Js);
        if (u == 0) { I = Is; J = Js; }
        if (Is != I || Js != J) { ++bad; I = Is; J = Js; }
        if (I < 0 || J < I || J >= T) ++bad;
        if (!seen.insert({I, J}).second) ++bad;
        unit_next(order, T, I, J);
      }
      if ((long)seen.size() != n_units) ++bad;
    }
  }
  for (long nwg : {1L, 7L, 8L, 9L, 136L, 1024L, 3072L, 8191L}) {
    std::set<long> s2;
    for (long b = 0; b < nwg; ++b) { long w = xcd_swizzle(b, nwg); if (w < 0 || w >= nwg) ++bad; s2.insert(w); }
    if ((long)s2.size() != nwg) ++bad;
  }
  {   // 1e6 blobs = 15 625 tiles: seek agrees with a run of nexts at a few places
    const int T = 15625; const long n_units = (long)T * (T + 1) / 2;
    for (long u0 : {0L, 123456789L, n_units / 2, n_units - 6000}) {
      int I, J; unit_seek(1, u0, T, I, J);
      for (long u = u0; u < u0 + 5000 && u < n_units; ++u) {
        int Is, Js; unit_seek(1, u, T, Is, Js);
        if (Is != I || Js != J) { ++bad; I = Is; J = Js; }
        unit_next(1, T, I, J);
      }
    }
  }
  printf("problems %ld\n", bad);
  return bad != 0;
}
'''


def test_unit_order_and_xcd_numbering(tmp_path):
  src = open(os.path.join(ROOT, "rigidmultiblobswall_amd", "csrc", "sym_kernels.h")).read()
  begin = src.index("__device__ __forceinline__ void unit_to_tiles(long u, int T, int& I, int& J) {")
  end = src.index("template <int KIND, bool WALL, bool PERIODIC>\n__global__", begin)
  code = ("#include <cmath>\n#include <cstdio>\n#include <set>\n#include <utility>\n#include <vector>\n"
          "#define __device__\n#define __forceinline__ inline\n" + src[begin:end] + HARNESS)
  cpp, exe = str(tmp_path / "unit_order.cpp"), str(tmp_path / "unit_order")
  with open(cpp, "w") as fh:
    fh.write(code)
  subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, cpp])
  res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
  assert res.returncode == 0 and "problems 0" in res.stdout, res.stdout + res.stderr


HARNESS2 = r'''
int main() {
  long bad = 0;
  std::vector<int> Ts;
  for (int T = 1; T <= 140; ++T) Ts.push_back(T);
  Ts.push_back(157); Ts.push_back(384); Ts.push_back(1563); Ts.push_back(2049);
  for (int T : Ts) {
    const long n_units = units2_total(T);
    long expect = 0;
    for (int p = 0; 2 * p < T; ++p) expect += T - 2 * p;
    if (expect != n_units) ++bad;
    for (int order = 0; order <= 1; ++order) {
      std::set<std::pair<int, int>> seen;
      int p = -1, J = -1;
      for (long u = 0; u < n_units; ++u) {
        int ps, Js;
        unit2_seek(order, u, T, ps, Js);
        if (u == 0) { p = ps; J = Js; }
        if (ps != p || Js != J) { ++bad; p = ps; J = Js; }
        if (p < 0 || 2 * p >= T || J < 2 * p || J >= T) ++bad;
        if (!seen.insert({p, J}).second) ++bad;
        unit2_next(order, T, p, J);
      }
      if ((long)seen.size() != n_units) ++bad;
    }
  }
  {   // 1e6 blobs = 15 625 tiles: seek agrees with a run of nexts at a few places
    const int T = 15625; const long n_units = units2_total(T);
    for (int order = 0; order <= 1; ++order)
      for (long u0 : {0L, 23456789L, n_units / 2, n_units - 6000}) {
        int p, J; unit2_seek(order, u0, T, p, J);
        for (long u = u0; u < u0 + 5000 && u < n_units; ++u) {
          int ps, Js; unit2_seek(order, u, T, ps, Js);
          if (ps != p || Js != J) { ++bad; p = ps; J = Js; }
          unit2_next(order, T, p, J);
        }
      }
  }
  printf("problems %ld\n", bad);
  return bad != 0;
}
'''


def test_row_pair_unit_order(tmp_path):
  """The unit grid of the two-targets-per-lane kernel (csrc/sym2t_kernels.h: unit2_seek / unit2_next, plain and blocked):
  every (row pair p, tile J >= 2p) exactly once, seek and next agree."""
  src = open(os.path.join(ROOT, "rigidmultiblobswall_amd", "csrc", "sym2t_kernels.h")).read()
  begin = src.index("// ---- unit order ----")
  end = src.index("template <int KIND, bool WALL>\n__global__", begin)
  code = ("#include <cmath>\n#include <cstdio>\n#include <set>\n#include <utility>\n#include <vector>\n"
          "#define __device__\n#define __host__\n#define __forceinline__ inline\nconstexpr int kOrdShift = 5;\n" + src[begin:end] + HARNESS2)
  cpp, exe = str(tmp_path / "unit2_order.cpp"), str(tmp_path / "unit2_order")
  with open(cpp, "w") as fh:
    fh.write(code)
  subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, cpp])
  res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
  assert res.returncode == 0 and "problems 0" in res.stdout, res.stdout + res.stderr
