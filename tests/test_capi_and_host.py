"""CPU-side checks of the boundary: the HIP shared library loads (no GPU needed), exports every
symbol include/rmb_mobility.h declares, reports "no device" loudly instead of falling back, and the
host-side helpers mirror the reference semantics."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols():
  src = open(os.path.join(ROOT, "include", "rmb_mobility.h")).read()
  src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
  return sorted(set(re.findall(r"\b(rmb_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
  from rigidmultiblobswall_amd import _lib
  lib = ctypes.CDLL(_lib.LIB_PATH)
  names = _declared_symbols()
  assert len(names) >= 20
  for n in names:
    assert hasattr(lib, n), "missing export: " + n
  # and the ctypes table covers exactly the header
  assert sorted(_lib.SYMBOLS) == names


def test_version_and_error_strings():
  from rigidmultiblobswall_amd import _lib
  lib = _lib.load()
  assert b"gfx950" in lib.rmb_version()
  assert isinstance(lib.rmb_last_error(), bytes)


def test_no_silent_cpu_fallback():
  """Without a GPU the product path must raise, never compute on the CPU."""
  import torch
  if torch.cuda.is_available():
    pytest.skip("GPU present")
  from rigidmultiblobswall_amd import _lib, mobility
  assert _lib.device_count() == 0
  r = np.random.rand(8, 3) + 1.0
  with pytest.raises(_lib.RmbError):
    mobility.single_wall_mobility_trans_times_force_hip(r, r, 1.0, 0.1)


def test_wall_regularisation_helpers_match_reference_semantics(oracle):
  """shift_heights uses `<=`, damping_matrix_B uses `<` (mobility/mobility.py:52-84)."""
  from rigidmultiblobswall_amd import mobility
  a = 0.25
  r = np.array([[0, 0, 1.0], [1, 0, a], [0, 1, 0.1], [2, 2, -0.05], [3, 3, a * (1 + 1e-16)]])
  re_ = mobility.shift_heights(r, a)
  B, overlap = mobility.damping_matrix_B(r, a)
  r_o, b_o, ov_o = oracle.wall_regularisation(r, a)
  assert np.array_equal(re_.reshape(-1), r_o)
  assert np.array_equal(B.diagonal(), np.repeat(b_o, 3))
  assert overlap is True and ov_o is True
  assert np.array_equal(r, np.array([[0, 0, 1.0], [1, 0, a], [0, 1, 0.1], [2, 2, -0.05], [3, 3, a * (1 + 1e-16)]]))
  B2, ov2 = mobility.damping_matrix_B(r[:2], a)
  assert ov2 is False and np.all(B2.diagonal() == 1.0)


def test_partition_covers_range_without_overlap():
  from rigidmultiblobswall_amd.distributed import partition
  for n in (0, 1, 5, 64, 1000, 24576, 10 ** 6 + 3):
    for g in (1, 2, 3, 4, 8):
      spans = [partition(n, g, r) for r in range(g)]
      assert spans[0][0] == 0 and spans[-1][1] == n
      for (b0, e0, blk), (b1, e1, _) in zip(spans, spans[1:]):
        assert e0 == b1 and e0 - b0 <= blk
      full = [e - b for b, e, _ in spans]
      # all blocks before the first short one are full (all-gather layout relies on it)
      short = [i for i, c in enumerate(full) if c < spans[0][2]]
      if short:
        assert all(c == 0 for c in full[short[0] + 1:])


def test_generated_single_precision_header_is_current():
  """csrc/pair_blocks32.h is generated from pair_blocks.h (tools/gen_pair_blocks32.py); a stale copy would make the
  single-precision mode compute with an older algebra than the double-precision one."""
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  rc = subprocess.call([sys.executable, os.path.join(root, "tools", "gen_pair_blocks32.py"), "--check"])
  assert rc == 0, "run `python tools/gen_pair_blocks32.py` after editing csrc/pair_blocks.h"


def test_multi_engine_and_device_selection_fail_loudly_without_a_gpu(monkeypatch):
  """rmb_multi_create / mobility.set_devices never fall back: no device -> error; the device list comes from
  set_devices, RMB_DEVICES, RMB_DEVICE, in that order."""
  import torch
  from rigidmultiblobswall_amd import _lib, mobility
  monkeypatch.delenv("RMB_DEVICES", raising=False)
  monkeypatch.delenv("RMB_DEVICE", raising=False)
  assert mobility.devices() == [0]
  monkeypatch.setenv("RMB_DEVICE", "3")
  assert mobility.devices() == [3]
  monkeypatch.setenv("RMB_DEVICES", "0, 1,2;3")
  assert mobility.devices() == [0, 1, 2, 3]
  assert mobility.active_devices(mobility.multi_min_blobs) == [0, 1, 2, 3] and mobility.active_devices(10) == [0]
  if torch.cuda.is_available():
    pytest.skip("GPU present")
  lib = _lib.load()
  h = ctypes.c_void_p()
  devs = (ctypes.c_int * 2)(0, 1)
  assert lib.rmb_multi_create(devs, 2, ctypes.byref(h)) == -4     # RMB_ERR_NO_DEVICE
  assert lib.rmb_multi_create(devs, 0, ctypes.byref(h)) == -1
  assert lib.rmb_multi_matvec(None, 0, 0, None, None, 1.0, None) == -1
  from rigidmultiblobswall_amd.multi import MultiContext
  with pytest.raises(_lib.RmbError):
    MultiContext([0, 1])


def test_library_tridiagonal_eigen_solver_matches_lapack():
  """rmb_lanczos_noise_coefficients (host function of the native Lanczos loop, csrc/rmb_gmres.hip: QL sweeps with implicit
  shifts) against numpy's eigh on the matrices the loop meets: Lanczos tridiagonals of SPD operators of every size the
  workspace allows, plus the awkward ones -- k = 1, decoupled blocks (zero off-diagonals), clustered and repeated
  eigenvalues, a slightly indefinite matrix (negative eigenvalues are clipped as stochastic.py does)."""
  import ctypes
  from rigidmultiblobswall_amd import _lib
  from rigidmultiblobswall_amd.stochastic import _noise_coefficients
  lib = _lib.load()
  dp = ctypes.POINTER(ctypes.c_double)

  def native(h_diag, h_sup, scale):
    k = len(h_diag)
    d = np.ascontiguousarray(h_diag, dtype=np.float64)
    e = np.ascontiguousarray(np.concatenate([h_sup, [0.0]])[:max(k, 1)], dtype=np.float64)
    out = np.empty(k)
    _lib.check(lib.rmb_lanczos_noise_coefficients(k, d.ctypes.data_as(dp), e.ctypes.data_as(dp), float(scale), out.ctypes.data_as(dp)))
    return out

  rng = np.random.RandomState(4)
  cases = []
  for n in (1, 2, 3, 7, 20, 48, 120, 254):
    # a real Lanczos run on an SPD matrix with a wide spectrum
    m = max(n, 4) + 6
    Q, _ = np.linalg.qr(rng.randn(m, m))
    A = (Q * np.logspace(-4, 1, m)) @ Q.T
    v = rng.randn(m); v /= np.linalg.norm(v)
    V, hd, hs = [v], [], []
    for i in range(n):
      w = A @ V[i] - (hs[i - 1] * V[i - 1] if i else 0.0)
      hd.append(float(w @ V[i])); w = w - hd[-1] * V[i]
      for u in V: w = w - (w @ u) * u
      hs.append(float(np.linalg.norm(w)))
      if hs[-1] < 1e-13 or len(V) == m: break
      V.append(w / hs[-1])
    cases.append((hd, hs[:len(hd)]))
  cases.append(([2.0, 3.0, 5.0, 7.0], [0.0, 0.0, 0.0, 0.0]))                        # diagonal
  cases.append(([1.0, 1.0, 1.0, 1.0, 1.0], [0.5, 0.0, 0.5, 1e-9, 0.0]))             # decoupled blocks, repeated eigenvalues
  cases.append(([1.0] * 30, [1e-7] * 30))                                            # one tight cluster
  cases.append(([1.0, -0.01, 2.0], [0.1, 0.2, 0.0]))                                 # indefinite: clipped
  for hd, hs in cases:
    k = len(hd)
    got = native(hd, hs[:max(k - 1, 0)], 1.7)
    ref = _noise_coefficients(hd, list(hs) + [0.0], k, 1.7)
    assert np.linalg.norm(got - ref) <= 1e-13 * max(np.linalg.norm(ref), 1e-300) + 1e-15, (k, np.linalg.norm(got - ref), np.linalg.norm(ref))
  with pytest.raises(_lib.RmbError):
    native([], [], 1.0)
