// exp_two_targets.hip -- would "two target blobs per lane" take the k-vector passes off the LDS pipe?  
//
// The symmetric kernels keep one target blob per lane in registers and walk the 64 source records of a staged tile in
// LDS: per pair one record read (3 + 3 K doubles) and 3 K ds_add_f64 of the transposed contribution.  With K >= 3 vectors
// the LDS pipe, not the VALU, bounds the pass (profiles/r4_coop_kernel_ab.txt).  Here a lane keeps TWO target blobs (rows
// 2p and 2p + 1 of the tile grid) against the same staged tile J: one record read and one set of ds_add_f64 serve two
// pairs.  This is a throughput experiment on the interior of the problem only (tile pairs with J >= 2p + 2, all
// off-diagonal; no diagonal units, no partial ranges, no finalize): the same work list is run by the ONE-target loop
// (each (row, J) unit staged, walked and flushed on its own -- the shape of symx_kernel) and by the TWO-target loop, wall
// tt x K, and the raw accumulators are compared.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I rigidmultiblobswall_amd/csrc \
//         -o /tmp/exp_two_targets tools/experiments/exp_two_targets.hip && /tmp/exp_two_targets [N]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "symx_kernels.h"

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Args {
  const double4* pos;
  const double* in[4];
  double* acc;          // [K][3][n_pad]
  long n, n_pad;
  const int2* work;     // (row pair p, tile J)
  long n_work;
  int work_per_wave;
  rmb::PairConsts k;
};

template <int K, bool TWO>
__global__ __launch_bounds__(256) void bench_kernel(const Args a) {
  using OP = rmb::OpKindK<rmb::KIND_TT, K>;
  constexpr int RD = 3 + 3 * K;                 // doubles per record
  constexpr int RD2 = (RD + 1) / 2;
  __shared__ double2 rec_all[4][64 * RD2];
  __shared__ double accj_all[4][3 * K * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double2* rec = rec_all[wave];
  double* accj = accj_all[wave];
  const long w = (long)blockIdx.x * 4 + wave;
  long e0 = w * a.work_per_wave, e1 = e0 + a.work_per_wave;
  if (e1 > a.n_work) e1 = a.n_work;
  int p_cur = -1;
  double x[2] = {0, 0}, y[2] = {0, 0}, z[2] = {1, 1};
  double v[2][3 * K], u[2][3 * K];
  long irow[2] = {0, 0};
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int c = 0; c < 3 * K; ++c) { v[s][c] = 0.0; u[s][c] = 0.0; }

  auto flush_rows = [&]() {
#pragma unroll
    for (int s = 0; s < 2; ++s)
      if (irow[s] < a.n) {
#pragma unroll
        for (int c = 0; c < 3 * K; ++c)
          __hip_atomic_fetch_add(&a.acc[(long)c * a.n_pad + irow[s]], u[s][c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
  };
  auto stage = [&](int J) {
    const long j = 64L * J + lane;
    double rd[2 * RD2];
#pragma unroll
    for (int c = 0; c < 2 * RD2; ++c) rd[c] = 0.0;
    rd[0] = -1e100; rd[1] = -1e100; rd[2] = 1.0;
    if (j < a.n) {
      const double4 p = a.pos[j];
      rd[0] = p.x; rd[1] = p.y; rd[2] = p.z;
#pragma unroll
      for (int q = 0; q < K; ++q) { rd[3 + 3 * q] = a.in[q][3 * j] * p.w; rd[4 + 3 * q] = a.in[q][3 * j + 1] * p.w; rd[5 + 3 * q] = a.in[q][3 * j + 2] * p.w; }
    }
#pragma unroll
    for (int c = 0; c < RD2; ++c) rec[lane * RD2 + c] = make_double2(rd[2 * c], rd[2 * c + 1]);
#pragma unroll
    for (int c = 0; c < 3 * K; ++c) accj[c * 64 + lane] = 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto flush_tile = [&](int J) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const long j = 64L * J + lane;
    if (j < a.n) {
#pragma unroll
      for (int c = 0; c < 3 * K; ++c)
        __hip_atomic_fetch_add(&a.acc[(long)c * a.n_pad + j], accj[c * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_wave_barrier();
  };
  const char* rec_bytes = reinterpret_cast<const char*>(rec);

  for (long e = e0; e < e1; ++e) {
    const int2 wk = a.work[e];
    const int p = wk.x, J = wk.y;
    if (p != p_cur) {
      if (p_cur >= 0) flush_rows();
      p_cur = p;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        irow[s] = 64L * (2 * p + s) + lane;
        x[s] = 1e100; y[s] = 1e100; z[s] = 1.0;
#pragma unroll
        for (int c = 0; c < 3 * K; ++c) { v[s][c] = 0.0; u[s][c] = 0.0; }
        if (irow[s] < a.n) {
          const double4 q = a.pos[irow[s]];
          x[s] = q.x; y[s] = q.y; z[s] = q.z;
#pragma unroll
          for (int c = 0; c < K; ++c) {
            v[s][3 * c] = a.in[c][3 * irow[s]] * q.w; v[s][3 * c + 1] = a.in[c][3 * irow[s] + 1] * q.w; v[s][3 * c + 2] = a.in[c][3 * irow[s] + 2] * q.w;
          }
        }
      }
    }
    if constexpr (TWO) {
      stage(J);
      for (int k = 0; k < 64; ++k) {
        const int jj = (lane + k) & 63;
        const double2* r = reinterpret_cast<const double2*>(rec_bytes + jj * (RD2 * 16));
        double rd[2 * RD2];
#pragma unroll
        for (int c = 0; c < RD2; ++c) { const double2 q = r[c]; rd[2 * c] = q.x; rd[2 * c + 1] = q.y; }
        double t0[3 * K], t1[3 * K];
        OP::template pair<true>(a.k, x[0] - rd[0], y[0] - rd[1], z[0] - rd[2], z[0], rd[2], v[0], rd + 3, u[0], t0);
        OP::template pair<true>(a.k, x[1] - rd[0], y[1] - rd[1], z[1] - rd[2], z[1], rd[2], v[1], rd + 3, u[1], t1);
#pragma unroll
        for (int c = 0; c < 3 * K; ++c)
          __hip_atomic_fetch_add(&accj[c * 64 + jj], t0[c] + t1[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
      flush_tile(J);
    } else {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        stage(J);
        for (int k = 0; k < 64; ++k) {
          const int jj = (lane + k) & 63;
          const double2* r = reinterpret_cast<const double2*>(rec_bytes + jj * (RD2 * 16));
          double rd[2 * RD2];
#pragma unroll
          for (int c = 0; c < RD2; ++c) { const double2 q = r[c]; rd[2 * c] = q.x; rd[2 * c + 1] = q.y; }
          double t[3 * K];
          OP::template pair<true>(a.k, x[s] - rd[0], y[s] - rd[1], z[s] - rd[2], z[s], rd[2], v[s], rd + 3, u[s], t);
#pragma unroll
          for (int c = 0; c < 3 * K; ++c)
            __hip_atomic_fetch_add(&accj[c * 64 + jj], t[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        flush_tile(J);
      }
    }
  }
  if (p_cur >= 0) flush_rows();
}

static rmb::PairConsts consts(double a) {
  rmb::PairConsts k;
  const double a2 = a * a, a3 = a2 * a, a4 = a2 * a2, a6 = a3 * a3;
  k.a2 = a2; k.four_a2 = 4.0 * a2; k.tt_k1 = 2.0 * a2 / 3.0; k.tt_k2 = 2.0 * a2; k.tt_k3 = a2 / 3.0;
  k.tt_n0 = 4.0 / (3.0 * a); k.tt_n1 = 3.0 / (8.0 * a2); k.tt_n2 = 1.0 / (8.0 * a2);
  k.rr_m0 = 1.0 / a3; k.rr_m1 = 27.0 / (32.0 * a4); k.rr_m2 = 5.0 / (64.0 * a6); k.rr_m3 = 9.0 / (32.0 * a4); k.rr_m4 = 3.0 / (64.0 * a6);
  k.c_q0 = 1.0 / (2.0 * a3); k.c_q1 = 3.0 / (16.0 * a4); k.m7 = -7.0; k.m6 = -6.0; k.c15 = 1.5; k.c30 = 30.0;
  return k;
}

template <int K, bool TWO>
static double run(const Args& a, unsigned blocks, int reps, int* regs) {
  hipFuncAttributes fa;
  CHK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&bench_kernel<K, TWO>)));
  *regs = fa.numRegs;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((bench_kernel<K, TWO>), dim3(blocks), dim3(256), 0, 0, a);
  CHK(hipDeviceSynchronize());
  CHK(hipMemset(a.acc, 0, sizeof(double) * 3 * K * a.n_pad));
  CHK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((bench_kernel<K, TWO>), dim3(blocks), dim3(256), 0, 0, a);
  CHK(hipEventRecord(e1, 0));
  CHK(hipEventSynchronize(e1));
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

template <int K>
static void compare(Args a, unsigned blocks, double pairs, int reps) {
  const size_t nacc = (size_t)3 * K * a.n_pad;
  std::vector<double> r1(nacc), r2(nacc);
  int regs1 = 0, regs2 = 0;
  const double t1 = run<K, false>(a, blocks, reps, &regs1);
  CHK(hipMemcpy(r1.data(), a.acc, nacc * sizeof(double), hipMemcpyDeviceToHost));
  const double t2 = run<K, true>(a, blocks, reps, &regs2);
  CHK(hipMemcpy(r2.data(), a.acc, nacc * sizeof(double), hipMemcpyDeviceToHost));
  double num = 0, den = 0;
  for (size_t i = 0; i < nacc; ++i) { num += (r1[i] - r2[i]) * (r1[i] - r2[i]); den += r1[i] * r1[i]; }
  printf("wall tt x%d: one target per lane %9.3f ms (%3d VGPRs, %.1f Gpairs/s) | two targets per lane %9.3f ms (%3d VGPRs, %.1f Gpairs/s)  -> x%.3f   rel diff %.1e\n",
         K, t1, regs1, pairs / t1 * 1e-6, t2, regs2, pairs / t2 * 1e-6, t1 / t2, std::sqrt(num / (den > 0 ? den : 1)));
  fflush(stdout);
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 100000;
  const long tiles = (n + 63) / 64, n_pad = 64 * tiles;
  const double rad = 0.5;
  srand(7);
  auto rnd = []() { return rand() / (double)RAND_MAX; };
  const double side = rad * std::cbrt((double)n / 0.05);       // ~ the density of bench.py's clouds
  std::vector<double4> pos(n);
  for (long i = 0; i < n; ++i) pos[i] = make_double4(side * rnd(), side * rnd(), rad * 1.1 + side * 0.25 * rnd(), 1.0);
  std::vector<double> vec[4];
  for (int q = 0; q < 4; ++q) { vec[q].resize(3 * n); for (auto& t : vec[q]) t = 2.0 * rnd() - 1.0; }
  std::vector<int2> work;
  for (int p = 0; 2 * p + 1 < tiles; ++p)
    for (int J = 2 * p + 2; J < tiles; ++J) work.push_back(make_int2(p, J));
  Args a;
  double4* dpos; int2* dwork; double* dacc; double* dvec[4];
  CHK(hipMalloc(&dpos, n * sizeof(double4))); CHK(hipMemcpy(dpos, pos.data(), n * sizeof(double4), hipMemcpyHostToDevice));
  CHK(hipMalloc(&dwork, work.size() * sizeof(int2))); CHK(hipMemcpy(dwork, work.data(), work.size() * sizeof(int2), hipMemcpyHostToDevice));
  CHK(hipMalloc(&dacc, sizeof(double) * 12 * n_pad));
  for (int q = 0; q < 4; ++q) { CHK(hipMalloc(&dvec[q], 3 * n * sizeof(double))); CHK(hipMemcpy(dvec[q], vec[q].data(), 3 * n * sizeof(double), hipMemcpyHostToDevice)); a.in[q] = dvec[q]; }
  a.pos = dpos; a.acc = dacc; a.n = n; a.n_pad = n_pad; a.work = dwork; a.n_work = (long)work.size(); a.work_per_wave = 8;
  a.k = consts(rad);
  const unsigned blocks = (unsigned)((work.size() + 4 * a.work_per_wave - 1) / (4 * a.work_per_wave));
  const double pairs = (double)work.size() * 2.0 * 64.0 * 64.0;
  printf("N = %ld blobs, %ld tiles, %zu double units (%.3e pair evaluations), %u workgroups\n", n, tiles, work.size(), pairs, blocks);
  const int reps = n <= 20000 ? 20 : 3;
  compare<1>(a, blocks, pairs, reps);
  compare<2>(a, blocks, pairs, reps);
  compare<3>(a, blocks, pairs, reps);
  compare<4>(a, blocks, pairs, reps);
  return 0;
}
