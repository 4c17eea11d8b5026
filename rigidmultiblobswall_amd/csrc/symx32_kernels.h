// symx32_kernels.h -- single-precision twins of the symmetric multi-block operations (gfx950, fp32 VALU).
//
// WHAT: context option "precision" = 32 (the reference's `precision = 'single'` build, mobility_pycuda.py:7-19) for the
// products besides tt: tr / rt / rr, the fused row M_tt f + M_tr tau (K11 / K12), the 6N grand mobility and the force
// column [M_tt; M_rt] f -- everything the single-blob roller steppers apply per step -- one block on k = 2..4
// vectors (lockstep solves) and the free-surface product, with open boundaries.
//
// HOW: symx_kernel's skeleton (symx_kernels.h: tile pairs, rotation, static balanced schedule, pair shards, fp64 global
// accumulators, fp64 finalize with self terms / B-damping / prefactor) with the pair arithmetic of pair_blocks32.h (the
// fp64 algebra regenerated in float).  As in sym32_kernels.h: tile J as float planes in LDS, at most 64 pair
// contributions per blob summed in fp32 before they are added in fp64 (ds_add_f64 / global_atomic_add_f64: ds_add_f32
// is an order of magnitude slower on this chip), so the error is that of the pair arithmetic, ~1e-6 relative.
#pragma once
#include "pair_blocks32.h"
#include "symx_kernels.h"
#include "sym32_kernels.h"

namespace rmb {

template <int KIND>
struct OpSingle32 {
  static constexpr int NIN = 1, NOUT = 1;
  template <bool WALL>
  static __device__ __forceinline__ void pair(const f32::PairConsts& k, float dx, float dy, float dz, float zi, float zj,
                                              const float* vi, const float* vj, float* ui, float* t) {
    const f32::Geom g = f32::make_geom<WALL>(dx, dy, dz, zi, zj);
    if constexpr (KIND == KIND_TT) { const f32::TTc c = f32::tt_coeffs<WALL>(k, g, zi, zj); f32::tt_apply<WALL, false>(c, g, vi, vj, ui, t); }
    if constexpr (KIND == KIND_RR) { const f32::RRc c = f32::rr_coeffs<WALL>(k, g); f32::rr_apply<WALL, false>(c, g, vi, vj, ui, t); }
    if constexpr (KIND == KIND_TR) { const f32::CPc C = f32::cpl_coeffs<WALL>(k, g, zi, zj); f32::tr_apply<WALL, false>(C, g, vi, vj, ui, t); }
    if constexpr (KIND == KIND_RT) { const f32::CPc C = f32::cpl_coeffs<WALL>(k, g, zi, zj); f32::rt_apply<WALL, false>(C, g, vi, vj, ui, t); }
  }
};

struct OpFusedRow32 {      // u = M_tt f + M_tr tau
  static constexpr int NIN = 2, NOUT = 1;
  template <bool WALL>
  static __device__ __forceinline__ void pair(const f32::PairConsts& k, float dx, float dy, float dz, float zi, float zj,
                                              const float* vi, const float* vj, float* ui, float* t) {
    const f32::Geom g = f32::make_geom<WALL>(dx, dy, dz, zi, zj);
    const f32::Rpy p = f32::rpy_coeffs<true, true, false>(k, g);
    const f32::TTc a = f32::tt_block<WALL>(k, g, zi, zj, p.cF, p.cD);
    f32::tt_apply<WALL, false>(a, g, vi, vj, ui, t);
    const f32::CPc C = f32::cpl_block<WALL>(k, g, zi, zj, p.c);
    f32::tr_apply<WALL, true>(C, g, vi + 3, vj + 3, ui, t);
  }
};

struct OpGrand32 {         // [u; w] = [[M_tt, M_tr], [M_rt, M_rr]] [f; tau]
  static constexpr int NIN = 2, NOUT = 2;
  template <bool WALL>
  static __device__ __forceinline__ void pair(const f32::PairConsts& k, float dx, float dy, float dz, float zi, float zj,
                                              const float* vi, const float* vj, float* ui, float* t) {
    const f32::Geom g = f32::make_geom<WALL>(dx, dy, dz, zi, zj);
    const f32::Rpy p = f32::rpy_coeffs<true, true, true>(k, g);
    const f32::TTc a = f32::tt_block<WALL>(k, g, zi, zj, p.cF, p.cD);
    f32::tt_apply<WALL, false>(a, g, vi, vj, ui, t);
    const f32::CPc C = f32::cpl_block<WALL>(k, g, zi, zj, p.c);
    f32::tr_apply<WALL, true>(C, g, vi + 3, vj + 3, ui, t);
    f32::rt_apply<WALL, false>(C, g, vi, vj, ui + 3, t + 3);
    const f32::RRc b = f32::rr_block<WALL>(k, g, p.rF, p.rD);
    f32::rr_apply<WALL, true>(b, g, vi + 3, vj + 3, ui + 3, t + 3);
  }
};

struct OpColumnF32 {       // [u; w] = [M_tt; M_rt] f
  static constexpr int NIN = 1, NOUT = 2;
  template <bool WALL>
  static __device__ __forceinline__ void pair(const f32::PairConsts& k, float dx, float dy, float dz, float zi, float zj,
                                              const float* vi, const float* vj, float* ui, float* t) {
    const f32::Geom g = f32::make_geom<WALL>(dx, dy, dz, zi, zj);
    const f32::Rpy p = f32::rpy_coeffs<true, true, false>(k, g);
    const f32::TTc a = f32::tt_block<WALL>(k, g, zi, zj, p.cF, p.cD);
    f32::tt_apply<WALL, false>(a, g, vi, vj, ui, t);
    const f32::CPc C = f32::cpl_block<WALL>(k, g, zi, zj, p.c);
    f32::rt_apply<WALL, false>(C, g, vi, vj, ui + 3, t + 3);
  }
};

template <int KIND, int K>
struct OpKindK32 {         // one block applied to K vectors (lockstep solves): the coefficients are built once
  static constexpr int NIN = K, NOUT = K;
  template <bool WALL>
  static __device__ __forceinline__ void pair(const f32::PairConsts& k, float dx, float dy, float dz, float zi, float zj,
                                              const float* vi, const float* vj, float* ui, float* t) {
    const f32::Geom g = f32::make_geom<WALL>(dx, dy, dz, zi, zj);
    if constexpr (KIND == KIND_TT) {
      const f32::TTc a = f32::tt_coeffs<WALL>(k, g, zi, zj);
#pragma unroll
      for (int v = 0; v < K; ++v) f32::tt_apply<WALL, false>(a, g, vi + 3 * v, vj + 3 * v, ui + 3 * v, t + 3 * v);
    } else if constexpr (KIND == KIND_RR) {
      const f32::RRc b = f32::rr_coeffs<WALL>(k, g);
#pragma unroll
      for (int v = 0; v < K; ++v) f32::rr_apply<WALL, false>(b, g, vi + 3 * v, vj + 3 * v, ui + 3 * v, t + 3 * v);
    } else {
      const f32::CPc C = f32::cpl_coeffs<WALL>(k, g, zi, zj);
#pragma unroll
      for (int v = 0; v < K; ++v) {
        if constexpr (KIND == KIND_TR) f32::tr_apply<WALL, false>(C, g, vi + 3 * v, vj + 3 * v, ui + 3 * v, t + 3 * v);
        else                           f32::rt_apply<WALL, false>(C, g, vi + 3 * v, vj + 3 * v, ui + 3 * v, t + 3 * v);
      }
    }
  }
};

// Free (stress-free) surface at z = 0 (OpFreeSurface of symx_kernels.h in float; mobility_numba.py:1846-1925, the
// reference's single-precision build covers it too, mobility_pycuda.py:1974): RPY(d) + RPY(R) P with the image block
// reciprocal, both directions from one set of coefficients.  Raw heights (set_positions with wall = 0).
struct OpFreeSurface32 {
  static constexpr int NIN = 1, NOUT = 1;
  template <bool WALL>
  static __device__ __forceinline__ void pair(const f32::PairConsts& k, float dx, float dy, float dz, float zi, float zj,
                                              const float* vi, const float* vj, float* ui, float* t) {
    const f32::Geom g = f32::make_geom<true>(dx, dy, dz, zi, zj);
    const f32::TTc a = f32::tt_coeffs<false>(k, g, zi, zj);
    f32::tt_apply<false, false>(a, g, vi, vj, ui, t);
    float cF, cD;
    f32::rpy_tt_coeffs(k, __builtin_fmaf(g.Rz, g.Rz, g.rho2), g.iR, g.iR2, cF, cD);
    const float sj = cD * __builtin_fmaf(-g.Rz, vj[2], __builtin_fmaf(dy, vj[1], dx * vj[0]));
    const float si = cD * __builtin_fmaf(g.Rz, vi[2], __builtin_fmaf(dy, vi[1], dx * vi[0]));
    ui[0] = __builtin_fmaf(cF, vj[0], ui[0]); ui[0] = __builtin_fmaf(sj, dx, ui[0]);
    ui[1] = __builtin_fmaf(cF, vj[1], ui[1]); ui[1] = __builtin_fmaf(sj, dy, ui[1]);
    ui[2] = __builtin_fmaf(-cF, vj[2], ui[2]); ui[2] = __builtin_fmaf(sj, g.Rz, ui[2]);
    t[0] = __builtin_fmaf(si, dx, __builtin_fmaf(cF, vi[0], t[0]));
    t[1] = __builtin_fmaf(si, dy, __builtin_fmaf(cF, vi[1], t[1]));
    t[2] = __builtin_fmaf(-si, g.Rz, __builtin_fmaf(-cF, vi[2], t[2]));
  }
};

// Translation mobility of blobs with DIFFERENT radii, sources == targets (OpRadiiTT of symx_kernels.h in float; the
// reference's single-precision build covers K13 too, mobility_pycuda.py:1841-2067 under `typedef float real`).
// vi[3] / vj[3] = radius of the blob (one extra LDS plane).
struct STc32 { float C1, C2, alpha, beta, gamma, delta, eps, rz; };

template <bool WALL>
__device__ __forceinline__ STc32 st_coeffs32(float dx, float dy, float dz, float x3, float y3, float at, float as) {
  STc32 c;
  const float rho2 = __builtin_fmaf(dy, dy, dx * dx);
  const float r2 = __builtin_fmaf(dz, dz, rho2);
  const float a2 = at * at, b2 = as * as, s = a2 + b2;
  const float ir = __builtin_amdgcn_rsqf(r2);
  const float ir2 = ir * ir;
  c.C1 = __builtin_fmaf(s * (1.0f / 3.0f), ir2, 1.0f) * ir;
  c.C2 = __builtin_fmaf(-s, ir2, 1.0f) * ir2 * ir;
  const float sum = at + as;
  if (__builtin_expect(__any(!(r2 > sum * sum)), 0)) {
    // overlapping blobs (rare): Zuk et al. regimes 2 and 3
    const float r = (r2 > 0.0f) ? __builtin_sqrtf(r2) : 0.0f;
    const float dm = (as - at) * (as - at);
    const float r3 = r2 * r;
    const float t = dm + 3.0f * r2, q = dm - r2;
    const float pre = (4.0f / 3.0f) / (as * at);
    const float C1m = ((16.0f * sum * r3 - t * t) / (32.0f * r3)) * pre;
    const float C2m = ((3.0f * q * q / (32.0f * r3)) / r2) * pre;
    const bool far = r > sum;
    const bool mid = r > __builtin_fabsf(as - at);
    c.C1 = far ? c.C1 : (mid ? C1m : (4.0f / 3.0f) / __builtin_fmaxf(at, as));
    c.C2 = far ? c.C2 : (mid ? C2m : 0.0f);
  }
  c.rz = x3 + y3;
  if constexpr (WALL) {
    const float rz = c.rz;
    const float R2 = __builtin_fmaf(rz, rz, rho2);
    const float i1 = __builtin_amdgcn_rsqf(R2);
    const float i2 = i1 * i1, i3 = i1 * i2, i5 = i3 * i2, i7 = i5 * i2, i9 = i7 * i2;
    const float ab = a2 * b2, xy = x3 * y3;
    const float m = rz * __builtin_fmaf(a2, y3, b2 * x3);
    const float ab23 = ab * (2.0f / 3.0f);
    const float rz2 = rz * rz;
    c.alpha = __builtin_fmaf(ab23, __builtin_fmaf(5.0f * rz2, i7, -i5),
                             __builtin_fmaf(-2.0f * m, i5, __builtin_fmaf(__builtin_fmaf(2.0f, xy, s * (1.0f / 3.0f)), i3, i1)));
    c.beta = __builtin_fmaf(ab23, __builtin_fmaf(-35.0f * rz2, i9, 5.0f * i7),
                            __builtin_fmaf(10.0f * m, i7, __builtin_fmaf(-__builtin_fmaf(6.0f, xy, s), i5, i3)));
    const float abz = ab * (20.0f / 3.0f) * rz * i7;
    const float dab = 2.0f * (a2 - b2) * i5;
    c.gamma = __builtin_fmaf(x3, __builtin_fmaf(-2.0f, i3, dab), abz);
    c.delta = __builtin_fmaf(y3, -__builtin_fmaf(2.0f, i3, dab), abz);
    c.eps = -__builtin_fmaf(ab * (4.0f / 3.0f), i5, __builtin_fmaf(s * (2.0f / 3.0f), i3, i1 + i1));
  } else {
    c.alpha = c.beta = c.gamma = c.delta = c.eps = 0.0f;
  }
  return c;
}

template <bool WALL>
__device__ __forceinline__ void st_apply32(const STc32& c, float dx, float dy, float dz, float sx, float gam, float del,
                                           const float* f, float* u) {
  const float pxy = __builtin_fmaf(dy, f[1], dx * f[0]);
  const float cD = c.C2 * __builtin_fmaf(dz, f[2], pxy);
  if constexpr (!WALL) {
    u[0] = __builtin_fmaf(c.C1, f[0], u[0]); u[0] = __builtin_fmaf(cD, dx, u[0]);
    u[1] = __builtin_fmaf(c.C1, f[1], u[1]); u[1] = __builtin_fmaf(cD, dy, u[1]);
    u[2] = __builtin_fmaf(c.C1, f[2], u[2]); u[2] = __builtin_fmaf(cD, dz, u[2]);
  } else {
    const float Rg = __builtin_fmaf(c.rz, f[2], -sx * pxy);
    const float cR = __builtin_fmaf(c.beta, Rg, gam * f[2]);
    const float cz = __builtin_fmaf(del, Rg, c.eps * f[2]);
    const float cFxy = c.C1 - c.alpha, cFz = c.C1 + c.alpha;
    const float cDR = __builtin_fmaf(sx, cR, cD);
    u[0] = __builtin_fmaf(cFxy, f[0], u[0]); u[0] = __builtin_fmaf(cDR, dx, u[0]);
    u[1] = __builtin_fmaf(cFxy, f[1], u[1]); u[1] = __builtin_fmaf(cDR, dy, u[1]);
    u[2] = __builtin_fmaf(cFz, f[2], u[2]); u[2] = __builtin_fmaf(cD, dz, u[2]);
    u[2] = __builtin_fmaf(cR, c.rz, u[2]); u[2] += cz;
  }
}

struct OpRadiiTT32 {
  static constexpr int NIN = 1, NOUT = 1, NEXTRA = 1;
  template <bool WALL>
  static __device__ __forceinline__ void pair(const f32::PairConsts&, float dx, float dy, float dz, float zi, float zj,
                                              const float* vi, const float* vj, float* ui, float* t) {
    const STc32 c = st_coeffs32<WALL>(dx, dy, dz, zi, zj, vi[3], vj[3]);
    st_apply32<WALL>(c, dx, dy, dz, 1.0f, c.gamma, c.delta, vj, ui);
    t[0] = 0.0f; t[1] = 0.0f; t[2] = 0.0f;
    st_apply32<WALL>(c, dx, dy, dz, -1.0f, c.delta, c.gamma, vi, t);
  }
};

// planes of a record: x, y, z heads, the NIN vectors, NEXTRA per-blob scalars (radius), then the three position tails
template <class OP> struct SymX32Lds {
  static constexpr int planes = 6 + 3 * OP::NIN + SymXExtra<OP>::value;
  static constexpr size_t bytes = (sizeof(float) * planes + sizeof(double) * 3 * OP::NOUT) * 64 * kSymWaves;
};

template <class OP, bool WALL>
__global__ __launch_bounds__(64 * kSymWaves) void symx32_kernel(const SymXArgs a, const f32::PairConsts kf) {
  constexpr int NI = OP::NIN, NO = OP::NOUT, NX = SymXExtra<OP>::value, NV = 3 * NI + NX, NP = 6 + NV;
  __shared__ float rec_all[kSymWaves][NP * 64];      // planes x, y, z (float heads), the NIN vectors (+ extras), then the position tails
  __shared__ double accj_all[kSymWaves][3 * NO * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* rec = rec_all[wave];
  double* accj = accj_all[wave];

  const long w = (a.xcd ? xcd_swizzle(blockIdx.x, gridDim.x) : (long)blockIdx.x) * kSymWaves + wave;
  // strided chunks: wave / workgroup `id` takes the step ranges id, id + n, id + 2 n, ... of `spw` steps each (one range when
  // the launch is planned that way: n spw >= the steps of the launch).  Waves that run at the same time then work on
  // NEIGHBOURING ranges whatever the size of the problem -- with the blocked unit order and the XCD-aware numbering
  // that keeps a launch's tile loads in one L2 (profiles/r4_unit_order.txt).
  for (long chunk = w;; chunk += (long)gridDim.x * kSymWaves) {
  long s = a.step_begin + chunk * a.steps_per_wave;
  if (s >= a.step_end) break;
  long s_end = s + a.steps_per_wave;
  if (s_end > a.step_end) s_end = a.step_end;
  int I = 0, J = 0;
  if (s < s_end) unit_seek(a.order, s >> 6, a.n_tiles, I, J);

  int I_cur = -1;
  long i = 0;
  bool vi_ok = false;
  float xi = 0, yi = 0, zi = 1.0f, xil = 0, yil = 0, zil = 0;
  float vi[NV], ui[3 * NO];
#pragma unroll
  for (int c = 0; c < NV; ++c) vi[c] = 0;
#pragma unroll
  for (int c = 0; c < 3 * NO; ++c) ui[c] = 0;

  auto flush_row = [&]() {
    if (!vi_ok) return;
#pragma unroll
    for (int c = 0; c < 3 * NO; ++c)
      __hip_atomic_fetch_add(&a.acc[(long)c * a.n_pad + i], (double)ui[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  while (s < s_end) {
    const int k0 = (int)(s & 63);
    const long left = s_end - s;
    const int k1 = (left < 64 - k0) ? (int)(k0 + left) : 64;
    s += k1 - k0;

    if (I != I_cur) {
      if (I_cur >= 0) flush_row();
      I_cur = I;
      i = 64L * I + lane;
      vi_ok = i < a.n;
      xi = 1e18f; yi = 1e18f; zi = 1.0f; xil = 0; yil = 0; zil = 0;   // padding: far away, 1/r^2 stays finite in float
#pragma unroll
      for (int c = 0; c < NV; ++c) vi[c] = 0;
      if (vi_ok) {
        const double4 p = a.pos[i];
        xi = (float)p.x; yi = (float)p.y; zi = (float)p.z;
        xil = (float)(p.x - (double)xi); yil = (float)(p.y - (double)yi); zil = (float)(p.z - (double)zi);
        if constexpr (NX > 0) vi[3 * NI] = (float)a.extra[i];
#pragma unroll
        for (int v = 0; v < NI; ++v) {
          vi[3 * v] = (float)(a.in[v][3 * i] * p.w); vi[3 * v + 1] = (float)(a.in[v][3 * i + 1] * p.w);
          vi[3 * v + 2] = a.in_plane ? 0.0f : (float)(a.in[v][3 * i + 2] * p.w);
        }
      }
#pragma unroll
      for (int c = 0; c < 3 * NO; ++c) ui[c] = 0;
    }
    {
      const long j = 64L * J + lane;
      float rd[NP];
#pragma unroll
      for (int c = 0; c < NP; ++c) rd[c] = 0;
      rd[0] = -1e18f; rd[1] = -1e18f; rd[2] = 1.0f;
      if (j < a.n) {
        const double4 p = a.pos[j];
        rd[0] = (float)p.x; rd[1] = (float)p.y; rd[2] = (float)p.z;
        rd[3 + NV] = (float)(p.x - (double)rd[0]); rd[4 + NV] = (float)(p.y - (double)rd[1]); rd[5 + NV] = (float)(p.z - (double)rd[2]);
        if constexpr (NX > 0) rd[3 + 3 * NI] = (float)a.extra[j];
#pragma unroll
        for (int v = 0; v < NI; ++v) {
          rd[3 + 3 * v] = (float)(a.in[v][3 * j] * p.w); rd[4 + 3 * v] = (float)(a.in[v][3 * j + 1] * p.w);
          rd[5 + 3 * v] = a.in_plane ? 0.0f : (float)(a.in[v][3 * j + 2] * p.w);
        }
      }
#pragma unroll
      for (int c = 0; c < NP; ++c) rec[c * 64 + lane] = rd[c];
#pragma unroll
      for (int c = 0; c < 3 * NO; ++c) accj[c * 64 + lane] = 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const bool diag = I == J;
    // diagonal units: every ordered pair of the tile once, forward only; step 0 (the blob itself) is the self term
    for (int k = (diag && k0 < 1) ? 1 : k0; k < k1; ++k) {
      const int jj = (lane + k) & 63;
      float rd[NP];
#pragma unroll
      for (int c = 0; c < NP; ++c) rd[c] = rec[c * 64 + jj];
      float t[3 * NO];
      // head / tail split of the fp64 positions (sym32_kernels.h): the error of d does not grow with the domain size
      const v2f xyj = {rd[0], rd[1]}, xyjl = {rd[3 + NV], rd[4 + NV]}, xyi = {xi, yi}, xyil = {xil, yil};
      const v2f dxy = (xyi - xyj) + (xyil - xyjl);
      const float dz = (zi - rd[2]) + (zil - rd[5 + NV]);
      OP::template pair<WALL>(kf, dxy.x, dxy.y, dz, zi, rd[2], vi, rd + 3, ui, t);
      if (!diag) {   // wave-uniform
#pragma unroll
        for (int c = 0; c < 3 * NO; ++c)
          __hip_atomic_fetch_add(&accj[c * 64 + jj], (double)t[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
    }
    if (!diag) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const long j = 64L * J + lane;
      if (j < a.n) {
#pragma unroll
        for (int c = 0; c < 3 * NO; ++c)
          __hip_atomic_fetch_add(&a.acc[(long)c * a.n_pad + j], accj[c * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __builtin_amdgcn_wave_barrier();   // accj / rec are rewritten by the next unit
    if (k1 == 64) {
      unit_next(a.order, a.n_tiles, I, J);
    }
  }
  if (I_cur >= 0) flush_row();
  }   // chunks
}

}  // namespace rmb
