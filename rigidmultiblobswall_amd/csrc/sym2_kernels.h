// sym2_kernels.h -- symmetric tt pair sweep applied to TWO source vectors in one pass (gfx950, fp64).
//
// WHY: a Brownian step of the rigid-multiblob integrators solves several saddle-point systems with the SAME
// mobility (quaternion_integrator_multi_bodies.py:985-996: the Brownian-slip solve and the RFD solve of
// stochastic_Slip_Trapz share configuration and operator).  Running their GMRES iterations in lockstep turns two
// M.v products into one sweep over the pairs with two vectors.  Of the ~110 VALU instructions sym_kernel spends per
// unordered pair, ~62 are geometry (differences, two rsqrt, RPY coefficients, the five wall polynomials) and do not
// depend on the vector; only the ~36-instruction contraction is per vector.  Two vectors: ~146 instead of 220.
//
// Same tile-pair rotation, per-wave LDS slab, static balanced step schedule, pair-shard ranges and global SoA
// accumulators as sym_kernel (sym_kernels.h); records are 80 bytes (x, y, z, v_a, v_b: conflict-free for per-lane
// ds_read_b128), accumulators [2][3][n_pad].
#pragma once
#include "sym_kernels.h"

namespace rmb {

struct Sym2Args {
  const double4* pos;
  const double* vec_a;   // [3n]
  const double* vec_b;   // [3n]
  double* acc;           // [2][3][n_pad], zero on entry, re-zeroed by finalize
  double* out_a;         // [3n]
  double* out_b;         // [3n]
  long n, n_pad;
  int n_tiles;
  long n_units;
  int order, xcd;       // as SymArgs
  long step_begin, step_end, steps_per_wave;
  long self_begin, self_end;
  double Lx, Ly, Lz, iLx, iLy, iLz;
  double prefactor;
  PairConsts k;
};

constexpr int kSym2RecBytes = 80;

// Vector-independent part of one tt pair: the block of pair_blocks.h (built once, contracted per vector).
struct TTPair { Geom g; TTc c; };

template <bool WALL>
__device__ __forceinline__ TTPair tt_pair_coefficients(const PairConsts& k, double dx, double dy, double dz, double zi,
                                                       double zj) {
  TTPair p;
  p.g = make_geom<WALL>(dx, dy, dz, zi, zj);
  p.c = tt_coeffs<WALL>(k, p.g, zi, zj);
  return p;
}

// Both directions for one vector pair: ui += M_ij v_j,  (tx,ty,tz) = M_ij^T v_i.
template <bool WALL>
__device__ __forceinline__ void tt_pair_apply(const TTPair& p, double vix, double viy, double viz, double vjx, double vjy,
                                              double vjz, Vec3& ui, double& tx, double& ty, double& tz) {
  const double vi[3] = {vix, viy, viz}, vj[3] = {vjx, vjy, vjz};
  double u[3] = {ui.x, ui.y, ui.z}, t[3];
  tt_apply<WALL, false>(p.c, p.g, vi, vj, u, t);
  ui.x = u[0]; ui.y = u[1]; ui.z = u[2];
  tx = t[0]; ty = t[1]; tz = t[2];
}

template <bool WALL, bool PERIODIC>
__global__ __launch_bounds__(64 * kSymWaves) __attribute__((amdgpu_waves_per_eu(kSymWavesPerEu, kSymWavesPerEu))) void sym2_kernel(const Sym2Args a) {
  __shared__ double2 rec_all[kSymWaves][64 * 5];
  __shared__ double accj_all[kSymWaves][6 * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double2* rec = rec_all[wave];
  double* accj = accj_all[wave];
  const char* rec_bytes = reinterpret_cast<const char*>(rec);
  double* acc_a = a.acc;
  double* acc_b = a.acc + 3 * a.n_pad;

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
  double xi = 0, yi = 0, zi = 1.0;
  double vax = 0, vay = 0, vaz = 0, vbx = 0, vby = 0, vbz = 0;
  Vec3 ua = {0.0, 0.0, 0.0}, ub = {0.0, 0.0, 0.0};

  auto flush_row = [&]() {
    __hip_atomic_fetch_add(&acc_a[i], ua.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&acc_a[a.n_pad + i], ua.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&acc_a[2 * a.n_pad + i], ua.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&acc_b[i], ub.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&acc_b[a.n_pad + i], ub.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&acc_b[2 * a.n_pad + i], ub.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  while (s < s_end) {
    const int k0 = (int)(s & 63);
    const long left = s_end - s;
    const int k1 = (left < 64 - k0) ? (int)(k0 + left) : 64;
    s += k1 - k0;

    if (I != I_cur) {
      if (I_cur >= 0 && vi_ok) flush_row();
      I_cur = I;
      i = 64L * I + lane;
      vi_ok = i < a.n;
      xi = 1e100; yi = 1e100; zi = 1.0;
      vax = vay = vaz = vbx = vby = vbz = 0.0;
      if (vi_ok) {
        const double4 p = a.pos[i];
        xi = p.x; yi = p.y; zi = p.z;
        vax = a.vec_a[3 * i] * p.w; vay = a.vec_a[3 * i + 1] * p.w; vaz = a.vec_a[3 * i + 2] * p.w;
        vbx = a.vec_b[3 * i] * p.w; vby = a.vec_b[3 * i + 1] * p.w; vbz = a.vec_b[3 * i + 2] * p.w;
      }
      ua.x = ua.y = ua.z = 0.0;
      ub.x = ub.y = ub.z = 0.0;
    }
    {
      const long j = 64L * J + lane;
      double xj = -1e100, yj = -1e100, zj = 1.0, ax = 0, ay = 0, az = 0, bx = 0, by = 0, bz = 0;
      if (j < a.n) {
        const double4 p = a.pos[j];
        xj = p.x; yj = p.y; zj = p.z;
        ax = a.vec_a[3 * j] * p.w; ay = a.vec_a[3 * j + 1] * p.w; az = a.vec_a[3 * j + 2] * p.w;
        bx = a.vec_b[3 * j] * p.w; by = a.vec_b[3 * j + 1] * p.w; bz = a.vec_b[3 * j + 2] * p.w;
      }
      rec[lane * 5 + 0] = make_double2(xj, yj);
      rec[lane * 5 + 1] = make_double2(zj, ax);
      rec[lane * 5 + 2] = make_double2(ay, az);
      rec[lane * 5 + 3] = make_double2(bx, by);
      rec[lane * 5 + 4] = make_double2(bz, 0.0);
#pragma unroll
      for (int c = 0; c < 6; ++c) accj[c * 64 + lane] = 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const int px = PERIODIC && a.Lx > 0, py = PERIODIC && a.Ly > 0, pz = PERIODIC && a.Lz > 0;
    const bool diag = I == J;
    for (int k = diag ? ((PERIODIC || k0 > 1) ? k0 : 1) : k0; k < k1; ++k) {
      const int jj = (lane + k) & 63;
      const double2* r = reinterpret_cast<const double2*>(rec_bytes + jj * kSym2RecBytes);
      const double2 q0 = r[0], q1 = r[1], q2 = r[2], q3 = r[3], q4 = r[4];
      double dx = xi - q0.x, dy = yi - q0.y, dz = zi - q1.x;
      double tax = 0, tay = 0, taz = 0, tbx = 0, tby = 0, tbz = 0;
      if constexpr (!PERIODIC) {
        const TTPair c = tt_pair_coefficients<WALL>(a.k, dx, dy, dz, zi, q1.x);
        tt_pair_apply<WALL>(c, vax, vay, vaz, q1.y, q2.x, q2.y, ua, tax, tay, taz);
        tt_pair_apply<WALL>(c, vbx, vby, vbz, q3.x, q3.y, q4.x, ub, tbx, tby, tbz);
      } else {
        if (px) dx = wrap_nearest_pad_safe(dx, a.Lx, a.iLx);
        if (py) dy = wrap_nearest_pad_safe(dy, a.Ly, a.iLy);
        if (pz) dz = wrap_nearest_pad_safe(dz, a.Lz, a.iLz);
        for (int bx = -px; bx <= px; ++bx)
          for (int by = -py; by <= py; ++by)
            for (int bz = -pz; bz <= pz; ++bz) {
              if (diag && k == 0 && bx == 0 && by == 0 && bz == 0) continue;   // the blob itself: self term (finalize)
              const double ex = dx + bx * a.Lx, ey = dy + by * a.Ly, ez = dz + bz * a.Lz;
              const TTPair c = tt_pair_coefficients<WALL>(a.k, ex, ey, ez, zi, q1.x);
              double sx, sy, sz;
              tt_pair_apply<WALL>(c, vax, vay, vaz, q1.y, q2.x, q2.y, ua, sx, sy, sz);
              tax += sx; tay += sy; taz += sz;
              tt_pair_apply<WALL>(c, vbx, vby, vbz, q3.x, q3.y, q4.x, ub, sx, sy, sz);
              tbx += sx; tby += sy; tbz += sz;
            }
      }
      if (!diag) {   // diagonal units visit every ordered pair of the tile: forward direction only
        __hip_atomic_fetch_add(&accj[jj], tax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(&accj[64 + jj], tay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(&accj[128 + jj], taz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(&accj[192 + jj], tbx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(&accj[256 + jj], tby, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(&accj[320 + jj], tbz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
    }
    if (!diag) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const long j = 64L * J + lane;
      if (j < a.n) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          __hip_atomic_fetch_add(&acc_a[c * a.n_pad + j], accj[c * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add(&acc_b[c * a.n_pad + j], accj[(3 + c) * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (k1 == 64) {
      unit_next(a.order, a.n_tiles, I, J);
    }
  }
  if (I_cur >= 0 && vi_ok) flush_row();
  }   // chunks
}

template <bool WALL>
__global__ __launch_bounds__(256) void sym2_finalize_kernel(const Sym2Args a) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const double4 p = a.pos[i];
  const double b = p.w;
  const double sc = a.prefactor * b;
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    double* acc = a.acc + (long)v * 3 * a.n_pad;
    const double* vec = v ? a.vec_b : a.vec_a;
    double* out = v ? a.out_b : a.out_a;
    Vec3 u = {acc[i], acc[a.n_pad + i], acc[2 * a.n_pad + i]};
    acc[i] = 0.0; acc[a.n_pad + i] = 0.0; acc[2 * a.n_pad + i] = 0.0;
    if (i >= a.self_begin && i < a.self_end)
      self_term<KIND_TT, WALL>(a.k, p.z, vec[3 * i] * b, vec[3 * i + 1] * b, vec[3 * i + 2] * b, 0, 0, 0, u);
    out[3 * i] = u.x * sc; out[3 * i + 1] = u.y * sc; out[3 * i + 2] = u.z * sc;
  }
}

}  // namespace rmb
