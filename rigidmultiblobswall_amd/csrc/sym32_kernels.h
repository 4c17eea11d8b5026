// sym32_kernels.h -- single-precision twin of the symmetric translation-translation sweep (gfx950, fp32 VALU).
//
// WHAT: the reference's GPU module has a precision switch (`precision = 'single' | 'double'` + `typedef float real`,
// mobility/mobility_pycuda.py:7-19): the same kernels compiled in fp32.  This is that mode for the product that
// dominates every caller, u = M_tt f (wall / no wall, open boundaries): context option "precision" = 32.  Everything
// else keeps running in fp64 whatever the option says.
//
// HOW: the schedule, the tile pairs, the rotation and the global accumulators are sym_kernel's (sym_kernels.h); the pair
// arithmetic is the closed-form block of pair_blocks.h (F, P, Q3, Q4, Szz) regenerated in float (pair_blocks32.h):
//   * positions and vectors are converted once per tile; each position travels as a float head and a float tail
//     (x = xh + xl to ~1e-15), and a separation is (xh_i - xh_j) + (xl_i - xl_j): exact head difference for near pairs,
//     so -- unlike the reference's float build, which subtracts rounded coordinates -- the error of d does not grow
//     with the size of the domain (6 extra fp32 instructions per pair);
//   * v_rsq_f32 is accurate to 1 ulp, so the two inverse square roots need no correction step (fp64: 5 instructions each);
//   * tile J sits in the wave's LDS slab as six float planes (conflict-free ds_read_b32 under the rotation); the
//     transposed contribution is converted and added to three fp64 planes with ds_add_f64 -- ds_add_f32 turned out
//     ~10x slower than ds_add_f64 on gfx950 (PMC: 45 vs 8 busy cycles per LDS instruction, no bank conflicts), which
//     made the first version of this kernel 5x SLOWER than the fp64 one;
//   * a wave's partial sums (64 pairs per blob and unit) are flushed into the SAME fp64 global accumulators the fp64
//     kernel uses, and the self term, the B-damping and the prefactor are applied by the same fp64 finalize kernel --
//     so the single-precision error is that of the pair arithmetic (~1e-6 relative), not of a long fp32 sum.
// 83 VALU instructions per unordered pair (wall tt; 3 of them conversions for the fp64 accumulation); fp32 issues 1.6x
// faster than fp64 on this chip (profiles/r1_ubench_fp64_issue_rates.txt: 785 vs 486 G wave-instr/s): 1.5-1.6x the fp64 kernel.
#pragma once
#include "pair_blocks32.h"
#include "sym_kernels.h"

namespace rmb {

typedef float v2f __attribute__((ext_vector_type(2)));

// Both directions of one pair in float: ui += M_ij vj, t = M_ij^T vi -- the generated single-precision algebra
// (pair_blocks32.h = pair_blocks.h in float, tools/gen_pair_blocks32.py)
template <bool WALL>
__device__ __forceinline__ void pair_tt_sym32(const f32::PairConsts& k, float dx, float dy, float dz, float zi, float zj,
                                              const float* vi, const float* vj, float* ui, float* t) {
  const f32::Geom g = f32::make_geom<WALL>(dx, dy, dz, zi, zj);
  const f32::TTc c = f32::tt_coeffs<WALL>(k, g, zi, zj);
  f32::tt_apply<WALL, false>(c, g, vi, vj, ui, t);
}

template <bool WALL>
__global__ __launch_bounds__(64 * kSymWaves) void sym32_tt_kernel(const SymArgs a, const f32::PairConsts kf) {
  __shared__ float rec_all[kSymWaves][9 * 64];     // planes x, y, z (float heads), their tails, vx, vy, vz of tile J
  __shared__ double accj_all[kSymWaves][3 * 64];   // fp64: ds_add_f32 is ~10x slower than ds_add_f64 on this chip (see header)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* rec = rec_all[wave];
  double* accj = accj_all[wave];

  // the static, exactly balanced schedule of sym_kernel
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
  float xi = 0, yi = 0, zi = 1.0f, xil = 0, yil = 0, zil = 0, vi[3] = {0, 0, 0};
  float ui[3] = {0, 0, 0};

  while (s < s_end) {
    const int k0 = (int)(s & 63);
    const long left = s_end - s;
    const int k1 = (left < 64 - k0) ? (int)(k0 + left) : 64;
    s += k1 - k0;

    if (I != I_cur) {
      if (I_cur >= 0 && vi_ok) {   // flush the previous row: float partial sums into the fp64 accumulators
        __hip_atomic_fetch_add(&a.acc[i], (double)ui[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&a.acc[a.n_pad + i], (double)ui[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&a.acc[2 * a.n_pad + i], (double)ui[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      I_cur = I;
      i = 64L * I + lane;
      vi_ok = i < a.n;
      xi = 1e18f; yi = 1e18f; zi = 1.0f; xil = 0; yil = 0; zil = 0; vi[0] = 0; vi[1] = 0; vi[2] = 0;   // padding: far away, 1/r^2 finite in float
      if (vi_ok) {
        const double4 p = a.pos[i];
        xi = (float)p.x; yi = (float)p.y; zi = (float)p.z;
        xil = (float)(p.x - (double)xi); yil = (float)(p.y - (double)yi); zil = (float)(p.z - (double)zi);
        vi[0] = (float)(a.vec[3 * i] * p.w); vi[1] = (float)(a.vec[3 * i + 1] * p.w); vi[2] = (float)(a.vec[3 * i + 2] * p.w);
      }
      ui[0] = 0; ui[1] = 0; ui[2] = 0;
    }
    {
      const long j = 64L * J + lane;
      float xj = -1e18f, yj = -1e18f, zj = 1.0f, xjl = 0, yjl = 0, zjl = 0, vjx = 0, vjy = 0, vjz = 0;
      if (j < a.n) {
        const double4 p = a.pos[j];
        xj = (float)p.x; yj = (float)p.y; zj = (float)p.z;
        xjl = (float)(p.x - (double)xj); yjl = (float)(p.y - (double)yj); zjl = (float)(p.z - (double)zj);
        vjx = (float)(a.vec[3 * j] * p.w); vjy = (float)(a.vec[3 * j + 1] * p.w); vjz = (float)(a.vec[3 * j + 2] * p.w);
      }
      rec[lane] = xj; rec[64 + lane] = yj; rec[128 + lane] = zj;
      rec[192 + lane] = vjx; rec[256 + lane] = vjy; rec[320 + lane] = vjz;
      rec[384 + lane] = xjl; rec[448 + lane] = yjl; rec[512 + lane] = zjl;
      accj[lane] = 0; accj[64 + lane] = 0; accj[128 + lane] = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const bool diag = I == J;
    // diagonal units: every ordered pair of the tile once, forward only, step 0 (the blob itself) left to finalize
    for (int k = (diag && k0 < 1) ? 1 : k0; k < k1; ++k) {
      const int jj = (lane + k) & 63;
      const float zj = rec[128 + jj];
      const float vj[3] = {rec[192 + jj], rec[256 + jj], rec[320 + jj]};
      float t[3];
      // separations from the head / tail split of the fp64 positions: the head difference is exact for near pairs
      // (Sterbenz), the tail difference restores what the rounding to float dropped -- the error of d no longer grows
      // with the size of the domain
      // (x, y) as packed pairs: the plane pairs come out of one ds_read2st64_b32 each into adjacent registers
      const v2f xyj = {rec[jj], rec[64 + jj]}, xyjl = {rec[384 + jj], rec[448 + jj]};
      const v2f xyi = {xi, yi}, xyil = {xil, yil};
      const v2f dxy = (xyi - xyj) + (xyil - xyjl);
      const float dz = (zi - zj) + (zil - rec[512 + jj]);
      pair_tt_sym32<WALL>(kf, dxy.x, dxy.y, dz, zi, zj, vi, vj, ui, t);
      if (!diag) {   // wave-uniform
        __hip_atomic_fetch_add(&accj[jj], (double)t[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(&accj[64 + jj], (double)t[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(&accj[128 + jj], (double)t[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
    }
    if (!diag) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const long j = 64L * J + lane;
      if (j < a.n) {
        __hip_atomic_fetch_add(&a.acc[j], accj[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&a.acc[a.n_pad + j], accj[64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&a.acc[2 * a.n_pad + j], accj[128 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __builtin_amdgcn_wave_barrier();   // accj / rec are rewritten by the next unit
    if (k1 == 64) {
      unit_next(a.order, a.n_tiles, I, J);
    }
  }
  if (I_cur >= 0 && vi_ok) {
    __hip_atomic_fetch_add(&a.acc[i], (double)ui[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&a.acc[a.n_pad + i], (double)ui[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&a.acc[2 * a.n_pad + i], (double)ui[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  }   // chunks
}

// ---------------------------------------------------------------------------------------------
// Blob-blob forces in single precision -- what the reference's GPU force kernel always computes in
// (multi_bodies/forces_pycuda.py:14, :21 `precision = 'single'`, `typedef float real`).  Same unit schedule as
// sym_force_kernel; pair arithmetic in float with the hardware exponential (v_exp_f32), partial sums of at most 64 pairs
// in fp32, then added in fp64 (LDS and global accumulators, same finalize).  Open boundaries; uniform radius or per-blob
// radii (RADII); separations from the head / tail split of the fp64 positions, so the exponent (2a - r)/b does not lose
// accuracy in large domains (the reference's float kernel subtracts rounded coordinates).  Context option "precision" = 32.
// ---------------------------------------------------------------------------------------------
template <bool RADII>
__global__ __launch_bounds__(64 * kSymWaves) void sym_force32_kernel(const SymForceArgs a) {
  __shared__ float4 rec_all[kSymWaves][64];       // (x, y, z) float heads of the positions, w = radius
  __shared__ float4 tail_all[kSymWaves][64];      // their float tails (x = head + tail to ~1e-15): see sym32_tt_kernel
  __shared__ double accj_all[kSymWaves][3 * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float4* rec = rec_all[wave];
  float4* tail = tail_all[wave];
  double* accj = accj_all[wave];
  const long n_waves = (long)gridDim.x * kSymWaves;
  const long w = (a.xcd ? xcd_swizzle(blockIdx.x, gridDim.x) : (long)blockIdx.x) * kSymWaves + wave;
  const long s_total = a.step_end - a.step_begin;
  const long spw = a.chunk_steps > 0 ? a.chunk_steps : (s_total + n_waves - 1) / n_waves;
  // strided chunks: wave / workgroup `id` takes the step ranges id, id + n, id + 2 n, ... of `spw` steps each (one range when
  // the launch is planned that way: n spw >= the steps of the launch).  Waves that run at the same time then work on
  // NEIGHBOURING ranges whatever the size of the problem -- with the blocked unit order and the XCD-aware numbering
  // that keeps a launch's tile loads in one L2 (profiles/r4_unit_order.txt).
  for (long chunk = w;; chunk += n_waves) {
  long s = a.step_begin + chunk * spw;
  if (s >= a.step_end) break;
  long s_end = s + spw;
  if (s_end > a.step_end) s_end = a.step_end;
  int I = 0, J = 0;
  if (s < s_end) unit_seek(a.order, s >> 6, a.n_tiles, I, J);
  int I_cur = -1;
  long i = 0;
  bool vi_ok = false;
  float xi = 0, yi = 0, zi = 0, xil = 0, yil = 0, zil = 0, ri = 0;
  float ax = 0, ay = 0, az = 0;
  const float eps_over_b = (float)a.eps_over_b, inv_b_log2e = (float)(a.inv_b * 1.4426950408889634), two_a0 = (float)a.two_a;
  auto flush_row = [&]() {
    if (!vi_ok) return;
    __hip_atomic_fetch_add(&a.acc[i], (double)ax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&a.acc[a.n_pad + i], (double)ay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&a.acc[2 * a.n_pad + i], (double)az, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  while (s < s_end) {
    const int k0 = (int)(s & 63);
    const long left = s_end - s;
    const int k1 = (left < 64 - k0) ? (int)(k0 + left) : 64;
    s += k1 - k0;
    if (a.bounds != nullptr && I != J && tile_gap2(a.bounds, I, J) > a.cull2) {
      // beyond the range of the float exponential (cull2 = (2a + 110 b)^2 here): the unit contributes exactly zero
      if (k1 == 64) {
        unit_next(a.order, a.n_tiles, I, J);
      }
      continue;
    }
    if (I != I_cur) {
      if (I_cur >= 0) flush_row();
      I_cur = I;
      i = 64L * I + lane;
      vi_ok = i < a.n;
      xi = 1e18f; yi = 1e18f; zi = 1e18f; xil = 0; yil = 0; zil = 0;
      if (vi_ok) {
        const double4 p = a.pos[i];
        xi = (float)p.x; yi = (float)p.y; zi = (float)p.z;
        xil = (float)(p.x - (double)xi); yil = (float)(p.y - (double)yi); zil = (float)(p.z - (double)zi);
      }
      if constexpr (RADII) ri = vi_ok ? (float)a.radii[i] : 0.0f;
      ax = 0; ay = 0; az = 0;
    }
    {
      const long j = 64L * J + lane;
      float4 q = make_float4(-1e18f, -1e18f, -1e18f, 0.0f), ql = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (j < a.n) {
        const double4 p = a.pos[j];
        q.x = (float)p.x; q.y = (float)p.y; q.z = (float)p.z;
        ql.x = (float)(p.x - (double)q.x); ql.y = (float)(p.y - (double)q.y); ql.z = (float)(p.z - (double)q.z);
      }
      if constexpr (RADII) q.w = (j < a.n) ? (float)a.radii[j] : 0.0f;
      rec[lane] = q;
      tail[lane] = ql;
      accj[lane] = 0.0; accj[64 + lane] = 0.0; accj[128 + lane] = 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool diag = (I == J);
    for (int k = (diag && k0 < 1) ? 1 : k0; k < k1; ++k) {
      const int jj = (lane + k) & 63;
      const float4 q = rec[jj], ql = tail[jj];
      const float dx = (q.x - xi) + (ql.x - xil), dy = (q.y - yi) + (ql.y - yil), dz = (q.z - zi) + (ql.z - zil);
      const float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
      const float ir = __builtin_amdgcn_rsqf(r2);
      const float r = r2 * ir;
      const float two_a = RADII ? ri + q.w : two_a0;
      // branch-free as pair_force (sym_kernels.h): exponent 0 exactly for r <= 2a, exp2(0) = 1, min(1/r, 1e25) = 1/r beyond
      const float e = __builtin_amdgcn_exp2f(__builtin_fminf((two_a - r) * inv_b_log2e, 0.0f));
      const float f0 = -eps_over_b * (e * __builtin_fminf(ir, 1e25f));
      const float fx = f0 * dx, fy = f0 * dy, fz = f0 * dz;
      ax += fx; ay += fy; az += fz;
      if (!diag) {   // wave-uniform; +f here, the sign of the reaction goes into the flush
        __hip_atomic_fetch_add(&accj[jj], (double)fx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(&accj[64 + jj], (double)fy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(&accj[128 + jj], (double)fz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
    }
    if (!diag) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const long j = 64L * J + lane;
      if (j < a.n) {
        __hip_atomic_fetch_add(&a.acc[j], -accj[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&a.acc[a.n_pad + j], -accj[64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&a.acc[2 * a.n_pad + j], -accj[128 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (k1 == 64) {
      unit_next(a.order, a.n_tiles, I, J);
    }
  }
  if (I_cur >= 0) flush_row();
  }   // chunks
}

}  // namespace rmb
