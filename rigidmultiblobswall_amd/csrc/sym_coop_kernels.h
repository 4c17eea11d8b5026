// sym_coop_kernels.h -- workgroup-cooperative variant of the symmetric pair sweep (sym_kernels.h).
//
// sym_kernel gives every WAVE its own range of rotation steps: each wave stages its own copy of tile J, and flushes its
// own partial of u_I and u_J with 2 x 192 global atomics.  That is free while a wave runs >= 64 steps per staged tile
// and three other waves of the SIMD hide the staging; it is most of the time of a launch that is smaller than one
// resident round -- a small suspension, or one rank's pair shard (1/8 shard of 1e4 blobs: 24 of 29.6 us without the pair
// arithmetic, profiles/r3_shard_plan.txt) -- where waves are cut to 16-32 steps to fill the chip and every wave still
// pays both tiles and 384 atomics on accumulators it shares with the other waves of its tile row.
//
// Here the schedule unit is the WORKGROUP: it owns a contiguous range of rotation steps and walks it tile pair by tile
// pair; the four waves split the steps of each piece (k in [k0, k1) -> four sub-ranges), read ONE staged copy of tile J
// and add into ONE LDS accumulator pair:
//   * tile J is staged once per piece (wave 0), not once per wave;
//   * the transposed contributions go to the shared u_J slab with ds_add_f64 (LDS atomics are atomic across the waves
//     of a workgroup; inside a wave the rotation still gives every lane its own j);
//   * every wave adds its register partial of u_I into a shared u_I slab after its steps; the slab lives across the
//     pieces of one tile row and is flushed when the row changes;
//   * per piece ONE flush of u_J (192 global atomics, wave 0) and per row ONE flush of u_I (192, wave 1) -- 4x fewer
//     than four 16-step waves issue, 2x fewer than two 32-step waves.
// Two workgroup barriers per piece.  Same pair arithmetic (pair_blocks.h through pair_sym / pair_apply), same global
// accumulators and finalize kernel as sym_kernel; results agree to rounding (atomic arrival order).
#pragma once
#include "sym_kernels.h"

namespace rmb {

template <int KIND, bool WALL, bool PERIODIC>
__global__ __launch_bounds__(64 * kSymWaves) __attribute__((amdgpu_waves_per_eu(kSymWavesPerEu, kSymWavesPerEu))) void sym_coop_kernel(const SymArgs a) {
  __shared__ double2 rec[64 * 3];     // tile J, 48-byte records
  __shared__ double accj[3 * 64];     // u_J of the current piece
  __shared__ double acci[3 * 64];     // u_I of the current tile row
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* rec_bytes = reinterpret_cast<const char*>(rec);

  // a.steps_per_wave carries the steps per WORKGROUP here (rmb_sym.hip)
  // strided chunks: wave / workgroup `id` takes the step ranges id, id + n, id + 2 n, ... of `spw` steps each (one range when
  // the launch is planned that way: n spw >= the steps of the launch).  Waves that run at the same time then work on
  // NEIGHBOURING ranges whatever the size of the problem -- with the blocked unit order and the XCD-aware numbering
  // that keeps a launch's tile loads in one L2 (profiles/r4_unit_order.txt).
  for (long chunk = (a.xcd ? xcd_swizzle(blockIdx.x, gridDim.x) : (long)blockIdx.x);; chunk += (long)gridDim.x) {
  long s = a.step_begin + chunk * a.steps_per_wave;
  if (s >= a.step_end) break;
  long s_end = s + a.steps_per_wave;
  if (s_end > a.step_end) s_end = a.step_end;
  int I = 0, J = 0;
  if (s < s_end) unit_seek(a.order, s >> 6, a.n_tiles, I, J);
  if (wave == 1) { acci[lane] = 0.0; acci[64 + lane] = 0.0; acci[128 + lane] = 0.0; }

  int I_cur = -1;
  long i = 0;
  double xi = 0, yi = 0, zi = 1.0, vix = 0, viy = 0, viz = 0;

  while (s < s_end) {
    const int k0 = (int)(s & 63);
    const long left = s_end - s;
    const int k1 = (left < 64 - k0) ? (int)(k0 + left) : 64;
    s += k1 - k0;

    if (I != I_cur) {
      if (I_cur >= 0 && wave == 1) {   // previous row: every wave's adds are behind the last barrier of its last piece
        if (i < a.n) {
          __hip_atomic_fetch_add(&a.acc[i], acci[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add(&a.acc[a.n_pad + i], acci[64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add(&a.acc[2 * a.n_pad + i], acci[128 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        acci[lane] = 0.0; acci[64 + lane] = 0.0; acci[128 + lane] = 0.0;
      }
      I_cur = I;
      i = 64L * I + lane;
      xi = 1e100; yi = 1e100; zi = 1.0; vix = 0; viy = 0; viz = 0;
      if (i < a.n) {
        const double4 p = a.pos[i];
        xi = p.x; yi = p.y; zi = p.z;
        vix = a.vec[3 * i] * p.w; viy = a.vec[3 * i + 1] * p.w; viz = a.vec[3 * i + 2] * p.w;
      }
    }
    if (wave == 0) {   // tile J -> the workgroup's slab (record l = blob 64 J + l), zero its accumulator
      const long j = 64L * J + lane;
      double xj = -1e100, yj = -1e100, zj = 1.0, vjx = 0, vjy = 0, vjz = 0;
      if (j < a.n) {
        const double4 p = a.pos[j];
        xj = p.x; yj = p.y; zj = p.z;
        vjx = a.vec[3 * j] * p.w; vjy = a.vec[3 * j + 1] * p.w; vjz = a.vec[3 * j + 2] * p.w;
      }
      rec[lane * 3 + 0] = make_double2(xj, yj);
      rec[lane * 3 + 1] = make_double2(zj, vjx);
      rec[lane * 3 + 2] = make_double2(vjy, vjz);
      accj[lane] = 0.0; accj[64 + lane] = 0.0; accj[128 + lane] = 0.0;
    }
    __syncthreads();

    // this wave's share of the piece [k0, k1)
    const int q = (k1 - k0 + kSymWaves - 1) / kSymWaves;
    int ka = k0 + wave * q;
    int kb = ka + q < k1 ? ka + q : k1;
    if (a.skip_pairs & 1) ka = kb;
    Vec3 ui = {0.0, 0.0, 0.0};
    const int px = PERIODIC && a.Lx > 0, py = PERIODIC && a.Ly > 0, pz = PERIODIC && a.Lz > 0;
    if (I != J) {
      for (int k = ka; k < kb; ++k) {
        const int jj = (lane + k) & 63;
        const double2* r = reinterpret_cast<const double2*>(rec_bytes + jj * kSymRecBytes);
        const double2 q0 = r[0], q1 = r[1], q2 = r[2];
        double dx = xi - q0.x, dy = yi - q0.y, dz = zi - q1.x;
        double tx, ty, tz;
        if constexpr (!PERIODIC) {
          pair_sym<KIND, WALL>(a.k, dx, dy, dz, zi, q1.x, vix, viy, viz, q1.y, q2.x, q2.y, ui, tx, ty, tz);
        } else {
          if (px) dx = wrap_nearest_pad_safe(dx, a.Lx, a.iLx);
          if (py) dy = wrap_nearest_pad_safe(dy, a.Ly, a.iLy);
          if (pz) dz = wrap_nearest_pad_safe(dz, a.Lz, a.iLz);
          tx = 0.0; ty = 0.0; tz = 0.0;
          for (int bx = -px; bx <= px; ++bx)
            for (int by = -py; by <= py; ++by)
              for (int bz = -pz; bz <= pz; ++bz) {
                pair_sym<KIND, WALL, true>(a.k, dx + bx * a.Lx, dy + by * a.Ly, dz + bz * a.Lz, zi, q1.x, vix, viy, viz, q1.y, q2.x,
                                           q2.y, ui, tx, ty, tz);      // accumulates (fused multiply-adds)
              }
        }
        __hip_atomic_fetch_add(&accj[jj], tx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&accj[64 + jj], ty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&accj[128 + jj], tz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    } else {
      // diagonal unit: every ordered pair of the tile once, forward only; step 0 is the blob itself (self term in
      // finalize; its periodic images use the pair formula)
      for (int k = (!PERIODIC && ka < 1) ? 1 : ka; k < kb; ++k) {
        const int jj = (lane + k) & 63;
        const double2* r = reinterpret_cast<const double2*>(rec_bytes + jj * kSymRecBytes);
        const double2 q0 = r[0], q1 = r[1], q2 = r[2];
        double dx = xi - q0.x, dy = yi - q0.y, dz = zi - q1.x;
        if constexpr (!PERIODIC) {
          pair_apply<KIND, WALL>(a.k, dx, dy, dz, zi, q1.x, q1.y, q2.x, q2.y, 0.0, 0.0, 0.0, ui);
        } else {
          if (px) dx = wrap_nearest_pad_safe(dx, a.Lx, a.iLx);
          if (py) dy = wrap_nearest_pad_safe(dy, a.Ly, a.iLy);
          if (pz) dz = wrap_nearest_pad_safe(dz, a.Lz, a.iLz);
          for (int bx = -px; bx <= px; ++bx)
            for (int by = -py; by <= py; ++by)
              for (int bz = -pz; bz <= pz; ++bz) {
                if (k == 0 && bx == 0 && by == 0 && bz == 0) continue;
                pair_apply<KIND, WALL>(a.k, dx + bx * a.Lx, dy + by * a.Ly, dz + bz * a.Lz, zi, q1.x, q1.y, q2.x, q2.y, 0.0, 0.0,
                                       0.0, ui);
              }
        }
      }
    }
    // this wave's partial of u_I joins the row's slab
    __hip_atomic_fetch_add(&acci[lane], ui.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(&acci[64 + lane], ui.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(&acci[128 + lane], ui.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    if (wave == 0 && I != J && !(a.skip_pairs & 2)) {   // one flush of u_J per piece; wave 0 re-stages the slab next
      const long j = 64L * J + lane;
      if (j < a.n) {
        __hip_atomic_fetch_add(&a.acc[j], accj[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&a.acc[a.n_pad + j], accj[64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&a.acc[2 * a.n_pad + j], accj[128 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (k1 == 64) {                    // next unit in row-major order
      unit_next(a.order, a.n_tiles, I, J);
    }
  }
  if (I_cur >= 0 && wave == 1 && i < a.n) {
    __hip_atomic_fetch_add(&a.acc[i], acci[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&a.acc[a.n_pad + i], acci[64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&a.acc[2 * a.n_pad + i], acci[128 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  }   // chunks
}

}  // namespace rmb
