// sym2t_kernels.h -- the symmetric pair sweep with TWO target blobs per lane (tt / tr / rt / rr, open boundaries, fp64).
//
// sym_kernel keeps one target blob per lane and walks the 64 records of a staged tile J in LDS: per pair ~80 VALU
// instructions plus three ds_read_b128, three ds_add_f64 and the address arithmetic of the rotation.  Here a lane keeps
// the blobs `lane` of TWO tile rows (2p and 2p + 1): a rotation step reads the record of blob jj once, evaluates both
// pairs and adds the SUM of the two transposed contributions with one set of ds_add_f64 -- half the LDS instructions,
// half the staging / flush work per pair (tools/experiments/exp_two_targets.hip: +4 % at 1e5 blobs, 114 VGPRs, still
// four waves per SIMD).
//
// Unit = (row pair p, tile J), J >= 2p, in the plain or the blocked order (unit2_seek / unit2_next); a unit has 64 rotation steps,
// a step evaluates the pairs (2p, jj) and (2p + 1, jj).  The step schedule of sym_kernel carries over unchanged: a
// launch is the step range [step_begin, step_end) of these units, cut into equal contiguous ranges (strided chunks), a
// range may begin and end inside a unit, a pair shard is a step range.  The two columns of a row pair that touch the
// diagonal -- J = 2p: (2p, 2p) diagonal, (2p + 1, 2p) below the diagonal and skipped; J = 2p + 1: (2p, 2p + 1) full,
// (2p + 1, 2p + 1) diagonal -- run row by row through the one-target loops of sym_kernel.
#pragma once
#include "sym_kernels.h"

namespace rmb {

// ---- unit order ------------------------------------------------------------------------------------------------
// units of the row-pair grid: sum over pairs p of (T - 2p)
__host__ __device__ inline long units2_before_pair(long p, long T) { return p * T - p * (p - 1); }
__host__ __device__ inline long units2_total(long T) { return units2_before_pair((T + 1) / 2, T); }

// order 0: pair by pair, J ascending.  order 1: the blocked order of sym_kernels.h on this grid -- super-blocks of
// 16 row pairs (32 tile rows) x 32 tile columns, walked super-row by super-row, pair by pair inside a super-block (the
// diagonal super-block is the staircase J >= 2p), so that waves which run at the same time share tiles in L2.
constexpr int kOrd2Pairs = 1 << (kOrdShift - 1);     // row pairs per super-block

// units before super-row B (all earlier super-rows are full): 16 B (T + 1 - 16 B)
__host__ __device__ inline long blk2_units_before_row(long B, long T) { return (long)kOrd2Pairs * B * (T + 1 - (long)kOrd2Pairs * B); }

__device__ __forceinline__ void unit2_seek(int order, long u, int T, int& p, int& J) {
  const double b = (double)T + 1.0;
  double disc = b * b - 4.0 * (double)u;
  if (disc < 0.0) disc = 0.0;
  const double y = (b - sqrt(disc)) * 0.5;             // smaller root of y (T + 1 - y) = u
  if (order == 0) {
    long q = (long)y;
    const long P = ((long)T + 1) / 2;
    if (q < 0) q = 0;
    if (q > P - 1) q = P - 1;
    while (q > 0 && units2_before_pair(q, T) > u) --q;
    while (q + 1 < P && units2_before_pair(q + 1, T) <= u) ++q;
    p = (int)q;
    J = (int)(2 * q + (u - units2_before_pair(q, T)));
    return;
  }
  const long NB = ((long)T + (1 << kOrdShift) - 1) >> kOrdShift;     // super-rows
  long B = (long)(y / kOrd2Pairs);
  if (B < 0) B = 0;
  if (B > NB - 1) B = NB - 1;
  while (B > 0 && blk2_units_before_row(B, T) > u) --B;
  while (B + 1 < NB && blk2_units_before_row(B + 1, T) <= u) ++B;
  long rem = u - blk2_units_before_row(B, T);
  const int row0 = (int)(B << kOrdShift);                               // first tile row (and first tile column) of the diagonal super-block
  const int w = (T - row0) < (1 << kOrdShift) ? (T - row0) : (1 << kOrdShift);
  const int sP = (w + 1) / 2;                                           // row pairs of this super-row
  const long tri = (long)sP * w - (long)sP * (sP - 1);
  if (rem < tri) {                                                      // diagonal super-block: pair lp has the columns 2 lp .. w - 1
    int lp = 0;
    while (lp + 1 < sP && (long)(lp + 1) * w - (long)(lp + 1) * lp <= rem) ++lp;
    const long before = (long)lp * w - (long)lp * (lp - 1);
    p = (int)(B * kOrd2Pairs) + lp;
    J = row0 + 2 * lp + (int)(rem - before);
    return;
  }
  rem -= tri;
  const int col0 = row0 + (1 << kOrdShift);                             // first column right of the diagonal super-block
  const long per = (long)sP << kOrdShift;                               // units of a full-width super-block
  const long q = rem / per;
  const int c0 = col0 + (int)(q << kOrdShift);
  const int wQ = (T - c0) < (1 << kOrdShift) ? (T - c0) : (1 << kOrdShift);
  const long rem2 = rem - q * per;
  const int lp = (int)(rem2 / wQ);
  p = (int)(B * kOrd2Pairs) + lp;
  J = c0 + (int)(rem2 - (long)lp * wQ);
}

__device__ __forceinline__ void unit2_next(int order, int T, int& p, int& J) {
  if (order == 0) {
    if (++J < T) return;
    ++p;
    J = 2 * p;
    return;
  }
  const int B = p / kOrd2Pairs, Q = J >> kOrdShift;
  const int col_end = ((Q + 1) << kOrdShift) < T ? ((Q + 1) << kOrdShift) : T;
  const int P = (T + 1) / 2;
  const int pair_end = (B + 1) * kOrd2Pairs < P ? (B + 1) * kOrd2Pairs : P;
  if (++J < col_end) return;                                            // same pair, same super-block
  if (++p < pair_end) { J = (Q == B) ? 2 * p : (Q << kOrdShift); return; }   // next pair of the super-block
  if (((Q + 1) << kOrdShift) < T) { p = B * kOrd2Pairs; J = (Q + 1) << kOrdShift; return; }   // next super-block of the super-row
  p = (B + 1) * kOrd2Pairs;                                             // next super-row: its diagonal super-block
  J = 2 * p;
}

template <int KIND, bool WALL>
__global__ __launch_bounds__(64 * kSymWaves) __attribute__((amdgpu_waves_per_eu(kSymWavesPerEu, kSymWavesPerEu))) void sym2t_kernel(const SymArgs a) {
  __shared__ double2 rec_all[kSymWaves][64 * 3];
  __shared__ double accj_all[kSymWaves][3 * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double2* rec = rec_all[wave];
  double* accj = accj_all[wave];
  const char* rec_bytes = reinterpret_cast<const char*>(rec);

  const long w = (a.xcd ? xcd_swizzle(blockIdx.x, gridDim.x) : (long)blockIdx.x) * kSymWaves + wave;
  // diagnostics build only (option "wave_clock"): wall-clock start / end of the wave and the shader-clock cycles it spends
  // staging tiles (loads issued -> records visible in LDS); a.wave_clock is null otherwise (wave-uniform branches)
  const long long t_start = a.wave_clock ? wall_clock64() : 0;
  const long long c_start = a.wave_clock ? (long long)__builtin_readcyclecounter() : 0;
  long long c_stage = 0;
  for (long chunk = w;; chunk += (long)gridDim.x * kSymWaves) {
  long s = a.step_begin + chunk * a.steps_per_wave;
  if (s >= a.step_end) break;
  long s_end = s + a.steps_per_wave;
  if (s_end > a.step_end) s_end = a.step_end;
  int p = 0, J = 0;
  unit2_seek(a.order, s >> 6, a.n_tiles, p, J);

  int p_cur = -1;
  long i0 = 0, i1 = 0;
  bool ok0 = false, ok1 = false;
  double x0 = 0, y0 = 0, z0 = 1.0, v0x = 0, v0y = 0, v0z = 0;
  double x1 = 0, y1 = 0, z1 = 1.0, v1x = 0, v1y = 0, v1z = 0;
  Vec3 u0 = {0.0, 0.0, 0.0}, u1 = {0.0, 0.0, 0.0};

  auto flush_rows = [&]() {
    if (ok0) {
      __hip_atomic_fetch_add(&a.acc[i0], u0.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&a.acc[a.n_pad + i0], u0.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&a.acc[2 * a.n_pad + i0], u0.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (ok1) {
      __hip_atomic_fetch_add(&a.acc[i1], u1.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&a.acc[a.n_pad + i1], u1.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&a.acc[2 * a.n_pad + i1], u1.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };

  while (s < s_end) {
    const int k0 = (int)(s & 63);
    const long left = s_end - s;
    const int k1 = (left < 64 - k0) ? (int)(k0 + left) : 64;
    s += k1 - k0;

    const long long c_s0 = a.wave_clock ? (long long)__builtin_readcyclecounter() : 0;
    if (p != p_cur) {
      if (p_cur >= 0) flush_rows();
      p_cur = p;
      i0 = 64L * (2 * p) + lane;
      i1 = i0 + 64;
      ok0 = i0 < a.n; ok1 = i1 < a.n;
      x0 = 1e100; y0 = 1e100; z0 = 1.0; v0x = 0; v0y = 0; v0z = 0;
      x1 = 1e100; y1 = 1e100; z1 = 1.0; v1x = 0; v1y = 0; v1z = 0;
      if (ok0) {
        const double4 q = a.pos[i0];
        x0 = q.x; y0 = q.y; z0 = q.z;
        v0x = a.vec[3 * i0] * q.w; v0y = a.vec[3 * i0 + 1] * q.w; v0z = a.vec[3 * i0 + 2] * q.w;
      }
      if (ok1) {
        const double4 q = a.pos[i1];
        x1 = q.x; y1 = q.y; z1 = q.z;
        v1x = a.vec[3 * i1] * q.w; v1y = a.vec[3 * i1 + 1] * q.w; v1z = a.vec[3 * i1 + 2] * q.w;
      }
      u0.x = 0.0; u0.y = 0.0; u0.z = 0.0;
      u1.x = 0.0; u1.y = 0.0; u1.z = 0.0;
    }
    {   // tile J -> this wave's LDS slab, its accumulators zeroed
      const long j = 64L * J + lane;
      double xj = -1e100, yj = -1e100, zj = 1.0, vjx = 0, vjy = 0, vjz = 0;
      if (j < a.n) {
        const double4 q = a.pos[j];
        xj = q.x; yj = q.y; zj = q.z;
        vjx = a.vec[3 * j] * q.w; vjy = a.vec[3 * j + 1] * q.w; vjz = a.vec[3 * j + 2] * q.w;
      }
      rec[lane * 3 + 0] = make_double2(xj, yj);
      rec[lane * 3 + 1] = make_double2(zj, vjx);
      rec[lane * 3 + 2] = make_double2(vjy, vjz);
      accj[lane] = 0.0; accj[64 + lane] = 0.0; accj[128 + lane] = 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (a.wave_clock) c_stage += (long long)__builtin_readcyclecounter() - c_s0;

    const int kb = (a.skip_pairs & 1) ? k1 : k0;
    if (J >= 2 * p + 2) {
      // both rows off-diagonal: one record read and one set of LDS adds for two pairs
      for (int k = kb; k < k1; ++k) {
        const int jj = (lane + k) & 63;
        const double2* r = reinterpret_cast<const double2*>(rec_bytes + jj * kSymRecBytes);
        const double2 q0 = r[0], q1 = r[1], q2 = r[2];
        // the second pair ACCUMULATES its transposed rows into the first one's (the multiplies that start them become fused
        // multiply-adds: three v_add_f64 less per step than summing two finished contributions)
        double ax, ay, az;
        pair_sym<KIND, WALL>(a.k, x0 - q0.x, y0 - q0.y, z0 - q1.x, z0, q1.x, v0x, v0y, v0z, q1.y, q2.x, q2.y, u0, ax, ay, az);
        pair_sym<KIND, WALL, true>(a.k, x1 - q0.x, y1 - q0.y, z1 - q1.x, z1, q1.x, v1x, v1y, v1z, q1.y, q2.x, q2.y, u1, ax, ay, az);
        __hip_atomic_fetch_add(&accj[jj], ax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(&accj[64 + jj], ay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(&accj[128 + jj], az, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
    } else {
      // the two columns at the diagonal of this row pair, row by row (the loops of sym_kernel)
      // row 2p: diagonal unit when J == 2p, full unit when J == 2p + 1
      if (J == 2 * p) {
        for (int k = (kb > 1 ? kb : ((a.skip_pairs & 1) ? k1 : 1)); k < k1; ++k) {
          const int jj = (lane + k) & 63;
          const double2* r = reinterpret_cast<const double2*>(rec_bytes + jj * kSymRecBytes);
          const double2 q0 = r[0], q1 = r[1], q2 = r[2];
          pair_apply<KIND, WALL>(a.k, x0 - q0.x, y0 - q0.y, z0 - q1.x, z0, q1.x, q1.y, q2.x, q2.y, 0.0, 0.0, 0.0, u0);
        }
      } else {
        for (int k = kb; k < k1; ++k) {
          const int jj = (lane + k) & 63;
          const double2* r = reinterpret_cast<const double2*>(rec_bytes + jj * kSymRecBytes);
          const double2 q0 = r[0], q1 = r[1], q2 = r[2];
          double tx, ty, tz;
          pair_sym<KIND, WALL>(a.k, x0 - q0.x, y0 - q0.y, z0 - q1.x, z0, q1.x, v0x, v0y, v0z, q1.y, q2.x, q2.y, u0, tx, ty, tz);
          __hip_atomic_fetch_add(&accj[jj], tx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          __hip_atomic_fetch_add(&accj[64 + jj], ty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          __hip_atomic_fetch_add(&accj[128 + jj], tz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        // row 2p + 1: its diagonal unit (J == 2p + 1); at J == 2p it lies below the diagonal
        for (int k = (kb > 1 ? kb : ((a.skip_pairs & 1) ? k1 : 1)); k < k1; ++k) {
          const int jj = (lane + k) & 63;
          const double2* r = reinterpret_cast<const double2*>(rec_bytes + jj * kSymRecBytes);
          const double2 q0 = r[0], q1 = r[1], q2 = r[2];
          pair_apply<KIND, WALL>(a.k, x1 - q0.x, y1 - q0.y, z1 - q1.x, z1, q1.x, q1.y, q2.x, q2.y, 0.0, 0.0, 0.0, u1);
        }
      }
    }
    if (J != 2 * p) {     // the slab holds transposed contributions (none in the pure diagonal column)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const long j = 64L * J + lane;
      if (j < a.n && !(a.skip_pairs & 2)) {
        __hip_atomic_fetch_add(&a.acc[j], accj[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&a.acc[a.n_pad + j], accj[64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&a.acc[2 * a.n_pad + j], accj[128 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __builtin_amdgcn_wave_barrier();   // accj / rec are rewritten by the next unit
    if (k1 == 64) unit2_next(a.order, a.n_tiles, p, J);
  }
  if (p_cur >= 0) flush_rows();
  }   // chunks
  if (a.wave_clock && lane == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((16 - 1) << 11));
    const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));
    const long n_waves = (long)gridDim.x * kSymWaves;
    a.wave_clock[2 * w] = t_start;
    a.wave_clock[2 * w + 1] = (wall_clock64() & 0xffffffffffLL) | ((long long)(hw & 0xffff) << 40) | ((long long)(xcc & 0xf) << 56);
    a.wave_clock[2 * (n_waves + w)] = c_stage;
    a.wave_clock[2 * (n_waves + w) + 1] = (long long)__builtin_readcyclecounter() - c_start;
  }
}

}  // namespace rmb
