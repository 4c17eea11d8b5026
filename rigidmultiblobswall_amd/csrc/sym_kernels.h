// sym_kernels.h -- symmetric pair sweep for the translation-translation mobility (gfx950, fp64).
//
// The blob mobility is symmetric as a 3N x 3N matrix, M_ji = M_ij^T (for the wall part this is the
// Swan-Brady reciprocity the reference's C++ twin also relies on, mobility/mobility.cpp:407-424).
// sweep_kernel evaluates every ORDERED pair; here every UNORDERED pair is evaluated once and applied
// to both blobs:   u_i += M_ij v_j   and   u_j += M_ij^T v_i.
// The expensive part (two rsqrt, RPY coefficients, the five wall polynomials G1..G5) is shared; only the
// cheap contractions are done twice.  With W = f1 I + f2 e e^T + f3 e z^T + f4 z e^T + f5 z z^T,
//   W v   = f1 v + [f2 (e.v) + f3 v_z] e + [f4 (e.v) + f5 v_z] z
//   W^T v = f1 v + [f2 (e.v) + f4 v_z] e + [f3 (e.v) + f5 v_z] z          (f3 <-> f4)
// => 80 VALU instructions per unordered pair (76 fp64; pair_blocks.h: five-entry block + closed-form wall
// polynomials through H) instead of 2 x 67 in the bulk loop of the one-sided sweep.
//
// Work decomposition: blobs are cut into tiles of 64; a work unit is a tile pair (I <= J).  One wave64
// owns a unit: lane l holds blob i = 64 I + l in registers (position, its own vector v_i, accumulator u_i);
// tile J sits in the wave's private LDS slab.  At step k lane l meets blob j = 64 J + ((l + k) & 63): the
// "rotation" makes every lane touch a different j, so the transposed contribution is added to the
// per-wave LDS accumulator of j without conflicts (ds_add_f64) and source records are plain per-lane
// ds_read_b128 (48-byte records: conflict-free).  Diagonal units (I == J) run forward-only over
// k = 1..63 (every ordered pair exactly once).  After 64 steps u_i (registers) and u_J (LDS) are flushed
// to global SoA accumulators with global_atomic_add_f64; finalize adds the self term and scales.
// Schedule: static and exactly balanced -- the n_units x 64 rotation steps are cut into equal contiguous
// ranges, one per wave (a range may start/end inside a unit; >= 64 steps each; 8x more workgroups than are
// resident at once), so every SIMD gets the same number of steps and no work counter is needed (a single dequeue word saturates at ~88 dequeues/us, which is
// about the unit rate of this kernel at N = 1e4).  Consecutive units share the row tile I, whose
// accumulator stays in registers until the row changes.  The order in which the atomics land is not
// fixed: results agree with the deterministic sweep_kernel to rounding (~1e-15 relative) but are not
// bit-reproducible; the "deterministic" context option selects sweep_kernel instead.
#pragma once
#include "pair_blocks.h"

namespace rmb {

struct SymArgs {
  const double4* pos;   // [n] packed positions
  const double* vec;    // [3n] source vector (AoS)
  double* acc;          // [3][n_pad] global SoA accumulators; zero on entry, re-zeroed by finalize
  double* out;          // [3n] final output (AoS)
  long n;
  long n_pad;           // 64 * n_tiles
  int n_tiles;
  long n_units;         // n_tiles (n_tiles + 1) / 2
  int order;            // unit order: 0 row-major, 1 blocked (unit_seek / unit_next); xcd != 0: XCD-aware workgroup numbering
  int xcd;
  long step_begin, step_end;  // rotation steps [begin, end) of the n_units*64 this launch covers (pair shard)
  long steps_per_wave;        // ceil((step_end - step_begin) / waves): wave w takes [begin + w spw, +spw)
  long self_begin, self_end;  // targets whose self term this launch adds (exactly one shard per target)
  double Lx, Ly, Lz, iLx, iLy, iLz;  // pseudo-periodic lengths (<= 0: open) and reciprocals
  double prefactor;
  int accumulate;         // finalize adds to `out` instead of overwriting it (second pass of the fused tt+tr product)
  int skip_pairs;         // diagnostics: run the schedule / loads / flushes but no pair arithmetic (timing only)
  long long* wave_clock;  // optional [n_waves][2] wall-clock stamps (start, end | placement bits) for schedule diagnostics, or nullptr
  PairConsts k;
};

constexpr int kSymRecBytes = 48;

// Both directions of one pair.  (vix..) = target's own vector, (vjx..) = source vector.
// Adds M_ij v_j to ui and returns M_ij^T v_i in (tx,ty,tz).  The block is built once in the form of pair_blocks.h
// (BlockM: F, P, Q3, Q4, Szz) and contracted with ten instructions per direction.
// ACC: the transposed rows are ADDED to (tx, ty, tz) instead of written -- the second target blob of sym2t_kernel folds its
// contribution into the first one's with the fused multiply-adds that build it (three v_add_f64 less per rotation step).
template <bool WALL, bool ACC = false>
__device__ __forceinline__ void pair_tt_sym(const PairConsts& k, double dx, double dy, double dz, double zi, double zj,
                                            double vix, double viy, double viz, double vjx, double vjy, double vjz,
                                            Vec3& ui, double& tx, double& ty, double& tz) {
  const Geom g = make_geom<WALL>(dx, dy, dz, zi, zj);
  const TTc c = tt_coeffs<WALL>(k, g, zi, zj);
  const double vi[3] = {vix, viy, viz}, vj[3] = {vjx, vjy, vjz};
  double u[3] = {ui.x, ui.y, ui.z}, t[3];
  if constexpr (ACC) { t[0] = tx; t[1] = ty; t[2] = tz; }
  tt_apply<WALL, ACC>(c, g, vi, vj, u, t);
  ui.x = u[0]; ui.y = u[1]; ui.z = u[2];
  tx = t[0]; ty = t[1]; tz = t[2];
}

// rr, both directions (same block form, pair_blocks.h: rr_coeffs).
template <bool WALL, bool ACC = false>
__device__ __forceinline__ void pair_rr_sym(const PairConsts& k, double dx, double dy, double dz, double zi, double zj,
                                            double vix, double viy, double viz, double vjx, double vjy, double vjz,
                                            Vec3& ui, double& tx, double& ty, double& tz) {
  const Geom g = make_geom<WALL>(dx, dy, dz, zi, zj);
  const RRc c = rr_coeffs<WALL>(k, g);
  const double vi[3] = {vix, viy, viz}, vj[3] = {vjx, vjy, vjz};
  double u[3] = {ui.x, ui.y, ui.z}, t[3];
  if constexpr (ACC) { t[0] = tx; t[1] = ty; t[2] = tz; }
  rr_apply<WALL, ACC>(c, g, vi, vj, u, t);
  ui.x = u[0]; ui.y = u[1]; ui.z = u[2];
  tx = t[0]; ty = t[1]; tz = t[2];
}

// Coupling blocks, both directions.  KIND_TR: u = M_tr tau, wall part anchored on the TARGET height of each
// direction (z_i forward, z_j transposed); KIND_RT: w = M_rt f, anchored on the SOURCE height (z_j forward,
// z_i transposed).  The two directions share both rsqrt, tau, e and differ only in g = z iR, which enters
// p, s, f3 linearly (pair_blocks.h: cpl_coeffs / tr_apply / rt_apply on the unnormalised separation).
template <bool TR, bool WALL, bool ACC = false>
__device__ __forceinline__ void pair_coupling_sym(const PairConsts& k, double dx, double dy, double dz, double zi, double zj,
                                                  double vix, double viy, double viz, double vjx, double vjy, double vjz,
                                                  Vec3& ui, double& tx, double& ty, double& tz) {
  const Geom g = make_geom<WALL>(dx, dy, dz, zi, zj);
  const CPc C = cpl_coeffs<WALL>(k, g, zi, zj);
  const double vi[3] = {vix, viy, viz}, vj[3] = {vjx, vjy, vjz};
  double u[3] = {ui.x, ui.y, ui.z}, t[3];
  if constexpr (ACC) { t[0] = tx; t[1] = ty; t[2] = tz; }
  if constexpr (TR) tr_apply<WALL, ACC>(C, g, vi, vj, u, t);
  else              rt_apply<WALL, ACC>(C, g, vi, vj, u, t);
  ui.x = u[0]; ui.y = u[1]; ui.z = u[2];
  tx = t[0]; ty = t[1]; tz = t[2];
}

// dispatcher
template <int KIND, bool WALL, bool ACC = false>
__device__ __forceinline__ void pair_sym(const PairConsts& k, double dx, double dy, double dz, double zi, double zj,
                                         double vix, double viy, double viz, double vjx, double vjy, double vjz,
                                         Vec3& ui, double& tx, double& ty, double& tz) {
  if constexpr (KIND == KIND_TT) pair_tt_sym<WALL, ACC>(k, dx, dy, dz, zi, zj, vix, viy, viz, vjx, vjy, vjz, ui, tx, ty, tz);
  if constexpr (KIND == KIND_RR) pair_rr_sym<WALL, ACC>(k, dx, dy, dz, zi, zj, vix, viy, viz, vjx, vjy, vjz, ui, tx, ty, tz);
  if constexpr (KIND == KIND_TR) pair_coupling_sym<true, WALL, ACC>(k, dx, dy, dz, zi, zj, vix, viy, viz, vjx, vjy, vjz, ui, tx, ty, tz);
  if constexpr (KIND == KIND_RT) pair_coupling_sym<false, WALL, ACC>(k, dx, dy, dz, zi, zj, vix, viy, viz, vjx, vjy, vjz, ui, tx, ty, tz);
}

__device__ __forceinline__ double wrap_nearest_sym(double r, double L, double invL) {
  const double q = r * invL;
  const double h = (r > 0.0) ? 0.5 : ((r < 0.0) ? -0.5 : 0.0);
  return __builtin_fma(-__builtin_trunc(q + h), L, r);
}

// Nearest image that leaves the padding sentinels (+-1e100) alone: wrapped, a sentinel can land exactly on a real
// blob (fma(-trunc(1e100/L), L, 1e100) == 0 for power-of-two L) and 1/r = inf would reach the accumulators.
__device__ __forceinline__ double wrap_nearest_pad_safe(double r, double L, double invL) {
  const double w = wrap_nearest_sym(r, L, invL);
  return (__builtin_fabs(r) < 1e50) ? w : r;
}

__device__ __forceinline__ void unit_to_tiles(long u, int T, int& I, int& J) {
  // row-major over the upper triangle: row I holds (T - I) units
  const double tt = 2.0 * T + 1.0;
  long i = (long)((tt - sqrt(tt * tt - 8.0 * (double)u)) * 0.5);
  if (i < 0) i = 0;
  if (i > T - 1) i = T - 1;
  while (i > 0 && i * T - i * (i - 1) / 2 > u) --i;
  while ((i + 1) * T - (i + 1) * i / 2 <= u) ++i;
  I = (int)i;
  J = (int)(u - (i * T - i * (i - 1) / 2) + i);
}

// ---- unit order and workgroup placement (round 4) ------------------------------------------------------------
// order 0: row-major over the tile triangle (what the deterministic mode's ordered reduction assumes).
// order 1: BLOCKED -- the triangle is cut into super-blocks of 32 x 32 tiles; super-blocks in row-major order, and
//   inside a super-block the units in row-major order (the triangle I <= J inside a diagonal one).  Waves that work on
//   neighbouring step ranges then touch the same 64 tiles for ~1000 units instead of sweeping a whole row of the
//   triangle, and with the XCD-aware numbering below those waves sit behind ONE L2: the tile-J loads (3.6 KB per 4096
//   pairs, all of them L2 misses at >= 1e5 blobs in row-major order) mostly hit.
constexpr int kOrdShift = 5;
constexpr int kOrdB = 1 << kOrdShift;

__device__ __forceinline__ long blk_units_before_row(long P, long T) {   // super-rows before P are all kOrdB tall
  const long B = kOrdB;
  return P * (B * (B + 1) / 2) + B * (P * T - B * (P * (P + 1) / 2));
}

__device__ __forceinline__ void unit_seek(int order, long u, int T, int& I, int& J) {
  if (order == 0) { unit_to_tiles(u, T, I, J); return; }
  const int SB = (T + kOrdB - 1) >> kOrdShift;
  int lo = 0, hi = SB - 1;
  while (lo < hi) {                       // largest super-row whose first unit is <= u
    const int mid = (lo + hi + 1) >> 1;
    if (blk_units_before_row(mid, T) <= u) lo = mid; else hi = mid - 1;
  }
  const int P = lo;
  long rem = u - blk_units_before_row(P, T);
  const int sP = (T - (P << kOrdShift)) < kOrdB ? (T - (P << kOrdShift)) : kOrdB;
  const long triP = (long)sP * (sP + 1) / 2;
  if (rem < triP) {                       // diagonal super-block: row-major triangle of sP tiles
    int li, lj;
    unit_to_tiles(rem, sP, li, lj);
    I = (P << kOrdShift) + li; J = (P << kOrdShift) + lj;
    return;
  }
  rem -= triP;
  const long per = (long)sP * kOrdB;      // every super-block right of the diagonal but the last is kOrdB wide
  const int q = (int)(rem / per);
  const int Q = P + 1 + q;
  const int wQ = (T - (Q << kOrdShift)) < kOrdB ? (T - (Q << kOrdShift)) : kOrdB;
  const long rem2 = rem - (long)q * per;
  const int li = (int)(rem2 / wQ);
  I = (P << kOrdShift) + li;
  J = (Q << kOrdShift) + (int)(rem2 - (long)li * wQ);
}

__device__ __forceinline__ void unit_next(int order, int T, int& I, int& J) {
  if (order == 0) {
    if (++J == T) { ++I; J = I; }
    return;
  }
  const int P = I >> kOrdShift, Q = J >> kOrdShift;
  const int row_end = ((P + 1) << kOrdShift) < T ? ((P + 1) << kOrdShift) : T;
  const int col_end = ((Q + 1) << kOrdShift) < T ? ((Q + 1) << kOrdShift) : T;
  if (++J < col_end) return;                                     // same row of the same super-block
  if (++I < row_end) { J = (P == Q) ? I : (Q << kOrdShift); return; }   // next row of the same super-block
  if (((Q + 1) << kOrdShift) < T) { I = P << kOrdShift; J = (Q + 1) << kOrdShift; return; }   // next super-block of the super-row
  I = (P + 1) << kOrdShift; J = I;                                // diagonal super-block of the next super-row
}

// XCD-aware numbering of the workgroups (cdna_hip_programming.md, T1): blocks are dealt round-robin over the 8 XCDs,
// so blocks b and b + 8 share an L2; this bijection gives every XCD one CONTIGUOUS eighth of the numbering, i.e. of the
// step range.  A speed choice only: any placement is correct.
__device__ __forceinline__ long xcd_swizzle(long bid, long nwg) {
  const long q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <int KIND, bool WALL, bool PERIODIC>
__global__ __launch_bounds__(64 * kSymWaves) __attribute__((amdgpu_waves_per_eu(kSymWavesPerEu, kSymWavesPerEu))) void sym_kernel(const SymArgs a) {
  __shared__ double2 rec_all[kSymWaves][64 * 3];
  __shared__ double accj_all[kSymWaves][3 * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double2* rec = rec_all[wave];
  double* accj = accj_all[wave];
  const char* rec_bytes = reinterpret_cast<const char*>(rec);

  // Static, exactly balanced schedule: the n_units * 64 rotation steps are cut into gridDim.x * 4 equal
  // contiguous ranges, one per wave; a range may begin and end inside a unit.
  const long w = (a.xcd ? xcd_swizzle(blockIdx.x, gridDim.x) : (long)blockIdx.x) * kSymWaves + wave;
  const long long t_start = a.wave_clock ? wall_clock64() : 0;
  // strided chunks: wave / workgroup `id` takes the step ranges id, id + n, id + 2 n, ... of `spw` steps each (one range when
  // the launch is planned that way: n spw >= the steps of the launch).  Waves that run at the same time then work on
  // NEIGHBOURING ranges whatever the size of the problem -- with the blocked unit order and the XCD-aware numbering
  // that keeps a launch's tile loads in one L2 (profiles/r4_unit_order.txt).
  for (long chunk = w;; chunk += (long)gridDim.x * kSymWaves) {
  long s = a.step_begin + chunk * a.steps_per_wave;
  if (s >= a.step_end) break;
  long s_end = s + a.steps_per_wave;
  if (s_end > a.step_end) s_end = a.step_end;
  // decode the first unit once (row-major upper triangle); later units follow by increment
  int I = 0, J = 0;
  if (s < s_end) unit_seek(a.order, s >> 6, a.n_tiles, I, J);

  int I_cur = -1;
  long i = 0;
  bool vi_ok = false;
  double xi = 0, yi = 0, zi = 1.0, vix = 0, viy = 0, viz = 0;
  Vec3 ui = {0.0, 0.0, 0.0};

  while (s < s_end) {
    const int k0 = (int)(s & 63);
    const long left = s_end - s;
    const int k1 = (left < 64 - k0) ? (int)(k0 + left) : 64;
    s += k1 - k0;

    if (I != I_cur) {
      if (I_cur >= 0 && vi_ok) {   // flush the previous row's accumulator
        __hip_atomic_fetch_add(&a.acc[i], ui.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&a.acc[a.n_pad + i], ui.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&a.acc[2 * a.n_pad + i], ui.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      I_cur = I;
      i = 64L * I + lane;
      vi_ok = i < a.n;
      xi = 1e100; yi = 1e100; zi = 1.0; vix = 0; viy = 0; viz = 0;
      if (vi_ok) {
        const double4 p = a.pos[i];
        xi = p.x; yi = p.y; zi = p.z;
        vix = a.vec[3 * i] * p.w; viy = a.vec[3 * i + 1] * p.w; viz = a.vec[3 * i + 2] * p.w;
      }
      ui.x = 0.0; ui.y = 0.0; ui.z = 0.0;
    }
    // tile J -> this wave's LDS slab (record l = blob 64 J + l), zero its accumulators
    {
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
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // Pseudo-periodic images (mobility_numba.py:170-197): nearest image once, then the 3^d neighbour boxes.
    // M_ji(box b) = M_ij(box -b)^T and the boxes are summed symmetrically, so both directions of every image
    // are still evaluated together.
    const int px = PERIODIC && a.Lx > 0, py = PERIODIC && a.Ly > 0, pz = PERIODIC && a.Lz > 0;
    if (I != J) {
      for (int k = (a.skip_pairs & 1) ? k1 : k0; k < k1; ++k) {
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
        __hip_atomic_fetch_add(&accj[jj], tx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(&accj[64 + jj], ty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(&accj[128 + jj], tz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const long j = 64L * J + lane;
      if (j < a.n && !(a.skip_pairs & 2)) {
        __hip_atomic_fetch_add(&a.acc[j], accj[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&a.acc[a.n_pad + j], accj[64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&a.acc[2 * a.n_pad + j], accj[128 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      // diagonal unit: every ordered pair of the tile once, forward only.  Step 0 is the blob itself: its
      // central-box term is the self term (finalize); its periodic images use the pair formula.
      for (int k = (a.skip_pairs & 1) ? k1 : ((PERIODIC || k0 > 1) ? k0 : 1); k < k1; ++k) {
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
    __builtin_amdgcn_wave_barrier();   // accj / rec are rewritten by the next unit
    if (k1 == 64) {                    // next unit in row-major order
      unit_next(a.order, a.n_tiles, I, J);
    }
  }
  if (I_cur >= 0 && vi_ok) {
    __hip_atomic_fetch_add(&a.acc[i], ui.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&a.acc[a.n_pad + i], ui.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&a.acc[2 * a.n_pad + i], ui.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  }   // chunks
  if (a.wave_clock && lane == 0) {   // stamps go to a buffer nothing else reads
    // placement: HW_ID (reg 4: simd [5:4], cu [11:8], sh [12], se [15:13]) and XCC_ID (reg 20) in the top 24 bits
    const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((16 - 1) << 11));
    const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));
    a.wave_clock[2 * w] = t_start;
    a.wave_clock[2 * w + 1] = (wall_clock64() & 0xffffffffffLL) | ((long long)(hw & 0xffff) << 40) | ((long long)(xcc & 0xf) << 56);
  }
}

// AoS result of a workgroup's 256 blobs through LDS, so that every store instruction writes 2 KB of CONSECUTIVE addresses
// (thread t writes doubles t, t + 256, t + 512 of the workgroup's 768) instead of 8 bytes at a stride of 24: the same
// bytes with a third of the write transactions, and the only way the result can go straight into page-locked host memory
// at a useful rate (rmb_matvec's zero-copy hand-off: partial-line writes over PCIe are 2x slower than a copy from ~4000
// blobs on, profiles/r4_exp_host_staging_rejected.txt).  `tile` = 768 doubles of LDS; base = first blob of the workgroup.
__device__ __forceinline__ void store_aos_coalesced(double* tile, double* out, long base, long n, double x, double y, double z,
                                                    bool valid, bool accumulate) {
  const int t = threadIdx.x;
  if (valid) { tile[3 * t] = x; tile[3 * t + 1] = y; tile[3 * t + 2] = z; }
  __syncthreads();
  const long lim = 3 * n;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const long g = 3 * base + c * 256 + t;
    if (g < lim) out[g] = accumulate ? out[g] + tile[c * 256 + t] : tile[c * 256 + t];
  }
  __syncthreads();          // the tile is reused by the caller's next output
}

template <int KIND, bool WALL>
__global__ __launch_bounds__(256) void sym_finalize_kernel(const SymArgs a) {
  __shared__ double tile[768];
  const long base = (long)blockIdx.x * blockDim.x;
  const long i = base + threadIdx.x;
  const bool valid = i < a.n;
  Vec3 acc = {0.0, 0.0, 0.0};
  double sc = 0.0;
  if (valid) {
    acc.x = a.acc[i]; acc.y = a.acc[a.n_pad + i]; acc.z = a.acc[2 * a.n_pad + i];
    a.acc[i] = 0.0; a.acc[a.n_pad + i] = 0.0; a.acc[2 * a.n_pad + i] = 0.0;   // ready for the next product
    const double4 p = a.pos[i];
    const double b = p.w;
    if (i >= a.self_begin && i < a.self_end)
      self_term<KIND, WALL>(a.k, p.z, a.vec[3 * i] * b, a.vec[3 * i + 1] * b, a.vec[3 * i + 2] * b, 0, 0, 0, acc);
    sc = a.prefactor * b;
  }
  store_aos_coalesced(tile, a.out, base, a.n, acc.x * sc, acc.y * sc, acc.z * sc, valid, a.accumulate != 0);
}


// ---------------------------------------------------------------------------------------------
// Symmetric blob-blob force sweep: F_ij = -F_ji, so each unordered pair is evaluated once
// (one rsqrt + one exp) and applied with opposite signs.  Same tile-pair rotation, LDS accumulation
// and static step schedule as sym_tt_kernel.  multi_bodies/forces_numba.py:12-55 semantics.
// ---------------------------------------------------------------------------------------------

struct SymForceArgs {
  const double4* pos;
  double* acc;          // [3][n_pad], zero on entry, re-zeroed by the finalize kernel
  double* out;          // [n][3]
  long n, n_pad;
  int n_tiles;
  long n_units;
  int order, xcd;       // as SymArgs
  long chunk_steps;     // > 0: steps per strided chunk of a wave; 0: one contiguous range per wave
  double Lx, Ly, Lz, iLx, iLy, iLz;
  double eps_over_b, inv_b, two_a;
  ExpConsts ec;
  const double* radii;  // RADII variant: one radius per blob, contact distance a_i + a_j (forces_numba.py:73-122)
  long step_begin, step_end;   // rotation steps [begin, end) of the n_units * 64 this launch covers (pair shard)
  // Tile culling (uniform radius; open or pseudo-periodic): bounds[T] = (xmin, ymin, zmin, xmax, ymax, zmax) of tile T
  // (tile_bounds_kernel), cull2 = (2a + 750 b)^2.  A tile pair whose boxes are further apart than that holds only
  // pairs with (r - 2a)/b > 750, for which exp underflows to exactly 0 here (exp_nonpositive) and in the reference
  // (exp(-745.2) is the smallest denormal): skipping the unit changes no bit of the result.  nullptr = no culling.
  const double* bounds;
  double cull2;
  // Spatially sorted configuration (rmb_sort.hip): `pos` is then the sorted copy and perm[s] the caller's index of
  // sorted slot s; the finalize kernel writes slot s to out[perm[s]].  nullptr = the caller's order.
  const unsigned* perm;
};

// Lower bound of the squared distance between any blob of tile I and any blob of tile J (wave-uniform: every lane
// reads the same twelve doubles).  Per direction the separations x_j - x_i fill the interval [lo_J - hi_I, hi_J - lo_I];
// in a pseudo-periodic direction (L > 0) the pair force takes the nearest image of every separation
// (d - rint(d/L) L, positions need not lie in one cell), so the interval is first moved by the multiple of L that
// centres it: it then lies inside (-L, L), and |nearest image| over it is smallest at the end nearer to zero -- or
// zero if the interval contains zero or is at least L long.
__device__ __forceinline__ double tile_gap2(const double* bounds, int I, int J, double Lx = 0.0, double Ly = 0.0, double Lz = 0.0) {
  const double* bi = bounds + 6L * I;
  const double* bj = bounds + 6L * J;
  const double L[3] = {Lx, Ly, Lz};
  double g2 = 0.0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    double lo = bj[d] - bi[3 + d], hi = bj[3 + d] - bi[d];
    if (L[d] > 0.0) {
      if (hi - lo >= L[d]) { lo = 0.0; hi = 0.0; }
      else {
        const double shift = __builtin_rint(0.5 * (lo + hi) / L[d]) * L[d];
        lo -= shift; hi -= shift;
      }
    }
    const double g = lo > 0.0 ? lo : (hi < 0.0 ? -hi : 0.0);
    g2 = __builtin_fma(g, g, g2);
  }
  return g2;
}

// bounding box of every 64-blob tile of the packed positions; one wave per tile
// (static: this header is part of two translation units, rmb_sym.hip and rmb_sym32.hip)
static __global__ __launch_bounds__(64) void tile_bounds_kernel(const double4* pos, long n, double* bounds) {
  const long T = blockIdx.x;
  const long i = 64 * T + threadIdx.x;
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  if (i < n) {
    const double4 p = pos[i];
    lo[0] = hi[0] = p.x; lo[1] = hi[1] = p.y; lo[2] = hi[2] = p.z;
  }
#pragma unroll
  for (int d = 0; d < 3; ++d)
    for (int off = 32; off > 0; off >>= 1) {
      lo[d] = fmin(lo[d], __shfl_xor(lo[d], off));
      hi[d] = fmax(hi[d], __shfl_xor(hi[d], off));
    }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { bounds[6 * T + d] = lo[d]; bounds[6 * T + 3 + d] = hi[d]; }
  }
}

// f0(r) dr for one pair; dr = r_j - r_i (minimal image), two_a = contact distance of the pair.
// Returns the force ON i; the force on j is minus it.
template <bool PERIODIC>
__device__ __forceinline__ void pair_force(const SymForceArgs& a, double two_a, double dx, double dy, double dz, double& fx,
                                           double& fy, double& fz) {
  if constexpr (PERIODIC) {
    if (a.Lx > 0) dx = wrap_nearest_pad_safe(dx, a.Lx, a.iLx);
    if (a.Ly > 0) dy = wrap_nearest_pad_safe(dy, a.Ly, a.iLy);
    if (a.Lz > 0) dz = wrap_nearest_pad_safe(dz, a.Lz, a.iLz);
  }
  const double r2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
  const double ir = rsqrt_f64(r2);
  const double r = r2 * ir;
  // far: -(eps/b) exp(-(r-2a)/b) / r ;  near (r <= 2a): -(eps/b) / max(r, 1e-25) = -(eps/b) min(1/r, 1e25)
  // Branch-free: x = min((2a - r)/b, 0) is 0 exactly for r <= 2a (and for r = NaN at coincident points, fmin keeps
  // the number), exp(0) = 1 exactly, and min(1/r, 1e25) = 1/r for every r > 2a -- one expression serves both ranges.
  const double x = fmin((two_a - r) * a.inv_b, 0.0);
  const double e = exp_nonpositive(a.ec, x);
  const double f0 = -a.eps_over_b * (e * fmin(ir, 1e25));
  fx = f0 * dx; fy = f0 * dy; fz = f0 * dz;
}

template <bool PERIODIC, bool RADII = false>
__global__ __launch_bounds__(64 * kSymWaves) void sym_force_kernel(const SymForceArgs a) {
  __shared__ double4 rec_all[kSymWaves][64];
  __shared__ double accj_all[kSymWaves][3 * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double4* rec = rec_all[wave];
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
  double xi = 0, yi = 0, zi = 0, ri = 0;
  double ax = 0, ay = 0, az = 0;
  while (s < s_end) {
    const int k0 = (int)(s & 63);
    const long left = s_end - s;
    const int k1 = (left < 64 - k0) ? (int)(k0 + left) : 64;
    s += k1 - k0;
    if (a.bounds != nullptr && I != J &&
        tile_gap2(a.bounds, I, J, PERIODIC ? a.Lx : 0.0, PERIODIC ? a.Ly : 0.0, PERIODIC ? a.Lz : 0.0) > a.cull2) {
      // every pair of this unit is beyond the range of the exponential: contributes exactly zero
      if (k1 == 64) {
        unit_next(a.order, a.n_tiles, I, J);
      }
      continue;
    }
    if (I != I_cur) {
      if (I_cur >= 0 && vi_ok) {
        __hip_atomic_fetch_add(&a.acc[i], ax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&a.acc[a.n_pad + i], ay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&a.acc[2 * a.n_pad + i], az, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      I_cur = I;
      i = 64L * I + lane;
      vi_ok = i < a.n;
      xi = 1e100; yi = 1e100; zi = 1e100;
      if (vi_ok) { const double4 p = a.pos[i]; xi = p.x; yi = p.y; zi = p.z; }
      if constexpr (RADII) ri = vi_ok ? a.radii[i] : 0.0;
      ax = 0.0; ay = 0.0; az = 0.0;
    }
    {
      const long j = 64L * J + lane;
      double4 p = make_double4(-1e100, -1e100, -1e100, 0.0);
      if (j < a.n) p = a.pos[j];
      if constexpr (RADII) p.w = (j < a.n) ? a.radii[j] : 0.0;   // w is free here: forces use unclamped positions (b = 1)
      rec[lane] = p;
      accj[lane] = 0.0; accj[64 + lane] = 0.0; accj[128 + lane] = 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool diag = (I == J);
    for (int k = (diag && k0 < 1) ? 1 : k0; k < k1; ++k) {
      const int jj = (lane + k) & 63;
      const double4 q = rec[jj];
      double fx, fy, fz;
      pair_force<PERIODIC>(a, RADII ? ri + q.w : a.two_a, q.x - xi, q.y - yi, q.z - zi, fx, fy, fz);
      ax += fx; ay += fy; az += fz;
      if (!diag) {   // wave-uniform
        // the LDS slab collects +f (ds_add_f64 has no negate modifier: -f would cost a v_xor + v_mov per component
        // and step); the sign of the reaction goes into the flush below
        __hip_atomic_fetch_add(&accj[jj], fx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(&accj[64 + jj], fy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(&accj[128 + jj], fz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
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
  if (I_cur >= 0 && vi_ok) {
    __hip_atomic_fetch_add(&a.acc[i], ax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&a.acc[a.n_pad + i], ay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&a.acc[2 * a.n_pad + i], az, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  }   // chunks
}

static __global__ __launch_bounds__(256) void sym_force_finalize_kernel(const SymForceArgs a) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const long o = a.perm ? (long)a.perm[i] : i;
  a.out[3 * o] = a.acc[i]; a.out[3 * o + 1] = a.acc[a.n_pad + i]; a.out[3 * o + 2] = a.acc[2 * a.n_pad + i];
  a.acc[i] = 0.0; a.acc[a.n_pad + i] = 0.0; a.acc[2 * a.n_pad + i] = 0.0;
}

}  // namespace rmb
