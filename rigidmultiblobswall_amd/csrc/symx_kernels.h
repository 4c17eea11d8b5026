// symx_kernels.h -- generic symmetric pair sweep: several blocks of the grand mobility per pair (gfx950, fp64).
//
// sym_kernel (sym_kernels.h) evaluates ONE block (tt, tr, rt or rr) on ONE vector per pass.  The callers of the
// path often need more from the same pair geometry:
//   * u = M_tt f + M_tr tau in one loop -- the reference's K11/K12 (mobility/mobility_pycuda.py:1266-1391,
//     :1394-1512 share the pair geometry between the UF and UT blocks);
//   * the 6N grand-mobility product [u; w] = [[M_tt, M_tr], [M_rt, M_rr]] [f; tau] that the roller integrator
//     applies once per Lanczos iteration as FOUR separate sweeps
//     (quaternion_integrator/quaternion_integrator_rollers.py:1114-1121);
//   * [u; w] = [M_tt; M_rt] f, the two random-finite-difference products of one draw (:1138-1160);
//   * one block (M_tt above all) applied to k vectors at once (solves and Lanczos recursions advanced in lockstep,
//     quaternion_integrator_multi_bodies.py:966-996);
//   * the in-plane products (mobility/mobility_numba.py:291, :690: a row/column mask of the symmetric matrix) and
//     the free-surface product (:1770-1937: the image block P RPY(R) is reciprocal too).
// All of them are symmetric operators on the stacked vector, so every unordered pair is still evaluated once and
// applied to both blobs.  Differences, both inverse square roots, tau, e and the heights are computed once per pair
// (Geom) and shared by all blocks: the grand product costs 2.1x one tt pass (220 against 92 + 77 + 77 + 73 VALU
// instructions per pair) instead of four passes.  OP::NEXTRA adds per-blob scalars to the record (per-blob radii),
// the DET variant stores per-unit partials for a fixed-order reduction instead of atomic flushes.
//
// Skeleton = sym_kernel's: tile pairs (I <= J) of 64 blobs, one wave64 per unit, rotation j = (lane + k) & 63,
// transposed contributions through ds_add_f64 into a per-wave LDS accumulator, static exactly balanced step
// schedule, pair-shard step ranges, global SoA accumulators + finalize.  An operation is a policy class OP:
//   OP::NIN / OP::NOUT   3-vectors per blob going in (LDS record) / coming out (accumulators)
//   OP::pair<WALL, ACC = false>(k, dx,dy,dz, zi,zj, vi, vj, ui, t)   ui += (M_ij v_j) rows,  t = (M_ji v_i) rows (ACC: t +=)
//   OP::self<WALL>(k, zi, vi, ui)                        the i == j term, added once per target in finalize
#pragma once
#include "sym_kernels.h"

namespace rmb {

struct SymXArgs {
  const double4* pos;
  const double* in[4];    // [NIN] source vectors (AoS, 3n)
  const double* extra;    // [n] one extra scalar per blob for operations with OP::NEXTRA = 1 (per-blob radius), else unused
  double* out[4];         // [NOUT] outputs (AoS, 3n)
  double* acc;            // [NOUT][3][n_pad] global SoA accumulators; zero on entry, re-zeroed by finalize
  long n, n_pad;
  int n_tiles;
  long n_units;
  int order, xcd;       // as SymArgs (the deterministic variant runs row-major)
  long step_begin, step_end, steps_per_wave;
  long self_begin, self_end;
  double Lx, Ly, Lz, iLx, iLy, iLz;
  double prefactor;
  int accumulate;         // bit c: finalize adds into out[c] instead of overwriting it
  int in_plane;           // zero the z component of every input on load and of every output on store
  int skip_pairs;         // diagnostics (timing only): 1 = no pair arithmetic, 2 = no flush of the LDS accumulators
  // deterministic variant (DET): whole units per wave; instead of atomics every unit stores its two partial results
  // (64 lanes x 3 NOUT doubles each) into a workspace that symx_det_reduce_kernel sums in a fixed order
  double* part_I;         // [unit - unit_begin][3 NOUT][64]  row-side partial, valid only on the last unit of a row run
  double* part_J;         // [unit - unit_begin][3 NOUT][64]  column-side partial of every off-diagonal unit
  double* det_seg;        // [n_tiles][det_segments][3 NOUT][64]  slice sums of the ordered reduction
  long unit_begin, unit_end, units_per_wave;
  int first_chunk;        // reduce: start from zero instead of the running accumulators
  PairConsts k;
};

// ---------------------------------------------------------------------------------------------
// Operations
// ---------------------------------------------------------------------------------------------

// One block on one vector through the verified pair_sym of sym_kernels.h (used for the in-plane products).
template <int KIND>
struct OpSingle {
  static constexpr int NIN = 1, NOUT = 1;
  static constexpr bool IMAGE_NO_WALL = false;
  template <bool WALL, bool ACC = false>
  static __device__ __forceinline__ void pair(const PairConsts& k, double dx, double dy, double dz, double zi, double zj,
                                              const double* vi, const double* vj, double* ui, double* t) {
    Vec3 u = {ui[0], ui[1], ui[2]};
    pair_sym<KIND, WALL, ACC>(k, dx, dy, dz, zi, zj, vi[0], vi[1], vi[2], vj[0], vj[1], vj[2], u, t[0], t[1], t[2]);
    ui[0] = u.x; ui[1] = u.y; ui[2] = u.z;
  }
  template <bool WALL>
  static __device__ __forceinline__ void self(const PairConsts& k, double zi, const double* vi, double* ui) {
    Vec3 u = {ui[0], ui[1], ui[2]};
    self_term<KIND, WALL>(k, zi, vi[0], vi[1], vi[2], 0.0, 0.0, 0.0, u);
    ui[0] = u.x; ui[1] = u.y; ui[2] = u.z;
  }
};

// u = M_tt f + M_tr tau    (in: f, tau; out: u)   -- K11 / K12
struct OpFusedRow {
  static constexpr int NIN = 2, NOUT = 1;
  template <bool WALL, bool ACC = false>
  static __device__ __forceinline__ void pair(const PairConsts& k, double dx, double dy, double dz, double zi, double zj,
                                              const double* vi, const double* vj, double* ui, double* t) {
    const Geom g = make_geom<WALL>(dx, dy, dz, zi, zj);
    const Rpy p = rpy_coeffs<true, true, false>(k, g);          // one overlap patch for both blocks
    const TTc a = tt_block<WALL>(k, g, zi, zj, p.cF, p.cD);
    tt_apply<WALL, ACC>(a, g, vi, vj, ui, t);
    const CPc C = cpl_block<WALL>(k, g, zi, zj, p.c);
    tr_apply<WALL, true>(C, g, vi + 3, vj + 3, ui, t);
  }
  template <bool WALL>
  static __device__ __forceinline__ void self(const PairConsts& k, double zi, const double* vi, double* ui) {
    Vec3 u = {ui[0], ui[1], ui[2]};
    self_term<KIND_TT_TR, WALL>(k, zi, vi[0], vi[1], vi[2], vi[3], vi[4], vi[5], u);
    ui[0] = u.x; ui[1] = u.y; ui[2] = u.z;
  }
};

// [u; w] = [[M_tt, M_tr], [M_rt, M_rr]] [f; tau]    (in: f, tau; out: u, w)
struct OpGrand {
  static constexpr int NIN = 2, NOUT = 2;
  template <bool WALL, bool ACC = false>
  static __device__ __forceinline__ void pair(const PairConsts& k, double dx, double dy, double dz, double zi, double zj,
                                              const double* vi, const double* vj, double* ui, double* t) {
    const Geom g = make_geom<WALL>(dx, dy, dz, zi, zj);
    const Rpy p = rpy_coeffs<true, true, true>(k, g);
    const TTc a = tt_block<WALL>(k, g, zi, zj, p.cF, p.cD);
    tt_apply<WALL, ACC>(a, g, vi, vj, ui, t);
    const CPc C = cpl_block<WALL>(k, g, zi, zj, p.c);
    tr_apply<WALL, true>(C, g, vi + 3, vj + 3, ui, t);
    rt_apply<WALL, ACC>(C, g, vi, vj, ui + 3, t + 3);
    const RRc b = rr_block<WALL>(k, g, p.rF, p.rD);
    rr_apply<WALL, true>(b, g, vi + 3, vj + 3, ui + 3, t + 3);
  }
  template <bool WALL>
  static __device__ __forceinline__ void self(const PairConsts& k, double zi, const double* vi, double* ui) {
    Vec3 u = {ui[0], ui[1], ui[2]}, w = {ui[3], ui[4], ui[5]};
    self_term<KIND_TT_TR, WALL>(k, zi, vi[0], vi[1], vi[2], vi[3], vi[4], vi[5], u);
    self_term<KIND_RT, WALL>(k, zi, vi[0], vi[1], vi[2], 0.0, 0.0, 0.0, w);
    self_term<KIND_RR, WALL>(k, zi, vi[3], vi[4], vi[5], 0.0, 0.0, 0.0, w);
    ui[0] = u.x; ui[1] = u.y; ui[2] = u.z; ui[3] = w.x; ui[4] = w.y; ui[5] = w.z;
  }
};

// [u; w] = [M_tt; M_rt] f    (in: f; out: u, w)  -- both random-finite-difference products of one draw
struct OpColumnF {
  static constexpr int NIN = 1, NOUT = 2;
  template <bool WALL, bool ACC = false>
  static __device__ __forceinline__ void pair(const PairConsts& k, double dx, double dy, double dz, double zi, double zj,
                                              const double* vi, const double* vj, double* ui, double* t) {
    const Geom g = make_geom<WALL>(dx, dy, dz, zi, zj);
    const Rpy p = rpy_coeffs<true, true, false>(k, g);
    const TTc a = tt_block<WALL>(k, g, zi, zj, p.cF, p.cD);
    tt_apply<WALL, ACC>(a, g, vi, vj, ui, t);
    const CPc C = cpl_block<WALL>(k, g, zi, zj, p.c);
    rt_apply<WALL, ACC>(C, g, vi, vj, ui + 3, t + 3);
  }
  template <bool WALL>
  static __device__ __forceinline__ void self(const PairConsts& k, double zi, const double* vi, double* ui) {
    Vec3 u = {ui[0], ui[1], ui[2]}, w = {ui[3], ui[4], ui[5]};
    self_term<KIND_TT, WALL>(k, zi, vi[0], vi[1], vi[2], 0.0, 0.0, 0.0, u);
    self_term<KIND_RT, WALL>(k, zi, vi[0], vi[1], vi[2], 0.0, 0.0, 0.0, w);
    ui[0] = u.x; ui[1] = u.y; ui[2] = u.z; ui[3] = w.x; ui[4] = w.y; ui[5] = w.z;
  }
};

// One block (tt, tr, rt or rr) applied to K vectors: geometry and coefficients once, contraction K times
// (KIND_TT, K = 2 is sym2_kernel's operation).
template <int KIND, int K>
struct OpKindK {
  static constexpr int NIN = K, NOUT = K;
  template <bool WALL, bool ACC = false>
  static __device__ __forceinline__ void pair(const PairConsts& k, double dx, double dy, double dz, double zi, double zj,
                                              const double* vi, const double* vj, double* ui, double* t) {
    const Geom g = make_geom<WALL>(dx, dy, dz, zi, zj);
    if constexpr (KIND == KIND_TT) {
      const TTc a = tt_coeffs<WALL>(k, g, zi, zj);
#pragma unroll
      for (int v = 0; v < K; ++v) tt_apply<WALL, ACC>(a, g, vi + 3 * v, vj + 3 * v, ui + 3 * v, t + 3 * v);
    } else if constexpr (KIND == KIND_RR) {
      const RRc b = rr_coeffs<WALL>(k, g);
#pragma unroll
      for (int v = 0; v < K; ++v) rr_apply<WALL, ACC>(b, g, vi + 3 * v, vj + 3 * v, ui + 3 * v, t + 3 * v);
    } else {
      const CPc C = cpl_coeffs<WALL>(k, g, zi, zj);
#pragma unroll
      for (int v = 0; v < K; ++v) {
        if constexpr (KIND == KIND_TR) tr_apply<WALL, ACC>(C, g, vi + 3 * v, vj + 3 * v, ui + 3 * v, t + 3 * v);
        else                           rt_apply<WALL, ACC>(C, g, vi + 3 * v, vj + 3 * v, ui + 3 * v, t + 3 * v);
      }
    }
  }
  template <bool WALL>
  static __device__ __forceinline__ void self(const PairConsts& k, double zi, const double* vi, double* ui) {
#pragma unroll
    for (int v = 0; v < K; ++v) {
      Vec3 u = {ui[3 * v], ui[3 * v + 1], ui[3 * v + 2]};
      self_term<KIND, WALL>(k, zi, vi[3 * v], vi[3 * v + 1], vi[3 * v + 2], 0.0, 0.0, 0.0, u);
      ui[3 * v] = u.x; ui[3 * v + 1] = u.y; ui[3 * v + 2] = u.z;
    }
  }
};
template <int K> using OpTTk = OpKindK<KIND_TT, K>;

// Free (stress-free) surface at z = 0: u_i = sum_j [RPY(d) + RPY(R) P] f_j, R = (d_x, d_y, z_i + z_j), P = diag(1,1,-1)
// (mobility_numba.py:1846-1925).  The reversed pair sees R' = (-d_x, -d_y, R_z), so (RPY(R) P)_ji = P RPY(R)_ij:
// reciprocal, both directions from one set of coefficients.  Raw heights (set_positions with wall = 0).
struct OpFreeSurface {
  static constexpr int NIN = 1, NOUT = 1;
  template <bool WALL, bool ACC = false>
  static __device__ __forceinline__ void pair(const PairConsts& k, double dx, double dy, double dz, double zi, double zj,
                                              const double* vi, const double* vj, double* ui, double* t) {
    const Geom g = make_geom<true>(dx, dy, dz, zi, zj);
    const TTc a = tt_coeffs<false>(k, g, zi, zj);
    tt_apply<false, ACC>(a, g, vi, vj, ui, t);
    double cF, cD;
    rpy_tt_coeffs(k, __builtin_fma(g.Rz, g.Rz, g.rho2), g.iR, g.iR2, cF, cD);
    // forward: cF P vj + cD (R . P vj) R ;  reversed: cF P vi + cD (R' . P vi) R'
    const double sj = cD * __builtin_fma(-g.Rz, vj[2], __builtin_fma(dy, vj[1], dx * vj[0]));
    const double si = cD * __builtin_fma(g.Rz, vi[2], __builtin_fma(dy, vi[1], dx * vi[0]));    // = -cD (R' . P vi)
    ui[0] = __builtin_fma(cF, vj[0], ui[0]); ui[0] = __builtin_fma(sj, dx, ui[0]);
    ui[1] = __builtin_fma(cF, vj[1], ui[1]); ui[1] = __builtin_fma(sj, dy, ui[1]);
    ui[2] = __builtin_fma(-cF, vj[2], ui[2]); ui[2] = __builtin_fma(sj, g.Rz, ui[2]);
    t[0] = __builtin_fma(si, dx, __builtin_fma(cF, vi[0], t[0]));
    t[1] = __builtin_fma(si, dy, __builtin_fma(cF, vi[1], t[1]));
    t[2] = __builtin_fma(-si, g.Rz, __builtin_fma(-cF, vi[2], t[2]));
  }
  template <bool WALL>
  static __device__ __forceinline__ void self(const PairConsts& k, double zi, const double* vi, double* ui) {
    Vec3 u = {ui[0], ui[1], ui[2]};
    self_term<KIND_TT_FREE, false>(k, zi, vi[0], vi[1], vi[2], 0.0, 0.0, 0.0, u);
    ui[0] = u.x; ui[1] = u.y; ui[2] = u.z;
  }
};

// Translation mobility of blobs with DIFFERENT radii, sources == targets (the reference's `radii_*` modes call the
// source->target kernel with the same array on both sides, mobility/mobility.py:1369-1374; pair formulas
// mobility_numba.py:1480-1658, restated one-sided in st_kernels.h).  The unbounded part (Zuk et al., three regimes)
// is symmetric in the two radii; of the five wall scalars alpha, beta, eps are symmetric under the exchange of the
// two blobs and gamma <-> delta swap, so the reversed pair costs one more contraction.  vi[3] / vj[3] = radius.
struct STc { double C1, C2, alpha, beta, gamma, delta, eps, rz; };

template <bool WALL>
__device__ __forceinline__ STc st_coeffs(double dx, double dy, double dz, double x3, double y3, double at, double as) {
  STc c;
  const double rho2 = __builtin_fma(dy, dy, dx * dx);
  const double r2 = __builtin_fma(dz, dz, rho2);
  const double a2 = at * at, b2 = as * as, s = a2 + b2;
  const double ir = rsqrt_f64(r2);
  const double ir2 = ir * ir;
  c.C1 = __builtin_fma(s * (1.0 / 3.0), ir2, 1.0) * ir;
  c.C2 = __builtin_fma(-s, ir2, 1.0) * ir2 * ir;
  const double sum = at + as;
  if (__builtin_expect(__any(!(r2 > sum * sum)), 0)) {
    // overlapping blobs (rare): Zuk et al. regimes 2 and 3, evaluated with true divisions
    const double r = (r2 > 0.0) ? sqrt(r2) : 0.0;
    const double dm = (as - at) * (as - at);
    const double r3 = r2 * r;
    const double t = dm + 3.0 * r2, q = dm - r2;
    const double pre = (4.0 / 3.0) / (as * at);
    const double C1m = ((16.0 * sum * r3 - t * t) / (32.0 * r3)) * pre;
    const double C2m = ((3.0 * q * q / (32.0 * r3)) / r2) * pre;
    const bool far = r > sum;
    const bool mid = r > fabs(as - at);
    c.C1 = far ? c.C1 : (mid ? C1m : (4.0 / 3.0) / fmax(at, as));
    c.C2 = far ? c.C2 : (mid ? C2m : 0.0);
  }
  c.rz = x3 + y3;
  if constexpr (WALL) {
    const double rz = c.rz;
    const double R2 = __builtin_fma(rz, rz, rho2);
    const double i1 = rsqrt_f64(R2);
    const double i2 = i1 * i1, i3 = i1 * i2, i5 = i3 * i2, i7 = i5 * i2, i9 = i7 * i2;
    const double ab = a2 * b2, xy = x3 * y3;
    const double m = rz * __builtin_fma(a2, y3, b2 * x3);
    const double ab23 = ab * (2.0 / 3.0);
    const double rz2 = rz * rz;
    c.alpha = __builtin_fma(ab23, __builtin_fma(5.0 * rz2, i7, -i5),
                            __builtin_fma(-2.0 * m, i5, __builtin_fma(__builtin_fma(2.0, xy, s * (1.0 / 3.0)), i3, i1)));
    c.beta = __builtin_fma(ab23, __builtin_fma(-35.0 * rz2, i9, 5.0 * i7),
                           __builtin_fma(10.0 * m, i7, __builtin_fma(-__builtin_fma(6.0, xy, s), i5, i3)));
    const double abz = ab * (20.0 / 3.0) * rz * i7;
    const double dab = 2.0 * (a2 - b2) * i5;
    c.gamma = __builtin_fma(x3, __builtin_fma(-2.0, i3, dab), abz);
    c.delta = __builtin_fma(y3, -__builtin_fma(2.0, i3, dab), abz);
    c.eps = -__builtin_fma(ab * (4.0 / 3.0), i5, __builtin_fma(s * (2.0 / 3.0), i3, i1 + i1));
  } else {
    c.alpha = c.beta = c.gamma = c.delta = c.eps = 0.0;
  }
  return c;
}

// u += M(target <- source) f with the wall scalars (gam, del) of that direction and the image separation
// R = (sx dx, sx dy, rz), sx = +1 forward / -1 reversed (g = (-f_x, -f_y, f_z))
template <bool WALL>
__device__ __forceinline__ void st_apply(const STc& c, double dx, double dy, double dz, double sx, double gam, double del,
                                         const double* f, double* u) {
  const double pxy = __builtin_fma(dy, f[1], dx * f[0]);
  const double cD = c.C2 * __builtin_fma(dz, f[2], pxy);
  if constexpr (!WALL) {
    u[0] = __builtin_fma(c.C1, f[0], u[0]); u[0] = __builtin_fma(cD, dx, u[0]);
    u[1] = __builtin_fma(c.C1, f[1], u[1]); u[1] = __builtin_fma(cD, dy, u[1]);
    u[2] = __builtin_fma(c.C1, f[2], u[2]); u[2] = __builtin_fma(cD, dz, u[2]);
  } else {
    const double Rg = __builtin_fma(c.rz, f[2], -sx * pxy);      // R . g
    const double cR = __builtin_fma(c.beta, Rg, gam * f[2]);
    const double cz = __builtin_fma(del, Rg, c.eps * f[2]);
    const double cFxy = c.C1 - c.alpha, cFz = c.C1 + c.alpha;
    const double cDR = __builtin_fma(sx, cR, cD);                // coefficient of (d_x, d_y)
    u[0] = __builtin_fma(cFxy, f[0], u[0]); u[0] = __builtin_fma(cDR, dx, u[0]);
    u[1] = __builtin_fma(cFxy, f[1], u[1]); u[1] = __builtin_fma(cDR, dy, u[1]);
    u[2] = __builtin_fma(cFz, f[2], u[2]); u[2] = __builtin_fma(cD, dz, u[2]);
    u[2] = __builtin_fma(cR, c.rz, u[2]); u[2] += cz;
  }
}

struct OpRadiiTT {
  static constexpr int NIN = 1, NOUT = 1, NEXTRA = 1;
  template <bool WALL, bool ACC = false>
  static __device__ __forceinline__ void pair(const PairConsts&, double dx, double dy, double dz, double zi, double zj,
                                              const double* vi, const double* vj, double* ui, double* t) {
    const STc c = st_coeffs<WALL>(dx, dy, dz, zi, zj, vi[3], vj[3]);
    st_apply<WALL>(c, dx, dy, dz, 1.0, c.gamma, c.delta, vj, ui);
    if constexpr (!ACC) { t[0] = 0.0; t[1] = 0.0; t[2] = 0.0; }
    st_apply<WALL>(c, dx, dy, dz, -1.0, c.delta, c.gamma, vi, t);   // reversed pair: gamma <-> delta, R' = (-d_x, -d_y, r_z)
  }
  // i == j is an ordinary pair of the formulas (r = 0 is the third Zuk regime; the blob's own wall image)
  template <bool WALL>
  static __device__ __forceinline__ void self(const PairConsts&, double zi, const double* vi, double* ui) {
    const STc c = st_coeffs<WALL>(0.0, 0.0, 0.0, zi, zi, vi[3], vi[3]);
    st_apply<WALL>(c, 0.0, 0.0, 0.0, 1.0, c.gamma, c.delta, vi, ui);
  }
};

// ---------------------------------------------------------------------------------------------
// Skeleton
// ---------------------------------------------------------------------------------------------

// LDS record of one blob: x, y, z, then NIN 3-vectors, in double2 units; an ODD number of double2 keeps the
// per-lane ds_read_b128 of consecutive records conflict-free (48 / 80 / 112 / 144 bytes).
template <int NIN, int NEXTRA = 0> struct SymXRec { static constexpr int nd = 3 + 3 * NIN + NEXTRA; static constexpr int d2 = ((nd + 1) / 2) | 1; };
// OP::NEXTRA (optional, default 0): per-blob scalars that travel with the vectors (vi / vj carry them after the 3 NIN
// vector components)
template <class OP, class = void> struct SymXExtra { static constexpr int value = 0; };
template <class OP> struct SymXExtra<OP, decltype((void)OP::NEXTRA)> { static constexpr int value = OP::NEXTRA; };

// (Measured and dropped: storing every record / accumulator slot twice so that the rotation index lane + k needs no
//  `& 63` wrap removes 2 of the 4 integer instructions per step -- 106 -> 104 -- and changes nothing in time,
//  0.1853 vs 0.1849 ms at 1e4 blobs, 17.17 vs 17.27 ms at 1e5: the integer work already hides under fp64 issue.)
template <class OP, bool WALL, bool PERIODIC, bool DET = false>
__global__ __launch_bounds__(64 * kSymWaves) void symx_kernel(const SymXArgs a) {
  constexpr int NI = OP::NIN, NO = OP::NOUT, NX = SymXExtra<OP>::value;
  constexpr int RD2 = SymXRec<NI, NX>::d2;
  constexpr int RECB = RD2 * 16;
  __shared__ double2 rec_all[kSymWaves][64 * RD2];
  __shared__ double accj_all[kSymWaves][3 * NO * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double2* rec = rec_all[wave];
  double* accj = accj_all[wave];
  const char* rec_bytes = reinterpret_cast<const char*>(rec);

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
  double vi[3 * NI + NX], ui[3 * NO];
#pragma unroll
  for (int c = 0; c < 3 * NI + NX; ++c) vi[c] = 0.0;
#pragma unroll
  for (int c = 0; c < 3 * NO; ++c) ui[c] = 0.0;

  // `u_last` = the unit the row run ends with (DET: its workspace slot receives the row partial)
  auto flush_row = [&](long u_last) {
    if constexpr (DET) {
      double* p = a.part_I + (u_last - a.unit_begin) * (3 * NO * 64) + lane;
#pragma unroll
      for (int c = 0; c < 3 * NO; ++c) p[c * 64] = ui[c];
    } else {
      if (!vi_ok) return;
#pragma unroll
      for (int c = 0; c < 3 * NO; ++c)
        __hip_atomic_fetch_add(&a.acc[(long)c * a.n_pad + i], ui[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };

  while (s < s_end) {
    const int k0 = (int)(s & 63);
    const long left = s_end - s;
    const int k1 = (left < 64 - k0) ? (int)(k0 + left) : 64;
    const long unit = s >> 6;
    s += k1 - k0;

    if (I != I_cur) {
      if (I_cur >= 0) flush_row(unit - 1);
      I_cur = I;
      i = 64L * I + lane;
      vi_ok = i < a.n;
      xi = 1e100; yi = 1e100; zi = 1.0;
#pragma unroll
      for (int c = 0; c < 3 * NI + NX; ++c) vi[c] = 0.0;
      if (vi_ok) {
        const double4 p = a.pos[i];
        xi = p.x; yi = p.y; zi = p.z;
        if constexpr (NX > 0) vi[3 * NI] = a.extra[i];
#pragma unroll
        for (int v = 0; v < NI; ++v) {
          vi[3 * v] = a.in[v][3 * i] * p.w; vi[3 * v + 1] = a.in[v][3 * i + 1] * p.w;
          vi[3 * v + 2] = a.in_plane ? 0.0 : a.in[v][3 * i + 2] * p.w;
        }
      }
#pragma unroll
      for (int c = 0; c < 3 * NO; ++c) ui[c] = 0.0;
    }
    {   // tile J -> this wave's LDS slab (record l = blob 64 J + l), zero its accumulators
      const long j = 64L * J + lane;
      double rd[2 * RD2];
#pragma unroll
      for (int c = 0; c < 2 * RD2; ++c) rd[c] = 0.0;
      rd[0] = -1e100; rd[1] = -1e100; rd[2] = 1.0;
      if (j < a.n) {
        const double4 p = a.pos[j];
        rd[0] = p.x; rd[1] = p.y; rd[2] = p.z;
#pragma unroll
        for (int v = 0; v < NI; ++v) {
          rd[3 + 3 * v] = a.in[v][3 * j] * p.w; rd[4 + 3 * v] = a.in[v][3 * j + 1] * p.w;
          rd[5 + 3 * v] = a.in_plane ? 0.0 : a.in[v][3 * j + 2] * p.w;
        }
        if constexpr (NX > 0) rd[3 + 3 * NI] = a.extra[j];
      }
#pragma unroll
      for (int c = 0; c < RD2; ++c) rec[lane * RD2 + c] = make_double2(rd[2 * c], rd[2 * c + 1]);
#pragma unroll
      for (int c = 0; c < 3 * NO; ++c) accj[c * 64 + lane] = 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const int px = PERIODIC && a.Lx > 0, py = PERIODIC && a.Ly > 0, pz = PERIODIC && a.Lz > 0;
    const bool diag = I == J;
    // diagonal units visit every ordered pair of the tile once (forward direction only); step 0 is the blob
    // itself: its central-box term is the self term (finalize), its periodic images use the pair formula
    int kb = diag ? ((PERIODIC || k0 > 1) ? k0 : 1) : k0;
    if (a.skip_pairs & 1) kb = k1;
    for (int k = kb; k < k1; ++k) {
      const int jj = (lane + k) & 63;
      const double2* r = reinterpret_cast<const double2*>(rec_bytes + jj * RECB);
      double rd[2 * RD2];
#pragma unroll
      for (int c = 0; c < RD2; ++c) { const double2 q = r[c]; rd[2 * c] = q.x; rd[2 * c + 1] = q.y; }
      double dx = xi - rd[0], dy = yi - rd[1], dz = zi - rd[2];
      double t[3 * NO];
      if constexpr (!PERIODIC) {
        OP::template pair<WALL>(a.k, dx, dy, dz, zi, rd[2], vi, rd + 3, ui, t);
      } else {
        if (px) dx = wrap_nearest_pad_safe(dx, a.Lx, a.iLx);
        if (py) dy = wrap_nearest_pad_safe(dy, a.Ly, a.iLy);
        if (pz) dz = wrap_nearest_pad_safe(dz, a.Lz, a.iLz);
#pragma unroll
        for (int c = 0; c < 3 * NO; ++c) t[c] = 0.0;
        for (int bx = -px; bx <= px; ++bx)
          for (int by = -py; by <= py; ++by)
            for (int bz = -pz; bz <= pz; ++bz) {
              if (diag && k == 0 && bx == 0 && by == 0 && bz == 0) continue;
              OP::template pair<WALL, true>(a.k, dx + bx * a.Lx, dy + by * a.Ly, dz + bz * a.Lz, zi, rd[2], vi, rd + 3, ui, t);      // accumulates (fused multiply-adds)
            }
      }
      if (!diag) {   // wave-uniform
#pragma unroll
        for (int c = 0; c < 3 * NO; ++c)
          __hip_atomic_fetch_add(&accj[c * 64 + jj], t[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
    }
    if (!diag) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const long j = 64L * J + lane;
      if constexpr (DET) {
        double* p = a.part_J + (unit - a.unit_begin) * (3 * NO * 64) + lane;
#pragma unroll
        for (int c = 0; c < 3 * NO; ++c) p[c * 64] = accj[c * 64 + lane];
      } else if (j < a.n && !(a.skip_pairs & 2)) {
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
  if (I_cur >= 0) flush_row(((s_end - 1) >> 6));
  }   // chunks
}

// Deterministic reduction of the per-unit partials of one chunk [unit_begin, unit_end) into the running accumulators
// acc[3 NOUT][n_pad].  For tile T the contributions are, in this fixed order: the row partials of row T (units (T, J),
// J ascending, only the slots that end a row run), then the column partials of column T (units (I, T), I ascending).
// Stage 1 (grid n_tiles x det_segments): workgroup (T, s) adds the s-th contiguous slice of that list sequentially,
// one thread per (component, lane), four independent loads in flight; stage 2 adds the slice sums in order.  Chunks
// are processed in order, so the summation order of every output depends only on N and the launch geometry --
// bit-reproducible, unlike the atomic flushes.
template <int NO>
__global__ __launch_bounds__(256) void symx_det_reduce_kernel(const SymXArgs a) {
  const long T = blockIdx.x;
  const int seg = blockIdx.y, S = gridDim.y;
  const int lane = threadIdx.x & 63;
  const int nt = a.n_tiles;
  const long ub = a.unit_begin, ue = a.unit_end, upw = a.units_per_wave;
  auto rowstart = [nt](long I) { return I * nt - I * (I - 1) / 2; };
  int I_lo, I_hi, dummy;
  unit_to_tiles(ub, nt, I_lo, dummy);       // rows that own units of this chunk
  unit_to_tiles(ue - 1, nt, I_hi, dummy);
  const long r0 = rowstart(T), r1 = r0 + (nt - T);
  const long lo = r0 > ub ? r0 : ub, hi = r1 < ue ? r1 : ue;
  const long n_row = hi > lo ? hi - lo : 0;
  const long Ia = I_lo, Ib = (I_hi < T - 1) ? I_hi : T - 1;
  const long n_col = Ib >= Ia ? Ib - Ia + 1 : 0;
  const long len = n_row + n_col;
  const long t0 = len * seg / S, t1 = len * (seg + 1) / S;
  constexpr long SLOT = 3 * NO * 64;
  // value of list entry t for component c (0 when the slot is not a contribution)
  auto entry = [&](long t, int c) -> double {
    if (t < n_row) {
      const long u = lo + t;
      const bool ends_run = (u == r1 - 1) || (u == ue - 1) || ((u - ub + 1) % upw == 0);
      return ends_run ? a.part_I[(u - ub) * SLOT + c * 64 + lane] : 0.0;
    }
    const long I = Ia + (t - n_row);
    const long u = rowstart(I) + (T - I);
    return (u >= ub && u < ue) ? a.part_J[(u - ub) * SLOT + c * 64 + lane] : 0.0;
  };
  for (int c = threadIdx.x >> 6; c < 3 * NO; c += (int)(blockDim.x >> 6)) {
    double acc = 0.0;
    long t = t0;
    for (; t + 4 <= t1; t += 4) {
      const double v0 = entry(t, c), v1 = entry(t + 1, c), v2 = entry(t + 2, c), v3 = entry(t + 3, c);
      acc += v0; acc += v1; acc += v2; acc += v3;
    }
    for (; t < t1; ++t) acc += entry(t, c);
    a.det_seg[((T * S + seg) * 3 * NO + c) * 64 + lane] = acc;
  }
}

template <int NO>
__global__ __launch_bounds__(256) void symx_det_combine_kernel(const SymXArgs a, int S) {
  const long T = blockIdx.x;
  const int lane = threadIdx.x & 63;
  for (int c = threadIdx.x >> 6; c < 3 * NO; c += (int)(blockDim.x >> 6)) {
    double acc = a.first_chunk ? 0.0 : a.acc[(long)c * a.n_pad + 64 * T + lane];
    for (int s = 0; s < S; ++s) acc += a.det_seg[((T * S + s) * 3 * NO + c) * 64 + lane];
    a.acc[(long)c * a.n_pad + 64 * T + lane] = acc;
  }
}

template <class OP, bool WALL>
__global__ __launch_bounds__(256) void symx_finalize_kernel(const SymXArgs a) {
  constexpr int NI = OP::NIN, NO = OP::NOUT, NX = SymXExtra<OP>::value;
  __shared__ double tile[768];
  const long base = (long)blockIdx.x * blockDim.x;
  const long i = base + threadIdx.x;
  const bool valid = i < a.n;
  double u[3 * NO];
#pragma unroll
  for (int c = 0; c < 3 * NO; ++c) u[c] = 0.0;
  double sc = 0.0;
  if (valid) {
    const double4 p = a.pos[i];
    const double b = p.w;
    double vi[3 * NI + NX];
    if constexpr (NX > 0) vi[3 * NI] = a.extra[i];
#pragma unroll
    for (int v = 0; v < NI; ++v) {
      vi[3 * v] = a.in[v][3 * i] * b; vi[3 * v + 1] = a.in[v][3 * i + 1] * b;
      vi[3 * v + 2] = a.in_plane ? 0.0 : a.in[v][3 * i + 2] * b;
    }
#pragma unroll
    for (int c = 0; c < 3 * NO; ++c) {
      u[c] = a.acc[(long)c * a.n_pad + i];
      a.acc[(long)c * a.n_pad + i] = 0.0;   // ready for the next product
    }
    if (i >= a.self_begin && i < a.self_end) OP::template self<WALL>(a.k, p.z, vi, u);
    sc = a.prefactor * b;
  }
#pragma unroll
  for (int o = 0; o < NO; ++o)      // coalesced AoS stores through LDS (store_aos_coalesced, sym_kernels.h)
    store_aos_coalesced(tile, a.out[o], base, a.n, u[3 * o] * sc, u[3 * o + 1] * sc, a.in_plane ? 0.0 : u[3 * o + 2] * sc, valid,
                        (a.accumulate & (1 << o)) != 0);
}

}  // namespace rmb
