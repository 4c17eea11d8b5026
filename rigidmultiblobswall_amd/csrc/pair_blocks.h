// pair_blocks.h -- the four blocks of the pair mobility on a SHARED pair geometry, both directions (gfx950, fp64).
//
// Kernels that evaluate each unordered pair once (sym_kernels.h, symx_kernels.h) need, per pair, M_ij v_j for blob i
// and M_ji v_i for blob j, possibly for several blocks (tt, tr, rt, rr) at once.  Everything that does not depend on
// the vectors -- differences, both inverse square roots, tau, the wall polynomials -- is computed once (Geom and the
// *_coeffs routines); the *_apply routines are the cheap contractions.  Reference formulas:
//   tt  mobility/mobility_numba.py:199-281    tr  :609-684    rt  :998-1071    rr  :1250-1326
#pragma once
#include "pair_ops.h"

namespace rmb {

// ---------------------------------------------------------------------------------------------
// Shared pair geometry and the per-block coefficient / contraction routines.
// vi / vj / ui / t are pointers to 3 consecutive doubles (registers after inlining).
// ACC = false: t is written; ACC = true: t is added to (second and later blocks of a fused operation).
// ---------------------------------------------------------------------------------------------
struct Geom {
  double dx, dy, dz, rho2, r2, ir, ir2;
  double Rz, iR, iR2;     // image separation R = (dx, dy, z_i + z_j); wall / free-surface operations only
};

template <bool IMAGE>
__device__ __forceinline__ Geom make_geom(double dx, double dy, double dz, double zi, double zj) {
  Geom g;
  g.dx = dx; g.dy = dy; g.dz = dz;
  g.rho2 = __builtin_fma(dy, dy, dx * dx);
  g.r2 = __builtin_fma(dz, dz, g.rho2);
  g.ir = rsqrt_f64(g.r2);
  g.ir2 = g.ir * g.ir;
  if constexpr (IMAGE) {
    g.Rz = zi + zj;
    g.iR = rsqrt_f64(__builtin_fma(g.Rz, g.Rz, g.rho2));
    g.iR2 = g.iR * g.iR;
  } else {
    g.Rz = 0.0; g.iR = 0.0; g.iR2 = 0.0;
  }
  return g;
}

// Unbounded (RPY) coefficients of the blocks a kernel asks for, far field + the overlap patch behind ONE wave-uniform
// branch.  An operation that evaluates several blocks of a pair patches them together, so that everything after the
// patch is a single basic block (instruction selection folds negations into source modifiers only within a block:
// with one branch per block the fused operations carried ~10 v_xor / v_mov copies per pair).
//   tt  cF I + cD r r^T       mobility_numba.py:209-239   (overlap: r <= 2a)
//   tr / rt  c (eps r)        :619-644 / :1010-1033       (overlap: r <  2a)
//   rr  rF I + rD r r^T       :1260-1286                  (overlap: r <  2a)
struct Rpy { double cF, cD, c, rF, rD; };

template <bool TT, bool CPL, bool RR>
__device__ __forceinline__ Rpy rpy_coeffs(const PairConsts& k, const Geom& g) {
  Rpy p;
  const double ir3 = g.ir2 * g.ir;
  p.cF = __builtin_fma(k.tt_k1, ir3, g.ir);                     // (1 + 2a^2/(3r^2))/r
  p.cD = __builtin_fma(-k.tt_k2, g.ir2, 1.0) * ir3;             // (1 - 2a^2/r^2)/r^3
  p.c = ir3;
  p.rF = -0.5 * ir3;
  p.rD = 1.5 * ir3 * g.ir2;
  const bool guard = TT ? (g.r2 <= k.four_a2) : (g.r2 < k.four_a2);      // one compare in the loop
  if (__builtin_expect(__any(guard), 0)) {
    const double r = g.r2 * g.ir;
    if constexpr (TT) {
      const bool near = g.r2 <= k.four_a2;
      p.cF = near ? __builtin_fma(-k.tt_n1, r, k.tt_n0) : p.cF;
      p.cD = near ? k.tt_n2 * g.ir : p.cD;
    }
    const bool near = TT ? (g.r2 < k.four_a2) : guard;
    if constexpr (CPL) p.c = near ? __builtin_fma(-k.c_q1, r, k.c_q0) : p.c;
    if constexpr (RR) {
      const double r3 = g.r2 * r;
      p.rF = near ? __builtin_fma(k.rr_m2, r3, __builtin_fma(-k.rr_m1, r, k.rr_m0)) : p.rF;
      p.rD = near ? __builtin_fma(-k.rr_m4, r, k.rr_m3 * g.ir) : p.rD;
    }
  }
  return p;
}

// RPY tt coefficients of an arbitrary separation (the image term of the free-surface operation)
__device__ __forceinline__ void rpy_tt_coeffs(const PairConsts& k, double r2, double ir, double ir2, double& cF, double& cD) {
  const double ir3 = ir2 * ir;
  cF = __builtin_fma(k.tt_k1, ir3, ir);
  cD = __builtin_fma(-k.tt_k2, ir2, 1.0) * ir3;
  if (__builtin_expect(__any(r2 <= k.four_a2), 0)) {
    const double r = r2 * ir;
    const bool near = r2 <= k.four_a2;
    cF = near ? __builtin_fma(-k.tt_n1, r, k.tt_n0) : cF;
    cD = near ? k.tt_n2 * ir : cD;
  }
}

// A pair block of the form (d = (d_x, d_y, d_z), R = (d_x, d_y, R_z); tt and rr with and without wall all have it)
//        | F + P dx dx    P dx dy     Q3 dx |
//   M =  |   P dx dy    F + P dy dy   Q3 dy |          M^T = the same with Q3 <-> Q4
//        |   Q4 dx        Q4 dy       Szz   |
// applied as  p = d_x v_x + d_y v_y,  s = P p + Q3 v_z,  (M v)_xy = F v_xy + s d_xy,  (M v)_z = Q4 p + Szz v_z :
// ten instructions per direction and vector, whatever the block costs to build.
struct BlockM { double F, P, Q3, Q4, Szz; };

// ui += M vj ;  t (+)= M^T vi
template <bool ACC>
__device__ __forceinline__ void block_apply(const BlockM& m, const Geom& g, const double* vi, const double* vj, double* ui, double* t) {
  const double pj = __builtin_fma(g.dy, vj[1], g.dx * vj[0]);
  const double sj = __builtin_fma(m.P, pj, m.Q3 * vj[2]);
  ui[0] = __builtin_fma(m.F, vj[0], ui[0]); ui[0] = __builtin_fma(sj, g.dx, ui[0]);
  ui[1] = __builtin_fma(m.F, vj[1], ui[1]); ui[1] = __builtin_fma(sj, g.dy, ui[1]);
  ui[2] = __builtin_fma(m.Q4, pj, ui[2]); ui[2] = __builtin_fma(m.Szz, vj[2], ui[2]);
  const double pi = __builtin_fma(g.dy, vi[1], g.dx * vi[0]);
  const double si = __builtin_fma(m.P, pi, m.Q4 * vi[2]);
  t[0] = __builtin_fma(si, g.dx, ACC ? __builtin_fma(m.F, vi[0], t[0]) : m.F * vi[0]);
  t[1] = __builtin_fma(si, g.dy, ACC ? __builtin_fma(m.F, vi[1], t[1]) : m.F * vi[1]);
  t[2] = __builtin_fma(m.Szz, vi[2], ACC ? __builtin_fma(m.Q3, pi, t[2]) : m.Q3 * pi);
}

// tt:  M = cF I + cD d d^T  (+ wall: -G1 iR I - G2 iR^3 R R^T + G3 iR^2 R z^T + G4 iR^2 z R^T + G5 iR z z^T with the
// polynomials of wall_tt_from_iR).  Substituting e_z = R_z/|R|, g = z_j/|R|, w = z_i z_j/R^2 into the block entries
// the z_j-dependent parts collapse (the 12 R_z w terms of Q3 cancel):  with s = 1/|R|, q = s^2, T = a^2 q,
// U = R_z^2 q = 1 - rho^2 q, W = z_i z_j q and
//   H  = 1 - 6W + T [ (10U - 2) + T (10 - 70U/3) ]                 (= G2 + 20 T^2/3;  Q3 + Q4 = 2 (cD - s q) d_z)
//   F   = cF - s { 1 + 2W + T [ (2/3 - 2U) + T (10U/3 - 2/3) ] }
//   P   = cD - s q ( H - 20 T^2/3 )
//   Q3  = cD d_z + s q [ 2 z_j + R_z (H - 2) ]
//   Q4  = cD d_z + s q [ 2 z_j - R_z H ]
//   Szz = F + cD d_z^2 + s { 4W - U (1 + 6W) + T [ U (10U - 6) + T (70 U (1 - U)/3 - 8/3) ] }
// 37 instructions for polynomials + block entries (the G-form needs 50; checked against it to rounding, and by every
// parity test).
typedef BlockM TTc;

template <bool WALL>
__device__ __forceinline__ TTc tt_block(const PairConsts& k, const Geom& g, double zi, double zj, double cF, double cD) {
  TTc m;
  if constexpr (WALL) {
    // every fma below has at most one non-inline constant (gfx9 VOP3 reads one SGPR / literal): no v_mov in the loop
    const double s = g.iR, q = g.iR2;
    const double q3 = s * q;
    // Base variables (round 3): Tq = T/3 = a^2 q / 3 and p30 = 30 U - 6 = 6 (5U - 1), so that T^2-terms need ONE multiply
    // (Tc = Tq^2 = T^2/9;  (2/3) T^2 = 6 Tc) and every product below has coefficient one:
    //   2T (5U - 1) = Tq p30,   (2/3) T^2 (5U - 1) = Tc p30,   2T (1/3 - U) = Tq (2 - 6U)
    const double Tq = k.tt_k3 * q;
    const double U = __builtin_fma(-g.rho2, q, 1.0);
    const double om = __builtin_fma(-g.r2, q, 1.0);           // 4W = 4 z_i z_j / R^2 = 1 - r^2/R^2
    const double p30 = __builtin_fma(U, k.c30, -6.0);         // k.c30 from the scalar file, -6 in a loop-invariant VGPR
    const double Tc = Tq * Tq;
    // H = 1 - 6W + 2T (5U - 1) + T^2 (10 - 70U/3),   T^2 (10 - 70U/3) = 6 Tc (15 - 35U) = Tc (48 - 7 p30)
    const double H = __builtin_fma(Tc, __builtin_fma(p30, k.m7, 48.0), __builtin_fma(Tq, p30, __builtin_fma(om, -1.5, 1.0)));
    const double cDdz = cD * g.dz;
    // Q3 = cD d_z + s q (R_z H - 2 z_i): 2 z_i does not change along a lane's row of pairs (hoisted);  Q3 + Q4 = 2 (cD - s q) d_z
    m.Q3 = __builtin_fma(q3, __builtin_fma(g.Rz, H, -(zi + zi)), cDdz);
    m.Q4 = __builtin_fma(__builtin_fma(-q3, g.dz, cDdz), 2.0, -m.Q3);
    m.P = __builtin_fma(-q3, __builtin_fma(Tc, -60.0, H), cD);                     // H - 20 T^2/3
    const double G1 = __builtin_fma(Tc, p30, __builtin_fma(Tq, __builtin_fma(U, k.m6, 2.0), __builtin_fma(om, 0.5, 1.0)));
    m.F = __builtin_fma(-G1, s, cF);
    // Zb = 4W - U (1 + 6W) + T U (10U - 6) + T^2 (70 U (1 - U)/3 - 8/3)  =  (om - 24 Tc) + U (H - 2 - 12 Tq + 120 Tc)
    // (the zz polynomial through H: U^2 enters both with the same coefficient 10T - 35 (2/3) T^2)
    const double Zb = __builtin_fma(U, __builtin_fma(Tc, 120.0, __builtin_fma(Tq, -12.0, H - 2.0)), __builtin_fma(Tc, -24.0, om));
    m.Szz = __builtin_fma(s, Zb, __builtin_fma(cDdz, g.dz, m.F));
  } else {
    m.F = cF; m.P = cD;          // tt_apply<false> contracts these two directly
    m.Q3 = m.Q4 = m.Szz = 0.0;
  }
  return m;
}

template <bool WALL>
__device__ __forceinline__ TTc tt_coeffs(const PairConsts& k, const Geom& g, double zi, double zj) {
  const Rpy p = rpy_coeffs<true, false, false>(k, g);
  return tt_block<WALL>(k, g, zi, zj, p.cF, p.cD);
}

// ui += M_tt,ij vj ;  t (+)= M_tt,ji vi
template <bool WALL, bool ACC>
__device__ __forceinline__ void tt_apply(const TTc& c, const Geom& g, const double* vi, const double* vj, double* ui, double* t) {
  if constexpr (WALL) {
    block_apply<ACC>(c, g, vi, vj, ui, t);
  } else {
    // no wall: F v + cD (d.v) d needs nothing built (tt_coeffs<false> leaves Q3, Q4, Szz unset)
    const double cDj = c.P * __builtin_fma(g.dz, vj[2], __builtin_fma(g.dy, vj[1], g.dx * vj[0]));
    const double cDi = c.P * __builtin_fma(g.dz, vi[2], __builtin_fma(g.dy, vi[1], g.dx * vi[0]));
    ui[0] = __builtin_fma(c.F, vj[0], ui[0]); ui[0] = __builtin_fma(cDj, g.dx, ui[0]);
    ui[1] = __builtin_fma(c.F, vj[1], ui[1]); ui[1] = __builtin_fma(cDj, g.dy, ui[1]);
    ui[2] = __builtin_fma(c.F, vj[2], ui[2]); ui[2] = __builtin_fma(cDj, g.dz, ui[2]);
    t[0] = __builtin_fma(cDi, g.dx, ACC ? __builtin_fma(c.F, vi[0], t[0]) : c.F * vi[0]);
    t[1] = __builtin_fma(cDi, g.dy, ACC ? __builtin_fma(c.F, vi[1], t[1]) : c.F * vi[1]);
    t[2] = __builtin_fma(cDi, g.dz, ACC ? __builtin_fma(c.F, vi[2], t[2]) : c.F * vi[2]);
  }
}

// Coupling blocks tr / rt.  With e = R/|R|, the reference's wall correction of M_tr tau (anchored on the target
// height, mobility_numba.py:646-679) is
//   ( f1 e_y t_z + p t_y - f3 c0 e_x,  -f1 e_x t_z - p t_x - f3 c0 e_y,  (f3 e_z + s) c0 ),   c0 = e_x t_y - e_y t_x
// and the RPY part is c (t x d).  Written on the UNNORMALISED separation (e_x = d_x/|R|, c0 = C0/|R| with
// C0 = d_x t_y - d_y t_x) the two parts share their terms:
//   u_x =  A d_y t_z + B t_y - D C0 d_x      A = 1/|R|^3 - c            (f1 = 1/R^2)
//   u_y = -A d_x t_z - B t_x - D C0 d_y      B = c d_z + p             D = f3 / R^2
//   u_z =  E C0                              E = D R_z + S,  S = s/|R| - c
// 11 instructions per direction instead of 25.  M_rt f (anchored on the source height, :1035-1066) is the transpose:
//   w_x = -(p - c d_z) f_y + K d_y,  w_y = (p - c d_z) f_x - K d_x,  w_z = A (d_x f_y - d_y f_x),  K = D (R.f) + S f_z.
// p, s, f3 depend linearly on the anchoring height g = z/|R|; `_i` is anchored on z_i, `_j` on z_j.  The reversed
// pair sees d' = -d, R' = (-d_x, -d_y, R_z).
struct CPc {
  double c;                    // RPY coupling coefficient: 1/r^3, or 1/(2a^3) - 3r/(16a^4) for overlapping blobs
  double A, B_i, Bjp;          // B_i = c d_z + p_i (forward, anchored on i);  Bjp = p_j - c d_z (reversed pair, anchored on j)
  double D_i, D_j, S_i, S_j, E_i, E_j;
};

template <bool WALL>
__device__ __forceinline__ CPc cpl_block(const PairConsts& k, const Geom& g, double zi, double zj, double c) {
  CPc C;
  C.c = c;
  if constexpr (WALL) {
    // p = (R_z (1 + 2 tau) - 2 z)/|R|^3,  D = f3/R^2 = (10 R_z tau - 6 z)/|R|^5,
    // S = s/|R| - c = [1 + (2 - 20 U) tau]/|R|^3 + 12 R_z z/|R|^5 - c      (z = anchoring height, U = R_z^2/R^2)
    const double q3 = g.iR2 * g.iR, q5 = q3 * g.iR2;
    const double T2 = k.tt_k2 * g.iR2;                          // 2 tau
    const double U = __builtin_fma(-g.rho2, g.iR2, 1.0);
    const double RT2 = g.Rz * T2;                               // 2 R_z tau
    const double RT10 = RT2 * 5.0;
    const double S0 = __builtin_fma(q3, __builtin_fma(__builtin_fma(U, -10.0, 1.0), T2, 1.0), -C.c);
    const double q5R = q5 * g.Rz;
    const double w12 = q5R * 12.0;
    C.D_i = q5 * __builtin_fma(zi, k.m6, RT10); C.D_j = q5 * __builtin_fma(zj, k.m6, RT10);     // k.m6: RT10 is read four times
    C.S_i = __builtin_fma(w12, zi, S0); C.S_j = __builtin_fma(w12, zj, S0);
    // E = D R_z + S = q5 R_z (10 R_z tau + 6 z) + S0   (tr alone needs neither S nor w12)
    C.E_i = __builtin_fma(q5R, __builtin_fma(zi, -k.m6, RT10), S0); C.E_j = __builtin_fma(q5R, __builtin_fma(zj, -k.m6, RT10), S0);
    C.A = q3 - C.c;
    // R_z - 2 z_i = -d_z and R_z - 2 z_j = d_z:  c d_z + p_i = 2 R_z tau/|R|^3 - A d_z,   p_j - c d_z = 2 R_z tau/|R|^3 + A d_z
    const double y2 = q3 * RT2;
    C.B_i = __builtin_fma(-C.A, g.dz, y2);
    C.Bjp = __builtin_fma(C.A, g.dz, y2);
  } else {
    C.A = C.B_i = C.Bjp = C.D_i = C.D_j = C.S_i = C.S_j = C.E_i = C.E_j = 0.0;
  }
  return C;
}

template <bool WALL>
__device__ __forceinline__ CPc cpl_coeffs(const PairConsts& k, const Geom& g, double zi, double zj) {
  return cpl_block<WALL>(k, g, zi, zj, rpy_coeffs<false, true, false>(k, g).c);
}

// RPY part alone (no wall): ui += c (vj x d),  t (+)= -c (vi x d)
template <bool ACC>
__device__ __forceinline__ void coupling_rpy_apply(double c, const Geom& g, const double* vi, const double* vj, double* ui, double* t) {
  ui[0] = __builtin_fma(__builtin_fma(vj[1], g.dz, -vj[2] * g.dy), c, ui[0]);
  ui[1] = __builtin_fma(__builtin_fma(vj[2], g.dx, -vj[0] * g.dz), c, ui[1]);
  ui[2] = __builtin_fma(__builtin_fma(vj[0], g.dy, -vj[1] * g.dx), c, ui[2]);
  const double bx = __builtin_fma(vi[2], g.dy, -vi[1] * g.dz), by = __builtin_fma(vi[0], g.dz, -vi[2] * g.dx),
               bz = __builtin_fma(vi[1], g.dx, -vi[0] * g.dy);
  if constexpr (ACC) { t[0] = __builtin_fma(bx, c, t[0]); t[1] = __builtin_fma(by, c, t[1]); t[2] = __builtin_fma(bz, c, t[2]); }
  else { t[0] = bx * c; t[1] = by * c; t[2] = bz * c; }
}

// tr: ui += M_tr,ij vj (torque of j -> velocity of i, wall part anchored on the TARGET height z_i);  t (+)= M_tr,ji vi
template <bool WALL, bool ACC>
__device__ __forceinline__ void tr_apply(const CPc& C, const Geom& g, const double* vi, const double* vj, double* ui, double* t) {
  if constexpr (!WALL) {
    coupling_rpy_apply<ACC>(C.c, g, vi, vj, ui, t);
  } else {
    const double C0 = __builtin_fma(g.dx, vj[1], -g.dy * vj[0]);
    const double Az = C.A * vj[2];
    const double DC = C.D_i * C0;
    ui[0] = __builtin_fma(Az, g.dy, ui[0]); ui[0] = __builtin_fma(C.B_i, vj[1], ui[0]); ui[0] = __builtin_fma(-DC, g.dx, ui[0]);
    ui[1] = __builtin_fma(-Az, g.dx, ui[1]); ui[1] = __builtin_fma(-C.B_i, vj[0], ui[1]); ui[1] = __builtin_fma(-DC, g.dy, ui[1]);
    ui[2] = __builtin_fma(C.E_i, C0, ui[2]);
    // reversed pair (d' = -d, anchored on z_j): (-A d_y t_z + B' t_y - D C0i d_x,  A d_x t_z - B' t_x - D C0i d_y,  -E C0i)
    const double C0i = __builtin_fma(g.dx, vi[1], -g.dy * vi[0]);
    const double Azi = C.A * vi[2];
    const double DCi = C.D_j * C0i;
    const double b0 = ACC ? __builtin_fma(-Azi, g.dy, t[0]) : -Azi * g.dy;
    const double b1 = ACC ? __builtin_fma(Azi, g.dx, t[1]) : Azi * g.dx;
    t[0] = __builtin_fma(-DCi, g.dx, __builtin_fma(C.Bjp, vi[1], b0));
    t[1] = __builtin_fma(-DCi, g.dy, __builtin_fma(-C.Bjp, vi[0], b1));
    t[2] = ACC ? __builtin_fma(-C.E_j, C0i, t[2]) : -C.E_j * C0i;
  }
}

// rt: ui += M_rt,ij vj (force of j -> angular velocity of i, wall part anchored on the SOURCE height z_j);  t (+)= M_rt,ji vi
template <bool WALL, bool ACC>
__device__ __forceinline__ void rt_apply(const CPc& C, const Geom& g, const double* vi, const double* vj, double* ui, double* t) {
  if constexpr (!WALL) {
    coupling_rpy_apply<ACC>(C.c, g, vi, vj, ui, t);
  } else {
    const double pj = __builtin_fma(g.dy, vj[1], g.dx * vj[0]);
    const double K = __builtin_fma(C.D_j, __builtin_fma(g.Rz, vj[2], pj), C.S_j * vj[2]);
    const double C0 = __builtin_fma(g.dx, vj[1], -g.dy * vj[0]);
    ui[0] = __builtin_fma(-C.Bjp, vj[1], ui[0]); ui[0] = __builtin_fma(K, g.dy, ui[0]);
    ui[1] = __builtin_fma(C.Bjp, vj[0], ui[1]); ui[1] = __builtin_fma(-K, g.dx, ui[1]);
    ui[2] = __builtin_fma(C.A, C0, ui[2]);
    // reversed pair (R' = (-d_x, -d_y, R_z), anchored on z_i)
    const double pi = __builtin_fma(g.dy, vi[1], g.dx * vi[0]);
    const double Ki = __builtin_fma(C.D_i, __builtin_fma(g.Rz, vi[2], -pi), C.S_i * vi[2]);
    const double C0i = __builtin_fma(g.dx, vi[1], -g.dy * vi[0]);
    t[0] = __builtin_fma(-Ki, g.dy, ACC ? __builtin_fma(-C.B_i, vi[1], t[0]) : -C.B_i * vi[1]);
    t[1] = __builtin_fma(Ki, g.dx, ACC ? __builtin_fma(C.B_i, vi[0], t[1]) : C.B_i * vi[0]);
    t[2] = ACC ? __builtin_fma(-C.A, C0i, t[2]) : -C.A * C0i;
  }
}

// rr:  M = cF I + cD d d^T  (+ wall, mobility_numba.py:1292-1321: with e = R/|R|, u = e_z^2
//   (W v)_xy = {(3.5 - 6u) v - (1.5 e.v + 3 e_par.v) e}/|R|^3,   (W v)_z = {(0.5 - 3u) v_z + 1.5 (e.v) e_z}/|R|^3 ).
// On the unnormalised R this is the block form with
//   F = cF + (3.5 - 6U)/|R|^3,  P = cD - 4.5/|R|^5,  Q3 = cD d_z - h,  Q4 = cD d_z + h,  h = 1.5 R_z/|R|^5,
//   Szz = cF + cD d_z^2 + (0.5 - 1.5U)/|R|^3
typedef BlockM RRc;

template <bool WALL>
__device__ __forceinline__ RRc rr_block(const PairConsts& k, const Geom& g, double cF, double cD) {
  RRc m;
  if constexpr (WALL) {
    const double q3 = g.iR2 * g.iR, q5 = q3 * g.iR2;
    const double U = __builtin_fma(-g.rho2, g.iR2, 1.0);
    const double cDdz = cD * g.dz;
    const double q5R = q5 * g.Rz;
    m.F = __builtin_fma(__builtin_fma(U, k.m6, 3.5), q3, cF);
    m.P = __builtin_fma(q5, -4.5, cD);
    m.Q3 = __builtin_fma(q5R, -k.c15, cDdz);                   // k.c15 from the scalar file: cDdz is read twice
    m.Q4 = __builtin_fma(q5R, k.c15, cDdz);
    m.Szz = __builtin_fma(q3, __builtin_fma(U, -1.5, 0.5), __builtin_fma(cDdz, g.dz, cF));
  } else {
    m.F = cF; m.P = cD;          // rr_apply<false> contracts these two directly
    m.Q3 = m.Q4 = m.Szz = 0.0;
  }
  return m;
}

template <bool WALL>
__device__ __forceinline__ RRc rr_coeffs(const PairConsts& k, const Geom& g) {
  const Rpy p = rpy_coeffs<false, false, true>(k, g);
  return rr_block<WALL>(k, g, p.rF, p.rD);
}

// rr: ui += M_rr,ij vj ;  t (+)= M_rr,ji vi
template <bool WALL, bool ACC>
__device__ __forceinline__ void rr_apply(const RRc& c, const Geom& g, const double* vi, const double* vj, double* ui, double* t) {
  tt_apply<WALL, ACC>(c, g, vi, vj, ui, t);     // same block form, same contraction
}

// ---------------------------------------------------------------------------------------------
// One ORDERED pair of kind KIND (one-sided kernels: sweep_kernel, dense builders, diagonal tile units).
// (vx,vy,vz) is the source vector; (wx,wy,wz) the source torque for the fused tt+tr kind
// (mobility/mobility_pycuda.py:1351-1375 evaluates both blocks in one pass).  The coupling kinds use the
// unnormalised-separation algebra above with the reversed direction left to dead-code elimination.
// ---------------------------------------------------------------------------------------------
template <bool TR, bool WALL>
__device__ __forceinline__ void pair_coupling_forward(const PairConsts& k, double dx, double dy, double dz, double zi, double zj,
                                                      double vx, double vy, double vz, Vec3& u) {
  const Geom g = make_geom<WALL>(dx, dy, dz, zi, zj);
  const CPc C = cpl_coeffs<WALL>(k, g, zi, zj);
  const double vi[3] = {0.0, 0.0, 0.0}, vj[3] = {vx, vy, vz};
  double u3[3] = {u.x, u.y, u.z}, t[3];
  if constexpr (TR) tr_apply<WALL, false>(C, g, vi, vj, u3, t);
  else              rt_apply<WALL, false>(C, g, vi, vj, u3, t);
  u.x = u3[0]; u.y = u3[1]; u.z = u3[2];
}

// One-sided tt through the same closed-form block (forward direction only; the reversed one is dead code)
template <bool WALL>
__device__ __forceinline__ void pair_tt_forward(const PairConsts& k, double dx, double dy, double dz, double zi, double zj,
                                                double vx, double vy, double vz, Vec3& u) {
  const Geom g = make_geom<WALL>(dx, dy, dz, zi, zj);
  const TTc c = tt_coeffs<WALL>(k, g, zi, zj);
  const double vi[3] = {0.0, 0.0, 0.0}, vj[3] = {vx, vy, vz};
  double u3[3] = {u.x, u.y, u.z}, t[3];
  tt_apply<WALL, false>(c, g, vi, vj, u3, t);
  u.x = u3[0]; u.y = u3[1]; u.z = u3[2];
}

template <int KIND, bool WALL>
__device__ __forceinline__ void pair_apply(const PairConsts& k, double dx, double dy, double dz, double zi,
                                           double zj, double vx, double vy, double vz, double wx, double wy,
                                           double wz, Vec3& u) {
  const double Rz = zi + zj;
  if constexpr (KIND == KIND_TT) pair_tt_forward<WALL>(k, dx, dy, dz, zi, zj, vx, vy, vz, u);
  if constexpr (KIND == KIND_TR) pair_coupling_forward<true, WALL>(k, dx, dy, dz, zi, zj, vx, vy, vz, u);
  if constexpr (KIND == KIND_RT) pair_coupling_forward<false, WALL>(k, dx, dy, dz, zi, zj, vx, vy, vz, u);
  if constexpr (KIND == KIND_RR) pair_rr<WALL>(k, dx, dy, dz, Rz, vx, vy, vz, u);
  if constexpr (KIND == KIND_TT_TR) {
    pair_tt_forward<WALL>(k, dx, dy, dz, zi, zj, vx, vy, vz, u);
    pair_coupling_forward<true, WALL>(k, dx, dy, dz, zi, zj, wx, wy, wz, u);
  }
  if constexpr (KIND == KIND_TT_FREE) {
    // free (stress-free) surface at z = 0: RPY(d) f + RPY(R) (f_x, f_y, -f_z), R = (d_x, d_y, z_i + z_j)
    // (mobility/mobility_numba.py:1846-1925; image block added with the z column negated, :1915-1923)
    pair_tt<false>(k, dx, dy, dz, Rz, zj, vx, vy, vz, u);
    pair_tt<false>(k, dx, dy, Rz, Rz, zj, vx, vy, -vz, u);
  }
}

}  // namespace rmb
