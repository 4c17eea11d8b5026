// pair_ops.h -- matrix-free pair operators for the blob-level mobility products (gfx950, fp64).
//
// WHAT (reference semantics): for a target blob i and a source blob j the reference builds a
// 3x3 block M_ij in units of the hydrodynamic radius a and multiplies it by the source vector
//   tt  mobility/mobility_numba.py:199-281   RPY + Swan-Brady wall correction, u = M_tt f
//   tr  mobility/mobility_numba.py:609-684   u = M_tr tau  (wall part anchored on the TARGET height)
//   rt  mobility/mobility_numba.py:998-1071  w = M_rt f
//   rr  mobility/mobility_numba.py:1250-1326 w = M_rr tau
//
// HOW (ours): the 3x3 block is never formed.  Every block is a combination of I, d d^T, R R^T,
// z R^T, R z^T, z z^T and cross products, so M_ij v is evaluated directly as
//   (scalar) v + (scalar) d + (scalar) R + (scalar) z
// with d = r_i - r_j and R = (d_x, d_y, z_i + z_j) (the image separation).  Positions stay
// UNSCALED (differences of the caller's coordinates are exact); every power of `a` is folded
// into uniform constants (PairConsts, SGPR-resident) and the common prefactor 1/(8 pi eta) is
// applied once per target in the epilogue.  The wall height ratio h = z/R_z only ever appears
// multiplied by e_z = R_z/|R|, so g = h e_z = z/|R| and no division by R_z is needed.  The two
// inverse square roots per pair are v_rsq_f64 + one cubic (Halley) correction step, not the
// 30-instruction IEEE sqrt+divide sequence.  The near-field (r < 2a, overlapping blobs) RPY
// branch is taken per wave only when some lane needs it.
//
// A wall-tt pair costs 95 VALU instructions in the one-sided sweep and 106 per UNORDERED pair (53 per ordered pair)
// in the symmetric kernel of sym_kernels.h (tools/isa_stats.py); the reference's as-written count is 211 flops.
#pragma once
#include <hip/hip_runtime.h>

namespace rmb {

// Launch geometry shared by the kernels and the host-side plans (rmb_plan.hip): a workgroup of every pair kernel is
// 4 waves = 256 threads, one wave per SIMD.
constexpr int kWaves = 4;            // one-sided sweeps (matvec_kernels.h, st_kernels.h, aux_kernels.h)
constexpr int kBlock = 64 * kWaves;
constexpr int kSymWaves = 4;         // symmetric kernels (sym*_kernels.h)
constexpr int kSymWavesPerEu = 4;    // register budget of sym_kernel / sym2_kernel: 4 waves per SIMD (the launch plan relies on it)

enum Kind : int { KIND_TT = 0, KIND_TR = 1, KIND_RT = 2, KIND_RR = 3, KIND_TT_TR = 4, KIND_TT_FREE = 5, KIND_COUNT = 6 };

// Uniform constants (host-computed from the blob radius a; see make_pair_consts in rmb_plan.hip).
struct PairConsts {
  double a2;       // a^2                (tau = a^2/R^2 in the wall corrections)
  double four_a2;  // (2a)^2             far/near switch on r^2
  // tt
  double tt_k1;    // 2 a^2 / 3
  double tt_k2;    // 2 a^2
  double tt_k3;    // a^2 / 3            (Tq = T/3 of the wall tt block, pair_blocks.h)
  double tt_n0;    // 4/(3a)             also the unbounded self term
  double tt_n1;    // 3/(8 a^2)
  double tt_n2;    // 1/(8 a^2)
  // rr
  double rr_m0;    // 1/a^3              also the unbounded self term
  double rr_m1;    // 27/(32 a^4)
  double rr_m2;    // 5/(64 a^6)
  double rr_m3;    // 9/(32 a^4)
  double rr_m4;    // 3/(64 a^6)
  // tr / rt
  double c_q0;     // 1/(2 a^3)
  double c_q1;     // 3/(16 a^4)
  // plain numbers handed over as kernel arguments (SGPRs) so that an fma with two non-inline constants reads one
  // from the scalar file and keeps the other in a loop-invariant VGPR (gfx9 VOP3: one SGPR / literal per instruction)
  double m7;       // -7
  double m6;       // -6
  double c15;      // 1.5
  double c30;      // 30
};

struct Vec3 { double x, y, z; };

// 1/sqrt(x): hardware seed (v_rsq_f64, ~2^-23 relative) + one third-order correction
//   y' = y (1 + e/2 + 3 e^2/8),  e = 1 - x y^2      -> relative error O(e^3), i.e. full fp64.
__device__ __forceinline__ double rsqrt_f64(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double t = x * y;
  double e = __builtin_fma(-t, y, 1.0);
  double p = __builtin_fma(0.375, e, 0.5) * e;
  return __builtin_fma(y, p, y);
}

// Wall-correction polynomials of the tt block.  With tau = a^2/R^2, u = e_z^2, g = z_j/|R| (= h e_z),
// w = g (e_z - g), q6 = e_z (e_z - g):
//   G1 = (1+2w) + tau [ (2/3)(1-3u) + tau (2/3)(5u-1) ]
//   G2 = (1-6w) + tau [ 2(5u-1) + tau (10/3)(1-7u) ]
//   G3 = 2g(1-6 q6) + 4 e_z tau [ (5u-1) + (5/3) tau (2-7u) ]
//   G4 = 2g - (20/3) e_z tau^2
//   G5 = -[ 4g^2 + 4 tau ( u + tau (2/3 - 5u) ) ]
// and  W = -G1 iR I - G2 iR e e^T + G3 iR e z^T + G4 iR z e^T + G5 iR z z^T      (times 1/(8 pi eta)).
// Every fma below has at most one non-inline constant, so the compiler needs no VGPR copies of
// constants (v_mov_b64 + v_fmac) in the pair loop.
struct WallTT { double iR, iR2, G1, G2, G3, G4, G5; };

// iR = 1/|R| supplied by the caller (kernels that evaluate several blocks per pair share it)
__device__ __forceinline__ WallTT wall_tt_from_iR(const PairConsts& k, double Rz, double iR, double zj) {
  WallTT W;
  W.iR = iR;
  W.iR2 = W.iR * W.iR;
  const double tau = k.a2 * W.iR2;
  const double ez = Rz * W.iR;
  const double g = zj * W.iR;
  const double uu = ez * ez;
  const double dd = ez - g;
  const double w = g * dd;
  const double q6 = ez * dd;
  const double tau23 = tau * (2.0 / 3.0);
  const double tau53 = tau * (5.0 / 3.0);
  const double tau2 = tau + tau;
  const double tau4 = tau * 4.0;
  const double p5 = __builtin_fma(uu, 5.0, -1.0);     // 5u - 1
  const double p3 = __builtin_fma(uu, -3.0, 1.0);     // 1 - 3u
  const double p7 = __builtin_fma(uu, -7.0, 1.0);     // 1 - 7u
  const double p7b = p7 + 1.0;                        // 2 - 7u
  const double g2 = g + g;
  W.G1 = __builtin_fma(tau23, __builtin_fma(tau, p5, p3), __builtin_fma(w, 2.0, 1.0));
  W.G2 = __builtin_fma(tau2, __builtin_fma(tau53, p7, p5), __builtin_fma(w, -6.0, 1.0));
  const double e4 = ez * tau4;                        // 4 e_z tau
  W.G3 = __builtin_fma(e4, __builtin_fma(tau53, p7b, p5), g2 * __builtin_fma(q6, -6.0, 1.0));
  W.G4 = __builtin_fma(-e4, tau53, g2);
  const double ut = uu * tau;
  W.G5 = -__builtin_fma(tau4, __builtin_fma(ut, -5.0, tau23) + uu, g2 * g2);
  return W;
}

__device__ __forceinline__ WallTT wall_tt_factors(const PairConsts& k, double rho2, double Rz, double zj) {
  return wall_tt_from_iR(k, Rz, rsqrt_f64(__builtin_fma(Rz, Rz, rho2)), zj);
}

// ---------------------------------------------------------------------------------------------
// tt:  u += [RPY_tt(d) + W_tt(d_x, d_y, R_z; z_j)] f          (common prefactor 1/(8 pi eta))
// ---------------------------------------------------------------------------------------------
template <bool WALL>
__device__ __forceinline__ void pair_tt(const PairConsts& k, double dx, double dy, double dz, double Rz,
                                        double zj, double fx, double fy, double fz, Vec3& u) {
  const double rho2 = __builtin_fma(dy, dy, dx * dx);
  const double r2 = __builtin_fma(dz, dz, rho2);
  const double ir = rsqrt_f64(r2);
  const double ir2 = ir * ir;
  const double pxy = __builtin_fma(dy, fy, dx * fx);
  const double df = __builtin_fma(dz, fz, pxy);
  // far field: (1/r) [ (1 + 2a^2/(3r^2)) I + (1 - 2a^2/r^2) d d^T / r^2 ]
  double cF = __builtin_fma(k.tt_k1, ir2, 1.0) * ir;
  double cD = __builtin_fma(-k.tt_k2, ir2, 1.0) * ir2 * ir;
  if (__builtin_expect(__any(r2 <= k.four_a2), 0)) {
    // overlapping blobs: (4/(3a) - 3r/(8a^2)) I + d d^T/(8 a^2 r)
    const double r = r2 * ir;
    const bool near = r2 <= k.four_a2;
    cF = near ? __builtin_fma(-k.tt_n1, r, k.tt_n0) : cF;
    cD = near ? k.tt_n2 * ir : cD;
  }
  cD *= df;
  if constexpr (!WALL) {
    u.x = __builtin_fma(cF, fx, u.x); u.x = __builtin_fma(cD, dx, u.x);
    u.y = __builtin_fma(cF, fy, u.y); u.y = __builtin_fma(cD, dy, u.y);
    u.z = __builtin_fma(cF, fz, u.z); u.z = __builtin_fma(cD, dz, u.z);
  } else {
    const WallTT W = wall_tt_factors(k, rho2, Rz, zj);
    const double E = W.iR * __builtin_fma(Rz, fz, pxy);      // e.f
    // coefficient of R: iR^2 (G3 f_z - G2 e.f);  of z: iR (G5 f_z + G4 e.f)
    const double cR = __builtin_fma(W.G3, fz, -W.G2 * E) * W.iR2;
    const double cb = __builtin_fma(W.G5, fz, W.G4 * E) * W.iR;
    cF = __builtin_fma(-W.G1, W.iR, cF);
    const double cDR = cD + cR;
    u.x = __builtin_fma(cF, fx, u.x); u.x = __builtin_fma(cDR, dx, u.x);
    u.y = __builtin_fma(cF, fy, u.y); u.y = __builtin_fma(cDR, dy, u.y);
    u.z = __builtin_fma(cF, fz, u.z); u.z = __builtin_fma(cD, dz, u.z);
    u.z = __builtin_fma(cR, Rz, u.z); u.z += cb;
  }
}

// (tr / rt: the coupling blocks live in pair_blocks.h -- cpl_coeffs / tr_apply / rt_apply on the unnormalised
//  separation, shared by the one-sided and the symmetric kernels.)

// rr:  w += [RPY_rr(d) + W_rr] tau
template <bool WALL>
__device__ __forceinline__ void pair_rr(const PairConsts& k, double dx, double dy, double dz, double Rz,
                                        double vx, double vy, double vz, Vec3& u) {
  const double rho2 = __builtin_fma(dy, dy, dx * dx);
  const double r2 = __builtin_fma(dz, dz, rho2);
  const double ir = rsqrt_f64(r2);
  const double ir2 = ir * ir;
  const double ir3 = ir2 * ir;
  const double pxy = __builtin_fma(dy, vy, dx * vx);
  const double dv = __builtin_fma(dz, vz, pxy);
  double cF = -0.5 * ir3;
  double cD = 1.5 * ir3 * ir2;
  if (__builtin_expect(__any(r2 < k.four_a2), 0)) {
    const double r = r2 * ir;
    const double r3 = r2 * r;
    const bool near = r2 < k.four_a2;
    cF = near ? __builtin_fma(k.rr_m2, r3, __builtin_fma(-k.rr_m1, r, k.rr_m0)) : cF;
    cD = near ? __builtin_fma(-k.rr_m4, r, k.rr_m3 * ir) : cD;
  }
  cD *= dv;
  if constexpr (!WALL) {
    u.x = __builtin_fma(cF, vx, u.x); u.x = __builtin_fma(cD, dx, u.x);
    u.y = __builtin_fma(cF, vy, u.y); u.y = __builtin_fma(cD, dy, u.y);
    u.z = __builtin_fma(cF, vz, u.z); u.z = __builtin_fma(cD, dz, u.z);
  } else {
    const double R2 = __builtin_fma(Rz, Rz, rho2);
    const double iR = rsqrt_f64(R2);
    const double iR2 = iR * iR;
    const double iR3 = iR2 * iR;
    const double uu = Rz * Rz * iR2;
    const double Rv = __builtin_fma(Rz, vz, pxy);
    // x,y: iR3 { (3.5 - 6u) v - (1.5 E + 3 E_par) e },  z: iR3 { (0.5 - 3u) v_z + 1.5 E e_z },  e = R iR
    const double iR5 = iR3 * iR2;
    const double cFxy = __builtin_fma(__builtin_fma(-6.0, uu, 3.5), iR3, cF);
    const double cFz = __builtin_fma(__builtin_fma(-3.0, uu, 0.5), iR3, cF);
    const double cRxy = -iR5 * __builtin_fma(1.5, Rv, 3.0 * pxy);
    const double cRz = 1.5 * iR5 * Rv;
    const double cDR = cD + cRxy;
    u.x = __builtin_fma(cFxy, vx, u.x); u.x = __builtin_fma(cDR, dx, u.x);
    u.y = __builtin_fma(cFxy, vy, u.y); u.y = __builtin_fma(cDR, dy, u.y);
    u.z = __builtin_fma(cFz, vz, u.z); u.z = __builtin_fma(cD, dz, u.z);
    u.z = __builtin_fma(cRz, Rz, u.z);
  }
}

// ---------------------------------------------------------------------------------------------
// Self terms (i == j, central box), added once per target in the epilogue.
//   tt: mobility_numba.py:203-208,:245-252   rr: :1254-1259,:1295-1300
//   tr: :653-657                             rt: :1040-1044
// ---------------------------------------------------------------------------------------------
template <int KIND, bool WALL>
__device__ __forceinline__ void self_term(const PairConsts& k, double zi, double vx, double vy, double vz,
                                          double wx, double wy, double wz, Vec3& u) {
  (void)wx; (void)wy; (void)wz; (void)vz;
  if constexpr (KIND == KIND_TT || KIND == KIND_TT_TR) {
    double sxx = k.tt_n0, szz = k.tt_n0;
    if constexpr (WALL) {
      const double iz = 1.0 / zi, iz2 = iz * iz, iz3 = iz2 * iz, iz5 = iz3 * iz2;
      const double a4 = k.a2 * k.a2;
      sxx -= (9.0 * iz - 2.0 * k.a2 * iz3 + a4 * iz5) / 12.0;
      szz -= (9.0 * iz - 4.0 * k.a2 * iz3 + a4 * iz5) / 6.0;
    }
    u.x = __builtin_fma(sxx, vx, u.x); u.y = __builtin_fma(sxx, vy, u.y); u.z = __builtin_fma(szz, vz, u.z);
  }
  if constexpr (KIND == KIND_TT_FREE) {
    // self: 4/(3a) I plus the blob's own image, evaluated with the pair formula at R = (0, 0, 2 z_i)
    u.x = __builtin_fma(k.tt_n0, vx, u.x); u.y = __builtin_fma(k.tt_n0, vy, u.y); u.z = __builtin_fma(k.tt_n0, vz, u.z);
    const double two_z = zi + zi;
    pair_tt<false>(k, 0.0, 0.0, two_z, two_z, zi, vx, vy, -vz, u);
  }
  if constexpr (KIND == KIND_RR) {
    double sxx = k.rr_m0, szz = k.rr_m0;
    if constexpr (WALL) {
      const double iz = 1.0 / zi, iz3 = iz * iz * iz;
      sxx -= 0.3125 * iz3;
      szz -= 0.125 * iz3;
    }
    u.x = __builtin_fma(sxx, vx, u.x); u.y = __builtin_fma(sxx, vy, u.y); u.z = __builtin_fma(szz, vz, u.z);
  }
  if constexpr (WALL && KIND == KIND_TR) {
    const double iz = 1.0 / zi, iz2 = iz * iz;
    const double c = 0.125 * k.a2 * iz2 * iz2;
    u.x = __builtin_fma(c, vy, u.x); u.y = __builtin_fma(-c, vx, u.y);
  }
  if constexpr (WALL && KIND == KIND_TT_TR) {
    const double iz = 1.0 / zi, iz2 = iz * iz;
    const double c = 0.125 * k.a2 * iz2 * iz2;
    u.x = __builtin_fma(c, wy, u.x); u.y = __builtin_fma(-c, wx, u.y);
  }
  if constexpr (WALL && KIND == KIND_RT) {
    const double iz = 1.0 / zi, iz2 = iz * iz;
    const double c = 0.125 * k.a2 * iz2 * iz2;
    u.x = __builtin_fma(-c, vy, u.x); u.y = __builtin_fma(c, vx, u.y);
  }
}

// exp(x) for x <= 0 in fp64 without ocml: n = rint(x log2 e), t = x - n ln2 (two-piece ln2), degree-13 Taylor
// polynomial in t (|t| <= ln2/2: truncation 4e-18), result = ldexp(p, n) (v_ldexp_f64 underflows gradually to 0).
// The coefficients travel as kernel arguments (SGPRs) so that every Horner step is one v_fma_f64 with a scalar
// addend; ocml's exp keeps them in VGPRs and pays a v_mov_b64 per step (seen in the ISA of the first version).
struct ExpConsts { double log2e, ln2_hi, ln2_lo, c[12]; };   // c[k] = 1/(k+2)!, k = 0..11

__device__ __forceinline__ double exp_nonpositive(const ExpConsts& e, double x) {
  x = fmax(x, -750.0);
  const double n = __builtin_rint(x * e.log2e);
  double t = __builtin_fma(n, -e.ln2_hi, x);
  t = __builtin_fma(n, -e.ln2_lo, t);
  double p = e.c[11];
#pragma unroll
  for (int k = 10; k >= 0; --k) p = __builtin_fma(p, t, e.c[k]);
  p = __builtin_fma(p, t, 1.0);      // 1 + t (1 + t q)
  p = __builtin_fma(p, t, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)n);
}

}  // namespace rmb
