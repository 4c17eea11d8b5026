/*
 * oracle_mobility.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the reference's blob-level pairwise operators, used
 * as the parity checker for the HIP path (tests/, __graft_entry__.smoke(),
 * and bench.py's cpu_baseline leg).  Nothing under rigidmultiblobswall_amd/
 * may link, import or call this file: the product path is the HIP library and
 * it fails loudly when that library is missing.
 *
 * Parity status: PINNED.  Every function here is checked against golden
 * vectors produced by running the reference's own Python
 * (mobility/mobility_numba.py under a numba identity stub, and the dense
 * builders in mobility/mobility.py) in the build container; see
 * oracle/gen_golden.py and tests/test_oracle_golden.py.
 *
 * Structure (ours, not the reference's): one generic driver loop
 * `matvec_driver` that walks targets x image boxes x sources and calls a
 * per-kind "pair block" routine that fills a 3x3 block in hydrodynamic-radius
 * units.  The reference instead has one 100+-line function per kind with the
 * loops and the algebra inlined; the arithmetic of every block follows the
 * reference formulas as written, cited per function below.
 *
 * All citations are relative to /root/reference/.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_PI 3.14159265358979323846264338327950288

/* kinds: which 3x3 block of the grand mobility */
enum { KIND_TT = 0, KIND_TR = 1, KIND_RT = 2, KIND_RR = 3 };

/* 3x3 block, row-major: m[3*row+col] */
typedef struct { double m[9]; } block3;

/* ------------------------------------------------------------------------ */
/* Unbounded RPY blocks.  Arguments are the separation in units of a.        */
/* ------------------------------------------------------------------------ */

/* translation-translation RPY block.
 * mobility/mobility_numba.py:202-239 (self 4/3; far c1 = 1 + 2/(3 r^2),
 * c2 = (1 - 2/r^2)/r^2, all times 1/r; near c1 = 4/3 (1 - 9r/32),
 * c2 = 4/3 * 3/(32 r)). */
static void rpy_tt(double rx, double ry, double rz, int self, block3 *B) {
  double Mxx, Mxy, Mxz, Myy, Myz, Mzz;
  if (self) {
    Mxx = 4.0 / 3.0; Mxy = 0; Mxz = 0; Myy = Mxx; Myz = 0; Mzz = Mxx;
  } else {
    double r2 = rx * rx + ry * ry + rz * rz;
    double r = sqrt(r2);
    double invr = 1.0 / r;
    double invr2 = invr * invr;
    if (r > 2) {
      double c1 = 1.0 + 2.0 / (3.0 * r2);
      double c2 = (1.0 - 2.0 * invr2) * invr2;
      Mxx = (c1 + c2 * rx * rx) * invr;
      Mxy = (c2 * rx * ry) * invr;
      Mxz = (c2 * rx * rz) * invr;
      Myy = (c1 + c2 * ry * ry) * invr;
      Myz = (c2 * ry * rz) * invr;
      Mzz = (c1 + c2 * rz * rz) * invr;
    } else {
      double c1 = (4.0 / 3.0) * (1.0 - 0.28125 * r);
      double c2 = (4.0 / 3.0) * 0.09375 * invr;
      Mxx = c1 + c2 * rx * rx;
      Mxy = c2 * rx * ry;
      Mxz = c2 * rx * rz;
      Myy = c1 + c2 * ry * ry;
      Myz = c2 * ry * rz;
      Mzz = c1 + c2 * rz * rz;
    }
  }
  B->m[0] = Mxx; B->m[1] = Mxy; B->m[2] = Mxz;
  B->m[3] = Mxy; B->m[4] = Myy; B->m[5] = Myz;
  B->m[6] = Mxz; B->m[7] = Myz; B->m[8] = Mzz;
}

/* translation-rotation / rotation-translation RPY block (antisymmetric).
 * mobility/mobility_numba.py:612-644 (tr) and :1001-1033 (rt): far
 * eps.r / r^3, near (1/2)(1 - 3r/8) eps.r; self 0. */
static void rpy_coupling(double rx, double ry, double rz, int self, block3 *B) {
  double Mxy, Mxz, Myz;
  if (self) {
    Mxy = 0; Mxz = 0; Myz = 0;
  } else {
    double r2 = rx * rx + ry * ry + rz * rz;
    double r = sqrt(r2);
    double r3 = r2 * r;
    double invr3 = 1.0 / r3;
    if (r >= 2) {
      Mxy = rz * invr3;
      Mxz = -ry * invr3;
      Myz = rx * invr3;
    } else {
      double c1 = 0.5 * (1.0 - 0.375 * r);
      Mxy = c1 * rz;
      Mxz = -c1 * ry;
      Myz = c1 * rx;
    }
  }
  B->m[0] = 0;    B->m[1] = Mxy;  B->m[2] = Mxz;
  B->m[3] = -Mxy; B->m[4] = 0;    B->m[5] = Myz;
  B->m[6] = -Mxz; B->m[7] = -Myz; B->m[8] = 0;
}

/* rotation-rotation RPY block.
 * mobility/mobility_numba.py:1253-1290: self 1; far (-1/2 + 3/2 rr/r^2)/r^3;
 * near c1 = 1 - 27r/32 + 5r^3/64, c2 = 9/(32r) - 3r/64. */
static void rpy_rr(double rx, double ry, double rz, int self, block3 *B) {
  double Mxx, Mxy, Mxz, Myy, Myz, Mzz;
  if (self) {
    Mxx = 1.0; Mxy = 0; Mxz = 0; Myy = 1.0; Myz = 0; Mzz = 1.0;
  } else {
    double r2 = rx * rx + ry * ry + rz * rz;
    double r = sqrt(r2);
    double r3 = r2 * r;
    double invr = 1.0 / r;
    double invr2 = 1.0 / r2;
    double invr3 = 1.0 / r3;
    if (r >= 2) {
      double c1 = -0.5;
      double c2 = 1.5 * invr2;
      Mxx = (c1 + c2 * rx * rx) * invr3;
      Mxy = (c2 * rx * ry) * invr3;
      Mxz = (c2 * rx * rz) * invr3;
      Myy = (c1 + c2 * ry * ry) * invr3;
      Myz = (c2 * ry * rz) * invr3;
      Mzz = (c1 + c2 * rz * rz) * invr3;
    } else {
      double c1 = 1.0 - 0.84375 * r + 0.078125 * r3;
      double c2 = 0.28125 * invr - 0.046875 * r;
      Mxx = c1 + c2 * rx * rx;
      Mxy = c2 * rx * ry;
      Mxz = c2 * rx * rz;
      Myy = c1 + c2 * ry * ry;
      Myz = c2 * ry * rz;
      Mzz = c1 + c2 * rz * rz;
    }
  }
  B->m[0] = Mxx; B->m[1] = Mxy; B->m[2] = Mxz;
  B->m[3] = Mxy; B->m[4] = Myy; B->m[5] = Myz;
  B->m[6] = Mxz; B->m[7] = Myz; B->m[8] = Mzz;
}

/* ------------------------------------------------------------------------ */
/* Single-wall (Swan & Brady) corrections, added in place to the RPY block.  */
/* (rx, ry) in-plane separation / a, Rz = (z_i + z_j)/a, h = height / a of   */
/* the blob the formula is anchored on (source for tt/rt/rr, target for tr). */
/* ------------------------------------------------------------------------ */

/* mobility/mobility_numba.py:241-276 */
static void wall_tt(double rx, double ry, double Rz, double hj, int self, block3 *B) {
  double *M = B->m;
  if (self) {
    double invZi = 1.0 / hj;
    double invZi3 = invZi * invZi * invZi;
    double invZi5 = invZi3 * invZi * invZi;
    M[0] += -(9.0 * invZi - 2.0 * invZi3 + invZi5) / 12.0;
    M[4] += -(9.0 * invZi - 2.0 * invZi3 + invZi5) / 12.0;
    M[8] += -(9.0 * invZi - 4.0 * invZi3 + invZi5) / 6.0;
  } else {
    double h_hat = hj / Rz;
    double invR = 1.0 / sqrt(rx * rx + ry * ry + Rz * Rz);
    double ex = rx * invR, ey = ry * invR, ez = Rz * invR;
    double invR3 = invR * invR * invR;
    double invR5 = invR3 * invR * invR;
    double ez2 = ez * ez;
    double fact1 = -(3.0 * (1.0 + 2.0 * h_hat * (1.0 - h_hat) * ez2) * invR + 2.0 * (1.0 - 3.0 * ez2) * invR3 - 2.0 * (1.0 - 5.0 * ez2) * invR5) / 3.0;
    double fact2 = -(3.0 * (1.0 - 6.0 * h_hat * (1.0 - h_hat) * ez2) * invR - 6.0 * (1.0 - 5.0 * ez2) * invR3 + 10.0 * (1.0 - 7.0 * ez2) * invR5) / 3.0;
    double fact3 = ez * (3.0 * h_hat * (1.0 - 6.0 * (1.0 - h_hat) * ez2) * invR - 6.0 * (1.0 - 5.0 * ez2) * invR3 + 10.0 * (2.0 - 7.0 * ez2) * invR5) * 2.0 / 3.0;
    double fact4 = ez * (3.0 * h_hat * invR - 10.0 * invR5) * 2.0 / 3.0;
    double fact5 = -(3.0 * h_hat * h_hat * ez2 * invR + 3.0 * ez2 * invR3 + (2.0 - 15.0 * ez2) * invR5) * 4.0 / 3.0;
    M[0] += fact1 + fact2 * ex * ex;
    M[1] += fact2 * ex * ey;
    M[2] += fact2 * ex * ez + fact3 * ex;
    M[3] += fact2 * ey * ex;
    M[4] += fact1 + fact2 * ey * ey;
    M[5] += fact2 * ey * ez + fact3 * ey;
    M[6] += fact2 * ez * ex + fact4 * ex;
    M[7] += fact2 * ez * ey + fact4 * ey;
    M[8] += fact1 + fact2 * ez * ez + fact3 * ez + fact4 * ez + fact5;
  }
}

/* mobility/mobility_numba.py:646-679.  Caller passes the NEGATED in-plane
 * separation and the TARGET height, as the reference does at :648-651. */
static void wall_tr(double rx, double ry, double Rz, double hi, int self, block3 *B) {
  double *M = B->m;
  if (self) {
    double invZi = 1.0 / hi;
    double invZi4 = invZi * invZi * invZi * invZi;
    M[1] -= -invZi4 * 0.125;
    M[3] -= invZi4 * 0.125;
  } else {
    double h_hat = hi / Rz;
    double invR = 1.0 / sqrt(rx * rx + ry * ry + Rz * Rz);
    double invR2 = invR * invR;
    double invR4 = invR2 * invR2;
    double ex = rx * invR, ey = ry * invR, ez = Rz * invR;
    double fact1 = invR2;
    double fact2 = (6.0 * h_hat * ez * ez * invR2 + (1.0 - 10.0 * ez * ez) * invR4) * 2.0;
    double fact3 = -ez * (3.0 * h_hat * invR2 - 5.0 * invR4) * 2.0;
    double fact4 = -ez * (h_hat * invR2 - invR4) * 2.0;
    M[0] -= -fact3 * ex * ey;
    M[1] -= -fact1 * ez + fact3 * ex * ex - fact4;
    M[2] -= fact1 * ey;
    M[3] -= fact1 * ez - fact3 * ey * ey + fact4;
    M[4] -= fact3 * ex * ey;
    M[5] -= -fact1 * ex;
    M[6] -= -fact1 * ey - fact2 * ey - fact3 * ey * ez;
    M[7] -= fact1 * ex + fact2 * ex + fact3 * ex * ez;
  }
}

/* mobility/mobility_numba.py:1035-1066 (source height) */
static void wall_rt(double rx, double ry, double Rz, double hj, int self, block3 *B) {
  double *M = B->m;
  if (self) {
    double invZi = 1.0 / hj;
    double invZi4 = invZi * invZi * invZi * invZi;
    M[1] += -invZi4 * 0.125;
    M[3] += invZi4 * 0.125;
  } else {
    double h_hat = hj / Rz;
    double invR = 1.0 / sqrt(rx * rx + ry * ry + Rz * Rz);
    double invR2 = invR * invR;
    double invR4 = invR2 * invR2;
    double ex = rx * invR, ey = ry * invR, ez = Rz * invR;
    double fact1 = invR2;
    double fact2 = (6.0 * h_hat * ez * ez * invR2 + (1.0 - 10.0 * ez * ez) * invR4) * 2.0;
    double fact3 = -ez * (3.0 * h_hat * invR2 - 5.0 * invR4) * 2.0;
    double fact4 = -ez * (h_hat * invR2 - invR4) * 2.0;
    M[0] -= -fact3 * ex * ey;
    M[1] -= fact1 * ez - fact3 * ey * ey + fact4;
    M[2] -= -fact1 * ey - fact2 * ey - fact3 * ey * ez;
    M[3] -= -fact1 * ez + fact3 * ex * ex - fact4;
    M[4] -= fact3 * ex * ey;
    M[5] -= fact1 * ex + fact2 * ex + fact3 * ex * ez;
    M[6] -= fact1 * ey;
    M[7] -= -fact1 * ex;
  }
}

/* mobility/mobility_numba.py:1292-1321 */
static void wall_rr(double rx, double ry, double Rz, double hj, int self, block3 *B) {
  double *M = B->m;
  if (self) {
    double invZi = 1.0 / hj;
    double invZi3 = invZi * invZi * invZi;
    M[0] += -invZi3 * 0.3125;
    M[4] += -invZi3 * 0.3125;
    M[8] += -invZi3 * 0.125;
  } else {
    double invR = 1.0 / sqrt(rx * rx + ry * ry + Rz * Rz);
    double invR3 = invR * invR * invR;
    double ex = rx * invR, ey = ry * invR, ez = Rz * invR;
    double fact1 = ((1.0 - 6.0 * ez * ez) * invR3) * 0.5;
    double fact2 = -(9.0 * invR3) / 6.0;
    double fact3 = 3.0 * invR3 * ez;
    double fact4 = 3.0 * invR3;
    M[0] += fact1 + fact2 * ex * ex + fact4 * ey * ey;
    M[1] += (fact2 - fact4) * ex * ey;
    M[2] += fact2 * ex * ez;
    M[3] += (fact2 - fact4) * ex * ey;
    M[4] += fact1 + fact2 * ey * ey + fact4 * ex * ex;
    M[5] += fact2 * ey * ez;
    M[6] += fact2 * ez * ex + fact3 * ex;
    M[7] += fact2 * ez * ey + fact3 * ey;
    M[8] += fact1 + fact2 * ez * ez + fact3 * ez;
  }
}

/* ------------------------------------------------------------------------ */
/* Pseudo-periodic nearest-image wrap.                                       */
/* mobility/mobility_numba.py:184-192: r -= int(r/L + 0.5*sgn(r))*L; C's     */
/* (long) cast truncates toward zero exactly as Python's int().              */
/* ------------------------------------------------------------------------ */
static inline double wrap_nearest(double r, double L) {
  int sg = (r > 0) - (r < 0);
  return r - (double)((long)(r / L + 0.5 * sg)) * L;
}

/* One (target i, source j, image box) block of kind `kind`, in units of a.
 * `in_plane` zeroes the z rows/cols as mobility_numba.py:291-435 / :690-828. */
static void pair_block(int kind, int wall, int in_plane, double rx, double ry, double rz,
                       double zi, double zj, double inva, int self, block3 *B) {
  double sx = rx * inva, sy = ry * inva, sz = rz * inva;
  double Rz = (zi + zj) * inva;
  switch (kind) {
    case KIND_TT:
      rpy_tt(sx, sy, sz, self, B);
      if (in_plane) { B->m[2] = B->m[5] = B->m[6] = B->m[7] = B->m[8] = 0; }
      if (wall) {
        block3 W; memset(&W, 0, sizeof W);
        wall_tt(sx, sy, Rz, zj * inva, self, &W);
        if (in_plane) { W.m[2] = W.m[5] = W.m[6] = W.m[7] = W.m[8] = 0; }
        for (int k = 0; k < 9; ++k) B->m[k] += W.m[k];
      }
      break;
    case KIND_TR:
      rpy_coupling(sx, sy, sz, self, B);
      if (in_plane) { B->m[2] = B->m[5] = B->m[6] = B->m[7] = 0; }
      if (wall) {
        block3 W; memset(&W, 0, sizeof W);
        wall_tr(-sx, -sy, Rz, zi * inva, self, &W);
        if (in_plane) { W.m[2] = W.m[5] = W.m[6] = W.m[7] = 0; }
        for (int k = 0; k < 9; ++k) B->m[k] += W.m[k];
      }
      break;
    case KIND_RT:
      rpy_coupling(sx, sy, sz, self, B);
      if (wall) wall_rt(sx, sy, Rz, zj * inva, self, B);
      break;
    default: /* KIND_RR */
      rpy_rr(sx, sy, sz, self, B);
      if (wall) wall_rr(sx, sy, Rz, zj * inva, self, B);
      break;
  }
}

/* ------------------------------------------------------------------------ */
/* Generic driver: out[3i..] = norm * sum_boxes sum_j block(i,j) . v[3j..]    */
/* Loop nest order (target, boxX, boxY, boxZ, source) and accumulation order */
/* follow mobility/mobility_numba.py:166-281.                                */
/* ------------------------------------------------------------------------ */
static int matvec_driver(int kind, int wall, int in_plane, long N, const double *r,
                         const double *v, double eta, double a, const double *L, double *out) {
  if (N < 0 || !out) return 1;
  if (N == 0) return 0;
  if (!r || !v || !L) return 1;
  const double inva = 1.0 / a;
  double norm;
  if (kind == KIND_TT) norm = 1.0 / (8.0 * ORACLE_PI * eta * a);
  else if (kind == KIND_RR) norm = 1.0 / (8.0 * ORACLE_PI * eta * a * a * a);
  else norm = 1.0 / (8.0 * ORACLE_PI * eta * a * a);
  const int px = L[0] > 0, py = L[1] > 0, pz = L[2] > 0;

#pragma omp parallel for schedule(dynamic, 16)
  for (long i = 0; i < N; ++i) {
    double ux = 0, uy = 0, uz = 0;
    const double xi = r[3 * i], yi = r[3 * i + 1], zi = r[3 * i + 2];
    for (int bx = -px; bx <= px; ++bx)
      for (int by = -py; by <= py; ++by)
        for (int bz = -pz; bz <= pz; ++bz)
          for (long j = 0; j < N; ++j) {
            double rx = xi - r[3 * j], ry = yi - r[3 * j + 1], rz = zi - r[3 * j + 2];
            if (px) rx = wrap_nearest(rx, L[0]) + bx * L[0];
            if (py) ry = wrap_nearest(ry, L[1]) + by * L[1];
            if (pz) rz = wrap_nearest(rz, L[2]) + bz * L[2];
            const int self = (i == j) && bx == 0 && by == 0 && bz == 0;
            block3 B;
            pair_block(kind, wall, in_plane, rx, ry, rz, zi, r[3 * j + 2], inva, self, &B);
            const double vx = v[3 * j], vy = v[3 * j + 1], vz = v[3 * j + 2];
            ux += (B.m[0] * vx + B.m[1] * vy + B.m[2] * vz) * norm;
            uy += (B.m[3] * vx + B.m[4] * vy + B.m[5] * vz) * norm;
            uz += (B.m[6] * vx + B.m[7] * vy + B.m[8] * vz) * norm;
          }
    out[3 * i] = ux; out[3 * i + 1] = uy; out[3 * i + 2] = uz;
  }
  return 0;
}

/* Exported entry point.  kind in {0 tt, 1 tr, 2 rt, 3 rr}; wall in {0,1};
 * in_plane only meaningful for tt/tr with wall=1.
 * Positions must already be height-clamped by the caller when wall=1 (the
 * reference does that in the Python wrapper, mobility/mobility.py:1150-1163). */
int oracle_mobility_matvec(int kind, int wall, int in_plane, long N, const double *r,
                           const double *v, double eta, double a, const double *L, double *out) {
  if (kind < 0 || kind > 3) return 2;
  return matvec_driver(kind, wall, in_plane, N, r, v, eta, a, L, out);
}

/* Subset variant for full-size spot checks: only the listed targets are
 * evaluated (all N sources each).  out has 3*n_targets entries. */
int oracle_mobility_matvec_targets(int kind, int wall, long N, const double *r, const double *v,
                                   double eta, double a, const double *L, long n_targets,
                                   const long *targets, double *out) {
  if (kind < 0 || kind > 3) return 2;
  const double inva = 1.0 / a;
  double norm;
  if (kind == KIND_TT) norm = 1.0 / (8.0 * ORACLE_PI * eta * a);
  else if (kind == KIND_RR) norm = 1.0 / (8.0 * ORACLE_PI * eta * a * a * a);
  else norm = 1.0 / (8.0 * ORACLE_PI * eta * a * a);
  const int px = L[0] > 0, py = L[1] > 0, pz = L[2] > 0;
#pragma omp parallel for schedule(dynamic, 1)
  for (long t = 0; t < n_targets; ++t) {
    const long i = targets[t];
    double ux = 0, uy = 0, uz = 0;
    const double xi = r[3 * i], yi = r[3 * i + 1], zi = r[3 * i + 2];
    for (int bx = -px; bx <= px; ++bx)
      for (int by = -py; by <= py; ++by)
        for (int bz = -pz; bz <= pz; ++bz)
          for (long j = 0; j < N; ++j) {
            double rx = xi - r[3 * j], ry = yi - r[3 * j + 1], rz = zi - r[3 * j + 2];
            if (px) rx = wrap_nearest(rx, L[0]) + bx * L[0];
            if (py) ry = wrap_nearest(ry, L[1]) + by * L[1];
            if (pz) rz = wrap_nearest(rz, L[2]) + bz * L[2];
            const int self = (i == j) && bx == 0 && by == 0 && bz == 0;
            block3 B;
            pair_block(kind, wall, 0, rx, ry, rz, zi, r[3 * j + 2], inva, self, &B);
            const double vx = v[3 * j], vy = v[3 * j + 1], vz = v[3 * j + 2];
            ux += (B.m[0] * vx + B.m[1] * vy + B.m[2] * vz) * norm;
            uy += (B.m[3] * vx + B.m[4] * vy + B.m[5] * vz) * norm;
            uz += (B.m[6] * vx + B.m[7] * vy + B.m[8] * vz) * norm;
          }
    out[3 * t] = ux; out[3 * t + 1] = uy; out[3 * t + 2] = uz;
  }
  return 0;
}

/* Free (stress-free) surface at z = 0: M = RPY(d) + RPY(R) P, P = diag(1,1,-1), R = (d_x, d_y, z_i+z_j).
 * mobility/mobility_numba.py:1770-1937: the image block is evaluated with the PAIR formula also for
 * i == j (:1890-1912) and added with its z COLUMN negated (:1915-1923).  No height clamp. */
int oracle_free_surface_matvec(long N, const double *r, const double *v, double eta, double a, const double *L,
                               double *out) {
  if (N < 0 || !out) return 1;
  if (N == 0) return 0;
  const double inva = 1.0 / a;
  const double norm = 1.0 / (8.0 * ORACLE_PI * eta * a);
  const int px = L[0] > 0, py = L[1] > 0, pz = L[2] > 0;
#pragma omp parallel for schedule(dynamic, 16)
  for (long i = 0; i < N; ++i) {
    double ux = 0, uy = 0, uz = 0;
    const double xi = r[3 * i], yi = r[3 * i + 1], zi = r[3 * i + 2];
    for (int bx = -px; bx <= px; ++bx)
      for (int by = -py; by <= py; ++by)
        for (int bz = -pz; bz <= pz; ++bz)
          for (long j = 0; j < N; ++j) {
            double rx = xi - r[3 * j], ry = yi - r[3 * j + 1], rz = zi - r[3 * j + 2];
            if (px) rx = wrap_nearest(rx, L[0]) + bx * L[0];
            if (py) ry = wrap_nearest(ry, L[1]) + by * L[1];
            if (pz) rz = wrap_nearest(rz, L[2]) + bz * L[2];
            const int self = (i == j) && bx == 0 && by == 0 && bz == 0;
            block3 B, I;
            rpy_tt(rx * inva, ry * inva, rz * inva, self, &B);
            rpy_tt(rx * inva, ry * inva, (zi + r[3 * j + 2]) * inva, 0, &I);
            const double vx = v[3 * j], vy = v[3 * j + 1], vz = v[3 * j + 2];
            ux += ((B.m[0] + I.m[0]) * vx + (B.m[1] + I.m[1]) * vy + (B.m[2] - I.m[2]) * vz) * norm;
            uy += ((B.m[3] + I.m[3]) * vx + (B.m[4] + I.m[4]) * vy + (B.m[5] - I.m[5]) * vz) * norm;
            uz += ((B.m[6] + I.m[6]) * vx + (B.m[7] + I.m[7]) * vy + (B.m[8] - I.m[8]) * vz) * norm;
          }
    out[3 * i] = ux; out[3 * i + 1] = uy; out[3 * i + 2] = uz;
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* Source -> target translation mobility with per-blob radii (K13).          */
/* mobility/mobility_numba.py:1480-1658 (and its pure-python twin             */
/* mobility/mobility.py:830-945): unbounded part after Zuk et al. (three       */
/* regimes in r vs a_t + a_s and |a_t - a_s|), wall part = -T(R) P with        */
/* P = diag(1,1,-1), R = target - image(source), T the sum of five tensors.    */
/* Restated here with explicit 3x3 outer products (the reference writes out    */
/* all 9 entries of every term by hand).  Heights already clamped by caller.   */
/* ------------------------------------------------------------------------ */
static void outer3(const double *u, const double *v, double c, double *T) {
  for (int l = 0; l < 3; ++l)
    for (int m = 0; m < 3; ++m) T[3 * l + m] += c * u[l] * v[m];
}

int oracle_source_target_matvec(long Ns, const double *src, const double *rad_s, long Nt, const double *tgt,
                                const double *rad_t, const double *force, double eta, const double *L, int wall,
                                double *out) {
  if (Ns < 0 || Nt < 0 || !out) return 1;
  const double pref = 1.0 / (8.0 * ORACLE_PI * eta);
  const int px = L[0] > 0, py = L[1] > 0, pz = L[2] > 0;
  static const double zhat[3] = {0.0, 0.0, 1.0};
#pragma omp parallel for schedule(dynamic, 8)
  for (long i = 0; i < Nt; ++i) {
    const double at = rad_t[i];
    double u[3] = {0, 0, 0};
    for (int bx = -px; bx <= px; ++bx)
      for (int by = -py; by <= py; ++by)
        for (int bz = -pz; bz <= pz; ++bz)
          for (long j = 0; j < Ns; ++j) {
            const double as = rad_s[j];
            double d[3] = {tgt[3 * i] - src[3 * j], tgt[3 * i + 1] - src[3 * j + 1], tgt[3 * i + 2] - src[3 * j + 2]};
            if (px) d[0] = wrap_nearest(d[0], L[0]) + bx * L[0];
            if (py) d[1] = wrap_nearest(d[1], L[1]) + by * L[1];
            if (pz) d[2] = wrap_nearest(d[2], L[2]) + bz * L[2];
            double M[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            /* unbounded part, mobility_numba.py:1556-1578 */
            const double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
            const double r = sqrt(r2);
            double C1, C2;
            if (r > at + as) {
              C1 = (1 + (as * as + at * at) / (3 * r2)) / r;
              C2 = ((1 - (as * as + at * at) / r2) / r2) / r;
            } else if (r > fabs(as - at)) {
              const double r3 = r2 * r, dm = (as - at) * (as - at);
              C1 = ((16 * (as + at) * r3 - (dm + 3 * r2) * (dm + 3 * r2)) / (32 * r3)) * (4.0 / 3.0) / (as * at);
              C2 = ((3 * (dm - r2) * (dm - r2) / (32 * r3)) / r2) * (4.0 / 3.0) / (as * at);
            } else {
              C1 = (4.0 / 3.0) / (at > as ? at : as);
              C2 = 0;
            }
            M[0] = M[4] = M[8] = C1;
            outer3(d, d, C2, M);
            if (wall == 2) {
              /* stress-free surface at z = 0 (mobility_numba.py:2040-2079): the same three-regime tensor evaluated at the
               * mirror image R = (dx, dy, z_t + z_s) -- raw heights, not wrapped in z -- added with its z column negated */
              const double Rv[3] = {d[0], d[1], tgt[3 * i + 2] + src[3 * j + 2]};
              const double q2 = Rv[0] * Rv[0] + Rv[1] * Rv[1] + Rv[2] * Rv[2];
              const double q = sqrt(q2);
              double D1, D2;
              if (q > at + as) {
                D1 = (1 + (as * as + at * at) / (3 * q2)) / q;
                D2 = ((1 - (as * as + at * at) / q2) / q2) / q;
              } else if (q > fabs(as - at)) {
                const double q3 = q2 * q, dm = (as - at) * (as - at);
                D1 = ((16 * (as + at) * q3 - (dm + 3 * q2) * (dm + 3 * q2)) / (32 * q3)) * (4.0 / 3.0) / (as * at);
                D2 = ((3 * (dm - q2) * (dm - q2) / (32 * q3)) / q2) * (4.0 / 3.0) / (as * at);
              } else {
                D1 = (4.0 / 3.0) / (at > as ? at : as);
                D2 = 0;
              }
              double T[9] = {D1, 0, 0, 0, D1, 0, 0, 0, D1};
              outer3(Rv, Rv, D2, T);
              for (int l = 0; l < 3; ++l) {
                M[3 * l] += T[3 * l];
                M[3 * l + 1] += T[3 * l + 1];
                M[3 * l + 2] -= T[3 * l + 2];
              }
            } else if (wall) {
              /* mobility_numba.py:1591-1644 */
              const double x3 = tgt[3 * i + 2], y3 = src[3 * j + 2];
              const double R[3] = {d[0], d[1], x3 + y3};
              const double R2 = R[0] * R[0] + R[1] * R[1] + R[2] * R[2];
              const double Rn = sqrt(R2), R3 = R2 * Rn, R5 = R3 * R2, R7 = R5 * R2, R9 = R7 * R2;
              const double a2 = at * at, b2 = as * as, rz = R[2];
              double T[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
              /* G_ab(R) */
              for (int k = 0; k < 3; ++k) T[4 * k] += (1 + (b2 + a2) / (3.0 * R2)) / Rn;
              outer3(R, R, (1 - (b2 + a2) / R2) / R2 / Rn, T);
              /* 2 (-J/r - R x3^T/r3 - y3 R^T/r3 + x3 y3 (I/r3 - 3 R R^T/r5)) */
              T[8] += -2.0 / Rn;
              outer3(R, zhat, -2.0 * x3 / R3, T);
              outer3(zhat, R, -2.0 * y3 / R3, T);
              for (int k = 0; k < 3; ++k) T[4 * k] += 2.0 * x3 * y3 / R3;
              outer3(R, R, -6.0 * x3 * y3 / R5, T);
              /* (2 a2/3)(-J/r3 + 3 R Rz^T/r5 - y3 (3 rz I/r5 + 3 z R^T/r5 + 3 R z^T/r5 - 15 rz R R^T/r7)) */
              {
                const double c = 2.0 * a2 / 3.0;
                T[8] += -c / R3;
                outer3(R, zhat, c * 3.0 * rz / R5, T);
                for (int k = 0; k < 3; ++k) T[4 * k] += -c * y3 * 3.0 * rz / R5;
                outer3(zhat, R, -c * y3 * 3.0 / R5, T);
                outer3(R, zhat, -c * y3 * 3.0 / R5, T);
                outer3(R, R, c * y3 * 15.0 * rz / R7, T);
              }
              /* (2 b2/3)(-J/r3 + 3 Rz R^T/r5 - x3 (same bracket)) */
              {
                const double c = 2.0 * b2 / 3.0;
                T[8] += -c / R3;
                outer3(zhat, R, c * 3.0 * rz / R5, T);
                for (int k = 0; k < 3; ++k) T[4 * k] += -c * x3 * 3.0 * rz / R5;
                outer3(zhat, R, -c * x3 * 3.0 / R5, T);
                outer3(R, zhat, -c * x3 * 3.0 / R5, T);
                outer3(R, R, c * x3 * 15.0 * rz / R7, T);
              }
              /* (2 a2 b2/3)(-I/r5 + 5 rz^2 I/r7 - 2J/r5 + 10 rz z R^T/r7 + 10 rz R z^T/r7 + 5 R R^T/r7 - 35 rz^2 R R^T/r9) */
              {
                const double c = 2.0 * a2 * b2 / 3.0;
                for (int k = 0; k < 3; ++k) T[4 * k] += c * (-1.0 / R5 + 5.0 * rz * rz / R7);
                T[8] += -2.0 * c / R5;
                outer3(zhat, R, c * 10.0 * rz / R7, T);
                outer3(R, zhat, c * 10.0 * rz / R7, T);
                outer3(R, R, c * (5.0 / R7 - 35.0 * rz * rz / R9), T);
              }
              /* M += -T P : columns x,y subtracted, column z added */
              for (int l = 0; l < 3; ++l) {
                M[3 * l] -= T[3 * l];
                M[3 * l + 1] -= T[3 * l + 1];
                M[3 * l + 2] += T[3 * l + 2];
              }
            }
            const double *f = force + 3 * j;
            for (int l = 0; l < 3; ++l) u[l] += (M[3 * l] * f[0] + M[3 * l + 1] * f[1] + M[3 * l + 2] * f[2]) * pref;
          }
    out[3 * i] = u[0]; out[3 * i + 1] = u[1]; out[3 * i + 2] = u[2];
  }
  return 0;
}

/* Dense 3N x 3N matrix of one kind (row-major), same blocks as the matvec.
 * Used to check symmetry / SPD properties and the dense builders
 * (mobility/mobility.py:967-1013 rotne_prager_tensor, :1018-1116
 * single_wall_fluid_mobility) on small N.  Non-periodic only. */
int oracle_mobility_dense(int kind, int wall, long N, const double *r, double eta, double a,
                          double *Mout) {
  if (kind < 0 || kind > 3) return 2;
  const double inva = 1.0 / a;
  double norm;
  if (kind == KIND_TT) norm = 1.0 / (8.0 * ORACLE_PI * eta * a);
  else if (kind == KIND_RR) norm = 1.0 / (8.0 * ORACLE_PI * eta * a * a * a);
  else norm = 1.0 / (8.0 * ORACLE_PI * eta * a * a);
  const long ld = 3 * N;
  for (long i = 0; i < N; ++i)
    for (long j = 0; j < N; ++j) {
      block3 B;
      pair_block(kind, wall, 0, r[3 * i] - r[3 * j], r[3 * i + 1] - r[3 * j + 1],
                 r[3 * i + 2] - r[3 * j + 2], r[3 * i + 2], r[3 * j + 2], inva, i == j, &B);
      for (int p = 0; p < 3; ++p)
        for (int q = 0; q < 3; ++q) Mout[(3 * i + p) * ld + 3 * j + q] = B.m[3 * p + q] * norm;
    }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* Blob-blob soft repulsion, all pairs, minimal image only.                   */
/* multi_bodies/forces_numba.py:12-55: F_i = sum_{j!=i} f0(r) (r_j - r_i),    */
/* f0 = -(eps/b) exp(-(r-2a)/b)/r for r > 2a, else -(eps/b)/max(r,1e-25).     */
/* ------------------------------------------------------------------------ */
int oracle_blob_blob_force(long N, const double *r, const double *L, double eps, double b,
                           double a, double *out) {
  if (N < 0 || !out) return 1;
#pragma omp parallel for schedule(dynamic, 16)
  for (long i = 0; i < N; ++i) {
    double fx = 0, fy = 0, fz = 0;
    for (long j = 0; j < N; ++j) {
      if (i == j) continue;
      double dr[3];
      for (int k = 0; k < 3; ++k) {
        dr[k] = r[3 * j + k] - r[3 * i + k];
        if (L[k] > 0) dr[k] = wrap_nearest(dr[k], L[k]);
      }
      double rn = sqrt(dr[0] * dr[0] + dr[1] * dr[1] + dr[2] * dr[2]);
      double f0;
      if (rn > 2 * a) f0 = -((eps / b) * exp(-(rn - 2.0 * a) / b) / rn);
      else f0 = -((eps / b) / fmax(rn, 1e-25));
      fx += f0 * dr[0]; fy += f0 * dr[1]; fz += f0 * dr[2];
    }
    out[3 * i] = fx; out[3 * i + 1] = fy; out[3 * i + 2] = fz;
  }
  return 0;
}

/* Subset variant for full-size spot checks (as oracle_mobility_matvec_targets): the force on the listed blobs
 * only, every other blob as a source.  out has 3*n_targets entries.  Same pair formula, forces_numba.py:12-55. */
int oracle_blob_blob_force_targets(long N, const double *r, const double *L, double eps, double b, double a,
                                   long n_targets, const long *targets, double *out) {
  if (N < 0 || n_targets < 0 || !out) return 1;
  for (long t = 0; t < n_targets; ++t)
    if (targets[t] < 0 || targets[t] >= N) return 2;
#pragma omp parallel for schedule(dynamic, 1)
  for (long t = 0; t < n_targets; ++t) {
    const long i = targets[t];
    double fx = 0, fy = 0, fz = 0;
    for (long j = 0; j < N; ++j) {
      if (i == j) continue;
      double dr[3];
      for (int k = 0; k < 3; ++k) {
        dr[k] = r[3 * j + k] - r[3 * i + k];
        if (L[k] > 0) dr[k] = wrap_nearest(dr[k], L[k]);
      }
      double rn = sqrt(dr[0] * dr[0] + dr[1] * dr[1] + dr[2] * dr[2]);
      double f0;
      if (rn > 2 * a) f0 = -((eps / b) * exp(-(rn - 2.0 * a) / b) / rn);
      else f0 = -((eps / b) / fmax(rn, 1e-25));
      fx += f0 * dr[0]; fy += f0 * dr[1]; fz += f0 * dr[2];
    }
    out[3 * t] = fx; out[3 * t + 1] = fy; out[3 * t + 2] = fz;
  }
  return 0;
}

/* One radius per blob: contact distance a_i + a_j (multi_bodies/forces_numba.py:73-122). */
int oracle_blob_blob_force_radii(long N, const double *r, const double *radii, const double *L, double eps, double b,
                                 double *out) {
  if (N < 0 || !out || !radii) return 1;
#pragma omp parallel for schedule(dynamic, 16)
  for (long i = 0; i < N; ++i) {
    double fx = 0, fy = 0, fz = 0;
    for (long j = 0; j < N; ++j) {
      if (i == j) continue;
      double dr[3];
      for (int k = 0; k < 3; ++k) {
        dr[k] = r[3 * j + k] - r[3 * i + k];
        if (L[k] > 0) dr[k] = wrap_nearest(dr[k], L[k]);
      }
      const double a = (radii[i] + radii[j]) * 0.5;
      double rn = sqrt(dr[0] * dr[0] + dr[1] * dr[1] + dr[2] * dr[2]);
      double f0;
      if (rn > 2 * a) f0 = -((eps / b) * exp(-(rn - 2.0 * a) / b) / rn);
      else f0 = -((eps / b) / fmax(rn, 1e-25));
      fx += f0 * dr[0]; fy += f0 * dr[1]; fz += f0 * dr[2];
    }
    out[3 * i] = fx; out[3 * i + 1] = fy; out[3 * i + 2] = fz;
  }
  return 0;
}

/* Height clamp + damping diagonal of the Python wrappers
 * (mobility/mobility.py:52-64 shift_heights uses `<=`, :67-84
 * damping_matrix_B uses `<`).  r_eff gets the clamped copy, bdiag (N entries)
 * the per-blob factor; returns 1 in *overlap if any blob is below z = a. */
int oracle_wall_regularisation(long N, const double *r, double a, double *r_eff, double *bdiag,
                               int *overlap) {
  int ov = 0;
  for (long i = 0; i < N; ++i) {
    double z = r[3 * i + 2];
    r_eff[3 * i] = r[3 * i];
    r_eff[3 * i + 1] = r[3 * i + 1];
    r_eff[3 * i + 2] = (z <= a) ? a : z;
    if (z < a) { bdiag[i] = z / a; ov = 1; } else bdiag[i] = 1.0;
  }
  if (overlap) *overlap = ov;
  return 0;
}

/* ---- Stokeslet pressure and Stokes double layer, source -> target ----------------------------------------------
 * The remaining O(N_s N_t) operators of mobility/mobility_numba.py.  Written as the reference writes them (explicit
 * divisions, sqrt, the nine-term contraction as (r.n)(r.v) r), unbounded or above a no-slip wall at z = 0.
 *
 * oracle_pressure_stokeslet: p_t = 1/(4 pi) sum_s f_s . r / |r|^3  (+ Blake image system, :1399-1476).  Two notes on
 * the reference's text: (i) its single-wall routine rescales the running sum by 1/(4 pi) INSIDE the source loop
 * (:1474), so for more than one source its result depends on the source order and decays geometrically; this
 * restatement applies the factor once, which is what the routine returns for a single source and what superposition
 * of single-source calls gives (that is how tests/golden pins it); (ii) the pseudo-periodic branch of both routines
 * divides by the UNWRAPPED distance (:1374-1375 before :1381-1389), so only L = 0 is restated. */
int oracle_pressure_stokeslet(long Ns, const double *src, long Nt, const double *tgt, const double *force, int wall,
                              double *p) {
  const double c = 1.0 / (4.0 * M_PI);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < Nt; ++i) {
    const double xi = tgt[3 * i], yi = tgt[3 * i + 1], zi = tgt[3 * i + 2];
    double acc = 0.0;
    for (long j = 0; j < Ns; ++j) {
      const double fx = force[3 * j], fy = force[3 * j + 1], fz = force[3 * j + 2];
      const double rx = xi - src[3 * j], ry = yi - src[3 * j + 1];
      double rz = zi - src[3 * j + 2];
      double r = sqrt(rx * rx + ry * ry + rz * rz);
      double r3 = r * r * r;
      acc += (fx * rx + fy * ry + fz * rz) / r3;                       /* :1391 / :1459 */
      if (wall) {                                                      /* :1461-1473 */
        const double h = src[3 * j + 2];
        rz = zi + h;
        r = sqrt(rx * rx + ry * ry + rz * rz);
        r3 = r * r * r;
        const double r5 = r3 * r * r;
        acc += -(fx * rx + fy * ry + fz * rz) / r3;
        acc += -fx * 2 * h * (-3 * rz * rx / r5);
        acc += -fy * 2 * h * (-3 * rz * ry / r5);
        acc += fz * 2 * h * (-3 * rz * rz / r5 + 1.0 / r3);
      }
    }
    p[i] = c * acc;
  }
  return 0;
}

/* oracle_double_layer: u_t = -3/(4 pi) sum_s w_s [ r (r.n_s)(r.v_s) / |r|^5 (r > 1e-14) + wall images ]
 *   (mobility_numba.py:1662-1766; images after Gimbutas et al. 2015: reflected double layer, derivative dipole /
 *   quadrupole, dipole, quadrupole -- the image terms are evaluated for r = 0 too).
 * blob_radius >= 0 selects the RPY-regularised unbounded operator (:2095-2168):
 *   (1 - 10 a^2/(3 r^2)) r (r.n)(r.v)/|r|^5 + (2 a^2/3) [(n.v) r + (r.v) n + (r.n) v]/|r|^5,  pairs with r < 1e-14 skipped. */
int oracle_double_layer(long Ns, const double *src, long Nt, const double *tgt, const double *normals,
                        const double *vector, const double *weights, int wall, double blob_radius, double *u) {
  const double factor = -3.0 / (4.0 * M_PI);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < Nt; ++i) {
    const double xi = tgt[3 * i], yi = tgt[3 * i + 1], zi = tgt[3 * i + 2];
    double ux = 0.0, uy = 0.0, uz = 0.0;
    for (long j = 0; j < Ns; ++j) {
      const double nx = normals[3 * j], ny = normals[3 * j + 1], nz = normals[3 * j + 2];
      const double vx = vector[3 * j], vy = vector[3 * j + 1], vz = vector[3 * j + 2];
      const double w = weights[j];
      const double rx = xi - src[3 * j], ry = yi - src[3 * j + 1];
      double rz = zi - src[3 * j + 2];
      double r2 = rx * rx + ry * ry + rz * rz;
      double r = sqrt(r2);
      double r5 = r2 * r2 * r;
      if (blob_radius >= 0.0) {
        if (r < 1e-14) continue;                                        /* :2139-2140 */
        const double a2 = blob_radius * blob_radius;
        const double rn = rx * nx + ry * ny + rz * nz, rv = rx * vx + ry * vy + rz * vz, nv = nx * vx + ny * vy + nz * vz;
        const double c0 = (1 - 10 * a2 / (3 * r2)) * rn * rv * w / r5;
        const double c1 = (2.0 * a2 / 3.0) * w / r5;
        ux += c0 * rx + c1 * (nv * rx + rv * nx + rn * vx);
        uy += c0 * ry + c1 * (nv * ry + rv * ny + rn * vy);
        uz += c0 * rz + c1 * (nv * rz + rv * nz + rn * vz);
        continue;
      }
      if (r > 1e-14) {                                                  /* :1715-1722 */
        const double c0 = (rx * nx + ry * ny + rz * nz) * (rx * vx + ry * vy + rz * vz) * w / r5;
        ux += rx * c0; uy += ry * c0; uz += rz * c0;
      }
      if (wall) {                                                       /* :1725-1759 */
        const double h = src[3 * j + 2];
        rz = zi + h;
        r2 = rx * rx + ry * ry + rz * rz;
        r = sqrt(r2);
        const double r3 = r2 * r;
        r5 = r3 * r2;
        const double rn = rx * nx + ry * ny - rz * nz, rv = rx * vx + ry * vy - rz * vz, nv = nx * vx + ny * vy + nz * vz;
        const double c0 = rn * rv * w / r5;
        ux -= rx * c0; uy -= ry * c0; uz -= rz * c0;
        ux += -2 * zi * nv * (-rx * rz / r2) * w / r3;
        uy += -2 * zi * nv * (-ry * rz / r2) * w / r3;
        uz += -2 * zi * nv * (1.0 / 3.0 - rz * rz / r2) * w / r3;
        ux += -2 * zi * h * (rx * nv + vx * rn + nx * rv - 5 * rx * rv * rn / r2) * w / r5;
        uy += -2 * zi * h * (ry * nv + vy * rn + ny * rv - 5 * ry * rv * rn / r2) * w / r5;
        uz += -2 * zi * h * (rz * nv - vz * rn - nz * rv - 5 * rz * rv * rn / r2) * w / r5;
        uz += 2 * nv * rz * w / (3 * r3);
        uz += 2 * h * (-nv / 3 + rv * rn / r2) * w / r3;
      }
    }
    u[3 * i] = factor * ux; u[3 * i + 1] = factor * uy; u[3 * i + 2] = factor * uz;
  }
  return 0;
}

/* Cap the OpenMP team (the GPU box exposes every hardware thread of the host but grants a CPU share of a few cores:
   an oversubscribed team makes the baseline slower than it is). */
void oracle_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
