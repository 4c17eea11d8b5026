// aux_kernels.h -- Stokeslet pressure and Stokes double layer, source -> target (gfx950, fp64).
//
// WHAT: the remaining O(N_s N_t) operators of mobility/mobility_numba.py:
//   pressure      p_t = 1/(4 pi) sum_s f_s.r / |r|^3  (+ Blake image system above a wall)   :1332-1476
//   double layer  u_t = -3/(4 pi) sum_s w_s r (r.n_s)(r.v_s) / |r|^5  (+ wall images)       :1662-1766
//   RPY double layer (unbounded, blob radius a)                                             :2095-2168
// Notes on the reference's text (restated in oracle/oracle_mobility.c, pinned by tests/golden/g11_*): the wall pressure
// routine rescales its running sum inside the source loop (:1474) -- the factor is applied once here, which is what
// the routine returns for one source and what superposition gives; only periodic_length = 0 (the reference's periodic
// branch divides by the unwrapped distance).
//
// HOW: same one-sided skeleton as st_kernels.h -- lane = target, an LDS tile of source records shared by the four waves
// of a workgroup (each wave takes every fourth source), source chunks over blockIdx.y with a fixed-order reduction, no
// atomics: bit-reproducible.  The nine-term contraction of the reference is (r.n)(r.v) r; divisions become one
// inverse square root per distance.  HBM traffic is the records once per target tile; the kernel is fp64-VALU bound
// like every pair sweep here (VALU instructions per source-target pair in this build: pressure 21 / 42 with the wall,
// double layer 35 / 82 with the wall images / 50 for the RPY form).
#pragma once
#include "matvec_kernels.h"

namespace rmb {

enum { AUX_P_FREE = 0, AUX_P_WALL = 1, AUX_DL_FREE = 2, AUX_DL_WALL = 3, AUX_DL_RPY = 4 };

struct AuxArgs {
  const double* src;    // [3 ns]
  const double* tgt;    // [3 nt]
  const double* v0;     // pressure: force [3 ns];  double layer: normals [3 ns]
  const double* v1;     // double layer: vector [3 ns]
  const double* w;      // double layer: weights [ns]
  double* out;          // pressure [nt];  double layer [3 nt]
  double* partial;      // [n_chunks][NOUT][n_tgt_pad]
  long ns, nt, n_tgt_pad, chunk_len;
  int n_chunks;
  double prefactor;     // 1/(4 pi)  |  -3/(4 pi)
  double a2;            // RPY double layer: blob radius squared
};

template <int MODE> struct AuxShape {
  static constexpr int NOUT = MODE <= AUX_P_WALL ? 1 : 3;
  static constexpr int REC2 = MODE <= AUX_P_WALL ? 3 : 5;     // double2 per source record
};

// One source at separation (dx, dy, dz) from the target (heights zt, zs), record q[] = what the tile holds.
template <int MODE>
__device__ __forceinline__ void aux_pair(const AuxArgs& a, double dx, double dy, double dz, double zt, double zs,
                                         const double* q, double* acc) {
  const double rho2 = __builtin_fma(dy, dy, dx * dx);
  const double r2 = __builtin_fma(dz, dz, rho2);
  if constexpr (MODE == AUX_P_FREE || MODE == AUX_P_WALL) {
    const double fx = q[0], fy = q[1], fz = q[2];
    const double ir = rsqrt_f64(r2);
    const double pxy = __builtin_fma(dy, fy, dx * fx);
    acc[0] = __builtin_fma(__builtin_fma(dz, fz, pxy), ir * ir * ir, acc[0]);
    if constexpr (MODE == AUX_P_WALL) {
      // -(f.R)/R^3 + 6 h R_z (f_x d_x + f_y d_y)/R^5 + 2 h f_z (1/R^3 - 3 R_z^2/R^5),  R = (d_x, d_y, z_t + z_s), h = z_s
      const double Rz = zt + zs;
      const double iR = rsqrt_f64(__builtin_fma(Rz, Rz, rho2));
      const double iR2 = iR * iR, iR3 = iR * iR2, iR5 = iR3 * iR2;
      const double h2 = zs + zs;
      const double t5 = (3.0 * h2) * Rz * iR5;                            // 6 h R_z / R^5
      const double s = __builtin_fma(t5, __builtin_fma(-Rz, fz, pxy), iR3 * __builtin_fma(h2 - Rz, fz, -pxy));
      acc[0] += s;
    }
  } else {
    const double nx = q[0], ny = q[1], nz = q[2], vx = q[3], vy = q[4], vz = q[5], w = q[6];
    const double nxy = __builtin_fma(dy, ny, dx * nx), vxy = __builtin_fma(dy, vy, dx * vx);
    const double ok = (r2 > 1e-28) ? 1.0 : 0.0;                           // r > 1e-14; the diagonal is skipped
    const double ir = rsqrt_f64((r2 > 1e-28) ? r2 : 1.0);
    const double ir2 = ir * ir;
    const double w5 = (w * ok) * (ir2 * ir2 * ir);                        // w / r^5
    const double rn = __builtin_fma(dz, nz, nxy), rv = __builtin_fma(dz, vz, vxy);
    if constexpr (MODE == AUX_DL_RPY) {
      const double nv = __builtin_fma(nz, vz, __builtin_fma(ny, vy, nx * vx));
      const double c0 = __builtin_fma(-(10.0 / 3.0) * a.a2, ir2, 1.0) * (rn * rv) * w5;
      const double c1 = (2.0 / 3.0) * a.a2 * w5;
      const double cr = __builtin_fma(c1, nv, c0), cn = c1 * rv, cv = c1 * rn;
      acc[0] = __builtin_fma(cr, dx, __builtin_fma(cn, nx, __builtin_fma(cv, vx, acc[0])));
      acc[1] = __builtin_fma(cr, dy, __builtin_fma(cn, ny, __builtin_fma(cv, vy, acc[1])));
      acc[2] = __builtin_fma(cr, dz, __builtin_fma(cn, nz, __builtin_fma(cv, vz, acc[2])));
    } else {
      const double c0 = (rn * rv) * w5;
      acc[0] = __builtin_fma(c0, dx, acc[0]); acc[1] = __builtin_fma(c0, dy, acc[1]); acc[2] = __builtin_fma(c0, dz, acc[2]);
      if constexpr (MODE == AUX_DL_WALL) {
        // image terms (kept for r = 0: a node sees its own image).  With R = (d_x, d_y, R_z), rn = R.(n_x, n_y, -n_z),
        // rv likewise, nv = n.v, h = z_s, and everything over a common w/R^5:
        //   xy:  d_xy { -rn rv + 2 z_t nv R_z - 2 z_t h (nv - 5 rn rv/R^2) } - 2 z_t h (rn v_xy + rv n_xy)
        //   z :  R_z { -rn rv + 2 z_t nv R_z - 2 z_t h (nv - 5 rn rv/R^2) } + 2 z_t h (rn v_z + rv n_z)
        //        - (2/3) z_t nv R^2 + (2/3) nv R_z R^2 + 2 h (rv rn - nv R^2/3)
        const double Rz = zt + zs;
        const double R2 = __builtin_fma(Rz, Rz, rho2);
        const double iR = rsqrt_f64(R2);
        const double iR2 = iR * iR;
        const double W5 = w * (iR2 * iR2 * iR);
        const double rnI = __builtin_fma(-Rz, nz, nxy), rvI = __builtin_fma(-Rz, vz, vxy);
        const double nv = __builtin_fma(nz, vz, __builtin_fma(ny, vy, nx * vx));
        const double rr = rnI * rvI;
        const double zh2 = (zt + zt) * zs;                                // 2 z_t h
        const double core = __builtin_fma(zh2, __builtin_fma(5.0 * rr, iR2, -nv), __builtin_fma((zt + zt) * nv, Rz, -rr));
        const double cn = -zh2 * rvI, cv = -zh2 * rnI;
        const double nvR2 = nv * R2 * (1.0 / 3.0);
        const double zex = __builtin_fma(zs + zs, rr - nvR2, (2.0 * nvR2) * (Rz - zt));
        acc[0] = __builtin_fma(W5, __builtin_fma(core, dx, __builtin_fma(cn, nx, cv * vx)), acc[0]);
        acc[1] = __builtin_fma(W5, __builtin_fma(core, dy, __builtin_fma(cn, ny, cv * vy)), acc[1]);
        acc[2] = __builtin_fma(W5, __builtin_fma(core, Rz, zex - __builtin_fma(cn, nz, cv * vz)), acc[2]);
      }
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(kBlock) void aux_sweep_kernel(const AuxArgs a) {
  constexpr int NOUT = AuxShape<MODE>::NOUT, R2 = AuxShape<MODE>::REC2;
  __shared__ double2 tile[kTile * R2];
  __shared__ double red[(kWaves - 1) * NOUT * 64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const long ti = 64L * blockIdx.x + lane;
  const bool valid = ti < a.nt;
  const long tc = valid ? ti : a.nt - 1;
  const double xt = a.tgt[3 * tc], yt = a.tgt[3 * tc + 1], zt = a.tgt[3 * tc + 2];
  const long c0 = (long)blockIdx.y * a.chunk_len;
  long c1 = c0 + a.chunk_len;
  if (c1 > a.ns) c1 = a.ns;
  double acc[NOUT];
#pragma unroll
  for (int c = 0; c < NOUT; ++c) acc[c] = 0.0;
  for (long j0 = c0; j0 < c1; j0 += kTile) {
    const int n = (int)((c1 - j0 < kTile) ? (c1 - j0) : kTile);
    __syncthreads();
    for (int t = threadIdx.x; t < n; t += kBlock) {
      const long j = j0 + t;
      double2* rec = tile + t * R2;
      rec[0] = make_double2(a.src[3 * j], a.src[3 * j + 1]);
      rec[1] = make_double2(a.src[3 * j + 2], a.v0[3 * j]);
      rec[2] = make_double2(a.v0[3 * j + 1], a.v0[3 * j + 2]);
      if constexpr (R2 == 5) {
        rec[3] = make_double2(a.v1[3 * j], a.v1[3 * j + 1]);
        rec[4] = make_double2(a.v1[3 * j + 2], a.w[j]);
      }
    }
    __syncthreads();
    for (int s = wave; s < n; s += kWaves) {
      const double2* rec = tile + s * R2;
      double q[7];
      const double2 p0 = rec[0], p1 = rec[1], p2 = rec[2];
      q[0] = p1.y; q[1] = p2.x; q[2] = p2.y;
      if constexpr (R2 == 5) { const double2 p3 = rec[3], p4 = rec[4]; q[3] = p3.x; q[4] = p3.y; q[5] = p4.x; q[6] = p4.y; }
      aux_pair<MODE>(a, xt - p0.x, yt - p0.y, zt - p1.x, zt, p1.x, q, acc);
    }
  }
  if (wave > 0) {
    double* r = red + (wave - 1) * NOUT * 64;
#pragma unroll
    for (int c = 0; c < NOUT; ++c) r[c * 64 + lane] = acc[c];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < kWaves - 1; ++w) {
    const double* r = red + w * NOUT * 64;
#pragma unroll
    for (int c = 0; c < NOUT; ++c) acc[c] += r[c * 64 + lane];
  }
  if (a.n_chunks == 1) {
    if (!valid) return;
#pragma unroll
    for (int c = 0; c < NOUT; ++c) a.out[NOUT * ti + c] = acc[c] * a.prefactor;
  } else {
    double* p = a.partial + (long)blockIdx.y * NOUT * a.n_tgt_pad;
#pragma unroll
    for (int c = 0; c < NOUT; ++c) p[c * a.n_tgt_pad + ti] = acc[c];
  }
}

template <int NOUT>
__global__ __launch_bounds__(256) void aux_finalize_kernel(const AuxArgs a) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.nt) return;
  double s[NOUT];
#pragma unroll
  for (int c = 0; c < NOUT; ++c) s[c] = 0.0;
  for (int k = 0; k < a.n_chunks; ++k) {
    const double* p = a.partial + (long)k * NOUT * a.n_tgt_pad;
#pragma unroll
    for (int c = 0; c < NOUT; ++c) s[c] += p[c * a.n_tgt_pad + t];
  }
#pragma unroll
  for (int c = 0; c < NOUT; ++c) a.out[NOUT * t + c] = s[c] * a.prefactor;
}

}  // namespace rmb
