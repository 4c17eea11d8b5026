// st_kernels.h -- source -> target translation mobility with per-blob radii (K13), gfx950 fp64.
//
// WHAT: u_t = sum_s M(x_t, a_t; y_s, a_s) f_s  for N_t targets != N_s sources, every blob with its own
// radius; unbounded part after Zuk et al. (three regimes), wall part = Blake/Swan-Brady image system for
// unequal radii (mobility/mobility_numba.py:1480-1658, CUDA twin mobility_pycuda.py:1841-2067).
// Used by the reference for `radii_*` mobility modes (mobility.py:1369-1374) and velocity-field output.
//
// HOW: the reference writes out 9 entries x 5 tensor terms per pair (686 flops as written).  All five
// terms are combinations of I, R R^T, R z^T, z R^T, z z^T, so with g = (-f_x, -f_y, f_z) (= -P f)
//     wall part = alpha g + (beta R.g + gamma g_z) R + (delta R.g + eps g_z) z
// with five scalars that are short polynomials in 1/|R|:
//   alpha = i1 + (s/3) i3 + 2 x3 y3 i3 - 2 rz (a2 y3 + b2 x3) i5 + (2 a2 b2/3)(-i5 + 5 rz^2 i7)
//   beta  = i3 - s i5 - 6 x3 y3 i5 + 10 rz (a2 y3 + b2 x3) i7 + (2 a2 b2/3)(5 i7 - 35 rz^2 i9)
//   gamma = -2 x3 i3 + 2 (a2 - b2) x3 i5 + (20/3) a2 b2 rz i7
//   delta = -2 y3 i3 + 2 (b2 - a2) y3 i5 + (20/3) a2 b2 rz i7
//   eps   = -2 i1 - (2/3) s i3 - (4/3) a2 b2 i5
// (a2 = a_t^2, b2 = a_s^2, s = a2 + b2, x3 / y3 = target / source height, rz = x3 + y3, i_n = |R|^-n).
// Same sweep skeleton as matvec_kernels.h: lane = target, 4 waves split an LDS tile of sources, source
// chunks over blockIdx.y, fixed-order reduction, atomic-free.  No self special case: r <= |a_t - a_s|
// (which includes r = 0) is the third regime of the unbounded part.
#pragma once
#include "matvec_kernels.h"

namespace rmb {

struct StArgs {
  const double4* src;   // [ns] (x, y, z_eff, b)
  const double* rad_s;  // [ns]
  const double* force;  // [3 ns]
  const double4* tgt;   // [nt] (x, y, z_eff, b)
  const double* rad_t;  // [nt]
  double* out;          // [3 nt]
  double* partial;      // [n_chunks][3][n_tgt_pad]
  long ns, nt, n_tgt_pad, chunk_len;
  int n_chunks;
  double prefactor;
  double Lx, Ly, Lz, iLx, iLy, iLz;
};

// positions + per-blob radius -> (x, y, z_eff, b)   (mobility/mobility.py:87-119)
__global__ void pack_positions_radii_kernel(const double* r, const double* rad, long n, int wall, double4* pos) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double z = r[3 * i + 2], a = rad[i];
  double4 p;
  p.x = r[3 * i]; p.y = r[3 * i + 1];
  if (wall) { p.z = (z > a) ? z : a; p.w = (z < a) ? z / a : 1.0; }
  else      { p.z = z; p.w = 1.0; }
  pos[i] = p;
}

template <bool WALL>
__device__ __forceinline__ void pair_st(double dx, double dy, double dz, double x3, double y3, double at, double as,
                                        double fx, double fy, double fz, Vec3& u) {
  const double rho2 = __builtin_fma(dy, dy, dx * dx);
  const double r2 = __builtin_fma(dz, dz, rho2);
  const double a2 = at * at, b2 = as * as, s = a2 + b2;
  const double ir = rsqrt_f64(r2);
  const double ir2 = ir * ir;
  double C1 = __builtin_fma(s * (1.0 / 3.0), ir2, 1.0) * ir;
  double C2 = __builtin_fma(-s, ir2, 1.0) * ir2 * ir;
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
    C1 = far ? C1 : (mid ? C1m : (4.0 / 3.0) / fmax(at, as));
    C2 = far ? C2 : (mid ? C2m : 0.0);
  }
  const double pxy = __builtin_fma(dy, fy, dx * fx);
  double cD = C2 * __builtin_fma(dz, fz, pxy);
  if constexpr (!WALL) {
    u.x = __builtin_fma(C1, fx, u.x); u.x = __builtin_fma(cD, dx, u.x);
    u.y = __builtin_fma(C1, fy, u.y); u.y = __builtin_fma(cD, dy, u.y);
    u.z = __builtin_fma(C1, fz, u.z); u.z = __builtin_fma(cD, dz, u.z);
  } else {
    const double rz = x3 + y3;
    const double R2 = __builtin_fma(rz, rz, rho2);
    const double i1 = rsqrt_f64(R2);
    const double i2 = i1 * i1, i3 = i1 * i2, i5 = i3 * i2, i7 = i5 * i2, i9 = i7 * i2;
    const double ab = a2 * b2, xy = x3 * y3;
    const double m = rz * __builtin_fma(a2, y3, b2 * x3);       // rz (a2 y3 + b2 x3)
    const double ab23 = ab * (2.0 / 3.0);
    const double rz2 = rz * rz;
    const double alpha = __builtin_fma(ab23, __builtin_fma(5.0 * rz2, i7, -i5),
                                       __builtin_fma(-2.0 * m, i5, __builtin_fma(__builtin_fma(2.0, xy, s * (1.0 / 3.0)), i3, i1)));
    const double beta = __builtin_fma(ab23, __builtin_fma(-35.0 * rz2, i9, 5.0 * i7),
                                      __builtin_fma(10.0 * m, i7, __builtin_fma(-__builtin_fma(6.0, xy, s), i5, i3)));
    const double abz = ab * (20.0 / 3.0) * rz * i7;
    const double dab = 2.0 * (a2 - b2) * i5;
    const double gamma = __builtin_fma(x3, __builtin_fma(-2.0, i3, dab), abz);
    const double delta = __builtin_fma(y3, -__builtin_fma(2.0, i3, dab), abz);
    const double eps = -__builtin_fma(ab * (4.0 / 3.0), i5, __builtin_fma(s * (2.0 / 3.0), i3, i1 + i1));
    // g = (-fx, -fy, fz);  R.g = -(dx fx + dy fy) + rz fz
    const double Rg = __builtin_fma(rz, fz, -pxy);
    const double cR = __builtin_fma(beta, Rg, gamma * fz);
    const double cz = __builtin_fma(delta, Rg, eps * fz);
    const double cFxy = C1 - alpha;           // C1 f + alpha g, x/y components
    const double cFz = C1 + alpha;
    const double cDR = cD + cR;
    u.x = __builtin_fma(cFxy, fx, u.x); u.x = __builtin_fma(cDR, dx, u.x);
    u.y = __builtin_fma(cFxy, fy, u.y); u.y = __builtin_fma(cDR, dy, u.y);
    u.z = __builtin_fma(cFz, fz, u.z); u.z = __builtin_fma(cD, dz, u.z);
    u.z = __builtin_fma(cR, rz, u.z); u.z += cz;
  }
}

// MODE 0: unbounded, 1: no-slip wall at z = 0 (Swan-Brady image system), 2: stress-free surface at z = 0
// (mobility_numba.py:1941-2091: the unbounded tensor of the mirror image R = (dx, dy, z_t + z_s), raw heights,
// added with its z column negated  =>  second pair_st<false> call with (fx, fy, -fz)).
template <int MODE>
__device__ __forceinline__ void pair_st_mode(double dx, double dy, double dz, double zt, double zs, double at, double as,
                                             double fx, double fy, double fz, Vec3& u) {
  pair_st<MODE == 1>(dx, dy, dz, zt, zs, at, as, fx, fy, fz, u);
  if constexpr (MODE == 2) pair_st<false>(dx, dy, zt + zs, zt, zs, at, as, fx, fy, -fz, u);
}

template <int MODE, bool PERIODIC>
__global__ __launch_bounds__(kBlock) void st_sweep_kernel(const StArgs a) {
  __shared__ double2 tile[kTile * 4];
  __shared__ double red[(kWaves - 1) * 3 * 64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const long ti = 64L * blockIdx.x + lane;
  const bool valid = ti < a.nt;
  const double4 tp = a.tgt[valid ? ti : a.nt - 1];
  const double at = a.rad_t[valid ? ti : a.nt - 1];
  const long c0 = (long)blockIdx.y * a.chunk_len;
  long c1 = c0 + a.chunk_len;
  if (c1 > a.ns) c1 = a.ns;
  Vec3 acc = {0.0, 0.0, 0.0};
  for (long j0 = c0; j0 < c1; j0 += kTile) {
    const int n = (int)((c1 - j0 < kTile) ? (c1 - j0) : kTile);
    __syncthreads();
    for (int t = threadIdx.x; t < n; t += kBlock) {
      const long j = j0 + t;
      const double4 p = a.src[j];
      tile[t * 4 + 0] = make_double2(p.x, p.y);
      tile[t * 4 + 1] = make_double2(p.z, a.rad_s[j]);
      tile[t * 4 + 2] = make_double2(a.force[3 * j] * p.w, a.force[3 * j + 1] * p.w);
      tile[t * 4 + 3] = make_double2(a.force[3 * j + 2] * p.w, 0.0);
    }
    __syncthreads();
    for (int s = wave; s < n; s += kWaves) {
      const double2 q0 = tile[s * 4 + 0], q1 = tile[s * 4 + 1], q2 = tile[s * 4 + 2], q3 = tile[s * 4 + 3];
      double dx = tp.x - q0.x, dy = tp.y - q0.y, dz = tp.z - q1.x;
      if constexpr (!PERIODIC) {
        pair_st_mode<MODE>(dx, dy, dz, tp.z, q1.x, at, q1.y, q2.x, q2.y, q3.x, acc);
      } else {
        const int px = a.Lx > 0, py = a.Ly > 0, pz = a.Lz > 0;
        if (px) dx = wrap_nearest(dx, a.Lx, a.iLx);
        if (py) dy = wrap_nearest(dy, a.Ly, a.iLy);
        if (pz) dz = wrap_nearest(dz, a.Lz, a.iLz);
        for (int bx = -px; bx <= px; ++bx)
          for (int by = -py; by <= py; ++by)
            for (int bz = -pz; bz <= pz; ++bz)
              pair_st_mode<MODE>(dx + bx * a.Lx, dy + by * a.Ly, dz + bz * a.Lz, tp.z, q1.x, at, q1.y, q2.x, q2.y, q3.x, acc);
      }
    }
  }
  if (wave > 0) {
    double* r = red + (wave - 1) * 3 * 64;
    r[lane] = acc.x; r[64 + lane] = acc.y; r[128 + lane] = acc.z;
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < kWaves - 1; ++w) {
    const double* r = red + w * 3 * 64;
    acc.x += r[lane]; acc.y += r[64 + lane]; acc.z += r[128 + lane];
  }
  if (a.n_chunks == 1) {
    if (!valid) return;
    const double sc = a.prefactor * tp.w;
    a.out[3 * ti] = acc.x * sc; a.out[3 * ti + 1] = acc.y * sc; a.out[3 * ti + 2] = acc.z * sc;
  } else {
    double* p = a.partial + (long)blockIdx.y * 3 * a.n_tgt_pad;
    p[ti] = acc.x; p[a.n_tgt_pad + ti] = acc.y; p[2 * a.n_tgt_pad + ti] = acc.z;
  }
}

__global__ __launch_bounds__(256) void st_finalize_kernel(const StArgs a) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.nt) return;
  double x = 0, y = 0, z = 0;
  for (int c = 0; c < a.n_chunks; ++c) {
    const double* p = a.partial + (long)c * 3 * a.n_tgt_pad;
    x += p[t]; y += p[a.n_tgt_pad + t]; z += p[2 * a.n_tgt_pad + t];
  }
  const double sc = a.prefactor * a.tgt[t].w;
  a.out[3 * t] = x * sc; a.out[3 * t + 1] = y * sc; a.out[3 * t + 2] = z * sc;
}

}  // namespace rmb
