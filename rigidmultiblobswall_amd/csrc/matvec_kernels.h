// matvec_kernels.h -- the O(N^2) pairwise sweep (gfx950).
//
// Decomposition (MI355X-first, not the reference's "one CUDA thread per target reading every
// source from global memory", mobility/mobility_pycuda.py:150-254):
//   * a workgroup = 256 threads = 4 wave64 = one wave per SIMD of a CU;
//   * lane l of EVERY wave owns target  i = tgt_begin + 64*blockIdx.x + l  (registers: x,y,z);
//   * blockIdx.y selects a contiguous chunk of sources; the chunk is streamed through a 512-record
//     LDS tile (x,y,z, b*v) staged with coalesced loads, B-damping fused into the staging;
//   * the 4 waves split each tile's sources 4-ways (wave w takes records w, w+4, ...), read them
//     as wave-uniform (broadcast) ds_read_b128, and accumulate in registers;
//   * the 4 partial sums are combined through LDS in a fixed order; with one chunk the result is
//     finalised in place (self term, 1/(8 pi eta), B_i), otherwise partials go to a workspace and
//     a small second kernel reduces the chunks in fixed order -> deterministic, atomic-free.
// So N=1e4 still yields ~2000 workgroups (157 target tiles x 13 chunks) instead of 157 waves.
// The pair loop has no self test: pairs with i == j contribute through `self_term` in the
// epilogue and are skipped only in the one tile that overlaps the workgroup's own targets.
#pragma once
#include "pair_blocks.h"

namespace rmb {

constexpr int kTile = 512;  // source records per LDS tile

struct SweepArgs {
  const double4* pos;   // [n_src]  (x, y, z_eff, b)   b = B-damping factor (1 if not overlapping / no wall)
  const double* vec;    // [3 n_src] source vector (force or torque), AoS as the reference's (N,3)
  const double* vec2;   // [3 n_src] torque for the fused tt+tr kind, else nullptr
  double* out;          // [3 (tgt_end - tgt_begin)] final output (AoS), used when n_chunks == 1
  double* partial;      // [n_chunks][3][n_tgt_pad] chunk partials, used when n_chunks > 1
  long n_src;
  long tgt_begin, tgt_end;  // owned target index range (multi-GPU shard); targets index into pos as well
  long n_tgt_pad;           // 64 * gridDim.x
  long chunk_len;           // sources per chunk (multiple of kWaves)
  int n_chunks;
  int in_plane;             // zero v_z on load and u_z on store (in_plane_* kernels of the reference)
  double prefactor;         // 1/(8 pi eta)
  double Lx, Ly, Lz;        // pseudo-periodic lengths (<= 0: open)
  double iLx, iLy, iLz;     // 1/L (0 when open)
  PairConsts k;
};

template <int KIND> struct Rec { static constexpr int n2 = (KIND == KIND_TT_TR) ? 5 : 3; };

__device__ __forceinline__ double wrap_nearest(double r, double L, double invL) {
  // r - trunc(r/L + 0.5 sgn(r)) L          (mobility/mobility_numba.py:184-192)
  const double q = r * invL;
  const double h = (r > 0.0) ? 0.5 : ((r < 0.0) ? -0.5 : 0.0);
  return __builtin_fma(-__builtin_trunc(q + h), L, r);
}

template <int KIND, bool WALL, bool PERIODIC, bool SKIP_SELF>
__device__ __forceinline__ void sweep_tile(const SweepArgs& a, const double2* tile, int n, int wave, long j0,
                                           long ti, double xi, double yi, double zi, Vec3& acc) {
  constexpr int R2 = Rec<KIND>::n2;
  for (int s = wave; s < n; s += kWaves) {
    const double2 q0 = tile[s * R2 + 0];
    const double2 q1 = tile[s * R2 + 1];
    const double2 q2 = tile[s * R2 + 2];
    double wx = 0, wy = 0, wz = 0;
    if constexpr (KIND == KIND_TT_TR) {
      const double2 q3 = tile[s * R2 + 3];
      const double2 q4 = tile[s * R2 + 4];
      wx = q3.x; wy = q3.y; wz = q4.x;
    }
    const double xj = q0.x, yj = q0.y, zj = q1.x, vx = q1.y, vy = q2.x, vz = q2.y;
    double dx = xi - xj, dy = yi - yj, dz = zi - zj;
    if constexpr (!PERIODIC) {
      if constexpr (SKIP_SELF) {
        if (j0 + s == ti) continue;
      }
      pair_apply<KIND, WALL>(a.k, dx, dy, dz, zi, zj, vx, vy, vz, wx, wy, wz, acc);
    } else {
      const int px = a.Lx > 0, py = a.Ly > 0, pz = a.Lz > 0;
      if (px) dx = wrap_nearest(dx, a.Lx, a.iLx);
      if (py) dy = wrap_nearest(dy, a.Ly, a.iLy);
      if (pz) dz = wrap_nearest(dz, a.Lz, a.iLz);
      for (int bx = -px; bx <= px; ++bx)
        for (int by = -py; by <= py; ++by)
          for (int bz = -pz; bz <= pz; ++bz) {
            if constexpr (SKIP_SELF) {
              if (j0 + s == ti && bx == 0 && by == 0 && bz == 0) continue;
            }
            pair_apply<KIND, WALL>(a.k, dx + bx * a.Lx, dy + by * a.Ly, dz + bz * a.Lz, zi, zj, vx, vy, vz,
                                   wx, wy, wz, acc);
          }
    }
  }
}

template <int KIND, bool WALL, bool PERIODIC>
__global__ __launch_bounds__(kBlock) void sweep_kernel(const SweepArgs a) {
  constexpr int R2 = Rec<KIND>::n2;
  __shared__ double2 tile[kTile * R2];
  __shared__ double red[(kWaves - 1) * 3 * 64];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: the source loop index stays scalar
  const long blk_t0 = a.tgt_begin + 64L * blockIdx.x;
  const long ti = blk_t0 + lane;
  const bool valid = ti < a.tgt_end;
  const double4 tp = a.pos[valid ? ti : a.tgt_end - 1];
  const double xi = tp.x, yi = tp.y, zi = tp.z;

  const long c0 = (long)blockIdx.y * a.chunk_len;
  long c1 = c0 + a.chunk_len;
  if (c1 > a.n_src) c1 = a.n_src;

  Vec3 acc = {0.0, 0.0, 0.0};
  for (long j0 = c0; j0 < c1; j0 += kTile) {
    const int n = (int)((c1 - j0 < kTile) ? (c1 - j0) : kTile);
    __syncthreads();
    for (int t = threadIdx.x; t < n; t += kBlock) {
      const long j = j0 + t;
      const double4 p = a.pos[j];
      const double b = p.w;
      double vx = a.vec[3 * j] * b, vy = a.vec[3 * j + 1] * b, vz = a.vec[3 * j + 2] * b;
      if (a.in_plane) vz = 0.0;
      tile[t * R2 + 0] = make_double2(p.x, p.y);
      tile[t * R2 + 1] = make_double2(p.z, vx);
      tile[t * R2 + 2] = make_double2(vy, vz);
      if constexpr (KIND == KIND_TT_TR) {
        double wx = a.vec2[3 * j] * b, wy = a.vec2[3 * j + 1] * b, wz = a.vec2[3 * j + 2] * b;
        if (a.in_plane) wz = 0.0;
        tile[t * R2 + 3] = make_double2(wx, wy);
        tile[t * R2 + 4] = make_double2(wz, 0.0);
      }
    }
    __syncthreads();
    const bool diag = (j0 < blk_t0 + 64) && (j0 + n > blk_t0);  // tile overlaps this block's own targets
    if (diag) sweep_tile<KIND, WALL, PERIODIC, true>(a, tile, n, wave, j0, ti, xi, yi, zi, acc);
    else      sweep_tile<KIND, WALL, PERIODIC, false>(a, tile, n, wave, j0, ti, xi, yi, zi, acc);
  }

  // combine the 4 waves (fixed order 0+1+2+3)
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
    const double b = tp.w;
    double vx = a.vec[3 * ti] * b, vy = a.vec[3 * ti + 1] * b, vz = a.vec[3 * ti + 2] * b;
    double wx = 0, wy = 0, wz = 0;
    if constexpr (KIND == KIND_TT_TR) { wx = a.vec2[3 * ti] * b; wy = a.vec2[3 * ti + 1] * b; wz = a.vec2[3 * ti + 2] * b; }
    if (a.in_plane) { vz = 0.0; wz = 0.0; }
    self_term<KIND, WALL>(a.k, zi, vx, vy, vz, wx, wy, wz, acc);
    const double sc = a.prefactor * b;
    const long o = 3 * (ti - a.tgt_begin);
    a.out[o] = acc.x * sc; a.out[o + 1] = acc.y * sc; a.out[o + 2] = a.in_plane ? 0.0 : acc.z * sc;
  } else {
    const long col = 64L * blockIdx.x + lane;
    double* p = a.partial + (long)blockIdx.y * 3 * a.n_tgt_pad;
    p[col] = acc.x; p[a.n_tgt_pad + col] = acc.y; p[2 * a.n_tgt_pad + col] = acc.z;
  }
}

// Second pass when the sources were split into chunks: fixed-order sum over chunks, self term,
// prefactor and B_i, AoS store.
template <int KIND, bool WALL>
__global__ __launch_bounds__(256) void finalize_kernel(const SweepArgs a) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long ti = a.tgt_begin + t;
  if (ti >= a.tgt_end) return;
  Vec3 acc = {0.0, 0.0, 0.0};
  for (int c = 0; c < a.n_chunks; ++c) {
    const double* p = a.partial + (long)c * 3 * a.n_tgt_pad;
    acc.x += p[t]; acc.y += p[a.n_tgt_pad + t]; acc.z += p[2 * a.n_tgt_pad + t];
  }
  const double4 tp = a.pos[ti];
  const double b = tp.w;
  double vx = a.vec[3 * ti] * b, vy = a.vec[3 * ti + 1] * b, vz = a.vec[3 * ti + 2] * b;
  double wx = 0, wy = 0, wz = 0;
  if constexpr (KIND == KIND_TT_TR) { wx = a.vec2[3 * ti] * b; wy = a.vec2[3 * ti + 1] * b; wz = a.vec2[3 * ti + 2] * b; }
  if (a.in_plane) { vz = 0.0; wz = 0.0; }
  self_term<KIND, WALL>(a.k, tp.z, vx, vy, vz, wx, wy, wz, acc);
  const double sc = a.prefactor * b;
  a.out[3 * t] = acc.x * sc; a.out[3 * t + 1] = acc.y * sc; a.out[3 * t + 2] = a.in_plane ? 0.0 : acc.z * sc;
}

// Positions: caller's (N,3) -> (x, y, z_eff, b).  Fuses shift_heights (mobility/mobility.py:52-64,
// clamp with `<=`) and damping_matrix_B (mobility/mobility.py:67-84, factor z/a for `z < a`), which
// the reference runs as an interpreted Python loop over N on every matvec.
__global__ void pack_positions_kernel(const double* r, long n, double a, int wall, double4* pos) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = r[3 * i], y = r[3 * i + 1], z = r[3 * i + 2];
  double4 p;
  p.x = x; p.y = y;
  if (wall) {
    p.z = (z <= a) ? a : z;
    p.w = (z < a) ? z / a : 1.0;
  } else {
    p.z = z; p.w = 1.0;
  }
  pos[i] = p;
}

// ---------------------------------------------------------------------------------------------
// Blob-blob soft repulsion, all pairs, minimal image (multi_bodies/forces_numba.py:12-55; the
// reference GPU twin multi_bodies/forces_pycuda.py:66-118 is float32, this one is fp64).
// Same skeleton: lane = target, 4 waves split the LDS tile, fixed-order combine.
// ---------------------------------------------------------------------------------------------
struct ForceArgs {
  const double4* pos;
  double* out;       // [3 (tgt_end - tgt_begin)]
  double* partial;
  long n_src, tgt_begin, tgt_end, n_tgt_pad, chunk_len;
  int n_chunks;
  double Lx, Ly, Lz, iLx, iLy, iLz;
  double eps_over_b, inv_b, two_a;
  ExpConsts ec;
  const double* radii;   // per-blob radii (RADII variant: contact distance a_i + a_j, forces_numba.py:73-122) or nullptr
};

template <bool PERIODIC, bool RADII = false>
__global__ __launch_bounds__(kBlock) void force_sweep_kernel(const ForceArgs a) {
  __shared__ double4 tile[kTile];
  __shared__ double red[(kWaves - 1) * 3 * 64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const long ti = a.tgt_begin + 64L * blockIdx.x + lane;
  const bool valid = ti < a.tgt_end;
  const double4 tp = a.pos[valid ? ti : a.tgt_end - 1];
  double ra = 0.0;
  if constexpr (RADII) ra = a.radii[valid ? ti : a.tgt_end - 1];
  const long c0 = (long)blockIdx.y * a.chunk_len;
  long c1 = c0 + a.chunk_len;
  if (c1 > a.n_src) c1 = a.n_src;
  double fx = 0, fy = 0, fz = 0;
  for (long j0 = c0; j0 < c1; j0 += kTile) {
    const int n = (int)((c1 - j0 < kTile) ? (c1 - j0) : kTile);
    __syncthreads();
    for (int t = threadIdx.x; t < n; t += kBlock) {
      double4 p = a.pos[j0 + t];
      if constexpr (RADII) p.w = a.radii[j0 + t];     // w is free here: forces use unclamped positions (b = 1)
      tile[t] = p;
    }
    __syncthreads();
    for (int s = wave; s < n; s += kWaves) {
      const double4 q = tile[s];
      double dx = q.x - tp.x, dy = q.y - tp.y, dz = q.z - tp.z;
      if constexpr (PERIODIC) {
        if (a.Lx > 0) dx = wrap_nearest(dx, a.Lx, a.iLx);
        if (a.Ly > 0) dy = wrap_nearest(dy, a.Ly, a.iLy);
        if (a.Lz > 0) dz = wrap_nearest(dz, a.Lz, a.iLz);
      }
      const double r2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
      const double ir = rsqrt_f64(r2);
      const double r = r2 * ir;
      // far: -(eps/b) exp(-(r-2a)/b)/r ; near (r <= 2a): -(eps/b)/max(r,1e-25) = -(eps/b) min(1/r, 1e25)
      const double two_a = RADII ? ra + q.w : a.two_a;
      // branch-free (sym_kernels.h pair_force): x = 0 exactly for r <= 2a, exp(0) = 1, min(1/r, 1e25) = 1/r beyond
      const double e = exp_nonpositive(a.ec, fmin((two_a - r) * a.inv_b, 0.0));
      double f0 = -a.eps_over_b * (e * fmin(ir, 1e25));
      if (j0 + s == ti) f0 = 0.0;  // i == j (r2 = 0 -> ir = inf; select, do not multiply)
      if (j0 + s == ti) { dx = 0.0; dy = 0.0; dz = 0.0; }
      fx = __builtin_fma(f0, dx, fx); fy = __builtin_fma(f0, dy, fy); fz = __builtin_fma(f0, dz, fz);
    }
  }
  if (wave > 0) {
    double* r = red + (wave - 1) * 3 * 64;
    r[lane] = fx; r[64 + lane] = fy; r[128 + lane] = fz;
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < kWaves - 1; ++w) {
    const double* r = red + w * 3 * 64;
    fx += r[lane]; fy += r[64 + lane]; fz += r[128 + lane];
  }
  if (a.n_chunks == 1) {
    if (!valid) return;
    const long o = 3 * (ti - a.tgt_begin);
    a.out[o] = fx; a.out[o + 1] = fy; a.out[o + 2] = fz;
  } else {
    const long col = 64L * blockIdx.x + lane;
    double* p = a.partial + (long)blockIdx.y * 3 * a.n_tgt_pad;
    p[col] = fx; p[a.n_tgt_pad + col] = fy; p[2 * a.n_tgt_pad + col] = fz;
  }
}

__global__ __launch_bounds__(256) void force_finalize_kernel(const ForceArgs a) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (a.tgt_begin + t >= a.tgt_end) return;
  double fx = 0, fy = 0, fz = 0;
  for (int c = 0; c < a.n_chunks; ++c) {
    const double* p = a.partial + (long)c * 3 * a.n_tgt_pad;
    fx += p[t]; fy += p[a.n_tgt_pad + t]; fz += p[2 * a.n_tgt_pad + t];
  }
  a.out[3 * t] = fx; a.out[3 * t + 1] = fy; a.out[3 * t + 2] = fz;
}

}  // namespace rmb
