// dense_kernels.h -- small dense per-body mobility blocks (preconditioner building block).
//
// The reference's block-diagonal preconditioner factorises, for every rigid body on its own, the dense
// blob mobility of that body's blobs (body/body.py:186-191 -> mobility/mobility.py:1018-1116
// single_wall_fluid_mobility, or :967-1013 rotne_prager_tensor without wall), then N = (K^T M^-1 K)^-1
// (multi_bodies/multi_bodies.py:516-531).  Here all bodies of one shape are built in one launch:
// workgroup = body, threads sweep its n_b x n_b blob pairs, each 3x3 block is obtained by applying the
// matrix-free pair operator of pair_ops.h to the three unit vectors.  Same regularisation as the
// products: M_body = B M(z_eff) B.
#pragma once
#include "pair_blocks.h"

namespace rmb {

struct DenseArgs {
  const double4* pos;      // packed positions of ALL blobs
  const long* first_blob;  // [n_bodies] index of each body's first blob (its blobs are contiguous)
  double* out;             // [n_bodies][3 n_b][3 n_b] row-major
  int n_b;
  long n_bodies;
  double prefactor;        // 1/(8 pi eta)
  PairConsts k;
};

template <bool WALL>
__global__ __launch_bounds__(256) void body_dense_tt_kernel(const DenseArgs a) {
  const long body = blockIdx.x;
  const long base = a.first_blob[body];
  const int nb = a.n_b;
  const int ld = 3 * nb;
  double* M = a.out + body * (long)ld * ld;
  for (long p = (long)blockIdx.y * blockDim.x + threadIdx.x; p < (long)nb * nb; p += (long)blockDim.x * gridDim.y) {
    const int i = (int)(p / nb), j = (int)(p - (long)i * nb);
    const double4 pi = a.pos[base + i];
    const double4 pj = a.pos[base + j];
    const double sc = a.prefactor * pi.w * pj.w;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double vx = c == 0 ? 1.0 : 0.0, vy = c == 1 ? 1.0 : 0.0, vz = c == 2 ? 1.0 : 0.0;
      Vec3 u = {0.0, 0.0, 0.0};
      if (i == j) self_term<KIND_TT, WALL>(a.k, pi.z, vx, vy, vz, 0, 0, 0, u);
      else pair_apply<KIND_TT, WALL>(a.k, pi.x - pj.x, pi.y - pj.y, pi.z - pj.z, pi.z, pj.z, vx, vy, vz, 0, 0, 0, u);
      M[(long)(3 * i + 0) * ld + 3 * j + c] = u.x * sc;
      M[(long)(3 * i + 1) * ld + 3 * j + c] = u.y * sc;
      M[(long)(3 * i + 2) * ld + 3 * j + c] = u.z * sc;
    }
  }
}

}  // namespace rmb
