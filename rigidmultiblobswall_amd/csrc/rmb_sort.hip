// rmb_sort.hip -- spatial ordering of the blobs for the force kernel's tile culling.
//
// sym_force_kernel skips a tile pair whose bounding boxes are further apart than the range of the exponential
// (sym_kernels.h: tile_gap2).  How much that skips depends on how compact a 64-blob tile is in space, i.e. on the ORDER
// in which the caller lists the blobs: a monolayer listed in lattice order loses 99 % of its tile pairs, the same
// monolayer listed in random order none (profiles/r3_force_tile_culling.txt).  The reference's answer to the short
// range of the force is a k-d tree (multi_bodies/forces_numba.py:141-271, `tree_numba`); here the blobs are sorted once
// per configuration along a Morton curve (16 bits per direction on an isotropic grid over the bounding box; wrapped
// coordinates in pseudo-periodic directions), the force kernel runs on the sorted copy, and its finalize kernel
// writes every result back to the caller's index.  The sort itself is the library radix sort (rocPRIM, 48-bit keys);
// everything else is four small kernels.  Result: the force on every blob is the same sum of the same pair terms,
// added in another order (equal to rounding, not to the bit: option "force_sort" = 0 keeps the caller's order).
#include "rmb_internal.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

namespace {

// box[0..2] = min, box[3..5] = max over all tiles' bounding boxes (tile_bounds layout: 6 doubles per tile)
__global__ __launch_bounds__(256) void global_bounds_kernel(const double* tile_bounds, long n_tiles, double* box) {
  __shared__ double lo[3][256], hi[3][256];
  double l[3] = {1e300, 1e300, 1e300}, h[3] = {-1e300, -1e300, -1e300};
  for (long t = threadIdx.x; t < n_tiles; t += 256)
    for (int d = 0; d < 3; ++d) {
      l[d] = fmin(l[d], tile_bounds[6 * t + d]);
      h[d] = fmax(h[d], tile_bounds[6 * t + 3 + d]);
    }
  for (int d = 0; d < 3; ++d) { lo[d][threadIdx.x] = l[d]; hi[d][threadIdx.x] = h[d]; }
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off)
      for (int d = 0; d < 3; ++d) {
        lo[d][threadIdx.x] = fmin(lo[d][threadIdx.x], lo[d][threadIdx.x + off]);
        hi[d][threadIdx.x] = fmax(hi[d][threadIdx.x], hi[d][threadIdx.x + off]);
      }
    __syncthreads();
  }
  if (threadIdx.x == 0)
    for (int d = 0; d < 3; ++d) { box[d] = lo[d][0]; box[3 + d] = hi[d][0]; }
}

// bounding box of every 64-blob tile (same layout as rmb::tile_bounds_kernel, which lives with the force kernel)
__global__ __launch_bounds__(64) void tile_box_kernel(const double4* pos, long n, double* bounds) {
  const long T = blockIdx.x;
  const long i = 64 * T + threadIdx.x;
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  if (i < n) {
    const double4 p = pos[i];
    lo[0] = hi[0] = p.x; lo[1] = hi[1] = p.y; lo[2] = hi[2] = p.z;
  }
  for (int d = 0; d < 3; ++d)
    for (int off = 32; off > 0; off >>= 1) {
      lo[d] = fmin(lo[d], __shfl_xor(lo[d], off));
      hi[d] = fmax(hi[d], __shfl_xor(hi[d], off));
    }
  if (threadIdx.x == 0)
    for (int d = 0; d < 3; ++d) { bounds[6 * T + d] = lo[d]; bounds[6 * T + 3 + d] = hi[d]; }
}

__device__ __forceinline__ unsigned long long spread16(unsigned v) {   // bit k of v -> bit 3k
  unsigned long long x = v & 0xffffu;
  x = (x | (x << 32)) & 0x00ff00000000ffffULL;    // not needed for 16 bits, kept general up to 21
  x = (x | (x << 16)) & 0x00ff0000ff0000ffULL;
  x = (x | (x << 8)) & 0xf00f00f00f00f00fULL;
  x = (x | (x << 4)) & 0x30c30c30c30c30c3ULL;
  x = (x | (x << 2)) & 0x9249249249249249ULL;
  return x;
}

// Morton key on an isotropic grid of 2^16 cells along the longest side of the box (a monolayer then gets a
// two-dimensional curve: its z index stays 0); a pseudo-periodic direction is wrapped into [0, L) first.
__global__ __launch_bounds__(256) void morton_keys_kernel(const double4* pos, long n, const double* box, double Lx, double Ly, double Lz,
                                                         unsigned long long* keys, unsigned* vals) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double L[3] = {Lx, Ly, Lz};
  const double4 p = pos[i];
  const double x[3] = {p.x, p.y, p.z};
  double lo[3], range = 0.0;
  for (int d = 0; d < 3; ++d) {
    lo[d] = L[d] > 0.0 ? 0.0 : box[d];
    const double r = L[d] > 0.0 ? L[d] : box[3 + d] - box[d];
    range = fmax(range, r);
  }
  const double inv_h = range > 0.0 ? 65536.0 / range : 0.0;
  unsigned long long key = 0;
  for (int d = 0; d < 3; ++d) {
    double w = x[d];
    if (L[d] > 0.0) w -= floor(w / L[d]) * L[d];
    double q = (w - lo[d]) * inv_h;
    q = q < 0.0 ? 0.0 : (q > 65535.0 ? 65535.0 : q);
    key |= spread16((unsigned)q) << d;
  }
  keys[i] = key;
  vals[i] = (unsigned)i;
}

__global__ __launch_bounds__(256) void gather_positions_kernel(const double4* pos, const unsigned* perm, long n, double4* sorted) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sorted[i] = pos[perm[i]];
}

}  // namespace

namespace rmbi {

// Builds c->fpos (packed positions along the Morton curve), c->fperm (sorted slot -> caller's index) and the tile
// bounds of the SORTED tiles for the resident configuration.  Asynchronous on the context's stream; nothing crosses
// PCIe (the bounding box stays on the device).
int force_sort_positions(rmb_ctx* c) {
  const long n = c->n, tiles = (n + 63) / 64;
  if (n > 0xffffffffL) return fail(RMB_ERR_ARG, "force_sort: more than 2^32 blobs");
  const double4* pos = (const double4*)c->pos.p;
  if (int rc = c->tile_bounds.reserve((size_t)6 * tiles * sizeof(double))) return rc;
  if (int rc = c->fsort_box.reserve(6 * sizeof(double))) return rc;
  if (int rc = c->fsort_keys.reserve((size_t)2 * n * sizeof(unsigned long long))) return rc;
  if (int rc = c->fsort_vals.reserve((size_t)n * sizeof(unsigned))) return rc;
  if (int rc = c->fperm.reserve((size_t)n * sizeof(unsigned))) return rc;
  if (int rc = c->fpos.reserve((size_t)n * sizeof(double4))) return rc;
  unsigned long long* keys_in = (unsigned long long*)c->fsort_keys.p;
  unsigned long long* keys_out = keys_in + n;
  hipLaunchKernelGGL(tile_box_kernel, dim3((unsigned)tiles), dim3(64), 0, c->stream, pos, n, (double*)c->tile_bounds.p);
  hipLaunchKernelGGL(global_bounds_kernel, dim3(1), dim3(256), 0, c->stream, (const double*)c->tile_bounds.p, tiles, (double*)c->fsort_box.p);
  hipLaunchKernelGGL(morton_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, pos, n, (const double*)c->fsort_box.p,
                     c->L[0], c->L[1], c->L[2], keys_in, (unsigned*)c->fsort_vals.p);
  RMB_HIP(hipGetLastError());
  size_t tmp_bytes = 0;
  RMB_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, keys_out, (unsigned*)c->fsort_vals.p, (unsigned*)c->fperm.p, (size_t)n, 0, 48,
                                    c->stream));
  if (int rc = c->fsort_tmp.reserve(tmp_bytes ? tmp_bytes : 16)) return rc;
  RMB_HIP(rocprim::radix_sort_pairs(c->fsort_tmp.p, tmp_bytes, keys_in, keys_out, (unsigned*)c->fsort_vals.p, (unsigned*)c->fperm.p, (size_t)n, 0,
                                    48, c->stream));
  hipLaunchKernelGGL(gather_positions_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, pos, (const unsigned*)c->fperm.p, n,
                     (double4*)c->fpos.p);
  hipLaunchKernelGGL(tile_box_kernel, dim3((unsigned)tiles), dim3(64), 0, c->stream, (const double4*)c->fpos.p, n, (double*)c->tile_bounds.p);
  RMB_HIP(hipGetLastError());
  return 0;
}

}  // namespace rmbi
