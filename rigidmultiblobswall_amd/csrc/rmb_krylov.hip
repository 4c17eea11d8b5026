// rmb_krylov.hip -- the O(N) pieces of the rigid-body saddle-point solve that sit between two blob sweeps, as HIP
// kernels on the context's stream (gfx950, fp64):
//
//   rmb_krylov_orthogonalize_device   one Arnoldi step's Gram-Schmidt: two classical passes against the Krylov basis,
//                                     the new Hessenberg column, |w| and the normalised next basis vector -- 4 launches
//                                     (what scipy's gmres does inside, general_application_utils.py:608-627 calls it;
//                                     as torch ops: 2 gemv-T, 2 gemv-N, add, norm, div = 7 launches, the two gemv-T of
//                                     a (j+1) x n basis alone 26 us at n = 4608 -- profiles/r4_gmres_graph.txt)
//   rmb_block_apply_device            y1_b = beta1 y1_b + alpha (A11_b x1_b + A12_b x2_b),
//                                     y2_b = beta2 y2_b + alpha (A21_b x1_b + A22_b x2_b)   for every body b,
//                                     one launch: the block-diagonal preconditioner (multi_bodies.py:548-560, four
//                                     batched GEMMs as torch ops) and the K / K^T products of the operator
//                                     (multi_bodies.py:327-375, 424-471; two batched GEMMs) -- blocks are addressed
//                                     with (batch, row, column) strides, so K^T is K with the strides exchanged.
//
// Both are HBM-latency-bound helpers of a few microseconds; they exist because up to a few thousand blobs the solver
// loop AROUND the sweep costs several times the sweep (profiles/r4_gmres_graph.txt).  Every reduction runs in a fixed
// order: results are bit-reproducible.
#include "rmb_internal.h"
#include "block_rows.h"

#include <cmath>
#include <cstring>

namespace rmbi {
namespace {

constexpr int kKrT = 256;            // threads per workgroup of the Gram-Schmidt kernels (4 waves)
constexpr int kKrWaves = kKrT / 64;
constexpr int kKrMaxRows = 256;      // basis vectors one call orthogonalises against (GMRES restart length + 1)
constexpr long kKrMaxChunk = 4096;   // doubles of w a workgroup keeps in LDS (32 KB)

struct OrthoArgs {
  long n, rows, ldv, chunk, n_chunks;
  int low_sync;     // 1: the first update launch also leaves |w1|^2 per chunk in part3; the LAST launch does the second update, the
                    // norm (|w2|^2 = |w1|^2 - |h2|^2: the basis is orthonormal) and the normalisation -- one launch less
  long n_part1;     // partials per row of pass 1: n_chunks, or the body count when the operator's finishing launch took them
  const double* V;
  double* w;
  double* col;      // rows coefficients, then |w|
  double* col_host; // the same column once more, in page-locked device-mapped host memory, or nullptr
  double* v_next;
  double* part1;    // [rows][n_chunks] partial dots of pass 1
  double* part2;    // [rows][n_chunks] partial dots of pass 2
  double* part3;    // [n_chunks] partial |w|^2
  double* h1;       // [rows] coefficients of pass 1
};

// These helpers are LATENCY-bound, not bandwidth-bound (a mid-size deck's whole basis is a few MB): every loop over basis
// rows / partials is written so that a group of independent loads is in flight before the first dependent instruction --
// four rows per wave and pass in the dots, eight rows per thread in the update, 64 partials per wave in the sums.  Written
// one load, one use per iteration they cost 8-13 us per launch at 256-512 bodies instead of ~5 (profiles/r5_gmres_step.txt).

// partial dots of this workgroup's chunk (in LDS) with every basis row: wave q takes rows 4q .. 4q + 3, then 4 (q + 4) ..
__device__ __forceinline__ void chunk_dots(const OrthoArgs& a, const double* wl, long base, long len, double* part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long r0 = 4L * wave; r0 < a.rows; r0 += 4L * kKrWaves) {
    const double* row[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) row[q] = a.V + (r0 + q < a.rows ? r0 + q : r0) * a.ldv + base;      // past the last row: row r0 again, not stored
    double s[4];
    four_row_sums<true>(row[0], row[1], row[2], row[3], wl, len, s, [](long e) { return e; });
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double t = wave_sum(s[q]);
      if (lane == 0 && r0 + q < a.rows) part[(r0 + q) * a.n_chunks + blockIdx.x] = t;
    }
  }
}

// coefficients = fixed-order sums of the partials over the chunks, into LDS: wave q takes rows 4q .. 4q + 3, ..., the lanes
// stride over the chunks (lane l adds partials l, l + 64, ... in that order), then the butterfly -- the same order in every
// workgroup and launch, so every workgroup holds the same bits
__device__ __forceinline__ void reduce_partials(const OrthoArgs& a, const double* part, long count, double* hl) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = (int)(blockDim.x >> 6);
  for (long r0 = 4L * wave; r0 < a.rows; r0 += 4L * n_waves) {
    const double* row[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) row[q] = part + (r0 + q < a.rows ? r0 + q : r0) * count;
    double s[4];
    four_row_sums<false>(row[0], row[1], row[2], row[3], nullptr, count, s, [](long e) { return e; });
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double t = wave_sum(s[q]);
      if (lane == 0 && r0 + q < a.rows) hl[r0 + q] = t;
    }
  }
}

// sum of n values by one wave (lanes stride, then the butterfly); every lane of the wave must call it; lane 0 holds the sum
__device__ __forceinline__ double wave_strided_sum(const double* v, long n) {
  double s = 0.0;
  for (long c = threadIdx.x & 63; c < n; c += 64) s += v[c];
  return wave_sum(s);
}

// wl[e] -= sum_r hl[r] V[r][base + e], rows in ascending order; eight rows' loads in flight per thread
__device__ __forceinline__ void chunk_update(const OrthoArgs& a, double* wl, const double* hl, long base, long len) {
  for (long e = threadIdx.x; e < len; e += kKrT) {
    double s = wl[e];
    const double* col = a.V + base + e;
    long r = 0;
    for (; r + 8 <= a.rows; r += 8) {
      double v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = col[(r + q) * a.ldv];
#pragma unroll
      for (int q = 0; q < 8; ++q) s -= hl[r + q] * v[q];
    }
    for (; r < a.rows; ++r) s -= hl[r] * col[r * a.ldv];
    wl[e] = s;
  }
}

__global__ __launch_bounds__(kKrT) void ortho_dots_kernel(const OrthoArgs a) {
  extern __shared__ double lds[];
  double* wl = lds;
  const long base = blockIdx.x * a.chunk;
  const long len = (a.n - base) < a.chunk ? (a.n - base) : a.chunk;
  for (long e = threadIdx.x; e < len; e += kKrT) wl[e] = a.w[base + e];
  __syncthreads();
  chunk_dots(a, wl, base, len, a.part1);
}

// pass: 1 = subtract the pass-1 projection and take the pass-2 dots; 2 = subtract the pass-2 projection, |w|^2, column
template <int PASS>
__global__ __launch_bounds__(kKrT) void ortho_update_kernel(const OrthoArgs a) {
  extern __shared__ double lds[];
  double* wl = lds;
  double* hl = lds + a.chunk;
  const long base = blockIdx.x * a.chunk;
  const long len = (a.n - base) < a.chunk ? (a.n - base) : a.chunk;
  for (long e = threadIdx.x; e < len; e += kKrT) wl[e] = a.w[base + e];
  reduce_partials(a, PASS == 1 ? a.part1 : a.part2, PASS == 1 ? a.n_part1 : a.n_chunks, hl);
  __syncthreads();
  if (blockIdx.x == 0) {
    for (long r = threadIdx.x; r < a.rows; r += kKrT) {
      if (PASS == 1) a.h1[r] = hl[r]; else a.col[r] = a.h1[r] + hl[r];
    }
  }
  chunk_update(a, wl, hl, base, len);
  __syncthreads();
  for (long e = threadIdx.x; e < len; e += kKrT) a.w[base + e] = wl[e];
  if (PASS == 1) chunk_dots(a, wl, base, len, a.part2);
  if (PASS == 2 || a.low_sync) {
    __shared__ double wsum[kKrWaves];
    double s = 0.0;
    for (long e = threadIdx.x; e < len; e += kKrT) s += wl[e] * wl[e];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int q = 0; q < kKrWaves; ++q) t += wsum[q];
      a.part3[blockIdx.x] = t;
    }
  }
}

__global__ __launch_bounds__(kKrT) void ortho_normalise_kernel(const OrthoArgs a) {
  __shared__ double nrm;
  if (threadIdx.x < 64) {
    const double s = wave_strided_sum(a.part3, a.n_chunks);
    if (threadIdx.x == 0) nrm = sqrt(s);
  }
  if (threadIdx.x == 0) {
    if (blockIdx.x == 0) {
      a.col[a.rows] = nrm;
      if (a.col_host) a.col_host[a.rows] = nrm;
    }
  }
  if (a.col_host && blockIdx.x == 0)      // col[0..rows) was written by the launch before this one
    for (long r = threadIdx.x; r < a.rows; r += kKrT) a.col_host[r] = a.col[r];
  __syncthreads();
  const double inv = 1.0 / nrm;      // |w| = 0 (exact breakdown): inf / nan in v_next, as w / |w| gives; the caller stops on col[rows] == 0
  const long base = blockIdx.x * a.chunk;
  const long len = (a.n - base) < a.chunk ? (a.n - base) : a.chunk;
  for (long e = threadIdx.x; e < len; e += kKrT) a.v_next[base + e] = a.w[base + e] * inv;
}

// The normalisation with the NEXT iteration's first launch fused in (rmb_rigid_gmres_device, rmb_rigid_lanczos_device):
// workgroup = body.  Every workgroup re-sums the chunk partials of |w|^2 (fixed order), normalises ITS slices of w -- the
// body's 3 n_b rows of the lambda part and, for the saddle-point system, its 6 rows of the U part: together the workgroups
// cover the whole vector -- into v_next, keeps them in LDS and applies the body's blocks to them: z = P^-1 v_next with the
// four blocks of [[M_b, -K], [-K^T, 0]]^-1 (GMRES), or z = L_b^-T v_next (Lanczos).
struct NormPcArgs {
  OrthoArgs o;
  long n_bodies, r1, r2, n_top;      // the body's slices: r1 rows at b * r1, r2 rows at n_top + b * r2 (n_top = n_bodies * r1)
  BlockRef a11, a12, a21, a22;
  double* z;
};

__global__ __launch_bounds__(1024) void ortho_normalise_pc_kernel(const NormPcArgs a) {
  extern __shared__ double xl[];          // r1 + r2: the body's slices of v_next; then r1 + r2 row sums (two_by_two_rows)
  __shared__ double nrm;
  const long b = blockIdx.x;
  if (threadIdx.x < 64) {
    const double s = wave_strided_sum(a.o.part3, a.o.n_chunks);
    if (threadIdx.x == 0) nrm = sqrt(s);
  }
  if (threadIdx.x == 0) {
    if (b == 0) {
      a.o.col[a.o.rows] = nrm;
      if (a.o.col_host) a.o.col_host[a.o.rows] = nrm;
    }
  }
  if (a.o.col_host && b == 0)
    for (long r = threadIdx.x; r < a.o.rows; r += blockDim.x) a.o.col_host[r] = a.o.col[r];
  __syncthreads();
  const double inv = 1.0 / nrm;
  const long r1 = a.r1, rows = a.r1 + a.r2;
  for (long k = threadIdx.x; k < rows; k += blockDim.x) {
    const long e = k < r1 ? b * r1 + k : a.n_top + a.r2 * b + (k - r1);
    const double v = a.o.w[e] * inv;
    a.o.v_next[e] = v;
    xl[k] = v;
  }
  __syncthreads();
  two_by_two_rows(a.a11, a.a12, a.a21, a.a22, b, a.r1, a.r1, a.r2, a.r2, xl, xl + rows,
                  [&](long row, double sum) { a.z[row < r1 ? b * r1 + row : a.n_top + a.r2 * b + (row - r1)] = sum; });
}

// ---- the low-synchronisation ending of a step (the native loops): second update + norm + normalisation in ONE launch ----------
// After the first update launch  h2 = V^T w1  and  |w1|^2  are known as per-chunk partials.  With an orthonormal basis
// |w1 - V h2|^2 = |w1|^2 - |h2|^2 (h2 is rounding-sized after the first pass, so the subtraction is benign; a negative result --
// an exact breakdown -- counts as zero and the loops stop or hand back as they do on |w| = 0), so the launch that subtracts the
// second projection can normalise at once: no separate norm / normalisation launch.  Every workgroup re-derives h2 and the norm
// from the partials in the same fixed order.
__device__ __forceinline__ double low_sync_norm(const OrthoArgs& a, const double* hl, double* scratch) {
  // wave 0: |w1|^2 - |h2|^2; everybody gets the root through `scratch` (one shared double) after a barrier
  if (threadIdx.x < 64) {
    const double s = wave_strided_sum(a.part3, a.n_chunks);
    double h = 0.0;
    for (long r = threadIdx.x; r < a.rows; r += 64) h += hl[r] * hl[r];
    h = wave_sum(h);
    if (threadIdx.x == 0) { const double d = s - h; *scratch = d > 0.0 ? sqrt(d) : (d == d ? 0.0 : d); }
  }
  __syncthreads();
  return *scratch;
}

// thread's entry e of the vector: w2 = w1 - sum_r hl[r] V[r][e], rows ascending, eight loads in flight
__device__ __forceinline__ double second_update(const OrthoArgs& a, const double* hl, long e) {
  double s = a.w[e];
  const double* col = a.V + e;
  long r = 0;
  for (; r + 8 <= a.rows; r += 8) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = col[(r + q) * a.ldv];
#pragma unroll
    for (int q = 0; q < 8; ++q) s -= hl[r + q] * v[q];
  }
  for (; r < a.rows; ++r) s -= hl[r] * col[r * a.ldv];
  return s;
}

__device__ __forceinline__ void low_sync_column(const OrthoArgs& a, const double* hl, double nrm) {
  for (long r = threadIdx.x; r < a.rows; r += blockDim.x) {
    const double v = a.h1[r] + hl[r];
    a.col[r] = v;
    if (a.col_host) a.col_host[r] = v;
  }
  if (threadIdx.x == 0) {
    a.col[a.rows] = nrm;
    if (a.col_host) a.col_host[a.rows] = nrm;
  }
}

__global__ __launch_bounds__(kKrT) void ortho_last_kernel(const OrthoArgs a) {
  extern __shared__ double lds[];
  double* hl = lds;                       // rows
  __shared__ double nrm_s;
  reduce_partials(a, a.part2, a.n_chunks, hl);
  __syncthreads();
  const double nrm = low_sync_norm(a, hl, &nrm_s);
  if (blockIdx.x == 0) low_sync_column(a, hl, nrm);
  const double inv = 1.0 / nrm;
  const long base = blockIdx.x * a.chunk;
  const long len = (a.n - base) < a.chunk ? (a.n - base) : a.chunk;
  for (long e = threadIdx.x; e < len; e += kKrT) {
    const double w2 = second_update(a, hl, base + e);
    a.w[base + e] = w2;
    a.v_next[base + e] = w2 * inv;
  }
}

__global__ __launch_bounds__(1024) void ortho_last_pc_kernel(const NormPcArgs a) {
  extern __shared__ double xl[];          // r1 + r2 slices of v_next, r1 + r2 row sums (two_by_two_rows), then `rows` coefficients
  __shared__ double nrm_s;
  const long b = blockIdx.x;
  const long r1 = a.r1, rws = a.r1 + a.r2;
  double* hl = xl + 2 * rws;
  reduce_partials(a.o, a.o.part2, a.o.n_chunks, hl);
  __syncthreads();
  const double nrm = low_sync_norm(a.o, hl, &nrm_s);
  if (b == 0) low_sync_column(a.o, hl, nrm);
  const double inv = 1.0 / nrm;
  for (long k = threadIdx.x; k < rws; k += blockDim.x) {
    const long e = k < r1 ? b * r1 + k : a.n_top + a.r2 * b + (k - r1);
    const double w2 = second_update(a.o, hl, e);
    a.o.w[e] = w2;
    const double v = w2 * inv;
    a.o.v_next[e] = v;
    xl[k] = v;
  }
  __syncthreads();
  two_by_two_rows(a.a11, a.a12, a.a21, a.a22, b, a.r1, a.r1, a.r2, a.r2, xl, xl + rws,
                  [&](long row, double sum) { a.z[row < r1 ? b * r1 + row : a.n_top + a.r2 * b + (row - r1)] = sum; });
}

// (Round 5 measured the whole step in ONE workgroup for systems of up to 6144 unknowns -- workgroup barriers instead of
//  kernel boundaries -- and dropped it: one CU pulls the (rows x n) basis four times through its own L2 port, 17-24 us per
//  step at 17 basis rows growing to 40 us at 60 (wave = row), 23-40 us with thread = unknown, against 4 x 4.5 us for the
//  four launches below, whose workgroups spread over the chip; and dependent launches of one stream start back to back
//  once they are queued -- what a tiny kernel costs is its ~4.5 us floor, not a gap.  profiles/r5_gmres_step.txt.)

// ---- batched two-by-two block matvec ---------------------------------------------------------------------------
struct BlockApplyArgs {
  long n_batch, r1, c1, r2, c2;
  BlockRef a11, a12, a21, a22;
  const double* x1; const double* x2;
  double* y1; double* y2;
  double alpha, beta1, beta2;
};

__global__ __launch_bounds__(1024) void block_apply_kernel(const BlockApplyArgs a) {
  extern __shared__ double xl[];          // x1_b then x2_b; then r1 + r2 row sums (two_by_two_rows)
  const long b = blockIdx.x;
  for (long k = threadIdx.x; k < a.c1; k += blockDim.x) xl[k] = a.x1[b * a.c1 + k];
  for (long k = threadIdx.x; k < a.c2; k += blockDim.x) xl[a.c1 + k] = a.x2[b * a.c2 + k];
  __syncthreads();
  two_by_two_rows(a.a11, a.a12, a.a21, a.a22, b, a.r1, a.c1, a.r2, a.c2, xl, xl + a.c1 + a.c2, [&](long row, double sum) {
    const bool top = row < a.r1;
    double* y = top ? a.y1 + b * a.r1 + row : a.y2 + b * a.r2 + (row - a.r1);
    const double beta = top ? a.beta1 : a.beta2;
    *y = (beta == 0.0 ? 0.0 : beta * *y) + a.alpha * sum;
  });
}

}  // namespace
}  // namespace rmbi

using namespace rmbi;

extern "C" {

int rmb_krylov_orthogonalize_device(rmb_ctx* c, long n, long rows, const double* V_dev, long ldv, double* w_dev, double* col_dev,
                                    double* v_next_dev) {
  return rmb_krylov_orthogonalize2_device(c, n, rows, V_dev, ldv, w_dev, col_dev, v_next_dev, nullptr);
}

int rmb_krylov_orthogonalize2_device(rmb_ctx* c, long n, long rows, const double* V_dev, long ldv, double* w_dev, double* col_dev,
                                     double* v_next_dev, double* col_mapped_dev) {
  return krylov_orthogonalize_impl(c, n, rows, V_dev, ldv, w_dev, col_dev, v_next_dev, col_mapped_dev, nullptr);
}

}  // extern "C"

namespace rmbi {
// chunk length / count of the Gram-Schmidt launches and the scratch they share: functions of n only
static void krylov_layout(long n, long* chunk_out, long* n_chunks_out, size_t* need_out) {
  // chunks of 256 doubles (one per thread: small systems are latency-bound, 4608 unknowns are 18 workgroups instead of 5)
  // while that gives at most 256 workgroups, larger ones (up to what fits LDS) beyond: every workgroup re-sums the
  // per-chunk partials, so their number stays bounded
  long chunk = 256;
  while ((n + chunk - 1) / chunk > 256 && chunk < kKrMaxChunk) chunk *= 2;
  const long n_chunks = (n + chunk - 1) / chunk;
  *chunk_out = chunk;
  *n_chunks_out = n_chunks;
  // Sized for the LARGEST basis (kKrMaxRows) whatever `rows` is: the scratch then depends on n only and never moves
  // while a solve walks up its iteration indices.  rigid.py captures one hipGraph per iteration index with these
  // addresses baked in; a buffer that grew with `rows` was freed and reallocated every 2-3 indices, and the graphs
  // captured for lower indices replayed into freed memory.  ~1 MB for up to 65 536 unknowns, + 512 KB for the
  // per-body partials of the fused first pass.
  *need_out = ((size_t)2 * kKrMaxRows * n_chunks + n_chunks + kKrMaxRows + (size_t)kKrMaxRows * kKrBodyPartialsMax) * sizeof(double);
}

int krylov_body_partials(rmb_ctx* c, long n, double** part_out) {
  long chunk, n_chunks; size_t need;
  krylov_layout(n, &chunk, &n_chunks, &need);
  if (int rc = c->krylov.reserve(need)) return rc;
  *part_out = (double*)c->krylov.p + (size_t)2 * kKrMaxRows * n_chunks + n_chunks + kKrMaxRows;
  return 0;
}

int krylov_orthogonalize_impl(rmb_ctx* c, long n, long rows, const double* V_dev, long ldv, double* w_dev, double* col_dev,
                              double* v_next_dev, double* col_mapped_dev, const PcBlocks* pc, long part1_bodies, bool low_sync) {
  if (!c) return fail(RMB_ERR_ARG, "null context");
  if (n < 1 || rows < 1 || rows > kKrMaxRows || ldv < n)
    return fail(RMB_ERR_ARG, "rmb_krylov_orthogonalize_device: need n >= 1, 1 <= rows <= 256, ldv >= n");
  if (!V_dev || !w_dev || !col_dev || !v_next_dev) return fail(RMB_ERR_ARG, "null pointer");
  RMB_HIP(hipSetDevice(c->device));
  OrthoArgs a;
  a.n = n; a.rows = rows; a.ldv = ldv;
  a.V = V_dev; a.w = w_dev; a.col = col_dev; a.v_next = v_next_dev; a.col_host = col_mapped_dev;
  if (part1_bodies < 0 || part1_bodies > kKrBodyPartialsMax) return fail(RMB_ERR_ARG, "krylov_orthogonalize_impl: too many per-body partials (internal)");
  long chunk; size_t need;
  krylov_layout(n, &chunk, &a.n_chunks, &need);
  a.chunk = chunk;
  if (int rc = c->krylov.reserve(need)) return rc;
  double* base = (double*)c->krylov.p;
  a.part1 = base;
  a.part2 = a.part1 + rows * a.n_chunks;
  a.part3 = a.part2 + rows * a.n_chunks;
  a.h1 = base + (size_t)2 * kKrMaxRows * a.n_chunks + a.n_chunks;
  a.n_part1 = a.n_chunks;
  a.low_sync = low_sync && c->opt_krylov_low_sync ? 1 : 0;
  if (part1_bodies > 0) { a.part1 = a.h1 + kKrMaxRows; a.n_part1 = part1_bodies; }       // krylov_body_partials' buffer
  const dim3 grid((unsigned)a.n_chunks), block(kKrT);
  const size_t lds_w = (size_t)chunk * sizeof(double), lds_wh = lds_w + (size_t)rows * sizeof(double);
  if (part1_bodies == 0) hipLaunchKernelGGL(ortho_dots_kernel, grid, block, lds_w, c->stream, a);
  hipLaunchKernelGGL(ortho_update_kernel<1>, grid, block, lds_wh, c->stream, a);
  if (a.low_sync) {
    if (pc) {
      if (pc->n_bodies * (pc->r1 + pc->r2) != n) return fail(RMB_ERR_ARG, "krylov_orthogonalize_impl: the blocks do not cover the vector (internal)");
      NormPcArgs q;
      q.o = a; q.n_bodies = pc->n_bodies; q.r1 = pc->r1; q.r2 = pc->r2; q.n_top = pc->n_bodies * pc->r1;
      q.a11 = pc->a11; q.a12 = pc->a12; q.a21 = pc->a21; q.a22 = pc->a22; q.z = pc->z;
      const long rws = pc->r1 + pc->r2;
      const bool any_wave = wave_rows(q.a11, q.r1, q.r1) || wave_rows(q.a12, q.r1, q.r2) || wave_rows(q.a21, q.r2, q.r1) || wave_rows(q.a22, q.r2, q.r2);
      unsigned threads = two_by_two_threads(rws, any_wave);
      if (threads < 256) threads = 256;                       // four waves for the partial sums
      hipLaunchKernelGGL(ortho_last_pc_kernel, dim3((unsigned)pc->n_bodies), dim3(threads), (size_t)(2 * rws + rows) * sizeof(double), c->stream, q);
    } else {
      hipLaunchKernelGGL(ortho_last_kernel, grid, block, (size_t)rows * sizeof(double), c->stream, a);
    }
    RMB_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(ortho_update_kernel<2>, grid, block, lds_wh, c->stream, a);
  if (pc) {
    if (pc->n_bodies * (pc->r1 + pc->r2) != n) return fail(RMB_ERR_ARG, "krylov_orthogonalize_impl: the blocks do not cover the vector (internal)");
    NormPcArgs q;
    q.o = a; q.n_bodies = pc->n_bodies; q.r1 = pc->r1; q.r2 = pc->r2; q.n_top = pc->n_bodies * pc->r1;
    q.a11 = pc->a11; q.a12 = pc->a12; q.a21 = pc->a21; q.a22 = pc->a22; q.z = pc->z;
    const long rws = pc->r1 + pc->r2;
    const bool any_wave = wave_rows(q.a11, q.r1, q.r1) || wave_rows(q.a12, q.r1, q.r2) || wave_rows(q.a21, q.r2, q.r1) || wave_rows(q.a22, q.r2, q.r2);
    const unsigned threads = two_by_two_threads(rws, any_wave);
    hipLaunchKernelGGL(ortho_normalise_pc_kernel, dim3((unsigned)pc->n_bodies), dim3(threads), (size_t)(2 * rws) * sizeof(double), c->stream, q);
  } else {
    hipLaunchKernelGGL(ortho_normalise_kernel, grid, block, 0, c->stream, a);
  }
  RMB_HIP(hipGetLastError());
  return 0;
}
}  // namespace rmbi

extern "C" {

// Page-locked host memory mapped into the device's address space: what a kernel may store into so that the host reads a
// result after one event / stream wait, without a copy command (the Hessenberg column of an Arnoldi step; rmb_matvec does
// the same internally).  *host_out = the address the host reads, *dev_out = the address kernels use.
int rmb_host_mapped_alloc(size_t bytes, void** host_out, void** dev_out) {
  if (!host_out || !dev_out || bytes == 0) return fail(RMB_ERR_ARG, "rmb_host_mapped_alloc: null pointer / zero size");
  void* h = nullptr;
  RMB_HIP(hipHostMalloc(&h, bytes, hipHostMallocMapped));
  void* d = nullptr;
  const hipError_t e = hipHostGetDevicePointer(&d, h, 0);
  if (e != hipSuccess) { (void)hipHostFree(h); return fail(RMB_ERR_HIP, std::string("hipHostGetDevicePointer: ") + hipGetErrorString(e)); }
  memset(h, 0, bytes);
  *host_out = h; *dev_out = d;
  return 0;
}

int rmb_host_mapped_free(void* host) {
  if (!host) return 0;
  RMB_HIP(hipHostFree(host));
  return 0;
}

int rmb_block_apply_device(rmb_ctx* c, long n_batch, long r1, long c1, long r2, long c2, const rmb_block* a11, const rmb_block* a12,
                           const rmb_block* a21, const rmb_block* a22, const double* x1_dev, const double* x2_dev, double alpha,
                           double beta1, double* y1_dev, double beta2, double* y2_dev) {
  if (!c) return fail(RMB_ERR_ARG, "null context");
  if (n_batch < 0 || r1 < 0 || c1 < 0 || r2 < 0 || c2 < 0) return fail(RMB_ERR_ARG, "rmb_block_apply_device: negative size");
  if (n_batch == 0 || r1 + r2 == 0) return 0;
  if ((size_t)(c1 + c2 + r1 + r2) * sizeof(double) > 64 * 1024)
    return fail(RMB_ERR_ARG, "rmb_block_apply_device: c1 + c2 + r1 + r2 above 8192 (operand and row sums of one batch entry are kept in LDS)");
  if ((c1 > 0 && !x1_dev) || (c2 > 0 && !x2_dev) || (r1 > 0 && !y1_dev) || (r2 > 0 && !y2_dev)) return fail(RMB_ERR_ARG, "null pointer");
  RMB_HIP(hipSetDevice(c->device));
  BlockApplyArgs a;
  a.n_batch = n_batch; a.r1 = r1; a.c1 = c1; a.r2 = r2; a.c2 = c2;
  auto ref = [](const rmb_block* b) { return b && b->p ? BlockRef{b->p, b->batch_stride, b->row_stride, b->col_stride} : BlockRef{nullptr, 0, 0, 0}; };
  a.a11 = ref(a11); a.a12 = ref(a12); a.a21 = ref(a21); a.a22 = ref(a22);
  a.x1 = x1_dev; a.x2 = x2_dev; a.y1 = y1_dev; a.y2 = y2_dev;
  a.alpha = alpha; a.beta1 = beta1; a.beta2 = beta2;
  const long rows = r1 + r2;
  const bool any_wave = wave_rows(a.a11, r1, c1) || wave_rows(a.a12, r1, c2) || wave_rows(a.a21, r2, c1) || wave_rows(a.a22, r2, c2);
  const unsigned threads = two_by_two_threads(rows, any_wave);
  hipLaunchKernelGGL(block_apply_kernel, dim3((unsigned)n_batch), dim3(threads), (size_t)(c1 + c2 + rows) * sizeof(double), c->stream, a);
  RMB_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
