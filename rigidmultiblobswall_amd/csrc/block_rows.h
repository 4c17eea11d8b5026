// block_rows.h -- one batch entry's two-by-two block matvec with the operand in LDS: shared by the batched block kernel and
// the fused Krylov-step kernels (rmb_krylov.hip, rmb_rigid.hip).  Device code, gfx950.
#pragma once
#include "rmb_internal.h"

namespace rmbi {

// (block_apply_kernel, ortho_normalise_pc_kernel, lanczos_finish_kernel)
// Two ways to walk a block.  THREAD = ROW: a thread runs along its row; right for narrow blocks (the 6 columns of K, A12)
// and for blocks of up to 96 columns, where every row a workgroup touches stays in the CU's cache.  WAVE = ROW: the lanes
// of a wave take the columns 64 at a time (coalesced when the row is contiguous) and a butterfly sums them; right for wide
// blocks -- the 126 x 126 blocks of the reference's 42-blob shells, where thread = row makes every load instruction touch 64
// different cache lines -- and for the few long rows of K^T / A21 (6 rows of 3 n_b entries).  Four rows per wave are loaded
// before the first butterfly so that their loads are in flight together.
constexpr long kWaveRowMinCols = 97;

__host__ __device__ inline bool wave_rows(const BlockRef& m, long rows, long cols) {
  return m.p && cols >= kWaveRowMinCols && (m.cs == 1 || rows <= 8);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;       // lane 0 holds the sum
}

// s[q] = sum over e = lane, lane + 64, ... < len of row_q[at(e)] * (x ? x[e] : 1), q = 0 .. 3, in ascending e: sixteen loads
// (four rows, four strides) are issued before the first multiply.  Entries past `len` read a valid address and count as zero.
// at(e): where entry e of the (logical) vector sits in a row (identity for a contiguous chunk).
template <bool WITH_X, class At>
__device__ __forceinline__ void four_row_sums(const double* r0, const double* r1, const double* r2, const double* r3, const double* x, long len,
                                              double* s, At at) {
  const long lane = threadIdx.x & 63;
  s[0] = s[1] = s[2] = s[3] = 0.0;
  for (long e0 = lane; e0 < len; e0 += 256) {
    long ee[4];
    double xv[4], v[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long e = e0 + 64 * j;
      const bool ok = e < len;
      const long el = ok ? e : e0;
      xv[j] = ok ? (WITH_X ? x[el] : 1.0) : 0.0;
      ee[j] = at(el);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[0][j] = r0[ee[j]]; v[1][j] = r1[ee[j]]; v[2][j] = r2[ee[j]]; v[3][j] = r3[ee[j]]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) s[q] += v[q][j] * xv[j];
    }
  }
}

// sum + m[0] x[0] + m[cs] x[1] + ..., in that order, eight loads in flight (a thread walking its row one load, one fma at a
// time waits out a cache latency per entry)
__device__ __forceinline__ double row_dot(const double* m, long cs, const double* x, long cols, double sum) {
  long k = 0;
  for (; k + 8 <= cols; k += 8) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = m[(k + q) * cs];
#pragma unroll
    for (int q = 0; q < 8; ++q) sum += v[q] * x[k + q];
  }
  for (; k < cols; ++k) sum += m[k * cs] * x[k];
  return sum;
}

// xl: operand (c1 then c2 entries); yl: r1 + r2 doubles of LDS (only touched when a block is walked wave = row);
// store(row, sum) is called once per row by the thread that owns it
template <class Store>
__device__ inline void two_by_two_rows(const BlockRef& a11, const BlockRef& a12, const BlockRef& a21, const BlockRef& a22, long b, long r1,
                                       long c1, long r2, long c2, const double* xl, double* yl, Store store) {
  const long rows = r1 + r2;
  const bool w11 = wave_rows(a11, r1, c1), w12 = wave_rows(a12, r1, c2), w21 = wave_rows(a21, r2, c1), w22 = wave_rows(a22, r2, c2);
  const bool any_wave = w11 || w12 || w21 || w22;
  if (any_wave) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    for (long row0 = 4L * wave; row0 < rows; row0 += 4L * n_waves) {
      double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const long row = row0 + q;
        if (row < rows) {
          const bool top = row < r1;
          const long r = top ? row : row - r1;
          const BlockRef& left = top ? a11 : a21;
          const BlockRef& right = top ? a12 : a22;
          if (top ? w11 : w21) {
            const double* m = left.p + b * left.bs + r * left.rs;
            for (long k = lane; k < c1; k += 64) s[q] += m[k * left.cs] * xl[k];
          }
          if (top ? w12 : w22) {
            const double* m = right.p + b * right.bs + r * right.rs;
            for (long k = lane; k < c2; k += 64) s[q] += m[k * right.cs] * xl[c1 + k];
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s[q] += __shfl_xor(s[q], off, 64);
        if (lane == 0 && row0 + q < rows) yl[row0 + q] = s[q];
      }
    }
    __syncthreads();
  }
  for (long row = threadIdx.x; row < rows; row += blockDim.x) {
    const bool top = row < r1;
    const long r = top ? row : row - r1;
    const BlockRef& left = top ? a11 : a21;
    const BlockRef& right = top ? a12 : a22;
    double sum = any_wave ? yl[row] : 0.0;
    if (left.p && !(top ? w11 : w21)) sum = row_dot(left.p + b * left.bs + r * left.rs, left.cs, xl, c1, sum);
    if (right.p && !(top ? w12 : w22)) sum = row_dot(right.p + b * right.bs + r * right.rs, right.cs, xl + c1, c2, sum);
    store(row, sum);
  }
}

// threads of a workgroup that runs two_by_two_rows
inline unsigned two_by_two_threads(long rows, bool any_wave) {
  if (any_wave) return rows > 64 ? 1024u : 256u;
  return rows <= 64 ? 64u : (rows <= 128 ? 128u : 256u);
}

}  // namespace rmbi
