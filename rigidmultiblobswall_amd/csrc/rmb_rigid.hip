// rmb_rigid.hip -- per-body geometry and the per-body factors of the block-diagonal preconditioner as ONE launch each
// (gfx950, fp64).  Between two solves a time step moves the bodies and refactorises every body's own blob mobility; as
// torch operations that is ~60 launches (quaternions -> rotations, batched GEMMs, strided fills of K, batched Cholesky /
// triangular solves / 6 x 6 inverses) costing 0.4 + 0.8 ms whatever the size -- as much as the whole GMRES solve of a
// 64-body deck, and 7 % of the step at 2048 bodies (tools/experiments/exp_step_breakdown.py, profiles/r4_gmres_graph.txt).
//
//   rmb_rigid_configuration_device    blob coordinates r = R(q) ref + x, body-frame offsets and K = [I, -(rel x)] of every
//                                     body (body/body.py:64-115; quaternion convention of quaternion.py:41-51)
//   rmb_rigid_advance_device          x + v dt, quaternion(omega dt) * q for every body (quaternion_integrator_multi_bodies.py:86-91)
//   rmb_rigid_preconditioner_device   per body: M_b = L L^T, L^-1, M_b^-1, N = (K^T M_b^-1 K)^-1 and the four blocks of
//                                     [[M_b, -K], [-K^T, 0]]^-1 (multi_bodies.py:516-531 builds L and N once per step,
//                                     :548-560 applies them); one workgroup per body, everything in LDS.
#include "rmb_internal.h"
#include "block_rows.h"

#include <cmath>

namespace rmbi {
namespace {

struct ConfigArgs {
  long n_bodies, n_b;
  const double* ref;    // (n_bodies, n_b, 3)
  const double* loc;    // (n_bodies, 3)
  const double* quat;   // (n_bodies, 4) as (s, p1, p2, p3)
  double* r;            // (n_bodies n_b, 3)
  double* rel;          // (n_bodies, n_b, 3) or null
  double* K;            // (n_bodies, 3 n_b, 6) or null
};

__global__ __launch_bounds__(256) void rigid_config_kernel(const ConfigArgs a) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= a.n_bodies * a.n_b) return;
  const long b = id / a.n_b;
  const double s = a.quat[4 * b], p0 = a.quat[4 * b + 1], p1 = a.quat[4 * b + 2], p2 = a.quat[4 * b + 3];
  const double d = s * s - 0.5;
  const double x = a.ref[3 * id], y = a.ref[3 * id + 1], z = a.ref[3 * id + 2];
  // R = 2 [[p0 p0 + d, p0 p1 - s p2, p0 p2 + s p1], [p1 p0 + s p2, p1 p1 + d, p1 p2 - s p0], [p2 p0 - s p1, p2 p1 + s p0, p2 p2 + d]]
  const double rx = 2.0 * ((p0 * p0 + d) * x + (p0 * p1 - s * p2) * y + (p0 * p2 + s * p1) * z);
  const double ry = 2.0 * ((p1 * p0 + s * p2) * x + (p1 * p1 + d) * y + (p1 * p2 - s * p0) * z);
  const double rz = 2.0 * ((p2 * p0 - s * p1) * x + (p2 * p1 + s * p0) * y + (p2 * p2 + d) * z);
  a.r[3 * id] = rx + a.loc[3 * b]; a.r[3 * id + 1] = ry + a.loc[3 * b + 1]; a.r[3 * id + 2] = rz + a.loc[3 * b + 2];
  if (a.rel) { a.rel[3 * id] = rx; a.rel[3 * id + 1] = ry; a.rel[3 * id + 2] = rz; }
  if (a.K) {
    double* k = a.K + 18 * id;     // rows 3 l .. 3 l + 2 of body b: 18 consecutive doubles
    k[0] = 1.0; k[1] = 0.0; k[2] = 0.0; k[3] = 0.0;  k[4] = rz;   k[5] = -ry;
    k[6] = 0.0; k[7] = 1.0; k[8] = 0.0; k[9] = -rz;  k[10] = 0.0; k[11] = rx;
    k[12] = 0.0; k[13] = 0.0; k[14] = 1.0; k[15] = ry; k[16] = -rx; k[17] = 0.0;
  }
}

struct AdvanceArgs {
  long n_bodies;
  const double *loc, *quat, *U;
  const double* dt_body;     // per-body step or null
  double dt;
  double *loc_out, *quat_out;
};

// x + v dt and quaternion(omega dt) * q (quaternion_integrator_multi_bodies.py:86-91; quaternion.py:17-39)
__global__ __launch_bounds__(256) void rigid_advance_kernel(const AdvanceArgs a) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.n_bodies) return;
  const double dt = a.dt_body ? a.dt_body[b] : a.dt;
  const double* u = a.U + 6 * b;
  a.loc_out[3 * b] = a.loc[3 * b] + u[0] * dt;
  a.loc_out[3 * b + 1] = a.loc[3 * b + 1] + u[1] * dt;
  a.loc_out[3 * b + 2] = a.loc[3 * b + 2] + u[2] * dt;
  const double px = u[3] * dt, py = u[4] * dt, pz = u[5] * dt;
  const double nrm = sqrt(px * px + py * py + pz * pz);
  const double qs = cos(0.5 * nrm), f = nrm > 0.0 ? sin(0.5 * nrm) / nrm : 0.0;
  const double qx = f * px, qy = f * py, qz = f * pz;
  const double rs = a.quat[4 * b], rx = a.quat[4 * b + 1], ry = a.quat[4 * b + 2], rz = a.quat[4 * b + 3];
  a.quat_out[4 * b] = qs * rs - (qx * rx + qy * ry + qz * rz);
  a.quat_out[4 * b + 1] = qs * rx + rs * qx + (qy * rz - qz * ry);
  a.quat_out[4 * b + 2] = qs * ry + rs * qy + (qz * rx - qx * rz);
  a.quat_out[4 * b + 3] = qs * rz + rs * qz + (qx * ry - qy * rx);
}

constexpr int kPcMaxN = 48;       // 3 n_b: two n x n LDS matrices + the n x 6 panels fit the default 64 KB of dynamic LDS

struct PcArgs {
  long n_bodies;
  int n;                // 3 n_b
  const double* Mb;     // (n_bodies, n, n)
  const double* K;      // (n_bodies, n, 6)
  double *Lchol, *Linv, *Minv, *Nbody, *A11, *A12, *A21, *A22;
  int* info;
};

constexpr int kPcT = 256;          // threads per body: four wavefronts

// N = R^-1 for the 6 x 6 resistance R = K^T M_b^-1 K of one body: Gauss-Jordan with partial pivoting on [R | I], run by ONE
// thread; the result is symmetrised.  Returns false when R has no accurate inverse (residual of R N - I above 1e-8): a
// rank-deficient resistance (single blobs, collinear rods) must take the pseudo-inverse route of the caller
// (multi_bodies.py:531 uses pinv).
__device__ bool invert_resistance(const double* R, double* Nl) {
  // Every index below is a compile-time constant after unrolling (the pivot row is brought up by conditional exchanges with
  // every candidate row, not by a run-time row index), so the 6 x 12 work matrix lives in registers.  With a run-time row
  // index it sat in scratch memory, and ONE thread's ~1000 dependent accesses to it were most of the preconditioner kernel's
  // time (profiles/r5_pc_42_blob_shells.txt, last part).  Same operations in the same order as before.
  bool ok = true;
  double w[6][12];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = 0; j < 6; ++j) { w[i][j] = R[i * 6 + j]; w[i][6 + j] = (i == j) ? 1.0 : 0.0; }
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    int piv = c;
    double best = fabs(w[c][c]);
#pragma unroll
    for (int i = c + 1; i < 6; ++i) {
      const double cand = fabs(w[i][c]);
      if (cand > best) { best = cand; piv = i; }
    }
#pragma unroll
    for (int i = c + 1; i < 6; ++i) {
      const bool sel = piv == i;
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const double up = w[c][j], down = w[i][j];
        w[c][j] = sel ? down : up;
        w[i][j] = sel ? up : down;
      }
    }
    const double dgl = w[c][c];
    if (!(fabs(dgl) > 0.0)) {
      ok = false;
    } else {
      const double inv = 1.0 / dgl;
#pragma unroll
      for (int j = 0; j < 12; ++j) w[c][j] *= inv;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        if (i == c) continue;
        const double f = w[i][c];
#pragma unroll
        for (int j = 0; j < 12; ++j) w[i][j] -= f * w[c][j];
      }
    }
  }
  double worst = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) s += R[i * 6 + k] * w[k][6 + j];
      const double e = fabs(s - (i == j ? 1.0 : 0.0));
      if (!(e <= worst)) worst = e;       // NaN-propagating maximum
    }
  }
  if (!(worst < 1e-8)) ok = false;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = 0; j < 6; ++j) Nl[i * 6 + j] = 0.5 * (w[i][6 + j] + w[j][6 + i]);
  }
  return ok;
}

// One workgroup per body, everything in LDS.  The O(n^3) pieces (trailing updates of the Cholesky factorisation, the
// forward substitutions of L^-1, M_b^-1 = L^-T L^-1, A11) are spread over all 256 threads element by element; the
// sequential dimension (n = 3 n_b <= 48 pivots / rows) costs one or two workgroup barriers per step.
__global__ __launch_bounds__(kPcT) void rigid_pc_kernel(const PcArgs a) {
  extern __shared__ double lds[];
  const int n = a.n, t = threadIdx.x;
  const long b = blockIdx.x;
  double* A = lds;                 // M_b -> L (lower) -> M_b^-1
  double* B = A + n * n;           // L^-1
  double* Kl = B + n * n;          // n x 6
  double* MK = Kl + n * 6;         // M_b^-1 K
  double* A12l = MK + n * 6;       // -M_b^-1 K N
  double* R = A12l + n * 6;        // 6 x 6: K^T M_b^-1 K
  double* Nl = R + 36;             // 6 x 6: its inverse
  __shared__ int bad;
  if (t == 0) bad = 0;
  const double* M = a.Mb + b * (long)n * n;
  for (int idx = t; idx < n * n; idx += kPcT) {
    const int i = idx / n, j = idx - i * n;
    A[idx] = 0.5 * (M[idx] + M[j * n + i]);
  }
  for (int idx = t; idx < n * n; idx += kPcT) B[idx] = 0.0;
  for (int idx = t; idx < n * 6; idx += kPcT) Kl[idx] = a.K[b * (long)n * 6 + idx];
  __syncthreads();
  // ---- Cholesky, right-looking, lower: column k scaled by 1 / sqrt(pivot), then the trailing triangle updated ----
  for (int k = 0; k < n; ++k) {
    const double piv = A[k * n + k];           // every thread reads the pivot before anyone overwrites it
    __syncthreads();
    if (t == 0 && !(piv > 0.0)) bad = 1;
    const double root = sqrt(piv);
    if (t == k) A[k * n + k] = root;
    else if (t > k && t < n) A[t * n + k] /= root;
    __syncthreads();
    // elements (i, j), k < j <= i < n, of the trailing triangle: m = n - k - 1 rows, numbered row by row
    const int m = n - k - 1;
    for (int e = t; e < m * (m + 1) / 2; e += kPcT) {
      int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
      while ((i + 1) * (i + 2) / 2 <= e) ++i;
      while (i * (i + 1) / 2 > e) --i;
      const int j = e - i * (i + 1) / 2;
      const int gi = k + 1 + i, gj = k + 1 + j;
      A[gi * n + gj] -= A[gi * n + k] * A[gj * n + k];
    }
    __syncthreads();
  }
  for (int idx = t; idx < n * n; idx += kPcT) {
    const int i = idx / n, j = idx - i * n;
    a.Lchol[b * (long)n * n + idx] = j <= i ? A[idx] : 0.0;
  }
  // ---- L^-1 row by row: (L^-1)[i][c] = (delta_ic - sum_{k=c}^{i-1} L[i][k] (L^-1)[k][c]) / L[i][i]; for a given i the
  //      columns c <= i are independent.  Thread (c, part): four threads share one column's dot product. ----
  for (int i = 0; i < n; ++i) {
    const int c = t >> 2, part = t & 3;
    double s = 0.0;
    if (c < i) for (int k = c + part; k < i; k += 4) s += A[i * n + k] * B[k * n + c];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (part == 0 && c <= i) B[i * n + c] = ((c == i ? 1.0 : 0.0) - s) / A[i * n + i];
    __syncthreads();
  }
  for (int idx = t; idx < n * n; idx += kPcT) a.Linv[b * (long)n * n + idx] = B[idx];
  __syncthreads();     // L has been written out and L^-1 is complete: A is free
  // ---- M_b^-1 = L^-T L^-1, element (i, j) = sum_{k >= max(i, j)} (L^-1)[k][i] (L^-1)[k][j]: symmetric term by term ----
  for (int idx = t; idx < n * n; idx += kPcT) {
    const int i = idx / n, j = idx - i * n;
    double s = 0.0;
    for (int k = (i > j ? i : j); k < n; ++k) s += B[k * n + i] * B[k * n + j];
    A[idx] = s;
    a.Minv[b * (long)n * n + idx] = s;
  }
  __syncthreads();
  for (int idx = t; idx < n * 6; idx += kPcT) {
    const int i = idx / 6, c = idx - 6 * i;
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += A[i * n + j] * Kl[j * 6 + c];
    MK[idx] = s;
  }
  __syncthreads();
  if (t < 36) {
    const int p = t / 6, q = t - 6 * p;
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += Kl[i * 6 + p] * MK[i * 6 + q];
    R[t] = s;
  }
  __syncthreads();
  // ---- N = R^-1: Gauss-Jordan with partial pivoting on [R | I], one thread (216 multiply-adds) ----
  if (t == 0 && !invert_resistance(R, Nl)) bad = 1;
  __syncthreads();
  if (t < 36) {
    a.Nbody[b * 36 + t] = Nl[t];
    a.A22[b * 36 + t] = -Nl[t];
  }
  // ---- A12 = -M_b^-1 K N,  A21 = A12^T,  A11 = M_b^-1 + A12 (M_b^-1 K)^T ----
  for (int idx = t; idx < n * 6; idx += kPcT) {
    const int i = idx / 6, c = idx - 6 * i;
    double s = 0.0;
    for (int k = 0; k < 6; ++k) s += MK[i * 6 + k] * Nl[k * 6 + c];
    A12l[idx] = -s;
    a.A12[b * (long)n * 6 + idx] = -s;
    a.A21[b * (long)n * 6 + (long)c * n + i] = -s;
  }
  __syncthreads();
  for (int idx = t; idx < n * n; idx += kPcT) {
    const int i = idx / n, j = idx - i * n;
    double s = A[idx];
    for (int c = 0; c < 6; ++c) s += A12l[i * 6 + c] * MK[j * 6 + c];
    a.A11[b * (long)n * n + idx] = s;
  }
  if (t == 0 && bad) atomicOr(a.info, 1);
}


// ---- bodies of 17 .. 42 blobs (n = 3 n_b <= 128): ONE n x n matrix in LDS (126^2 doubles = 127 KB of the 160 KB) ----------
// The reference's own shells have 12 / 42 / 162 blobs (multi_bodies/Structures/shell_N_42_Rg_0_225.vertex; its timing
// harness multi_bodies/examples/Mobility_Prod_Timing uses them); two matrices as above do not fit for 42.  So everything
// runs IN PLACE on one matrix: Cholesky (lower) -> L written out -> L^-1 in place (column by column from the last one,
// LAPACK dtrti2's order; the column being replaced is first copied to a vector) -> L^-1 written out; M_b^-1 = L^-T L^-1 is
// never stored in LDS: K-panels go through Y = L^-1 K (M_b^-1 K = L^-T Y, R = Y^T Y) and the last pass forms every element
// of M_b^-1 from two columns of L^-1, stores it and adds the rank-6 term of A11.  1024 threads (16 waves) per body; the
// trailing update of the factorisation runs wave = row, lane = column (conflict-free LDS reads through a copy of the
// pivot column).
constexpr int kPcLargeT = 1024;
constexpr int kPcLargeMaxN = 128;

__global__ __launch_bounds__(kPcLargeT) void rigid_pc_large_kernel(const PcArgs a) {
  extern __shared__ double lds[];
  const int n = a.n, t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  constexpr int kWaves = kPcLargeT / 64;
  const long b = blockIdx.x;
  double* A = lds;                 // M_b -> L (lower) -> L^-1 (lower); the strict upper triangle is never read after the load
  double* col = A + n * n;         // n: the pivot column (factorisation) / the column being inverted
  double* Kl = col + n;            // n x 6
  double* Y = Kl + n * 6;          // L^-1 K
  double* MK = Y + n * 6;          // M_b^-1 K
  double* A12l = MK + n * 6;       // -M_b^-1 K N
  double* R = A12l + n * 6;        // 6 x 6
  double* Nl = R + 36;             // 6 x 6
  __shared__ int bad;
  if (t == 0) bad = 0;
  const double* M = a.Mb + b * (long)n * n;
  for (int idx = t; idx < n * n; idx += kPcLargeT) {
    const int i = idx / n, j = idx - i * n;
    A[idx] = 0.5 * (M[idx] + M[j * n + i]);
  }
  for (int idx = t; idx < n * 6; idx += kPcLargeT) Kl[idx] = a.K[b * (long)n * 6 + idx];
  __syncthreads();
  // ---- Cholesky, right-looking, lower: two barriers per pivot ----
  for (int k = 0; k < n; ++k) {
    __syncthreads();                               // the trailing update of the previous pivot is complete
    const double piv = A[k * n + k];               // nobody writes A[k][k] before the next barrier
    if (t == 0 && !(piv > 0.0)) bad = 1;
    const double root = sqrt(piv);
    if (t > k && t < n) { const double v = A[t * n + k] / root; A[t * n + k] = v; col[t] = v; }
    __syncthreads();
    if (t == k) A[k * n + k] = root;               // nobody reads A[k][k] any more: the update touches rows / columns > k
    for (int gi = k + 1 + wave; gi < n; gi += kWaves) {
      const double ci = col[gi];
      for (int gj = k + 1 + lane; gj <= gi; gj += 64) A[gi * n + gj] -= ci * col[gj];
    }
  }
  __syncthreads();
  for (int idx = t; idx < n * n; idx += kPcLargeT) {
    const int i = idx / n, j = idx - i * n;
    a.Lchol[b * (long)n * n + idx] = j <= i ? A[idx] : 0.0;
  }
  __syncthreads();
  // ---- L^-1 in place, columns from the last to the first: X = L^-1 satisfies X[j][j] = 1 / L[j][j] and
  //      X[j+1:, j] = -X[j][j] * (X[j+1:, j+1:] L[j+1:, j]); the trailing block X[j+1:, j+1:] is already in place ----
  for (int j = n - 1; j >= 0; --j) {
    const double dj = 1.0 / A[j * n + j];
    if (t > j && t < n) col[t] = A[t * n + j];
    __syncthreads();
    if (t == 0) A[j * n + j] = dj;
    {
      // row i = j + 1 + (t / 8): eight threads share its dot product over k = j + 1 .. i
      const int i = j + 1 + (t >> 3), part = t & 7;
      double s = 0.0;
      if (i < n) for (int k = j + 1 + part; k <= i; k += 8) s += A[i * n + k] * col[k];
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      if (i < n && part == 0) A[i * n + j] = -dj * s;
    }
    __syncthreads();
  }
  for (int idx = t; idx < n * n; idx += kPcLargeT) {
    const int i = idx / n, j = idx - i * n;
    a.Linv[b * (long)n * n + idx] = j <= i ? A[idx] : 0.0;
  }
  // ---- Y = L^-1 K,  M_b^-1 K = L^-T Y,  R = K^T M_b^-1 K = Y^T Y ----
  for (int idx = t; idx < n * 6; idx += kPcLargeT) {
    const int i = idx / 6, c = idx - 6 * i;
    double s = 0.0;
    for (int k = 0; k <= i; ++k) s += A[i * n + k] * Kl[k * 6 + c];
    Y[idx] = s;
  }
  __syncthreads();
  for (int idx = t; idx < n * 6; idx += kPcLargeT) {
    const int i = idx / 6, c = idx - 6 * i;
    double s = 0.0;
    for (int k = i; k < n; ++k) s += A[k * n + i] * Y[k * 6 + c];
    MK[idx] = s;
  }
  if (t < 36) {
    const int p = t / 6, q = t - 6 * p;
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += Y[i * 6 + p] * Y[i * 6 + q];
    R[t] = s;
  }
  __syncthreads();
  if (t == 0 && !invert_resistance(R, Nl)) bad = 1;
  __syncthreads();
  if (t < 36) {
    a.Nbody[b * 36 + t] = Nl[t];
    a.A22[b * 36 + t] = -Nl[t];
  }
  for (int idx = t; idx < n * 6; idx += kPcLargeT) {
    const int i = idx / 6, c = idx - 6 * i;
    double s = 0.0;
    for (int k = 0; k < 6; ++k) s += MK[i * 6 + k] * Nl[k * 6 + c];
    A12l[idx] = -s;
    a.A12[b * (long)n * 6 + idx] = -s;
    a.A21[b * (long)n * 6 + (long)c * n + i] = -s;
  }
  __syncthreads();
  // ---- M_b^-1[i][j] = sum_{k >= max(i, j)} (L^-1)[k][i] (L^-1)[k][j] (symmetric term by term),  A11 = M_b^-1 + A12 (M_b^-1 K)^T ----
  for (int idx = t; idx < n * n; idx += kPcLargeT) {
    const int i = idx / n, j = idx - i * n;
    double s = 0.0;
    for (int k = (i > j ? i : j); k < n; ++k) s += A[k * n + i] * A[k * n + j];
    a.Minv[b * (long)n * n + idx] = s;
    double s11 = s;
    for (int c = 0; c < 6; ++c) s11 += A12l[i * 6 + c] * MK[j * 6 + c];
    a.A11[b * (long)n * n + idx] = s11;
  }
  if (t == 0 && bad) atomicOr(a.info, 1);
}


// ---- the saddle-point operator's finishing launch: workgroup = body ------------------------------------------------
struct OpFinArgs {
  const double4* pos;
  const double* x;        // [lambda (3N); U (6 n_bodies)]
  const double* K;        // (n_bodies, 3 n_b, 6)
  double* acc;            // [3][n_pad] raw sums of the symmetric tt sweep; re-zeroed here
  double* out;            // [M lambda - K U (3N); -K^T lambda (6 n_bodies)]
  long n, n_pad, n_bodies;
  int n_b;
  double prefactor;
  rmb::PairConsts k;
  // optional (part != nullptr): the first Gram-Schmidt pass's partial dots of the body's slices of `out` with the basis
  // rows V[0 .. rows): part[r * n_bodies + body]
  const double* V;
  long ldv, rows;
  double* part;
};

// Thread l < n_b finishes blob i = body n_b + l exactly as sym_finalize_kernel does (self term of the B-damped lambda,
// scaling by b_i / (8 pi eta)), subtracts row i of K U, and contributes its three rows of K^T lambda to a fixed-order
// reduction over the body.
template <bool WALL>
__global__ __launch_bounds__(256) void rigid_operator_finish_kernel(const OpFinArgs a) {
  extern __shared__ double wl[];          // 3 n_b + 6: the body's slices of the result (only with a.part)
  __shared__ double part[4][6];
  const long body = blockIdx.x;
  const int l = threadIdx.x, lane = l & 63, wave = l >> 6;
  const long n3 = 3 * a.n;
  double kt[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (l < a.n_b) {
    const long i = body * a.n_b + l;
    rmb::Vec3 acc = {a.acc[i], a.acc[a.n_pad + i], a.acc[2 * a.n_pad + i]};
    a.acc[i] = 0.0; a.acc[a.n_pad + i] = 0.0; a.acc[2 * a.n_pad + i] = 0.0;     // ready for the next product
    const double4 p = a.pos[i];
    const double b = p.w;
    const double lx = a.x[3 * i], ly = a.x[3 * i + 1], lz = a.x[3 * i + 2];
    rmb::self_term<rmb::KIND_TT, WALL>(a.k, p.z, lx * b, ly * b, lz * b, 0, 0, 0, acc);
    const double sc = a.prefactor * b;
    const double* U = a.x + n3 + 6 * body;
    const double* Kr = a.K + (body * 3 * a.n_b + 3 * l) * 6;
    double u[3] = {acc.x * sc, acc.y * sc, acc.z * sc};
    const double lam[3] = {lx, ly, lz};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        s += Kr[6 * c + q] * U[q];
        kt[q] += Kr[6 * c + q] * lam[c];
      }
      a.out[3 * i + c] = u[c] - s;
      if (a.part) wl[3 * l + c] = u[c] - s;
    }
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    double v = kt[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) part[wave][q] = v;
  }
  __syncthreads();
  if (l < 6) {
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += part[w][l];
    a.out[n3 + 6 * body + l] = -s;
    if (a.part) wl[3 * a.n_b + l] = -s;
  }
  if (a.part) {
    // the body's share of V[r] . out for every basis row: wave q takes rows 4q .. 4q + 3, ... (sixteen loads in flight)
    __syncthreads();
    const long nn = 3L * a.n_b, len = nn + 6, top = body * nn, bot = n3 + 6 * body - nn;
    const int n_waves = (int)(blockDim.x >> 6);
    for (long r0 = 4L * wave; r0 < a.rows; r0 += 4L * n_waves) {
      const double* row[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) row[q] = a.V + (r0 + q < a.rows ? r0 + q : r0) * a.ldv;
      double s[4];
      four_row_sums<true>(row[0], row[1], row[2], row[3], wl, len, s, [&](long k) { return k < nn ? top + k : bot + k; });
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double t = wave_sum(s[q]);
        if (lane == 0 && r0 + q < a.rows) a.part[(r0 + q) * a.n_bodies + body] = t;
      }
    }
  }
}

// The Lanczos step's finishing launch: workgroup = body.  Thread l < n_b finishes blob i = body n_b + l as sym_finalize_kernel
// does (self term of the B-damped input, scaling by b_i / (8 pi eta)) into LDS, then the workgroup multiplies the body's 3 n_b
// entries by L_b^-1:  d = P^T (M pv),  P = blockdiag(L_b^-T) -- the sweep's finalize and the block product in one launch.
struct LanFinArgs {
  const double4* pos;
  const double* x;        // the sweep's input (3N)
  double* acc;            // [3][n_pad] raw sums of the symmetric tt sweep; re-zeroed here
  BlockRef linv;          // (n_bodies, 3 n_b, 3 n_b)
  double* out;            // 3N
  long n, n_pad;
  int n_b;
  double prefactor;
  rmb::PairConsts k;
  // optional (part != nullptr): the first Gram-Schmidt pass's partial dots of the body's 3 n_b entries of `out` with the
  // basis rows V[0 .. rows): part[r * n_bodies + body]
  const double* V;
  long ldv, rows, n_bodies;
  double* part;
};

template <bool WALL>
__global__ __launch_bounds__(1024) void lanczos_finish_kernel(const LanFinArgs a) {
  extern __shared__ double xl[];          // 3 n_b finished entries, 3 n_b row sums (two_by_two_rows), 3 n_b results (with a.part)
  const long body = blockIdx.x;
  const int l = threadIdx.x;
  if (l < a.n_b) {
    const long i = body * a.n_b + l;
    rmb::Vec3 acc = {a.acc[i], a.acc[a.n_pad + i], a.acc[2 * a.n_pad + i]};
    a.acc[i] = 0.0; a.acc[a.n_pad + i] = 0.0; a.acc[2 * a.n_pad + i] = 0.0;     // ready for the next product
    const double4 p = a.pos[i];
    const double b = p.w;
    rmb::self_term<rmb::KIND_TT, WALL>(a.k, p.z, a.x[3 * i] * b, a.x[3 * i + 1] * b, a.x[3 * i + 2] * b, 0, 0, 0, acc);
    const double sc = a.prefactor * b;
    xl[3 * l] = acc.x * sc; xl[3 * l + 1] = acc.y * sc; xl[3 * l + 2] = acc.z * sc;
  }
  __syncthreads();
  const long nn = 3L * a.n_b;
  const BlockRef none{nullptr, 0, 0, 0};
  double* res = xl + 2 * nn;
  two_by_two_rows(a.linv, none, none, none, body, nn, nn, 0, 0, xl, xl + nn, [&](long row, double sum) {
    a.out[body * nn + row] = sum;
    if (a.part) res[row] = sum;
  });
  if (a.part) {
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = (int)(blockDim.x >> 6);
    const long top = body * nn;
    for (long r0 = 4L * wave; r0 < a.rows; r0 += 4L * n_waves) {
      const double* row[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) row[q] = a.V + (r0 + q < a.rows ? r0 + q : r0) * a.ldv + top;
      double s[4];
      four_row_sums<true>(row[0], row[1], row[2], row[3], res, nn, s, [](long e) { return e; });
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double t = wave_sum(s[q]);
        if (lane == 0 && r0 + q < a.rows) a.part[(r0 + q) * a.n_bodies + body] = t;
      }
    }
  }
}

// The plain product's finishing launch with the first Gram-Schmidt pass's dots (rmb_lanczos_device, M_tt): workgroup = tile of
// 64 blobs.  Thread l < 64 finishes blob i = 64 tile + l as sym_finalize_kernel does, the 192 results go to LDS and from there
// to `out` with consecutive addresses, then four waves take the tile's share of V[r] . out for every basis row.
struct PlainFinArgs {
  const double4* pos;
  const double* x;        // the sweep's input (3N)
  double* acc;            // [3][n_pad] raw sums of the symmetric tt sweep; re-zeroed here
  double* out;            // 3N
  long n, n_pad, n_tiles;
  double prefactor;
  rmb::PairConsts k;
  const double* V;
  long ldv, rows;
  double* part;           // part[r * n_tiles + tile]
};

template <bool WALL>
__global__ __launch_bounds__(256) void plain_finish_kernel(const PlainFinArgs a) {
  __shared__ double res[192];
  const long tile = blockIdx.x, first = 64 * tile;
  const long cnt = (a.n - first) < 64 ? (a.n - first) : 64;
  const int l = threadIdx.x, lane = l & 63, wave = l >> 6;
  if (l < cnt) {
    const long i = first + l;
    rmb::Vec3 acc = {a.acc[i], a.acc[a.n_pad + i], a.acc[2 * a.n_pad + i]};
    a.acc[i] = 0.0; a.acc[a.n_pad + i] = 0.0; a.acc[2 * a.n_pad + i] = 0.0;     // ready for the next product
    const double4 p = a.pos[i];
    const double b = p.w;
    rmb::self_term<rmb::KIND_TT, WALL>(a.k, p.z, a.x[3 * i] * b, a.x[3 * i + 1] * b, a.x[3 * i + 2] * b, 0, 0, 0, acc);
    const double sc = a.prefactor * b;
    res[3 * l] = acc.x * sc; res[3 * l + 1] = acc.y * sc; res[3 * l + 2] = acc.z * sc;
  }
  __syncthreads();
  const long len = 3 * cnt, base = 3 * first;
  for (long e = l; e < len; e += blockDim.x) a.out[base + e] = res[e];
  for (long r0 = 4L * wave; r0 < a.rows; r0 += 16) {
    const double* row[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) row[q] = a.V + (r0 + q < a.rows ? r0 + q : r0) * a.ldv + base;
    double s[4];
    four_row_sums<true>(row[0], row[1], row[2], row[3], res, len, s, [](long e) { return e; });
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double t = wave_sum(s[q]);
      if (lane == 0 && r0 + q < a.rows) a.part[(r0 + q) * a.n_tiles + tile] = t;
    }
  }
}

}  // namespace

// out = M_tt v on the resident configuration, with the partial dots of `out` against V[0 .. rows) left for the Gram-Schmidt
// step (krylov_body_partials' buffer, one partial per tile of 64 blobs): *tiles_done = that count, or 0 when the plain path ran
int plain_tt_with_dots(rmb_ctx* c, const double* v_dev, double eta, double* out_dev, const double* V_dev, long ldv, long rows, long* tiles_done) {
  *tiles_done = 0;
  const long n = c->n, n_tiles = (n + 63) / 64;
  const bool periodic = c->L[0] > 0 || c->L[1] > 0 || c->L[2] > 0;
  if (!(c->opt_gmres_fuse_dots && sym_applies(c) && !periodic && c->opt_deterministic == 0 && c->opt_precision == 64 && c->tgt_begin == 0 &&
        c->tgt_end == n && n_tiles <= kKrBodyPartialsMax))
    return matvec_device_impl(c, rmb::KIND_TT, 0, v_dev, nullptr, eta, out_dev);
  PlainFinArgs a;
  if (int rc = krylov_body_partials(c, 3 * n, &a.part)) return rc;
  if (int rc = sym_device(c, rmb::KIND_TT, v_dev, eta, out_dev, 0, 1, false, true)) return rc;
  a.pos = (const double4*)c->pos.p; a.x = v_dev; a.acc = (double*)c->symbuf.p; a.out = out_dev;
  a.n = n; a.n_pad = 64 * n_tiles; a.n_tiles = n_tiles;
  a.prefactor = 1.0 / (8.0 * M_PI * eta);
  a.k = make_pair_consts(c->a);
  a.V = V_dev; a.ldv = ldv; a.rows = rows;
  if (c->wall) hipLaunchKernelGGL(plain_finish_kernel<true>, dim3((unsigned)n_tiles), dim3(256), 0, c->stream, a);
  else         hipLaunchKernelGGL(plain_finish_kernel<false>, dim3((unsigned)n_tiles), dim3(256), 0, c->stream, a);
  RMB_HIP(hipGetLastError());
  *tiles_done = n_tiles;
  return 0;
}

int lanczos_step_impl(rmb_ctx* c, long n_bodies, long n_b, const double* Linv_dev, double* V_dev, long ldv, long i, double eta, double* pv_dev,
                      double* mw_dev, double* d_dev, double* col_dev, double* col_mapped_dev, bool pv_ready, bool fuse_next) {
  if (int rc = check_ready(c)) return rc;
  if (n_bodies < 1 || n_b < 1 || i < 0) return fail(RMB_ERR_ARG, "rmb_rigid_lanczos_step_device: bad n_bodies / n_b / i");
  if (n_bodies * n_b != c->n) return fail(RMB_ERR_STATE, "rmb_rigid_lanczos_step_device: the resident configuration does not hold n_bodies x n_b blobs");
  if (!Linv_dev || !V_dev || !pv_dev || !mw_dev || !d_dev || !col_dev) return fail(RMB_ERR_ARG, "null pointer");
  if (!(eta > 0.0)) return fail(RMB_ERR_ARG, "eta must be positive");
  const long nn = 3 * n_b, n = c->n, n3 = 3 * n;
  if (ldv < n3) return fail(RMB_ERR_ARG, "rmb_rigid_lanczos_step_device: ldv < 3 N");
  RMB_HIP(hipSetDevice(c->device));
  const double* v = V_dev + i * ldv;
  // pv = P v with P = blockdiag(L_b^-T): the transposed block is the same memory with the two strides exchanged
  const rmb_block lt{Linv_dev, nn * nn, 1, nn}, l{Linv_dev, nn * nn, nn, 1};
  if (!pv_ready)
    if (int rc = rmb_block_apply_device(c, n_bodies, nn, nn, 0, 0, &lt, nullptr, nullptr, nullptr, v, nullptr, 1.0, 0.0, pv_dev, 0.0, nullptr)) return rc;
  // d = P^T (M pv)
  long dots_bodies = 0;
  const bool periodic = c->L[0] > 0 || c->L[1] > 0 || c->L[2] > 0;
  if (c->opt_lanczos_fuse_finish && sym_applies(c) && !periodic && c->opt_deterministic == 0 && c->opt_precision == 64 && c->tgt_begin == 0 && c->tgt_end == n &&
      n_b <= 256) {
    // the pair sweep with the raw sums left in the accumulators, then ONE finishing launch (finalize + L_b^-1)
    if (int rc = sym_device(c, rmb::KIND_TT, pv_dev, eta, mw_dev, 0, 1, false, true)) return rc;
    LanFinArgs a;
    a.pos = (const double4*)c->pos.p; a.x = pv_dev; a.acc = (double*)c->symbuf.p; a.linv = BlockRef{Linv_dev, nn * nn, nn, 1}; a.out = d_dev;
    a.n = n; a.n_pad = 64 * ((n + 63) / 64); a.n_b = (int)n_b;
    a.prefactor = 1.0 / (8.0 * M_PI * eta);
    a.k = make_pair_consts(c->a);
    unsigned threads = two_by_two_threads(nn, wave_rows(a.linv, nn, nn));
    if (threads < (unsigned)(64 * ((n_b + 63) / 64))) threads = (unsigned)(64 * ((n_b + 63) / 64));
    // small decks: this launch also takes the first Gram-Schmidt pass's dots (one partial per body and basis row)
    a.V = nullptr; a.ldv = 0; a.rows = 0; a.n_bodies = n_bodies; a.part = nullptr;
    if (c->opt_gmres_fuse_dots && n_bodies <= kKrBodyPartialsMax) {
      if (int rc = krylov_body_partials(c, n3, &a.part)) return rc;
      a.V = V_dev; a.ldv = ldv; a.rows = i + 1;
      if (threads < 256) threads = 256;
      dots_bodies = n_bodies;
    }
    const size_t lds = (size_t)(3 * nn) * sizeof(double);
    if (c->wall) hipLaunchKernelGGL(lanczos_finish_kernel<true>, dim3((unsigned)n_bodies), dim3(threads), lds, c->stream, a);
    else         hipLaunchKernelGGL(lanczos_finish_kernel<false>, dim3((unsigned)n_bodies), dim3(threads), lds, c->stream, a);
    RMB_HIP(hipGetLastError());
  } else {
    if (int rc = matvec_device_impl(c, rmb::KIND_TT, 0, pv_dev, nullptr, eta, mw_dev)) return rc;
    if (int rc = rmb_block_apply_device(c, n_bodies, nn, nn, 0, 0, &l, nullptr, nullptr, nullptr, mw_dev, nullptr, 1.0, 0.0, d_dev, 0.0, nullptr)) return rc;
  }
  // full re-orthogonalisation against v_0 .. v_i: col[i] = h_ii, col[i + 1] = h_{i+1,i} (the other coefficients vanish to
  // rounding with an orthonormal basis), v_{i+1} = d / |d| (and, fused, pv = P v_{i+1})
  const BlockRef none{nullptr, 0, 0, 0};
  const PcBlocks pc{n_bodies, nn, 0, {Linv_dev, nn * nn, 1, nn}, none, none, none, pv_dev};
  return krylov_orthogonalize_impl(c, n3, i + 1, V_dev, ldv, d_dev, col_dev, V_dev + (i + 1) * ldv, col_mapped_dev, fuse_next ? &pc : nullptr,
                                   dots_bodies, true);
}

int arnoldi_step_impl(rmb_ctx* c, long n_bodies, long n_b, const double* A11_dev, const double* A12_dev, const double* A21_dev,
                      const double* A22_dev, const double* K_dev, double* V_dev, long ldv, long j, double eta, double* z_dev, double* w_dev,
                      double* col_dev, double* col_mapped_dev, bool z_ready, bool fuse_pc) {
  if (int rc = check_ready(c)) return rc;
  if (n_bodies < 1 || n_b < 1 || j < 0) return fail(RMB_ERR_ARG, "rmb_rigid_arnoldi_step_device: bad n_bodies / n_b / j");
  if (!A11_dev || !A12_dev || !A21_dev || !A22_dev || !K_dev || !V_dev || !z_dev || !w_dev || !col_dev) return fail(RMB_ERR_ARG, "null pointer");
  const long nn = 3 * n_b, n3 = 3 * n_bodies * n_b, n = n3 + 6 * n_bodies;
  if (ldv < n) return fail(RMB_ERR_ARG, "rmb_rigid_arnoldi_step_device: ldv < 3 N + 6 n_bodies");
  if (!z_ready) {
    // z = P^-1 v_j: the four blocks of every body's [[M_b, -K], [-K^T, 0]]^-1 in one launch
    const double* v = V_dev + j * ldv;
    const rmb_block b11{A11_dev, nn * nn, nn, 1}, b12{A12_dev, nn * 6, 6, 1}, b21{A21_dev, 6 * nn, nn, 1}, b22{A22_dev, 36, 6, 1};
    if (int rc = rmb_block_apply_device(c, n_bodies, nn, nn, 6, 6, &b11, &b12, &b21, &b22, v, v + n3, 1.0, 0.0, z_dev, 0.0, z_dev + n3)) return rc;
  }
  // w = A z: pair sweep + one finishing launch, which (small decks, inside the native GMRES) also takes the first pass's dots
  DotsFuse df{V_dev, ldv, j + 1, nullptr};
  bool dots_done = false;
  if (fuse_pc && c->opt_gmres_fuse_dots && n_bodies <= kKrBodyPartialsMax)
    if (int rc = krylov_body_partials(c, n, &df.part)) return rc;
  if (int rc = rigid_operator_impl(c, n_bodies, n_b, K_dev, z_dev, eta, w_dev, df.part ? &df : nullptr, &dots_done)) return rc;
  // two Gram-Schmidt passes against v_0 .. v_j, Hessenberg column, |w|, v_{j+1} (and, fused, z = P^-1 v_{j+1})
  const PcBlocks pc{n_bodies, nn, 6, {A11_dev, nn * nn, nn, 1}, {A12_dev, nn * 6, 6, 1}, {A21_dev, 6 * nn, nn, 1}, {A22_dev, 36, 6, 1}, z_dev};
  return krylov_orthogonalize_impl(c, n, j + 1, V_dev, ldv, w_dev, col_dev, V_dev + (j + 1) * ldv, col_mapped_dev, fuse_pc ? &pc : nullptr,
                                   dots_done ? n_bodies : 0, true);
}

}  // namespace rmbi

using namespace rmbi;

extern "C" {

int rmb_rigid_configuration_device(rmb_ctx* c, long n_bodies, long n_b, const double* ref_dev, const double* loc_dev,
                                   const double* quat_dev, double* r_dev, double* rel_dev, double* K_dev) {
  if (!c) return fail(RMB_ERR_ARG, "null context");
  if (n_bodies < 0 || n_b < 1) return fail(RMB_ERR_ARG, "rmb_rigid_configuration_device: bad n_bodies / blobs per body");
  if (n_bodies == 0) return 0;
  if (!ref_dev || !loc_dev || !quat_dev || !r_dev) return fail(RMB_ERR_ARG, "null pointer");
  RMB_HIP(hipSetDevice(c->device));
  ConfigArgs a{n_bodies, n_b, ref_dev, loc_dev, quat_dev, r_dev, rel_dev, K_dev};
  const long total = n_bodies * n_b;
  hipLaunchKernelGGL(rigid_config_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, a);
  RMB_HIP(hipGetLastError());
  return 0;
}

int rmb_rigid_advance_device(rmb_ctx* c, long n_bodies, const double* loc_dev, const double* quat_dev, const double* U_dev, double dt,
                             const double* dt_body_dev, double* loc_out_dev, double* quat_out_dev) {
  if (!c) return fail(RMB_ERR_ARG, "null context");
  if (n_bodies < 0) return fail(RMB_ERR_ARG, "rmb_rigid_advance_device: negative n_bodies");
  if (n_bodies == 0) return 0;
  if (!loc_dev || !quat_dev || !U_dev || !loc_out_dev || !quat_out_dev) return fail(RMB_ERR_ARG, "null pointer");
  RMB_HIP(hipSetDevice(c->device));
  AdvanceArgs a{n_bodies, loc_dev, quat_dev, U_dev, dt_body_dev, dt, loc_out_dev, quat_out_dev};
  hipLaunchKernelGGL(rigid_advance_kernel, dim3((unsigned)((n_bodies + 255) / 256)), dim3(256), 0, c->stream, a);
  RMB_HIP(hipGetLastError());
  return 0;
}

int rmb_rigid_preconditioner_device(rmb_ctx* c, long n_bodies, long n_b, const double* Mb_dev, const double* K_dev, double* Lchol_dev,
                                    double* Linv_dev, double* Minv_dev, double* Nbody_dev, double* A11_dev, double* A12_dev,
                                    double* A21_dev, double* A22_dev, int* info_dev) {
  if (!c) return fail(RMB_ERR_ARG, "null context");
  if (n_bodies < 0 || n_b < 1) return fail(RMB_ERR_ARG, "rmb_rigid_preconditioner_device: bad n_bodies / blobs per body");
  if (3 * n_b > kPcLargeMaxN)
    return fail(RMB_ERR_ARG, "rmb_rigid_preconditioner_device: at most 42 blobs per body (the factors of one body are kept in LDS)");
  if (n_bodies == 0) return 0;
  if (!Mb_dev || !K_dev || !Lchol_dev || !Linv_dev || !Minv_dev || !Nbody_dev || !A11_dev || !A12_dev || !A21_dev || !A22_dev || !info_dev)
    return fail(RMB_ERR_ARG, "null pointer");
  RMB_HIP(hipSetDevice(c->device));
  PcArgs a;
  a.n_bodies = n_bodies; a.n = (int)(3 * n_b);
  a.Mb = Mb_dev; a.K = K_dev; a.Lchol = Lchol_dev; a.Linv = Linv_dev; a.Minv = Minv_dev; a.Nbody = Nbody_dev;
  a.A11 = A11_dev; a.A12 = A12_dev; a.A21 = A21_dev; a.A22 = A22_dev; a.info = info_dev;
  RMB_HIP(hipMemsetAsync(info_dev, 0, sizeof(int), c->stream));
  if (a.n <= kPcMaxN) {
    const size_t lds = ((size_t)2 * a.n * a.n + (size_t)18 * a.n + 72) * sizeof(double);
    hipLaunchKernelGGL(rigid_pc_kernel, dim3((unsigned)n_bodies), dim3(kPcT), lds, c->stream, a);
  } else {
    // one matrix + the pivot column + four n x 6 panels + two 6 x 6: 146.7 KB at n = 126 (one workgroup per CU)
    const size_t lds = ((size_t)a.n * a.n + (size_t)25 * a.n + 72) * sizeof(double);
    if (lds + 1024 > c->lds_per_cu) return fail(RMB_ERR_STATE, "rmb_rigid_preconditioner_device: the device's LDS is too small for this body size");
    RMB_HIP(hipFuncSetAttribute((const void*)rigid_pc_large_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(rigid_pc_large_kernel, dim3((unsigned)n_bodies), dim3(kPcLargeT), lds, c->stream, a);
  }
  RMB_HIP(hipGetLastError());
  return 0;
}

int rmb_rigid_operator_device(rmb_ctx* c, long n_bodies, long n_b, const double* K_dev, const double* x_dev, double eta,
                              double* out_dev) {
  return rigid_operator_impl(c, n_bodies, n_b, K_dev, x_dev, eta, out_dev, nullptr, nullptr);
}

}  // extern "C"

namespace rmbi {
int rigid_operator_impl(rmb_ctx* c, long n_bodies, long n_b, const double* K_dev, const double* x_dev, double eta, double* out_dev,
                        const DotsFuse* dots, bool* dots_done) {
  if (dots_done) *dots_done = false;
  if (int rc = check_ready(c)) return rc;
  if (n_bodies < 1 || n_b < 1 || n_b > 256) return fail(RMB_ERR_ARG, "rmb_rigid_operator_device: need n_bodies >= 1 and 1 <= n_b <= 256");
  if (n_bodies * n_b != c->n) return fail(RMB_ERR_STATE, "rmb_rigid_operator_device: the resident configuration does not hold n_bodies x n_b blobs");
  if (c->tgt_begin != 0 || c->tgt_end != c->n) return fail(RMB_ERR_STATE, "rmb_rigid_operator_device: needs the full target range");
  if (!K_dev || !x_dev || !out_dev) return fail(RMB_ERR_ARG, "null pointer");
  if (!(eta > 0.0)) return fail(RMB_ERR_ARG, "eta must be positive");
  RMB_HIP(hipSetDevice(c->device));
  const long n = c->n, n3 = 3 * n;
  const bool periodic = c->L[0] > 0 || c->L[1] > 0 || c->L[2] > 0;
  if (sym_applies(c) && !periodic && c->opt_deterministic == 0 && c->opt_precision == 64) {
    // pair sweep with the raw sums left in the accumulators, then ONE finishing launch
    if (int rc = sym_device(c, rmb::KIND_TT, x_dev, eta, out_dev, 0, 1, false, true)) return rc;
    OpFinArgs a;
    a.pos = (const double4*)c->pos.p; a.x = x_dev; a.K = K_dev; a.acc = (double*)c->symbuf.p; a.out = out_dev;
    a.n = n; a.n_pad = 64 * ((n + 63) / 64); a.n_bodies = n_bodies; a.n_b = (int)n_b;
    a.prefactor = 1.0 / (8.0 * M_PI * eta);
    a.k = make_pair_consts(c->a);
    a.V = nullptr; a.ldv = 0; a.rows = 0; a.part = nullptr;
    unsigned threads = (unsigned)(64 * ((n_b + 63) / 64));
    size_t lds = 0;
    if (dots && dots->part && dots->rows > 0) {
      a.V = dots->V; a.ldv = dots->ldv; a.rows = dots->rows; a.part = dots->part;
      threads = 256;                                    // four waves share the basis rows
      lds = (size_t)(3 * n_b + 6) * sizeof(double);
    }
    if (c->wall) hipLaunchKernelGGL(rigid_operator_finish_kernel<true>, dim3((unsigned)n_bodies), dim3(threads), lds, c->stream, a);
    else         hipLaunchKernelGGL(rigid_operator_finish_kernel<false>, dim3((unsigned)n_bodies), dim3(threads), lds, c->stream, a);
    RMB_HIP(hipGetLastError());
    if (a.part && dots_done) *dots_done = true;
    return 0;
  }
  // any other mode (one-sided sweep below 128 blobs, periodic images, deterministic / single-precision options): the
  // product, then top -= K U and bottom = -K^T lambda through the block kernel -- the same result in three launches
  if (int rc = matvec_device_impl(c, rmb::KIND_TT, 0, x_dev, nullptr, eta, out_dev)) return rc;
  rmb_block kb{K_dev, 3 * n_b * 6, 6, 1};            // K as (batch, row, col)
  rmb_block kt{K_dev, 3 * n_b * 6, 1, 6};            // K^T: strides exchanged
  return rmb_block_apply_device(c, n_bodies, 3 * n_b, 3 * n_b, 6, 6, nullptr, &kb, &kt, nullptr, x_dev, x_dev + n3, -1.0, 1.0, out_dev, 0.0,
                                out_dev + n3);
}
}  // namespace rmbi

extern "C" {

int rmb_rigid_arnoldi_step_device(rmb_ctx* c, long n_bodies, long n_b, const double* A11_dev, const double* A12_dev,
                                  const double* A21_dev, const double* A22_dev, const double* K_dev, double* V_dev, long ldv, long j,
                                  double eta, double* z_dev, double* w_dev, double* col_dev, double* col_mapped_dev) {
  return arnoldi_step_impl(c, n_bodies, n_b, A11_dev, A12_dev, A21_dev, A22_dev, K_dev, V_dev, ldv, j, eta, z_dev, w_dev, col_dev,
                           col_mapped_dev, false, false);
}

int rmb_rigid_lanczos_step_device(rmb_ctx* c, long n_bodies, long n_b, const double* Linv_dev, double* V_dev, long ldv, long i,
                                  double eta, double* y_dev, double* w_dev, double* col_dev, double* col_mapped_dev) {
  // y = P v_i, w = M y (or the raw sums of the sweep), then y <- P^T M y in y's place and its orthogonalisation
  return lanczos_step_impl(c, n_bodies, n_b, Linv_dev, V_dev, ldv, i, eta, y_dev, w_dev, y_dev, col_dev, col_mapped_dev, false, false);
}

}  // extern "C"
