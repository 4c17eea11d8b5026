// rmb_gmres.hip -- the whole right-preconditioned GMRES(restart) of the rigid-body saddle-point problem as ONE library call
// (gfx950, fp64): rmb_rigid_gmres_device.
//
// The reference solves [[M, -K], [-K^T, 0]] [lambda; U] = [slip; -F] with scipy's compiled GMRES around Python callbacks
// (quaternion_integrator/quaternion_integrator_multi_bodies.py:1441-1547 -> general_application_utils.py:608-627, restart
// 60, right preconditioning with the per-body block inverse of multi_bodies.py:548-560).  On the decks it is mostly run on
// -- tens to hundreds of bodies -- an iteration is seven launches of a few microseconds, so what an iteration costs is
// the HOST: with the loop in Python (rigid.py: _gmres_steps, one rmb_rigid_arnoldi_step_device call per iteration) 49 us
// per iteration against 32-36 us of GPU time (profiles/r5_gmres_step.txt).  Here the loop itself is native: per iteration
// one rmb_rigid_arnoldi_step_device, one event record, and -- ONE ITERATION LATE, while the device runs the next step --
// the Givens rotations and the convergence test on the column the Gram-Schmidt kernel stored into mapped host memory.
// Same algorithm, same stopping rule, same lag policy as rigid.py's loop (which stays for everything this entry does not
// cover: several body shapes, prescribed bodies, an initial guess, multi-rank facades):
//   * two passes of classical Gram-Schmidt per step (rmb_krylov_orthogonalize2_device);
//   * stop when |g_{j+1}| <= tol |b| (scipy's `tol`, atol = 0), or on an exact breakdown, or after maxiter inner iterations;
//   * the host reads column j after step j + 1 has been enqueued, unless the last observed reduction rate says column j may
//     already meet the tolerance (then it waits first): at most one discarded step per solve;
//   * at a restart the TRUE residual b - A P^-1 y is formed.
#include "rmb_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace rmbi {
namespace {

constexpr int kVecT = 256;

__global__ __launch_bounds__(kVecT) void vec_scale_kernel(double* out, const double* in, double alpha, long n) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) out[e] = in[e] * alpha;
}

__global__ __launch_bounds__(kVecT) void vec_sub_kernel(double* out, const double* a, const double* b, long n) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) out[e] = a[e] - b[e];
}

__global__ __launch_bounds__(kVecT) void vec_zero_kernel(double* out, long n) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) out[e] = 0.0;
}

// y[e] += sum_{i < k} coef[i] V[i][e]; coef lives in mapped host memory (k <= 256 doubles read once per workgroup)
__global__ __launch_bounds__(kVecT) void vec_lincomb_kernel(double* y, const double* V, long ldv, const double* coef, int k, long n) {
  __shared__ double cl[256];
  for (int i = threadIdx.x; i < k; i += kVecT) cl[i] = coef[i];
  __syncthreads();
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  double s = y[e];
  const double* col = V + e;
  int i = 0;
  for (; i + 8 <= k; i += 8) {             // eight rows' loads in flight (one load, one fma at a time waits out a latency per row)
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = col[(long)(i + q) * ldv];
#pragma unroll
    for (int q = 0; q < 8; ++q) s += cl[i + q] * v[q];
  }
  for (; i < k; ++i) s += cl[i] * col[(long)i * ldv];
  y[e] = s;
}

// |v| to mapped host memory: one workgroup of 1024 threads, fixed-order sums (systems here are <= ~1e6 unknowns)
__global__ __launch_bounds__(1024) void vec_norm_kernel(const double* v, long n, double* out_mapped) {
  __shared__ double ws[16];
  double s = 0.0;
  for (long e = threadIdx.x; e < n; e += 1024) s += v[e] * v[e];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int q = 0; q < 16; ++q) t += ws[q];
    *out_mapped = sqrt(t);
  }
}

__global__ __launch_bounds__(kVecT) void vec_div_kernel(double* out, const double* in, double alpha, long n) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) out[e] = in[e] / alpha;
}

inline unsigned blocks_of(long n) { return (unsigned)((n + kVecT - 1) / kVecT); }

// Square-root function of a symmetric tridiagonal matrix applied to e_1, by QL sweeps with implicit Wilkinson shifts (the classical
// tql2 scheme) WITHOUT accumulating the eigenvector matrix: T = Q L Q^T with Q = G_1 G_2 ... G_m a product of plane rotations,
// so  f(T) e_1 = Q (f(L) Q^T e_1)  needs the first ROW of Q -- carried along in O(1) per rotation -- and then the rotations
// applied once more, in reverse order, to the k-vector f(L) q_1: O(1) per rotation again.  The whole solve is O(k^2) instead
// of O(k^3); the unpreconditioned forcings of the roller schemes take 40-60 iterations, and the solve of every iteration has
// to fit into the ~35 us the device needs for one (k = 50: 180 us with the accumulated matrix, tests/test_capi_and_host.py
// holds the result against LAPACK).  d = diagonal (n), e = sub-diagonal in e[0 .. n-2] (e[n-1] is scratch), q1 = e_1 on entry.
// On exit d holds the eigenvalues (unsorted), q1 the first components of the eigenvectors, rot the rotations in the order
// applied.  false = a sweep did not converge in 60 rounds.
struct PlaneRotation { double c, s; int i; };      // acts on columns i, i + 1:  new_i = c old_i - s old_{i+1},  new_{i+1} = s old_i + c old_{i+1}

bool tridiagonal_ql(int n, double* d, double* e, double* q1, std::vector<PlaneRotation>& rot) {
  rot.clear();
  if (n > 0) e[n - 1] = 0.0;
  for (int l = 0; l < n; ++l) {
    int iter = 0, m;
    do {
      for (m = l; m < n - 1; ++m) {
        const double dd = fabs(d[m]) + fabs(d[m + 1]);
        if (fabs(e[m]) <= 2.220446049250313e-16 * dd) break;
      }
      if (m != l) {
        if (iter++ == 60) return false;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = sqrt(g * g + 1.0);
        g = d[m] - d[l] + e[l] / (g + copysign(r, g));
        double s = 1.0, c = 1.0, p = 0.0;
        int i;
        for (i = m - 1; i >= l; --i) {
          double f = s * e[i];
          const double b = c * e[i];
          r = sqrt(f * f + g * g);      // (entries of a Lanczos matrix of the mobility: no overflow to guard against; hypot() costs a third of a rotation)
          e[i + 1] = r;
          if (r == 0.0) {          // an exact zero on the sub-diagonal: deflate and start the sweep again
            d[i + 1] -= p;
            e[m] = 0.0;
            break;
          }
          const double ir = 1.0 / r;
          s = f * ir;
          c = g * ir;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * b;
          p = s * r;
          d[i + 1] = g + p;
          g = c * r - b;
          rot.push_back(PlaneRotation{c, s, i});
          const double fk = q1[i + 1];
          q1[i + 1] = s * q1[i] + c * fk;
          q1[i] = c * q1[i] - s * fk;
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p;
        e[l] = g;
        e[m] = 0.0;
      }
    } while (m != l);
  }
  return true;
}

// coef = scale * Q sqrt(max(lambda, 0)) Q^T e_1 for the k x k tridiagonal (h_diag, h_sup): the Lanczos noise estimate's
// coordinates in the Krylov basis (stochastic_forcing/stochastic_forcing.py:215-229 forms Q sqrt(L) Q^T e_1 |z| the same way)
struct NoiseWork { std::vector<double> buf; std::vector<PlaneRotation> rot; };

bool noise_coefficients(long k, const double* h_diag, const double* h_sup, double scale, double* coef, NoiseWork& work) {
  work.buf.resize(3 * (size_t)k);
  double* d = work.buf.data();
  double* e = d + k;
  double* q1 = e + k;
  for (long i = 0; i < k; ++i) { d[i] = h_diag[i]; e[i] = i + 1 < k ? h_sup[i] : 0.0; q1[i] = 0.0; }
  q1[0] = 1.0;
  if (!tridiagonal_ql((int)k, d, e, q1, work.rot)) return false;
  for (long j = 0; j < k; ++j) coef[j] = sqrt(d[j] > 0.0 ? d[j] : 0.0) * q1[j];          // f(L) Q^T e_1
  for (size_t t = work.rot.size(); t-- > 0;) {                                             // Q v = G_1 (G_2 (... (G_m v)))
    const PlaneRotation& g = work.rot[t];
    const double a = coef[g.i], b = coef[g.i + 1];
    coef[g.i] = g.c * a + g.s * b;
    coef[g.i + 1] = g.c * b - g.s * a;
  }
  for (long r = 0; r < k; ++r) coef[r] *= scale;
  return true;
}

struct Mapped {
  void* host = nullptr;
  void* dev = nullptr;
  size_t cap = 0;
};

}  // namespace
}  // namespace rmbi

using namespace rmbi;

// Per-context workspace of the native GMRES: device vectors in one DevBuf, mapped host memory for the Hessenberg columns, the
// back-substitution coefficients and one scalar.  Grows, never shrinks; freed with the context (rmb_ctx_destroy calls
// rmb_gmres_release).
struct rmb_gmres_ws {
  DevBuf dev;
  Mapped mapped;
  hipEvent_t ev[2] = {nullptr, nullptr};
};

namespace rmbi {
void gmres_release(rmb_ctx* c) {
  rmb_gmres_ws* w = (rmb_gmres_ws*)c->gmres_ws;
  if (!w) return;
  w->dev.release();
  if (w->mapped.host) (void)hipHostFree(w->mapped.host);
  for (auto& e : w->ev) if (e) (void)hipEventDestroy(e);
  delete w;
  c->gmres_ws = nullptr;
}
}  // namespace rmbi

extern "C" {

int rmb_rigid_gmres_device(rmb_ctx* c, long n_bodies, long n_b, const double* A11_dev, const double* A12_dev, const double* A21_dev,
                           const double* A22_dev, const double* K_dev, const double* b_dev, double tol, long restart, long maxiter,
                           double eta, double* x_dev, long* iterations, double* residual, long* discarded, long* products,
                           double* history, long history_cap, double* rhs_norm) {
  if (int rc = check_ready(c)) return rc;
  if (n_bodies < 1 || n_b < 1) return fail(RMB_ERR_ARG, "rmb_rigid_gmres_device: bad n_bodies / n_b");
  if (n_bodies * n_b != c->n) return fail(RMB_ERR_STATE, "rmb_rigid_gmres_device: the resident configuration does not hold n_bodies x n_b blobs");
  if (!A11_dev || !A12_dev || !A21_dev || !A22_dev || !K_dev || !b_dev || !x_dev || !iterations || !residual)
    return fail(RMB_ERR_ARG, "null pointer");
  if (restart < 1 || restart > 255 || maxiter < 0 || !(tol >= 0.0)) return fail(RMB_ERR_ARG, "rmb_rigid_gmres_device: need 1 <= restart <= 255, maxiter >= 0, tol >= 0");
  RMB_HIP(hipSetDevice(c->device));
  const long nn = 3 * n_b, n3 = 3 * c->n, n = n3 + 6 * n_bodies, ldv = n;
  const long m_max = restart;
  // ---- workspace ----
  if (!c->gmres_ws) c->gmres_ws = new rmb_gmres_ws();
  rmb_gmres_ws* ws = (rmb_gmres_ws*)c->gmres_ws;
  const size_t col_row = (size_t)(m_max + 2);
  const size_t dev_doubles = (size_t)(m_max + 1) * n + (size_t)5 * n + (size_t)m_max * col_row;
  if (int rc = ws->dev.reserve(dev_doubles * sizeof(double))) return rc;
  const size_t map_doubles = (size_t)m_max * col_row + 256 + 8;
  if (map_doubles * sizeof(double) > ws->mapped.cap) {
    if (ws->mapped.host) { RMB_HIP(hipStreamSynchronize(c->stream)); (void)hipHostFree(ws->mapped.host); ws->mapped = Mapped(); }
    RMB_HIP(hipHostMalloc(&ws->mapped.host, map_doubles * sizeof(double), hipHostMallocMapped));
    RMB_HIP(hipHostGetDevicePointer(&ws->mapped.dev, ws->mapped.host, 0));
    ws->mapped.cap = map_doubles * sizeof(double);
  }
  for (auto& e : ws->ev) if (!e) RMB_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  double* V = (double*)ws->dev.p;
  double* z = V + (size_t)(m_max + 1) * n;
  double* w = z + n;
  double* y = w + n;
  double* r = y + n;
  double* bs = r + n;                                          // the right-hand side scaled to unit norm (rhs_norm != NULL)
  double* cols = bs + n;                                       // [m_max][m_max + 2] on the device
  double* hcols = (double*)ws->mapped.host;                    // the same rows in mapped host memory
  double* hcols_dev = (double*)ws->mapped.dev;
  double* hcoef = hcols + (size_t)m_max * col_row;             // back-substitution coefficients (host writes, kernel reads)
  double* hcoef_dev = hcols_dev + (size_t)m_max * col_row;
  double* hscal = hcoef + 256;                                 // one scalar (a norm)
  double* hscal_dev = hcoef_dev + 256;
  hipStream_t s = c->stream;
  const rmb_block b11{A11_dev, nn * nn, nn, 1}, b12{A12_dev, nn * 6, 6, 1}, b21{A21_dev, 6 * nn, nn, 1}, b22{A22_dev, 36, 6, 1};
  long n_products = 0;

  auto host_norm = [&](const double* v, double* out) -> int {
    hipLaunchKernelGGL(vec_norm_kernel, dim3(1), dim3(1024), 0, s, v, n, hscal_dev);
    RMB_HIP(hipGetLastError());
    RMB_HIP(hipStreamSynchronize(s));
    *out = *hscal;
    return 0;
  };
  auto precondition = [&](const double* in, double* out) -> int {       // out = P^-1 in
    return rmb_block_apply_device(c, n_bodies, nn, nn, 6, 6, &b11, &b12, &b21, &b22, in, in + n3, 1.0, 0.0, out, 0.0, out + n3);
  };

  // rhs_norm != NULL: b_dev is the raw right-hand side; it is scaled to unit norm here, as the reference does before its
  // GMRES (:1518-1521), and the solution scaled back -- one norm + host wait instead of the caller's and ours
  double bnorm = 0.0, scale_back = 1.0;
  if (int rc = host_norm(b_dev, &bnorm)) return rc;
  const double* b_use = b_dev;
  if (rhs_norm) {
    *rhs_norm = bnorm;
    if (bnorm == 0.0) {
      hipLaunchKernelGGL(vec_zero_kernel, dim3(blocks_of(n)), dim3(kVecT), 0, s, x_dev, n);
      RMB_HIP(hipGetLastError());
      *iterations = 0; *residual = 0.0;
      if (discarded) *discarded = 0;
      if (products) *products = 0;
      return 0;
    }
    hipLaunchKernelGGL(vec_scale_kernel, dim3(blocks_of(n)), dim3(kVecT), 0, s, bs, b_dev, 1.0 / bnorm, n);
    b_use = bs;
    scale_back = bnorm;
    bnorm = 1.0;       // |rhs / |rhs|| (the Python loop measures it again and finds 1 to rounding)
  }
  hipLaunchKernelGGL(vec_zero_kernel, dim3(blocks_of(n)), dim3(kVecT), 0, s, y, n);
  RMB_HIP(hipMemcpyAsync(r, b_use, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
  double beta = bnorm;
  long its = 0, wasted = 0, n_hist = 0;
  double res = bnorm > 0.0 ? beta / bnorm : 0.0;
  std::vector<double> H((size_t)(m_max + 1) * m_max), cs(m_max), sn(m_max), g(m_max + 1), col(m_max + 2);

  while (its < maxiter && res > tol) {
    const long m = restart < maxiter - its ? restart : maxiter - its;
    hipLaunchKernelGGL(vec_scale_kernel, dim3(blocks_of(n)), dim3(kVecT), 0, s, V, r, 1.0 / beta, n);
    RMB_HIP(hipGetLastError());
    std::fill(H.begin(), H.end(), 0.0);
    std::fill(g.begin(), g.end(), 0.0);
    g[0] = beta;
    long k_used = 0;
    double prev_res = -1.0;           // < 0: none yet
    long pending = -1;
    bool stop = false;

    // Host side of iteration j: wait for its column, rotate, test.  true = stop after this column.
    auto finish = [&](long j, bool* out_stop) -> int {
      RMB_HIP(hipEventSynchronize(ws->ev[j & 1]));
      memcpy(col.data(), hcols + (size_t)j * col_row, (size_t)(j + 2) * sizeof(double));
      const double w_norm = col[j + 1];
      for (long i = 0; i < j; ++i) {
        const double t = cs[i] * col[i] + sn[i] * col[i + 1];
        col[i + 1] = -sn[i] * col[i] + cs[i] * col[i + 1];
        col[i] = t;
      }
      const double d = hypot(col[j], col[j + 1]);
      if (d > 0.0) { cs[j] = col[j] / d; sn[j] = col[j + 1] / d; } else { cs[j] = 1.0; sn[j] = 0.0; }
      col[j] = d;
      col[j + 1] = 0.0;
      for (long i = 0; i < j + 2; ++i) H[(size_t)i * m_max + j] = col[i];
      g[j + 1] = -sn[j] * g[j];
      g[j] = cs[j] * g[j];
      its += 1;
      k_used = j + 1;
      prev_res = res;
      res = fabs(g[j + 1]) / bnorm;
      if (history && n_hist < history_cap) history[n_hist] = res;
      ++n_hist;
      *out_stop = res <= tol || w_norm == 0.0 || !std::isfinite(w_norm);
      return 0;
    };
    auto may_defer = [&]() {
      const double rate = prev_res > 0.0 ? (res / prev_res < 1.0 ? res / prev_res : 1.0) : 1.0;
      return res * rate > 20.0 * tol;
    };

    // (wide blocks -- bodies of more than 32 blobs -- keep the separate launch: their product is no longer a 4.5 us kernel
    //  and the fused one measured 4 % slower, profiles/r5_gmres_fuse_pc.txt)
    const bool fuse_pc = c->opt_gmres_fuse_pc != 0 && 3 * n_b <= 96;
    for (long j = 0; j < m; ++j) {
      if (pending >= 0 && !may_defer()) {
        if (int rc = finish(pending, &stop)) return rc;
        pending = -1;
        if (stop) break;
      }
      // (from the second step of a cycle on, z = P^-1 v_j was left behind by the previous step's last launch: six launches)
      if (int rc = arnoldi_step_impl(c, n_bodies, n_b, A11_dev, A12_dev, A21_dev, A22_dev, K_dev, V, ldv, j, eta, z, w,
                                     cols + (size_t)j * col_row, hcols_dev + (size_t)j * col_row, fuse_pc && j > 0, fuse_pc))
        return rc;
      ++n_products;
      RMB_HIP(hipEventRecord(ws->ev[j & 1], s));
      if (pending >= 0) {
        if (int rc = finish(pending, &stop)) return rc;
        pending = -1;
        if (stop) { ++wasted; break; }      // step j was enqueued for nothing
      }
      pending = j;
    }
    if (pending >= 0 && !stop) {
      if (int rc = finish(pending, &stop)) return rc;
    }
    // y += V[:k]^T coef,  coef = triu(H[:k, :k])^-1 g[:k]
    if (k_used > 0) {
      for (long i = k_used - 1; i >= 0; --i) {
        double t = g[i];
        for (long q = i + 1; q < k_used; ++q) t -= H[(size_t)i * m_max + q] * hcoef[q];
        hcoef[i] = t / H[(size_t)i * m_max + i];
      }
      hipLaunchKernelGGL(vec_lincomb_kernel, dim3(blocks_of(n)), dim3(kVecT), 0, s, y, V, ldv, hcoef_dev, (int)k_used, n);
      RMB_HIP(hipGetLastError());
      // (no wait here: the host writes hcoef again only after it has waited for an event of a LATER step of this stream --
      //  of the next cycle, or of the next call on this context -- and by then this kernel has run)
    }
    if (res > tol && its < maxiter) {           // restart: true residual
      if (int rc = precondition(y, z)) return rc;
      if (int rc = rmb_rigid_operator_device(c, n_bodies, n_b, K_dev, z, eta, w)) return rc;
      ++n_products;
      hipLaunchKernelGGL(vec_sub_kernel, dim3(blocks_of(n)), dim3(kVecT), 0, s, r, b_use, w, n);
      RMB_HIP(hipGetLastError());
      if (int rc = host_norm(r, &beta)) return rc;
      res = beta / bnorm;
    }
  }
  if (scale_back != 1.0) {                                // x = |rhs| P^-1 y
    hipLaunchKernelGGL(vec_scale_kernel, dim3(blocks_of(n)), dim3(kVecT), 0, s, y, y, scale_back, n);
    RMB_HIP(hipGetLastError());
  }
  if (int rc = precondition(y, x_dev)) return rc;        // x = P^-1 y
  *iterations = its;
  *residual = res;
  if (discarded) *discarded = wasted;
  if (products) *products = n_products;
  return 0;
}

// Host function (no GPU work): the coefficients rmb_rigid_lanczos_device derives from the tridiagonal matrix after k
// iterations; exported so that the CPU test suite can hold the library's eigen-solver against LAPACK.
int rmb_lanczos_noise_coefficients(long k, const double* h_diag, const double* h_sup, double scale, double* coef_out) {
  if (k < 1 || k > 4096 || !h_diag || (k > 1 && !h_sup) || !coef_out) return fail(RMB_ERR_ARG, "rmb_lanczos_noise_coefficients: bad arguments");
  NoiseWork work;
  if (!noise_coefficients(k, h_diag, h_sup, scale, coef_out, work)) return fail(RMB_ERR_STATE, "rmb_lanczos_noise_coefficients: the QL sweeps did not converge");
  return 0;
}

// ---- the Lanczos forcing loops ------------------------------------------------------------------------------------
// factor * f(z) with the square root of an SPD operator applied through a Krylov basis (stochastic_forcing/stochastic_forcing.py:
// 112-264): per iteration ONE step call (product + full re-orthogonalisation, which also stores h_ii and h_{i+1,i} into mapped
// host memory) and one event; the host side -- the small tridiagonal eigenproblem and the reference's stopping rule (:239-255:
// relative change of the noise estimate, measured on its coordinates in the orthonormal basis) -- runs ONE ITERATION LATE,
// while the device works on the next step (a Python iteration costs 60-120 us of host time against ~35 us of launches).
// status: 0 = done; 1 = exact breakdown or a failed eigen-solve, 2 = more than max_rows basis vectors needed -- in both cases
// nothing was written to the result, the stream has been drained and the caller runs its general loop.
}  // extern "C"

namespace rmbi {
namespace {
struct LanczosBuffers {
  double* V; long ldv;               // basis, (cap + 1) rows
  double *x0, *x1, *x2;              // three work vectors of the step
  double* cols; double* hcols_dev;   // coefficient rows: device copy and the mapped one
  size_t col_row;
};

// step(i, buffers): enqueue iteration i (its coefficients go to cols + i * col_row and hcols_dev + i * col_row);
// finish(combo, buffers): combo = V[:k]^T coef is enqueued in buffers.x1; write the result
template <class Step, class Finish>
int lanczos_loop(rmb_ctx* c, const char* who, long dim, const double* z_dev, double factor, double tol, long max_iter, long cap, Step step,
                 Finish finish_result, long* iterations, long* products, int* status) {
  if (!c->gmres_ws) c->gmres_ws = new rmb_gmres_ws();
  rmb_gmres_ws* ws = (rmb_gmres_ws*)c->gmres_ws;
  const size_t col_row = (size_t)(cap + 2);
  const size_t dev_doubles = (size_t)(cap + 1) * dim + (size_t)3 * dim + (size_t)cap * col_row;
  if (int rc = ws->dev.reserve(dev_doubles * sizeof(double))) return rc;
  const size_t map_doubles = (size_t)cap * col_row + 256 + 8;
  if (map_doubles * sizeof(double) > ws->mapped.cap) {
    if (ws->mapped.host) { RMB_HIP(hipStreamSynchronize(c->stream)); (void)hipHostFree(ws->mapped.host); ws->mapped = Mapped(); }
    RMB_HIP(hipHostMalloc(&ws->mapped.host, map_doubles * sizeof(double), hipHostMallocMapped));
    RMB_HIP(hipHostGetDevicePointer(&ws->mapped.dev, ws->mapped.host, 0));
    ws->mapped.cap = map_doubles * sizeof(double);
  }
  for (auto& e : ws->ev) if (!e) RMB_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  LanczosBuffers b;
  b.V = (double*)ws->dev.p; b.ldv = dim;
  b.x0 = b.V + (size_t)(cap + 1) * dim;
  b.x1 = b.x0 + dim;
  b.x2 = b.x1 + dim;
  b.cols = b.x2 + dim;
  b.col_row = col_row;
  double* hcols = (double*)ws->mapped.host;
  b.hcols_dev = (double*)ws->mapped.dev;
  double* hcoef = hcols + (size_t)cap * col_row;
  double* hcoef_dev = b.hcols_dev + (size_t)cap * col_row;
  double* hscal = hcoef + 256;
  double* hscal_dev = hcoef_dev + 256;
  hipStream_t s = c->stream;
  (void)who;
  *status = 0;
  *iterations = 0;
  long n_products = 0;
  auto give_up = [&](int st) -> int {
    RMB_HIP(hipStreamSynchronize(s));
    *status = st;
    if (products) *products = n_products;
    return 0;
  };

  hipLaunchKernelGGL(vec_norm_kernel, dim3(1), dim3(1024), 0, s, z_dev, dim, hscal_dev);
  RMB_HIP(hipGetLastError());
  RMB_HIP(hipStreamSynchronize(s));
  const double v_norm = *hscal;
  if (!(v_norm > 0.0) || !std::isfinite(v_norm)) return give_up(1);
  hipLaunchKernelGGL(vec_div_kernel, dim3(blocks_of(dim)), dim3(kVecT), 0, s, b.V, z_dev, v_norm, dim);
  RMB_HIP(hipGetLastError());

  std::vector<double> h_diag, h_sup, coef, coef_old;
  NoiseWork work;
  auto enqueue = [&](long i) -> int {
    if (int rc = step(i, b)) return rc;
    ++n_products;
    RMB_HIP(hipEventRecord(ws->ev[i & 1], s));
    return 0;
  };
  // host side of iteration i: 1 = stop (converged), 0 = go on, -1 = breakdown
  long its = -1;
  auto finish = [&](long i, int* verdict) -> int {
    RMB_HIP(hipEventSynchronize(ws->ev[i & 1]));
    const double hd = hcols[(size_t)i * col_row + i], hs = hcols[(size_t)i * col_row + i + 1];
    if (!(hs > 0.0) || !std::isfinite(hs)) { *verdict = -1; return 0; }
    h_diag.push_back(hd);
    h_sup.push_back(hs);
    coef.resize((size_t)i + 1);
    if (!noise_coefficients(i + 1, h_diag.data(), h_sup.data(), v_norm * factor, coef.data(), work)) { *verdict = -1; return 0; }
    if (i > 0) {
      double old2 = 0.0, diff2 = 0.0;
      for (long q = 0; q < i; ++q) { old2 += coef_old[q] * coef_old[q]; const double t = coef[q] - coef_old[q]; diff2 += t * t; }
      diff2 += coef[i] * coef[i];
      const double old_norm = sqrt(old2), diff = sqrt(diff2);
      if (diff / (old_norm > 2.220446049250313e-16 ? old_norm : 2.220446049250313e-16) < tol) { its = i; *verdict = 1; return 0; }
    }
    coef_old = coef;
    *verdict = 0;
    return 0;
  };

  if (int rc = enqueue(0)) return rc;
  long i = 0;
  while (true) {
    const long nxt = i + 1;
    if (nxt < cap && nxt <= max_iter) {
      if (int rc = enqueue(nxt)) return rc;        // the device goes on while the host looks at iteration i
    }
    int verdict = 0;
    if (int rc = finish(i, &verdict)) return rc;
    if (verdict < 0) return give_up(1);
    if (verdict > 0) break;
    if (nxt > max_iter) { its = max_iter; break; }
    if (nxt >= cap) return give_up(2);
    i = nxt;
  }
  // V[:k]^T coef
  const long k = (long)coef.size();
  memcpy(hcoef, coef.data(), (size_t)k * sizeof(double));
  hipLaunchKernelGGL(vec_zero_kernel, dim3(blocks_of(dim)), dim3(kVecT), 0, s, b.x1, dim);
  hipLaunchKernelGGL(vec_lincomb_kernel, dim3(blocks_of(dim)), dim3(kVecT), 0, s, b.x1, b.V, b.ldv, hcoef_dev, (int)k, dim);
  RMB_HIP(hipGetLastError());
  if (int rc = finish_result(b.x1, b)) return rc;
  // (no wait: whoever writes hcoef next -- a Lanczos entry or rmb_rigid_gmres_device -- has by then waited for an event recorded
  //  later on this stream; the result is enqueued, the scalars are final)
  *iterations = its;
  if (products) *products = n_products;
  return 0;
}
}  // namespace
}  // namespace rmbi

extern "C" {

// The preconditioned forcing of the rigid-body schemes:  noise = factor * blockdiag(L_b) (P^T M P)^{1/2} z,  P = blockdiag(L_b^-T)
// (quaternion_integrator_multi_bodies.py:966-973 with the preconditioner of multi_bodies.py:590-614): rigid.py's _lanczos_native
// loop, natively.
int rmb_rigid_lanczos_device(rmb_ctx* c, long n_bodies, long n_b, const double* Linv_dev, const double* Lchol_dev, const double* z_dev,
                             double factor, double tol, long max_iter, long max_rows, double eta, double* noise_dev, long* iterations,
                             long* products, int* status) {
  if (int rc = check_ready(c)) return rc;
  if (n_bodies < 1 || n_b < 1) return fail(RMB_ERR_ARG, "rmb_rigid_lanczos_device: bad n_bodies / n_b");
  if (n_bodies * n_b != c->n) return fail(RMB_ERR_STATE, "rmb_rigid_lanczos_device: the resident configuration does not hold n_bodies x n_b blobs");
  if (!Linv_dev || !Lchol_dev || !z_dev || !noise_dev || !iterations || !status) return fail(RMB_ERR_ARG, "null pointer");
  if (max_rows < 2 || max_rows > 254 || max_iter < 1 || !(tol >= 0.0)) return fail(RMB_ERR_ARG, "rmb_rigid_lanczos_device: need 2 <= max_rows <= 254, max_iter >= 1, tol >= 0");
  RMB_HIP(hipSetDevice(c->device));
  const long nn = 3 * n_b, n3 = 3 * c->n;
  const bool fuse_next = c->opt_gmres_fuse_pc != 0 && nn <= 96;      // (as in rmb_rigid_gmres_device)
  auto step = [&](long i, const LanczosBuffers& b) -> int {
    // x0: P v_i (from the second step on left there by the previous step's normalisation launch), x1: the sweep's raw sums,
    // x2: P^T M P v_i, orthogonalised in place -- five launches per iteration
    return lanczos_step_impl(c, n_bodies, n_b, Linv_dev, b.V, b.ldv, i, eta, b.x0, b.x1, b.x2, b.cols + (size_t)i * b.col_row,
                             b.hcols_dev + (size_t)i * b.col_row, fuse_next && i > 0, fuse_next);
  };
  auto result = [&](double* combo, const LanczosBuffers&) -> int {       // noise = blockdiag(L_b) V[:k]^T coef
    const rmb_block l{Lchol_dev, nn * nn, nn, 1};
    return rmb_block_apply_device(c, n_bodies, nn, nn, 0, 0, &l, nullptr, nullptr, nullptr, combo, nullptr, 1.0, 0.0, noise_dev, 0.0, nullptr);
  };
  return lanczos_loop(c, "rmb_rigid_lanczos_device", n3, z_dev, factor, tol, max_iter, max_rows, step, result, iterations, products, status);
}

// The plain forcing of the single-blob schemes:  noise = factor * M^{1/2} z  with M = M_tt (product 0; in_plane: its in-plane
// variant) over 3 N unknowns, or the 6 N x 6 N grand mobility [[M_tt, M_tr], [M_rt, M_rr]] (product 1, z = [z_f; z_tau])
// (quaternion_integrator_rollers.py:1082-1121, :1203-1260, :1315-1353 -> stochastic_forcing_lanczos without preconditioner).
int rmb_lanczos_device(rmb_ctx* c, int product, int in_plane, const double* z_dev, double factor, double tol, long max_iter, long max_rows,
                       double eta, double* noise_dev, long* iterations, long* products, int* status) {
  if (int rc = check_ready(c)) return rc;
  if (product != 0 && product != 1) return fail(RMB_ERR_ARG, "rmb_lanczos_device: product must be 0 (M_tt) or 1 (grand mobility)");
  if (product == 1 && in_plane) return fail(RMB_ERR_ARG, "rmb_lanczos_device: the grand mobility has no in-plane variant");
  if (c->tgt_begin != 0 || c->tgt_end != c->n) return fail(RMB_ERR_STATE, "rmb_lanczos_device: needs the full target range");
  if (!z_dev || !noise_dev || !iterations || !status) return fail(RMB_ERR_ARG, "null pointer");
  if (max_rows < 2 || max_rows > 254 || max_iter < 1 || !(tol >= 0.0)) return fail(RMB_ERR_ARG, "rmb_lanczos_device: need 2 <= max_rows <= 254, max_iter >= 1, tol >= 0");
  if (!(eta > 0.0)) return fail(RMB_ERR_ARG, "eta must be positive");
  RMB_HIP(hipSetDevice(c->device));
  const long n3 = 3 * c->n, dim = product == 1 ? 2 * n3 : n3;
  auto step = [&](long i, const LanczosBuffers& b) -> int {
    const double* v = b.V + i * b.ldv;
    long tiles = 0;
    if (product == 0 && !in_plane) {
      // (small decks: the finishing launch also takes the first Gram-Schmidt pass's dots, five launches per iteration)
      if (int rc = plain_tt_with_dots(c, v, eta, b.x1, b.V, b.ldv, i + 1, &tiles)) return rc;
    } else if (product == 0) {
      if (int rc = matvec_device_impl(c, rmb::KIND_TT, 1, v, nullptr, eta, b.x1)) return rc;
    } else {
      const double* in[2] = {v, v + n3};
      double* out[2] = {b.x1, b.x1 + n3};
      if (int rc = rmb_matvec_op_device(c, RMB_OP_GRAND, 0, 2, in, 2, out, eta)) return rc;
    }
    return krylov_orthogonalize_impl(c, dim, i + 1, b.V, b.ldv, b.x1, b.cols + (size_t)i * b.col_row, b.V + (i + 1) * b.ldv,
                                     b.hcols_dev + (size_t)i * b.col_row, nullptr, tiles, true);
  };
  auto result = [&](double* combo, const LanczosBuffers&) -> int {
    RMB_HIP(hipMemcpyAsync(noise_dev, combo, (size_t)dim * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return 0;
  };
  return lanczos_loop(c, "rmb_lanczos_device", dim, z_dev, factor, tol, max_iter, max_rows, step, result, iterations, products, status);
}

}  // extern "C"
