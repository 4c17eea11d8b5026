// rmb_sweep.hip -- launchers of the one-sided kernels (lane = target, sources streamed through an LDS tile, source
// chunks + fixed-order reduction, atomic-free): the pair sweep of every kind, the force sweep, the source->target
// operators with per-blob radii, Stokeslet pressure / Stokes double layer, the dense per-body blocks; position packing.
#include "rmb_internal.h"

#include <cmath>

#include "matvec_kernels.h"
#include "dense_kernels.h"
#include "st_kernels.h"
#include "aux_kernels.h"
#include "diag_kernels.h"

namespace rmbi {

namespace {
typedef void (*sweep_fn)(const rmb::SweepArgs);
typedef void (*final_fn)(const rmb::SweepArgs);

struct KernelEntry { sweep_fn sweep; final_fn fin; int blocks_per_cu; };

template <int KIND, bool WALL, bool PER>
KernelEntry make_entry() {
  KernelEntry e;
  e.sweep = rmb::sweep_kernel<KIND, WALL, PER>;
  e.fin = rmb::finalize_kernel<KIND, WALL>;
  e.blocks_per_cu = 0;
  return e;
}

// [kind][wall][periodic]
KernelEntry g_kernels[rmb::KIND_COUNT][2][2] = {
#define RMB_ROW(K) {{make_entry<K, false, false>(), make_entry<K, false, true>()}, {make_entry<K, true, false>(), make_entry<K, true, true>()}}
    RMB_ROW(rmb::KIND_TT), RMB_ROW(rmb::KIND_TR), RMB_ROW(rmb::KIND_RT), RMB_ROW(rmb::KIND_RR), RMB_ROW(rmb::KIND_TT_TR),
    RMB_ROW(rmb::KIND_TT_FREE)
#undef RMB_ROW
};
}  // namespace

// One-sided sweep of targets [tgt_begin, tgt_end) against all n sources (every ordered pair, fixed summation order).
int sweep_device(rmb_ctx* c, int kind, int in_plane, const double* v, const double* v2, double eta, double* out) {
  const bool periodic = c->L[0] > 0 || c->L[1] > 0 || c->L[2] > 0;
  const long n_tgt = c->tgt_end - c->tgt_begin;
  c->last_path = 0;
  KernelEntry& ke = g_kernels[kind][c->wall ? 1 : 0][periodic ? 1 : 0];
  const long slots = c->n_cu * resident_blocks((const void*)ke.sweep, &ke.blocks_per_cu);
  long n_chunks, chunk_len;
  choose_chunks(n_tgt, c->n, c->opt_chunks, slots, &n_chunks, &chunk_len);
  const long tiles = (n_tgt + 63) / 64;
  if (tiles > 0x7fffffffL || n_chunks > 65535) return fail(RMB_ERR_ARG, "problem too large for one launch");

  rmb::SweepArgs a;
  a.pos = (const double4*)c->pos.p;
  a.vec = v;
  a.vec2 = v2;
  a.out = out;
  a.partial = nullptr;
  a.n_src = c->n;
  a.tgt_begin = c->tgt_begin;
  a.tgt_end = c->tgt_end;
  a.n_tgt_pad = 64 * tiles;
  a.chunk_len = chunk_len;
  a.n_chunks = (int)n_chunks;
  a.in_plane = in_plane ? 1 : 0;
  a.prefactor = 1.0 / (8.0 * M_PI * eta);
  a.Lx = c->L[0]; a.Ly = c->L[1]; a.Lz = c->L[2];
  a.iLx = c->L[0] > 0 ? 1.0 / c->L[0] : 0.0;
  a.iLy = c->L[1] > 0 ? 1.0 / c->L[1] : 0.0;
  a.iLz = c->L[2] > 0 ? 1.0 / c->L[2] : 0.0;
  a.k = make_pair_consts(c->a);
  if (n_chunks > 1) {
    if (int rc = c->partial.reserve((size_t)n_chunks * 3 * a.n_tgt_pad * sizeof(double))) return rc;
    a.partial = (double*)c->partial.p;
  }
  const dim3 grid((unsigned)tiles, (unsigned)n_chunks);
  c->last_tiles = tiles; c->last_chunks = n_chunks; c->last_wgs = tiles * n_chunks;

  int slot;
  if (int rc = timing_begin(c, &slot)) return rc;
  hipLaunchKernelGGL(ke.sweep, grid, dim3(rmb::kBlock), 0, c->stream, a);
  RMB_HIP(hipGetLastError());
  if (int rc = timing_end(c, slot)) return rc;
  if (n_chunks > 1) {
    hipLaunchKernelGGL(ke.fin, dim3((unsigned)((n_tgt + 255) / 256)), dim3(256), 0, c->stream, a);
    RMB_HIP(hipGetLastError());
  }
  return 0;
}

// One-sided force sweep of the context's target range (multi_bodies/forces_numba.py:12-55), atomic-free.
int force_sweep_device(rmb_ctx* c, double eps, double b, double blob_radius, double* out, const double* radii) {
  const bool periodic = c->L[0] > 0 || c->L[1] > 0 || c->L[2] > 0;
  const long n_tgt = c->tgt_end - c->tgt_begin;
  c->last_path = 0;
  static int force_occ[2][2] = {{0, 0}, {0, 0}};
  typedef void (*force_fn)(const rmb::ForceArgs);
  const force_fn ffn = radii ? (periodic ? (force_fn)rmb::force_sweep_kernel<true, true> : (force_fn)rmb::force_sweep_kernel<false, true>)
                             : (periodic ? (force_fn)rmb::force_sweep_kernel<true, false> : (force_fn)rmb::force_sweep_kernel<false, false>);
  const long slots = c->n_cu * resident_blocks((const void*)ffn, &force_occ[radii ? 1 : 0][periodic ? 1 : 0]);
  long n_chunks, chunk_len;
  choose_chunks(n_tgt, c->n, c->opt_chunks, slots, &n_chunks, &chunk_len);
  const long tiles = (n_tgt + 63) / 64;
  if (tiles > 0x7fffffffL || n_chunks > 65535) return fail(RMB_ERR_ARG, "problem too large for one launch");
  rmb::ForceArgs a;
  a.pos = (const double4*)c->pos.p;
  a.out = out;
  a.partial = nullptr;
  a.n_src = c->n;
  a.tgt_begin = c->tgt_begin; a.tgt_end = c->tgt_end;
  a.n_tgt_pad = 64 * tiles;
  a.chunk_len = chunk_len;
  a.n_chunks = (int)n_chunks;
  a.Lx = c->L[0]; a.Ly = c->L[1]; a.Lz = c->L[2];
  a.iLx = c->L[0] > 0 ? 1.0 / c->L[0] : 0.0;
  a.iLy = c->L[1] > 0 ? 1.0 / c->L[1] : 0.0;
  a.iLz = c->L[2] > 0 ? 1.0 / c->L[2] : 0.0;
  a.eps_over_b = eps / b;
  a.inv_b = 1.0 / b;
  a.two_a = 2.0 * blob_radius;
  a.ec = exp_consts();
  a.radii = radii;
  if (n_chunks > 1) {
    if (int rc = c->partial.reserve((size_t)n_chunks * 3 * a.n_tgt_pad * sizeof(double))) return rc;
    a.partial = (double*)c->partial.p;
  }
  const dim3 grid((unsigned)tiles, (unsigned)n_chunks), block(rmb::kBlock);
  c->last_tiles = tiles; c->last_chunks = n_chunks; c->last_wgs = tiles * n_chunks;
  int slot;
  if (int rc = timing_begin(c, &slot)) return rc;
  hipLaunchKernelGGL(ffn, grid, block, 0, c->stream, a);
  RMB_HIP(hipGetLastError());
  if (int rc = timing_end(c, slot)) return rc;
  if (n_chunks > 1) {
    hipLaunchKernelGGL(rmb::force_finalize_kernel, dim3((unsigned)((n_tgt + 255) / 256)), dim3(256), 0, c->stream, a);
    RMB_HIP(hipGetLastError());
  }
  return 0;
}

int pack_positions(rmb_ctx* c, const double* r_dev, long n, double a, const double* L, int wall) {
  if (int rc = c->pos.reserve((size_t)(n > 0 ? n : 1) * sizeof(double4))) return rc;
  if (n > 0) {
    hipLaunchKernelGGL(rmb::pack_positions_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, r_dev, n,
                       a, wall ? 1 : 0, (double4*)c->pos.p);
    RMB_HIP(hipGetLastError());
  }
  c->n = n;
  c->a = a;
  c->tile_bounds_valid = false;
  for (int k = 0; k < 3; ++k) c->L[k] = L ? L[k] : 0.0;
  c->wall = wall ? 1 : 0;
  c->tgt_begin = 0;
  c->tgt_end = n;
  c->have_positions = true;
  return 0;
}

int pack_positions_radii(rmb_ctx* c, const double* r_dev, const double* rad_dev, long n, int wall, double4* dst) {
  hipLaunchKernelGGL(rmb::pack_positions_radii_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, r_dev,
                     rad_dev, n, wall, dst);
  RMB_HIP(hipGetLastError());
  return 0;
}

namespace {
__global__ void add_inplace_kernel(double* y, const double* x, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += x[i];
}
}  // namespace

namespace {
// dst (device memory) = src (page-locked host memory mapped into the device's address space): 16 bytes per lane,
// consecutive lanes consecutive addresses -- the upload of rmb_matvec's input vectors as a kernel of the SAME queue
__global__ __launch_bounds__(256) void pull_mapped_kernel(double* dst, const double* src, long n) {
  const long i = 2 * ((long)blockIdx.x * blockDim.x + threadIdx.x);
  if (i + 1 < n) *reinterpret_cast<double2*>(dst + i) = *reinterpret_cast<const double2*>(src + i);
  else if (i < n) dst[i] = src[i];
}
}  // namespace

int pull_mapped(rmb_ctx* c, double* dst_dev, const double* src_mapped_dev, long n) {
  hipLaunchKernelGGL(pull_mapped_kernel, dim3((unsigned)((n / 2 + 256) / 256)), dim3(256), 0, c->stream, dst_dev, src_mapped_dev, n);
  RMB_HIP(hipGetLastError());
  return 0;
}

int add_inplace(rmb_ctx* c, double* y, const double* x, long n) {
  hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, y, x, n);
  RMB_HIP(hipGetLastError());
  return 0;
}

int body_dense_device(rmb_ctx* c, const long* first_blob_dev, long n_bodies, int n_b, double eta, double* out_dev) {
  rmb::DenseArgs a;
  a.pos = (const double4*)c->pos.p;
  a.first_blob = first_blob_dev;
  a.out = out_dev;
  a.n_b = n_b;
  a.n_bodies = n_bodies;
  a.prefactor = 1.0 / (8.0 * M_PI * eta);
  a.k = make_pair_consts(c->a);
  // blockIdx.y splits the n_b^2 blob pairs of a body so that one big "body" (the dense builders) still fills the chip
  long ysplit = ((long)n_b * n_b + 256L * 16 - 1) / (256L * 16);   // 256 threads x 16 blob pairs each
  if (ysplit < 1) ysplit = 1;
  if (ysplit > 4096) ysplit = 4096;
  const dim3 grid((unsigned)n_bodies, (unsigned)ysplit);
  if (c->wall) hipLaunchKernelGGL(rmb::body_dense_tt_kernel<true>, grid, dim3(256), 0, c->stream, a);
  else         hipLaunchKernelGGL(rmb::body_dense_tt_kernel<false>, grid, dim3(256), 0, c->stream, a);
  RMB_HIP(hipGetLastError());
  return 0;
}


// One-sided source -> target sweep with per-blob radii (st_kernels.h); positions already packed (clamped when wall = 1).
int st_sweep_device(rmb_ctx* c, long ns, const double4* src_packed, const double* rad_s, const double* force, long nt,
                    const double4* tgt_packed, const double* rad_t, double eta, const double* L, int wall, double* out) {
  rmb::StArgs a;
  a.src = src_packed; a.rad_s = rad_s; a.force = force;
  a.tgt = tgt_packed; a.rad_t = rad_t; a.out = out; a.partial = nullptr;
  a.ns = ns; a.nt = nt;
  const long tiles = (nt + 63) / 64;
  a.n_tgt_pad = 64 * tiles;
  a.prefactor = 1.0 / (8.0 * M_PI * eta);
  const double Lx = L ? L[0] : 0.0, Ly = L ? L[1] : 0.0, Lz = L ? L[2] : 0.0;
  a.Lx = Lx; a.Ly = Ly; a.Lz = Lz;
  a.iLx = Lx > 0 ? 1.0 / Lx : 0.0; a.iLy = Ly > 0 ? 1.0 / Ly : 0.0; a.iLz = Lz > 0 ? 1.0 / Lz : 0.0;
  const bool periodic = Lx > 0 || Ly > 0 || Lz > 0;
  static int occ[3][2] = {{0, 0}, {0, 0}, {0, 0}};
  typedef void (*st_fn)(const rmb::StArgs);
  st_fn fn = wall == 2 ? (periodic ? (st_fn)rmb::st_sweep_kernel<2, true> : (st_fn)rmb::st_sweep_kernel<2, false>)
           : wall      ? (periodic ? (st_fn)rmb::st_sweep_kernel<1, true> : (st_fn)rmb::st_sweep_kernel<1, false>)
                       : (periodic ? (st_fn)rmb::st_sweep_kernel<0, true> : (st_fn)rmb::st_sweep_kernel<0, false>);
  const long slots = c->n_cu * resident_blocks((const void*)fn, &occ[wall == 2 ? 2 : (wall ? 1 : 0)][periodic ? 1 : 0]);
  long n_chunks, chunk_len;
  choose_chunks(nt, ns, c->opt_chunks, slots, &n_chunks, &chunk_len);
  if (tiles > 0x7fffffffL || n_chunks > 65535) return fail(RMB_ERR_ARG, "problem too large for one launch");
  a.chunk_len = chunk_len; a.n_chunks = (int)n_chunks;
  if (n_chunks > 1) {
    if (int rc = c->partial.reserve((size_t)n_chunks * 3 * a.n_tgt_pad * sizeof(double))) return rc;
    a.partial = (double*)c->partial.p;
  }
  c->last_path = 0; c->last_tiles = tiles; c->last_chunks = n_chunks; c->last_wgs = tiles * n_chunks;
  int slot;
  if (int rc = timing_begin(c, &slot)) return rc;
  hipLaunchKernelGGL(fn, dim3((unsigned)tiles, (unsigned)n_chunks), dim3(rmb::kBlock), 0, c->stream, a);
  RMB_HIP(hipGetLastError());
  if (int rc = timing_end(c, slot)) return rc;
  if (n_chunks > 1) {
    hipLaunchKernelGGL(rmb::st_finalize_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, c->stream, a);
    RMB_HIP(hipGetLastError());
  }
  return 0;
}

namespace {
template <int MODE>
int aux_launch(rmb_ctx* c, rmb::AuxArgs a) {
  typedef void (*aux_fn)(const rmb::AuxArgs);
  constexpr int NOUT = rmb::AuxShape<MODE>::NOUT;
  static int occ = 0;
  aux_fn fn = (aux_fn)rmb::aux_sweep_kernel<MODE>;
  const long tiles = (a.nt + 63) / 64;
  a.n_tgt_pad = 64 * tiles;
  const long slots = c->n_cu * resident_blocks((const void*)fn, &occ);
  long n_chunks, chunk_len;
  choose_chunks(a.nt, a.ns, c->opt_chunks, slots, &n_chunks, &chunk_len);
  if (tiles > 0x7fffffffL || n_chunks > 65535) return fail(RMB_ERR_ARG, "problem too large for one launch");
  a.chunk_len = chunk_len; a.n_chunks = (int)n_chunks; a.partial = nullptr;
  if (n_chunks > 1) {
    if (int rc = c->partial.reserve((size_t)n_chunks * NOUT * a.n_tgt_pad * sizeof(double))) return rc;
    a.partial = (double*)c->partial.p;
  }
  c->last_path = 0; c->last_tiles = tiles; c->last_chunks = n_chunks; c->last_wgs = tiles * n_chunks;
  int slot;
  if (int rc = timing_begin(c, &slot)) return rc;
  hipLaunchKernelGGL(fn, dim3((unsigned)tiles, (unsigned)n_chunks), dim3(rmb::kBlock), 0, c->stream, a);
  RMB_HIP(hipGetLastError());
  if (int rc = timing_end(c, slot)) return rc;
  if (n_chunks > 1) {
    hipLaunchKernelGGL(rmb::aux_finalize_kernel<NOUT>, dim3((unsigned)((a.nt + 255) / 256)), dim3(256), 0, c->stream, a);
    RMB_HIP(hipGetLastError());
  }
  return 0;
}
}  // namespace

int pressure_device(rmb_ctx* c, long ns, const double* src_dev, long nt, const double* tgt_dev, const double* force_dev,
                    int wall, double* out_dev) {
  rmb::AuxArgs a{};
  a.src = src_dev; a.tgt = tgt_dev; a.v0 = force_dev; a.v1 = nullptr; a.w = nullptr; a.out = out_dev;
  a.ns = ns; a.nt = nt; a.prefactor = 1.0 / (4.0 * M_PI); a.a2 = 0.0;
  return wall ? aux_launch<rmb::AUX_P_WALL>(c, a) : aux_launch<rmb::AUX_P_FREE>(c, a);
}

int double_layer_device(rmb_ctx* c, long ns, const double* src_dev, long nt, const double* tgt_dev,
                        const double* normals_dev, const double* vector_dev, const double* weights_dev, int wall,
                        double blob_radius, double* out_dev) {
  rmb::AuxArgs a{};
  a.src = src_dev; a.tgt = tgt_dev; a.v0 = normals_dev; a.v1 = vector_dev; a.w = weights_dev; a.out = out_dev;
  a.ns = ns; a.nt = nt; a.prefactor = -3.0 / (4.0 * M_PI); a.a2 = blob_radius >= 0.0 ? blob_radius * blob_radius : 0.0;
  if (blob_radius >= 0.0) return aux_launch<rmb::AUX_DL_RPY>(c, a);
  return wall ? aux_launch<rmb::AUX_DL_WALL>(c, a) : aux_launch<rmb::AUX_DL_FREE>(c, a);
}

int ubench_fp64_issue(rmb_ctx* c, int launches, double* g_wave_instr_per_s) {
  RMB_HIP(hipSetDevice(c->device));
  const long blocks = c->n_cu * 4;                    // 4 workgroups of 4 waves per CU = 4 waves per SIMD
  if (int rc = c->tmp3n.reserve((size_t)blocks * 256 * sizeof(double))) return rc;
  hipEvent_t e0, e1;
  RMB_HIP(hipEventCreate(&e0));
  RMB_HIP(hipEventCreate(&e1));
  hipLaunchKernelGGL(rmb::ubench_fma64_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, (double*)c->tmp3n.p, 1.0000001, 1e-9);
  RMB_HIP(hipEventRecord(e0, c->stream));
  for (int i = 0; i < launches; ++i)
    hipLaunchKernelGGL(rmb::ubench_fma64_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, (double*)c->tmp3n.p, 1.0000001, 1e-9);
  RMB_HIP(hipEventRecord(e1, c->stream));
  RMB_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  RMB_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  const double instr = (double)launches * blocks * 4 * rmb::kUbenchIters * rmb::kUbenchFmaPerIter;
  *g_wave_instr_per_s = instr / (ms * 1e-3) / 1e9;
  return 0;
}

}  // namespace rmbi
