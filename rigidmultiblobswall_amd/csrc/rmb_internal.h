// rmb_internal.h -- what the translation units of librmb_mobility.so share (never installed; the boundary is
// include/rmb_mobility.h).
//
//   rmb_context.hip  error state, context life cycle, streams, options, timing ring, diagnostics, default context
//   rmb_plan.hip     launch plans: source chunks, residency, the balanced step schedule of the symmetric kernels,
//                    pair-shard ranges, kernel-uniform constants
//   rmb_sym.hip      launchers of the symmetric (each unordered pair once) fp64 kernels: sym / sym2 / symx / symx_det,
//                    the symmetric force kernel
//   rmb_sym32.hip    their single-precision twins (handed over as launch thunks)
//   rmb_symx_coop.hip  workgroup-cooperative instances of the generic symmetric skeleton (launch thunks too)
//   rmb_symx2t.hip, rmb_symx2t_per.hip  two-targets-per-lane instances of the generic skeleton, open / pseudo-periodic
//   rmb_sort.hip     Morton ordering of the blobs for the force kernel's tile culling (rocPRIM radix sort)
//   rmb_sweep.hip    launchers of the one-sided kernels: sweep, force sweep, source->target, pressure / double layer,
//                    dense body blocks, position packing
//   rmb_entry.hip    the extern "C" products: argument checks, routing between the two families, host staging
//   rmb_multi.hip    the single-process multi-device engine (rmb_multi_*)
//   rmb_rigid.hip    per-body geometry (positions, K) and the per-body factors of the block-diagonal preconditioner
//   rmb_krylov.hip   O(N) helpers of the rigid-body solve: batched 2 x 2 block product, fused Gram-Schmidt step
//   rmb_gmres.hip    the whole right-preconditioned GMRES of the rigid-body problem as one call (host loop native too)
#pragma once
#include "../../include/rmb_mobility.h"

#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <vector>

#include "pair_ops.h"

namespace rmbi {

// ---- errors ------------------------------------------------------------------------------------------------
int fail(int code, const std::string& msg);   // stores the message for rmb_last_error() of this thread, returns code

#define RMB_HIP(call)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return rmbi::fail(RMB_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));     \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes);
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

constexpr int kTimingRing = 8192;

}  // namespace rmbi

struct rmb_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t stream_switch = nullptr;  // orders a newly set stream after the work queued on the previous one
  // device properties (hipDeviceProp_t): a partitioned (CPX) device or another SKU changes both
  long n_cu = 256;               // multiProcessorCount
  size_t lds_per_cu = 160 * 1024;  // maxSharedMemoryPerMultiProcessor
  // resident configuration
  long n = 0;
  double a = 0.0;
  double L[3] = {0, 0, 0};
  int wall = 0;
  bool have_positions = false;
  long tgt_begin = 0, tgt_end = 0;
  // device memory
  rmbi::DevBuf pos;      // double4[n]
  rmbi::DevBuf r_stage;  // raw positions staging (host entry)
  rmbi::DevBuf vec, vec2, out, partial, tmp3n;
  rmbi::DevBuf tile_bounds;      // bounding boxes of the 64-blob tiles (force kernel's tile culling); valid for the packed positions
  bool tile_bounds_valid = false;
  long opt_force_cull = 1;       // blob-blob forces: skip tile pairs beyond the range of the exponential (bit-exact)
  // spatially sorted copy of the configuration for the force kernel (rmb_sort.hip): valid together with tile_bounds
  rmbi::DevBuf fpos, fperm, fsort_keys, fsort_vals, fsort_tmp, fsort_box;
  bool force_sorted = false;     // tile_bounds / fpos / fperm describe the SORTED configuration
  long opt_force_sort = 1;       // sort the blobs along a Morton curve for the force kernel's tile culling
  rmbi::DevBuf det_ws;           // per-unit partials of the deterministic symmetric pass
  long opt_det_workspace_mb = 8192;   // cap on the partial-result workspace of deterministic = 2 (symx_det_device)
  rmbi::DevBuf st[8];    // scratch of the source->target entry point
  rmbi::DevBuf wave_clock;  // optional per-wave (start, end) wall-clock stamps of the symmetric kernel
  long wave_clock_n = 0;
  long opt_wave_clock = 0;
  long opt_skip_pairs = 0;
  rmbi::DevBuf krylov;   // partial sums of rmb_krylov_orthogonalize_device
  void* gmres_ws = nullptr;   // workspace of rmb_rigid_gmres_device (rmb_gmres.hip), freed by gmres_release
  rmbi::DevBuf symbuf;   // acc[3][n_pad] doubles for the symmetric tt kernel (kept zero between calls)
  // result hand-off of the synchronous host entry point (rmb_matvec): page-locked, device-mapped host memory the finalize
  // kernel stores into directly (coalesced), for results up to opt_host_zero_copy bytes
  void* host_out = nullptr;
  double* host_out_dev = nullptr;
  size_t host_out_cap = 0;
  void* host_in = nullptr;       // ... and the same for its input vectors (two of them: RMB_TT_TR)
  double* host_in_dev = nullptr;
  size_t host_in_cap = 0;
  long opt_lanczos_fuse_finish = 1; // rmb_rigid_lanczos_step_device: finalize of the sweep + L_b^-1 product in one launch
  long opt_krylov_low_sync = 1;     // native GMRES / Lanczos steps: second update + norm (by Pythagoras) + normalisation in one launch
  long opt_gmres_fuse_dots = 1;     // rmb_rigid_gmres_device: the operator's finishing launch also takes the first Gram-Schmidt dots (<= 256 bodies)
  long opt_gmres_fuse_pc = 1;       // rmb_rigid_gmres_device: the normalisation launch also applies the preconditioner for the next step
  long opt_host_zero_copy_in = 1;   // inputs of rmb_matvec through mapped memory + a pull kernel (sizes as host_zero_copy)
  long opt_host_zero_copy = 768 << 10;   // bytes (32 768 blobs: level at 43 000, +1 % at 1e5); 0 = always a device-to-host copy command
  long symbuf_zeroed_for = -1;
  // options
  long opt_chunks = 0;
  long opt_timing = 0;
  long opt_symmetric = 1;      // use the symmetric (each unordered pair once) kernel where applicable
  long opt_fused_symmetric = 1;  // tt+tr: 1 = single symmetric pass (symx_kernels.h), 2 = two symmetric passes, 0 = one-sided fused sweep
  long opt_symx_single = 0;      // route tt / tr / rt / rr through the generic skeleton (A/B against sym_kernel)
  long opt_deterministic = 0;  // force the atomic-free sweep kernel everywhere
  int last_path = 0;           // 0 = sweep, 1 = symmetric (per wave), 2 = deterministic symmetric, 3 = symmetric, workgroup-cooperative
  long opt_sym_wps = 0;        // cap on resident workgroups per CU for the symmetric kernel (0 = occupancy limit)
  long opt_sym_pin = 1;        // pad dynamic LDS so residency is exactly that number
  long opt_precision = 64;     // 32: M_tt f (open boundaries) in single precision (sym32_kernels.h); everything else fp64
  long opt_force_precision = 0;  // blob-blob forces: 0 = follow "precision", 32 / 64 = pinned
  long opt_sym_min_steps = 64; // floor on rotation steps per wave (a unit is 64 steps)
  long opt_sym_fine_steps = 0;   // floor on steps per wave when less than one resident round is left (pair shards, small N); 0 = 16 or 32, chosen in plan_sym
  long opt_sym_coop = 1;       // workgroup-cooperative symmetric kernel (sym_coop_kernels.h): 0 = never, 1 = launches of at most
                               // kCoopMaxRounds resident rounds (small suspensions, pair shards, up to ~1e4 blobs), 2 = always
  long opt_sym_chunk_steps = 1024;  // symmetric kernels: a wave's steps are cut into strided chunks of about this many (0 = one range)
  long opt_sym_two_targets = 1;   // sym2t_kernel (two target blobs per lane) for tt / tr / rt / rr, open boundaries: 0 off, 1 from one resident round on, 2 always
  long opt_sym_order = 1;      // unit order of the symmetric kernels: 1 = blocked (32 x 32 tile super-blocks), 0 = row-major
  long opt_sym_xcd = 1;        // XCD-aware workgroup numbering (each XCD a contiguous eighth of the step range)
  long opt_sym_oversub = 8;    // launch this many times the resident workgroup count (measured: -4..8 % kernel time;
                               // waves of one SIMD finish oldest-first, more rounds keep every SIMD at >= 3 active waves)
  // timing ring (events around the sweep kernel)
  std::vector<hipEvent_t> ev0, ev1;
  int ev_count = 0;  // events recorded since last reset (capped at ring size)
  long timing_launches = 0;  // sweeps seen since the last reset (sampling stride of the "timing" option)
  // last launch
  long last_tiles = 0, last_chunks = 0, last_wgs = 0;
  double host_us[4] = {0, 0, 0, 0};   // last rmb_matvec: upload, launch, wait + download, whole call (host wall clock, us)
};

namespace rmbi {

// ---- rmb_context.hip ---------------------------------------------------------------------------------------
int timing_begin(rmb_ctx* c, int* slot);
int timing_end(rmb_ctx* c, int slot);
int check_ready(rmb_ctx* c);
// the library's default context (stateless entry points); created on first use on the device RMB_DEVICE names (0)
extern std::mutex g_default_mu;
int default_ctx(rmb_ctx** out);   // call with g_default_mu held

// ---- rmb_gmres.hip ----------------------------------------------------------------------------------------
void gmres_release(rmb_ctx* c);
// ---- rmb_krylov.hip / rmb_rigid.hip: pieces the native GMRES composes ------------------------------------------------
// Per-body blocks (the preconditioner's four of the saddle-point system, or L_b^-T alone for the Lanczos forcing: r2 = 0,
// absent blocks with p = nullptr) and where z = blocks * v goes: handed to the Gram-Schmidt step, its LAST launch
// (normalisation, workgroup = body instead of chunk) also applies the blocks to the vector it has just normalised -- the
// next iteration's first launch.  The vector is laid out as [n_bodies x r1; n_bodies x r2].
struct BlockRef { const double* p; long bs, rs, cs; };       // one batched block: entry b at p + b * bs, element (r, c) at + r * rs + c * cs
struct PcBlocks { long n_bodies, r1, r2; BlockRef a11, a12, a21, a22; double* z; };      // square: r1 x r1, r1 x r2, r2 x r1, r2 x r2
// part1_bodies > 0: the first pass's partial dots are already there, one per body and basis row (krylov_body_partials'
// buffer, written by the operator's finishing launch): the step starts with the first update launch
int krylov_orthogonalize_impl(rmb_ctx* c, long n, long rows, const double* V_dev, long ldv, double* w_dev, double* col_dev,
                              double* v_next_dev, double* col_mapped_dev, const PcBlocks* pc, long part1_bodies = 0, bool low_sync = false);
constexpr long kKrBodyPartialsMax = 256;      // bodies up to which the finishing launch takes the first dots (every update workgroup re-sums them)
int krylov_body_partials(rmb_ctx* c, long n, double** part_out);
// the basis and where the partial dots of the vector the operator has just produced go: part[r * n_bodies + body]
struct DotsFuse { const double* V; long ldv, rows; double* part; };
int plain_tt_with_dots(rmb_ctx* c, const double* v_dev, double eta, double* out_dev, const double* V_dev, long ldv, long rows, long* tiles_done);
int rigid_operator_impl(rmb_ctx* c, long n_bodies, long n_b, const double* K_dev, const double* x_dev, double eta, double* out_dev,
                        const DotsFuse* dots, bool* dots_done);
// z_ready: z_dev already holds P^-1 v_j (the previous step's fused launch); fuse_pc: leave P^-1 v_{j+1} in z_dev
// One step of the preconditioned Lanczos forcing.  pv: P v_i (input of the sweep; pv_ready: left there by the previous step's
// fused launch), mw: the sweep's result, d: P^T M P v_i, the vector that is orthogonalised (d may be pv when fuse_next is
// false); fuse_next: the normalisation launch also leaves P v_{i+1} in pv.
int lanczos_step_impl(rmb_ctx* c, long n_bodies, long n_b, const double* Linv_dev, double* V_dev, long ldv, long i, double eta, double* pv_dev,
                      double* mw_dev, double* d_dev, double* col_dev, double* col_mapped_dev, bool pv_ready, bool fuse_next);
int arnoldi_step_impl(rmb_ctx* c, long n_bodies, long n_b, const double* A11_dev, const double* A12_dev, const double* A21_dev,
                      const double* A22_dev, const double* K_dev, double* V_dev, long ldv, long j, double eta, double* z_dev, double* w_dev,
                      double* col_dev, double* col_mapped_dev, bool z_ready, bool fuse_pc);

// ---- rmb_plan.hip ------------------------------------------------------------------------------------------
rmb::PairConsts make_pair_consts(double a);
rmb::ExpConsts exp_consts();
void choose_chunks(long n_tgt, long n_src, long forced, long slots, long* n_chunks, long* chunk_len);
int resident_blocks(const void* fn, int* cache);   // workgroups of 256 threads per CU, capped at 8
// Launch plan of a symmetric sweep: `total` rotation steps over `blocks` workgroups of 4 waves.
struct SymPlan { long blocks; long steps_per_wave; size_t dyn_lds; bool sub_round; long round; };   // round: resident workgroups; sub_round: less work than that
int plan_sym(rmb_ctx* c, const void* fn, int* occ_cache, size_t static_lds, long total, bool pin, SymPlan* out,
             int declared_waves = 0, long fine_auto = 0);
void shard_ranges(long n, long n_units, long shard, long nshards, long* step_begin, long* step_end, long* self_begin,
                  long* self_end);
int sym_accumulators(rmb_ctx* c, long n_pad);
// Steps per strided chunk: `spw` = steps per schedule unit (wave, or workgroup of the cooperative kernels) of a plan over
// `n_sched` units; cut into R >= 1 equal chunks of about `target` steps (exactly balanced: every unit gets R chunks).
long chunked_steps(const rmb_ctx* c, long total, long n_sched, long spw, long target);
// whether the symmetric (each unordered pair once) path applies to the resident configuration
bool sym_applies(const rmb_ctx* c);

// ---- rmb_sym.hip -------------------------------------------------------------------------------------------
// SX_K2 + 4 (k - 2) + kind: one block on k = 2..4 vectors
enum SymXOp { SX_TT = 0, SX_TR, SX_RT, SX_RR, SX_FUSED, SX_GRAND, SX_COLF, SX_FREE, SX_RADII, SX_K2, SX_COUNT = SX_K2 + 12 };
// Configuration a symmetric pass runs on: the context's resident one, or a caller-packed one (per-blob radii)
struct SymConf { const double4* pos; long n; double L[3]; int wall; const double* extra; };
// no_finalize: leave the raw sums in the accumulators (c->symbuf: [3][n_pad], unscaled, no self term) -- the caller
// launches its own finishing kernel (rmb_rigid_operator_device); not for the pseudo-periodic / fp32 routes
int sym_device(rmb_ctx* c, int kind, const double* v, double eta, double* out, long shard = 0, long nshards = 1,
               bool accumulate = false, bool no_finalize = false);
int sym2_device(rmb_ctx* c, const double* va, const double* vb, double eta, double* out_a, double* out_b, long shard = 0,
                long nshards = 1);
int symx_device(rmb_ctx* c, int op, const double* const* in, double* const* out, double eta, int in_plane, long shard,
                long nshards, int accumulate_mask = 0, const SymConf* conf_in = nullptr);
int symx_det_device(rmb_ctx* c, int op, const double* const* in, double* const* out, double eta, int in_plane,
                    long shard = 0, long nshards = 1);
int sym_force_device(rmb_ctx* c, double eps, double b, double blob_radius, double* out, const double* radii, long shard,
                     long nshards);

// ---- rmb_sort.hip ------------------------------------------------------------------------------------------
int force_sort_positions(rmb_ctx* c);

// ---- rmb_sym32.hip: single-precision twins as launch thunks ---------------------------------------------------
// fn = host handle of the kernel (occupancy / attributes), nullptr when the operation has no fp32 twin;
// launch() converts the kernel-uniform constants to float and enqueues; `args` points to the fp64 kernel's argument
// struct (rmb::SymArgs / rmb::SymXArgs / rmb::SymForceArgs).
struct Kernel32 {
  const void* fn;
  size_t static_lds;
  int* occ;
  void (*launch)(const void* args, const rmb::PairConsts& k, unsigned blocks, size_t dyn_lds, hipStream_t s);
};
Kernel32 sym32_tt(bool wall);
Kernel32 symx32(int sx, bool wall);
Kernel32 sym_force32(bool radii);
// ---- rmb_symx_coop.hip: workgroup-cooperative instances of the generic skeleton, same thunk shape (fp64) ---------
Kernel32 symx_coop(int sx, bool wall, bool periodic);
// ---- rmb_symx2t.hip / rmb_symx2t_per.hip: two-targets-per-lane instances of the generic skeleton (symx2t_kernels.h) ------
// fn == nullptr when the operation has none; *waves_per_eu = what the instance was compiled for (plan_sym's residency cap)
Kernel32 symx_two_open(int sx, bool wall, int* waves_per_eu);
Kernel32 symx_two_periodic(int sx, bool wall, int* waves_per_eu);

// ---- rmb_sweep.hip -----------------------------------------------------------------------------------------
int pack_positions(rmb_ctx* c, const double* r_dev, long n, double a, const double* L, int wall);
int pack_positions_radii(rmb_ctx* c, const double* r_dev, const double* rad_dev, long n, int wall, double4* dst);
int sweep_device(rmb_ctx* c, int kind, int in_plane, const double* v, const double* v2, double eta, double* out);
int force_sweep_device(rmb_ctx* c, double eps, double b, double blob_radius, double* out, const double* radii);
int add_inplace(rmb_ctx* c, double* y, const double* x, long n);
int pull_mapped(rmb_ctx* c, double* dst_dev, const double* src_mapped_dev, long n);
int body_dense_device(rmb_ctx* c, const long* first_blob_dev, long n_bodies, int n_b, double eta, double* out_dev);
int st_sweep_device(rmb_ctx* c, long ns, const double4* src_packed, const double* rad_s, const double* force, long nt,
                    const double4* tgt_packed, const double* rad_t, double eta, const double* L, int wall, double* out);
int pressure_device(rmb_ctx* c, long ns, const double* src, long nt, const double* tgt, const double* force, int wall,
                    double* out);
int double_layer_device(rmb_ctx* c, long ns, const double* src, long nt, const double* tgt, const double* normals,
                        const double* vector, const double* weights, int wall, double blob_radius, double* out);
int ubench_fp64_issue(rmb_ctx* c, int launches, double* g_wave_instr_per_s);

// ---- rmb_entry.hip (used by the multi-device engine too) ------------------------------------------------------
int matvec_device_impl(rmb_ctx* c, int kind, int in_plane, const double* v, const double* v2, double eta, double* out);
int matvec_pairshard_impl(rmb_ctx* c, int kind, int in_plane, const double* v, double eta, double* out, long shard, long nshards);
int matvec_op_impl(rmb_ctx* c, int op, int in_plane, int n_in, const double* const* in, int n_out, double* const* out,
                   double eta, long shard, long nshards);
int force_device_impl(rmb_ctx* c, double eps, double b, double blob_radius, double* out, const double* radii = nullptr,
                      long shard = 0, long nshards = 1);

}  // namespace rmbi
