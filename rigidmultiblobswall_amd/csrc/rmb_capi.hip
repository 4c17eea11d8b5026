// rmb_capi.hip -- C ABI (include/rmb_mobility.h) over the gfx950 sweep kernels.
// Host side: context with resident packed positions, workspace, launch geometry, HIP-event timing.
#include "../../include/rmb_mobility.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "matvec_kernels.h"
#include "sym_kernels.h"
#include "sym2_kernels.h"
#include "sym32_kernels.h"
#include "symx_kernels.h"
#include "symx32_kernels.h"
#include "dense_kernels.h"
#include "st_kernels.h"
#include "aux_kernels.h"
#include "diag_kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define RMB_HIP(call)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(RMB_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));           \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { p = nullptr; return fail(RMB_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    cap = want;
    return 0;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

constexpr int kTimingRing = 8192;

}  // namespace

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
  DevBuf pos;      // double4[n]
  DevBuf r_stage;  // raw positions staging (host entry)
  DevBuf vec, vec2, out, partial, tmp3n;
  DevBuf tile_bounds;            // bounding boxes of the 64-blob tiles (force kernel's tile culling); valid for the packed positions
  bool tile_bounds_valid = false;
  long opt_force_cull = 1;       // blob-blob forces: skip tile pairs beyond the range of the exponential (bit-exact)
  DevBuf det_ws;                 // per-unit partials of the deterministic symmetric pass
  long opt_det_workspace_mb = 8192;   // cap on the partial-result workspace of deterministic = 2 (symx_det_device)
  DevBuf st[8];    // scratch of the source->target entry point
  DevBuf wave_clock;  // optional per-wave (start, end) wall-clock stamps of the symmetric kernel
  long wave_clock_n = 0;
  long opt_wave_clock = 0;
  long opt_skip_pairs = 0;
  DevBuf symbuf;   // acc[3][n_pad] doubles for the symmetric tt kernel (kept zero between calls)
  long symbuf_zeroed_for = -1;
  // options
  long opt_chunks = 0;
  long opt_timing = 0;
  long opt_symmetric = 1;      // use the symmetric (each unordered pair once) kernel where applicable
  long opt_fused_symmetric = 1;  // tt+tr: 1 = single symmetric pass (symx_kernels.h), 2 = two symmetric passes, 0 = one-sided fused sweep
  long opt_symx_single = 0;      // route tt / tr / rt / rr through the generic skeleton (A/B against sym_kernel)
  long opt_deterministic = 0;  // force the atomic-free sweep kernel everywhere
  int last_path = 0;           // 0 = sweep, 1 = symmetric
  long opt_sym_wps = 0;        // cap on resident workgroups per CU for the symmetric kernel (0 = occupancy limit)
  long opt_sym_pin = 1;        // pad dynamic LDS so residency is exactly that number
  long opt_precision = 64;     // 32: M_tt f (open boundaries) in single precision (sym32_kernels.h); everything else fp64
  long opt_force_precision = 0;  // blob-blob forces: 0 = follow "precision", 32 / 64 = pinned
  long opt_sym_min_steps = 64; // floor on rotation steps per wave (a unit is 64 steps)
  long opt_sym_fine_steps = 0;   // floor on steps per wave when less than one resident round is left (pair shards, small N); 0 = 16 or 32, chosen in plan_sym
  long opt_sym_oversub = 8;    // launch this many times the resident workgroup count (measured: -4..8 % kernel time;
                               // waves of one SIMD finish oldest-first, more rounds keep every SIMD at >= 3 active waves)
  // timing ring (events around the sweep kernel)
  std::vector<hipEvent_t> ev0, ev1;
  int ev_count = 0;  // events recorded since last reset (capped at ring size)
  long timing_launches = 0;  // sweeps seen since the last reset (sampling stride of the "timing" option)
  // last launch
  long last_tiles = 0, last_chunks = 0, last_wgs = 0;
};

namespace {

rmb::PairConsts make_pair_consts(double a) {
  rmb::PairConsts k;
  const double a2 = a * a, a3 = a2 * a, a4 = a2 * a2, a6 = a3 * a3;
  k.a2 = a2;
  k.four_a2 = 4.0 * a2;
  k.tt_k1 = 2.0 * a2 / 3.0;
  k.tt_k2 = 2.0 * a2;
  k.tt_k3 = a2 / 3.0;
  k.tt_n0 = 4.0 / (3.0 * a);
  k.tt_n1 = 3.0 / (8.0 * a2);
  k.tt_n2 = 1.0 / (8.0 * a2);
  k.rr_m0 = 1.0 / a3;
  k.rr_m1 = 27.0 / (32.0 * a4);
  k.rr_m2 = 5.0 / (64.0 * a6);
  k.rr_m3 = 9.0 / (32.0 * a4);
  k.rr_m4 = 3.0 / (64.0 * a6);
  k.c_q0 = 1.0 / (2.0 * a3);
  k.c_q1 = 3.0 / (16.0 * a4);
  k.m7 = -7.0;
  k.m6 = -6.0;
  k.c15 = 1.5;
  k.c30 = 30.0;
  return k;
}

// Source-chunk count.  A workgroup is 4 waves (one per SIMD); `slots` = 256 CUs x resident
// workgroups per CU for this kernel.  Either everything is resident at once in one balanced round
// (tiles*c just under `slots`), or there are enough rounds (>= 6) that the tail is small.
void choose_chunks(long n_tgt, long n_src, long forced, long slots, long* n_chunks, long* chunk_len) {
  const long tiles = (n_tgt + 63) / 64;
  long c = 1;
  if (forced > 0) {
    c = forced;
  } else if (tiles <= slots) {
    c = slots / tiles;
  } else if (tiles < 6 * slots) {
    c = (6 * slots + tiles - 1) / tiles;
  }
  const long max_chunks = (n_src + 127) / 128;  // >= 32 sources per wave
  if (c > max_chunks) c = max_chunks;
  if (c < 1) c = 1;
  long len = (n_src + c - 1) / c;
  len = ((len + rmb::kWaves - 1) / rmb::kWaves) * rmb::kWaves;
  c = (n_src + len - 1) / len;
  *n_chunks = c;
  *chunk_len = len;
}

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

int resident_blocks(const void* fn, int* cache) {
  if (*cache > 0) return *cache;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, rmb::kBlock, 0) != hipSuccess || nb < 1) nb = 4;
  if (nb > 8) nb = 8;
  if (getenv("RMB_DEBUG_OCC")) {
    hipFuncAttributes at;
    if (hipFuncGetAttributes(&at, fn) == hipSuccess)
      fprintf(stderr, "[rmb] occupancy api %d blocks/CU | numRegs %d sharedSizeBytes %zu maxThreadsPerBlock %d localSizeBytes %zu\n", nb,
              at.numRegs, (size_t)at.sharedSizeBytes, at.maxThreadsPerBlock, (size_t)at.localSizeBytes);
  }
  *cache = nb;
  return nb;
}

int timing_begin(rmb_ctx* c, int* slot) {
  *slot = -1;
  if (!c->opt_timing) return 0;
  // "timing" = n > 1: bracket every n-th sweep only.  An event pair costs ~4-8 us of serialisation around a launch
  // (tools/exp_graph.py: 188 us per 1e4-blob step without events, 199 us with), so a throughput measurement samples.
  if (c->opt_timing > 1 && (c->timing_launches++ % c->opt_timing) != 0) return 0;
  if (c->ev0.empty()) {
    c->ev0.resize(kTimingRing);
    c->ev1.resize(kTimingRing);
    for (int i = 0; i < kTimingRing; ++i) {
      RMB_HIP(hipEventCreate(&c->ev0[i]));
      RMB_HIP(hipEventCreate(&c->ev1[i]));
    }
  }
  *slot = c->ev_count % kTimingRing;
  RMB_HIP(hipEventRecord(c->ev0[*slot], c->stream));
  return 0;
}

int timing_end(rmb_ctx* c, int slot) {
  if (slot < 0) return 0;
  RMB_HIP(hipEventRecord(c->ev1[slot], c->stream));
  c->ev_count++;
  return 0;
}

int check_ready(rmb_ctx* c) {
  if (!c) return fail(RMB_ERR_ARG, "null context");
  if (!c->have_positions) return fail(RMB_ERR_STATE, "rmb_set_positions has not been called");
  return 0;
}

typedef void (*sym_fn)(const rmb::SymArgs);
struct SymEntry { sym_fn sweep; sym_fn fin; int occ; };
template <int KIND, bool WALL, bool PER> SymEntry make_sym_entry() {
  return SymEntry{rmb::sym_kernel<KIND, WALL, PER>, rmb::sym_finalize_kernel<KIND, WALL>, 0};
}
// [kind tt,tr,rt,rr][wall][periodic]
#define RMB_SYM_ROW(K) {{make_sym_entry<K, false, false>(), make_sym_entry<K, false, true>()}, {make_sym_entry<K, true, false>(), make_sym_entry<K, true, true>()}}
SymEntry g_sym[4][2][2] = {RMB_SYM_ROW(rmb::KIND_TT), RMB_SYM_ROW(rmb::KIND_TR), RMB_SYM_ROW(rmb::KIND_RT), RMB_SYM_ROW(rmb::KIND_RR)};
#undef RMB_SYM_ROW

// Global SoA accumulators of the symmetric kernels: [4][3][n_pad] doubles (up to four output vectors per pass),
// zeroed once; every finalize kernel re-zeroes what its sweep touched.
constexpr int kSymMaxOut = 4;
int sym_accumulators(rmb_ctx* c, long n_pad) {
  const size_t acc_bytes = (size_t)3 * kSymMaxOut * n_pad * sizeof(double);
  if (acc_bytes > c->symbuf.cap || c->symbuf_zeroed_for != n_pad) {
    if (int rc = c->symbuf.reserve(acc_bytes)) return rc;
    RMB_HIP(hipMemsetAsync(c->symbuf.p, 0, acc_bytes, c->stream));
    c->symbuf_zeroed_for = n_pad;
  }
  return 0;
}

rmb::f32::PairConsts pair_consts32(const rmb::PairConsts& k) {
  rmb::f32::PairConsts f;
  f.a2 = (float)k.a2; f.four_a2 = (float)k.four_a2; f.tt_k1 = (float)k.tt_k1; f.tt_k2 = (float)k.tt_k2; f.tt_k3 = (float)k.tt_k3;
  f.tt_n0 = (float)k.tt_n0; f.tt_n1 = (float)k.tt_n1; f.tt_n2 = (float)k.tt_n2;
  f.rr_m0 = (float)k.rr_m0; f.rr_m1 = (float)k.rr_m1; f.rr_m2 = (float)k.rr_m2; f.rr_m3 = (float)k.rr_m3; f.rr_m4 = (float)k.rr_m4;
  f.c_q0 = (float)k.c_q0; f.c_q1 = (float)k.c_q1; f.m7 = (float)k.m7; f.m6 = (float)k.m6; f.c15 = (float)k.c15; f.c30 = (float)k.c30;
  return f;
}

// Launch plan of a symmetric sweep: `total` rotation steps over `blocks` workgroups of 4 waves.
struct SymPlan { long blocks; long steps_per_wave; size_t dyn_lds; };

// Static, exactly balanced schedule (sym_kernels.h): whole multiples of the resident workgroup count so that every
// SIMD gets the same number of steps.  `pin` pads dynamic LDS so that exactly `wps` workgroups fit a CU (equal steps
// per wave is then equal work per SIMD); CU count and LDS size come from hipDeviceProp_t (rmb_ctx_create).
// `declared_waves`: the kernel's amdgpu_waves_per_eu bound (0 = none).  The occupancy API prices a kernel by its
// ARCHITECTURAL VGPRs only (hipFuncAttributes::numRegs); a kernel compiled under waves_per_eu(4, 4) parks values in
// AGPRs up to the 128-register budget (sym_kernel<TT>: 86 + 9 -> allocates 104, <RR>: 72 + 25), so the API reports 5
// and 7 workgroups per CU where the hardware holds 4 (per-wave start stamps, profiles/r3_shard_wave_placement.txt) and
// every plan built on "whole resident rounds" was off after the kernels lost registers in round 2.
int plan_sym(rmb_ctx* c, const void* fn, int* occ_cache, size_t static_lds, long total, bool pin, SymPlan* out,
             int declared_waves = 0) {
  int wps = resident_blocks(fn, occ_cache);
  if (declared_waves > 0 && wps > declared_waves) wps = declared_waves;
  if (c->opt_sym_wps > 0 && c->opt_sym_wps < wps) wps = (int)c->opt_sym_wps;
  size_t pad = 0;
  if (pin && c->opt_sym_pin) {
    const size_t per_block = c->lds_per_cu / (size_t)wps;
    if (per_block > static_lds + 1024) pad = per_block - static_lds - 512;
    if (pad > 48 * 1024) RMB_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad));
  }
  const long round = c->n_cu * wps;
  long blocks = round * c->opt_sym_oversub;
  const long per_wg = rmb::kSymWaves * c->opt_sym_min_steps;
  const long need = (total + per_wg - 1) / per_wg > 0 ? (total + per_wg - 1) / per_wg : 1;
  if (blocks > need) blocks = need;
  if (blocks > round) blocks -= blocks % round;   // whole rounds only: a partial last round is a tail
  if (blocks < round) {
    // Less than one resident round at 64 steps per wave (small N, or one rank's pair shard): shorter waves beat
    // leaving SIMDs with one or two waves and no latency hiding, but every wave pays its own loads and 384 global
    // atomics on accumulators it shares with the other waves of its tile row.
    // Two regimes (tools/exp_small_n.py, tools/exp_shard_plan.py; profiles/r3_shard_plan.txt): while 16-step waves
    // do not fill the chip (small N: <= 1500 blobs) the launch is latency-bound and more, shorter waves win
    // (1000 blobs: 9.6 us at 16 steps, 13.0 at 32); once they would overfill it, every extra wave only adds its
    // loads and flushes (1/8 shard of 1e4 blobs: 36.6 us at 16 steps x 1024 workgroups, 29.3 at 32 x 776).
    long fine_steps = c->opt_sym_fine_steps;
    if (fine_steps <= 0) fine_steps = (total + rmb::kSymWaves * 16L - 1) / (rmb::kSymWaves * 16L) <= round ? 16 : 32;
    const long per_wg_fine = rmb::kSymWaves * fine_steps;
    long fine = (total + per_wg_fine - 1) / per_wg_fine;
    if (fine > round) fine = round;
    if (fine > blocks) blocks = fine;
  }
  if (blocks < 1) blocks = 1;
  const long waves = blocks * rmb::kSymWaves;
  out->blocks = blocks;
  out->steps_per_wave = (total + waves - 1) / waves;
  out->dyn_lds = pad;
  return 0;
}

// step range and self-term ownership of pair shard `shard` of `nshards`
void shard_ranges(long n, long n_units, long shard, long nshards, long* step_begin, long* step_end, long* self_begin,
                  long* self_end) {
  const __int128 s_total = (__int128)n_units * 64;
  *step_begin = (long)(s_total * shard / nshards);
  *step_end = (long)(s_total * (shard + 1) / nshards);
  const long block = (n + nshards - 1) / nshards;       // same block partition as distributed.partition()
  *self_begin = block * shard < n ? block * shard : n;
  *self_end = block * (shard + 1) < n ? block * (shard + 1) : n;
}

int sym_device(rmb_ctx* c, int kind, const double* v, double eta, double* out, long shard = 0, long nshards = 1,
               bool accumulate = false) {
  const bool periodic = c->L[0] > 0 || c->L[1] > 0 || c->L[2] > 0;
  SymEntry& se = g_sym[kind][c->wall ? 1 : 0][periodic ? 1 : 0];
  const long n = c->n;
  const long tiles = (n + 63) / 64;
  const long n_pad = 64 * tiles;
  if (int rc = sym_accumulators(c, n_pad)) return rc;
  rmb::SymArgs a;
  a.pos = (const double4*)c->pos.p;
  a.vec = v;
  a.acc = (double*)c->symbuf.p;
  a.out = out;
  a.n = n;
  a.n_pad = n_pad;
  a.n_tiles = (int)tiles;
  a.n_units = tiles * (tiles + 1) / 2;
  shard_ranges(n, a.n_units, shard, nshards, &a.step_begin, &a.step_end, &a.self_begin, &a.self_end);
  a.Lx = c->L[0]; a.Ly = c->L[1]; a.Lz = c->L[2];
  a.iLx = c->L[0] > 0 ? 1.0 / c->L[0] : 0.0;
  a.iLy = c->L[1] > 0 ? 1.0 / c->L[1] : 0.0;
  a.iLz = c->L[2] > 0 ? 1.0 / c->L[2] : 0.0;
  a.prefactor = 1.0 / (8.0 * M_PI * eta);
  a.k = make_pair_consts(c->a);
  SymPlan plan;
  // single-precision mode (mobility_pycuda.py:7-19 `precision = 'single'`): tt with open boundaries only
  const bool f32 = c->opt_precision == 32 && kind == RMB_TT && !periodic;
  if (f32 && (c->opt_wave_clock || c->opt_skip_pairs))
    return fail(RMB_ERR_STATE, "the \"wave_clock\" / \"skip_pairs\" diagnostics exist in the fp64 kernels only: set \"precision\" = 64");
  typedef void (*sym32_fn)(const rmb::SymArgs, const rmb::f32::PairConsts);
  const sym32_fn fn32 = c->wall ? (sym32_fn)rmb::sym32_tt_kernel<true> : (sym32_fn)rmb::sym32_tt_kernel<false>;
  static int occ32[2] = {0, 0};
  const size_t stat = f32 ? (sizeof(float) * 9 + sizeof(double) * 3) * rmb::kSymWaves * 64
                          : sizeof(double2) * rmb::kSymWaves * 64 * 3 + sizeof(double) * rmb::kSymWaves * 3 * 64;
  if (int rc = plan_sym(c, f32 ? (const void*)fn32 : (const void*)se.sweep, f32 ? &occ32[c->wall ? 1 : 0] : &se.occ, stat,
                        a.step_end - a.step_begin, true, &plan, f32 ? 0 : rmb::kSymWavesPerEu))
    return rc;
  const long blocks = plan.blocks;
  a.steps_per_wave = plan.steps_per_wave;
  c->last_tiles = tiles; c->last_chunks = 0; c->last_wgs = blocks;
  a.skip_pairs = (int)c->opt_skip_pairs;
  a.accumulate = accumulate ? 1 : 0;
  a.wave_clock = nullptr;
  if (c->opt_wave_clock) {
    c->wave_clock_n = blocks * rmb::kSymWaves;
    if (int rc = c->wave_clock.reserve((size_t)2 * c->wave_clock_n * sizeof(long long))) return rc;
    a.wave_clock = (long long*)c->wave_clock.p;
  }
  int slot;
  if (int rc = timing_begin(c, &slot)) return rc;
  if (f32) {
    const rmb::f32::PairConsts kf = pair_consts32(a.k);
    hipLaunchKernelGGL(fn32, dim3((unsigned)blocks), dim3(64 * rmb::kSymWaves), plan.dyn_lds, c->stream, a, kf);
  } else {
    hipLaunchKernelGGL(se.sweep, dim3((unsigned)blocks), dim3(64 * rmb::kSymWaves), plan.dyn_lds, c->stream, a);
  }
  RMB_HIP(hipGetLastError());
  if (int rc = timing_end(c, slot)) return rc;
  const dim3 fgrid((unsigned)((n + 255) / 256));
  hipLaunchKernelGGL(se.fin, fgrid, dim3(256), 0, c->stream, a);
  RMB_HIP(hipGetLastError());
  return 0;
}

// Two source vectors, one pass over the unordered pairs (sym2_kernels.h).  Same schedule rules as sym_device.
int sym2_device(rmb_ctx* c, const double* va, const double* vb, double eta, double* out_a, double* out_b, long shard = 0,
                long nshards = 1) {
  const bool periodic = c->L[0] > 0 || c->L[1] > 0 || c->L[2] > 0;
  const long n = c->n, tiles = (n + 63) / 64, n_pad = 64 * tiles;
  if (int rc = sym_accumulators(c, n_pad)) return rc;
  rmb::Sym2Args a;
  a.pos = (const double4*)c->pos.p;
  a.vec_a = va; a.vec_b = vb;
  a.acc = (double*)c->symbuf.p;
  a.out_a = out_a; a.out_b = out_b;
  a.n = n; a.n_pad = n_pad; a.n_tiles = (int)tiles; a.n_units = tiles * (tiles + 1) / 2;
  shard_ranges(n, a.n_units, shard, nshards, &a.step_begin, &a.step_end, &a.self_begin, &a.self_end);
  a.Lx = c->L[0]; a.Ly = c->L[1]; a.Lz = c->L[2];
  a.iLx = c->L[0] > 0 ? 1.0 / c->L[0] : 0.0;
  a.iLy = c->L[1] > 0 ? 1.0 / c->L[1] : 0.0;
  a.iLz = c->L[2] > 0 ? 1.0 / c->L[2] : 0.0;
  a.prefactor = 1.0 / (8.0 * M_PI * eta);
  a.k = make_pair_consts(c->a);
  typedef void (*k2_fn)(const rmb::Sym2Args);
  static int occ2[2][2] = {{0, 0}, {0, 0}};
  const k2_fn fn = c->wall ? (periodic ? (k2_fn)rmb::sym2_kernel<true, true> : (k2_fn)rmb::sym2_kernel<true, false>)
                           : (periodic ? (k2_fn)rmb::sym2_kernel<false, true> : (k2_fn)rmb::sym2_kernel<false, false>);
  SymPlan plan;
  if (int rc = plan_sym(c, (const void*)fn, &occ2[c->wall ? 1 : 0][periodic ? 1 : 0], 0, a.step_end - a.step_begin, false, &plan,
                        rmb::kSymWavesPerEu))
    return rc;
  const long blocks = plan.blocks;
  a.steps_per_wave = plan.steps_per_wave;
  c->last_path = 1; c->last_tiles = tiles; c->last_chunks = 0; c->last_wgs = blocks;
  int slot;
  if (int rc = timing_begin(c, &slot)) return rc;
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(64 * rmb::kSymWaves), 0, c->stream, a);
  RMB_HIP(hipGetLastError());
  if (int rc = timing_end(c, slot)) return rc;
  const dim3 fgrid((unsigned)((n + 255) / 256));
  if (c->wall) hipLaunchKernelGGL(rmb::sym2_finalize_kernel<true>, fgrid, dim3(256), 0, c->stream, a);
  else         hipLaunchKernelGGL(rmb::sym2_finalize_kernel<false>, fgrid, dim3(256), 0, c->stream, a);
  RMB_HIP(hipGetLastError());
  return 0;
}

// ---- generic symmetric operations (symx_kernels.h) ---------------------------------------------------------
typedef void (*symx_fn)(const rmb::SymXArgs);
typedef void (*symx_combine_fn)(const rmb::SymXArgs, int);
// single-precision twin of an operation (symx32_kernels.h; open boundaries only), or none
typedef void (*symx32_fn)(const rmb::SymXArgs, const rmb::f32::PairConsts);
template <class OP32, bool WALL> struct SymX32 {
  static symx32_fn fn() { return rmb::symx32_kernel<OP32, WALL>; }
  static size_t lds() { return rmb::SymX32Lds<OP32>::bytes; }
};
template <bool WALL> struct SymX32<void, WALL> {
  static symx32_fn fn() { return nullptr; }
  static size_t lds() { return 0; }
};
struct SymXEntry { symx_fn sweep; symx_fn fin; int occ; size_t static_lds; int n_in, n_out; symx_fn det_sweep; symx_fn det_reduce;
                   symx_combine_fn det_combine; int det_occ; symx32_fn sweep32; size_t static_lds32; int occ32; };
template <class OP, bool WALL, bool PER, class OP32 = void> SymXEntry make_symx_entry() {
  return SymXEntry{rmb::symx_kernel<OP, WALL, PER, false>, rmb::symx_finalize_kernel<OP, WALL>, 0,
                   sizeof(double2) * rmb::kSymWaves * 64 * rmb::SymXRec<OP::NIN, rmb::SymXExtra<OP>::value>::d2 +
                       sizeof(double) * rmb::kSymWaves * 3 * OP::NOUT * 64,
                   OP::NIN, OP::NOUT, rmb::symx_kernel<OP, WALL, PER, true>, rmb::symx_det_reduce_kernel<OP::NOUT>,
                   rmb::symx_det_combine_kernel<OP::NOUT>, 0,
                   PER ? nullptr : SymX32<OP32, WALL>::fn(), SymX32<OP32, WALL>::lds(), 0};
}
// SX_K2 + 4 (k - 2) + kind: one block on k = 2..4 vectors
enum SymXOp { SX_TT = 0, SX_TR, SX_RT, SX_RR, SX_FUSED, SX_GRAND, SX_COLF, SX_FREE, SX_RADII, SX_K2, SX_COUNT = SX_K2 + 12 };
// [op][wall][periodic]
#define RMB_SX_ROW(OP) {{make_symx_entry<OP, false, false>(), make_symx_entry<OP, false, true>()}, {make_symx_entry<OP, true, false>(), make_symx_entry<OP, true, true>()}}
#define RMB_SX_ROW32(OP, OP32) {{make_symx_entry<OP, false, false, OP32>(), make_symx_entry<OP, false, true>()}, {make_symx_entry<OP, true, false, OP32>(), make_symx_entry<OP, true, true>()}}
SymXEntry g_symx[SX_COUNT][2][2] = {
    RMB_SX_ROW32(rmb::OpSingle<rmb::KIND_TT>, rmb::OpSingle32<rmb::KIND_TT>), RMB_SX_ROW32(rmb::OpSingle<rmb::KIND_TR>, rmb::OpSingle32<rmb::KIND_TR>),
    RMB_SX_ROW32(rmb::OpSingle<rmb::KIND_RT>, rmb::OpSingle32<rmb::KIND_RT>), RMB_SX_ROW32(rmb::OpSingle<rmb::KIND_RR>, rmb::OpSingle32<rmb::KIND_RR>),
    RMB_SX_ROW32(rmb::OpFusedRow, rmb::OpFusedRow32), RMB_SX_ROW32(rmb::OpGrand, rmb::OpGrand32), RMB_SX_ROW32(rmb::OpColumnF, rmb::OpColumnF32),
    // the free-surface operation takes raw heights: only the wall = 0 column is ever launched
    {{make_symx_entry<rmb::OpFreeSurface, false, false, rmb::OpFreeSurface32>(), make_symx_entry<rmb::OpFreeSurface, false, true>()},
     {make_symx_entry<rmb::OpFreeSurface, false, false, rmb::OpFreeSurface32>(), make_symx_entry<rmb::OpFreeSurface, false, true>()}},
    RMB_SX_ROW32(rmb::OpRadiiTT, rmb::OpRadiiTT32),
#define RMB_SX_K(K) RMB_SX_ROW32(RMB_SX_KIND(rmb::KIND_TT, K), RMB_SX_KIND32(rmb::KIND_TT, K)), RMB_SX_ROW32(RMB_SX_KIND(rmb::KIND_TR, K), RMB_SX_KIND32(rmb::KIND_TR, K)), \
                    RMB_SX_ROW32(RMB_SX_KIND(rmb::KIND_RT, K), RMB_SX_KIND32(rmb::KIND_RT, K)), RMB_SX_ROW32(RMB_SX_KIND(rmb::KIND_RR, K), RMB_SX_KIND32(rmb::KIND_RR, K))
#define RMB_SX_KIND(KIND, K) rmb::OpKindK<KIND, K>
#define RMB_SX_KIND32(KIND, K) rmb::OpKindK32<KIND, K>
    RMB_SX_K(2), RMB_SX_K(3), RMB_SX_K(4)};
#undef RMB_SX_K
#undef RMB_SX_KIND
#undef RMB_SX_KIND32
#undef RMB_SX_ROW
#undef RMB_SX_ROW32

// Configuration a symmetric pass runs on: the context's resident one, or a caller-packed one (per-blob radii)
struct SymConf { const double4* pos; long n; double L[3]; int wall; const double* extra; };
SymConf conf_of(const rmb_ctx* c) {
  return SymConf{(const double4*)c->pos.p, c->n, {c->L[0], c->L[1], c->L[2]}, c->wall, nullptr};
}

int symx_device(rmb_ctx* c, int op, const double* const* in, double* const* out, double eta, int in_plane, long shard,
                long nshards, int accumulate_mask = 0, const SymConf* conf_in = nullptr) {
  const SymConf cf = conf_in ? *conf_in : conf_of(c);
  const bool periodic = cf.L[0] > 0 || cf.L[1] > 0 || cf.L[2] > 0;
  SymXEntry& se = g_symx[op][cf.wall ? 1 : 0][periodic ? 1 : 0];
  const long n = cf.n, tiles = (n + 63) / 64, n_pad = 64 * tiles;
  if (int rc = sym_accumulators(c, n_pad)) return rc;
  rmb::SymXArgs a;
  a.pos = cf.pos;
  a.extra = cf.extra;
  for (int v = 0; v < 4; ++v) { a.in[v] = v < se.n_in ? in[v] : nullptr; a.out[v] = v < se.n_out ? out[v] : nullptr; }
  a.acc = (double*)c->symbuf.p;
  a.n = n; a.n_pad = n_pad; a.n_tiles = (int)tiles; a.n_units = tiles * (tiles + 1) / 2;
  shard_ranges(n, a.n_units, shard, nshards, &a.step_begin, &a.step_end, &a.self_begin, &a.self_end);
  a.Lx = cf.L[0]; a.Ly = cf.L[1]; a.Lz = cf.L[2];
  a.iLx = cf.L[0] > 0 ? 1.0 / cf.L[0] : 0.0;
  a.iLy = cf.L[1] > 0 ? 1.0 / cf.L[1] : 0.0;
  a.iLz = cf.L[2] > 0 ? 1.0 / cf.L[2] : 0.0;
  a.prefactor = 1.0 / (8.0 * M_PI * eta);
  a.accumulate = accumulate_mask;
  a.in_plane = in_plane ? 1 : 0;
  a.skip_pairs = (int)c->opt_skip_pairs;
  a.k = make_pair_consts(c->a > 0.0 ? c->a : 1.0);   // unused by the per-blob-radii operation
  SymPlan plan;
  // "precision" = 32: the operation's single-precision twin where it has one (open boundaries)
  const bool f32 = c->opt_precision == 32 && se.sweep32 != nullptr;
  if (f32 && c->opt_skip_pairs)
    return fail(RMB_ERR_STATE, "the \"skip_pairs\" diagnostic exists in the fp64 kernels only: set \"precision\" = 64");
  if (int rc = plan_sym(c, f32 ? (const void*)se.sweep32 : (const void*)se.sweep, f32 ? &se.occ32 : &se.occ,
                        f32 ? se.static_lds32 : se.static_lds, a.step_end - a.step_begin, true, &plan))
    return rc;
  a.steps_per_wave = plan.steps_per_wave;
  c->last_path = 1; c->last_tiles = tiles; c->last_chunks = 0; c->last_wgs = plan.blocks;
  int slot;
  if (int rc = timing_begin(c, &slot)) return rc;
  if (f32) hipLaunchKernelGGL(se.sweep32, dim3((unsigned)plan.blocks), dim3(64 * rmb::kSymWaves), plan.dyn_lds, c->stream, a, pair_consts32(a.k));
  else     hipLaunchKernelGGL(se.sweep, dim3((unsigned)plan.blocks), dim3(64 * rmb::kSymWaves), plan.dyn_lds, c->stream, a);
  RMB_HIP(hipGetLastError());
  if (int rc = timing_end(c, slot)) return rc;
  hipLaunchKernelGGL(se.fin, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, a);
  RMB_HIP(hipGetLastError());
  return 0;
}

// Deterministic symmetric pass ("deterministic" = 2): same pair arithmetic as symx_device, but whole units per wave and
// per-unit partial results in a bounded workspace instead of atomics, summed in a fixed order by
// symx_det_reduce_kernel; the unit list is processed in chunks that fit the workspace ("det_workspace_mb").
// Pair shard `shard` of `nshards`: whole units [n_units shard / nshards, n_units (shard + 1) / nshards) -- the fixed order
// then holds per rank, and a G-rank run is bit-reproducible as long as the all-reduce is (same ranks, same algorithm).
int symx_det_device(rmb_ctx* c, int op, const double* const* in, double* const* out, double eta, int in_plane,
                    long shard = 0, long nshards = 1) {
  const bool periodic = c->L[0] > 0 || c->L[1] > 0 || c->L[2] > 0;
  SymXEntry& se = g_symx[op][c->wall ? 1 : 0][periodic ? 1 : 0];
  const long n = c->n, tiles = (n + 63) / 64, n_pad = 64 * tiles;
  if (int rc = sym_accumulators(c, n_pad)) return rc;
  rmb::SymXArgs a;
  a.pos = (const double4*)c->pos.p;
  a.extra = nullptr;
  for (int v = 0; v < 4; ++v) { a.in[v] = v < se.n_in ? in[v] : nullptr; a.out[v] = v < se.n_out ? out[v] : nullptr; }
  a.acc = (double*)c->symbuf.p;
  a.n = n; a.n_pad = n_pad; a.n_tiles = (int)tiles; a.n_units = tiles * (tiles + 1) / 2;
  long sb_unused, se_unused;
  shard_ranges(n, a.n_units, shard, nshards, &sb_unused, &se_unused, &a.self_begin, &a.self_end);
  const long shard_ub = (long)((__int128)a.n_units * shard / nshards), shard_ue = (long)((__int128)a.n_units * (shard + 1) / nshards);
  a.Lx = c->L[0]; a.Ly = c->L[1]; a.Lz = c->L[2];
  a.iLx = c->L[0] > 0 ? 1.0 / c->L[0] : 0.0;
  a.iLy = c->L[1] > 0 ? 1.0 / c->L[1] : 0.0;
  a.iLz = c->L[2] > 0 ? 1.0 / c->L[2] : 0.0;
  a.prefactor = 1.0 / (8.0 * M_PI * eta);
  a.accumulate = 0;
  a.in_plane = in_plane ? 1 : 0;
  a.skip_pairs = 0;
  a.k = make_pair_consts(c->a);
  // Chunk = as many units as the workspace holds; whole units per wave, as many waves PER CHUNK LAUNCH as `sym_oversub`
  // resident rounds (so that every chunk fills the chip), never less than one unit each.
  const int wps = resident_blocks((const void*)se.det_sweep, &se.det_occ);
  const long max_waves = c->n_cu * wps * rmb::kSymWaves * c->opt_sym_oversub;
  const size_t slot = (size_t)3 * se.n_out * 64 * sizeof(double);
  long chunk_units = (long)(((size_t)c->opt_det_workspace_mb << 20) / (2 * slot));
  if (chunk_units > shard_ue - shard_ub) chunk_units = shard_ue - shard_ub;
  if (chunk_units < 1) chunk_units = 1;
  const long upw = (chunk_units + max_waves - 1) / max_waves;
  chunk_units = ((chunk_units + upw - 1) / upw) * upw;
  // slices per tile in the ordered reduction: enough workgroups to fill the chip when there are few tiles
  long segs = (4 * c->n_cu + tiles - 1) / tiles;
  if (segs > 32) segs = 32;
  if (segs < 1) segs = 1;
  if (int rc = c->det_ws.reserve((size_t)2 * chunk_units * slot + (size_t)tiles * segs * slot)) return rc;
  a.part_I = (double*)c->det_ws.p;
  a.part_J = a.part_I + chunk_units * (3L * se.n_out * 64);
  a.det_seg = a.part_J + chunk_units * (3L * se.n_out * 64);
  a.units_per_wave = upw;
  a.steps_per_wave = 64 * upw;
  c->last_path = 2; c->last_tiles = tiles; c->last_chunks = 0; c->last_wgs = 0;
  for (long ub = shard_ub; ub < shard_ue; ub += chunk_units) {
    const long ue = ub + chunk_units < shard_ue ? ub + chunk_units : shard_ue;
    a.unit_begin = ub; a.unit_end = ue;
    a.step_begin = 64 * ub; a.step_end = 64 * ue;
    a.first_chunk = ub == shard_ub ? 1 : 0;
    const long waves = (ue - ub + upw - 1) / upw;
    const long blocks = (waves + rmb::kSymWaves - 1) / rmb::kSymWaves;
    c->last_wgs += blocks;
    int slot_t;
    if (int rc = timing_begin(c, &slot_t)) return rc;
    hipLaunchKernelGGL(se.det_sweep, dim3((unsigned)blocks), dim3(64 * rmb::kSymWaves), 0, c->stream, a);
    RMB_HIP(hipGetLastError());
    hipLaunchKernelGGL(se.det_reduce, dim3((unsigned)tiles, (unsigned)segs), dim3(256), 0, c->stream, a);
    hipLaunchKernelGGL(se.det_combine, dim3((unsigned)tiles), dim3(256), 0, c->stream, a, (int)segs);
    RMB_HIP(hipGetLastError());
    if (int rc = timing_end(c, slot_t)) return rc;
  }
  hipLaunchKernelGGL(se.fin, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, a);
  RMB_HIP(hipGetLastError());
  return 0;
}

// whether the symmetric (each unordered pair once) path applies to the resident configuration
bool sym_applies(const rmb_ctx* c) {
  return c->opt_symmetric && c->opt_deterministic != 1 && c->tgt_begin == 0 && c->tgt_end == c->n && c->n >= 128;
}

int matvec_device_impl(rmb_ctx* c, int kind, int in_plane, const double* v, const double* v2, double eta,
                       double* out) {
  if (int rc = check_ready(c)) return rc;
  if (kind < 0 || kind >= rmb::KIND_COUNT) return fail(RMB_ERR_ARG, "kind must be 0..5");
  if (kind == rmb::KIND_TT_FREE && c->wall)
    return fail(RMB_ERR_STATE, "RMB_TT_FREE_SURFACE uses raw heights: call rmb_set_positions with wall = 0");
  const long n_tgt = c->tgt_end - c->tgt_begin;
  if (n_tgt == 0) return 0;
  if (!v || !out) return fail(RMB_ERR_ARG, "null vector / output pointer");
  if (kind == rmb::KIND_TT_TR && !v2) return fail(RMB_ERR_ARG, "RMB_TT_TR needs vec2 (torque)");
  if (!(eta > 0.0)) return fail(RMB_ERR_ARG, "eta must be positive");
  RMB_HIP(hipSetDevice(c->device));

  const bool periodic = c->L[0] > 0 || c->L[1] > 0 || c->L[2] > 0;
  c->last_path = 0;
  if (sym_applies(c)) {
    // every product of the surface is a symmetric operator: each unordered pair once, applied to both blobs
    c->last_path = 1;
    const double* in[2] = {v, v2};
    double* outs[1] = {out};
    if (c->opt_deterministic == 2) {   // bit-reproducible AND symmetric: ordered reduction instead of atomics
      const int sx = kind <= rmb::KIND_RR ? SX_TT + kind : (kind == rmb::KIND_TT_TR ? SX_FUSED : SX_FREE);
      return symx_det_device(c, sx, in, outs, eta, in_plane);
    }
    if (kind <= rmb::KIND_RR) {
      // tr / rt / rr in single precision run on the generic skeleton's fp32 twin (tt has its own kernel in sym_device)
      const bool x32 = c->opt_precision == 32 && kind != rmb::KIND_TT && !periodic;
      if (in_plane || c->opt_symx_single || x32) return symx_device(c, SX_TT + kind, in, outs, eta, in_plane, 0, 1);
      return sym_device(c, kind, v, eta, out);
    }
    if (kind == rmb::KIND_TT_TR) {
      if (c->opt_fused_symmetric == 2) {   // round-1 path, kept for A/B: two symmetric passes into one output
        if (in_plane) return fail(RMB_ERR_ARG, "fused_symmetric = 2 has no in-plane variant");
        if (int rc = sym_device(c, rmb::KIND_TT, v, eta, out)) return rc;
        return sym_device(c, rmb::KIND_TR, v2, eta, out, 0, 1, true);
      }
      if (c->opt_fused_symmetric) return symx_device(c, SX_FUSED, in, outs, eta, in_plane, 0, 1);
    }
    if (kind == rmb::KIND_TT_FREE) return symx_device(c, SX_FREE, in, outs, eta, in_plane, 0, 1);
    c->last_path = 0;
  }
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

rmb::ExpConsts exp_consts() {
  rmb::ExpConsts e;
  e.log2e = 1.4426950408889634;
  e.ln2_hi = 6.93147180369123816490e-01;  // ln2 in two pieces; the high one has 33 significant bits, so n * ln2_hi is exact
  e.ln2_lo = 1.90821492927058770002e-10;
  double f = 2.0;
  for (int k = 0; k < 12; ++k) { e.c[k] = 1.0 / f; f *= (double)(k + 3); }
  return e;
}

// shard / nshards: pair shard of the unordered pairs (F_ji = -F_ij needs no self term) into a full-length partial; a
// shard always takes the symmetric kernel, whatever n and the target range (as rmb_matvec_pairshard_device).
int force_device_impl(rmb_ctx* c, double eps, double b, double blob_radius, double* out, const double* radii = nullptr,
                      long shard = 0, long nshards = 1) {
  if (int rc = check_ready(c)) return rc;
  const long n_tgt = nshards > 1 ? c->n : c->tgt_end - c->tgt_begin;
  if (n_tgt == 0) return 0;
  if (!out) return fail(RMB_ERR_ARG, "null output pointer");
  if (!(b > 0.0)) return fail(RMB_ERR_ARG, "debye_length must be positive");
  if (nshards < 1 || shard < 0 || shard >= nshards) return fail(RMB_ERR_ARG, "bad shard / nshards");
  RMB_HIP(hipSetDevice(c->device));
  const bool periodic = c->L[0] > 0 || c->L[1] > 0 || c->L[2] > 0;
  c->last_path = 0;
  if (nshards > 1 || (sym_applies(c) && c->opt_deterministic == 0)) {
    // symmetric path: each unordered pair once (F_ji = -F_ij); its flushes are atomics, so both deterministic modes
    // take the one-sided sweep below
    const long n = c->n, tiles = (n + 63) / 64, n_pad = 64 * tiles;
    if (int rc = sym_accumulators(c, n_pad)) return rc;
    rmb::SymForceArgs a;
    a.pos = (const double4*)c->pos.p;
    a.acc = (double*)c->symbuf.p;
    a.out = out;
    a.n = n; a.n_pad = n_pad; a.n_tiles = (int)tiles; a.n_units = tiles * (tiles + 1) / 2;
    a.Lx = c->L[0]; a.Ly = c->L[1]; a.Lz = c->L[2];
    a.iLx = c->L[0] > 0 ? 1.0 / c->L[0] : 0.0;
    a.iLy = c->L[1] > 0 ? 1.0 / c->L[1] : 0.0;
    a.iLz = c->L[2] > 0 ? 1.0 / c->L[2] : 0.0;
    a.eps_over_b = eps / b; a.inv_b = 1.0 / b; a.two_a = 2.0 * blob_radius;
    a.ec = exp_consts();
    a.radii = radii;
    {
      const __int128 s_all = (__int128)a.n_units * 64;
      a.step_begin = (long)(s_all * shard / nshards);
      a.step_end = (long)(s_all * (shard + 1) / nshards);
    }
    // "precision" = 32, open boundaries: the single-precision kernel -- the arithmetic of the reference's own GPU force
    // kernel (forces_pycuda.py:14-21)
    const bool f32 = (c->opt_force_precision ? c->opt_force_precision : c->opt_precision) == 32 && !periodic;
    // Tile culling: the force has the range of its exponential.  exp(-(r - 2a)/b) is exactly 0 in double precision
    // beyond (r - 2a)/b = 745.2 (750 here; 110 for the float kernel), so a tile pair whose bounding boxes are further
    // apart contributes nothing, bit for bit.  In a 262 144-roller monolayer that is 99 % of the tile pairs -- the
    // reference's own answer to this is a k-d tree (`blob_blob_force_implementation tree_numba`).
    a.bounds = nullptr; a.cull2 = 0.0;
    if (c->opt_force_cull && !radii && tiles > 1) {
      if (!c->tile_bounds_valid) {
        if (int rc = c->tile_bounds.reserve((size_t)6 * tiles * sizeof(double))) return rc;
        hipLaunchKernelGGL(rmb::tile_bounds_kernel, dim3((unsigned)tiles), dim3(64), 0, c->stream, (const double4*)c->pos.p, n,
                           (double*)c->tile_bounds.p);
        RMB_HIP(hipGetLastError());
        c->tile_bounds_valid = true;
      }
      const double reach = 2.0 * blob_radius + (f32 ? 110.0 : 750.0) * b;
      a.bounds = (const double*)c->tile_bounds.p;
      a.cull2 = reach * reach;
    }
    static int socc[2][2] = {{0, 0}, {0, 0}};
    typedef void (*sforce_fn)(const rmb::SymForceArgs);
    const sforce_fn sfn = radii ? (periodic ? (sforce_fn)rmb::sym_force_kernel<true, true> : (sforce_fn)rmb::sym_force_kernel<false, true>)
                                : (periodic ? (sforce_fn)rmb::sym_force_kernel<true, false> : (sforce_fn)rmb::sym_force_kernel<false, false>);
    static int socc32[2] = {0, 0};
    const sforce_fn sfn32 = radii ? (sforce_fn)rmb::sym_force32_kernel<true> : (sforce_fn)rmb::sym_force32_kernel<false>;
    const void* fn = f32 ? (const void*)sfn32 : (const void*)sfn;
    long blocks = c->n_cu * resident_blocks(fn, f32 ? &socc32[radii ? 1 : 0] : &socc[radii ? 1 : 0][periodic ? 1 : 0]) * c->opt_sym_oversub;
    long need = (a.step_end - a.step_begin + 255) / 256;
    if (need < 1) need = 1;
    if (blocks > need) blocks = need;
    c->last_path = 1; c->last_tiles = tiles; c->last_chunks = 0; c->last_wgs = blocks;
    int slot;
    if (int rc = timing_begin(c, &slot)) return rc;
    hipLaunchKernelGGL(f32 ? sfn32 : sfn, dim3((unsigned)blocks), dim3(64 * rmb::kSymWaves), 0, c->stream, a);
    RMB_HIP(hipGetLastError());
    if (int rc = timing_end(c, slot)) return rc;
    hipLaunchKernelGGL(rmb::sym_force_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, a);
    RMB_HIP(hipGetLastError());
    return 0;
  }
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

int set_positions_impl(rmb_ctx* c, const double* r_dev, long n, double a, const double* L, int wall) {
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

__global__ void add_inplace_kernel(double* y, const double* x, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += x[i];
}

// Multi-block operations (include/rmb_mobility.h, enum rmb_op).  One symmetric pass when that path applies (or for a
// pair shard); otherwise composed from the one-sided sweeps (target sub-ranges, "deterministic", n < 128).
int matvec_op_impl(rmb_ctx* c, int op, int in_plane, int n_in, const double* const* in, int n_out, double* const* out,
                   double eta, long shard, long nshards) {
  if (int rc = check_ready(c)) return rc;
  if (!in || !out) return fail(RMB_ERR_ARG, "null vector / output list");
  if (!(eta > 0.0)) return fail(RMB_ERR_ARG, "eta must be positive");
  if (nshards < 1 || shard < 0 || shard >= nshards) return fail(RMB_ERR_ARG, "bad shard / nshards");
  int want_in = 0, want_out = 0, sx = -1, multi_kind = rmb::KIND_TT;
  switch (op) {
    case RMB_OP_VELOCITY_FROM_FORCE_TORQUE: want_in = 2; want_out = 1; sx = SX_FUSED; break;
    case RMB_OP_GRAND: want_in = 2; want_out = 2; sx = SX_GRAND; break;
    case RMB_OP_FORCE_COLUMN: want_in = 1; want_out = 2; sx = SX_COLF; break;
    case RMB_OP_TT_MULTI: case RMB_OP_TR_MULTI: case RMB_OP_RT_MULTI: case RMB_OP_RR_MULTI: {
      if (n_in < 1 || n_in > 4) return fail(RMB_ERR_ARG, "RMB_OP_*_MULTI takes 1..4 vectors");
      want_in = want_out = n_in;
      multi_kind = op - RMB_OP_TT_MULTI;      // rmb_kind of the block
      sx = n_in == 1 ? SX_TT + multi_kind : SX_K2 + 4 * (n_in - 2) + multi_kind;
      break;
    }
    default: return fail(RMB_ERR_ARG, "unknown rmb_op");
  }
  if (n_in != want_in || n_out != want_out) return fail(RMB_ERR_ARG, "wrong number of input / output vectors for this rmb_op");
  for (int v = 0; v < n_in; ++v) if (!in[v]) return fail(RMB_ERR_ARG, "null input vector");
  for (int v = 0; v < n_out; ++v) if (!out[v]) return fail(RMB_ERR_ARG, "null output vector");
  if (c->n == 0) return 0;
  RMB_HIP(hipSetDevice(c->device));
  // a pair shard always writes all n targets, whatever target range is set (as rmb_matvec_pairshard_device)
  if (c->opt_deterministic == 2 && (nshards > 1 || sym_applies(c))) return symx_det_device(c, sx, in, out, eta, in_plane, shard, nshards);
  if (sym_applies(c) || nshards > 1) return symx_device(c, sx, in, out, eta, in_plane, shard, nshards);
  const long n_tgt = c->tgt_end - c->tgt_begin;
  if (n_tgt == 0) return 0;
  switch (op) {
    case RMB_OP_VELOCITY_FROM_FORCE_TORQUE:
      return matvec_device_impl(c, rmb::KIND_TT_TR, in_plane, in[0], in[1], eta, out[0]);
    case RMB_OP_FORCE_COLUMN:
      if (int rc = matvec_device_impl(c, rmb::KIND_TT, in_plane, in[0], nullptr, eta, out[0])) return rc;
      return matvec_device_impl(c, rmb::KIND_RT, in_plane, in[0], nullptr, eta, out[1]);
    case RMB_OP_GRAND: {
      if (int rc = matvec_device_impl(c, rmb::KIND_TT_TR, in_plane, in[0], in[1], eta, out[0])) return rc;
      if (int rc = matvec_device_impl(c, rmb::KIND_RT, in_plane, in[0], nullptr, eta, out[1])) return rc;
      if (int rc = c->tmp3n.reserve((size_t)3 * n_tgt * sizeof(double))) return rc;
      if (int rc = matvec_device_impl(c, rmb::KIND_RR, in_plane, in[1], nullptr, eta, (double*)c->tmp3n.p)) return rc;
      hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((3 * n_tgt + 255) / 256)), dim3(256), 0, c->stream, out[1],
                         (const double*)c->tmp3n.p, 3 * n_tgt);
      RMB_HIP(hipGetLastError());
      return 0;
    }
    default:
      for (int v = 0; v < n_in; ++v)
        if (int rc = matvec_device_impl(c, multi_kind, in_plane, in[v], nullptr, eta, out[v])) return rc;
      return 0;
  }
}

std::mutex g_default_mu;
rmb_ctx* g_default_ctx = nullptr;

}  // namespace

// ---- Stokeslet pressure / Stokes double layer, source -> target (aux_kernels.h) ---------------------------------
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

// host arrays -> staging buffers of the default context; returns device pointers in dev[]
int aux_stage(rmb_ctx* c, int n, const double* const* host, const size_t* bytes, const int* slot, const double** dev) {
  for (int k = 0; k < n; ++k) {
    if (int rc = c->st[slot[k]].reserve(bytes[k] ? bytes[k] : sizeof(double))) return rc;
    if (bytes[k]) RMB_HIP(hipMemcpyAsync(c->st[slot[k]].p, host[k], bytes[k], hipMemcpyHostToDevice, c->stream));
    dev[k] = (const double*)c->st[slot[k]].p;
  }
  return 0;
}
}  // namespace


extern "C" {

const char* rmb_version(void) { return "rmb_mobility 0.1 (gfx950)"; }
const char* rmb_last_error(void) { return g_err.c_str(); }

int rmb_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int rmb_ctx_create(int device, rmb_ctx** out) {
  if (!out) return fail(RMB_ERR_ARG, "null ctx out pointer");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0)
    return fail(RMB_ERR_NO_DEVICE, std::string("no HIP device visible (") + hipGetErrorString(e) + ")");
  if (device < 0 || device >= n) return fail(RMB_ERR_ARG, "device index out of range");
  RMB_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  RMB_HIP(hipGetDeviceProperties(&prop, device));
  rmb_ctx* c = new rmb_ctx();
  c->device = device;
  if (prop.multiProcessorCount > 0) c->n_cu = prop.multiProcessorCount;
  if (prop.maxSharedMemoryPerMultiProcessor > 0) c->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor;
  *out = c;
  return 0;
}

int rmb_ctx_destroy(rmb_ctx* c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  c->wave_clock.release(); c->tile_bounds.release(); for (auto& b : c->st) b.release(); c->symbuf.release(); c->pos.release(); c->r_stage.release(); c->vec.release(); c->vec2.release(); c->out.release(); c->partial.release(); c->tmp3n.release(); c->det_ws.release();
  if (c->stream_switch) (void)hipEventDestroy(c->stream_switch);
  for (auto e : c->ev0) (void)hipEventDestroy(e);
  for (auto e : c->ev1) (void)hipEventDestroy(e);
  delete c;
  return 0;
}

int rmb_ctx_set_stream(rmb_ctx* c, void* s) {
  if (!c) return fail(RMB_ERR_ARG, "null context");
  const hipStream_t next = (hipStream_t)s;
  if (next != c->stream) {
    // A context is single-stream at a time: accumulators, workspaces and the packed positions are re-used from call
    // to call, so the new stream must not start before what was queued on the previous one has finished.
    RMB_HIP(hipSetDevice(c->device));
    if (!c->stream_switch) RMB_HIP(hipEventCreateWithFlags(&c->stream_switch, hipEventDisableTiming));
    // The previous handle must still be alive here (HIP does not validate stream handles: recording on a destroyed
    // one is a use-after-free, it crashed in the round-3 test) -- a host that destroys its streams calls
    // rmb_ctx_release_stream() first.  Whatever the record returns, the new handle is adopted: a context never stays
    // bound to a stream it failed to fence.
    hipError_t e = hipEventRecord(c->stream_switch, c->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(next, c->stream_switch, 0);
    c->stream = next;
    if (e != hipSuccess) {
      (void)hipGetLastError();
      RMB_HIP(hipDeviceSynchronize());
    }
  }
  return 0;
}

int rmb_ctx_release_stream(rmb_ctx* c) {
  if (!c) return fail(RMB_ERR_ARG, "null context");
  RMB_HIP(hipSetDevice(c->device));
  hipError_t e = hipStreamSynchronize(c->stream);     // the stream is still alive: its owner calls this BEFORE destroying it
  c->stream = nullptr;                                // from here on the context does not know the old handle any more
  if (e != hipSuccess) return fail(RMB_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
  return 0;
}

int rmb_ctx_set_option(rmb_ctx* c, const char* key, long value) {
  if (!c || !key) return fail(RMB_ERR_ARG, "null context / key");
  if (!strcmp(key, "chunks")) { c->opt_chunks = value; return 0; }
  if (!strcmp(key, "timing")) { c->opt_timing = value; return 0; }
  if (!strcmp(key, "symmetric")) { c->opt_symmetric = value; return 0; }
  if (!strcmp(key, "fused_symmetric")) { c->opt_fused_symmetric = value; return 0; }
  if (!strcmp(key, "symx_single")) { c->opt_symx_single = value; return 0; }
  if (!strcmp(key, "deterministic")) { c->opt_deterministic = value; return 0; }
  if (!strcmp(key, "det_workspace_mb")) { c->opt_det_workspace_mb = value < 1 ? 1 : value; return 0; }
  if (!strcmp(key, "sym_wps")) { c->opt_sym_wps = value; return 0; }
  if (!strcmp(key, "wave_clock")) { c->opt_wave_clock = value; return 0; }
  if (!strcmp(key, "skip_pairs")) { c->opt_skip_pairs = value; return 0; }
  if (!strcmp(key, "sym_pin")) { c->opt_sym_pin = value; return 0; }
  if (!strcmp(key, "precision")) {
    if (value != 32 && value != 64) return fail(RMB_ERR_ARG, "precision must be 32 or 64");
    c->opt_precision = value;
    return 0;
  }
  if (!strcmp(key, "force_cull")) { c->opt_force_cull = value ? 1 : 0; return 0; }
  if (!strcmp(key, "force_precision")) {
    if (value != 0 && value != 32 && value != 64) return fail(RMB_ERR_ARG, "force_precision must be 0 (follow \"precision\"), 32 or 64");
    c->opt_force_precision = value;
    return 0;
  }
  if (!strcmp(key, "sym_fine_steps")) { c->opt_sym_fine_steps = value < 0 ? 0 : value; return 0; }
  if (!strcmp(key, "sym_oversub")) { c->opt_sym_oversub = value < 1 ? 1 : value; return 0; }
  if (!strcmp(key, "sym_min_steps")) { c->opt_sym_min_steps = value < 1 ? 1 : value; return 0; }
  return fail(RMB_ERR_ARG, std::string("unknown option: ") + key);
}

int rmb_ctx_get_option(rmb_ctx* c, const char* key, long* value) {
  if (!c || !key || !value) return fail(RMB_ERR_ARG, "null context / key / value");
  const struct { const char* name; const long* v; } table[] = {
      {"chunks", &c->opt_chunks}, {"timing", &c->opt_timing}, {"symmetric", &c->opt_symmetric},
      {"fused_symmetric", &c->opt_fused_symmetric}, {"symx_single", &c->opt_symx_single},
      {"deterministic", &c->opt_deterministic}, {"det_workspace_mb", &c->opt_det_workspace_mb}, {"sym_wps", &c->opt_sym_wps},
      {"wave_clock", &c->opt_wave_clock}, {"skip_pairs", &c->opt_skip_pairs}, {"sym_pin", &c->opt_sym_pin},
      {"precision", &c->opt_precision}, {"force_precision", &c->opt_force_precision}, {"force_cull", &c->opt_force_cull}, {"sym_oversub", &c->opt_sym_oversub}, {"sym_fine_steps", &c->opt_sym_fine_steps},
      {"sym_min_steps", &c->opt_sym_min_steps}};
  for (const auto& e : table)
    if (!strcmp(key, e.name)) { *value = *e.v; return 0; }
  return fail(RMB_ERR_ARG, std::string("unknown option: ") + key);
}

int rmb_set_positions(rmb_ctx* c, const double* r, long n, double a, const double* L, int wall) {
  if (!c) return fail(RMB_ERR_ARG, "null context");
  if (n < 0) return fail(RMB_ERR_ARG, "negative n");
  if (n > 0 && !r) return fail(RMB_ERR_ARG, "null positions");
  if (!(a > 0.0)) return fail(RMB_ERR_ARG, "blob radius must be positive");
  RMB_HIP(hipSetDevice(c->device));
  if (n > 0) {
    if (int rc = c->r_stage.reserve((size_t)3 * n * sizeof(double))) return rc;
    RMB_HIP(hipMemcpyAsync(c->r_stage.p, r, (size_t)3 * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  if (int rc = set_positions_impl(c, (const double*)c->r_stage.p, n, a, L, wall)) return rc;
  RMB_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int rmb_set_positions_device(rmb_ctx* c, const double* r_dev, long n, double a, const double* L, int wall) {
  if (!c) return fail(RMB_ERR_ARG, "null context");
  if (n < 0) return fail(RMB_ERR_ARG, "negative n");
  if (n > 0 && !r_dev) return fail(RMB_ERR_ARG, "null positions");
  if (!(a > 0.0)) return fail(RMB_ERR_ARG, "blob radius must be positive");
  RMB_HIP(hipSetDevice(c->device));
  return set_positions_impl(c, r_dev, n, a, L, wall);
}

int rmb_set_target_range(rmb_ctx* c, long begin, long end) {
  if (int rc = check_ready(c)) return rc;
  if (begin < 0 || end < begin || end > c->n) return fail(RMB_ERR_STATE, "target range must satisfy 0 <= begin <= end <= n");
  c->tgt_begin = begin;
  c->tgt_end = end;
  return 0;
}

int rmb_matvec_device(rmb_ctx* c, int kind, int in_plane, const double* v, const double* v2, double eta, double* out) {
  return matvec_device_impl(c, kind, in_plane, v, v2, eta, out);
}

int rmb_matvec2_pairshard_device(rmb_ctx* c, int kind, const double* vec_a, const double* vec_b, double eta,
                                 double* out_a, double* out_b, long shard, long nshards) {
  if (int rc = check_ready(c)) return rc;
  if (kind != rmb::KIND_TT) return fail(RMB_ERR_ARG, "two-vector products exist for RMB_TT only");
  if (!vec_a || !vec_b || !out_a || !out_b) return fail(RMB_ERR_ARG, "null vector / output pointer");
  if (!(eta > 0.0)) return fail(RMB_ERR_ARG, "eta must be positive");
  if (nshards < 1 || shard < 0 || shard >= nshards) return fail(RMB_ERR_ARG, "bad shard");
  RMB_HIP(hipSetDevice(c->device));
  if (!sym_applies(c) && nshards == 1) {
    if (int rc = matvec_device_impl(c, kind, 0, vec_a, nullptr, eta, out_a)) return rc;
    return matvec_device_impl(c, kind, 0, vec_b, nullptr, eta, out_b);
  }
  // a pair shard (nshards > 1) always runs the symmetric kernel, whatever n: it is the only kernel that can
  // evaluate a slice of the unordered pairs (rmb_matvec_pairshard_device does the same)
  const bool periodic = c->L[0] > 0 || c->L[1] > 0 || c->L[2] > 0;
  const bool x32 = c->opt_precision == 32 && !periodic;     // the generic skeleton has the single-precision twin
  if (c->opt_symx_single || x32 || c->opt_deterministic == 2) {
    const double* in[2] = {vec_a, vec_b};
    double* outs[2] = {out_a, out_b};
    if (c->opt_deterministic == 2) return symx_det_device(c, SX_K2, in, outs, eta, 0, shard, nshards);
    return symx_device(c, SX_K2, in, outs, eta, 0, shard, nshards);
  }
  return sym2_device(c, vec_a, vec_b, eta, out_a, out_b, shard, nshards);
}

int rmb_matvec2_device(rmb_ctx* c, int kind, const double* vec_a, const double* vec_b, double eta, double* out_a,
                       double* out_b) {
  if (int rc = check_ready(c)) return rc;
  if (c->tgt_begin != 0 || c->tgt_end != c->n) {      // target shards: two one-sided sweeps
    if (int rc = matvec_device_impl(c, kind, 0, vec_a, nullptr, eta, out_a)) return rc;
    return matvec_device_impl(c, kind, 0, vec_b, nullptr, eta, out_b);
  }
  return rmb_matvec2_pairshard_device(c, kind, vec_a, vec_b, eta, out_a, out_b, 0, 1);
}

int rmb_matvec_pairshard_device(rmb_ctx* c, int kind, const double* v, double eta, double* out, long shard, long nshards) {
  if (int rc = check_ready(c)) return rc;
  if ((kind < 0 || kind > rmb::KIND_RR) && kind != rmb::KIND_TT_FREE)
    return fail(RMB_ERR_ARG, "pair sharding is implemented for RMB_TT / TR / RT / RR / TT_FREE_SURFACE (RMB_TT_TR: rmb_matvec_op_pairshard_device)");
  if (kind == rmb::KIND_TT_FREE && c->wall)
    return fail(RMB_ERR_STATE, "RMB_TT_FREE_SURFACE uses raw heights: call rmb_set_positions with wall = 0");
  if (nshards < 1 || shard < 0 || shard >= nshards) return fail(RMB_ERR_ARG, "bad shard / nshards");
  if (c->n == 0) return 0;
  if (!v || !out) return fail(RMB_ERR_ARG, "null vector / output pointer");
  if (!(eta > 0.0)) return fail(RMB_ERR_ARG, "eta must be positive");
  RMB_HIP(hipSetDevice(c->device));
  c->last_path = 1;
  const int sx = kind == rmb::KIND_TT_FREE ? SX_FREE : SX_TT + kind;
  const double* in[2] = {v, nullptr};
  double* outs[1] = {out};
  if (c->opt_deterministic == 2)        // bit-reproducible shard: whole units, ordered reduction (symx_det_device)
    return symx_det_device(c, sx, in, outs, eta, 0, shard, nshards);
  if (kind == rmb::KIND_TT_FREE) return symx_device(c, SX_FREE, in, outs, eta, 0, shard, nshards);
  return sym_device(c, kind, v, eta, out, shard, nshards);
}

int rmb_matvec_op_device(rmb_ctx* c, int op, int in_plane, int n_in, const double* const* in_dev, int n_out,
                         double* const* out_dev, double eta) {
  return matvec_op_impl(c, op, in_plane, n_in, in_dev, n_out, out_dev, eta, 0, 1);
}

int rmb_matvec_op_pairshard_device(rmb_ctx* c, int op, int in_plane, int n_in, const double* const* in_dev, int n_out,
                                   double* const* out_dev, double eta, long shard, long nshards) {
  return matvec_op_impl(c, op, in_plane, n_in, in_dev, n_out, out_dev, eta, shard, nshards);
}

int rmb_body_mobility_dense_device(rmb_ctx* c, const long* first_blob_dev, long n_bodies, int n_b, double eta,
                                   double* out_dev) {
  if (int rc = check_ready(c)) return rc;
  if (n_bodies < 0 || n_b < 1) return fail(RMB_ERR_ARG, "bad n_bodies / blobs per body");
  if (n_bodies == 0) return 0;
  if (!first_blob_dev || !out_dev) return fail(RMB_ERR_ARG, "null pointer");
  if (!(eta > 0.0)) return fail(RMB_ERR_ARG, "eta must be positive");
  // Periodic contexts are accepted: a body's own block never includes images (the reference's per-body
  // b.calc_mobility_blobs, body/body.py:186-191, has no periodic_length either).
  RMB_HIP(hipSetDevice(c->device));
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

int rmb_matvec(rmb_ctx* c, int kind, int in_plane, const double* v, const double* v2, double eta, double* out) {
  if (int rc = check_ready(c)) return rc;
  const long n = c->n, n_tgt = c->tgt_end - c->tgt_begin;
  if (n_tgt == 0) return 0;
  if (!v || !out) return fail(RMB_ERR_ARG, "null vector / output pointer");
  if (kind == rmb::KIND_TT_TR && !v2) return fail(RMB_ERR_ARG, "RMB_TT_TR needs vec2 (torque)");
  RMB_HIP(hipSetDevice(c->device));
  const size_t vb = (size_t)3 * n * sizeof(double), ob = (size_t)3 * n_tgt * sizeof(double);
  if (int rc = c->vec.reserve(vb)) return rc;
  if (int rc = c->out.reserve(ob)) return rc;
  RMB_HIP(hipMemcpyAsync(c->vec.p, v, vb, hipMemcpyHostToDevice, c->stream));
  const double* v2d = nullptr;
  if (kind == rmb::KIND_TT_TR) {
    if (int rc = c->vec2.reserve(vb)) return rc;
    RMB_HIP(hipMemcpyAsync(c->vec2.p, v2, vb, hipMemcpyHostToDevice, c->stream));
    v2d = (const double*)c->vec2.p;
  }
  if (int rc = matvec_device_impl(c, kind, in_plane, (const double*)c->vec.p, v2d, eta, (double*)c->out.p)) return rc;
  RMB_HIP(hipMemcpyAsync(out, c->out.p, ob, hipMemcpyDeviceToHost, c->stream));
  RMB_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int rmb_blob_blob_force_device(rmb_ctx* c, double eps, double b, double blob_radius, double* out) {
  return force_device_impl(c, eps, b, blob_radius, out);
}

int rmb_blob_blob_force_pairshard_device(rmb_ctx* c, double eps, double b, double blob_radius, double* out, long shard,
                                         long nshards) {
  return force_device_impl(c, eps, b, blob_radius, out, nullptr, shard, nshards);
}

int rmb_blob_blob_force_radii_device(rmb_ctx* c, const double* radii_dev, double eps, double b, double* out) {
  if (!radii_dev) return fail(RMB_ERR_ARG, "null radii pointer");
  return force_device_impl(c, eps, b, 0.0, out, radii_dev);
}

int rmb_blob_blob_force_radii(rmb_ctx* c, const double* radii, double eps, double b, double* out) {
  if (int rc = check_ready(c)) return rc;
  const long n_tgt = c->tgt_end - c->tgt_begin;
  if (n_tgt == 0) return 0;
  if (!out || !radii) return fail(RMB_ERR_ARG, "null pointer");
  RMB_HIP(hipSetDevice(c->device));
  const size_t ob = (size_t)3 * n_tgt * sizeof(double), rb = (size_t)c->n * sizeof(double);
  if (int rc = c->out.reserve(ob)) return rc;
  if (int rc = c->vec2.reserve(rb)) return rc;
  RMB_HIP(hipMemcpyAsync(c->vec2.p, radii, rb, hipMemcpyHostToDevice, c->stream));
  if (int rc = force_device_impl(c, eps, b, 0.0, (double*)c->out.p, (const double*)c->vec2.p)) return rc;
  RMB_HIP(hipMemcpyAsync(out, c->out.p, ob, hipMemcpyDeviceToHost, c->stream));
  RMB_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int rmb_blob_blob_force(rmb_ctx* c, double eps, double b, double blob_radius, double* out) {
  if (int rc = check_ready(c)) return rc;
  const long n_tgt = c->tgt_end - c->tgt_begin;
  if (n_tgt == 0) return 0;
  if (!out) return fail(RMB_ERR_ARG, "null output pointer");
  RMB_HIP(hipSetDevice(c->device));
  const size_t ob = (size_t)3 * n_tgt * sizeof(double);
  if (int rc = c->out.reserve(ob)) return rc;
  if (int rc = force_device_impl(c, eps, b, blob_radius, (double*)c->out.p)) return rc;
  RMB_HIP(hipMemcpyAsync(out, c->out.p, ob, hipMemcpyDeviceToHost, c->stream));
  RMB_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int rmb_mobility_source_target_device(rmb_ctx* c, long ns, const double* src_dev, const double* rad_s_dev, long nt,
                                      const double* tgt_dev, const double* rad_t_dev, const double* force_dev,
                                      double eta, const double* L, int wall, double* out_dev) {
  if (!c) return fail(RMB_ERR_ARG, "null context");
  if (ns < 0 || nt < 0) return fail(RMB_ERR_ARG, "negative size");
  if (nt == 0) return 0;
  if (!out_dev || !tgt_dev || !rad_t_dev) return fail(RMB_ERR_ARG, "null target pointer");
  if (ns > 0 && (!src_dev || !rad_s_dev || !force_dev)) return fail(RMB_ERR_ARG, "null source pointer");
  if (!(eta > 0.0)) return fail(RMB_ERR_ARG, "eta must be positive");
  RMB_HIP(hipSetDevice(c->device));
  if (ns == 0) { RMB_HIP(hipMemsetAsync(out_dev, 0, (size_t)3 * nt * sizeof(double), c->stream)); return 0; }
  if (src_dev == tgt_dev && rad_s_dev == rad_t_dev && ns == nt && ns >= 128 && (wall == 0 || wall == 1) &&
      c->opt_symmetric && c->opt_deterministic == 0) {
    // Sources == targets (the reference's `radii_*` mobility modes, mobility/mobility.py:1369-1374): the operator is
    // symmetric, each unordered pair once on the generic symmetric skeleton (symx_kernels.h, OpRadiiTT)
    if (int rc = c->st[0].reserve((size_t)ns * sizeof(double4))) return rc;
    hipLaunchKernelGGL(rmb::pack_positions_radii_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, c->stream, src_dev,
                       rad_s_dev, ns, wall, (double4*)c->st[0].p);
    RMB_HIP(hipGetLastError());
    SymConf cf{(const double4*)c->st[0].p, ns, {L ? L[0] : 0.0, L ? L[1] : 0.0, L ? L[2] : 0.0}, wall, rad_s_dev};
    const double* in[1] = {force_dev};
    double* outs[1] = {out_dev};
    return symx_device(c, SX_RADII, in, outs, eta, 0, 0, 1, 0, &cf);
  }
  if (int rc = c->st[0].reserve((size_t)ns * sizeof(double4))) return rc;
  if (int rc = c->st[1].reserve((size_t)nt * sizeof(double4))) return rc;
  hipLaunchKernelGGL(rmb::pack_positions_radii_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, c->stream, src_dev,
                     rad_s_dev, ns, wall == 1 ? 1 : 0, (double4*)c->st[0].p);
  hipLaunchKernelGGL(rmb::pack_positions_radii_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, c->stream, tgt_dev,
                     rad_t_dev, nt, wall == 1 ? 1 : 0, (double4*)c->st[1].p);
  RMB_HIP(hipGetLastError());
  rmb::StArgs a;
  a.src = (const double4*)c->st[0].p; a.rad_s = rad_s_dev; a.force = force_dev;
  a.tgt = (const double4*)c->st[1].p; a.rad_t = rad_t_dev; a.out = out_dev; a.partial = nullptr;
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

int rmb_mobility_source_target(long ns, const double* src, const double* rad_s, long nt, const double* tgt,
                               const double* rad_t, const double* force, double eta, const double* L, int wall,
                               double* out) {
  std::lock_guard<std::mutex> lk(g_default_mu);
  if (!g_default_ctx) {
    if (int rc = rmb_ctx_create(0, &g_default_ctx)) return rc;
  }
  rmb_ctx* c = g_default_ctx;
  if (ns < 0 || nt < 0) return fail(RMB_ERR_ARG, "negative size");
  if (nt == 0) return 0;
  if (!out || !tgt || !rad_t || (ns > 0 && (!src || !rad_s || !force))) return fail(RMB_ERR_ARG, "null pointer");
  RMB_HIP(hipSetDevice(c->device));
  const size_t bs3 = (size_t)3 * (ns > 0 ? ns : 1) * sizeof(double), bt3 = (size_t)3 * nt * sizeof(double);
  const size_t bs1 = (size_t)(ns > 0 ? ns : 1) * sizeof(double), bt1 = (size_t)nt * sizeof(double);
  if (int rc = c->st[2].reserve(bs3)) return rc;   // src
  if (int rc = c->st[3].reserve(bs1)) return rc;   // rad_s
  if (int rc = c->st[4].reserve(bt3)) return rc;   // tgt
  if (int rc = c->st[5].reserve(bt1)) return rc;   // rad_t
  if (int rc = c->st[6].reserve(bs3)) return rc;   // force
  if (int rc = c->st[7].reserve(bt3)) return rc;   // out
  if (ns > 0) {
    RMB_HIP(hipMemcpyAsync(c->st[2].p, src, (size_t)3 * ns * sizeof(double), hipMemcpyHostToDevice, c->stream));
    RMB_HIP(hipMemcpyAsync(c->st[3].p, rad_s, (size_t)ns * sizeof(double), hipMemcpyHostToDevice, c->stream));
    RMB_HIP(hipMemcpyAsync(c->st[6].p, force, (size_t)3 * ns * sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  RMB_HIP(hipMemcpyAsync(c->st[4].p, tgt, bt3, hipMemcpyHostToDevice, c->stream));
  RMB_HIP(hipMemcpyAsync(c->st[5].p, rad_t, bt1, hipMemcpyHostToDevice, c->stream));
  // sources == targets (same arrays, or equal contents): hand the device entry the SAME pointers, which selects its
  // symmetric path
  const bool same = ns == nt && ns > 0 && (src == tgt || !memcmp(src, tgt, (size_t)3 * ns * sizeof(double))) &&
                    (rad_s == rad_t || !memcmp(rad_s, rad_t, (size_t)ns * sizeof(double)));
  const double* tgt_d = same ? (const double*)c->st[2].p : (const double*)c->st[4].p;
  const double* radt_d = same ? (const double*)c->st[3].p : (const double*)c->st[5].p;
  if (int rc = rmb_mobility_source_target_device(c, ns, (const double*)c->st[2].p, (const double*)c->st[3].p, nt, tgt_d, radt_d,
                                                 (const double*)c->st[6].p, eta, L, wall, (double*)c->st[7].p))
    return rc;
  RMB_HIP(hipMemcpyAsync(out, c->st[7].p, bt3, hipMemcpyDeviceToHost, c->stream));
  RMB_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int rmb_pressure_stokeslet_device(rmb_ctx* c, long ns, const double* src_dev, long nt, const double* tgt_dev,
                                  const double* force_dev, const double* L, int wall, double* out_dev) {
  if (!c) return fail(RMB_ERR_ARG, "null context");
  if (ns < 0 || nt < 0) return fail(RMB_ERR_ARG, "negative size");
  if (wall != 0 && wall != 1) return fail(RMB_ERR_ARG, "wall must be 0 or 1");
  if (L && (L[0] > 0 || L[1] > 0 || L[2] > 0))
    return fail(RMB_ERR_ARG, "pressure: periodic_length must be zero (the reference's periodic branch divides by the unwrapped distance)");
  if (nt == 0) return 0;
  if (!out_dev || !tgt_dev || (ns > 0 && (!src_dev || !force_dev))) return fail(RMB_ERR_ARG, "null pointer");
  RMB_HIP(hipSetDevice(c->device));
  if (ns == 0) { RMB_HIP(hipMemsetAsync(out_dev, 0, (size_t)nt * sizeof(double), c->stream)); return 0; }
  rmb::AuxArgs a{};
  a.src = src_dev; a.tgt = tgt_dev; a.v0 = force_dev; a.v1 = nullptr; a.w = nullptr; a.out = out_dev;
  a.ns = ns; a.nt = nt; a.prefactor = 1.0 / (4.0 * M_PI); a.a2 = 0.0;
  return wall ? aux_launch<rmb::AUX_P_WALL>(c, a) : aux_launch<rmb::AUX_P_FREE>(c, a);
}

int rmb_double_layer_device(rmb_ctx* c, long ns, const double* src_dev, long nt, const double* tgt_dev,
                            const double* normals_dev, const double* vector_dev, const double* weights_dev, int wall,
                            double blob_radius, double* out_dev) {
  if (!c) return fail(RMB_ERR_ARG, "null context");
  if (ns < 0 || nt < 0) return fail(RMB_ERR_ARG, "negative size");
  if (wall != 0 && wall != 1) return fail(RMB_ERR_ARG, "wall must be 0 or 1");
  if (wall && blob_radius >= 0.0) return fail(RMB_ERR_ARG, "the RPY double layer is unbounded only (mobility_numba.py:2095)");
  if (nt == 0) return 0;
  if (!out_dev || !tgt_dev || (ns > 0 && (!src_dev || !normals_dev || !vector_dev || !weights_dev)))
    return fail(RMB_ERR_ARG, "null pointer");
  RMB_HIP(hipSetDevice(c->device));
  if (ns == 0) { RMB_HIP(hipMemsetAsync(out_dev, 0, (size_t)3 * nt * sizeof(double), c->stream)); return 0; }
  rmb::AuxArgs a{};
  a.src = src_dev; a.tgt = tgt_dev; a.v0 = normals_dev; a.v1 = vector_dev; a.w = weights_dev; a.out = out_dev;
  a.ns = ns; a.nt = nt; a.prefactor = -3.0 / (4.0 * M_PI); a.a2 = blob_radius >= 0.0 ? blob_radius * blob_radius : 0.0;
  if (blob_radius >= 0.0) return aux_launch<rmb::AUX_DL_RPY>(c, a);
  return wall ? aux_launch<rmb::AUX_DL_WALL>(c, a) : aux_launch<rmb::AUX_DL_FREE>(c, a);
}

int rmb_pressure_stokeslet(long ns, const double* src, long nt, const double* tgt, const double* force, const double* L,
                           int wall, double* out) {
  std::lock_guard<std::mutex> lk(g_default_mu);
  if (!g_default_ctx) {
    if (int rc = rmb_ctx_create(0, &g_default_ctx)) return rc;
  }
  rmb_ctx* c = g_default_ctx;
  if (ns < 0 || nt < 0) return fail(RMB_ERR_ARG, "negative size");
  if (nt == 0) return 0;
  if (!out || !tgt || (ns > 0 && (!src || !force))) return fail(RMB_ERR_ARG, "null pointer");
  RMB_HIP(hipSetDevice(c->device));
  const size_t b3s = (size_t)3 * ns * sizeof(double), b3t = (size_t)3 * nt * sizeof(double);
  const double* host[3] = {src, tgt, force};
  const size_t bytes[3] = {b3s, b3t, b3s};
  const int slot[3] = {2, 4, 6};
  const double* dev[3];
  if (int rc = aux_stage(c, 3, host, bytes, slot, dev)) return rc;
  if (int rc = c->st[7].reserve((size_t)nt * sizeof(double))) return rc;
  if (int rc = rmb_pressure_stokeslet_device(c, ns, dev[0], nt, dev[1], dev[2], L, wall, (double*)c->st[7].p)) return rc;
  RMB_HIP(hipMemcpyAsync(out, c->st[7].p, (size_t)nt * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  RMB_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int rmb_double_layer(long ns, const double* src, long nt, const double* tgt, const double* normals, const double* vector,
                     const double* weights, int wall, double blob_radius, double* out) {
  std::lock_guard<std::mutex> lk(g_default_mu);
  if (!g_default_ctx) {
    if (int rc = rmb_ctx_create(0, &g_default_ctx)) return rc;
  }
  rmb_ctx* c = g_default_ctx;
  if (ns < 0 || nt < 0) return fail(RMB_ERR_ARG, "negative size");
  if (nt == 0) return 0;
  if (!out || !tgt || (ns > 0 && (!src || !normals || !vector || !weights))) return fail(RMB_ERR_ARG, "null pointer");
  RMB_HIP(hipSetDevice(c->device));
  const size_t b3s = (size_t)3 * ns * sizeof(double), b3t = (size_t)3 * nt * sizeof(double);
  const double* host[5] = {src, tgt, normals, vector, weights};
  const size_t bytes[5] = {b3s, b3t, b3s, b3s, (size_t)ns * sizeof(double)};
  const int slot[5] = {2, 4, 6, 5, 3};
  const double* dev[5];
  if (int rc = aux_stage(c, 5, host, bytes, slot, dev)) return rc;
  if (int rc = c->st[7].reserve(b3t)) return rc;
  if (int rc = rmb_double_layer_device(c, ns, dev[0], nt, dev[1], dev[2], dev[3], dev[4], wall, blob_radius, (double*)c->st[7].p))
    return rc;
  RMB_HIP(hipMemcpyAsync(out, c->st[7].p, b3t, hipMemcpyDeviceToHost, c->stream));
  RMB_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int rmb_timing_collect(rmb_ctx* c, double* ms, int max_n) {
  if (!c || !ms || max_n < 0) return fail(RMB_ERR_ARG, "bad timing_collect arguments");
  if (hipSetDevice(c->device) != hipSuccess) return fail(RMB_ERR_HIP, "hipSetDevice failed");
  if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(RMB_ERR_HIP, "hipStreamSynchronize failed");
  int have = c->ev_count < kTimingRing ? c->ev_count : kTimingRing;
  if (have > max_n) have = max_n;
  // most recent `have` entries, oldest first
  for (int i = 0; i < have; ++i) {
    const int idx = (c->ev_count - have + i) % kTimingRing;
    float t = 0.f;
    if (hipEventElapsedTime(&t, c->ev0[idx], c->ev1[idx]) != hipSuccess) return fail(RMB_ERR_HIP, "hipEventElapsedTime failed");
    ms[i] = (double)t;
  }
  return have;
}

int rmb_wave_clock_collect(rmb_ctx* c, long long* stamps, long max_waves) {
  if (!c || !stamps || max_waves < 0) return fail(RMB_ERR_ARG, "bad wave_clock_collect arguments");
  RMB_HIP(hipSetDevice(c->device));
  RMB_HIP(hipStreamSynchronize(c->stream));
  long n = c->wave_clock_n < max_waves ? c->wave_clock_n : max_waves;
  if (n > 0 && c->wave_clock.p) RMB_HIP(hipMemcpy(stamps, c->wave_clock.p, (size_t)2 * n * sizeof(long long), hipMemcpyDeviceToHost));
  else n = 0;
  return (int)n;
}

int rmb_ubench_fp64_issue(rmb_ctx* c, int launches, double* g_wave_instr_per_s) {
  if (!c || !g_wave_instr_per_s || launches < 1) return fail(RMB_ERR_ARG, "bad ubench arguments");
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

int rmb_timing_reset(rmb_ctx* c) {
  if (!c) return fail(RMB_ERR_ARG, "null context");
  c->ev_count = 0;
  c->timing_launches = 0;
  return 0;
}

int rmb_last_launch(rmb_ctx* c, long* tiles, long* chunks, long* wgs) {
  if (!c) return fail(RMB_ERR_ARG, "null context");
  if (tiles) *tiles = c->last_tiles;
  if (chunks) *chunks = c->last_chunks;
  if (wgs) *wgs = c->last_wgs;
  return 0;
}

int rmb_ctx_synchronize(rmb_ctx* c) {
  if (!c) return fail(RMB_ERR_ARG, "null context");
  RMB_HIP(hipSetDevice(c->device));
  RMB_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

static int default_ctx(rmb_ctx** out) {
  if (!g_default_ctx) {
    if (int rc = rmb_ctx_create(0, &g_default_ctx)) return rc;
  }
  *out = g_default_ctx;
  return 0;
}

int rmb_default_ctx_set_option(const char* key, long value) {
  std::lock_guard<std::mutex> lk(g_default_mu);
  rmb_ctx* c;
  if (int rc = default_ctx(&c)) return rc;
  return rmb_ctx_set_option(c, key, value);
}

int rmb_mobility_oneshot(int kind, int wall, int in_plane, long n, const double* r, const double* vec,
                         const double* vec2, double eta, double a, const double* L, double* out) {
  std::lock_guard<std::mutex> lk(g_default_mu);
  rmb_ctx* c;
  if (int rc = default_ctx(&c)) return rc;
  if (int rc = rmb_set_positions(c, r, n, a, L, wall)) return rc;
  return rmb_matvec(c, kind, in_plane, vec, vec2, eta, out);
}

int rmb_forces_oneshot(long n, const double* r, const double* L, double eps, double b, double blob_radius, double* out) {
  std::lock_guard<std::mutex> lk(g_default_mu);
  rmb_ctx* c;
  if (int rc = default_ctx(&c)) return rc;
  if (int rc = rmb_set_positions(c, r, n, blob_radius, L, 0)) return rc;
  return rmb_blob_blob_force(c, eps, b, blob_radius, out);
}

}  // extern "C"
