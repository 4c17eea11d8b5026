// rmb_context.hip -- error state, context life cycle, streams, options, the HIP-event timing ring, schedule
// diagnostics and the library's default context (include/rmb_mobility.h: "library / device", "persistent context").
#include "rmb_internal.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace rmbi {

namespace { thread_local std::string g_err; }

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

int DevBuf::reserve(size_t bytes) {
  if (bytes <= cap) return 0;
  if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
  size_t want = bytes + bytes / 8 + 256;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) { p = nullptr; return fail(RMB_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e)); }
  cap = want;
  return 0;
}

int timing_begin(rmb_ctx* c, int* slot) {
  *slot = -1;
  if (!c->opt_timing) return 0;
  // "timing" = n > 1: bracket every n-th sweep only.  An event pair costs ~4-8 us of serialisation around a launch
  // (tools/experiments/exp_graph.py: 188 us per 1e4-blob step without events, 199 us with), so a throughput measurement samples.
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

std::mutex g_default_mu;
namespace {
rmb_ctx* g_default_ctx = nullptr;
int g_default_device = -1;      // -1 = RMB_DEVICE, or 0 (rmb_ctx_create); rmb_default_ctx_set_device() pins another
}

// Device of the default context (the stateless entry points) and of contexts created with device = -1: the
// environment variable RMB_DEVICE (an index into the visible devices), 0 when it is unset.
int default_device() {
  const char* e = getenv("RMB_DEVICE");
  if (!e || !*e) return 0;
  char* end = nullptr;
  const long v = strtol(e, &end, 10);
  return (end && *end == 0 && v >= 0) ? (int)v : -1;     // garbage -> -1 -> rmb_ctx_create reports "out of range"
}

int default_ctx(rmb_ctx** out) {
  if (!g_default_ctx) {
    if (int rc = rmb_ctx_create(g_default_device, &g_default_ctx)) return rc;
  }
  *out = g_default_ctx;
  return 0;
}

}  // namespace rmbi

using namespace rmbi;

extern "C" {

const char* rmb_version(void) { return "rmb_mobility 0.1 (gfx950)"; }
const char* rmb_last_error(void) { return rmbi::g_err.c_str(); }

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
  if (device == -1) {       // "the default device": RMB_DEVICE, or 0
    device = default_device();
    if (device < 0) return fail(RMB_ERR_ARG, "RMB_DEVICE must be a non-negative device index");
  }
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
  rmbi::gmres_release(c);
  c->wave_clock.release(); c->tile_bounds.release(); c->fpos.release(); c->fperm.release(); c->fsort_keys.release(); c->fsort_vals.release(); c->fsort_tmp.release(); c->fsort_box.release(); for (auto& b : c->st) b.release(); c->symbuf.release(); if (c->host_out) { (void)hipHostFree(c->host_out); c->host_out = nullptr; c->host_out_cap = 0; } if (c->host_in) { (void)hipHostFree(c->host_in); c->host_in = nullptr; c->host_in_cap = 0; } c->pos.release(); c->r_stage.release(); c->vec.release(); c->vec2.release(); c->out.release(); c->partial.release(); c->tmp3n.release(); c->det_ws.release(); c->krylov.release();
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
#ifdef RMB_DIAGNOSTICS
  // Diagnostics that make results WRONG ("skip_pairs") or change which kernel runs ("wave_clock") exist only in the
  // diagnostics build of the library (librmb_mobility_diag.so, tools/ only): the release library's option table does
  // not know them, so the boundary cannot be talked into a silent wrong answer.
  if (!strcmp(key, "wave_clock")) { c->opt_wave_clock = value; return 0; }
  if (!strcmp(key, "skip_pairs")) { c->opt_skip_pairs = value; return 0; }
#else
  if (!strcmp(key, "wave_clock") || !strcmp(key, "skip_pairs"))
    return fail(RMB_ERR_ARG, std::string("option \"") + key + "\" exists only in the diagnostics build (librmb_mobility_diag.so, RMB_DIAGNOSTICS=1)");
#endif
  if (!strcmp(key, "sym_pin")) { c->opt_sym_pin = value; return 0; }
  if (!strcmp(key, "precision")) {
    if (value != 32 && value != 64) return fail(RMB_ERR_ARG, "precision must be 32 or 64");
    c->opt_precision = value;
    return 0;
  }
  if (!strcmp(key, "force_cull")) { c->opt_force_cull = value ? 1 : 0; return 0; }
  if (!strcmp(key, "force_sort")) { c->opt_force_sort = value ? 1 : 0; return 0; }
  if (!strcmp(key, "force_precision")) {
    if (value != 0 && value != 32 && value != 64) return fail(RMB_ERR_ARG, "force_precision must be 0 (follow \"precision\"), 32 or 64");
    c->opt_force_precision = value;
    return 0;
  }
  if (!strcmp(key, "sym_fine_steps")) { c->opt_sym_fine_steps = value < 0 ? 0 : value; return 0; }
  if (!strcmp(key, "sym_coop")) {
    if (value < 0 || value > 2) return fail(RMB_ERR_ARG, "sym_coop must be 0 (never), 1 (launches below one resident round) or 2 (always)");
    c->opt_sym_coop = value;
    return 0;
  }
  if (!strcmp(key, "sym_chunk_steps")) { c->opt_sym_chunk_steps = value < 0 ? 0 : value; return 0; }
  if (!strcmp(key, "sym_two_targets")) { c->opt_sym_two_targets = value < 0 ? 0 : (value > 2 ? 2 : value); return 0; }
  if (!strcmp(key, "host_zero_copy_in")) { c->opt_host_zero_copy_in = value ? 1 : 0; return 0; }
  if (!strcmp(key, "gmres_fuse_pc")) { c->opt_gmres_fuse_pc = value ? 1 : 0; return 0; }
  if (!strcmp(key, "gmres_fuse_dots")) { c->opt_gmres_fuse_dots = value ? 1 : 0; return 0; }
  if (!strcmp(key, "krylov_low_sync")) { c->opt_krylov_low_sync = value ? 1 : 0; return 0; }
  if (!strcmp(key, "lanczos_fuse_finish")) { c->opt_lanczos_fuse_finish = value ? 1 : 0; return 0; }
  if (!strcmp(key, "host_zero_copy")) { c->opt_host_zero_copy = value < 0 ? 0 : value; return 0; }
  if (!strcmp(key, "sym_order")) { c->opt_sym_order = value ? 1 : 0; return 0; }
  if (!strcmp(key, "sym_xcd")) { c->opt_sym_xcd = value ? 1 : 0; return 0; }
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
      {"precision", &c->opt_precision}, {"force_precision", &c->opt_force_precision}, {"force_cull", &c->opt_force_cull}, {"force_sort", &c->opt_force_sort}, {"sym_oversub", &c->opt_sym_oversub}, {"sym_fine_steps", &c->opt_sym_fine_steps}, {"sym_coop", &c->opt_sym_coop}, {"sym_order", &c->opt_sym_order}, {"host_zero_copy", &c->opt_host_zero_copy}, {"host_zero_copy_in", &c->opt_host_zero_copy_in}, {"gmres_fuse_pc", &c->opt_gmres_fuse_pc}, {"gmres_fuse_dots", &c->opt_gmres_fuse_dots}, {"krylov_low_sync", &c->opt_krylov_low_sync}, {"lanczos_fuse_finish", &c->opt_lanczos_fuse_finish}, {"sym_two_targets", &c->opt_sym_two_targets}, {"sym_chunk_steps", &c->opt_sym_chunk_steps}, {"sym_xcd", &c->opt_sym_xcd},
      {"sym_min_steps", &c->opt_sym_min_steps}};
  // read-only: which kernel family the last product ran on (0 one-sided sweep, 1 symmetric per-wave, 2 deterministic
  // symmetric, 3 symmetric workgroup-cooperative)
  if (!strcmp(key, "last_path")) { *value = c->last_path; return 0; }
  if (!strcmp(key, "diagnostics_build")) {
#ifdef RMB_DIAGNOSTICS
    *value = 1;
#else
    *value = 0;
#endif
    return 0;
  }
  // read-only: a hash of the addresses of every device buffer the library owns for this context.  A captured hipGraph
  // of device-path calls holds those addresses by value; it stays valid exactly while this number is unchanged (a
  // buffer that grows is freed and reallocated, DevBuf::reserve).
  if (!strcmp(key, "buffers_signature")) {
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&h](const rmbi::DevBuf& b) { h = (h ^ (unsigned long long)(uintptr_t)b.p) * 1099511628211ull; };
    mix(c->pos); mix(c->r_stage); mix(c->vec); mix(c->vec2); mix(c->out); mix(c->partial); mix(c->tmp3n); mix(c->tile_bounds);
    mix(c->fpos); mix(c->fperm); mix(c->fsort_keys); mix(c->fsort_vals); mix(c->fsort_tmp); mix(c->fsort_box); mix(c->det_ws);
    for (const auto& b : c->st) mix(b);
    mix(c->wave_clock); mix(c->krylov); mix(c->symbuf);
    *value = (long)(h >> 1);
    return 0;
  }
  for (const auto& e : table)
    if (!strcmp(key, e.name)) { *value = *e.v; return 0; }
  return fail(RMB_ERR_ARG, std::string("unknown option: ") + key);
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

int rmb_ubench_fp64_issue(rmb_ctx* c, int launches, double* g_wave_instr_per_s) {
  if (!c || !g_wave_instr_per_s || launches < 1) return fail(RMB_ERR_ARG, "bad ubench arguments");
  return ubench_fp64_issue(c, launches, g_wave_instr_per_s);
}

int rmb_default_ctx_set_device(int device) {
  std::lock_guard<std::mutex> lk(g_default_mu);
  if (device < -1) return fail(RMB_ERR_ARG, "device must be an index, or -1 for RMB_DEVICE / 0");
  if (g_default_ctx && (device == g_default_ctx->device || (device == -1 && g_default_device == -1))) return 0;
  if (device >= 0) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return fail(RMB_ERR_NO_DEVICE, "no HIP device visible");
    if (device >= n) return fail(RMB_ERR_ARG, "device index out of range");
  }
  if (g_default_ctx) { rmb_ctx_destroy(g_default_ctx); g_default_ctx = nullptr; }   // re-created on the next call
  g_default_device = device;
  return 0;
}

int rmb_default_ctx_set_option(const char* key, long value) {
  std::lock_guard<std::mutex> lk(g_default_mu);
  rmb_ctx* c;
  if (int rc = default_ctx(&c)) return rc;
  return rmb_ctx_set_option(c, key, value);
}

}  // extern "C"
