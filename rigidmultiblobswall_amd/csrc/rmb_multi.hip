// rmb_multi.hip -- single-process multi-device engine (include/rmb_mobility.h, "rmb_multi_*").
//
// The reference's callers are ONE Python process (multi_bodies/multi_bodies.py:233-287 picks a function,
// :445 / :599 call it; mobility/mobility.py:222-252 is the call shape), so "use the whole node" has to live behind the
// same call.  One engine owns G shards; shard g = one rmb_ctx on devices[g] with its own stream.  A product is
//   1. inputs -> every device (host: one pinned staging copy, then G async uploads; device: each shard pulls the vectors
//      from devices[0] on its own stream with hipMemcpyPeerAsync -- caller-owned memory is only touched by runtime copies),
//   2. shard g sweeps pair shard g of G (rmb_matvec_pairshard_device & co.: each unordered pair once, both blobs updated)
//      into a full-length partial on its device,
//   3. device g adds slice g of the G partials IN FIXED ORDER (h = 0 .. G-1) through peer-mapped reads of the other
//      devices' partials (memory this engine allocated after enabling peer access) and hands the slice to where the
//      result is wanted: a peer copy to devices[0] for the *_device entry points, a download to pinned memory otherwise.
// With "deterministic" = 2 the partials are bit-reproducible and so is the sum.  Option "reduce" = 1 replaces step 3 by a
// grouped RCCL all-reduce (ncclCommInitAll; librccl is dlopen()ed on first use so the library does not link it).
// The same device may be listed several times (rehearsal of the G-device code path on one GPU: shards then share the
// device, the peer reads are local reads).
#include "rmb_internal.h"

#include <dlfcn.h>

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace {

using rmbi::DevBuf;
using rmbi::fail;

constexpr int kMaxShards = 16;
constexpr int kMaxVec = 4;

struct ReduceArgs {
  const double* part[kMaxShards];   // partial outputs of every shard: n_out vectors of `len` doubles each
  double* out[kMaxVec];             // where vector v of the result goes (indexed like the partials)
  int n_shards, n_out;
  long len;                         // 3n
  long lo, hi;                      // this launch's slice of [0, len)
};

// Peer-mapped memory is read and written with system-scope accesses (sc0 sc1 on gfx950: served by the owning device's
// memory, never by a line this device's L2 kept from the previous product).  Ordering between devices is by events at
// kernel boundaries; the scope only keeps a stale cached copy out of the picture.
__device__ __forceinline__ double peer_load(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void peer_store(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// out[v][i] = sum over shards, ascending shard index: a fixed order, so bit-reproducible partials give a
// bit-reproducible product.  part[h] may live on a peer device (peer-mapped loads), out[v] too (peer-mapped stores).
__global__ __launch_bounds__(256) void reduce_slices_kernel(const ReduceArgs a) {
  const long i = a.lo + (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.hi) return;
  for (int v = 0; v < a.n_out; ++v) {
    double s = peer_load(&a.part[0][v * a.len + i]);
    for (int h = 1; h < a.n_shards; ++h) s += peer_load(&a.part[h][v * a.len + i]);
    peer_store(&a.out[v][i], s);
  }
}

struct Shard {
  int device = 0;
  rmb_ctx* ctx = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev_partial = nullptr;   // this shard's partial is complete
  hipEvent_t ev_reduced = nullptr;   // this shard's slice of the result is written
  DevBuf in[kMaxVec];                // local copies of the input vectors
  DevBuf part;                       // partial outputs [n_out][3n]
  DevBuf r_stage;                    // raw positions
  DevBuf gather;                     // staged path (no peer access): slice of every other shard's partial
  DevBuf red;                        // staged path / host entry: this shard's reduced slice
  int rc = 0;                        // status of this shard's part of the current job (worker thread -> caller)
  std::string err;
};

// RCCL, resolved at run time (torch bundles its own librccl.so under the same SONAME; dlopen returns the copy that is
// already mapped, so a process never ends up with two).
struct Rccl {
  void* handle = nullptr;
  int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
  int (*CommDestroy)(void* comm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*AllReduce)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t s) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok() const { return handle && CommInitAll && CommDestroy && GroupStart && GroupEnd && AllReduce; }
};

// What one product is, seen by the engine: inputs (3n doubles each), outputs (3n each), and how shard g of G launches
// its pair shard on its own context with LOCAL input pointers into a LOCAL full-length partial.
struct Product {
  int n_in = 0, n_out = 0;
  int (*launch)(rmb_ctx* c, const Product& p, const double* const* in, double* const* out, long shard, long nshards) = nullptr;
  int kind = 0, op = 0, in_plane = 0;
  double eta = 1.0, eps = 0.0, b = 1.0, a = 0.0;
};

// One call in flight: the product, where its inputs are and where its results go.
//   device entry: in[v] / out[v] are pointers on devices[0]; results are made visible to m->primary
//   host entry  : in[v] = staged copy in m->pinned, results land at m->host_out + v * len (out[] unused)
struct Job {
  Product p;
  bool host = false;
  const double* in[kMaxVec] = {nullptr, nullptr, nullptr, nullptr};
  double* out[kMaxVec] = {nullptr, nullptr, nullptr, nullptr};
};

}  // namespace

struct rmb_multi {
  std::vector<Shard> sh;
  hipStream_t primary = nullptr;    // the caller's stream on devices[0] (the *_device entry points are ordered on it)
  hipEvent_t ev_in = nullptr;       // inputs of the current call are ready (recorded on `primary`)
  hipEvent_t ev_done = nullptr;     // everything the last *_device call enqueued (recorded on `primary`)
  bool done_recorded = false;
  bool peer = true;                 // every pair of distinct devices has peer access enabled
  bool distinct = true;             // no device listed twice (RCCL needs that)
  bool force_remote = false;        // RMB_MULTI_FORCE_REMOTE=1 (tests): treat every shard but the first as if it sat on another
                                    // device than devices[0], so a one-GPU rehearsal runs the copies a node runs
  long opt_reduce = 0;              // 0 = fixed-order slice reduction (peer reads / staged copies), 1 = RCCL all-reduce
  long n = 0;
  bool have_positions = false;
  void* pinned = nullptr;           // host staging of the synchronous entry points: inputs first, results behind them
  size_t pinned_cap = 0;
  double* host_out = nullptr;       // where in `pinned` the current host call's results go
  Rccl rccl;
  std::vector<void*> comms;
  // worker threads (one per shard when there are several): job hand-over, phase barrier, completion
  std::vector<std::thread> workers;
  std::mutex mu, mu_done;
  std::condition_variable cv, cv_done;
  std::atomic<unsigned long> job_seq{0};
  std::atomic<int> quit{0}, failed{0}, done_count{0}, bar_count{0};
  std::atomic<unsigned> bar_gen{0};
  Job job;
};

namespace {

int pinned_reserve(rmb_multi* m, size_t bytes) {
  if (bytes <= m->pinned_cap) return 0;
  if (m->pinned) { (void)hipHostFree(m->pinned); m->pinned = nullptr; m->pinned_cap = 0; }
  const size_t want = bytes + bytes / 8 + 4096;
  RMB_HIP(hipHostMalloc(&m->pinned, want, hipHostMallocPortable));
  m->pinned_cap = want;
  return 0;
}

// Every shard's partial for the largest product (kMaxVec outputs of 3n doubles).  Called from rmb_multi_set_positions*:
// if a buffer has to grow, everything the previous calls left in flight on ANY device is drained first, because peers
// read these buffers.
int reserve_partials(rmb_multi* m, long n) {
  const size_t need = (size_t)kMaxVec * 3 * (size_t)n * sizeof(double);
  bool grow = false;
  for (const Shard& s : m->sh) grow = grow || need > s.part.cap;
  if (!grow) return 0;
  for (Shard& s : m->sh) {
    RMB_HIP(hipSetDevice(s.device));
    RMB_HIP(hipStreamSynchronize(s.stream));
  }
  if (m->done_recorded) { RMB_HIP(hipEventSynchronize(m->ev_done)); }
  for (Shard& s : m->sh) {
    RMB_HIP(hipSetDevice(s.device));
    if (int rc = s.part.reserve(need)) return rc;
  }
  return 0;
}

// The engine sets the calling thread's current device as it goes (devices[g] for every shard); every rmb_multi_* entry
// point puts back what was current on entry, on every return path -- an in-process torch caller would otherwise find its
// implicit-device allocations and current stream on another GPU.
struct DeviceGuard {
  int saved = -1;
  DeviceGuard() { if (hipGetDevice(&saved) != hipSuccess) { saved = -1; (void)hipGetLastError(); } }
  ~DeviceGuard() { if (saved >= 0) (void)hipSetDevice(saved); }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

int check_multi(rmb_multi* m, bool need_positions) {
  if (!m) return fail(RMB_ERR_ARG, "null multi-device engine");
  if (need_positions && !m->have_positions) return fail(RMB_ERR_STATE, "rmb_multi_set_positions has not been called");
  return 0;
}

// does shard g share the memory of devices[0] (the caller's vectors are then used in place)?
bool local_to_primary(const rmb_multi* m, int g) {
  return m->sh[g].device == m->sh[0].device && !(m->force_remote && g > 0);
}

void slice_of(long len, int g, int G, long* lo, long* hi) {
  *lo = (long)((__int128)len * g / G);
  *hi = (long)((__int128)len * (g + 1) / G);
}

int rccl_load(rmb_multi* m) {
  if (m->rccl.ok()) return 0;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* nm : names) {
    m->rccl.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (m->rccl.handle) break;
  }
  if (!m->rccl.handle) return fail(RMB_ERR_STATE, std::string("\"reduce\" = 1 needs librccl.so: ") + dlerror());
  void* h = m->rccl.handle;
  *(void**)&m->rccl.CommInitAll = dlsym(h, "ncclCommInitAll");
  *(void**)&m->rccl.CommDestroy = dlsym(h, "ncclCommDestroy");
  *(void**)&m->rccl.GroupStart = dlsym(h, "ncclGroupStart");
  *(void**)&m->rccl.GroupEnd = dlsym(h, "ncclGroupEnd");
  *(void**)&m->rccl.AllReduce = dlsym(h, "ncclAllReduce");
  *(void**)&m->rccl.GetErrorString = dlsym(h, "ncclGetErrorString");
  if (!m->rccl.ok()) return fail(RMB_ERR_STATE, "librccl.so lacks the ncclCommInitAll / ncclAllReduce entry points");
  return 0;
}

int rccl_fail(rmb_multi* m, const char* what, int rc) {
  return fail(RMB_ERR_HIP, std::string(what) + ": " + (m->rccl.GetErrorString ? m->rccl.GetErrorString(rc) : "RCCL error"));
}

int rccl_comms(rmb_multi* m) {
  if (!m->comms.empty()) return 0;
  if (!m->distinct) return fail(RMB_ERR_ARG, "\"reduce\" = 1 (RCCL) needs distinct devices: a communicator holds one rank per device");
  if (int rc = rccl_load(m)) return rc;
  std::vector<int> devs;
  for (const Shard& s : m->sh) devs.push_back(s.device);
  m->comms.assign(devs.size(), nullptr);
  const int rc = m->rccl.CommInitAll(m->comms.data(), (int)devs.size(), devs.data());
  if (rc != 0) { m->comms.clear(); return rccl_fail(m, "ncclCommInitAll", rc); }
  return 0;
}

int launch_kind(rmb_ctx* c, const Product& p, const double* const* in, double* const* out, long g, long G) {
  return rmbi::matvec_pairshard_impl(c, p.kind, p.in_plane, in[0], p.eta, out[0], g, G);   // in_plane: free surface only (product_of_kind)
}
int launch_op(rmb_ctx* c, const Product& p, const double* const* in, double* const* out, long g, long G) {
  return rmbi::matvec_op_impl(c, p.op, p.in_plane, p.n_in, in, p.n_out, out, p.eta, g, G);
}
int launch_force(rmb_ctx* c, const Product& p, const double* const*, double* const* out, long g, long G) {
  return rmbi::force_device_impl(c, p.eps, p.b, p.a, out[0], nullptr, g, G);
}

// Phase 1 of shard g: inputs -> this device, sweep pair shard g of G into the local partial, record ev_partial.
int shard_sweep(rmb_multi* m, int g, const Job& job) {
  const int G = (int)m->sh.size();
  const long len = 3 * m->n;
  const Product& p = job.p;
  Shard& s = m->sh[g];
  Shard& s0 = m->sh[0];
  RMB_HIP(hipSetDevice(s.device));
  const double* in_local[kMaxVec] = {nullptr, nullptr, nullptr, nullptr};
  if (!job.host) RMB_HIP(hipStreamWaitEvent(s.stream, m->ev_in, 0));
  for (int v = 0; v < p.n_in; ++v) {
    if (!job.host && local_to_primary(m, g)) { in_local[v] = job.in[v]; continue; }     // same memory: no copy
    if (int rc = s.in[v].reserve((size_t)len * sizeof(double))) return rc;
    if (job.host) {
      RMB_HIP(hipMemcpyAsync(s.in[v].p, job.in[v], (size_t)len * sizeof(double), hipMemcpyHostToDevice, s.stream));
    } else {
      // The caller's vectors were allocated by the caller (torch's allocator, say), possibly before this engine enabled
      // peer access: only the runtime's own copy is safe on them.  Peer-mapped loads are kept for memory the engine
      // allocated itself after enabling access (the partials, shard_reduce).
      RMB_HIP(hipMemcpyPeerAsync(s.in[v].p, s.device, job.in[v], s0.device, (size_t)len * sizeof(double), s.stream));
    }
    in_local[v] = (const double*)s.in[v].p;
  }
  // `part` is the one buffer OTHER devices read (peer-mapped loads of reduce_slices_kernel): it is sized for kMaxVec
  // outputs when the positions are set (reserve_partials, after the previous job has drained everywhere) and never
  // reallocated by a product -- a hipFree here could pull it from under a peer's reduction of the previous product.
  if ((size_t)p.n_out * len * sizeof(double) > s.part.cap) return fail(RMB_ERR_STATE, "engine partial buffer smaller than the product (internal)");
  double* outs[kMaxVec];
  for (int v = 0; v < p.n_out; ++v) outs[v] = (double*)s.part.p + v * len;
  if (int rc = p.launch(s.ctx, p, in_local, outs, g, G)) return rc;
  RMB_HIP(hipEventRecord(s.ev_partial, s.stream));
  return 0;
}

// Phase 2 of shard g (every shard has RECORDED its ev_partial by now): slice g of the sum, stored where it is wanted.
int shard_reduce(rmb_multi* m, int g, const Job& job) {
  const int G = (int)m->sh.size();
  const long len = 3 * m->n;
  const Product& p = job.p;
  Shard& s = m->sh[g];
  RMB_HIP(hipSetDevice(s.device));
  for (int h = 0; h < G; ++h)
    if (h != g) RMB_HIP(hipStreamWaitEvent(s.stream, m->sh[h].ev_partial, 0));
  long lo, hi;
  slice_of(len, g, G, &lo, &hi);
  const long cnt = hi - lo;
  ReduceArgs ra;
  ra.n_shards = G; ra.n_out = p.n_out; ra.len = len; ra.lo = lo; ra.hi = hi;
  // Only devices[0] itself stores straight into the caller's result (caller-owned memory: see shard_sweep); the other
  // devices reduce into a buffer of their own and hand the slice over with the runtime's peer copy.
  const bool direct = !job.host && local_to_primary(m, g);
  for (int h = 0; h < G; ++h) ra.part[h] = (const double*)m->sh[h].part.p;
  if (!m->peer && cnt > 0) {
    // no peer access: bring slice g of every other device's partial here with copies the runtime routes itself
    if (int rc = s.gather.reserve((size_t)G * p.n_out * len * sizeof(double))) return rc;   // indexed like a partial: no offset arithmetic in the kernel
    for (int h = 0; h < G; ++h) {
      if (m->sh[h].device == s.device) continue;
      double* base = (double*)s.gather.p + (size_t)h * p.n_out * len;
      for (int v = 0; v < p.n_out; ++v)
        RMB_HIP(hipMemcpyPeerAsync(base + v * len + lo, s.device, (const double*)m->sh[h].part.p + v * len + lo, m->sh[h].device,
                                   (size_t)cnt * sizeof(double), s.stream));
      ra.part[h] = base;
    }
  }
  if (direct) {
    for (int v = 0; v < p.n_out; ++v) ra.out[v] = job.out[v];
  } else {
    if (int rc = s.red.reserve((size_t)p.n_out * len * sizeof(double))) return rc;
    for (int v = 0; v < p.n_out; ++v) ra.out[v] = (double*)s.red.p + v * len;
  }
  if (cnt > 0) {
    hipLaunchKernelGGL(reduce_slices_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s.stream, ra);
    RMB_HIP(hipGetLastError());
    for (int v = 0; v < p.n_out; ++v) {
      if (job.host)
        RMB_HIP(hipMemcpyAsync(m->host_out + v * len + lo, ra.out[v] + lo, (size_t)cnt * sizeof(double), hipMemcpyDeviceToHost, s.stream));
      else if (!direct)
        RMB_HIP(hipMemcpyPeerAsync(job.out[v] + lo, m->sh[0].device, ra.out[v] + lo, s.device, (size_t)cnt * sizeof(double), s.stream));
    }
  }
  RMB_HIP(hipEventRecord(s.ev_reduced, s.stream));
  if (job.host) RMB_HIP(hipStreamSynchronize(s.stream));     // synchronous entry: the shards wait in parallel
  return 0;
}

// "reduce" = 1: RCCL all-reduce of the partials in place (one grouped call from the calling thread), then the primary
// shard hands the sum over.  Every shard has recorded its ev_partial; RCCL orders itself on the shard streams.
int rccl_reduce(rmb_multi* m, const Job& job) {
  const int G = (int)m->sh.size();
  const long len = 3 * m->n;
  const Product& p = job.p;
  if (int rc = rccl_comms(m)) return rc;
  if (int rc = m->rccl.GroupStart()) return rccl_fail(m, "ncclGroupStart", rc);
  for (int g = 0; g < G; ++g) {
    Shard& s = m->sh[g];
    const int rc = m->rccl.AllReduce(s.part.p, s.part.p, (size_t)p.n_out * len, /*ncclFloat64*/ 8, /*ncclSum*/ 0, m->comms[g], s.stream);
    if (rc != 0) { (void)m->rccl.GroupEnd(); return rccl_fail(m, "ncclAllReduce", rc); }
  }
  if (int rc = m->rccl.GroupEnd()) return rccl_fail(m, "ncclGroupEnd", rc);
  Shard& s0 = m->sh[0];
  RMB_HIP(hipSetDevice(s0.device));
  for (int v = 0; v < p.n_out; ++v) {
    if (!job.host) RMB_HIP(hipMemcpyAsync(job.out[v], (double*)s0.part.p + v * len, (size_t)len * sizeof(double), hipMemcpyDeviceToDevice, s0.stream));
    else RMB_HIP(hipMemcpyAsync(m->host_out + v * len, (double*)s0.part.p + v * len, (size_t)len * sizeof(double), hipMemcpyDeviceToHost, s0.stream));
  }
  for (int g = 0; g < G; ++g) {
    RMB_HIP(hipSetDevice(m->sh[g].device));
    RMB_HIP(hipEventRecord(m->sh[g].ev_reduced, m->sh[g].stream));
    if (job.host) RMB_HIP(hipStreamSynchronize(m->sh[g].stream));
  }
  return 0;
}

// ---- executors ---------------------------------------------------------------------------------------------------
// Enqueueing one shard's part of a product is ~25 us of HIP calls (waits, two or three launches, records); issued from
// one thread that is 25 us x G before the last device even starts (tools/experiments/exp_multi_engine.py).  With more
// than one shard every shard therefore has a worker thread that issues its own calls; the calling thread publishes the
// job and waits.  Between the two phases all workers meet at a barrier: hipStreamWaitEvent refers to the event's most
// recent RECORD, so every ev_partial has to be recorded before any shard waits on it.

void spin_pause() {
#if defined(__x86_64__)
  __builtin_ia32_pause();
#endif
}

// every worker calls this once per phase; returns when all `n` have arrived
void workers_barrier(rmb_multi* m) {
  const int n = (int)m->sh.size();
  const unsigned gen = m->bar_gen.load(std::memory_order_acquire);
  if (m->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
    m->bar_count.store(0, std::memory_order_relaxed);
    m->bar_gen.fetch_add(1, std::memory_order_release);
  } else {
    int spins = 0;
    while (m->bar_gen.load(std::memory_order_acquire) == gen) {
      if (++spins < 4096) spin_pause(); else std::this_thread::yield();
    }
  }
}

void worker_main(rmb_multi* m, int g) {
  (void)hipSetDevice(m->sh[g].device);
  unsigned long seen = 0;
  for (;;) {
    {
      // short spin first (a Krylov loop calls every few hundred microseconds), then sleep
      int spins = 0;
      while (m->job_seq.load(std::memory_order_acquire) == seen && !m->quit.load(std::memory_order_acquire) && ++spins < 20000) spin_pause();
      if (m->job_seq.load(std::memory_order_acquire) == seen && !m->quit.load(std::memory_order_acquire)) {
        std::unique_lock<std::mutex> lk(m->mu);
        m->cv.wait(lk, [&] { return m->job_seq.load(std::memory_order_acquire) != seen || m->quit.load(std::memory_order_acquire); });
      }
    }
    if (m->quit.load(std::memory_order_acquire)) return;
    seen = m->job_seq.load(std::memory_order_acquire);
    const Job& job = m->job;
    int rc = shard_sweep(m, g, job);
    if (rc != 0) { m->sh[g].rc = rc; m->sh[g].err = rmb_last_error(); m->failed.store(1, std::memory_order_release); }
    workers_barrier(m);
    if (!m->failed.load(std::memory_order_acquire) && m->opt_reduce == 0) {
      rc = shard_reduce(m, g, job);
      if (rc != 0) { m->sh[g].rc = rc; m->sh[g].err = rmb_last_error(); m->failed.store(1, std::memory_order_release); }
    }
    if (m->done_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (int)m->sh.size()) {
      std::lock_guard<std::mutex> lk(m->mu_done);
      m->cv_done.notify_one();
    }
  }
}

int run_shards(rmb_multi* m, const Job& job) {
  const int G = (int)m->sh.size();
  if (m->workers.empty()) {
    // one shard, or RMB_MULTI_THREADS=0: everything from the calling thread
    for (int g = 0; g < G; ++g)
      if (int rc = shard_sweep(m, g, job)) return rc;
    if (m->opt_reduce == 1) return rccl_reduce(m, job);
    for (int g = 0; g < G; ++g)
      if (int rc = shard_reduce(m, g, job)) return rc;
    return 0;
  }
  m->job = job;
  m->failed.store(0, std::memory_order_relaxed);
  m->done_count.store(0, std::memory_order_relaxed);
  for (Shard& s : m->sh) s.rc = 0;
  {
    std::lock_guard<std::mutex> lk(m->mu);
    m->job_seq.fetch_add(1, std::memory_order_release);
  }
  m->cv.notify_all();
  {
    int spins = 0;
    while (m->done_count.load(std::memory_order_acquire) != G && ++spins < 200000) spin_pause();
    if (m->done_count.load(std::memory_order_acquire) != G) {
      std::unique_lock<std::mutex> lk(m->mu_done);
      m->cv_done.wait(lk, [&] { return m->done_count.load(std::memory_order_acquire) == G; });
    }
  }
  if (m->failed.load(std::memory_order_acquire))
    for (Shard& s : m->sh)
      if (s.rc != 0) return fail(s.rc, s.err);
  if (m->opt_reduce == 1) return rccl_reduce(m, job);
  return 0;
}

// device entry: order the shards after the caller's stream, run, order the caller's stream after the shards
int run_device(rmb_multi* m, const Product& p, const double* const* in_dev, double* const* out_dev) {
  Shard& s0 = m->sh[0];
  RMB_HIP(hipSetDevice(s0.device));
  RMB_HIP(hipEventRecord(m->ev_in, m->primary));
  Job job;
  job.p = p; job.host = false;
  for (int v = 0; v < p.n_in; ++v) job.in[v] = in_dev[v];
  for (int v = 0; v < p.n_out; ++v) job.out[v] = out_dev[v];
  if (int rc = run_shards(m, job)) return rc;
  RMB_HIP(hipSetDevice(s0.device));
  for (Shard& s : m->sh) RMB_HIP(hipStreamWaitEvent(m->primary, s.ev_reduced, 0));
  RMB_HIP(hipEventRecord(m->ev_done, m->primary));
  m->done_recorded = true;
  return 0;
}

// host entry: synchronous, like the reference's wrappers
int run_host(rmb_multi* m, const Product& p, const double* const* in_host, double* const* out_host) {
  const long len = 3 * m->n;
  const size_t vb = (size_t)len * sizeof(double);
  if (int rc = pinned_reserve(m, (size_t)(p.n_in + p.n_out) * vb)) return rc;
  m->host_out = (double*)m->pinned + (size_t)p.n_in * len;
  // everything a previous *_device call left in flight must be through before the partials are re-used
  if (m->done_recorded) { RMB_HIP(hipEventSynchronize(m->ev_done)); }
  Job job;
  job.p = p; job.host = true;
  for (int v = 0; v < p.n_in; ++v) {
    memcpy((char*)m->pinned + v * vb, in_host[v], vb);
    job.in[v] = (const double*)((char*)m->pinned + v * vb);
  }
  if (int rc = run_shards(m, job)) return rc;      // every shard stream is synchronised when this returns
  for (int v = 0; v < p.n_out; ++v) memcpy(out_host[v], m->host_out + v * len, vb);
  return 0;
}

int product_of_kind(rmb_multi* m, int kind, int in_plane, bool have_vec2, double eta, Product* p) {
  if (kind < 0 || kind >= rmb::KIND_COUNT) return fail(RMB_ERR_ARG, "kind must be 0..5");
  if (!(eta > 0.0)) return fail(RMB_ERR_ARG, "eta must be positive");
  if (kind == rmb::KIND_TT_TR && !have_vec2) return fail(RMB_ERR_ARG, "RMB_TT_TR needs vec2 (torque)");
  p->eta = eta; p->in_plane = in_plane ? 1 : 0; p->kind = kind;
  p->n_in = kind == rmb::KIND_TT_TR ? 2 : 1;
  p->n_out = 1;
  if (kind == rmb::KIND_TT_TR) { p->op = RMB_OP_VELOCITY_FROM_FORCE_TORQUE; p->launch = launch_op; }
  else if (in_plane && kind <= rmb::KIND_RR) { p->op = RMB_OP_TT_MULTI + kind; p->launch = launch_op; }   // row / column mask of the symmetric block
  else p->launch = launch_kind;      // RMB_TT_FREE_SURFACE keeps its in_plane mask (matvec_pairshard_impl), like the single context
  (void)m;
  return 0;
}

int product_of_op(int op, int in_plane, int n_in, int n_out, double eta, Product* p) {
  if (!(eta > 0.0)) return fail(RMB_ERR_ARG, "eta must be positive");
  if (n_in < 1 || n_in > kMaxVec || n_out < 1 || n_out > kMaxVec) return fail(RMB_ERR_ARG, "an operation takes 1..4 vectors");
  p->op = op; p->in_plane = in_plane ? 1 : 0; p->n_in = n_in; p->n_out = n_out; p->eta = eta; p->launch = launch_op;
  return 0;    // the shard launch checks op / counts (matvec_op_impl)
}

}  // namespace

extern "C" {

int rmb_multi_create(const int* devices, int n_dev, rmb_multi** out) {
  DeviceGuard restore_device_;
  if (!out) return fail(RMB_ERR_ARG, "null engine out pointer");
  if (!devices || n_dev < 1 || n_dev > kMaxShards) return fail(RMB_ERR_ARG, "device list must hold 1..16 entries");
  int n_vis = 0;
  hipError_t e = hipGetDeviceCount(&n_vis);
  if (e != hipSuccess || n_vis == 0) return fail(RMB_ERR_NO_DEVICE, std::string("no HIP device visible (") + hipGetErrorString(e) + ")");
  for (int g = 0; g < n_dev; ++g)
    if (devices[g] < 0 || devices[g] >= n_vis) return fail(RMB_ERR_ARG, "device index out of range");
  rmb_multi* m = new rmb_multi();
  m->sh.resize(n_dev);
  int rc = 0;
  for (int g = 0; g < n_dev && rc == 0; ++g) {
    Shard& s = m->sh[g];
    s.device = devices[g];
    for (int h = 0; h < g; ++h) if (devices[h] == devices[g]) m->distinct = false;
    if ((rc = rmb_ctx_create(s.device, &s.ctx))) break;
    if (hipSetDevice(s.device) != hipSuccess || hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&s.ev_partial, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s.ev_reduced, hipEventDisableTiming) != hipSuccess) {
      rc = fail(RMB_ERR_HIP, "stream / event creation failed");
      break;
    }
    rc = rmb_ctx_set_stream(s.ctx, s.stream);
  }
  if (rc == 0) {
    if (hipSetDevice(m->sh[0].device) != hipSuccess || hipEventCreateWithFlags(&m->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_done, hipEventDisableTiming) != hipSuccess)
      rc = fail(RMB_ERR_HIP, "event creation failed");
  }
  if (rc == 0) {
    // peer access between every pair of distinct devices (xGMI on an MI355X node); without it the engine stages
    // through copies.  RMB_MULTI_NO_PEER=1 forces the staged path (rehearsal of that path on any box).
    const char* np = getenv("RMB_MULTI_NO_PEER");
    if (np && *np && *np != '0') m->peer = false;
    const char* fr = getenv("RMB_MULTI_FORCE_REMOTE");
    if (fr && *fr && *fr != '0') m->force_remote = true;
    for (int g = 0; g < n_dev && m->peer; ++g)
      for (int h = 0; h < n_dev && m->peer; ++h) {
        const int a = devices[g], b = devices[h];
        if (a == b) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can) { m->peer = false; break; }
        if (hipSetDevice(a) != hipSuccess) { m->peer = false; break; }
        const hipError_t pe = hipDeviceEnablePeerAccess(b, 0);
        if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) m->peer = false;
        (void)hipGetLastError();
      }
  }
  if (rc != 0) { rmb_multi_destroy(m); return rc; }
  // one worker thread per shard issues that shard's HIP calls (RMB_MULTI_THREADS=0: everything from the calling thread)
  const char* th = getenv("RMB_MULTI_THREADS");
  if (n_dev > 1 && !(th && *th == '0')) {
    try {
      for (int g = 0; g < n_dev; ++g) m->workers.emplace_back(worker_main, m, g);
    } catch (...) {
      rmb_multi_destroy(m);
      return fail(RMB_ERR_STATE, "could not start the shard worker threads");
    }
  }
  *out = m;
  return 0;
}

int rmb_multi_destroy(rmb_multi* m) {
  DeviceGuard restore_device_;
  if (!m) return 0;
  if (!m->workers.empty()) {
    {
      std::lock_guard<std::mutex> lk(m->mu);
      m->quit.store(1, std::memory_order_release);
    }
    m->cv.notify_all();
    for (std::thread& t : m->workers) if (t.joinable()) t.join();
    m->workers.clear();
  }
  for (Shard& s : m->sh) {
    (void)hipSetDevice(s.device);
    if (s.stream) (void)hipStreamSynchronize(s.stream);
  }
  if (!m->comms.empty() && m->rccl.CommDestroy)
    for (void* c : m->comms) if (c) (void)m->rccl.CommDestroy(c);
  for (Shard& s : m->sh) {
    (void)hipSetDevice(s.device);
    if (s.ctx) { s.ctx->stream = nullptr; rmb_ctx_destroy(s.ctx); }     // its stream goes next: forget the handle first
    for (auto& b : s.in) b.release();
    s.part.release(); s.r_stage.release(); s.gather.release(); s.red.release();
    if (s.ev_partial) (void)hipEventDestroy(s.ev_partial);
    if (s.ev_reduced) (void)hipEventDestroy(s.ev_reduced);
    if (s.stream) (void)hipStreamDestroy(s.stream);
  }
  if (m->ev_in) (void)hipEventDestroy(m->ev_in);
  if (m->ev_done) (void)hipEventDestroy(m->ev_done);
  if (m->pinned) (void)hipHostFree(m->pinned);
  delete m;
  return 0;
}

int rmb_multi_n_shards(rmb_multi* m) { return m ? (int)m->sh.size() : 0; }

int rmb_multi_shard_ctx(rmb_multi* m, int shard, rmb_ctx** ctx) {
  if (!m || !ctx || shard < 0 || shard >= (int)m->sh.size()) return fail(RMB_ERR_ARG, "bad engine / shard index");
  *ctx = m->sh[shard].ctx;
  return 0;
}

int rmb_multi_set_stream(rmb_multi* m, void* hip_stream) {
  DeviceGuard restore_device_;
  if (int rc = check_multi(m, false)) return rc;
  const hipStream_t next = (hipStream_t)hip_stream;
  if (next != m->primary) {
    // the new stream continues after what the last call enqueued; the event was recorded while the previous stream was
    // alive, so nothing here touches the old handle (it may have been destroyed since)
    if (m->done_recorded) {
      RMB_HIP(hipSetDevice(m->sh[0].device));
      RMB_HIP(hipStreamWaitEvent(next, m->ev_done, 0));
    }
    m->primary = next;
  }
  return 0;
}

int rmb_multi_set_option(rmb_multi* m, const char* key, long value) {
  DeviceGuard restore_device_;
  if (int rc = check_multi(m, false)) return rc;
  if (!key) return fail(RMB_ERR_ARG, "null key");
  if (!strcmp(key, "reduce")) {
    if (value != 0 && value != 1) return fail(RMB_ERR_ARG, "reduce must be 0 (fixed-order slices) or 1 (RCCL all-reduce)");
    if (value == 1 && !m->distinct) return fail(RMB_ERR_ARG, "\"reduce\" = 1 (RCCL) needs distinct devices");
    m->opt_reduce = value;
    return 0;
  }
  for (Shard& s : m->sh)
    if (int rc = rmb_ctx_set_option(s.ctx, key, value)) return rc;
  return 0;
}

int rmb_multi_get_option(rmb_multi* m, const char* key, long* value) {
  if (int rc = check_multi(m, false)) return rc;
  if (!key || !value) return fail(RMB_ERR_ARG, "null key / value");
  if (!strcmp(key, "reduce")) { *value = m->opt_reduce; return 0; }
  if (!strcmp(key, "peer")) { *value = m->peer ? 1 : 0; return 0; }
  if (!strcmp(key, "threads")) { *value = (long)m->workers.size(); return 0; }
  return rmb_ctx_get_option(m->sh[0].ctx, key, value);
}

int rmb_multi_set_positions(rmb_multi* m, const double* r_host, long n, double a, const double* L, int wall) {
  DeviceGuard restore_device_;
  if (int rc = check_multi(m, false)) return rc;
  if (n < 0) return fail(RMB_ERR_ARG, "negative n");
  if (n > 0 && !r_host) return fail(RMB_ERR_ARG, "null positions");
  if (!(a > 0.0)) return fail(RMB_ERR_ARG, "blob radius must be positive");
  const size_t rb = (size_t)3 * n * sizeof(double);
  if (m->done_recorded) { RMB_HIP(hipEventSynchronize(m->ev_done)); }
  if (int rc = reserve_partials(m, n)) return rc;
  if (n > 0) {
    if (int rc = pinned_reserve(m, rb)) return rc;
    memcpy(m->pinned, r_host, rb);
  }
  for (Shard& s : m->sh) {
    RMB_HIP(hipSetDevice(s.device));
    if (n > 0) {
      if (int rc = s.r_stage.reserve(rb)) return rc;
      RMB_HIP(hipMemcpyAsync(s.r_stage.p, m->pinned, rb, hipMemcpyHostToDevice, s.stream));
    }
    if (int rc = rmb_set_positions_device(s.ctx, (const double*)s.r_stage.p, n, a, L, wall)) return rc;
  }
  for (Shard& s : m->sh) {
    RMB_HIP(hipSetDevice(s.device));
    RMB_HIP(hipStreamSynchronize(s.stream));
  }
  m->n = n;
  m->have_positions = true;
  return 0;
}

int rmb_multi_set_positions_device(rmb_multi* m, const double* r_dev, long n, double a, const double* L, int wall) {
  DeviceGuard restore_device_;
  if (int rc = check_multi(m, false)) return rc;
  if (n < 0) return fail(RMB_ERR_ARG, "negative n");
  if (n > 0 && !r_dev) return fail(RMB_ERR_ARG, "null positions");
  if (!(a > 0.0)) return fail(RMB_ERR_ARG, "blob radius must be positive");
  const long len = 3 * n;
  Shard& s0 = m->sh[0];
  if (int rc = reserve_partials(m, n)) return rc;
  RMB_HIP(hipSetDevice(s0.device));
  RMB_HIP(hipEventRecord(m->ev_in, m->primary));
  for (Shard& s : m->sh) {
    RMB_HIP(hipSetDevice(s.device));
    RMB_HIP(hipStreamWaitEvent(s.stream, m->ev_in, 0));
    // every shard keeps its OWN copy of the raw positions (the caller may overwrite r_dev as soon as this returns
    // and its stream moves on: the copy below is ordered before that by ev_reduced -> primary)
    if (n > 0) {
      if (int rc = s.r_stage.reserve((size_t)len * sizeof(double))) return rc;
      if (s.device == s0.device && !(m->force_remote && &s != &s0))
        RMB_HIP(hipMemcpyAsync(s.r_stage.p, r_dev, (size_t)len * sizeof(double), hipMemcpyDeviceToDevice, s.stream));
      else
        RMB_HIP(hipMemcpyPeerAsync(s.r_stage.p, s.device, r_dev, s0.device, (size_t)len * sizeof(double), s.stream));   // caller-owned memory
    }
    if (int rc = rmb_set_positions_device(s.ctx, (const double*)s.r_stage.p, n, a, L, wall)) return rc;
    RMB_HIP(hipEventRecord(s.ev_reduced, s.stream));
  }
  RMB_HIP(hipSetDevice(s0.device));
  for (Shard& s : m->sh) RMB_HIP(hipStreamWaitEvent(m->primary, s.ev_reduced, 0));
  RMB_HIP(hipEventRecord(m->ev_done, m->primary));
  m->done_recorded = true;
  m->n = n;
  m->have_positions = true;
  return 0;
}

int rmb_multi_matvec(rmb_multi* m, int kind, int in_plane, const double* vec_host, const double* vec2_host, double eta,
                     double* out_host) {
  DeviceGuard restore_device_;
  if (int rc = check_multi(m, true)) return rc;
  if (m->n == 0) return 0;
  if (!vec_host || !out_host) return fail(RMB_ERR_ARG, "null vector / output pointer");
  Product p;
  if (int rc = product_of_kind(m, kind, in_plane, vec2_host != nullptr, eta, &p)) return rc;
  const double* in[2] = {vec_host, vec2_host};
  double* out[1] = {out_host};
  return run_host(m, p, in, out);
}

int rmb_multi_matvec_device(rmb_multi* m, int kind, int in_plane, const double* vec_dev, const double* vec2_dev, double eta,
                            double* out_dev) {
  DeviceGuard restore_device_;
  if (int rc = check_multi(m, true)) return rc;
  if (m->n == 0) return 0;
  if (!vec_dev || !out_dev) return fail(RMB_ERR_ARG, "null vector / output pointer");
  Product p;
  if (int rc = product_of_kind(m, kind, in_plane, vec2_dev != nullptr, eta, &p)) return rc;
  const double* in[2] = {vec_dev, vec2_dev};
  double* out[1] = {out_dev};
  return run_device(m, p, in, out);
}

int rmb_multi_matvec_op_device(rmb_multi* m, int op, int in_plane, int n_in, const double* const* in_dev, int n_out,
                               double* const* out_dev, double eta) {
  DeviceGuard restore_device_;
  if (int rc = check_multi(m, true)) return rc;
  if (!in_dev || !out_dev) return fail(RMB_ERR_ARG, "null vector / output list");
  Product p;
  if (int rc = product_of_op(op, in_plane, n_in, n_out, eta, &p)) return rc;
  for (int v = 0; v < n_in; ++v) if (!in_dev[v]) return fail(RMB_ERR_ARG, "null input vector");
  for (int v = 0; v < n_out; ++v) if (!out_dev[v]) return fail(RMB_ERR_ARG, "null output vector");
  if (m->n == 0) return 0;
  return run_device(m, p, in_dev, out_dev);
}

int rmb_multi_blob_blob_force(rmb_multi* m, double repulsion_strength, double debye_length, double blob_radius, double* out_host) {
  DeviceGuard restore_device_;
  if (int rc = check_multi(m, true)) return rc;
  if (m->n == 0) return 0;
  if (!out_host) return fail(RMB_ERR_ARG, "null output pointer");
  Product p;
  p.n_in = 0; p.n_out = 1; p.eps = repulsion_strength; p.b = debye_length; p.a = blob_radius; p.launch = launch_force;
  double* out[1] = {out_host};
  return run_host(m, p, nullptr, out);
}

int rmb_multi_blob_blob_force_device(rmb_multi* m, double repulsion_strength, double debye_length, double blob_radius,
                                     double* out_dev) {
  DeviceGuard restore_device_;
  if (int rc = check_multi(m, true)) return rc;
  if (m->n == 0) return 0;
  if (!out_dev) return fail(RMB_ERR_ARG, "null output pointer");
  Product p;
  p.n_in = 0; p.n_out = 1; p.eps = repulsion_strength; p.b = debye_length; p.a = blob_radius; p.launch = launch_force;
  double* out[1] = {out_dev};
  return run_device(m, p, nullptr, out);
}

int rmb_multi_synchronize(rmb_multi* m) {
  DeviceGuard restore_device_;
  if (int rc = check_multi(m, false)) return rc;
  for (Shard& s : m->sh) {
    RMB_HIP(hipSetDevice(s.device));
    RMB_HIP(hipStreamSynchronize(s.stream));
  }
  RMB_HIP(hipSetDevice(m->sh[0].device));
  if (m->done_recorded) { RMB_HIP(hipEventSynchronize(m->ev_done)); }
  return 0;
}

}  // extern "C"
