// rmb_entry.hip -- the extern "C" products of include/rmb_mobility.h: argument checks, routing between the symmetric
// and the one-sided kernel families (rmb_sym.hip / rmb_sweep.hip), host staging of the synchronous entry points.
#include "rmb_internal.h"

#include <chrono>
#include <cmath>
#include <cstring>

namespace rmbi {

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
  return sweep_device(c, kind, in_plane, v, v2, eta, out);
}

// shard / nshards: pair shard of the unordered pairs (F_ji = -F_ij needs no self term) into a full-length partial; a
// shard always takes the symmetric kernel, whatever n and the target range (as rmb_matvec_pairshard_device).
int force_device_impl(rmb_ctx* c, double eps, double b, double blob_radius, double* out, const double* radii, long shard,
                      long nshards) {
  if (int rc = check_ready(c)) return rc;
  const long n_tgt = nshards > 1 ? c->n : c->tgt_end - c->tgt_begin;
  if (n_tgt == 0) return 0;
  if (!out) return fail(RMB_ERR_ARG, "null output pointer");
  if (!(b > 0.0)) return fail(RMB_ERR_ARG, "debye_length must be positive");
  if (nshards < 1 || shard < 0 || shard >= nshards) return fail(RMB_ERR_ARG, "bad shard / nshards");
  RMB_HIP(hipSetDevice(c->device));
  const bool periodic = c->L[0] > 0 || c->L[1] > 0 || c->L[2] > 0;
  c->last_path = 0;
  // symmetric path: each unordered pair once (F_ji = -F_ij); its flushes are atomics, so both deterministic modes
  // take the one-sided sweep
  if (nshards > 1 || (sym_applies(c) && c->opt_deterministic == 0))
    return sym_force_device(c, eps, b, blob_radius, out, radii, shard, nshards);
  return force_sweep_device(c, eps, b, blob_radius, out, radii);
}

// Multi-block operations (include/rmb_mobility.h, enum rmb_op).  One symmetric pass when that path applies (or for a
// pair shard); otherwise composed from the one-sided sweeps (target sub-ranges, "deterministic", n < 128).
// One pair shard of a single-vector product.  `in_plane` (the reference's in_plane_* wrappers: z row / column masked) is
// honoured for the free-surface block only -- the one kind the multi-device engine routes here with it (the wall / open
// kinds with in_plane go through matvec_op_impl's *_MULTI operations); the public rmb_matvec_pairshard_device passes 0.
int matvec_pairshard_impl(rmb_ctx* c, int kind, int in_plane, const double* v, double eta, double* out, long shard, long nshards) {
  if (int rc = check_ready(c)) return rc;
  if ((kind < 0 || kind > rmb::KIND_RR) && kind != rmb::KIND_TT_FREE)
    return fail(RMB_ERR_ARG, "pair sharding is implemented for RMB_TT / TR / RT / RR / TT_FREE_SURFACE (RMB_TT_TR: rmb_matvec_op_pairshard_device)");
  if (kind == rmb::KIND_TT_FREE && c->wall)
    return fail(RMB_ERR_STATE, "RMB_TT_FREE_SURFACE uses raw heights: call rmb_set_positions with wall = 0");
  if (in_plane && kind != rmb::KIND_TT_FREE)
    return fail(RMB_ERR_ARG, "pair shards with in_plane: use rmb_matvec_op_pairshard_device (RMB_OP_*_MULTI) for RMB_TT / TR / RT / RR");
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
    return symx_det_device(c, sx, in, outs, eta, in_plane ? 1 : 0, shard, nshards);
  if (kind == rmb::KIND_TT_FREE) return symx_device(c, SX_FREE, in, outs, eta, in_plane ? 1 : 0, shard, nshards);
  return sym_device(c, kind, v, eta, out, shard, nshards);
}

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
      return add_inplace(c, out[1], (const double*)c->tmp3n.p, 3 * n_tgt);
    }
    default:
      for (int v = 0; v < n_in; ++v)
        if (int rc = matvec_device_impl(c, multi_kind, in_plane, in[v], nullptr, eta, out[v])) return rc;
      return 0;
  }
}

namespace {
// host arrays -> staging buffers of the default context; returns device pointers in dev[]
int aux_stage(rmb_ctx* c, int n, const double* const* host, const size_t* bytes, const int* slot, const double** dev) {
  for (int k = 0; k < n; ++k) {
    if (int rc = c->st[slot[k]].reserve(bytes[k] ? bytes[k] : sizeof(double))) return rc;
    if (bytes[k]) RMB_HIP(hipMemcpyAsync(c->st[slot[k]].p, host[k], bytes[k], hipMemcpyHostToDevice, c->stream));
    dev[k] = (const double*)c->st[slot[k]].p;
  }
  return 0;
}

// One-blob forces of the rigid-multiblob driver (multi_bodies/multi_bodies_functions.py:153-188): weight along -z and the
// screened repulsion from the wall, f_z = -weight + (eps_w / b_w) exp(-(h - a) / b_w) for h > a, -weight + eps_w / b_w
// at contact.  One thread per blob, on the caller's own coordinates (no clamp).
__global__ __launch_bounds__(256) void one_blob_force_kernel(const double* r, long n, double a, double weight, double eps_wall,
                                                             double debye_wall, int accumulate, double* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double fz = -weight;
  if (eps_wall != 0.0) {
    const double h = r[3 * i + 2], e = eps_wall / debye_wall;
    fz = __dadd_rn(fz, h > a ? __dmul_rn(e, exp(-(h - a) / debye_wall)) : e);      // (no contraction: the bits of the tensor formula)
  }
  if (accumulate) {
    out[3 * i + 2] += fz;
  } else {
    out[3 * i] = 0.0;
    out[3 * i + 1] = 0.0;
    out[3 * i + 2] = fz;
  }
}
}  // namespace

}  // namespace rmbi

using namespace rmbi;

extern "C" {

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
  if (int rc = pack_positions(c, (const double*)c->r_stage.p, n, a, L, wall)) return rc;
  RMB_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int rmb_set_positions_device(rmb_ctx* c, const double* r_dev, long n, double a, const double* L, int wall) {
  if (!c) return fail(RMB_ERR_ARG, "null context");
  if (n < 0) return fail(RMB_ERR_ARG, "negative n");
  if (n > 0 && !r_dev) return fail(RMB_ERR_ARG, "null positions");
  if (!(a > 0.0)) return fail(RMB_ERR_ARG, "blob radius must be positive");
  RMB_HIP(hipSetDevice(c->device));
  return pack_positions(c, r_dev, n, a, L, wall);
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
  return matvec_pairshard_impl(c, kind, 0, v, eta, out, shard, nshards);
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
  return body_dense_device(c, first_blob_dev, n_bodies, n_b, eta, out_dev);
}

int rmb_matvec(rmb_ctx* c, int kind, int in_plane, const double* v, const double* v2, double eta, double* out) {
  if (int rc = check_ready(c)) return rc;
  const long n = c->n, n_tgt = c->tgt_end - c->tgt_begin;
  if (n_tgt == 0) return 0;
  if (!v || !out) return fail(RMB_ERR_ARG, "null vector / output pointer");
  if (kind == rmb::KIND_TT_TR && !v2) return fail(RMB_ERR_ARG, "RMB_TT_TR needs vec2 (torque)");
  RMB_HIP(hipSetDevice(c->device));
  // host wall clock of the four stages of this synchronous call (rmb_last_host_timing): where the time through the
  // reference's call shape goes beyond the kernel
  const auto t0 = std::chrono::steady_clock::now();
  const size_t vb = (size_t)3 * n * sizeof(double), ob = (size_t)3 * n_tgt * sizeof(double);
  if (int rc = c->vec.reserve(vb)) return rc;
  if (int rc = c->out.reserve(ob)) return rc;
  // Inputs.  Small vectors (as the result below): host memcpy into page-locked, device-mapped memory + a pull kernel on
  // the product's own queue, instead of a host-to-device copy command on the copy queue.
  const bool pull_in = c->opt_host_zero_copy_in && c->opt_host_zero_copy > 0 && vb <= (size_t)c->opt_host_zero_copy && n >= 128;
  const int n_vec = kind == rmb::KIND_TT_TR ? 2 : 1;
  if (pull_in && (size_t)n_vec * vb > c->host_in_cap) {
    if (c->host_in) { (void)hipHostFree(c->host_in); c->host_in = nullptr; c->host_in_cap = 0; }
    const size_t want = 2 * (vb + vb / 8) + 4096;
    RMB_HIP(hipHostMalloc(&c->host_in, want, hipHostMallocMapped));
    RMB_HIP(hipHostGetDevicePointer((void**)&c->host_in_dev, c->host_in, 0));
    c->host_in_cap = want;
  }
  if (pull_in) {
    memcpy(c->host_in, v, vb);
    if (int rc = pull_mapped(c, (double*)c->vec.p, c->host_in_dev, 3 * n)) return rc;
  } else {
    RMB_HIP(hipMemcpyAsync(c->vec.p, v, vb, hipMemcpyHostToDevice, c->stream));
  }
  const double* v2d = nullptr;
  if (kind == rmb::KIND_TT_TR) {
    if (int rc = c->vec2.reserve(vb)) return rc;
    if (pull_in) {
      memcpy((char*)c->host_in + vb, v2, vb);
      if (int rc = pull_mapped(c, (double*)c->vec2.p, c->host_in_dev + 3 * n, 3 * n)) return rc;
    } else {
      RMB_HIP(hipMemcpyAsync(c->vec2.p, v2, vb, hipMemcpyHostToDevice, c->stream));
    }
    v2d = (const double*)c->vec2.p;
  }
  // Result hand-off.  Small results (<= "host_zero_copy" bytes, 768 KB): the finalize kernel stores straight into page-locked,
  // device-mapped host memory (coalesced 2 KB store instructions, store_aos_coalesced) and the call ends with ONE stream
  // wait + a host memcpy into the caller's array -- no device-to-host copy command, whose hand-over between the compute and
  // the copy queue costs more than 240 KB over PCIe do (tools/ubench_hostwrite.hip: -11 us per call at 1e3 and 1e4 blobs).
  // Larger results take the copy command (mapped stores lose from ~1e5 blobs on); so does the one-sided sweep
  // ("deterministic" = 1), whose final stores are 8 bytes at a stride of 24.
  const bool zero_copy = c->opt_host_zero_copy > 0 && ob <= (size_t)c->opt_host_zero_copy && c->opt_deterministic != 1 && n >= 128;
  if (zero_copy && ob > c->host_out_cap) {
    if (c->host_out) { (void)hipHostFree(c->host_out); c->host_out = nullptr; c->host_out_cap = 0; }
    const size_t want = ob + ob / 8 + 4096;
    RMB_HIP(hipHostMalloc(&c->host_out, want, hipHostMallocMapped));
    RMB_HIP(hipHostGetDevicePointer((void**)&c->host_out_dev, c->host_out, 0));
    c->host_out_cap = want;
  }
  const auto t1 = std::chrono::steady_clock::now();
  if (int rc = matvec_device_impl(c, kind, in_plane, (const double*)c->vec.p, v2d, eta, zero_copy ? c->host_out_dev : (double*)c->out.p)) return rc;
  const auto t2 = std::chrono::steady_clock::now();
  if (zero_copy) {
    RMB_HIP(hipStreamSynchronize(c->stream));
    memcpy(out, c->host_out, ob);
  } else {
    RMB_HIP(hipMemcpyAsync(out, c->out.p, ob, hipMemcpyDeviceToHost, c->stream));
    RMB_HIP(hipStreamSynchronize(c->stream));
  }
  const auto t3 = std::chrono::steady_clock::now();
  const auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::micro>(b - a).count();
  };
  c->host_us[0] = us(t0, t1); c->host_us[1] = us(t1, t2); c->host_us[2] = us(t2, t3); c->host_us[3] = us(t0, t3);
  return 0;
}

int rmb_last_host_timing(rmb_ctx* c, double* us4) {
  if (!c || !us4) return fail(RMB_ERR_ARG, "null context / output");
  for (int k = 0; k < 4; ++k) us4[k] = c->host_us[k];
  return 0;
}

int rmb_blob_blob_force_device(rmb_ctx* c, double eps, double b, double blob_radius, double* out) {
  return force_device_impl(c, eps, b, blob_radius, out);
}

int rmb_one_blob_force_device(rmb_ctx* c, long n, const double* r_dev, double blob_radius, double weight, double eps_wall,
                              double debye_wall, int accumulate, double* out_dev) {
  if (!c) return fail(RMB_ERR_ARG, "null context");
  if (n < 0 || (n > 0 && (!r_dev || !out_dev))) return fail(RMB_ERR_ARG, "rmb_one_blob_force_device: bad n / null pointer");
  if (eps_wall != 0.0 && !(debye_wall > 0.0)) return fail(RMB_ERR_ARG, "rmb_one_blob_force_device: debye_wall must be positive");
  if (n == 0) return 0;
  RMB_HIP(hipSetDevice(c->device));
  hipLaunchKernelGGL(one_blob_force_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, r_dev, n, blob_radius, weight,
                     eps_wall, debye_wall, accumulate, out_dev);
  RMB_HIP(hipGetLastError());
  return 0;
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
    if (int rc = pack_positions_radii(c, src_dev, rad_s_dev, ns, wall, (double4*)c->st[0].p)) return rc;
    SymConf cf{(const double4*)c->st[0].p, ns, {L ? L[0] : 0.0, L ? L[1] : 0.0, L ? L[2] : 0.0}, wall, rad_s_dev};
    const double* in[1] = {force_dev};
    double* outs[1] = {out_dev};
    return symx_device(c, SX_RADII, in, outs, eta, 0, 0, 1, 0, &cf);
  }
  if (int rc = c->st[0].reserve((size_t)ns * sizeof(double4))) return rc;
  if (int rc = c->st[1].reserve((size_t)nt * sizeof(double4))) return rc;
  if (int rc = pack_positions_radii(c, src_dev, rad_s_dev, ns, wall == 1 ? 1 : 0, (double4*)c->st[0].p)) return rc;
  if (int rc = pack_positions_radii(c, tgt_dev, rad_t_dev, nt, wall == 1 ? 1 : 0, (double4*)c->st[1].p)) return rc;
  return st_sweep_device(c, ns, (const double4*)c->st[0].p, rad_s_dev, force_dev, nt, (const double4*)c->st[1].p, rad_t_dev, eta,
                         L, wall, out_dev);
}

int rmb_mobility_source_target(long ns, const double* src, const double* rad_s, long nt, const double* tgt,
                               const double* rad_t, const double* force, double eta, const double* L, int wall,
                               double* out) {
  std::lock_guard<std::mutex> lk(g_default_mu);
  rmb_ctx* c;
  if (int rc = default_ctx(&c)) return rc;
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
  return pressure_device(c, ns, src_dev, nt, tgt_dev, force_dev, wall, out_dev);
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
  return double_layer_device(c, ns, src_dev, nt, tgt_dev, normals_dev, vector_dev, weights_dev, wall, blob_radius, out_dev);
}

int rmb_pressure_stokeslet(long ns, const double* src, long nt, const double* tgt, const double* force, const double* L,
                           int wall, double* out) {
  std::lock_guard<std::mutex> lk(g_default_mu);
  rmb_ctx* c;
  if (int rc = default_ctx(&c)) return rc;
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
  rmb_ctx* c;
  if (int rc = default_ctx(&c)) return rc;
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

