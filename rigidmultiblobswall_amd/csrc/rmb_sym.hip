// rmb_sym.hip -- launchers of the symmetric (each unordered pair once, both blobs updated) fp64 kernels:
// sym_kernel (tt / tr / rt / rr), sym2_kernel (two vectors), symx_kernel (multi-block / multi-vector operations, the
// deterministic variant with its ordered reduction) and the symmetric blob-blob force kernel.
#include "rmb_internal.h"

#include <cmath>

#include "sym_kernels.h"
#include "sym_coop_kernels.h"
#include "sym2t_kernels.h"
#include "sym2_kernels.h"
#include "symx_kernels.h"

namespace rmbi {

namespace {
typedef void (*sym_fn)(const rmb::SymArgs);
struct SymEntry { sym_fn sweep; sym_fn fin; int occ; sym_fn coop; int coop_occ; sym_fn two; int two_occ; };
template <int KIND, bool WALL, bool PER> SymEntry make_sym_entry() {
  sym_fn two = nullptr;
  if constexpr (!PER) two = rmb::sym2t_kernel<KIND, WALL>;      // two target blobs per lane: open boundaries only
  return SymEntry{rmb::sym_kernel<KIND, WALL, PER>, rmb::sym_finalize_kernel<KIND, WALL>, 0, rmb::sym_coop_kernel<KIND, WALL, PER>, 0, two, 0};
}
// [kind tt,tr,rt,rr][wall][periodic]
#define RMB_SYM_ROW(K) {{make_sym_entry<K, false, false>(), make_sym_entry<K, false, true>()}, {make_sym_entry<K, true, false>(), make_sym_entry<K, true, true>()}}
SymEntry g_sym[4][2][2] = {RMB_SYM_ROW(rmb::KIND_TT), RMB_SYM_ROW(rmb::KIND_TR), RMB_SYM_ROW(rmb::KIND_RT), RMB_SYM_ROW(rmb::KIND_RR)};
#undef RMB_SYM_ROW
}  // namespace

int sym_device(rmb_ctx* c, int kind, const double* v, double eta, double* out, long shard, long nshards, bool accumulate,
               bool no_finalize) {
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
  a.order = (int)c->opt_sym_order; a.xcd = (int)c->opt_sym_xcd;
  shard_ranges(n, a.n_units, shard, nshards, &a.step_begin, &a.step_end, &a.self_begin, &a.self_end);
  a.Lx = c->L[0]; a.Ly = c->L[1]; a.Lz = c->L[2];
  a.iLx = c->L[0] > 0 ? 1.0 / c->L[0] : 0.0;
  a.iLy = c->L[1] > 0 ? 1.0 / c->L[1] : 0.0;
  a.iLz = c->L[2] > 0 ? 1.0 / c->L[2] : 0.0;
  a.prefactor = 1.0 / (8.0 * M_PI * eta);
  a.k = make_pair_consts(c->a);
  // Pseudo-periodic single-vector products: the two-targets-per-lane instance of the generic skeleton (symx2t_kernels.h:
  // one record read and one set of LDS adds for the 2 x 3^d image pairs of a step) from half a unit per resident wave on,
  // the same rule as below; smaller launches stay here (cooperative kernel).  The wave_clock diagnostic lives in sym_kernel.
  if (periodic && no_finalize) return fail(RMB_ERR_STATE, "sym_device: no_finalize is for open boundaries (internal)");
  if (periodic && c->opt_sym_two_targets && c->opt_sym_coop != 2 && tiles >= 4 && !c->opt_wave_clock) {
    int wpe = 0;
    const Kernel32 cand = symx_two_periodic(SX_TT + kind, c->wall != 0, &wpe);
    if (cand.fn) {
      long sb, se_, qb, qe;
      shard_ranges(n, rmb::units2_total(tiles), shard, nshards, &sb, &se_, &qb, &qe);
      SymPlan plan2;
      if (int rc = plan_sym(c, cand.fn, cand.occ, cand.static_lds, se_ - sb, true, &plan2, wpe)) return rc;
      if (!plan2.sub_round || (se_ - sb) >= 32 * plan2.round * rmb::kSymWaves || c->opt_sym_two_targets == 2) {
        const double* in[2] = {v, nullptr};
        double* outs[1] = {out};
        return symx_device(c, SX_TT + kind, in, outs, eta, 0, shard, nshards, accumulate ? 1 : 0);
      }
    }
  }
  SymPlan plan;
  // single-precision mode (mobility_pycuda.py:7-19 `precision = 'single'`): tt with open boundaries only
  const bool f32 = c->opt_precision == 32 && kind == RMB_TT && !periodic;
  if (f32 && (c->opt_wave_clock || c->opt_skip_pairs))
    return fail(RMB_ERR_STATE, "the \"wave_clock\" / \"skip_pairs\" diagnostics exist in the fp64 kernels only: set \"precision\" = 64");
  const Kernel32 k32 = f32 ? sym32_tt(c->wall != 0) : Kernel32{nullptr, 0, nullptr, nullptr};
  const size_t stat = f32 ? k32.static_lds
                          : sizeof(double2) * rmb::kSymWaves * 64 * 3 + sizeof(double) * rmb::kSymWaves * 3 * 64;
  if (int rc = plan_sym(c, f32 ? k32.fn : (const void*)se.sweep, f32 ? k32.occ : &se.occ, stat,
                        a.step_end - a.step_begin, true, &plan, f32 ? 0 : rmb::kSymWavesPerEu))
    return rc;
  // Workgroup-cooperative variant (sym_coop_kernels.h): the workgroup owns the step range, its four waves share one
  // staged tile J and one flush per tile.  Faster below one resident round (1/8 pair shard of 1e4 blobs 29.6 -> 25.0 us,
  // 1000 blobs 10.4 -> 9.0 us), the same time up to a few rounds with HALF the atomic flush traffic (1e4 blobs:
  // WRITE_SIZE 54.4 -> 27.4 MB per launch, 146.7 vs 147.8 us), 0.5-1 % slower at >= 8 rounds
  // (profiles/r4_coop_kernel_ab.txt): by default up to kCoopMaxRounds.
  constexpr long kCoopMaxRounds = 4;
  // Two target blobs per lane (sym2t_kernels.h): units are (row pair, tile) -- half as many steps, two pairs per step.
  // From half a resident round on (smaller launches stay with the cooperative kernel, whose four waves share a unit).
  bool two = false;
  if (se.two && !f32 && c->opt_sym_two_targets && c->opt_sym_coop != 2 && tiles >= 4) {
    rmb::SymArgs t = a;
    t.n_units = rmb::units2_total(tiles);
    shard_ranges(n, t.n_units, shard, nshards, &t.step_begin, &t.step_end, &t.self_begin, &t.self_end);
    SymPlan plan2;
    if (int rc = plan_sym(c, (const void*)se.two, &se.two_occ, stat, t.step_end - t.step_begin, true, &plan2, rmb::kSymWavesPerEu))
      return rc;
    // from half a unit (32 steps) per resident wave on: 6000 blobs on a whole MI355X (59.4 vs 60.3 us; 76.4 vs 80.4 at 7000,
    // 97.2 vs 101.8 at 8000; below, the cooperative kernel wins: 50.7 vs 44.5 us at 5000 -- tools/experiments/exp_sym2t_threshold.py)
    if (!plan2.sub_round || (t.step_end - t.step_begin) >= 32 * plan2.round * rmb::kSymWaves || c->opt_sym_two_targets == 2) {
      two = true;
      a = t;
      plan = plan2;
    }
  }
  const bool coop = !two && !f32 && !c->opt_wave_clock &&
                    (c->opt_sym_coop == 2 || (c->opt_sym_coop == 1 && (plan.sub_round || plan.blocks <= kCoopMaxRounds * plan.round)));
  if (coop) {
    const size_t coop_lds = sizeof(double2) * 64 * 3 + sizeof(double) * 2 * 3 * 64;
    // steps per wave below one resident round: 8, and 4 for the smallest launches (<= 12288 rotation steps, i.e. up to
    // ~1200 blobs: 1000 blobs 8.98 -> 8.26 us; profiles/r4_coop_kernel_ab.txt)
    const long fine = (a.step_end - a.step_begin) <= 12288 ? 4 : 8;
    if (int rc = plan_sym(c, (const void*)se.coop, &se.coop_occ, coop_lds, a.step_end - a.step_begin, true, &plan, rmb::kSymWavesPerEu, fine))
      return rc;
  }
  const long blocks = plan.blocks;
  const long total_steps = a.step_end - a.step_begin;
  a.steps_per_wave = coop ? (total_steps + blocks - 1) / blocks : plan.steps_per_wave;    // coop: steps per WORKGROUP
  a.steps_per_wave = chunked_steps(c, total_steps, coop ? blocks : blocks * rmb::kSymWaves, a.steps_per_wave,
                                   c->opt_sym_chunk_steps * (coop ? rmb::kSymWaves : 1));
  c->last_tiles = tiles; c->last_chunks = 0; c->last_wgs = blocks;
  c->last_path = coop ? 3 : (two ? 4 : 1);
  a.skip_pairs = (int)c->opt_skip_pairs;
  a.accumulate = accumulate ? 1 : 0;
  a.wave_clock = nullptr;
  if (c->opt_wave_clock) {
    // [waves][2] stamps (start, end | placement); sym2t_kernel adds a second region [waves][2] of per-wave phase totals in
    // shader-clock cycles (staging incl. its wait, everything) -- rmb_wave_clock_collect hands out both as 2 x waves rows
    c->wave_clock_n = blocks * rmb::kSymWaves * (two ? 2 : 1);
    if (int rc = c->wave_clock.reserve((size_t)2 * c->wave_clock_n * sizeof(long long))) return rc;
    a.wave_clock = (long long*)c->wave_clock.p;
  }
  int slot;
  if (int rc = timing_begin(c, &slot)) return rc;
  if (f32) {
    k32.launch(&a, a.k, (unsigned)blocks, plan.dyn_lds, c->stream);
  } else {
    hipLaunchKernelGGL(coop ? se.coop : (two ? se.two : se.sweep), dim3((unsigned)blocks), dim3(64 * rmb::kSymWaves), plan.dyn_lds, c->stream, a);
  }
  RMB_HIP(hipGetLastError());
  if (int rc = timing_end(c, slot)) return rc;
  if (no_finalize) return 0;          // the caller finishes the accumulators itself
  const dim3 fgrid((unsigned)((n + 255) / 256));
  hipLaunchKernelGGL(se.fin, fgrid, dim3(256), 0, c->stream, a);
  RMB_HIP(hipGetLastError());
  return 0;
}

// Two source vectors, one pass over the unordered pairs (sym2_kernels.h).  Same schedule rules as sym_device.
int sym2_device(rmb_ctx* c, const double* va, const double* vb, double eta, double* out_a, double* out_b, long shard,
                long nshards) {
  const bool periodic = c->L[0] > 0 || c->L[1] > 0 || c->L[2] > 0;
  const long n = c->n, tiles = (n + 63) / 64, n_pad = 64 * tiles;
  if (int rc = sym_accumulators(c, n_pad)) return rc;
  rmb::Sym2Args a;
  a.pos = (const double4*)c->pos.p;
  a.vec_a = va; a.vec_b = vb;
  a.acc = (double*)c->symbuf.p;
  a.out_a = out_a; a.out_b = out_b;
  a.n = n; a.n_pad = n_pad; a.n_tiles = (int)tiles; a.n_units = tiles * (tiles + 1) / 2;
  a.order = (int)c->opt_sym_order; a.xcd = (int)c->opt_sym_xcd;
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
  a.steps_per_wave = chunked_steps(c, a.step_end - a.step_begin, blocks * rmb::kSymWaves, plan.steps_per_wave, c->opt_sym_chunk_steps);
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
namespace {
typedef void (*symx_fn)(const rmb::SymXArgs);
typedef void (*symx_combine_fn)(const rmb::SymXArgs, int);
struct SymXEntry { symx_fn sweep; symx_fn fin; int occ; size_t static_lds; int n_in, n_out; symx_fn det_sweep; symx_fn det_reduce;
                   symx_combine_fn det_combine; int det_occ; };
template <class OP, bool WALL, bool PER> SymXEntry make_symx_entry() {
  return SymXEntry{rmb::symx_kernel<OP, WALL, PER, false>, rmb::symx_finalize_kernel<OP, WALL>, 0,
                   sizeof(double2) * rmb::kSymWaves * 64 * rmb::SymXRec<OP::NIN, rmb::SymXExtra<OP>::value>::d2 +
                       sizeof(double) * rmb::kSymWaves * 3 * OP::NOUT * 64,
                   OP::NIN, OP::NOUT, rmb::symx_kernel<OP, WALL, PER, true>, rmb::symx_det_reduce_kernel<OP::NOUT>,
                   rmb::symx_det_combine_kernel<OP::NOUT>, 0};
}
// [op][wall][periodic]
#define RMB_SX_ROW(OP) {{make_symx_entry<OP, false, false>(), make_symx_entry<OP, false, true>()}, {make_symx_entry<OP, true, false>(), make_symx_entry<OP, true, true>()}}
SymXEntry g_symx[SX_COUNT][2][2] = {
    RMB_SX_ROW(rmb::OpSingle<rmb::KIND_TT>), RMB_SX_ROW(rmb::OpSingle<rmb::KIND_TR>), RMB_SX_ROW(rmb::OpSingle<rmb::KIND_RT>),
    RMB_SX_ROW(rmb::OpSingle<rmb::KIND_RR>), RMB_SX_ROW(rmb::OpFusedRow), RMB_SX_ROW(rmb::OpGrand), RMB_SX_ROW(rmb::OpColumnF),
    // the free-surface operation takes raw heights: only the wall = 0 column is ever launched
    {{make_symx_entry<rmb::OpFreeSurface, false, false>(), make_symx_entry<rmb::OpFreeSurface, false, true>()},
     {make_symx_entry<rmb::OpFreeSurface, false, false>(), make_symx_entry<rmb::OpFreeSurface, false, true>()}},
    RMB_SX_ROW(rmb::OpRadiiTT),
#define RMB_SX_K(K) RMB_SX_ROW(RMB_SX_KIND(rmb::KIND_TT, K)), RMB_SX_ROW(RMB_SX_KIND(rmb::KIND_TR, K)), \
                    RMB_SX_ROW(RMB_SX_KIND(rmb::KIND_RT, K)), RMB_SX_ROW(RMB_SX_KIND(rmb::KIND_RR, K))
#define RMB_SX_KIND(KIND, K) rmb::OpKindK<KIND, K>
    RMB_SX_K(2), RMB_SX_K(3), RMB_SX_K(4)};
#undef RMB_SX_K
#undef RMB_SX_KIND
#undef RMB_SX_ROW

bool coop_forced(const rmb_ctx* c) { return c->opt_sym_coop == 2; }

SymConf conf_of(const rmb_ctx* c) {
  return SymConf{(const double4*)c->pos.p, c->n, {c->L[0], c->L[1], c->L[2]}, c->wall, nullptr};
}
}  // namespace

int symx_device(rmb_ctx* c, int op, const double* const* in, double* const* out, double eta, int in_plane, long shard,
                long nshards, int accumulate_mask, const SymConf* conf_in) {
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
  a.order = (int)c->opt_sym_order; a.xcd = (int)c->opt_sym_xcd;
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
  const Kernel32 k32 = (c->opt_precision == 32 && !periodic) ? symx32(op, cf.wall != 0) : Kernel32{nullptr, 0, nullptr, nullptr};
  const bool f32 = k32.fn != nullptr;
  if (f32 && c->opt_skip_pairs)
    return fail(RMB_ERR_STATE, "the \"skip_pairs\" diagnostic exists in the fp64 kernels only: set \"precision\" = 64");
  if (int rc = plan_sym(c, f32 ? k32.fn : (const void*)se.sweep, f32 ? k32.occ : &se.occ,
                        f32 ? k32.static_lds : se.static_lds, a.step_end - a.step_begin, true, &plan))
    return rc;
  // Workgroup-cooperative instance (symx_coop_kernels.h).  Measured (profiles/r4_coop_kernel_ab.txt): faster below one
  // resident round (pair shards, small suspensions); for the three-vector passes at every size (their per-wave slabs,
  // 47 KB per workgroup, hold residency at three workgroups per CU where the registers allow four: -7 % at 1e4 blobs,
  // -1.6 % at 1e5); no gain for the other operations above one round, and 2-7 % SLOWER for four vectors at 1e5 blobs
  // (those passes are bound by the LDS pipe itself -- 20 LDS instructions per step -- and a third wave per SIMD only
  // adds contention).
  const bool three_vectors = op >= SX_K2 + 4 && op < SX_K2 + 8;
  Kernel32 kc{nullptr, 0, nullptr, nullptr};
  if (!f32 && (c->opt_sym_coop == 2 || (c->opt_sym_coop == 1 && (plan.sub_round || three_vectors))))
    kc = symx_coop(op, cf.wall != 0, periodic);
  // Two target blobs per lane (symx2t_kernels.h; option "sym_two_targets"): fused row, grand, force column, one block on
  // two vectors, and every pseudo-periodic single-vector product.  Same rule as sym_device: from half a unit per resident
  // wave on (smaller launches stay with the cooperative instances), units = (row pair, tile).
  Kernel32 k2{nullptr, 0, nullptr, nullptr};
  if (!f32 && !coop_forced(c) && c->opt_sym_two_targets && tiles >= 4) {
    int wpe = 0;
    const Kernel32 cand = periodic ? symx_two_periodic(op, cf.wall != 0, &wpe) : symx_two_open(op, cf.wall != 0, &wpe);
    if (cand.fn) {
      rmb::SymXArgs t = a;
      t.n_units = rmb::units2_total(tiles);
      shard_ranges(n, t.n_units, shard, nshards, &t.step_begin, &t.step_end, &t.self_begin, &t.self_end);
      SymPlan plan2;
      if (int rc = plan_sym(c, cand.fn, cand.occ, cand.static_lds, t.step_end - t.step_begin, true, &plan2, wpe)) return rc;
      if (!plan2.sub_round || (t.step_end - t.step_begin) >= 32 * plan2.round * rmb::kSymWaves || c->opt_sym_two_targets == 2) {
        k2 = cand;
        a = t;
        plan = plan2;
        kc = Kernel32{nullptr, 0, nullptr, nullptr};
      }
    }
  }
  const bool two = k2.fn != nullptr;
  const bool coop = !two && kc.fn != nullptr;
  if (coop) {
    const long fine = (a.step_end - a.step_begin) <= 12288 ? 4 : 8;     // as sym_device
    if (int rc = plan_sym(c, kc.fn, kc.occ, kc.static_lds, a.step_end - a.step_begin, true, &plan, 0, fine)) return rc;
  }
  const long total_steps = a.step_end - a.step_begin;
  a.steps_per_wave = coop ? (total_steps + plan.blocks - 1) / plan.blocks : plan.steps_per_wave;     // coop: steps per WORKGROUP
  a.steps_per_wave = chunked_steps(c, total_steps, coop ? plan.blocks : plan.blocks * rmb::kSymWaves, a.steps_per_wave,
                                   c->opt_sym_chunk_steps * (coop ? rmb::kSymWaves : 1));
  c->last_path = coop ? 3 : (two ? 4 : 1); c->last_tiles = tiles; c->last_chunks = 0; c->last_wgs = plan.blocks;
  int slot;
  if (int rc = timing_begin(c, &slot)) return rc;
  if (f32) k32.launch(&a, a.k, (unsigned)plan.blocks, plan.dyn_lds, c->stream);
  else if (two) k2.launch(&a, a.k, (unsigned)plan.blocks, plan.dyn_lds, c->stream);
  else if (coop) kc.launch(&a, a.k, (unsigned)plan.blocks, plan.dyn_lds, c->stream);
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
                    long shard, long nshards) {
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
  a.order = 0; a.xcd = 0;        // the ordered reduction enumerates the units row-major
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

// Symmetric blob-blob force sweep (sym_kernels.h: each unordered pair once, F_ji = -F_ij) of pair shard `shard` of
// `nshards` into a full-length result; atomic flushes.  Tile culling and the fp32 twin as the options say.
int sym_force_device(rmb_ctx* c, double eps, double b, double blob_radius, double* out, const double* radii, long shard,
                     long nshards) {
  const bool periodic = c->L[0] > 0 || c->L[1] > 0 || c->L[2] > 0;
  const long n = c->n, tiles = (n + 63) / 64, n_pad = 64 * tiles;
  if (int rc = sym_accumulators(c, n_pad)) return rc;
  rmb::SymForceArgs a;
  a.pos = (const double4*)c->pos.p;
  a.acc = (double*)c->symbuf.p;
  a.out = out;
  a.n = n; a.n_pad = n_pad; a.n_tiles = (int)tiles; a.n_units = tiles * (tiles + 1) / 2;
  // Row-major units and plain workgroup numbering, whatever "sym_order" / "sym_xcd" say: with tile culling most units cost
  // nothing and the ones that do sit next to the diagonal (after the Morton sort), so the blocked order hands neighbouring
  // waves equally heavy runs of super-block rows; measured 3.37 vs 5.01 ms on a 3D cloud of 1e5 blobs, 2.39 vs 2.64 ms on
  // the 262 144-roller monolayer (tools/experiments/exp_force_ab.py).  The strided chunks stay.
  a.order = 0; a.xcd = 0;
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
  a.bounds = nullptr; a.cull2 = 0.0; a.perm = nullptr;
  if (c->opt_force_cull && !radii && tiles > 1) {
    // Spatial order first ("force_sort"): how much the culling skips depends on how compact a 64-blob tile is, i.e. on
    // the order in which the caller lists the blobs; from 32 tiles on the blobs are sorted along a Morton curve once
    // per configuration (rmb_sort.hip) and the kernel runs on the sorted copy.
    const bool want_sorted = c->opt_force_sort && tiles >= 32;
    if (!c->tile_bounds_valid || c->force_sorted != want_sorted) {
      if (want_sorted) {
        if (int rc = force_sort_positions(c)) return rc;
      } else {
        if (int rc = c->tile_bounds.reserve((size_t)6 * tiles * sizeof(double))) return rc;
        hipLaunchKernelGGL(rmb::tile_bounds_kernel, dim3((unsigned)tiles), dim3(64), 0, c->stream, (const double4*)c->pos.p, n,
                           (double*)c->tile_bounds.p);
        RMB_HIP(hipGetLastError());
      }
      c->force_sorted = want_sorted;
      c->tile_bounds_valid = true;
    }
    if (c->force_sorted) {
      a.pos = (const double4*)c->fpos.p;
      a.perm = (const unsigned*)c->fperm.p;
    }
    const double reach = 2.0 * blob_radius + (f32 ? 110.0 : 750.0) * b;
    a.bounds = (const double*)c->tile_bounds.p;
    a.cull2 = reach * reach;
  }
  static int socc[2][2] = {{0, 0}, {0, 0}};
  typedef void (*sforce_fn)(const rmb::SymForceArgs);
  const sforce_fn sfn = radii ? (periodic ? (sforce_fn)rmb::sym_force_kernel<true, true> : (sforce_fn)rmb::sym_force_kernel<false, true>)
                              : (periodic ? (sforce_fn)rmb::sym_force_kernel<true, false> : (sforce_fn)rmb::sym_force_kernel<false, false>);
  const Kernel32 k32 = f32 ? sym_force32(radii != nullptr) : Kernel32{nullptr, 0, nullptr, nullptr};
  const void* fn = f32 ? k32.fn : (const void*)sfn;
  long blocks = c->n_cu * resident_blocks(fn, f32 ? k32.occ : &socc[radii ? 1 : 0][periodic ? 1 : 0]) * c->opt_sym_oversub;
  long need = (a.step_end - a.step_begin + 255) / 256;
  if (need < 1) need = 1;
  if (blocks > need) blocks = need;
  c->last_path = 1; c->last_tiles = tiles; c->last_chunks = 0; c->last_wgs = blocks;
  {
    const long total_steps = a.step_end - a.step_begin, waves = blocks * rmb::kSymWaves;
    const long spw = (total_steps + waves - 1) / waves;
    // a quarter of the mobility kernels' chunk: the surviving units are few and uneven, shorter chunks balance them
    // (3D cloud of 1e5 blobs 3.36 -> 3.05 ms, monolayer unchanged; tools/experiments/exp_force_ab.py)
    const long ch = chunked_steps(c, total_steps, waves, spw, c->opt_sym_chunk_steps / 4);
    a.chunk_steps = ch < spw ? ch : 0;
  }
  int slot;
  if (int rc = timing_begin(c, &slot)) return rc;
  if (f32) k32.launch(&a, rmb::PairConsts{}, (unsigned)blocks, 0, c->stream);
  else     hipLaunchKernelGGL(sfn, dim3((unsigned)blocks), dim3(64 * rmb::kSymWaves), 0, c->stream, a);
  RMB_HIP(hipGetLastError());
  if (int rc = timing_end(c, slot)) return rc;
  hipLaunchKernelGGL(rmb::sym_force_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, a);
  RMB_HIP(hipGetLastError());
  return 0;
}

}  // namespace rmbi
