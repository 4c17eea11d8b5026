// rmb_plan.hip -- launch plans (host only): source chunks of the one-sided sweeps, residency, the exactly balanced
// step schedule of the symmetric kernels, pair-shard ranges, kernel-uniform constants.
#include "rmb_internal.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace rmbi {

// A CU has 4 SIMDs and a workgroup of the symmetric kernels is 4 waves, one per SIMD: "workgroups per CU" and "waves per
// SIMD" (the unit of amdgpu_waves_per_eu, which plan_sym's `declared_waves` carries) are then the same number.  Change
// either constant and the cap in plan_sym has to become declared_waves * kSimdsPerCu / kSymWaves.
constexpr int kSimdsPerCu = 4;
static_assert(rmb::kSymWaves == kSimdsPerCu, "plan_sym caps workgroups per CU with a waves-per-SIMD bound: units coincide only for 4-wave workgroups");

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

rmb::ExpConsts exp_consts() {
  rmb::ExpConsts e;
  e.log2e = 1.4426950408889634;
  e.ln2_hi = 6.93147180369123816490e-01;  // ln2 in two pieces; the high one has 33 significant bits, so n * ln2_hi is exact
  e.ln2_lo = 1.90821492927058770002e-10;
  double f = 2.0;
  for (int k = 0; k < 12; ++k) { e.c[k] = 1.0 / f; f *= (double)(k + 3); }
  return e;
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

int resident_blocks(const void* fn, int* cache) {
  if (*cache > 0) return *cache;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, rmb::kBlock, 0) != hipSuccess || nb < 1) nb = 4;
  if (nb > 8) nb = 8;
#ifdef RMB_DEBUG_OCC
  {
    hipFuncAttributes at;
    if (hipFuncGetAttributes(&at, fn) == hipSuccess)
      fprintf(stderr, "[rmb] occupancy api %d blocks/CU | numRegs %d sharedSizeBytes %zu maxThreadsPerBlock %d localSizeBytes %zu\n", nb,
              at.numRegs, (size_t)at.sharedSizeBytes, at.maxThreadsPerBlock, (size_t)at.localSizeBytes);
  }
#endif
  *cache = nb;
  return nb;
}

// Global SoA accumulators of the symmetric kernels: [4][3][n_pad] doubles (up to four output vectors per pass),
// zeroed once; every finalize kernel re-zeroes what its sweep touched.
constexpr int kSymMaxOut = 4;
int sym_accumulators(rmb_ctx* c, long n_pad) {
  // The accumulators are zero between products whatever layout the last product used: the kernels only add into entries
  // the finalize kernel of the same product reads and re-zeroes.  So the buffer is cleared ONCE, in full, when it is
  // (re)allocated -- never because n_pad changed: a context shared by two suspensions would otherwise put a memset into
  // whichever captured hipGraph happens to run first after the switch (rigid.py, _ArnoldiGraphs).
  const size_t acc_bytes = (size_t)3 * kSymMaxOut * n_pad * sizeof(double);
  if (acc_bytes > c->symbuf.cap) {
    if (int rc = c->symbuf.reserve(acc_bytes)) return rc;
    RMB_HIP(hipMemsetAsync(c->symbuf.p, 0, c->symbuf.cap, c->stream));
  }
  c->symbuf_zeroed_for = n_pad;
  return 0;
}


// Static, exactly balanced schedule (sym_kernels.h): whole multiples of the resident workgroup count so that every
// SIMD gets the same number of steps.  `pin` pads dynamic LDS so that exactly `wps` workgroups fit a CU (equal steps
// per wave is then equal work per SIMD); CU count and LDS size come from hipDeviceProp_t (rmb_ctx_create).
// `declared_waves`: the kernel's amdgpu_waves_per_eu bound (0 = none).  The occupancy API prices a kernel by its
// ARCHITECTURAL VGPRs only (hipFuncAttributes::numRegs); a kernel compiled under waves_per_eu(4, 4) parks values in
// AGPRs up to the 128-register budget (sym_kernel<TT>: 86 + 9 -> allocates 104, <RR>: 72 + 25), so the API reports 5
// and 7 workgroups per CU where the hardware holds 4 (per-wave start stamps, profiles/r3_shard_wave_placement.txt) and
// every plan built on "whole resident rounds" was off after the kernels lost registers in round 2.
int plan_sym(rmb_ctx* c, const void* fn, int* occ_cache, size_t static_lds, long total, bool pin, SymPlan* out,
             int declared_waves, long fine_auto) {
  int wps = resident_blocks(fn, occ_cache);
  if (declared_waves > 0 && wps > declared_waves * kSimdsPerCu / rmb::kSymWaves) wps = declared_waves * kSimdsPerCu / rmb::kSymWaves;
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
  out->sub_round = blocks < round;
  out->round = round;
  if (blocks < round) {
    // Less than one resident round at 64 steps per wave (small N, or one rank's pair shard): shorter waves beat
    // leaving SIMDs with one or two waves and no latency hiding, but every wave pays its own loads and 384 global
    // atomics on accumulators it shares with the other waves of its tile row.
    // Two regimes (tools/experiments/exp_small_n.py, tools/experiments/exp_shard_plan.py; profiles/r3_shard_plan.txt): while 16-step waves
    // do not fill the chip (small N: <= 1500 blobs) the launch is latency-bound and more, shorter waves win
    // (1000 blobs: 9.6 us at 16 steps, 13.0 at 32); once they would overfill it, every extra wave only adds its
    // loads and flushes (1/8 shard of 1e4 blobs: 36.6 us at 16 steps x 1024 workgroups, 29.3 at 32 x 776).
    long fine_steps = c->opt_sym_fine_steps;
    // `fine_auto` > 0: the caller's own default (the workgroup-cooperative kernel shares loads and flushes between
    // the waves of a workgroup, so shorter waves cost it nothing: 8 steps per wave, tools/experiments/exp_coop.py)
    if (fine_steps <= 0 && fine_auto > 0) fine_steps = fine_auto;
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

long chunked_steps(const rmb_ctx* c, long total, long n_sched, long spw, long target) {
  if (c->opt_sym_chunk_steps <= 0 || target <= 0 || n_sched < 1) return spw;
  const long rounds = (spw + target / 2) / target;       // chunks per schedule unit
  if (rounds < 2) return spw;
  return (total + n_sched * rounds - 1) / (n_sched * rounds);
}

// whether the symmetric (each unordered pair once) path applies to the resident configuration
bool sym_applies(const rmb_ctx* c) {
  return c->opt_symmetric && c->opt_deterministic != 1 && c->tgt_begin == 0 && c->tgt_end == c->n && c->n >= 128;
}

}  // namespace rmbi
