#!/usr/bin/env python
"""Headline benchmark: RPY-wall M.f matvecs/s on MI355X (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W

N > 1 launched plainly (no torchrun environment): this process spawns the N ranks itself -- child processes with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, started BEFORE anything here touches the
GPU (the parent never initialises HIP, never exec()s) -- waits for them and exits non-zero if any rank failed.
Launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` it is one rank.

A "step" is one single_wall_mobility_trans_times_force product (configs[1]: 1e4 random blobs above a wall, fp64) with
positions and the force vector already resident in HBM.  N > 1: the unordered blob pairs are sharded over the ranks
and the partial velocities all-reduced over RCCL (fixed total work => "strong" scaling); the line also carries both
decompositions (pair shard + all-reduce; target shard + all-gather of f, the one north_star describes) at 1e4 / 1e5 /
1e6 blobs.  Rank 0 prints ONE JSON line.

Before the W warm-up steps every rank primes the clocks with an untimed, DECLARED pre-warm (`prewarm` in the line):
from idle the fp64 clock needs ~100 launches / 25 ms to settle (tools/experiments/exp_prewarm.py: 0.207 -> 0.230 -> 0.189 ms per
launch), so `--warmup 5` alone would time the power-management ramp, not the kernel.

Objects on the line
  roofline      dominant kernel (the pair sweep).  The path is fp64-VALU bound, not HBM bound.
                `achieved` / `frac` are the UTILISATION, <= 1 by construction: fp64 flops the kernel really executes
                per launch (FMA = 2, mul/add/rsq = 1, counted over the pair loop of THIS build's ISA by
                tools/isa_stats.py) x the real unordered pairs of the launch / kernel time / 78.6 TF (`executed` holds
                the inputs of that number).  `algorithmic_throughput` is SURVEY 8(d)'s unit -- 211 as-written flops per
                ORDERED pair x N^2 / kernel time: a speed in the reference's currency that exceeds the peak because
                the symmetric kernel evaluates each unordered pair once with 123 flops; it is never `frac`.
                `issue` = VALU wave-instructions per launch from the same ISA count / kernel time against the fp64
                issue ceiling measured live in this process (rmb_ubench_fp64_issue).
                `traffic` (HBM bytes per launch) cannot be measured inside this process: the `traffic_live` stage runs
                two child passes of this script under rocprofv3 --pmc (FETCH_SIZE | WRITE_SIZE) and puts the measured
                figure on the line (`traffic_provenance.measured_in_this_run` true, the committed figure beside it);
                without rocprofv3 it stays the figure of the committed passes, tagged with its source.
  value_unprimed  the same W + K steps run first thing, before the declared pre-warm (clock still ramping)
  host_surface  matvecs/s through single_wall_mobility_trans_times_force_hip with numpy in / out (the reference's
                call shape, PCIe-inclusive) -- reported beside `value`, never `value`
                `devices` = what the call ran on (mobility.set_devices / RMB_DEVICES), `breakdown_us` = where one call's
                time goes (position compare, upload, enqueue, sweep by HIP events, download + sync, Python)
  two_targets_ab  one-rank run: the headline product on the one-target kernels and with two target blobs per lane, alternating
                in this process (a same-box A/B of the round's kernel change)
  two_targets_ops_ab  one-rank run: the fused row / grand product at 1e4 and 1e5 blobs and the pseudo-periodic tt / fused row at
                24 576 blobs, one target per lane against two (the default), alternating in this process
  small_deck_steps  one-rank run: whole time steps of the rigid-multiblob integrators on 64 / 256 shells (the reference's
                usual sizes), where the solver loop around the sweep decides (the loop itself inside the library)
  rccl_one_rank  one-rank run: the N > 1 step's fp64 all-reduce through RCCL in a one-rank group, step timed with and
                without it (child process, tools/rccl_one_rank_probe.py)
  multi_device_surface  one-rank run with several devices visible: the same call on the single-process multi-device
                engine over all of them (child process with a timeout, tools/multi_surface_probe.py)
  cpu_baseline  the CPU oracle's -O3 -ffast-math OpenMP build timed on this host (rank 0, N = 1 only); `by_size`: every
                size of SURVEY 8(d) up to 1e5 blobs measured (bounded repetitions), 262 144 / 1e6 extrapolated with N^2
  build         what __graft_entry__.build() did for the library this run loaded (compiled / reused)

The run cannot hang and cannot lose its headline (class Guard): the headline goes to stderr and to a file the moment
it exists; every later stage is started only if all ranks agree that it fits the wall-clock budget (--budget-s, 420 s);
a rank whose stage raises drops a marker file and every rank's watchdog thread ends the run within a second -- rank 0
prints the line with what exists plus `extras_aborted`; the same watchdog cuts a stage that never returns, at the stage's
own limit (--stage-limit-s, counted from the stage's start) or at the budget.  All clocks are time.monotonic().
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

FLOPS_PER_PAIR = {"tt_wall": 211.0}     # reference as-written op count, SURVEY.md 8(d)
FP64_VECTOR_PEAK_TFLOPS = 78.6          # MI355X fp64 vector = 1/2 of the 157.3 TF fp32 vector peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0
TIMING_STRIDE = 4                       # HIP events bracket every 4th sweep launch of the timed region


def d2_cloud(N, seed=0):
  """5 % volume fraction, a = 0.5, eta = 1, z in [1.1a, 1.1a + Lbox)."""
  rng = np.random.RandomState(seed)
  a, eta = 0.5, 1.0
  Lbox = (N * (4.0 / 3.0) * np.pi * a ** 3 / 0.05) ** (1.0 / 3.0)
  r = rng.rand(N, 3) * Lbox
  r[:, 2] += 1.1 * a
  return r, rng.randn(N, 3), eta, a


def parse_args():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=200)
  ap.add_argument("--warmup", type=int, default=20)
  ap.add_argument("--blobs", type=int, default=10000, help="N_blobs of the headline workload (configs[1] = 1e4)")
  ap.add_argument("--prewarm-ms", type=float, default=300.0,
                  help="untimed, declared clock pre-warm before the warm-up steps (0 disables)")
  ap.add_argument("--no-sweep", action="store_true", help="skip the N_blobs sweep, decompositions and config extras")
  ap.add_argument("--ctx-option", action="append", default=[], metavar="KEY=VALUE",
                  help="context option for this run (A/B and profiling passes), e.g. sym_coop=2; recorded on the line")
  ap.add_argument("--no-cpu", action="store_true")
  ap.add_argument("--budget-s", type=float, default=420.0,
                  help="wall-clock budget of the whole run: stages that would not fit are skipped on all ranks, and at the "
                       "budget the watchdog prints the line with what exists and ends the run (the driver's limit is 600 s)")
  ap.add_argument("--stage-limit-s", type=float, default=240.0,
                  help="no single extra after the headline may run longer than this, counted from the moment the stage starts "
                       "(the watchdog then prints the line with what exists and ends the run); 0 disables")
  ap.add_argument("--no-host-surface", action="store_true",
                  help="skip the host_surface extra (profiling runs: the last K dispatches of the trace are then the timed ones)")
  return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: spawn the ranks (the parent stays off the GPU)
# ---------------------------------------------------------------------------------------------------------------
def spawn_ranks(args):
  with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  import tempfile
  run_dir = tempfile.mkdtemp(prefix="rmb_bench_")      # side channel of the ranks' watchdogs (Guard)
  procs = []
  for rank in range(args.gpus):
    env = dict(os.environ)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # The hosts of this pool only support dmabuf IPC: with the legacy mode RCCL's (and torch's) cross-process buffer
    # registration fails with `hipIpcGetMemHandle: invalid argument`.  The image exports HSA_ENABLE_IPC_MODE_LEGACY=0
    # already (and a launcher such as torch.distributed.run passes its environment on); setdefault only covers a
    # shell that dropped it, it never overrides a value the operator chose.  The value every rank ran with is on the
    # line ("ipc_mode_legacy_env").
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["RMB_BENCH_RUN_DIR"] = run_dir
    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
  rc = 0
  alive = set(range(len(procs)))
  while alive:
    for i in sorted(alive):
      r = procs[i].poll()
      if r is None:
        continue
      alive.discard(i)
      if r != 0 and rc == 0:
        rc = r if r > 0 else 1
        sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n" % (i, r))
        for j in alive:
          procs[j].terminate()          # exact PIDs we started
    time.sleep(0.05)
  return rc


# ---------------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------------
def timed_region(torch, dist, world, device, backend, step, steps, warmup):
  """W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize; MAX over ranks."""
  for _ in range(warmup):
    step()
  torch.cuda.synchronize(device)
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize(device)
  backend.ctx.timing_reset()      # the first timed launch is sampled, then every TIMING_STRIDE-th
  t0 = time.perf_counter()
  for _ in range(steps):
    step()
  torch.cuda.synchronize(device)
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize(device)
  dt = time.perf_counter() - t0
  kern_ms = backend.ctx.timing_collect(steps)     # the sampled launches of the timed region
  kern_ms_avg = float(np.mean(kern_ms)) if len(kern_ms) else float("nan")
  if world > 1:
    t = torch.tensor([dt, kern_ms_avg], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, kern_ms_avg = float(t[0].item()), float(t[1].item())
  return dt, kern_ms_avg


def run_config(torch, dist, sm, backend, n_blobs, steps, warmup, world, rank, device, decomposition="pair", prewarm_ms=0.0,
               unprimed=False):
  """decomposition "pair": every rank holds f, evaluates its slice of the unordered pairs (each once, both blobs
  updated) into a full-length partial, one all-reduce of u.  "target": rank g owns a block of targets, all-gather of
  the f blocks, one-sided sweep of its targets against all sources, no reduction (north_star's layout)."""
  from rigidmultiblobswall_amd.distributed import partition
  r, f, eta, a = d2_cloud(n_blobs, seed=0)
  b, e, _ = partition(n_blobs, world, rank)
  sm.set_local_positions(torch.as_tensor(r[b:e].reshape(-1), device=device), n_blobs, a, wall=True)
  if decomposition == "pair":
    f_full = torch.as_tensor(f.reshape(-1), device=device)
    out = torch.empty(3 * n_blobs, dtype=torch.float64, device=device)
    if world == 1:
      def step():
        sm.matvec_local("tt", f_full, eta, out=out)
    else:
      def step():
        sm.matvec_replicated("tt", f_full, eta, out=out)
  else:
    f_local = torch.as_tensor(f[b:e].reshape(-1), device=device)
    out = torch.empty(3 * (e - b), dtype=torch.float64, device=device)
    if world == 1:
      backend.ctx.set_option("deterministic", 1)     # what one rank of a target shard runs: the one-sided sweep

    def step():
      sm.matvec_local("tt", f_local, eta, out=out)
  prewarm = None
  cold = None
  try:
    if unprimed:
      # the driver's bare flags on a chip that has not run this kernel yet: W warm-up steps, then the same K steps
      cold = timed_region(torch, dist, world, device, backend, step, steps, warmup)
    if prewarm_ms > 0:
      t0 = time.perf_counter()
      n_pre = 0
      while time.perf_counter() - t0 < prewarm_ms * 1e-3:
        for _ in range(20):
          step()
        torch.cuda.synchronize(device)
        n_pre += 20
      prewarm = {"ms": round(1e3 * (time.perf_counter() - t0), 1), "steps": n_pre, "timed": False}
    dt, kern_ms = timed_region(torch, dist, world, device, backend, step, steps, warmup)
  finally:
    backend.ctx.set_option("deterministic", 0)
  return dict(dt=dt, kern_ms=kern_ms, launch=backend.ctx.last_launch(), path=backend.ctx.get_option("last_path"), out=out, r=r, f=f,
              eta=eta, a=a,
              n_local=e - b, begin=b, end=e, prewarm=prewarm, cold=cold)


def committed_traffic(N, sym):
  """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/traffic.json), with their provenance."""
  try:
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
      tj = json.load(fh)
    key = ("sym_tt_wall_N%d" if sym else "sweep_tt_wall_N%d") % N
    cand = [(k, v) for k, v in tj.items() if isinstance(v, dict) and k.endswith(key)]
    if cand:
      k, v = cand[-1]
      prov = {"measured_in_this_run": False, "source": "profiles/traffic.json[%s]" % k,
              "files": v.get("files"), "commit": v.get("commit"), "command": v.get("command"),
              "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this command, "
                        "gfx950 corrections of MI355X_MICROARCH.md"}
      if "valu_active_per_busy_cycle" in v:
        # VALU utilisation from the same PMC passes: SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES, ceiling 8
        prov["valu_active_per_busy_cycle"] = {"kernel": v["valu_active_per_busy_cycle"], "ceiling": 8.0,
                                              "pure_v_fma_f64_ubench": v.get("ubench_valu_active_per_busy_cycle")}
      return v["traffic_bytes"], prov
  except (OSError, ValueError, KeyError):
    pass
  return None, {"measured_in_this_run": False, "source": None}


def load_isa():
  """Instruction mix of the pair loops of THIS build (tools/isa_stats.py; hashed against csrc/), or None."""
  try:
    import isa_stats
    return isa_stats.load()
  except Exception:
    return None


def isa_key(path):
  """librmb_mobility.isa.json entry of the kernel family a product ran on (context option "last_path")."""
  return {0: "sweep_tt_wall", 1: "sym_tt_wall", 2: "symx_single_tt_wall", 3: "sym_coop_tt_wall", 4: "sym2t_tt_wall"}.get(path, "sym_tt_wall")


def executed_views(isa, sym, N, world, n_local, kern_s, issue_peak=None, key=None):
  """(executed, issue): what one launch of the wall-tt sweep really executes, priced against the fp64 vector peak
  and against the live fp64 issue ceiling.  Both <= 1 by construction."""
  st = isa["kernels"].get(key or ("sym_tt_wall" if sym else "sweep_tt_wall")) if isa else None
  if st is None:
    return None, None
  pps = st.get("pairs_per_step", 1)
  valu_launch = None
  if sym:
    tiles = (N + 63) // 64
    wave_steps = (tiles * (tiles + 1) // 2 * 64 - tiles) / world       # rotation steps (diagonal units skip k = 0)
    pair_evals = float(N) * (N - 1) / 2 / world                         # real (unpadded) unordered pairs
    if pps == 2:
      # two target blobs per lane (sym2t_kernel): a step of a unit (row pair p, tile J >= 2p + 2) evaluates two pairs; the
      # two columns at the diagonal of every row pair run through one-target loops (priced with sym_kernel's count)
      pairs_rows = (tiles + 1) // 2
      fused_steps = 64 * sum(max(tiles - 2 * p - 2, 0) for p in range(pairs_rows))
      single_steps = wave_steps * world - 2 * fused_steps
      one = isa["kernels"].get("sym_tt_wall", st)
      valu_launch = (st["valu_per_step"] * fused_steps + one["valu_per_step"] * single_steps) / world
  else:
    wave_steps = float(-(-n_local // 64)) * N
    pair_evals = float(n_local) * N
  flops_pair = st["flops_per_lane_step"] / float(pps)
  ex_tf = flops_pair * pair_evals / kern_s / 1e12
  executed = {"flops_per_pair_evaluation": flops_pair, "pair_evaluations_per_launch": pair_evals,
              "achieved": round(ex_tf, 3), "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
              "frac": round(ex_tf / FP64_VECTOR_PEAK_TFLOPS, 4),
              "source": "FMA = 2, mul/add/rsq = 1 flop, counted over the pair loop of this build's ISA "
                        "(tools/isa_stats.py, csrc sha1 %s); padded lanes not counted" % isa["source_sha1"][:12],
              "instruction_mix_per_step": st["classes"]}
  issue = None
  if issue_peak:
    valu = valu_launch if valu_launch is not None else st["valu_per_step"] * wave_steps
    issue = {"valu_wave_instr_per_launch": valu, "achieved": round(valu / kern_s / 1e9, 1), "peak": round(issue_peak, 1),
             "unit": "G wave-instr/s", "frac": round(valu / kern_s / 1e9 / issue_peak, 4),
             "peak_source": "rmb_ubench_fp64_issue: independent v_fma_f64, 4 waves per SIMD on every CU, 40 launches, "
                            "measured in this process right after the timed loop",
             "note": "%d of the %d VALU instructions per step are v_rsq_f64, which issue at ~0.3x the FMA rate "
                     "(profiles/r1_ubench_fp64_issue_rates.txt)" % (2 * pps, st["valu_per_step"])}
  return executed, issue


def _exec_frac(isa, sym, N, world, n_local, kern_s, key=None):
  ex, _ = executed_views(isa, sym, N, world, n_local, kern_s, key=key)
  return None if ex is None else ex["frac"]


# ---------------------------------------------------------------------------------------------------------------
# never hang, never lose the headline
# ---------------------------------------------------------------------------------------------------------------
class Guard(object):
  """Keeps a run from hanging and from losing its headline (one rank that leaves a collective sequence
  strands the others until the driver's limit, and the line was only printed at the very end).

    * the headline is written to stderr and to <run dir>/headline.json the moment it exists;
    * every stage after it is announced with begin(): the ranks agree (MAX of their clocks) whether it still fits the
      wall-clock budget, otherwise it is skipped on all of them;
    * a rank whose stage raises calls fail(): it drops a marker file in the run directory (a side channel that needs
      neither the GPU nor a collective);
    * a watchdog thread on every rank polls for markers and the budget four times a second; on either, rank 0 prints
      the line with what it has (plus `extras_aborted`) and every rank leaves with os._exit -- whatever collective
      the main thread is stuck in."""

  def __init__(self, rank, world, budget_s, run_dir, t0, stage_limit_s=None):
    import threading
    self.rank, self.world, self.budget, self.run_dir, self.t0 = rank, world, float(budget_s), run_dir, t0
    self.stage_limit = float(stage_limit_s) if stage_limit_s else None
    self.line = None
    self.stage = "headline"
    self.stage_t0 = None            # monotonic clock at the last begin() that let a stage start
    self._lock = threading.Lock()
    self._finished = False
    self._stop = threading.Event()
    os.makedirs(run_dir, exist_ok=True)
    self._thread = threading.Thread(target=self._watch, name="bench-watchdog", daemon=True)
    self._thread.start()

  def elapsed(self):
    """Seconds since the rank started, on the MONOTONIC clock (a wall-clock step on a freshly leased box must not
    trip the budget)."""
    return time.monotonic() - self.t0

  def publish_headline(self, line, full=False):
    self.line = line
    if self.rank == 0:
      txt = json.dumps(line if full else {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                                              "roofline") if k in line})
      sys.stderr.write("[bench %s] %s\n" % ("line so far" if full else "headline", txt))
      sys.stderr.flush()
      for path in (os.path.join(self.run_dir, "headline.json"), os.path.join(ROOT, "gpurun_out", "bench_headline_last.json")):
        try:
          if os.path.isdir(os.path.dirname(path)):
            with open(path, "w") as fh:
              fh.write(txt + "\n")
        except OSError:
          pass

  def begin(self, stage, expect_s, torch=None, dist=None, device=None):
    """All ranks: may `stage` (expected to take `expect_s`) still start?  Decided on the slowest rank's clock."""
    el = self.elapsed()
    if self.world > 1:
      t = torch.tensor([el], dtype=torch.float64, device=device)
      dist.all_reduce(t, op=dist.ReduceOp.MAX)
      el = float(t.item())
    ok = el + expect_s < self.budget
    if ok:
      self.stage = stage
      self.stage_t0 = time.monotonic()
    return ok

  def end(self):
    """The stage begin() admitted has returned: its own time limit no longer applies."""
    self.stage_t0 = None

  def fail(self, stage, exc):
    """A stage raised on this rank of a multi-rank run: tell every rank's watchdog, then wait to be taken down."""
    try:
      with open(os.path.join(self.run_dir, "abort_rank%d" % self.rank), "w") as fh:
        fh.write("rank %d failed in stage %r: %s: %s" % (self.rank, stage, type(exc).__name__, exc))
    except OSError:
      os._exit(3)
    time.sleep(15)      # the watchdogs (this rank's included) end the run within a second; if they are gone, leave anyway
    os._exit(3)

  def _emit(self, reason):
    if self.rank == 0:
      code = 3
      for _ in range(5):
        try:
          line = dict(self.line) if self.line is not None else {"error": "aborted before the headline existed"}
          line["extras_aborted"] = {"reason": reason, "stage": self.stage, "elapsed_s": round(self.elapsed(), 1),
                                    "budget_s": self.budget}
          sys.stdout.write(json.dumps(line) + "\n")
          sys.stdout.flush()
          code = 0 if "value" in line else 3
          break
        except RuntimeError:      # the main thread added a key while we copied: again
          time.sleep(0.01)
      os._exit(code)
    time.sleep(2.0)               # let rank 0 print first: a launcher may stop everybody when the first rank leaves
    os._exit(0)

  def _watch(self):
    import glob
    while not self._stop.wait(0.25):
      reason = None
      st0 = self.stage_t0
      if self.elapsed() > self.budget:
        reason = "wall-clock budget exhausted"
      elif self.stage_limit is not None and st0 is not None and time.monotonic() - st0 > self.stage_limit:
        # counted from the stage's own begin(), not from interpreter start: independent of how long import torch,
        # the rendezvous and the headline took on this box
        reason = "stage exceeded its time limit (%.0f s)" % self.stage_limit
      else:
        marks = sorted(glob.glob(os.path.join(self.run_dir, "abort_rank*")))
        if marks:
          try:
            with open(marks[0]) as fh:
              reason = fh.read() or "a rank failed"
          except OSError:
            reason = "a rank failed"
      if reason:
        with self._lock:
          if self._finished:
            return
          self._finished = True
        self._emit(reason)

  def finish(self):
    """Main thread, normal end: True if it may print (the watchdog has not taken over)."""
    with self._lock:
      taken_over = self._finished
      self._finished = True
    if taken_over:          # the watchdog is printing / leaving: do not print a second line
      time.sleep(15)
      os._exit(3)
    self._stop.set()
    return True


def _inject(stage, rank):
  """Test hook (tests/test_gpu_distributed.py): RMB_BENCH_INJECT="fail:<rank>:<stage>" raises in that rank's stage,
  "hang:<rank>:<stage>" blocks it."""
  spec = os.environ.get("RMB_BENCH_INJECT", "")
  if not spec:
    return
  what, r, st = spec.split(":")
  if int(r) == rank and st == stage:
    if what == "fail":
      raise RuntimeError("injected failure")
    if what == "hang":
      time.sleep(3600)


def rank_main(args):
  t_start = time.monotonic()
  import torch
  import torch.distributed as dist
  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if world != args.gpus:
    raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
  # rehearsal on a 1-GPU box: RMB_BENCH_BACKEND=gloo lets several ranks share cuda:0 (RCCL refuses that)
  backend_name = os.environ.get("RMB_BENCH_BACKEND", "nccl")
  n_dev = torch.cuda.device_count()
  if backend_name == "nccl" and world > n_dev:
    raise SystemExit("bench.py: --gpus %d but only %d device(s) visible" % (world, n_dev))
  device = torch.device("cuda:%d" % (local_rank % n_dev))
  torch.cuda.set_device(device)
  run_dir = os.environ.get("RMB_BENCH_RUN_DIR") or os.path.join(
      "/tmp", "rmb_bench_%d_%s" % (os.getppid() if world > 1 else os.getpid(), os.environ.get("MASTER_PORT", "0")))
  guard = Guard(rank, world, args.budget_s, run_dir, t_start, stage_limit_s=args.stage_limit_s)
  if world > 1:
    if backend_name == "nccl":
      dist.init_process_group("nccl", device_id=device)
    else:
      dist.init_process_group(backend_name)

  from rigidmultiblobswall_amd.distributed import HipBackend, ShardedMobility
  backend = HipBackend(device)
  # HIP events around every 4th sweep of the timed region: an event pair serialises 4-8 us around a 190 us launch
  # (tools/experiments/exp_graph.py), so bracketing every launch would lower `value` by ~4 %; the sample gives kernel_ms_avg
  backend.ctx.set_option("timing", TIMING_STRIDE)
  for kv in args.ctx_option:
    k_, v_ = kv.split("=")
    backend.ctx.set_option(k_, int(v_))
  sm = ShardedMobility(backend, device=device)

  N = args.blobs
  _inject("headline", rank)       # test hook: a rank that never produces its headline (cut by the whole-run budget)
  res = run_config(torch, dist, sm, backend, N, args.steps, args.warmup, world, rank, device, "pair", args.prewarm_ms,
                   unprimed=True)
  ms_per_step = 1e3 * res["dt"] / args.steps
  value = args.steps / res["dt"]
  sym = res["launch"]["chunks"] == 0
  issue_peak = backend.ctx.ubench_fp64_issue(40)      # same process, clocks still primed by the timed loop
  if world > 1:
    t = torch.tensor([issue_peak], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    issue_peak = float(t.item())

  # ---- roofline of the dominant kernel: one rank's launch covers 1/world of the pairs -------------------------
  kern_s = res["kern_ms"] * 1e-3
  pairs_ordered = float(N) * N / world
  alg_tf = FLOPS_PER_PAIR["tt_wall"] * pairs_ordered / kern_s / 1e12
  alg_bytes = 48.0 * N + 24.0 * (res["n_local"] if world == 1 else N)
  traffic, traffic_src = committed_traffic(N, sym) if world == 1 else (None, {"measured_in_this_run": False, "source": None})
  isa = load_isa()
  executed, issue = executed_views(isa, sym, N, world, res["n_local"], kern_s, issue_peak, key=isa_key(res["path"]))
  if executed is None:
    # no instruction count for this build (tools/isa_stats.py needs hipcc): the utilisation cannot be priced; say so
    # rather than fall back to the algorithmic unit, which is not a utilisation
    executed = {"flops_per_pair_evaluation": 0, "pair_evaluations_per_launch": 0, "achieved": None, "frac": None,
                "error": "librmb_mobility.isa.json missing and could not be regenerated"}
  roofline = {
      "bound": "valu_fp64",
      "kernel": {3: "rmb::sym_coop_kernel<TT,wall> (each unordered pair once, both blobs updated; the four waves of a workgroup "
                    "share one staged tile and one flush per tile)",
                 4: "rmb::sym2t_kernel<TT,wall> (each unordered pair once, both blobs updated; two target blobs per lane share one "
                    "record read and one set of LDS adds per rotation step)",
                 1: "rmb::sym_kernel<TT,wall> (each unordered pair once, both blobs updated)"}.get(res["path"], "rmb::sweep_kernel<TT,wall>"),
      # utilisation, <= 1 by construction: fp64 flops this kernel EXECUTES per launch / kernel time / fp64 vector peak
      "achieved": executed["achieved"], "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": executed["frac"],
      "unit_of_work": "EXECUTED: %g fp64 flops per pair evaluation (FMA = 2, mul/add/rsq = 1, counted over the pair loop "
                      "of this build's ISA) x %.0f pair evaluations per launch (N(N-1)/2 unordered pairs / ranks for the "
                      "symmetric kernel)" % (executed["flops_per_pair_evaluation"], executed["pair_evaluations_per_launch"]),
      "kernel_ms_avg": round(res["kern_ms"], 5),
      "kernel_ms_avg_source": "HIP events on the launch stream around every %d-th sweep launch of the timed region "
                              "(%d launches sampled)" % (TIMING_STRIDE, -(-args.steps // TIMING_STRIDE)),
      "launch": res["launch"],
      "executed": executed,
      # throughput in the reference's unit: 211 as-written flops per ORDERED pair x N^2.  The symmetric
      # kernel needs 3.4x fewer flops for the same result, so this exceeds the peak: a speed in the reference's
      # currency, NOT a utilisation -- never read it as `frac`
      "algorithmic_throughput": {"flops_per_ordered_pair": FLOPS_PER_PAIR["tt_wall"], "ordered_pairs_per_launch": pairs_ordered,
                                 "value": round(alg_tf, 3), "unit": "TFLOP/s (reference's as-written arithmetic)",
                                 "ratio_to_fp64_vector_peak": round(alg_tf / FP64_VECTOR_PEAK_TFLOPS, 4)},
      "issue": issue,
      "traffic": traffic, "traffic_provenance": traffic_src,
      "hbm": {"algorithmic_bytes_per_launch": alg_bytes,
              "achieved": round(alg_bytes / kern_s / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
              "frac": round(alg_bytes / kern_s / 1e9 / HBM_PEAK_GBS, 6)},
  }

  line = {
      "metric": "RPY-wall M.f matvecs/sec (single_wall_mobility_trans_times_force, N_blobs=%d)" % N,
      "value": round(value, 3), "unit": "matvecs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
      "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
      "dtype": "f64", "data": "synthetic",
      "config": {"workload": "configs[1]: %d random blobs above a wall (D2 cloud, 5%% volume fraction, seed 0), fp64, "
                             "single_wall_mobility_trans_times_force; vectors resident in HBM" % N,
                 "n_blobs": N,
                 "parallelism": ("single GPU" if world == 1 else
                                 "unordered blob pairs sharded over %d ranks (each pair once, both blobs updated), "
                                 "f replicated, one RCCL all-reduce of u per matvec" % world)},
      "world_size": world,
      "collective_backend": None if world == 1 else ("nccl (RCCL)" if backend_name == "nccl" else backend_name),
      "ipc_mode_legacy_env": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
      "prewarm": res["prewarm"],
      "roofline": roofline,
      "build": build_record(),
  }
  if args.ctx_option:
    line["ctx_options"] = list(args.ctx_option)      # not the defaults: an A/B or profiling pass
  if res["cold"] is not None:
    # the same K steps after the same W warm-up steps, run BEFORE the declared pre-warm on a chip that had not run the
    # kernel yet: what the driver's bare flags measure without priming (the fp64 clock is still ramping)
    cdt, ckern = res["cold"]
    line["value_unprimed"] = {"value": round(args.steps / cdt, 3), "unit": "matvecs/s", "ms_per_step": round(1e3 * cdt / args.steps, 5),
                              "kernel_ms_avg": round(ckern, 5), "steps": args.steps, "warmup": args.warmup,
                              "when": "first thing this process ran on the GPU, before `prewarm`"}
  # From here on nothing can cost the headline: it is on stderr and on disk, and the watchdog prints the line if a
  # later stage hangs or a rank fails.
  guard.publish_headline(line)

  def stage(key, expect_s, fn, single_rank_only=False):
    """One extra.  world == 1: an exception is recorded and the run goes on.  world > 1: the failing rank raises the
    abort marker and every rank leaves within a second -- no rank ever skips ahead of the others' collectives."""
    if single_rank_only and not (rank == 0 and world == 1):
      return
    if not guard.begin(key, expect_s, torch, dist, device):
      line[key] = {"skipped": "would not fit the wall-clock budget (%.0f s of %.0f s used)" % (guard.elapsed(), guard.budget)}
      return
    try:
      _inject(key, rank)
      out = fn()
      if out is not None:
        line[key] = out
    except Exception as exc:      # an extra must never cost the headline line
      line[key] = {"error": "%s: %s" % (type(exc).__name__, exc)}
      if world > 1:
        guard.fail(key, exc)
    finally:
      guard.end()

  # ---- extras ---------------------------------------------------------------------------------------------------
  def host_surface():
    # End to end through the plugin surface, the call shape of the reference's callers (mobility/mobility.py:222-252,
    # multi_bodies/multi_bodies.py:445): numpy arrays in, a new numpy array out, synchronous, PCIe-inclusive.  Never
    # `value`.  The positions stay resident while the caller passes the same r_vectors (one compare per call).
    from rigidmultiblobswall_amd import mobility as mob
    r_h, f_h = np.ascontiguousarray(res["r"]), np.ascontiguousarray(res["f"])
    for _ in range(max(args.warmup, 3)):
      u_h = mob.single_wall_mobility_trans_times_force_hip(r_h, f_h, res["eta"], res["a"])
    n_host = max(args.steps, 50)
    t0 = time.perf_counter()
    for _ in range(n_host):
      u_h = mob.single_wall_mobility_trans_times_force_hip(r_h, f_h, res["eta"], res["a"])
    dt_h = time.perf_counter() - t0
    out = {"value": round(n_host / dt_h, 3), "unit": "matvecs/s", "ms_per_call": round(1e3 * dt_h / n_host, 5),
           "calls": n_host, "function": "single_wall_mobility_trans_times_force_hip(r_vectors, force, eta, a)",
           "devices": mob.active_devices(N), "bytes_over_pcie_per_call": 48 * N,
           "max_abs_diff_vs_timed_device_output": float(np.max(np.abs(u_h - res["out"].cpu().numpy())))}
    # where the time of one call goes: the position compare + option set of the Python wrapper, the C call split by
    # the library's own host clock (rmb_last_host_timing), the sweep by HIP events, the rest is Python / ctypes
    ctx = mob._context(N)
    if hasattr(ctx, "last_host_timing"):
      n_b = 200
      t0 = time.perf_counter()
      for _ in range(n_b):
        mob._bind_positions(r_h, res["a"], np.zeros(3), True)
      bind_us = (time.perf_counter() - t0) / n_b * 1e6
      ctx.set_option("timing", 1)
      ctx.timing_reset()
      acc = np.zeros(4)
      for _ in range(n_b):
        mob.single_wall_mobility_trans_times_force_hip(r_h, f_h, res["eta"], res["a"])
        ht = ctx.last_host_timing()
        acc += [ht["upload_us"], ht["launch_us"], ht["wait_and_download_us"], ht["c_call_us"]]
      acc /= n_b
      k_us = float(np.mean(ctx.timing_collect(n_b))) * 1e3
      ctx.set_option("timing", 0)
      total_us = 1e3 * out["ms_per_call"]
      out["breakdown_us"] = {
          "position_compare_and_options": round(bind_us, 1),
          "upload_24N_bytes_host_memcpy_and_pull_kernel_enqueue": round(acc[0], 1), "kernel_enqueue": round(acc[1], 1),
          "sweep_kernel_hip_events": round(k_us, 1),
          "finalize_into_mapped_memory_stream_wait_and_memcpy_24N_bytes": round(max(acc[2] - k_us, 0.0), 1),
          "python_ctypes_numpy_alloc": round(max(total_us - bind_us - acc[3], 0.0), 1),
          "c_call_total": round(acc[3], 1), "call_total": round(total_us, 1),
          "note": "upload / enqueue / wait+download: host wall clock inside rmb_matvec (rmb_last_host_timing), measured with "
                  "per-launch HIP events on (adds ~5 us to enqueue and ~15 us to the call, so c_call_total exceeds call_total, which "
                  "is timed without them).  Up to 768 KB per vector the input goes through page-locked mapped memory + a pull kernel on "
                  "the product's queue and the finalize kernel stores the result straight into mapped memory: one stream wait and a "
                  "host memcpy, no copy-queue command either way (options host_zero_copy / host_zero_copy_in)"}
    mob.reset()
    return out
  if not args.no_host_surface:
    stage("host_surface", 10, host_surface, single_rank_only=True)

  def traffic_live():
    # HBM-side traffic of the dominant kernel measured NOW: two child passes of this same script under
    # `rocprofv3 --pmc` (FETCH_SIZE, then WRITE_SIZE, separately, the program directly after `--`, as
    # MI355X_MICROARCH.md prescribes; gfx950: FETCH_SIZE counts half of the wide coalesced reads -> doubled; KB -> x1024).
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
      return {"skipped": "rocprofv3 not on PATH: roofline.traffic stays the committed figure"}
    if "rocprofiler" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
      return {"skipped": "this run is itself being profiled: no nested rocprofv3 passes"}
    fam = {3: "sym_coop_kernel<0, true, false>", 1: "sym_kernel<0, true, false>", 4: "sym2t_kernel<0, true>"}.get(res["path"], "sweep_kernel<0, true, false>")
    out_dir = tempfile.mkdtemp(prefix="rmb_pmc_")
    got = {}
    try:
      for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out_dir, counter)
        cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
               "--blobs", str(N), "--steps", "20", "--warmup", "2", "--prewarm-ms", "0", "--no-sweep", "--no-cpu", "--no-host-surface"]
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "RMB_BENCH_INJECT"):
          env.pop(k, None)
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=150, cwd="/tmp", env=env)
        vals = []
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
          with open(f) as fh:
            for row in csv.DictReader(fh):
              if row.get("Counter_Name") == counter and fam in row.get("Kernel_Name", ""):
                vals.append(float(row["Counter_Value"]))
        if p.returncode != 0 or not vals:
          return {"error": "rocprofv3 --pmc %s pass gave no samples of %s (rc %d): %s" % (counter, fam, p.returncode, p.stderr[-300:])}
        got[counter] = (sum(vals) / len(vals), len(vals))
    finally:
      shutil.rmtree(out_dir, ignore_errors=True)
    live = (2.0 * got["FETCH_SIZE"][0] + got["WRITE_SIZE"][0]) * 1024.0
    committed = line["roofline"].get("traffic")
    line["roofline"]["traffic"] = int(round(live))
    line["roofline"]["traffic_provenance"] = {
        "measured_in_this_run": True, "kernel": "rmb::" + fam,
        "method": "two child passes of this script under rocprofv3 --pmc (FETCH_SIZE | WRITE_SIZE separately, program directly "
                  "after `--`): per-launch averages; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 correction of MI355X_MICROARCH.md)",
        "FETCH_SIZE_kb": round(got["FETCH_SIZE"][0], 2), "WRITE_SIZE_kb": round(got["WRITE_SIZE"][0], 2),
        "launches_sampled": [got["FETCH_SIZE"][1], got["WRITE_SIZE"][1]],
        "committed_figure_for_comparison": committed, "committed_source": traffic_src.get("source"),
        "ratio_to_algorithmic_bytes": round(live / alg_bytes, 1)}
    return None
  if not args.no_sweep and not args.no_host_surface:
    stage("traffic_live", 60, traffic_live, single_rank_only=True)

  def multi_device_surface():
    # More than one device visible to a one-rank run (a whole node): the same plugin call on the single-process
    # multi-device engine (mobility.set_devices(all): pair shards on every GPU, fixed-order slice reduction over xGMI).
    # Run in a CHILD process with a timeout: it touches devices this process does not own, and a fault there must not
    # take the line down with it.
    # With ONE device visible the same probe lists it twice: the engine's code path (two shard contexts, two streams,
    # the slice reduction) on this box -- a rehearsal that shows its overhead, not a speed-up.
    rehearsal = n_dev < 2
    cmd = [sys.executable, os.path.join(ROOT, "tools", "multi_surface_probe.py"), "0,0" if rehearsal else str(n_dev)] + \
          (["24576"] if rehearsal else [])
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=150)
    rows = [l for l in p.stdout.split("\n") if l.startswith("{")]
    if p.returncode != 0 or not rows:
      return {"error": "probe exited with %d: %s" % (p.returncode, (p.stderr or p.stdout)[-400:])}
    out = json.loads(rows[-1])
    out["rehearsal_on_one_gpu"] = rehearsal
    if rehearsal:
      out["note"] = ("one device visible: it is listed twice, so the two shards share the chip and `speedup` < 1 is the engine's "
                     "hand-off overhead; on a node the list is every visible device")
    return out

  def rccl_one_rank():
    # RCCL cannot meet a second rank on a one-GPU box, but the N > 1 step's all-reduce can RUN there: a one-rank "nccl"
    # group in a CHILD process (this one holds no process group at N = 1) times the step with and without it.
    cmd = [sys.executable, os.path.join(ROOT, "tools", "rccl_one_rank_probe.py"), str(N)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=110)
    rows = [l for l in p.stdout.split("\n") if l.startswith("{")]
    if p.returncode != 0 or not rows:
      return {"error": "probe exited with %d: %s" % (p.returncode, (p.stderr or p.stdout)[-400:])}
    return json.loads(rows[-1])

  def parity_and_cpu():
    from oracle import oracle
    r, f, eta, a = res["r"], res["f"], res["eta"], res["a"]
    # parity guard on the very output that was timed (subset of targets, all sources)
    tg = np.random.RandomState(1).choice(N, min(64, N), replace=False)
    r_eff, bdiag, _ = oracle.wall_regularisation(r, a)
    ref = oracle.raw_matvec_targets("tt", 1, r_eff, f, eta, a, tg)
    got = res["out"].cpu().numpy().reshape(-1, 3)[tg].reshape(-1)
    line["parity_rel_err_vs_oracle"] = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    # CPU baseline: same workload, fast-math OpenMP port, bounded to ~10-30 s.  The OpenMP team is sized to the CPUs
    # this process may really use (affinity capped by the cgroup quota), not to every hardware thread of the host.
    hw_threads = oracle.num_threads()
    cores = min(hw_threads, oracle.usable_cpus())
    oracle.set_num_threads(cores)
    oracle.single_wall_mobility_trans_times_force_oracle(r, f, eta, a, fast=True)   # warm-up
    times = []
    t_start_ = time.perf_counter()
    while len(times) < 40 and (time.perf_counter() - t_start_ < 12.0 or len(times) < 3):
      t0 = time.perf_counter()
      oracle.single_wall_mobility_trans_times_force_oracle(r, f, eta, a, fast=True)
      times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    cb = {"value": round(1.0 / med, 4), "unit": "matvecs/s", "cores": cores, "kind": "port",
          "sample": "%d full matvecs of the same %d-blob workload (median), oracle C port "
                    "-O3 -ffast-math -fopenmp, %d OpenMP threads (host exposes %d hardware threads)"
                    % (len(times), N, cores, hw_threads)}
    # the CPU baseline at every size <= 1e5, 1e6 extrapolated with N^2 and flagged.  Bounded: the
    # larger sizes run fewer repetitions (one matvec at 1e5 blobs is ~10-15 s on 16 cores).
    by_size = [{"n_blobs": N, "matvecs_per_s": cb["value"], "reps": len(times), "extrapolated": False}]
    if not args.no_sweep:
      last = None
      for nb, reps in ((24576, 3), (32000, 3), (100000, 2)):
        if guard.elapsed() + reps * med * (nb / float(N)) ** 2 * 1.3 > 0.6 * guard.budget:
          by_size.append({"n_blobs": nb, "matvecs_per_s": round(1.0 / (med * (nb / float(N)) ** 2), 5), "reps": 0, "extrapolated": True})
          continue
        rb, fb, eb, ab = d2_cloud(nb, seed=0)
        ts = []
        for _ in range(reps):
          t0 = time.perf_counter()
          oracle.single_wall_mobility_trans_times_force_oracle(rb, fb, eb, ab, fast=True)
          ts.append(time.perf_counter() - t0)
        last = (nb, float(np.median(ts)))
        by_size.append({"n_blobs": nb, "matvecs_per_s": round(1.0 / last[1], 5), "reps": reps, "extrapolated": False})
      base_n, base_t = last if last is not None else (N, med)
      for nb in (262144, 1000000):
        by_size.append({"n_blobs": nb, "matvecs_per_s": round(1.0 / (base_t * (nb / float(base_n)) ** 2), 6), "reps": 0,
                        "extrapolated": True, "from_n_blobs": base_n})
    cb["by_size"] = by_size
    line["cpu_baseline"] = cb
    return None
  if not args.no_cpu:
    stage("cpu_baseline", 75 if not args.no_sweep else 20, parity_and_cpu, single_rank_only=True)

  def decompositions():
    # both decompositions at 1e4 / 1e5 / 1e6 blobs (the metric is "... vs N_blobs at 1/2/4/8 MI355X")
    dec = {"pair_shard_allreduce": [], "target_shard_allgather": []}
    # the sizes of SURVEY 8(d): 1e4, 24 576 (configs[2]), 3.2e4, 1e5, 262 144 (configs[4]), 1e6 (configs[3])
    for nb, st_, wu in ((10000, 50, 5), (24576, 20, 3), (32000, 20, 3), (100000, 5, 1), (262144, 3, 1), (1000000, 2, 1)):
      rs = run_config(torch, dist, sm, backend, nb, st_, wu, world, rank, device, "pair", 100.0 if nb == 10000 else 0.0)
      tb, prov = committed_traffic(nb, True) if world == 1 else (None, {})
      row = {
          "n_blobs": nb, "matvecs_per_s": round(st_ / rs["dt"], 4), "ms_per_step": round(1e3 * rs["dt"] / st_, 4),
          "kernel_ms_avg": round(rs["kern_ms"], 4), "allreduce_bytes": 0 if world == 1 else 24 * nb,
          "algorithmic_tflops_all_ranks": round(211.0 * float(nb) * nb / (rs["dt"] / st_) / 1e12, 2),
          "executed_frac": _exec_frac(isa, rs["launch"]["chunks"] == 0, nb, world, rs["n_local"], rs["kern_ms"] * 1e-3, isa_key(rs["path"])),
          "kernel_family": {0: "one-sided sweep", 1: "symmetric, per wave", 3: "symmetric, workgroup-cooperative",
                            4: "symmetric, two target blobs per lane"}.get(rs["path"]),
          "hbm_algorithmic_gbps": round(72.0 * nb / (rs["kern_ms"] * 1e-3) / 1e9, 4), "launch": rs["launch"]}
      if tb is not None:
        # HBM-side traffic of one launch from the committed rocprofv3 --pmc passes (not measured in this run) over
        # this run's kernel time: what the metric's "HBM GB/s vs N_blobs" is for a VALU-bound kernel
        row["hbm_traffic_gbps"] = round(tb / (rs["kern_ms"] * 1e-3) / 1e9, 1)
        row["hbm_traffic_source"] = prov.get("source")
      dec["pair_shard_allreduce"].append(row)
    for nb, st_, wu in ((10000, 50, 5), (100000, 3, 1), (1000000, 1, 1)):
      rs = run_config(torch, dist, sm, backend, nb, st_, wu, world, rank, device, "target", 100.0 if nb == 10000 else 0.0)
      dec["target_shard_allgather"].append({
          "n_blobs": nb, "matvecs_per_s": round(st_ / rs["dt"], 4), "ms_per_step": round(1e3 * rs["dt"] / st_, 4),
          "kernel_ms_avg": round(rs["kern_ms"], 4), "allgather_bytes": 0 if world == 1 else 24 * nb,
          "executed_frac": _exec_frac(isa, False, nb, world, rs["n_local"], rs["kern_ms"] * 1e-3),
          "algorithmic_tflops_all_ranks": round(211.0 * float(nb) * nb / (rs["dt"] / st_) / 1e12, 2),
          "launch": rs["launch"]})
    return dec
  if not args.no_sweep:
    stage("decompositions", 40, decompositions)

  def config3_gmres():
    # BASELINE.json configs[2]: 2048 rollers x 12-blob shells, full GMRES mobility solve on 1 GPU (reported
    # beside the headline, not part of `value`)
    from rigidmultiblobswall_amd import structures as st
    from rigidmultiblobswall_amd.rigid import RigidSuspension
    R, eta3 = 1.0155, 0.957e-3
    shell = st.icosahedron_shell(0.792079207921 * R)
    a3 = st.min_blob_separation(shell) / 2
    nb = 2048
    loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=5)
    FT = np.zeros((nb, 6)); FT[:, 2] = -0.05; FT[:, 4] = 1.0
    rs = RigidSuspension([shell] * nb, loc, quat, a3, eta3, device=device)
    for _ in range(2):
      rs.solve_mobility_problem(force_torque=FT, tol=1e-8)      # warm-up (library initialisation, clocks)
    torch.cuda.synchronize(device)
    n_timed = 3
    t0 = time.perf_counter()
    for _ in range(n_timed):
      U, lam, info = rs.solve_mobility_problem(force_torque=FT, tol=1e-8)
    torch.cuda.synchronize(device)
    out = {"bodies": nb, "blobs": rs.n_blobs, "tolerance": 1e-8, "iterations": info["iterations"],
           "residual": float(info["residual"]), "ms_per_solve": round(1e3 * (time.perf_counter() - t0) / n_timed, 3),
           "solves_timed": n_timed,
           "helpers": "csrc/rmb_krylov.hip + rmb_rigid.hip (block products, fused Gram-Schmidt, per-body factors)" if rs._native_blocks() is not None else "torch operations"}
    # the same solve by iterative refinement with fp32 inner products (RigidSuspension.solve_mixed_precision): same
    # tolerance on the true fp64 residual; an option, reported beside the reference's algorithm above
    rs.solve_mobility_problem(force_torque=FT, tol=1e-8, mixed_precision=True)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(n_timed):
      U2, lam2, info2 = rs.solve_mobility_problem(force_torque=FT, tol=1e-8, mixed_precision=True)
    torch.cuda.synchronize(device)
    out["mixed_precision_option"] = {
        "ms_per_solve": round(1e3 * (time.perf_counter() - t0) / n_timed, 3), "inner_iterations_fp32": info2["iterations"],
        "outer_iterations_fp64": info2["outer_iterations"], "residual_fp64": float(info2["residual"]),
        "velocity_rel_diff_vs_fp64_solve": float(np.linalg.norm(U2 - U) / np.linalg.norm(U))}
    rs.close()
    return out
  if not args.no_sweep:
    stage("config3_gmres", 15, config3_gmres, single_rank_only=True)

  def two_targets_ab():
    # Same box, same process, alternating: the headline product on the one-target symmetric kernels (round 3 / early
    # round 4: cooperative up to four resident rounds, per wave above) and with two target blobs per lane (the default).
    # Boxes of the pool differ by several per cent, so a kernel change is only visible in a same-run comparison.
    ctx = backend.ctx
    r_, f_, eta_, a_ = d2_cloud(N, seed=0)
    ctx.set_positions(torch.as_tensor(r_.reshape(-1), device=device), a_, None, wall=True)
    fd_ = torch.as_tensor(f_.reshape(-1), device=device)
    out_ = torch.empty(3 * N, dtype=torch.float64, device=device)
    keep_timing = ctx.get_option("timing")
    ctx.set_option("timing", 1)
    res_ab = {0: [], 1: []}
    try:
      # the stages before this one leave the GPU idle for seconds (CPU baseline, child processes): prime the clocks as the
      # headline does, or whichever mode is measured first pays the ramp
      t_prime = time.perf_counter()
      while time.perf_counter() - t_prime < 0.3:
        for _ in range(20):
          ctx.matvec_device("tt", fd_, eta_, out=out_)
        torch.cuda.synchronize(device)
      for rnd in range(4):
        for mode in ((0, 1) if rnd % 2 == 0 else (1, 0)):
          ctx.set_option("sym_two_targets", mode)
          for _ in range(5):
            ctx.matvec_device("tt", fd_, eta_, out=out_)
          torch.cuda.synchronize(device)
          ctx.timing_reset()
          for _ in range(100):
            ctx.matvec_device("tt", fd_, eta_, out=out_)
          torch.cuda.synchronize(device)
          res_ab[mode].append(float(np.median(ctx.timing_collect(100))) * 1e3)
    finally:
      ctx.set_option("sym_two_targets", 1)
      ctx.set_option("timing", keep_timing)
    one, two = float(np.median(res_ab[0])), float(np.median(res_ab[1]))
    return {"n_blobs": N, "kernel_us_one_target_per_lane": round(one, 2), "kernel_us_two_targets_per_lane": round(two, 2),
            "speedup": round(one / two, 4), "rounds": 4, "launches_per_round": 100,
            "per_round_us": {"one_target": [round(x, 2) for x in res_ab[0]], "two_targets": [round(x, 2) for x in res_ab[1]]},
            "note": "median of the per-round medians; HIP events around every launch (adds ~8 us to both); clocks primed for "
                    "0.3 s first, the order of the two modes alternates from round to round; option sym_two_targets 0 / 1"}
  if world == 1 and not args.no_sweep and not any(kv.startswith("sym_two_targets=") for kv in args.ctx_option):
    stage("two_targets_ab", 3, two_targets_ab, single_rank_only=True)

  def two_targets_ops_ab():
    # Round 5: the multi-block / multi-vector operations and the pseudo-periodic products with two target blobs per lane
    # (csrc/symx2t_kernels.h, the default) against the one-target kernels, same box, same process, alternating rounds.
    # These are what the configs[4] steps are made of (rollers: fused row + grand; mobility_pycuda.py:1266-1391,
    # quaternion_integrator_rollers.py:1114-1121; periodic images mobility_numba.py:170-197).
    ctx = backend.ctx
    keep_timing = ctx.get_option("timing")
    ctx.set_option("timing", 1)
    rows = []
    try:
      for nb, periodic, reps in ((10000, False, 40), (100000, False, 4), (24576, True, 6)):
        r_, f_, eta_, a_ = d2_cloud(nb, seed=0)
        box = (nb * (4.0 / 3.0) * np.pi * a_ ** 3 / 0.05) ** (1.0 / 3.0)
        L_ = np.array([box, box, 0.0]) if periodic else None
        ctx.set_positions(torch.as_tensor(r_.reshape(-1), device=device), a_, L_, wall=True)
        vs = [torch.as_tensor(x.reshape(-1), device=device) for x in (f_, np.random.RandomState(1).randn(nb, 3))]
        outs = [torch.empty(3 * nb, dtype=torch.float64, device=device) for _ in range(2)]
        calls = [("fused row: u = M_tt f + M_tr tau", lambda: ctx.matvec_op_device("velocity_from_force_torque", vs, eta_, outs=outs[:1])),
                 ("grand: [u; w] = M [f; tau]", lambda: ctx.matvec_op_device("grand", vs, eta_, outs=outs))]
        if periodic:
          calls = [("tt (9 images per pair)", lambda: ctx.matvec_device("tt", vs[0], eta_, out=outs[0]))] + calls[:1]
        t_prime = time.perf_counter()
        while time.perf_counter() - t_prime < 0.2:
          calls[0][1](); torch.cuda.synchronize(device)
        for label, call in calls:
          res_ab, path = {0: [], 1: []}, {}
          for rnd in range(2):
            for mode in ((0, 1) if rnd % 2 == 0 else (1, 0)):
              ctx.set_option("sym_two_targets", mode)
              for _ in range(2):
                call()
              torch.cuda.synchronize(device); ctx.timing_reset()
              for _ in range(reps):
                call()
              torch.cuda.synchronize(device)
              res_ab[mode].append(float(np.median(ctx.timing_collect(reps))) * 1e3)
              path[mode] = ctx.get_option("last_path")
          one, two = float(np.median(res_ab[0])), float(np.median(res_ab[1]))
          rows.append({"n_blobs": nb, "periodic_xy": periodic, "operation": label, "kernel_us_one_target_per_lane": round(one, 2),
                       "kernel_us_two_targets_per_lane": round(two, 2), "speedup": round(one / two, 4),
                       "kernel_family": {1: "per wave", 3: "cooperative", 4: "two targets per lane"}.get(path[1])})
    finally:
      ctx.set_option("sym_two_targets", 1)
      ctx.set_option("timing", keep_timing)
    return {"rows": rows, "note": "median of 2 alternating rounds, HIP events around every sweep launch; option sym_two_targets 0 / 1"}
  if world == 1 and not args.no_sweep and not any(kv.startswith("sym_two_targets=") for kv in args.ctx_option):
    stage("two_targets_ops_ab", 8, two_targets_ops_ab, single_rank_only=True)

  def small_deck_steps():
    # The reference's usual sizes (tens to hundreds of bodies): whole time steps of the rigid-multiblob integrators, where
    # the loop AROUND the sweep decides (helper kernels + captured Arnoldi iterations, profiles/r4_gmres_graph.txt).
    from rigidmultiblobswall_amd import structures as st
    from rigidmultiblobswall_amd.rigid_integrator import RigidIntegrator
    R, eta_s = 1.0155, 0.957e-3
    shell = st.icosahedron_shell(0.792079207921 * R)
    a_s = st.min_blob_separation(shell) / 2
    rows = []
    for nb, scheme, tol, n_steps in ((64, "deterministic_adams_bashforth", 1e-8, 30), (64, "stochastic_Slip_Trapz", 1e-6, 10),
                                     (256, "deterministic_adams_bashforth", 1e-8, 30)):
      loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=5)
      integ = RigidIntegrator([shell] * nb, loc, quat, scheme, a_s, eta_s, tolerance=tol, device=device, seed=9)
      integ.kT, integ.g = 0.0041419464, 0.0024892 * 12
      integ.repulsion_strength_wall, integ.debye_length_wall = 0.0165677856, 0.0656
      integ.repulsion_strength, integ.debye_length = 0.0165677856, 0.0656
      for step in range(4):
        integ.advance_time_step(0.002, step=step)
      torch.cuda.synchronize(device)
      it0 = (integ.det_iterations_count, integ.stoch_iterations_count)
      t0 = time.perf_counter()
      for step in range(4, 4 + n_steps):
        integ.advance_time_step(0.002, step=step)
      torch.cuda.synchronize(device)
      rows.append({"bodies": nb, "blobs": 12 * nb, "scheme": scheme, "solver_tolerance": tol, "steps": n_steps,
                   "ms_per_step": round(1e3 * (time.perf_counter() - t0) / n_steps, 3),
                   "gmres_iterations_per_step": round((integ.det_iterations_count - it0[0]) / n_steps, 1),
                   "lanczos_iterations_per_step": round((integ.stoch_iterations_count - it0[1]) / n_steps, 1),
                   "rejected_steps": integ.invalid_configuration_count})
      integ.close()
    # single-blob rollers (quaternion_integrator_rollers.py schemes): the Brownian Adams-Bashforth step of a small deck, whose cost is
    # the unpreconditioned Lanczos forcing (rmb_lanczos_device; tensor operations under a Python loop in round 4: 7.2 ms per step)
    from rigidmultiblobswall_amd.rollers import RollersIntegrator
    n_r, a_r, n_steps = 1000, 0.656, 20
    loc_r, _, _ = st.roller_monolayer(n_r, radius=a_r, seed=7)
    integ = RollersIntegrator(loc_r, "stochastic_adams_bashforth_rollers", a_r, 1.0e-3, tolerance=1e-6, device=device, seed=11)
    integ.kT, integ.g = 0.0041419464, 0.0024892
    integ.repulsion_strength = integ.repulsion_strength_wall = 0.0165677856
    integ.debye_length = integ.debye_length_wall = 0.0656
    integ.omega_one_roller = np.array([0.0, 62.8, 0.0])
    integ.report_rejections = False
    for _ in range(3):
      integ.advance_time_step(0.016)
    torch.cuda.synchronize(device)
    l0, t0 = integ.stoch_iterations_count, time.perf_counter()
    for _ in range(n_steps):
      integ.advance_time_step(0.016)
    torch.cuda.synchronize(device)
    rows.append({"bodies": n_r, "blobs": n_r, "scheme": "stochastic_adams_bashforth_rollers", "solver_tolerance": 1e-6, "steps": n_steps,
                 "ms_per_step": round(1e3 * (time.perf_counter() - t0) / n_steps, 3), "gmres_iterations_per_step": 0.0,
                 "lanczos_iterations_per_step": round((integ.stoch_iterations_count - l0) / n_steps, 1),
                 "rejected_steps": integ.invalid_configuration_count})
    integ.close()
    return {"decks": rows, "note": "12-blob shells in a monolayer (rows 1-3) and single-blob rollers (row 4); defaults: the GMRES and Lanczos "
                                    "loops inside the library (rmb_rigid_gmres_device / rmb_rigid_lanczos_device / rmb_lanczos_device, four O(N) "
                                    "launches per iteration); round 4, captured graphs / tensor operations: 1.69 / 6.33 / 2.25 / 7.2 ms per step"}
  if not args.no_sweep:
    stage("small_deck_steps", 8, small_deck_steps, single_rank_only=True)

  def sync_max(dt):
    if world > 1:
      t = torch.tensor([dt], dtype=torch.float64, device=device)
      dist.all_reduce(t, op=dist.ReduceOp.MAX)
      return float(t.item())
    return dt

  def fence():
    torch.cuda.synchronize(device)
    if world > 1:
      dist.barrier()

  def config5_rollers():
    # BASELINE.json configs[4] recipe: 2.6e5 single-blob rollers, Brownian Adams-Bashforth steps = forces kernel +
    # M_tt F + M_tr T + Lanczos M^{1/2} z + 2 random-finite-difference products per step; physical parameters of
    # multi_bodies/examples/rollers/inputfile_rollers.dat.  On N ranks the same replicated stepper runs on every
    # rank and only the pair sweeps are divided (ReplicatedContext).  Reported beside the headline; the full
    # 100-step run is tools/run_config5.py (profiles/).
    from rigidmultiblobswall_amd import structures as st
    from rigidmultiblobswall_amd.distributed import ReplicatedContext
    from rigidmultiblobswall_amd.rollers import RollersIntegrator
    n5, a5 = 262144, 0.656
    loc5, _, _ = st.roller_monolayer(n5, radius=a5, seed=7)
    integ = RollersIntegrator(loc5, "stochastic_adams_bashforth_rollers", a5, 1.0e-3, tolerance=1e-3, device=device,
                              ctx=ReplicatedContext(sm), seed=11)
    integ.kT, integ.g = 0.0041419464, 0.0024892
    integ.repulsion_strength = integ.repulsion_strength_wall = 0.0165677856
    integ.debye_length = integ.debye_length_wall = 0.0656
    integ.omega_one_roller = np.array([0.0, 62.8, 0.0])
    integ.report_rejections = False     # stdout carries ONE JSON line; rejected steps are counted on it
    integ.advance_time_step(0.016)      # warm-up (first step is forward Euler)
    fence()
    p0, l0, n5_steps = integ.mobility_products, integ.stoch_iterations_count, 2
    t0 = time.perf_counter()
    for _ in range(n5_steps):
      integ.advance_time_step(0.016)
    fence()
    dt5 = sync_max(time.perf_counter() - t0)
    out = {"rollers": n5, "scheme": integ.scheme, "lanczos_tolerance": 1e-3, "steps": n5_steps,
           "s_per_step": round(dt5 / n5_steps, 4),
           "mobility_products_per_step": (integ.mobility_products - p0) / n5_steps,
           "lanczos_iterations_per_step": (integ.stoch_iterations_count - l0) / n5_steps,
           "rejected_steps": integ.invalid_configuration_count}
    # the same two steps with the reference GPU module's precision switch on 'single' (fp32 twins of the fused row and
    # the grand mobility, fp64 accumulation): an option, reported beside the double-precision figure
    integ.precision = "single"
    l1 = integ.stoch_iterations_count
    fence()
    t0 = time.perf_counter()
    for _ in range(n5_steps):
      integ.advance_time_step(0.016)
    fence()
    dt5s = sync_max(time.perf_counter() - t0)
    integ.precision = "double"
    out["single_precision_option"] = {
        "s_per_step": round(dt5s / n5_steps, 4), "lanczos_iterations_per_step": (integ.stoch_iterations_count - l1) / n5_steps,
        "rejected_steps": integ.invalid_configuration_count}
    return out
  if not args.no_sweep:
    stage("config5_rollers", 40, config5_rollers)

  def config5_multiblob():
    # The same config with rigid multiblobs instead of single-blob rollers: 21845 shells x 12 blobs = 262140 blobs,
    # stochastic_Slip_Trapz (3 GMRES rigid solves + preconditioned Lanczos + forces kernel per step), parameters of
    # multi_bodies/examples/Spectral_Multiblob_Roller/inputfile_2048_rollers.dat (tolerance 1e-4, constant torque).
    import math
    from rigidmultiblobswall_amd import structures as st
    from rigidmultiblobswall_amd.distributed import ReplicatedContext
    from rigidmultiblobswall_amd.rigid_integrator import RigidIntegrator
    R5, eta5, nb5 = 1.0155, 0.957e-3, 21845
    shell5 = st.icosahedron_shell(0.792079207921 * R5)
    a5b = st.min_blob_separation(shell5) / 2
    loc5, quat5, _ = st.roller_monolayer(nb5, radius=R5, seed=5)
    ri = RigidIntegrator([shell5] * nb5, loc5, quat5, "stochastic_Slip_Trapz", a5b, eta5, tolerance=1e-4, device=device,
                         ctx=ReplicatedContext(sm), seed=1)
    ri.kT, ri.g = 0.0040749841, 0.0303 / 12
    ri.repulsion_strength_wall = ri.repulsion_strength = 0.0326
    ri.debye_length_wall = ri.debye_length = 0.0406
    FT5 = torch.zeros((nb5, 6), dtype=torch.float64, device=device)
    FT5[:, 4] = 8 * math.pi * eta5 * R5 ** 3 * 62.8
    ri.external_force_torque = lambda it: FT5
    ri.report_rejections = False
    ri.advance_time_step(0.01, step=0)          # warm-up
    fence()
    d0, l0, m0, p0 = ri.det_iterations_count, ri.stoch_iterations_count, ri.susp.matvec_count, ri.susp.sweep_count
    t0 = time.perf_counter()
    ri.advance_time_step(0.01, step=1)
    fence()
    dt5 = sync_max(time.perf_counter() - t0)
    out = {"bodies": nb5, "blobs": ri.Nblobs, "scheme": ri.scheme, "solver_tolerance": 1e-4,
           "steps": 1, "s_per_step": round(dt5, 4),
           "gmres_iterations_per_step": ri.det_iterations_count - d0,
           "lanczos_iterations_per_step": ri.stoch_iterations_count - l0,
           "mobility_products_per_step": ri.susp.matvec_count - m0,
           "passes_over_the_pairs_per_step": ri.susp.sweep_count - p0,
           "rejected_steps": ri.invalid_configuration_count}
    ri.precision = "single"       # single-vector M_tt passes in fp32 (the k-vector lockstep passes stay fp64)
    d1, l1 = ri.det_iterations_count, ri.stoch_iterations_count
    fence()
    t0 = time.perf_counter()
    ri.advance_time_step(0.01, step=2)
    fence()
    dt5s = sync_max(time.perf_counter() - t0)
    ri.precision = "double"
    out["single_precision_option"] = {
        "s_per_step": round(dt5s, 4), "gmres_iterations_per_step": ri.det_iterations_count - d1,
        "lanczos_iterations_per_step": ri.stoch_iterations_count - l1, "rejected_steps": ri.invalid_configuration_count}
    return out
  if not args.no_sweep:
    stage("config5_multiblob_brownian", 60, config5_multiblob)

  if not args.no_host_surface and rank == 0 and world == 1:
    # Last of all: first contact of the one-process engine with real peer devices happens in a child process, after
    # every other number exists; the line as it stands goes to stderr and to the headline file first.
    guard.publish_headline(line, full=True)
    stage("multi_device_surface", 160, multi_device_surface, single_rank_only=True)
    stage("rccl_one_rank", 120, rccl_one_rank, single_rank_only=True)
  line["wall_s"] = round(guard.elapsed(), 1)
  guard.finish()
  if rank == 0:
    print(json.dumps(line), flush=True)
  if world > 1:
    dist.destroy_process_group()


def build_record():
  """What __graft_entry__.build() did for the library this run loads: compiled or reused (librmb_mobility.build.json)."""
  try:
    with open(os.path.join(ROOT, "rigidmultiblobswall_amd", "librmb_mobility.build.json")) as fh:
      return json.load(fh)
  except (OSError, ValueError):
    return {"mode": "unknown (no build record next to the library)"}


def main():
  args = parse_args()
  if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    sys.exit(spawn_ranks(args))
  rank_main(args)


if __name__ == "__main__":
  main()
