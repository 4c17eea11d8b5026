#!/usr/bin/env python
"""Headline benchmark: RPY-wall M.f matvecs/s on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one single_wall_mobility_trans_times_force product (BASELINE.json configs[1]: 1e4
random blobs above a wall, fp64) with positions and the force vector already resident in HBM:
all-gather of the force blocks (N > 1 only) + pair sweep + chunk reduction.  Targets are sharded
over the ranks; total work is fixed => "strong" scaling.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      dominant kernel (the pair sweep): algorithmic flops 211 N^2 per launch
                / average launch duration from HIP events on the launch stream, against the fp64
                vector peak.  The path is FP64-VALU bound, not HBM bound; the HBM view
                (algorithmic 72 N bytes per launch, PMC traffic) is reported beside it.
  cpu_baseline  the CPU oracle's -O3 -ffast-math OpenMP build timed on this host (rank 0, N=1)
  sweep         the same product at larger N_blobs (the metric is "... vs N_blobs")
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOPS_PER_PAIR = {"tt_wall": 211.0}     # reference as-written op count, SURVEY.md 8(d)
FP64_VECTOR_PEAK_TFLOPS = 78.6          # MI355X fp64 vector = 1/2 of the 157.3 TF fp32 vector peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


def d2_cloud(N, seed=0):
  """5 % volume fraction, a = 0.5, eta = 1, z in [1.1a, 1.1a + Lbox)."""
  rng = np.random.RandomState(seed)
  a, eta = 0.5, 1.0
  Lbox = (N * (4.0 / 3.0) * np.pi * a ** 3 / 0.05) ** (1.0 / 3.0)
  r = rng.rand(N, 3) * Lbox
  r[:, 2] += 1.1 * a
  return r, rng.randn(N, 3), eta, a


def run_config(torch, dist, sm, backend, n_blobs, steps, warmup, world, rank, device):
  from rigidmultiblobswall_amd.distributed import partition
  r, f, eta, a = d2_cloud(n_blobs, seed=0)
  b, e, _ = partition(n_blobs, world, rank)
  sm.set_local_positions(torch.as_tensor(r[b:e].reshape(-1), device=device), n_blobs, a, wall=True)
  if world == 1:
    f_local = torch.as_tensor(f[b:e].reshape(-1), device=device)
    out = torch.empty(3 * (e - b), dtype=torch.float64, device=device)

    def step():
      sm.matvec_local("tt", f_local, eta, out=out)
  else:
    # replicated vectors: every rank holds f and receives u; unordered pairs are sharded over the
    # ranks (each pair evaluated once, applied to both blobs) and u is all-reduced (RCCL)
    f_full = torch.as_tensor(f.reshape(-1), device=device)
    out = torch.empty(3 * n_blobs, dtype=torch.float64, device=device)

    def step():
      sm.matvec_replicated("tt", f_full, eta, out=out)

  for _ in range(warmup):
    step()
  torch.cuda.synchronize(device)
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize(device)
  backend.ctx.timing_reset()
  t0 = time.perf_counter()
  for _ in range(steps):
    step()
  torch.cuda.synchronize(device)
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize(device)
  dt = time.perf_counter() - t0
  if world > 1:
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
  kern_ms = backend.ctx.timing_collect(steps)
  kern_ms_avg = float(np.mean(kern_ms)) if len(kern_ms) else float("nan")
  if world > 1:
    t = torch.tensor([kern_ms_avg], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    kern_ms_avg = float(t.item())
  return dict(dt=dt, kern_ms=kern_ms_avg, launch=backend.ctx.last_launch(), out=out, r=r, f=f, eta=eta, a=a,
              n_local=e - b, begin=b, end=e)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=200)
  ap.add_argument("--warmup", type=int, default=20)
  ap.add_argument("--blobs", type=int, default=10000, help="N_blobs of the headline workload (configs[1] = 1e4)")
  ap.add_argument("--no-sweep", action="store_true")
  ap.add_argument("--no-cpu", action="store_true")
  ap.add_argument("--traffic-bytes", type=float, default=None,
                  help="HBM bytes per sweep launch from a separate rocprofv3 --pmc run (profiles/), if known")
  args = ap.parse_args()

  import torch
  import torch.distributed as dist
  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
  # rehearsal on a 1-GPU box: RMB_BENCH_BACKEND=gloo lets several ranks share cuda:0 (RCCL refuses that)
  backend_name = os.environ.get("RMB_BENCH_BACKEND", "nccl")
  device = torch.device("cuda:%d" % (local_rank % torch.cuda.device_count()))
  torch.cuda.set_device(device)
  if world > 1:
    if backend_name == "nccl":
      dist.init_process_group("nccl", device_id=device)
    else:
      dist.init_process_group(backend_name)
  assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"

  from rigidmultiblobswall_amd.distributed import HipBackend, ShardedMobility
  backend = HipBackend(device)
  backend.ctx.set_option("timing", 1)
  sm = ShardedMobility(backend, device=device)

  N = args.blobs
  res = run_config(torch, dist, sm, backend, N, args.steps, args.warmup, world, rank, device)
  ms_per_step = 1e3 * res["dt"] / args.steps
  value = args.steps / res["dt"]

  # roofline of the dominant kernel: one rank's launch covers 1/world of the N x N ordered pairs
  pairs_per_launch = float(N) * N / world
  flops = FLOPS_PER_PAIR["tt_wall"] * pairs_per_launch
  achieved_tf = flops / (res["kern_ms"] * 1e-3) / 1e12
  alg_bytes = 48.0 * N + 24.0 * res["n_local"]       # read r,f of all sources; write u of own targets
  traffic = args.traffic_bytes
  valu_instr = issue_ceiling = None
  if traffic is None:
    # HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
    try:
      with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
        tj = json.load(fh)
      key = "sym_tt_wall_N%d" % N if res["launch"]["chunks"] == 0 else "sweep_tt_wall_N%d" % N
      cand = [v for k, v in tj.items() if k.endswith(key)]
      if cand and world == 1:
        traffic = cand[-1]["traffic_bytes"]
        valu_instr = cand[-1].get("SQ_INSTS_VALU_per_launch")
        issue_ceiling = tj.get("_fp64_issue_ceiling_G_wave_instr_per_s")
    except (OSError, ValueError, KeyError):
      traffic = None
  sym = res["launch"]["chunks"] == 0
  roofline = {
      "bound": "valu_fp64",
      "kernel": "rmb::sym_kernel<TT,wall> (each unordered pair once)" if sym else "rmb::sweep_kernel<TT,wall>",
      "note": "achieved = ALGORITHMIC flops (211 per ordered pair, the reference's as-written count, SURVEY 8d) / measured "
              "kernel time; the kernel executes ~55 (symmetric) or ~93 (sweep) fp64 VALU instructions per ordered pair, so "
              "frac can exceed 1; measured fp64 issue ceiling of the chip: 479 G wave-instr/s (profiles/r1_ubench_*)",
      "achieved": round(achieved_tf, 3), "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
      "frac": round(achieved_tf / FP64_VECTOR_PEAK_TFLOPS, 4),
      "flops_per_pair": FLOPS_PER_PAIR["tt_wall"], "pairs_per_launch": pairs_per_launch,
      "kernel_ms_avg": round(res["kern_ms"], 5), "launch": res["launch"],
      "traffic": traffic,
      # executed-instruction view (what the kernel actually issues vs what the chip can issue): SQ_INSTS_VALU per
      # launch from the committed rocprofv3 --pmc pass of this command / live kernel time, against the fp64 VALU
      # issue ceiling measured by tools/ubench.hip on the same chip family
      "issue": None if not valu_instr else {
          "valu_wave_instr_per_launch": valu_instr, "achieved": round(valu_instr / (res["kern_ms"] * 1e-3) / 1e9, 1),
          "peak": issue_ceiling, "unit": "G wave-instr/s", "frac": round(valu_instr / (res["kern_ms"] * 1e-3) / 1e9 / issue_ceiling, 4)},
      "hbm": {"algorithmic_bytes_per_launch": alg_bytes,
              "achieved": round(alg_bytes / (res["kern_ms"] * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
              "frac": round(alg_bytes / (res["kern_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)},
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
      "roofline": roofline,
  }

  if rank == 0 and world == 1 and not args.no_cpu:
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
    t_start = time.perf_counter()
    while len(times) < 40 and (time.perf_counter() - t_start < 12.0 or len(times) < 3):
      t0 = time.perf_counter()
      oracle.single_wall_mobility_trans_times_force_oracle(r, f, eta, a, fast=True)
      times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    line["cpu_baseline"] = {"value": round(1.0 / med, 4), "unit": "matvecs/s", "cores": cores,
                            "kind": "port",
                            "sample": "%d full matvecs of the same %d-blob workload (median), oracle C port "
                                      "-O3 -ffast-math -fopenmp, %d OpenMP threads (host exposes %d hardware threads)"
                                      % (len(times), N, cores, hw_threads)}

  if not args.no_sweep:
    try:
      sweep = []
      for nb, st, wu in ((100000, 5, 1), (262144, 3, 1), (1000000, 2, 1)):
        rs = run_config(torch, dist, sm, backend, nb, st, wu, world, rank, device)
        pairs = float(nb) * nb / world
        sweep.append({"n_blobs": nb, "matvecs_per_s": round(st / rs["dt"], 4), "ms_per_step": round(1e3 * rs["dt"] / st, 3),
                      "kernel_ms_avg": round(rs["kern_ms"], 3),
                      "valu_fp64_tflops": round(211.0 * pairs / (rs["kern_ms"] * 1e-3) / 1e12, 2),
                      "hbm_algorithmic_gbps": round(72.0 * nb / (rs["kern_ms"] * 1e-3) / 1e9, 4),
                      "launch": rs["launch"]})
      line["sweep"] = sweep
    except Exception as exc:      # an extra must never cost the headline line
      line['sweep'] = {"error": "%s: %s" % (type(exc).__name__, exc)}

  if world == 1 and not args.no_sweep:
    try:
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
      rs.solve_mobility_problem(force_torque=FT, tol=1e-8)        # warm-up (library initialisation)
      torch.cuda.synchronize(device)
      t0 = time.perf_counter()
      U, lam, info = rs.solve_mobility_problem(force_torque=FT, tol=1e-8)
      torch.cuda.synchronize(device)
      line["config3_gmres"] = {"bodies": nb, "blobs": rs.n_blobs, "tolerance": 1e-8, "iterations": info["iterations"],
                               "residual": float(info["residual"]), "ms_per_solve": round(1e3 * (time.perf_counter() - t0), 3)}
      rs.close()
    except Exception as exc:      # an extra must never cost the headline line
      line['config3_gmres'] = {"error": "%s: %s" % (type(exc).__name__, exc)}

  if not args.no_sweep:
    try:
      # BASELINE.json configs[4] recipe: 2.6e5 single-blob rollers, Brownian Adams-Bashforth steps = forces kernel +
      # M_tt F + M_tr T + Lanczos M^{1/2} z + 2 random-finite-difference products per step; physical parameters of
      # multi_bodies/examples/rollers/inputfile_rollers.dat.  On N ranks the same replicated stepper runs on every
      # rank and only the pair sweeps are divided (ReplicatedContext).  Reported beside the headline.
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
      integ.advance_time_step(0.016)      # warm-up (first step is forward Euler)
      torch.cuda.synchronize(device)
      if world > 1:
        dist.barrier()
      p0, l0, n5_steps = integ.mobility_products, integ.stoch_iterations_count, 2
      t0 = time.perf_counter()
      for _ in range(n5_steps):
        integ.advance_time_step(0.016)
      torch.cuda.synchronize(device)
      if world > 1:
        dist.barrier()
      dt5 = time.perf_counter() - t0
      if world > 1:
        t = torch.tensor([dt5], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt5 = float(t.item())
      line["config5_rollers"] = {"rollers": n5, "scheme": integ.scheme, "lanczos_tolerance": 1e-3, "steps": n5_steps,
                                 "s_per_step": round(dt5 / n5_steps, 4),
                                 "mobility_products_per_step": (integ.mobility_products - p0) / n5_steps,
                                 "lanczos_iterations_per_step": (integ.stoch_iterations_count - l0) / n5_steps,
                                 "rejected_steps": integ.invalid_configuration_count}
    except Exception as exc:      # an extra must never cost the headline line
      line['config5_rollers'] = {"error": "%s: %s" % (type(exc).__name__, exc)}

  if not args.no_sweep:
    try:
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
      ri.advance_time_step(0.01, step=0)          # warm-up
      torch.cuda.synchronize(device)
      if world > 1:
        dist.barrier()
      d0, l0, m0 = ri.det_iterations_count, ri.stoch_iterations_count, ri.susp.matvec_count
      t0 = time.perf_counter()
      ri.advance_time_step(0.01, step=1)
      torch.cuda.synchronize(device)
      if world > 1:
        dist.barrier()
      dt5 = time.perf_counter() - t0
      if world > 1:
        t = torch.tensor([dt5], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt5 = float(t.item())
      line["config5_multiblob_brownian"] = {"bodies": nb5, "blobs": ri.Nblobs, "scheme": ri.scheme, "solver_tolerance": 1e-4,
                                            "steps": 1, "s_per_step": round(dt5, 4),
                                            "gmres_iterations_per_step": ri.det_iterations_count - d0,
                                            "lanczos_iterations_per_step": ri.stoch_iterations_count - l0,
                                            "pair_sweeps_per_step": ri.susp.matvec_count - m0,
                                            "rejected_steps": ri.invalid_configuration_count}
    except Exception as exc:      # an extra must never cost the headline line
      line['config5_multiblob_brownian'] = {"error": "%s: %s" % (type(exc).__name__, exc)}

  if rank == 0:
    print(json.dumps(line), flush=True)
  if world > 1:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
