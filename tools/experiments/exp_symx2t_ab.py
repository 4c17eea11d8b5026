"""Same-box A/B of the two-targets-per-lane instances of the generic symmetric skeleton (csrc/symx2t_kernels.h) against the
one-target kernels: option sym_two_targets 0 / 1, alternating rounds, HIP events around the sweep, clocks primed.

  python tools/experiments/exp_symx2t_ab.py [sizes] [periodic sizes]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud

sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [10000, 100000]
psizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [24576]
ctx = MobilityContext(0)
ctx.set_option("timing", 1)


def ab(label, call, reps):
  res = {0: [], 1: []}
  path = {}
  for rnd in range(4):
    for mode in ((0, 1) if rnd % 2 == 0 else (1, 0)):
      ctx.set_option("sym_two_targets", mode)
      for _ in range(3):
        call()
      torch.cuda.synchronize(); ctx.timing_reset()
      for _ in range(reps):
        call()
      torch.cuda.synchronize()
      res[mode].append(float(np.median(ctx.timing_collect(reps))) * 1e3)
      path[mode] = ctx.get_option("last_path")
  one, two = np.median(res[0]), np.median(res[1])
  print("%-44s one target %9.2f us (path %d)   two targets %9.2f us (path %d)   x %.3f" % (label, one, path[0], two, path[1], one / two), flush=True)


for N in sizes:
  r, f, eta, a = d2_cloud(N)
  t = np.random.RandomState(1).randn(N, 3)
  rd = torch.as_tensor(r.reshape(-1), device="cuda")
  vs = [torch.as_tensor(x.reshape(-1), device="cuda") for x in (f, t)]
  outs = [torch.empty(3 * N, dtype=torch.float64, device="cuda") for _ in range(2)]
  ctx.set_positions(rd, a, None, wall=True)
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < 0.4:
    ctx.matvec_device("tt", vs[0], eta, out=outs[0]); torch.cuda.synchronize()
  reps = 60 if N <= 30000 else 6
  ab("N %d wall fused row (f, tau -> u)" % N, lambda: ctx.matvec_op_device("velocity_from_force_torque", vs, eta, outs=outs[:1]), reps)
  ab("N %d wall grand (f, tau -> u, w)" % N, lambda: ctx.matvec_op_device("grand", vs, eta, outs=outs), reps)
  ab("N %d wall force column (f -> u, w)" % N, lambda: ctx.matvec_op_device("force_column", vs[:1], eta, outs=outs), reps)
  ab("N %d wall tt x 2 vectors" % N, lambda: ctx.matvec_op_device("tt_multi", vs, eta, outs=outs), reps)
  ab("N %d wall rr x 2 vectors" % N, lambda: ctx.matvec_op_device("rr_multi", vs, eta, outs=outs), reps)
  ctx.set_positions(rd, a, None, wall=False)
  ab("N %d no wall grand" % N, lambda: ctx.matvec_op_device("grand", vs, eta, outs=outs), reps)
for N in psizes:
  r, f, eta, a = d2_cloud(N)
  t = np.random.RandomState(1).randn(N, 3)
  box = (N * (4.0 / 3.0) * np.pi * a ** 3 / 0.05) ** (1.0 / 3.0)
  rd = torch.as_tensor(r.reshape(-1), device="cuda")
  vs = [torch.as_tensor(x.reshape(-1), device="cuda") for x in (f, t)]
  outs = [torch.empty(3 * N, dtype=torch.float64, device="cuda") for _ in range(2)]
  reps = 10
  for L, name in ((np.array([box, box, 0.0]), "xy"), (np.array([box, 0.0, 0.0]), "x")):
    ctx.set_positions(rd, a, L, wall=True)
    ab("N %d periodic %s wall tt" % (N, name), lambda: ctx.matvec_device("tt", vs[0], eta, out=outs[0]), reps)
    ab("N %d periodic %s wall rr" % (N, name), lambda: ctx.matvec_device("rr", vs[0], eta, out=outs[0]), reps)
    ab("N %d periodic %s wall fused row" % (N, name), lambda: ctx.matvec_op_device("velocity_from_force_torque", vs, eta, outs=outs[:1]), reps)
    ab("N %d periodic %s wall grand" % (N, name), lambda: ctx.matvec_op_device("grand", vs, eta, outs=outs), reps)
ctx.close()
