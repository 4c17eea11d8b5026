"""sym2t_kernel at N blobs: kernel time (HIP events, clocks primed, alternating rounds) against the floor on rotation steps per
wave ("sym_min_steps": shorter waves = more rounds of workgroups = a shorter phase in which a SIMD is left with one or two
waves, paid with more staging / flush work per step)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud

sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [10000]
steps = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [64, 48, 32, 24, 16, 12, 8]
for N in sizes:
  r, f, eta, a = d2_cloud(N)
  rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
  out = torch.empty(3 * N, dtype=torch.float64, device="cuda")
  ctx = MobilityContext(0)
  ctx.set_positions(rd, a, None, wall=True)
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < 0.4:
    for _ in range(20):
      ctx.matvec_device("tt", fd, eta, out=out)
    torch.cuda.synchronize()
  ctx.set_option("timing", 1)
  res = {m: [] for m in steps}
  wgs = {}
  reps = 100 if N <= 30000 else 10
  for rnd in range(4):
    for m in (steps if rnd % 2 == 0 else steps[::-1]):
      ctx.set_option("sym_min_steps", m)
      for _ in range(5):
        ctx.matvec_device("tt", fd, eta, out=out)
      torch.cuda.synchronize(); ctx.timing_reset()
      for _ in range(reps):
        ctx.matvec_device("tt", fd, eta, out=out)
      torch.cuda.synchronize()
      res[m].append(float(np.median(ctx.timing_collect(reps))) * 1e3)
      wgs[m] = (ctx.last_launch()["workgroups"], ctx.get_option("last_path"))
  for m in steps:
    print("N %d sym_min_steps %3d: kernel %.2f us (rounds %s) workgroups %d path %d" % (N, m, np.median(res[m]), ["%.1f" % x for x in res[m]], wgs[m][0], wgs[m][1]), flush=True)
  ctx.close()
