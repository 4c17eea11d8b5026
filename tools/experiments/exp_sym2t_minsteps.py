"""Launch shape of the two-targets-per-lane kernel: floor on steps per wave (option sym_min_steps; a step is two pairs) at
1e4 ... 32 000 blobs."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud
ctx = MobilityContext(0); ctx.set_option("timing", 1)
default = ctx.get_option("sym_min_steps")
for N in [int(x) for x in sys.argv[1:]] or [8000, 10000, 12000, 16000, 24576, 32000]:
  r, f, eta, a = d2_cloud(N)
  rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
  out = torch.empty(3 * N, dtype=torch.float64, device="cuda")
  ctx.set_positions(rd, a, None, True)
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < 0.2:
    for _ in range(20): ctx.matvec_device("tt", fd, eta, out=out)
    torch.cuda.synchronize()
  cells = []
  for ms in (16, 24, 32, 48, 64, 96, 128):
    ctx.set_option("sym_min_steps", ms)
    best = []
    for rnd in range(3):
      for _ in range(5): ctx.matvec_device("tt", fd, eta, out=out)
      torch.cuda.synchronize(); ctx.timing_reset()
      for _ in range(100): ctx.matvec_device("tt", fd, eta, out=out)
      torch.cuda.synchronize()
      best.append(float(np.median(ctx.timing_collect(100))) * 1e3)
    cells.append("%3d: %8.2f us (%4d wgs)" % (ms, np.median(best), ctx.last_launch()["workgroups"]))
  ctx.set_option("sym_min_steps", default)
  print("N=%6d (default %d, path %d)  %s" % (N, default, ctx.get_option("last_path"), "  ".join(cells)), flush=True)
ctx.close()
