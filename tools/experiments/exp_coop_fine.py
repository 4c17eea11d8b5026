"""Steps per wave of the workgroup-cooperative kernel below one resident round (option sym_fine_steps; the cooperative
default is 8): small suspensions, HIP-event kernel time."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud
ctx = MobilityContext(0); ctx.set_option("timing", 1)
first = True
for N in (200, 500, 1000, 1500, 2000, 3000, 4000):
  r, f, eta, a = d2_cloud(N)
  rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
  out = torch.empty(3 * N, dtype=torch.float64, device="cuda")
  ctx.set_positions(rd, a, None, True)
  if first:
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
      for _ in range(50): ctx.matvec_device("tt", fd, eta, out=out)
      torch.cuda.synchronize()
    first = False
  cells = []
  for fine in (2, 4, 8, 16):
    ctx.set_option("sym_fine_steps", fine)
    for _ in range(20): ctx.matvec_device("tt", fd, eta, out=out)
    torch.cuda.synchronize(); ctx.timing_reset()
    for _ in range(400): ctx.matvec_device("tt", fd, eta, out=out)
    torch.cuda.synchronize()
    cells.append("fine %2d: %6.2f us (%4d wgs)" % (fine, float(np.mean(ctx.timing_collect(400))) * 1e3, ctx.last_launch()["workgroups"]))
  print("N=%5d  %s" % (N, "   ".join(cells)), flush=True)
ctx.close()
