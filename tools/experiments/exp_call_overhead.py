"""Host cost of one device-entry product of a small suspension (resident vectors, back-to-back, no synchronisation in the loop)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud
ctx = MobilityContext(0)
for N in (128, 512, 1000, 2000):
  r, f, eta, a = d2_cloud(N)
  rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
  out = torch.empty(3 * N, dtype=torch.float64, device="cuda")
  ctx.set_positions(rd, a, None, wall=True)
  for _ in range(200): ctx.matvec_device("tt", fd, eta, out=out)
  torch.cuda.synchronize()
  best = 1e9
  for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(2000): ctx.matvec_device("tt", fd, eta, out=out)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 2000 * 1e6)
  print("N=%5d: %.2f us per product (enqueue alone %.2f us)" % (N, best, t_enq / 2000 * 1e6), flush=True)
ctx.close()
