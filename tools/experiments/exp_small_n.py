"""Small suspensions (one launch is less than a resident round): symmetric kernel with the sub-round floor on steps per
wave at 16 / 32 / 64 against the one-sided sweep; sweep + finalize end to end per call, no events, clocks primed."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud
ctx = MobilityContext(0)
r, f, eta, a = d2_cloud(10000)
ctx.set_positions(r, a, None, wall=True)
fd = torch.as_tensor(f.reshape(-1), device="cuda")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
  for _ in range(20): ctx.matvec_device("tt", fd, eta)
  torch.cuda.synchronize()
print("N      one-sided   sym fine=16   fine=24   fine=32   fine=48   fine=64   default (0)   (us per call, workgroups)")
for N in (128, 192, 256, 384, 512, 768, 1000, 1500, 2000, 3000, 4000, 5000, 6000, 8000):
  r, f, eta, a = d2_cloud(N)
  rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
  out = torch.empty(3 * N, dtype=torch.float64, device="cuda")
  ctx.set_positions(rd, a, None, wall=True)
  cells = []
  for mode in ("sweep", 16, 24, 32, 48, 64, 0):
    ctx.set_option("symmetric", 0 if mode == "sweep" else 1)
    if mode != "sweep": ctx.set_option("sym_fine_steps", mode)
    for _ in range(20): ctx.matvec_device("tt", fd, eta, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(400): ctx.matvec_device("tt", fd, eta, out=out)
    torch.cuda.synchronize()
    cells.append("%6.1f (%4d)" % ((time.perf_counter() - t0) / 400 * 1e6, ctx.last_launch()["workgroups"]))
  ctx.set_option("symmetric", 1); ctx.set_option("sym_fine_steps", 0)
  print("%-6d %s" % (N, "  ".join(cells)), flush=True)
ctx.close()
