"""Launch plan of a pair shard that is smaller than one resident round: residency cap ("sym_wps") x floor on steps per
wave ("sym_fine_steps").  HIP events around every sweep, clocks primed."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
r, f, eta, a = d2_cloud(N)
rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
out = torch.empty(3 * N, dtype=torch.float64, device="cuda")
ctx = MobilityContext(0)
ctx.set_positions(rd, a, None, wall=True)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
  for _ in range(20):
    ctx.matvec_device("tt", fd, eta, out=out)
  torch.cuda.synchronize()
ctx.set_option("timing", 1)
for G in [int(x) for x in sys.argv[2:]] or [2, 4, 8]:
  print("N=%d G=%d   rows: sym_wps, columns: sym_fine_steps 8 16 24 32 48 64 96  -> us (workgroups)" % (N, G))
  for wps in (0, 4, 3, 2, 1):
    ctx.set_option("sym_wps", wps)
    cells = []
    for fine in (8, 16, 24, 32, 48, 64, 96):
      ctx.set_option("sym_fine_steps", fine)
      for _ in range(5): ctx.matvec_pairshard_device("tt", fd, eta, G // 2, G, out=out)
      torch.cuda.synchronize(); ctx.timing_reset()
      for _ in range(200): ctx.matvec_pairshard_device("tt", fd, eta, G // 2, G, out=out)
      torch.cuda.synchronize()
      cells.append("%6.1f (%4d)" % (float(np.mean(ctx.timing_collect(200))) * 1e3, ctx.last_launch()["workgroups"]))
    print("  wps %d: %s" % (wps, "  ".join(cells)), flush=True)
ctx.close()
