"""Where the two-targets-per-lane kernel should take over from the cooperative one: wall tt kernel time for
sym_two_targets = 0 (cooperative / per wave), 1 (the default rule: from one resident round on) and 2 (always)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud
ctx = MobilityContext(0); ctx.set_option("timing", 1)
r, f, eta, a = d2_cloud(10000)
ctx.set_positions(torch.as_tensor(r.reshape(-1), device="cuda"), a, None, True)
fd = torch.as_tensor(f.reshape(-1), device="cuda")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
  for _ in range(20): ctx.matvec_device("tt", fd, eta)
  torch.cuda.synchronize()
for N in [int(x) for x in sys.argv[1:]] or [1000, 2000, 3000, 4000, 5000, 6000, 7000, 8000, 9000, 10000, 12000, 16000]:
  r, f, eta, a = d2_cloud(N)
  rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
  out = torch.empty(3 * N, dtype=torch.float64, device="cuda")
  ctx.set_positions(rd, a, None, True)
  cells = []
  for mode in (0, 1, 2):
    ctx.set_option("sym_two_targets", mode)
    best = []
    for rnd in range(3):
      for _ in range(5): ctx.matvec_device("tt", fd, eta, out=out)
      torch.cuda.synchronize(); ctx.timing_reset()
      for _ in range(200): ctx.matvec_device("tt", fd, eta, out=out)
      torch.cuda.synchronize()
      best.append(float(np.median(ctx.timing_collect(200))) * 1e3)
    cells.append("mode %d: %7.2f us (path %d)" % (mode, np.median(best), ctx.get_option("last_path")))
  print("N=%6d  %s" % (N, "   ".join(cells)), flush=True)
ctx.close()
