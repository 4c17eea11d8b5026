"""Schedule knobs of the symmetric kernel at the headline size with primed clocks: rounds of resident workgroups
(sym_oversub), minimum rotation steps per wave (sym_min_steps), resident workgroups per CU (sym_wps)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
r, f, eta, a = d2_cloud(N)
rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
out = torch.empty_like(fd)
ctx = MobilityContext(0); ctx.set_option("timing", 1)
ctx.set_positions(rd, a, wall=True)
for _ in range(1500):
  ctx.matvec_device("tt", fd, eta, out=out)
torch.cuda.synchronize()
res = []
for rep in range(2):
  for wps in (0, 3, 2):
    for over in (1, 2, 3, 4, 8):
      for ms_ in (32, 48, 64, 96, 128, 194):
        ctx.set_option("sym_wps", wps); ctx.set_option("sym_oversub", over); ctx.set_option("sym_min_steps", ms_)
        for _ in range(20):
          ctx.matvec_device("tt", fd, eta, out=out)
        torch.cuda.synchronize(); ctx.timing_reset()
        for _ in range(100):
          ctx.matvec_device("tt", fd, eta, out=out)
        torch.cuda.synchronize()
        t = ctx.timing_collect(100)
        res.append((float(np.mean(t)), wps, over, ms_, ctx.last_launch()["workgroups"]))
res.sort()
for t, wps, over, ms_, wg in res[:25]:
  print("%.4f ms  wps=%d oversub=%d min_steps=%d workgroups=%d" % (t, wps, over, ms_, wg))
print("default (wps=0 oversub=8 min_steps=64):", [x for x in res if x[1:4] == (0, 8, 64)])
ctx.close()
