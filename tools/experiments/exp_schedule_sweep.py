"""Schedule knobs of the symmetric kernel with primed clocks: rounds of resident workgroups (sym_oversub), minimum
rotation steps per wave (sym_min_steps), resident workgroups per CU (sym_wps).
  python tools/experiments/exp_schedule_sweep.py [N]            headline size: the full grid
  python tools/experiments/exp_schedule_sweep.py N coarse       other sizes: oversub x min_steps at the default residency"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
r, f, eta, a = d2_cloud(N)
rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
out = torch.empty_like(fd)
ctx = MobilityContext(0); ctx.set_option("timing", 1)
ctx.set_positions(rd, a, wall=True)
coarse = len(sys.argv) > 2 and sys.argv[2] == "coarse"
scale = max(1, int((N / 10000.0) ** 2))
for _ in range(max(3, 1500 // scale)):
  ctx.matvec_device("tt", fd, eta, out=out)
torch.cuda.synchronize()
res = []
n_rep = max(3, 100 // scale)
for rep in range(2):
  for wps in ((0,) if coarse else (0, 3, 2)):
    for over in ((1, 2, 3, 4, 6, 8, 12, 16) if coarse else (1, 2, 3, 4, 8)):
      for ms_ in ((64, 128, 256, 512) if coarse else (32, 48, 64, 96, 128, 194)):
        ctx.set_option("sym_wps", wps); ctx.set_option("sym_oversub", over); ctx.set_option("sym_min_steps", ms_)
        for _ in range(max(2, 20 // scale)):
          ctx.matvec_device("tt", fd, eta, out=out)
        torch.cuda.synchronize(); ctx.timing_reset()
        for _ in range(n_rep):
          ctx.matvec_device("tt", fd, eta, out=out)
        torch.cuda.synchronize()
        t = ctx.timing_collect(n_rep)
        res.append((float(np.mean(t)), wps, over, ms_, ctx.last_launch()["workgroups"]))
res.sort()
for t, wps, over, ms_, wg in res[:25]:
  print("%.4f ms  wps=%d oversub=%d min_steps=%d workgroups=%d" % (t, wps, over, ms_, wg))
print("default (wps=0 oversub=8 min_steps=64):", [x for x in res if x[1:4] == (0, 8, 64)])
ctx.close()
