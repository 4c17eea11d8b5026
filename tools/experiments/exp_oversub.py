import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud
for N in (10000, 24576, 100000):
  r, f, eta, a = d2_cloud(N)
  rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
  for k in (1, 2, 3, 4, 6, 8):
    ctx = MobilityContext(0); ctx.set_option("timing", 1); ctx.set_option("sym_oversub", k)
    ctx.set_positions(rd, a, wall=True)
    reps = 30 if N < 50000 else 8
    for _ in range(3):
      ctx.matvec_device("tt", fd, eta)
    torch.cuda.synchronize(); ctx.timing_reset()
    for _ in range(reps):
      ctx.matvec_device("tt", fd, eta)
    torch.cuda.synchronize()
    ms = ctx.timing_collect(reps)
    print("N=%d oversub=%d wgs=%d kernel avg %.4f min %.4f ms" % (N, k, ctx.last_launch()["workgroups"], ms.mean(), ms.min()), flush=True)
    ctx.close()
