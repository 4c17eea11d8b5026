import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud
for N in (10000, 24576):
  r, f, eta, a = d2_cloud(N)
  rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
  ctxs = {}
  for ms_ in (64, 48, 32, 24, 16):
    ctx = MobilityContext(0); ctx.set_option("timing", 1); ctx.set_option("sym_min_steps", ms_); ctx.set_option("sym_oversub", 16)
    ctx.set_positions(rd, a, wall=True)
    for _ in range(3):
      ctx.matvec_device("tt", fd, eta)
    ctxs[ms_] = ctx
  torch.cuda.synchronize()
  res = {k: [] for k in ctxs}
  for rnd in range(5):                       # interleaved rounds in one process
    for k, ctx in ctxs.items():
      ctx.timing_reset()
      for _ in range(10):
        ctx.matvec_device("tt", fd, eta)
      torch.cuda.synchronize()
      res[k].append(ctx.timing_collect(10).mean())
  for k, ctx in ctxs.items():
    print("N=%d min_steps=%d wgs=%d kernel median %.4f min %.4f ms" % (N, k, ctx.last_launch()["workgroups"], np.median(res[k]), np.min(res[k])), flush=True)
    ctx.close()
