import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud
N = 100000
r, f, eta, a = d2_cloud(N)
rd = torch.as_tensor(r.reshape(-1), device="cuda")
ctx = MobilityContext(0); ctx.set_option("timing", 1)
ctx.set_positions(rd, a, wall=False)
for prec in (64, 32, 64, 32):
  ctx.set_option("precision", prec)
  for _ in range(3): ctx.blob_blob_force_device(3.92, 0.1 * a, a)
  torch.cuda.synchronize(); ctx.timing_reset()
  for _ in range(6): ctx.blob_blob_force_device(3.92, 0.1 * a, a)
  torch.cuda.synchronize()
  print(prec, float(np.mean(ctx.timing_collect(6))), "ms")
