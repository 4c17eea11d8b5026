"""Price of the two bit-reproducible modes against the default (atomic flushes): wall tt, clocks primed."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud
for N in (10000, 24576, 100000, 262144):
  r, f, eta, a = d2_cloud(N)
  rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
  out = torch.empty_like(fd)
  ctx = MobilityContext(0)
  ctx.set_positions(rd, a, wall=True)
  for _ in range(600 if N <= 30000 else 3):
    ctx.matvec_device("tt", fd, eta, out=out)
  torch.cuda.synchronize()
  reps = 200 if N <= 30000 else (10 if N <= 100000 else 3)
  line = "N=%d:" % N
  for det in (0, 2, 1):
    ctx.set_option("deterministic", det)
    for _ in range(3):
      ctx.matvec_device("tt", fd, eta, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
      ctx.matvec_device("tt", fd, eta, out=out)
    e1.record(); torch.cuda.synchronize()
    line += "  deterministic=%d %.4f ms/product" % (det, e0.elapsed_time(e1) / reps)
  print(line, flush=True)
  ctx.close()
