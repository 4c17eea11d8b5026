"""Cost of the two-vector symmetric tt sweep against two single sweeps (kernel time, HIP events)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud
for N in (10000, 24576, 100000, 262144):
  r, f, eta, a = d2_cloud(N)
  rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
  gd = torch.randn_like(fd)
  for wall in (True, False):
    ctx = MobilityContext(0); ctx.set_option("timing", 1)
    ctx.set_positions(rd, a, None, wall=wall)
    reps = 20 if N <= 24576 else 4
    for _ in range(2): ctx.matvec_device("tt", fd, eta)
    torch.cuda.synchronize(); ctx.timing_reset()
    for _ in range(reps): ctx.matvec_device("tt", fd, eta)
    torch.cuda.synchronize(); one = float(np.mean(ctx.timing_collect(reps)))
    oa, ob = torch.empty_like(fd), torch.empty_like(fd)
    for _ in range(2): ctx.matvec2_device("tt", fd, gd, eta, out_a=oa, out_b=ob)
    torch.cuda.synchronize(); ctx.timing_reset()
    for _ in range(reps): ctx.matvec2_device("tt", fd, gd, eta, out_a=oa, out_b=ob)
    torch.cuda.synchronize(); two = float(np.mean(ctx.timing_collect(reps)))
    print("N=%d wall=%s: one vector %.3f ms, two vectors in one pass %.3f ms = %.2f x one = %.2f of two passes" %
          (N, wall, one, two, two / one, two / (2 * one)), flush=True)
    ctx.close()
