"""A few launches of the single-precision tt product (for rocprofv3 --pmc passes)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
r, f, eta, a = d2_cloud(N)
rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
ctx = MobilityContext(0); ctx.set_option("timing", 1)
ctx.set_positions(rd, a, wall=True)
for prec in (64, 32):
  ctx.set_option("precision", prec)
  for _ in range(30): ctx.matvec_device("tt", fd, eta)
  torch.cuda.synchronize(); ctx.timing_reset()
  for _ in range(20): ctx.matvec_device("tt", fd, eta)
  torch.cuda.synchronize()
  print(prec, float(np.mean(ctx.timing_collect(20))), "ms", ctx.last_launch())
ctx.close()
