"""Per-rank kernel time of the pair-sharded symmetric product at N = 1e4 (what one of G ranks executes), for
several values of the sym_min_steps option."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
r, f, eta, a = d2_cloud(N)
rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
for min_steps in (64, 32, 16):
  for G in (1, 2, 4, 8):
    ctx = MobilityContext(0); ctx.set_option("timing", 1); ctx.set_option("sym_min_steps", min_steps)
    ctx.set_positions(rd, a, None, wall=True)
    out = torch.empty(3 * N, dtype=torch.float64, device="cuda")
    for _ in range(3): ctx.matvec_pairshard_device("tt", fd, eta, 0, G, out=out)
    torch.cuda.synchronize(); ctx.timing_reset()
    for _ in range(50): ctx.matvec_pairshard_device("tt", fd, eta, G // 2, G, out=out)
    torch.cuda.synchronize()
    ms = float(np.mean(ctx.timing_collect(50)))
    print("min_steps=%d G=%d: shard kernel %.1f us (ideal %.1f us), workgroups %d" %
          (min_steps, G, ms * 1e3, 195.0 / G * (N / 1e4) ** 2, ctx.last_launch()["workgroups"]), flush=True)
    ctx.close()
