"""Pseudo-periodic products (3^d image boxes): kernel time of the symmetric path vs the open-boundary one."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud
for N in (10000, 24576):
  r, f, eta, a = d2_cloud(N)
  Lbox = float(r[:, 0].max()) * 1.02
  rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
  for L in (None, np.array([Lbox, 0.0, 0.0]), np.array([Lbox, Lbox, 0.0])):
    for kind in ("tt", "rr"):
      ctx = MobilityContext(0); ctx.set_option("timing", 1)
      ctx.set_positions(rd, a, L, wall=True)
      for _ in range(2): ctx.matvec_device(kind, fd, eta)
      torch.cuda.synchronize(); ctx.timing_reset()
      for _ in range(5): ctx.matvec_device(kind, fd, eta)
      torch.cuda.synchronize()
      ms = float(np.mean(ctx.timing_collect(5)))
      boxes = 1 if L is None else 3 ** int((L > 0).sum())
      print("N=%d kind=%s boxes=%d path=%s kernel %.3f ms (%.3f ms per box)" %
            (N, kind, boxes, "symmetric" if ctx.last_launch()["chunks"] == 0 else "sweep", ms, ms / boxes), flush=True)
      ctx.close()
