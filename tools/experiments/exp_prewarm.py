"""Two measurements behind bench.py's declared pre-warm and DESIGN's atomic-flush claim (run on the GPU box).

1. Ramp: per-launch HIP-event time of the wall-tt product at N = 1e4 from a cold start (first launches after
   context creation) -- how many launches / milliseconds the clocks need to settle.
2. skip_pairs A/B at N = 1e4: 0 = full kernel, 2 = no flush of u_J to the global accumulators (the 55 MB of
   global_atomic_add_f64 per launch), 1 = schedule + loads + flushes without pair arithmetic, 3 = neither.
   Results with skip_pairs != 0 are WRONG by construction; only the time is used.
"""
import os
os.environ.setdefault("RMB_DIAGNOSTICS", "1")   # skip_pairs / wave_clock exist in the diagnostics build only (librmb_mobility_diag.so)
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
r, f, eta, a = d2_cloud(N)
rd = torch.as_tensor(r.reshape(-1), device="cuda")
fd = torch.as_tensor(f.reshape(-1), device="cuda")
out = torch.empty_like(fd)

ctx = MobilityContext(0)
ctx.set_option("timing", 1)
ctx.set_positions(rd, a, wall=True)
torch.cuda.synchronize()
time.sleep(2.0)    # let the device idle down
ctx.timing_reset()
t0 = time.perf_counter()
n_cold = 600
for _ in range(n_cold):
  ctx.matvec_device("tt", fd, eta, out=out)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
ms = ctx.timing_collect(n_cold)
print("cold start, %d launches in %.1f ms wall" % (n_cold, 1e3 * wall))
for lo, hi in ((0, 5), (5, 25), (25, 50), (50, 100), (100, 200), (200, 400), (400, 600)):
  print("  launches %3d..%3d: kernel avg %.4f ms  min %.4f  max %.4f" % (lo, hi, ms[lo:hi].mean(), ms[lo:hi].min(), ms[lo:hi].max()))

print("idle 0.5 s, then 20 launches (what --warmup 5 --steps 20 sees in a fresh process after init):")
time.sleep(0.5)
ctx.timing_reset()
for _ in range(25):
  ctx.matvec_device("tt", fd, eta, out=out)
torch.cuda.synchronize()
ms = ctx.timing_collect(25)
print("  first 5: %.4f   next 20: %.4f ms" % (ms[:5].mean(), ms[5:].mean()))

print("skip_pairs A/B (N = %d, 200 launches each after 100 warm launches):" % N)
for rep in range(2):
  for sp in (0, 2, 1, 3):
    ctx.set_option("skip_pairs", sp)
    for _ in range(100):
      ctx.matvec_device("tt", fd, eta, out=out)
    torch.cuda.synchronize()
    ctx.timing_reset()
    for _ in range(200):
      ctx.matvec_device("tt", fd, eta, out=out)
    torch.cuda.synchronize()
    ms = ctx.timing_collect(200)
    print("  skip_pairs=%d: kernel avg %.4f ms  median %.4f  min %.4f" % (sp, ms.mean(), np.median(ms), ms.min()), flush=True)
ctx.set_option("skip_pairs", 0)
ctx.close()
