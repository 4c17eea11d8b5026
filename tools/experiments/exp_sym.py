"""GPU experiment: symmetric-kernel residency (waves per SIMD) and pinning vs kernel time."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud

def run(N, opts, reps=30):
  r, f, eta, a = d2_cloud(N)
  ctx = MobilityContext(0)
  ctx.set_option("timing", 1)
  for k, v in opts.items():
    ctx.set_option(k, v)
  rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
  ctx.set_positions(rd, a, wall=True)
  out = torch.empty(3 * N, dtype=torch.float64, device="cuda")
  for _ in range(3):
    ctx.matvec_device("tt", fd, eta, out=out)
  torch.cuda.synchronize(); ctx.timing_reset()
  for _ in range(reps):
    ctx.matvec_device("tt", fd, eta, out=out)
  torch.cuda.synchronize()
  ms = ctx.timing_collect(reps)
  ll = ctx.last_launch(); ctx.close()
  return float(np.mean(ms)), float(np.min(ms)), ll

for N in [int(x) for x in sys.argv[1:]] or [10000, 24576, 100000]:
  base = run(N, {"deterministic": 1}, reps=10 if N > 50000 else 30)
  print("N=%d sweep           avg=%.4f min=%.4f ms %s" % (N, base[0], base[1], base[2]), flush=True)
  for pin in (1, 0):
    for wps in (2, 3, 4, 5, 6):
      t = run(N, {"sym_wps": wps, "sym_pin": pin}, reps=10 if N > 50000 else 30)
      print("N=%d sym wps=%d pin=%d avg=%.4f min=%.4f ms  (%.2fx vs sweep) %s" % (N, wps, pin, t[0], t[1], base[0] / t[0], t[2]), flush=True)
