"""True residency of the symmetric kernels against what the launch plan assumes (round 3: the occupancy API counts
the 86 architectural VGPRs of sym_kernel and says 5 waves per SIMD; with its 9 AGPR spill slots the kernel allocates
104 registers and the hardware holds 4).  Times the headline product and one pair shard with the plan left alone and
with the residency capped ("sym_wps"), for the library in RMB_AB_LIB (default: the in-tree build)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import _lib as _rmb_lib
if os.environ.get("RMB_AB_LIB"):
  _rmb_lib.LIB_PATH = os.path.abspath(os.environ["RMB_AB_LIB"])
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud

tag = os.environ.get("RMB_AB_TAG", "in-tree")
for N in [int(x) for x in sys.argv[1:]] or [10000, 24576, 100000]:
  r, f, eta, a = d2_cloud(N)
  rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
  out = torch.empty(3 * N, dtype=torch.float64, device="cuda")
  ctx = MobilityContext(0)
  ctx.set_positions(rd, a, None, wall=True)
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < 0.3:
    for _ in range(20):
      ctx.matvec_device("tt", fd, eta, out=out)
    torch.cuda.synchronize()
  ctx.set_option("timing", 1)
  reps = 200 if N <= 30000 else 20
  for G in (1, 8):
    for wps in (0, 5, 4, 3):
      ctx.set_option("sym_wps", wps)
      for kind in ("tt", "rr"):
        for _ in range(5): ctx.matvec_pairshard_device(kind, fd, eta, G // 2, G, out=out)
        torch.cuda.synchronize(); ctx.timing_reset()
        for _ in range(reps): ctx.matvec_pairshard_device(kind, fd, eta, G // 2, G, out=out)
        torch.cuda.synchronize()
        t = ctx.timing_collect(reps)
        print("%s N=%d G=%d sym_wps=%d %s: %.2f us (min %.2f), workgroups %d" %
              (tag, N, G, wps, kind, float(np.mean(t)) * 1e3, float(np.min(t)) * 1e3, ctx.last_launch()["workgroups"]), flush=True)
  ctx.close()
