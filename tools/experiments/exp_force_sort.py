"""Force kernel with tile culling on a 262 144-roller monolayer (configs[4] geometry) listed in lattice order and in
random order, with and without the device Morton sort (option "force_sort"); cost of the sort itself."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext, structures as st

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
a = 0.656
loc, _, _ = st.roller_monolayer(n, radius=a, seed=7)
eps, b = 0.0165677856, 0.0656
rng = np.random.RandomState(0)
perm = rng.permutation(n)
ctx = MobilityContext(0)
ctx.set_option("timing", 1)
for label, r in (("lattice order", loc), ("random order", loc[perm])):
  rd = torch.as_tensor(np.ascontiguousarray(r).reshape(-1), device="cuda")
  res = {}
  for sort in (0, 1):
    ctx.set_option("force_sort", sort)
    ctx.set_positions(rd, a, None, wall=False)
    out = ctx.blob_blob_force_device(eps, b, a)
    torch.cuda.synchronize()
    # per call with the sorted copy cached (same configuration), and per call including a new configuration
    ctx.timing_reset()
    for _ in range(5): ctx.blob_blob_force_device(eps, b, a, out=out)
    torch.cuda.synchronize()
    k = float(np.mean(ctx.timing_collect(5)))
    t0 = time.perf_counter()
    for _ in range(5):
      ctx.set_positions(rd, a, None, wall=False)
      ctx.blob_blob_force_device(eps, b, a, out=out)
    torch.cuda.synchronize()
    e2e = (time.perf_counter() - t0) / 5 * 1e3
    res[sort] = out.clone()
    print("%-14s force_sort %d: force kernel %8.3f ms   set_positions + forces end to end %8.3f ms" % (label, sort, k, e2e), flush=True)
  print("               rel diff sorted vs caller's order: %.2e" % float(torch.linalg.norm(res[1] - res[0]) / torch.linalg.norm(res[0])))
ctx.close()
