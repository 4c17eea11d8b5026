"""Blob-blob forces with and without tile culling ("force_cull") on the configs[4] monolayer (262 144 rollers, the roller
deck's debye length) and on the 5 % cloud; HIP events around the force kernel, clocks primed."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext, structures as st
from bench import d2_cloud
ctx = MobilityContext(0)
for name, N in (("monolayer", 262144), ("monolayer", 20000), ("cloud", 100000)):
  if name == "monolayer":
    a, eps, b = 0.656, 0.0165677856, 0.0656
    r, _, _ = st.roller_monolayer(N, radius=a, seed=7)
  else:
    r, _, _, a = d2_cloud(N); eps, b = 0.3, 0.1 * a
  rd = torch.as_tensor(r.reshape(-1), device="cuda")
  ctx.set_positions(rd, a, None, wall=False)
  out = torch.empty(3 * N, dtype=torch.float64, device="cuda")
  res = {}
  for cull in (0, 1):
    ctx.set_option("force_cull", cull); ctx.set_option("timing", 1)
    for _ in range(3): ctx.blob_blob_force_device(eps, b, a, out=out)
    torch.cuda.synchronize(); ctx.timing_reset()
    for _ in range(10): ctx.blob_blob_force_device(eps, b, a, out=out)
    torch.cuda.synchronize()
    res[cull] = (float(np.mean(ctx.timing_collect(10))), out.clone())
  diff = float((res[1][1] - res[0][1]).abs().max() / res[0][1].abs().max())
  print("%s N=%d: force kernel %.3f ms without culling, %.3f ms with (%.1fx), max difference %.1e of the largest force" %
        (name, N, res[0][0], res[1][0], res[0][0] / res[1][0], diff), flush=True)
ctx.close()
