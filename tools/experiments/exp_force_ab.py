"""Force kernel (tile culling + Morton sort) under the schedule options of the symmetric kernels: 3D cloud of bench.py and
the configs[4] monolayer, kernel time by HIP events."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext, structures as st
from bench import d2_cloud
ctx = MobilityContext(0); ctx.set_option("timing", 1)
cases = []
for N in (10000, 100000):
  r, f, eta, a = d2_cloud(N); cases.append(("3D cloud %d" % N, r, a))
loc, _, _ = st.roller_monolayer(262144, radius=0.656, seed=7)
perm = np.random.RandomState(1).permutation(len(loc))
cases.append(("monolayer 262144, random order", loc[perm], 0.656))
for name, r, a in cases:
  rd = torch.as_tensor(np.ascontiguousarray(r).reshape(-1), device="cuda")
  ctx.set_positions(rd, a, None, wall=False)
  row = []
  for order, xcd, chunk in ((0, 0, 0), (1, 1, 0), (1, 1, 1024), (0, 0, 1024), (1, 1, 4096), (1, 1, 256)):
    ctx.set_option("sym_order", order); ctx.set_option("sym_xcd", xcd); ctx.set_option("sym_chunk_steps", chunk)
    for _ in range(3): ctx.blob_blob_force_device(3.92, 0.1 * a, a)
    torch.cuda.synchronize(); ctx.timing_reset()
    for _ in range(10): ctx.blob_blob_force_device(3.92, 0.1 * a, a)
    torch.cuda.synchronize()
    t = ctx.timing_collect(100)
    row.append("o%d x%d c%-4d %7.3f ms" % (order, xcd, chunk, float(np.mean(t))))
  print("%-32s %s" % (name, " | ".join(row)), flush=True)
ctx.close()
