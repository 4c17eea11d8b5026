"""Four tile-J accumulators per wave (build flag RMB_SYM_JSLOTS=4) with unit order 2 (4 x 4 sub-blocks inside the 32 x 32
super-blocks) against the default build: HIP-event kernel time of the wall tt product, result compared with the default
order of the same build.  One process per library (RMB_AB_LIB), configurations "order:chunk" from argv after the sizes.

  RMB_AB_LIB=build/ab/j4.so python tools/experiments/exp_jslots.py 10000 100000 -- 1:1024 2:1024 2:2048
"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import _lib as _rmb_lib
if os.environ.get("RMB_AB_LIB"):
  _rmb_lib.LIB_PATH = os.path.abspath(os.environ["RMB_AB_LIB"])
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud
args = sys.argv[1:]
cut = args.index("--") if "--" in args else len(args)
sizes = [int(x) for x in args[:cut]] or [10000, 100000]
cfgs = [tuple(int(v) for v in c.split(":")) for c in args[cut + 1:]] or [(1, 1024)]
tag = os.path.basename(os.environ.get("RMB_AB_LIB", "default"))
ctx = MobilityContext(0); ctx.set_option("timing", 1); ctx.set_option("sym_coop", 0)
for N in sizes:
  r, f, eta, a = d2_cloud(N)
  rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
  out = torch.empty(3 * N, dtype=torch.float64, device="cuda"); ref = None
  ctx.set_positions(rd, a, None, True)
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < 0.3:
    ctx.matvec_device("tt", fd, eta, out=out); torch.cuda.synchronize()
  reps = 200 if N <= 10000 else (40 if N <= 30000 else (8 if N <= 100000 else 3))
  for rnd in range(2):
    for order, chunk in cfgs:
      ctx.set_option("sym_order", order); ctx.set_option("sym_chunk_steps", chunk)
      for _ in range(3): ctx.matvec_device("tt", fd, eta, out=out)
      torch.cuda.synchronize(); ctx.timing_reset()
      for _ in range(reps): ctx.matvec_device("tt", fd, eta, out=out)
      torch.cuda.synchronize()
      k = float(np.mean(ctx.timing_collect(reps))) * 1e3
      if ref is None: ref = out.clone()
      print("%-8s N=%7d order %d chunk %4d: %10.2f us   diff %.1e" % (tag, N, order, chunk, k, float(torch.linalg.norm(out - ref) / torch.linalg.norm(ref))), flush=True)
ctx.close()
