"""Pair shards of the multi-block operations at 1e4 blobs (what one of G GPUs runs): per-wave vs cooperative skeleton."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud
ctx = MobilityContext(0); ctx.set_option("timing", 1)
N = 10000
r, f, eta, a = d2_cloud(N)
rd = torch.as_tensor(r.reshape(-1), device="cuda")
vs = [torch.as_tensor(np.random.RandomState(k).randn(3 * N), device="cuda") for k in range(4)]
ctx.set_positions(rd, a, None, True)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
  ctx.matvec_device("tt", vs[0], eta); torch.cuda.synchronize()
for G in (4, 8):
  for name, op, vecs in (("fused row", "velocity_from_force_torque", vs[:2]), ("grand", "grand", vs[:2]), ("tt x2", "tt_multi", vs[:2]), ("tt x4", "tt_multi", vs[:4])):
    row = []
    for coop in (0, 1):
      ctx.set_option("sym_coop", coop)
      for _ in range(5): ctx.matvec_op_device(op, vecs, eta, shard=G // 2, nshards=G)
      torch.cuda.synchronize(); ctx.timing_reset()
      for _ in range(200): ctx.matvec_op_device(op, vecs, eta, shard=G // 2, nshards=G)
      torch.cuda.synchronize()
      row.append("coop %d: %7.1f us (path %d, %d wgs)" % (coop, float(np.mean(ctx.timing_collect(200))) * 1e3, ctx.get_option("last_path"), ctx.last_launch()["workgroups"]))
    print("1/%d shard  %-10s %s" % (G, name, "   ".join(row)), flush=True)
for n_small in (1000, 2000):
  r, f, eta, a = d2_cloud(n_small)
  ctx.set_positions(torch.as_tensor(r.reshape(-1), device="cuda"), a, None, True)
  v2 = [v[:3 * n_small].contiguous() for v in vs]
  for name, op, vecs in (("fused row", "velocity_from_force_torque", v2[:2]), ("grand", "grand", v2[:2]), ("tt x3", "tt_multi", v2[:3])):
    row = []
    for coop in (0, 1):
      ctx.set_option("sym_coop", coop)
      for _ in range(5): ctx.matvec_op_device(op, vecs, eta)
      torch.cuda.synchronize(); ctx.timing_reset()
      for _ in range(200): ctx.matvec_op_device(op, vecs, eta)
      torch.cuda.synchronize()
      row.append("coop %d: %7.1f us (path %d)" % (coop, float(np.mean(ctx.timing_collect(200))) * 1e3, ctx.get_option("last_path")))
    print("N=%d  %-10s %s" % (n_small, name, "   ".join(row)), flush=True)
ctx.close()
