"""Two target blobs per lane (context option sym_two_targets) against the default symmetric kernels: parity for every
kind / size / pair shard, then HIP-event kernel time of the wall tt product, alternating, same box."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud
ctx = MobilityContext(0)
rel = lambda a, b: float(torch.linalg.norm(a - b) / torch.linalg.norm(b))
worst = 0.0
for N in (256, 257, 300, 449, 1000, 4097, 10000):
  for wall in (True, False):
    r, f, eta, a = d2_cloud(N)
    rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
    ctx.set_positions(rd, a, None, wall)
    for kind in ("tt", "tr", "rt", "rr"):
      ctx.set_option("sym_two_targets", 0)
      ref = ctx.matvec_device(kind, fd, eta).clone()
      ctx.set_option("sym_two_targets", 2)
      got = ctx.matvec_device(kind, fd, eta).clone()
      assert ctx.get_option("last_path") == 4, ctx.get_option("last_path")
      e = rel(got, ref); worst = max(worst, e)
      assert e < 1e-13, (N, wall, kind, e)
      for G in (2, 3, 8):
        tot = torch.zeros_like(ref)
        for g in range(G):
          tot += ctx.matvec_pairshard_device(kind, fd, eta, g, G)
        e = rel(tot, ref); worst = max(worst, e)
        assert e < 1e-13, (N, wall, kind, G, e)
print("parity of sym_two_targets = 2 against the default kernels (7 sizes x wall / no wall x 4 kinds x whole + 2 / 3 / 8 pair shards): worst rel diff %.1e" % worst, flush=True)
ctx.set_option("timing", 1)
for N in [int(x) for x in sys.argv[1:]] or [10000, 24576, 100000, 262144]:
  r, f, eta, a = d2_cloud(N)
  rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
  out = torch.empty(3 * N, dtype=torch.float64, device="cuda")
  ctx.set_positions(rd, a, None, True)
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < 0.3:
    ctx.matvec_device("tt", fd, eta, out=out); torch.cuda.synchronize()
  reps = 200 if N <= 10000 else (40 if N <= 30000 else (8 if N <= 100000 else 3))
  res = {0: [], 1: []}
  for rnd in range(3):
    for mode in (0, 1):
      ctx.set_option("sym_two_targets", mode)
      for _ in range(3): ctx.matvec_device("tt", fd, eta, out=out)
      torch.cuda.synchronize(); ctx.timing_reset()
      for _ in range(reps): ctx.matvec_device("tt", fd, eta, out=out)
      torch.cuda.synchronize()
      res[mode].append(float(np.mean(ctx.timing_collect(reps))) * 1e3)
      path = ctx.get_option("last_path")
  print("N=%7d wall tt: default %s us | two targets per lane %s us (path %d)  -> x%.3f" % (
      N, " ".join("%.1f" % x for x in res[0]), " ".join("%.1f" % x for x in res[1]), path, np.mean(res[0]) / np.mean(res[1])), flush=True)
ctx.close()
