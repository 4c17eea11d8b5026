"""Workgroup-cooperative symmetric kernel (option "sym_coop") against the per-wave one: HIP-event kernel time and
end-to-end time per call for small suspensions, pair shards of 1e4 blobs and full products; results compared."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud

ctx = MobilityContext(0)


def measure(fn, reps):
  for _ in range(10): fn()
  torch.cuda.synchronize(); ctx.timing_reset()
  t0 = time.perf_counter()
  for _ in range(reps): fn()
  torch.cuda.synchronize()
  e2e = (time.perf_counter() - t0) / reps * 1e6
  return float(np.mean(ctx.timing_collect(reps))) * 1e3, e2e


def prime(fd, eta, out):
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < 0.3:
    for _ in range(20): ctx.matvec_device("tt", fd, eta, out=out)
    torch.cuda.synchronize()


ctx.set_option("timing", 1)
for N in [int(x) for x in sys.argv[1:]] or [1000, 2000, 4000, 10000, 24576, 100000]:
  r, f, eta, a = d2_cloud(N)
  rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
  out = torch.empty(3 * N, dtype=torch.float64, device="cuda"); ref = torch.empty_like(out)
  ctx.set_positions(rd, a, None, True)
  ctx.set_option("sym_coop", 0); ctx.set_option("sym_fine_steps", 0)
  prime(fd, eta, out)
  reps = 300 if N <= 10000 else (50 if N <= 30000 else 8)
  ctx.matvec_device("tt", fd, eta, out=ref)
  print("N=%d full product" % N)
  for coop, fine in ((0, 0), (2, 0), (2, 16), (2, 32), (2, 8)):
    ctx.set_option("sym_coop", coop); ctx.set_option("sym_fine_steps", fine)
    k, e = measure(lambda: ctx.matvec_device("tt", fd, eta, out=out), reps)
    err = float(torch.linalg.norm(out - ref) / torch.linalg.norm(ref))
    print("   coop %d fine %2d: kernel %9.2f us   end-to-end %9.2f us   wgs %5d   rel diff vs per-wave %.1e"
          % (coop, fine, k, e, ctx.last_launch()["workgroups"], err), flush=True)
  if N == 10000:
    for kind in ("rr", "tr"):
      for coop in (0, 2):
        ctx.set_option("sym_coop", coop); ctx.set_option("sym_fine_steps", 0)
        k, e = measure(lambda: ctx.matvec_device(kind, fd, eta, out=out), reps)
        print("   %s coop %d: kernel %9.2f us" % (kind, coop, k), flush=True)
    for G in (2, 4, 8):
      ctx.set_option("sym_coop", 0); ctx.set_option("sym_fine_steps", 0)
      ctx.matvec_pairshard_device("tt", fd, eta, G // 2, G, out=ref)
      print("N=%d pair shard %d of %d" % (N, G // 2, G))
      for coop, fine in ((0, 0), (1, 0), (1, 16), (1, 24), (1, 32), (1, 8)):
        ctx.set_option("sym_coop", coop); ctx.set_option("sym_fine_steps", fine)
        k, e = measure(lambda: ctx.matvec_pairshard_device("tt", fd, eta, G // 2, G, out=out), reps)
        err = float(torch.linalg.norm(out - ref) / torch.linalg.norm(ref))
        print("   coop %d fine %2d: kernel %9.2f us   end-to-end %9.2f us   wgs %5d   rel diff %.1e"
              % (coop, fine, k, e, ctx.last_launch()["workgroups"], err), flush=True)
ctx.close()
