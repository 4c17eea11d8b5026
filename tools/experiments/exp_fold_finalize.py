"""Finalize folded into the sweep (sym2t_kernel<FIN>, option sym_fold_finalize) against sweep + finalize as two dependent
launches: end-to-end time per product (no events: wall clock over many back-to-back products, what `value` measures) and the
sweep kernel alone by HIP events; same box, alternating rounds.  Also a correctness hammer: 300 folded products in a row
must equal the two-launch result to rounding every time (the per-tile counters and the accumulators go back to zero)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud

sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [6500, 10000, 24576, 100000]
ctx = MobilityContext(0)
for N in sizes:
  r, f, eta, a = d2_cloud(N)
  rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
  out = torch.empty(3 * N, dtype=torch.float64, device="cuda")
  ctx.set_positions(rd, a, None, wall=True)
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < 0.4:
    for _ in range(20):
      ctx.matvec_device("tt", fd, eta, out=out)
    torch.cuda.synchronize()
  reps = 400 if N <= 30000 else 20
  e2e, ker = {0: [], 2: []}, {0: [], 2: []}
  for rnd in range(4):
    for mode in ((0, 2) if rnd % 2 == 0 else (2, 0)):
      ctx.set_option("sym_fold_finalize", mode)
      ctx.set_option("timing", 0)
      for _ in range(10):
        ctx.matvec_device("tt", fd, eta, out=out)
      torch.cuda.synchronize(); t0 = time.perf_counter()
      for _ in range(reps):
        ctx.matvec_device("tt", fd, eta, out=out)
      torch.cuda.synchronize()
      e2e[mode].append((time.perf_counter() - t0) / reps * 1e6)
      assert ctx.get_option("last_folded") == (1 if mode else 0) and ctx.get_option("last_path") == 4
      ctx.set_option("timing", 1); ctx.timing_reset()
      for _ in range(min(reps, 100)):
        ctx.matvec_device("tt", fd, eta, out=out)
      torch.cuda.synchronize()
      ker[mode].append(float(np.median(ctx.timing_collect(min(reps, 100)))) * 1e3)
  ctx.set_option("timing", 0)
  print("N %7d: product end to end  two launches %9.2f us  folded %9.2f us  (x %.3f) | sweep kernel by events %9.2f -> %9.2f us" %
        (N, np.median(e2e[0]), np.median(e2e[2]), np.median(e2e[0]) / np.median(e2e[2]), np.median(ker[0]), np.median(ker[2])), flush=True)
  # hammer
  ctx.set_option("sym_fold_finalize", 0)
  ref = ctx.matvec_device("tt", fd, eta).clone()
  ctx.set_option("sym_fold_finalize", 2)
  worst = 0.0
  for k in range(300 if N <= 30000 else 20):
    u = ctx.matvec_device("tt", fd, eta, out=out)
    worst = max(worst, float((u - ref).abs().max() / ref.abs().max()))
  ctx.set_option("sym_fold_finalize", 0)
  u = ctx.matvec_device("tt", fd, eta, out=out)       # and the two-launch path finds clean accumulators afterwards
  worst2 = float((u - ref).abs().max() / ref.abs().max())
  print("          hammer: worst folded-vs-two-launch difference %.2e; two-launch product afterwards %.2e" % (worst, worst2), flush=True)
  ctx.set_option("sym_fold_finalize", 1)
ctx.close()
