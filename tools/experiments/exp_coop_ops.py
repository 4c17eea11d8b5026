"""Workgroup-cooperative instances of the generic symmetric skeleton (symx_coop_kernels.h) against the per-wave ones:
multi-vector / multi-block operations, HIP-event kernel time, results compared.  sym_coop 0 = per wave, 1 = default
rule (<= 4 resident rounds, or LDS-bound operations), 2 = always."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud

ctx = MobilityContext(0)
ctx.set_option("timing", 1)
for N in [int(x) for x in sys.argv[1:]] or [10000, 100000]:
  r, f, eta, a = d2_cloud(N)
  rd = torch.as_tensor(r.reshape(-1), device="cuda")
  vs = [torch.as_tensor(np.random.RandomState(k).randn(3 * N), device="cuda") for k in range(4)]
  ctx.set_positions(rd, a, None, True)
  ctx.set_option("sym_coop", 0)
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < 0.3:
    ctx.matvec_device("tt", vs[0], eta); torch.cuda.synchronize()
  reps = 100 if N <= 10000 else 6
  base = None
  print("N=%d" % N)
  for name, op, vecs in (("tt x1 (sym_kernel)", None, vs[:1]), ("tt x2", "tt_multi", vs[:2]), ("tt x3", "tt_multi", vs[:3]),
                         ("tt x4", "tt_multi", vs[:4]), ("rr x4", "rr_multi", vs[:4]), ("fused row", "velocity_from_force_torque", vs[:2]),
                         ("force column", "force_column", vs[:1]), ("grand", "grand", vs[:2])):
    row = []
    ref = None
    for coop in (0, 1, 2):
      ctx.set_option("sym_coop", coop)
      fn = (lambda: (ctx.matvec_device("tt", vecs[0], eta),)) if op is None else (lambda: ctx.matvec_op_device(op, vecs, eta))
      for _ in range(3): out = fn()
      torch.cuda.synchronize(); ctx.timing_reset()
      for _ in range(reps): out = fn()
      torch.cuda.synchronize()
      k = float(np.mean(ctx.timing_collect(reps))) * 1e3
      if ref is None: ref = [o.clone() for o in out]
      err = max(float(torch.linalg.norm(o - q) / torch.linalg.norm(q)) for o, q in zip(out, ref))
      row.append("coop %d: %9.1f us (path %d, diff %.0e)" % (coop, k, ctx.get_option("last_path"), err))
    print("  %-20s %s" % (name, "   ".join(row)), flush=True)
ctx.close()
