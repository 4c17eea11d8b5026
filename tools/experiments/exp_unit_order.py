"""Unit order (row-major vs blocked, option sym_order) x workgroup numbering (plain vs XCD-aware, option sym_xcd) of the
symmetric kernels: HIP-event kernel time at several sizes; results compared.  (HBM fetch traffic: run this script under
rocprofv3 --pmc FETCH_SIZE with one configuration per launch group, see profiles/r4_unit_order.txt.)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud
sizes = [int(x) for x in sys.argv[1:]] or [10000, 24576, 100000, 262144]
ctx = MobilityContext(0); ctx.set_option("timing", 1)
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
    for order, xcd, chunk in ((0, 0, 0), (1, 1, 0), (1, 1, 512), (1, 1, 1024), (1, 1, 2048), (0, 0, 1024)):
      ctx.set_option("sym_order", order); ctx.set_option("sym_xcd", xcd); ctx.set_option("sym_chunk_steps", chunk)
      for _ in range(3): ctx.matvec_device("tt", fd, eta, out=out)
      torch.cuda.synchronize(); ctx.timing_reset()
      for _ in range(reps): ctx.matvec_device("tt", fd, eta, out=out)
      torch.cuda.synchronize()
      k = float(np.mean(ctx.timing_collect(reps))) * 1e3
      if ref is None: ref = out.clone()
      print("N=%7d order %d xcd %d chunk %4d: %10.2f us   diff %.1e" % (N, order, xcd, chunk, k, float(torch.linalg.norm(out - ref) / torch.linalg.norm(ref))), flush=True)
ctx.close()
