"""Per-kernel measurement table: every product of the surface at several N (device-resident vectors,
HIP-event kernel time), algorithmic flops per pair from SURVEY.md 8(d), plus the host-surface
(numpy in/out, PCIe-inclusive) rate of the headline product."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rigidmultiblobswall_amd import MobilityContext, mobility as mob
from bench import d2_cloud

FLOPS = {("tt", True): 211, ("tt", False): 62, ("tr", True): 110, ("tr", False): 38, ("rt", True): 110, ("rt", False): 38,
         ("rr", True): 128, ("rr", False): 59, ("tt_tr", True): 321, ("tt_tr", False): 100, ("tt_free", False): 124}
rows = []
for N in (10000, 100000):
  r, f, eta, a = d2_cloud(N)
  rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
  for wall in (True, False):
    ctx = MobilityContext(0); ctx.set_option("timing", 1)
    ctx.set_positions(rd, a, wall=wall)
    for kind in ("tt", "tr", "rt", "rr", "tt_tr", "tt_free"):
      if (kind, wall) not in FLOPS:
        continue
      reps = 20 if N <= 10000 else 5
      v2 = fd if kind == "tt_tr" else None
      for det in ((0, 1) if kind in ("tt", "tr", "rt", "rr", "tt_tr") else (0,)):
        ctx.set_option("deterministic", det)
        for _ in range(2):
          ctx.matvec_device(kind, fd, eta, vec2=v2)
        torch.cuda.synchronize(); ctx.timing_reset()
        for _ in range(reps):
          ctx.matvec_device(kind, fd, eta, vec2=v2)
        torch.cuda.synchronize()
        # kernel time per PRODUCT: the symmetric tt+tr product is two timed launches
        ms = float(np.sum(ctx.timing_collect(4 * reps))) / reps
        tf = FLOPS[(kind, wall)] * float(N) * N / (ms * 1e-3) / 1e12
        rows.append(dict(N=N, kind=kind, wall=wall, path="symmetric" if ctx.last_launch()["chunks"] == 0 else "sweep", kernel_ms=round(ms, 4),
                         flops_per_pair=FLOPS[(kind, wall)], alg_tflops=round(tf, 2), frac_fp64_peak=round(tf / 78.6, 3),
                         gpairs_per_s=round(float(N) * N / (ms * 1e-3) / 1e9, 1)))
        print(rows[-1], flush=True)
      ctx.set_option("deterministic", 0)
    ctx.close()
  # forces
  ctx = MobilityContext(0); ctx.set_option("timing", 1)
  ctx.set_positions(rd, a, wall=False)
  for _ in range(2):
    ctx.blob_blob_force_device(3.92, 0.1 * a, a)
  torch.cuda.synchronize(); ctx.timing_reset()
  reps = 20 if N <= 10000 else 5
  for _ in range(reps):
    ctx.blob_blob_force_device(3.92, 0.1 * a, a)
  torch.cuda.synchronize()
  ms = float(np.mean(ctx.timing_collect(reps)))
  rows.append(dict(N=N, kind="forces", wall=False, path="symmetric" if ctx.last_launch()["chunks"] == 0 else "sweep", kernel_ms=round(ms, 4), flops_per_pair=22,
                   alg_tflops=round(22 * float(N) * N / (ms * 1e-3) / 1e12, 2), gpairs_per_s=round(float(N) * N / (ms * 1e-3) / 1e9, 1)))
  print(rows[-1], flush=True)
  ctx.close()
# host surface (numpy in / numpy out, positions cached after the first call): PCIe-inclusive
for N in (10000, 100000):
  r, f, eta, a = d2_cloud(N)
  mob.single_wall_mobility_trans_times_force_hip(r, f, eta, a)
  t0 = time.perf_counter(); reps = 30 if N <= 10000 else 5
  for _ in range(reps):
    mob.single_wall_mobility_trans_times_force_hip(r, f, eta, a)
  dt = (time.perf_counter() - t0) / reps
  rows.append(dict(N=N, kind="tt", wall=True, path="host surface (numpy in/out, PCIe-inclusive)", ms_per_call=round(dt * 1e3, 4),
                   matvecs_per_s=round(1.0 / dt, 2)))
  print(rows[-1], flush=True)
json.dump(rows, open("gpurun_out/bench_all_kinds.json", "w"), indent=1)
