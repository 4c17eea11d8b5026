"""Timing of the source->target product (K13) through the host surface: ns = nt = N, per-blob radii."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import mobility as mob
from bench import d2_cloud
for N in (10000, 100000):
  r, f, eta, a = d2_cloud(N)
  rng = np.random.RandomState(1)
  tgt = r + rng.randn(N, 3) * 0.3; tgt[:, 2] = np.abs(tgt[:, 2]) + 0.1
  rs = a * (0.5 + rng.rand(N)); rt = a * (0.5 + rng.rand(N))
  for wall, fn in ((True, mob.single_wall_mobility_trans_times_force_source_target_hip),
                   (False, mob.no_wall_mobility_trans_times_force_source_target_hip)):
    fn(r, tgt, f, rs, rt, eta)
    reps = 10 if N <= 10000 else 3
    t0 = time.perf_counter()
    for _ in range(reps):
      fn(r, tgt, f, rs, rt, eta)
    dt = (time.perf_counter() - t0) / reps
    print("N=%d wall=%s: %.3f ms/call (host surface), %.1f Gpairs/s" % (N, wall, dt * 1e3, N * N / dt / 1e9), flush=True)
