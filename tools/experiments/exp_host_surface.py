"""Per-call time of the reference-shaped host surface (numpy in, numpy out, synchronous; positions cached after the
first call) against the kernel time: what a RigidMultiblobsWall caller that keeps its vectors on the host pays."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import mobility as mob
from bench import d2_cloud
for N in (1000, 10000, 24576, 100000):
  r, f, eta, a = d2_cloud(N)
  t = np.random.RandomState(1).randn(N, 3)
  for _ in range(200 if N <= 30000 else 5):
    mob.single_wall_mobility_trans_times_force_hip(r, f, eta, a)
  reps = 500 if N <= 30000 else 20
  t0 = time.perf_counter()
  for _ in range(reps):
    u = mob.single_wall_mobility_trans_times_force_hip(r, f, eta, a)
  dt = (time.perf_counter() - t0) / reps
  t0 = time.perf_counter()
  for _ in range(reps):
    u2 = mob.single_wall_mobility_trans_times_force_torque_hip(r, f, t, eta, a)
  dt2 = (time.perf_counter() - t0) / reps
  print("N=%d  tt through the host surface %.1f us/call (%.0f matvecs/s);  fused tt+tr %.1f us/call" % (N, 1e6 * dt, 1.0 / dt, 1e6 * dt2), flush=True)
