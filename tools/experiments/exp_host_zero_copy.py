"""Synchronous host entry point rmb_matvec: result stored by the kernels straight into page-locked, device-mapped host
memory + a host memcpy (option host_zero_copy = largest n it applies to) against a device-to-host copy command (0)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud
ctx = MobilityContext(0)
for N, reps in ((100, 1000), (1000, 1000), (4000, 500), (10000, 400), (24576, 100)):
  r, f, eta, a = d2_cloud(N)
  ctx.set_positions(r, a, None, True)
  ref = None
  for _ in range(200 if N <= 10000 else 20): ctx.matvec("tt", f, eta)
  for zc in (0, 1 << 30, 0, 1 << 30):
    ctx.set_option("host_zero_copy", zc)
    for _ in range(5): u = ctx.matvec("tt", f, eta)
    if ref is None: ref = u
    acc = np.zeros(4)
    t0 = time.perf_counter()
    for _ in range(reps):
      u = ctx.matvec("tt", f, eta)
      ht = ctx.last_host_timing(); acc += [ht["upload_us"], ht["launch_us"], ht["wait_and_download_us"], ht["c_call_us"]]
    dt = (time.perf_counter() - t0) / reps * 1e6
    acc /= reps
    print("N=%6d zero_copy %d: %8.1f us per call   (upload %.1f, enqueue %.1f, wait + download %.1f, C call %.1f)   diff %.1e"
          % (N, int(zc > 0), dt, *acc, np.linalg.norm(u - ref) / np.linalg.norm(ref)), flush=True)
ctx.close()
