"""Host entry point rmb_matvec: result returned through a page-locked staging buffer + memcpy (option host_staging = 1)
against a device-to-host copy straight into the caller's pageable array (0)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud
ctx = MobilityContext(0)
for N, reps in ((1000, 500), (10000, 400), (100000, 10)):
  r, f, eta, a = d2_cloud(N)
  ctx.set_positions(r, a, None, True)
  for _ in range(200 if N <= 10000 else 3): ctx.matvec("tt", f, eta)
  for staging in (0, 1, 0, 1):
    ctx.set_option("host_staging", staging)
    for _ in range(5): ctx.matvec("tt", f, eta)
    acc = np.zeros(4)
    t0 = time.perf_counter()
    for _ in range(reps):
      ctx.matvec("tt", f, eta)
      ht = ctx.last_host_timing(); acc += [ht["upload_us"], ht["launch_us"], ht["wait_and_download_us"], ht["c_call_us"]]
    dt = (time.perf_counter() - t0) / reps * 1e6
    acc /= reps
    print("N=%6d host_staging %d: %8.1f us per call   (upload %.1f, enqueue %.1f, wait + download %.1f, C call %.1f)" % (N, staging, dt, *acc), flush=True)
ctx.close()
