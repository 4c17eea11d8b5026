"""rmb_matvec result hand-off, round 5: finalize storing (coalesced) into mapped host memory + one stream wait + memcpy
(option host_zero_copy = 1 MB, the default) against the device-to-host copy command (0).  Alternating passes per size; the
ordinary path is re-measured after every mapped pass (round 4 saw it slow down after mapped-memory stores with the old
8-byte strided finalize stores)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud
ctx = MobilityContext(0)
for N, reps in ((200, 1000), (1000, 1000), (4000, 500), (10000, 400), (24576, 100), (43000, 50), (100000, 10)):
  r, f, eta, a = d2_cloud(N)
  ctx.set_positions(r, a, None, True)
  ref = None
  for _ in range(100 if N <= 10000 else 10): ctx.matvec("tt", f, eta)
  fd = torch.as_tensor(f.reshape(-1), device="cuda"); od = torch.empty(3 * N, dtype=torch.float64, device="cuda")
  reps = min(reps, 150)
  for zc, zin in ((0, 0), (1 << 22, 0), (1 << 22, 1), (0, 0), (1 << 22, 0), (1 << 22, 1)):
    ctx.set_option("host_zero_copy", zc); ctx.set_option("host_zero_copy_in", zin)
    # The synchronous host path keeps the GPU ~70 % busy at best and the fp64 clock sags within tens of milliseconds of such a
    # duty cycle (a first version of this script, like round 4's, read that sag as an after-effect of mapped memory): prime
    # the clocks with back-to-back device products before EVERY measurement and keep the measurement short.
    t_p = time.perf_counter()
    while time.perf_counter() - t_p < 0.3:
      for _ in range(20): ctx.matvec_device("tt", fd, eta, out=od)
      torch.cuda.synchronize()
    for _ in range(3): u = ctx.matvec("tt", f, eta)
    if ref is None: ref = u
    acc = np.zeros(4)
    t0 = time.perf_counter()
    for _ in range(reps):
      u = ctx.matvec("tt", f, eta)
      ht = ctx.last_host_timing(); acc += [ht["upload_us"], ht["launch_us"], ht["wait_and_download_us"], ht["c_call_us"]]
    dt = (time.perf_counter() - t0) / reps * 1e6
    acc /= reps
    print("N=%6d zero_copy out %d in %d: %8.1f us per call   (upload %.1f, enqueue %.1f, wait + download %.1f, C call %.1f)   diff %.1e"
          % (N, int(zc > 0), zin, dt, *acc, np.linalg.norm(u - ref) / np.linalg.norm(ref)), flush=True)
ctx.close()
