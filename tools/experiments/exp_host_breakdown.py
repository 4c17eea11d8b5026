"""Where one call of the reference's call shape goes at N blobs (numpy in / out, synchronous), WITHOUT per-launch events:
the library's own host clock (rmb_last_host_timing: upload, enqueue, wait + download, whole C call) and the Python wrapper."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import mobility as mob
from bench import d2_cloud
for N in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1000", "10000", "24576"])]:
  r, f, eta, a = d2_cloud(N)
  for _ in range(50):
    u = mob.single_wall_mobility_trans_times_force_hip(r, f, eta, a)
  ctx = mob._context(N)
  acc = np.zeros(4); n = 300
  t0 = time.perf_counter()
  for _ in range(n):
    u = mob.single_wall_mobility_trans_times_force_hip(r, f, eta, a)
    ht = ctx.last_host_timing(); acc += [ht["upload_us"], ht["launch_us"], ht["wait_and_download_us"], ht["c_call_us"]]
  tot = (time.perf_counter() - t0) / n * 1e6
  acc /= n
  t0 = time.perf_counter()
  for _ in range(n):
    mob._bind_positions(r, a, np.zeros(3), True)
  bind = (time.perf_counter() - t0) / n * 1e6
  print("N %6d: call %.1f us = bind positions %.1f + C call %.1f (upload %.1f, enqueue %.1f, wait + download %.1f) + python/numpy %.1f" %
        (N, tot, bind, acc[3], acc[0], acc[1], acc[2], tot - bind - acc[3]), flush=True)
