#!/usr/bin/env python
"""Child process of bench.py's `multi_device_surface` extra: the reference-shaped call
single_wall_mobility_trans_times_force_hip(r_vectors, force, eta, a) on ONE device and on ALL visible devices through
the single-process multi-device engine (mobility.set_devices), same inputs, results compared.  Prints one JSON line.

  python tools/multi_surface_probe.py N_DEVICES|d0,d1,... [N_BLOBS ...]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
  from bench import d2_cloud
  from rigidmultiblobswall_amd import mobility as mob
  # N_DEVICES, or an explicit list "0,0,0" (rehearsal on one GPU: the same device several times)
  all_devs = [int(x) for x in sys.argv[1].split(",")] if "," in sys.argv[1] else list(range(int(sys.argv[1])))
  sizes = [int(x) for x in sys.argv[2:]] or [24576, 100000]
  mob.multi_min_blobs = 0
  rows = []
  for n in sizes:
    r, f, eta, a = d2_cloud(n, seed=0)
    per = {}
    u_ref = None
    for label, devs in (("one_device", all_devs[:1]), ("all_devices", all_devs)):
      mob.set_devices(devs)
      for _ in range(3):
        u = mob.single_wall_mobility_trans_times_force_hip(r, f, eta, a)
      reps = 20 if n <= 30000 else 5
      t0 = time.perf_counter()
      for _ in range(reps):
        u = mob.single_wall_mobility_trans_times_force_hip(r, f, eta, a)
      per[label] = {"devices": devs, "ms_per_call": round(1e3 * (time.perf_counter() - t0) / reps, 4)}
      if u_ref is None:
        u_ref = u
      else:
        per["rel_diff_all_vs_one"] = float(np.linalg.norm(u - u_ref) / np.linalg.norm(u_ref))
        ctx = mob._context(n)
        per["peer_access"] = bool(ctx.get_option("peer")) if hasattr(ctx, "n_shards") else None
    per["n_blobs"] = n
    per["speedup"] = round(per["one_device"]["ms_per_call"] / per["all_devices"]["ms_per_call"], 3)
    rows.append(per)
  mob.set_devices(None)
  print(json.dumps({"function": "single_wall_mobility_trans_times_force_hip(r_vectors, force, eta, a), numpy in / out, one process",
                    "engine": "rmb_multi_*: pair shard per device, fixed-order slice reduction through peer reads",
                    "sizes": rows}), flush=True)


if __name__ == "__main__":
  main()
