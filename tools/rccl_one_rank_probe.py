#!/usr/bin/env python
"""Child process of bench.py's `rccl_one_rank` extra: what the all-reduce of the N > 1 step costs on the stream.

A one-GPU box cannot give RCCL a second rank, but a ONE-rank "nccl" group runs the same code: ShardedMobility with
always_exchange issues the fp64 all-reduce of u after every pair-shard product through torch.distributed -> RCCL
(communicator set-up, ProcessGroupNCCL's bookkeeping, the event hand-off between the library's stream and RCCL's; with
one rank RCCL itself has nothing to move for an in-place all-reduce).  The step is timed with and without it; the
difference is the per-step price an N-rank run pays BEFORE any byte crosses xGMI.  Prints one JSON line.

  python tools/rccl_one_rank_probe.py [N_BLOBS ...]
"""
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
  import torch
  import torch.distributed as dist
  from bench import d2_cloud
  from rigidmultiblobswall_amd.distributed import HipBackend, ShardedMobility
  sizes = [int(x) for x in sys.argv[1:]] or [10000]
  s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
  os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
  dev = torch.device("cuda:0")
  torch.cuda.set_device(dev)
  dist.init_process_group("nccl", device_id=dev)
  backend = HipBackend(dev)
  sm = ShardedMobility(backend, device=dev, always_exchange=True)
  plain = ShardedMobility(backend, device=dev)       # same context, the one-rank default: no collective at all
  rows = []
  for n in sizes:
    r, f, eta, a = d2_cloud(n, seed=0)
    sm.set_local_positions(torch.as_tensor(r.reshape(-1), device=dev), n, a, wall=True)
    fd = torch.as_tensor(f.reshape(-1), device=dev)
    out = torch.empty(3 * n, dtype=torch.float64, device=dev)
    plain.n, plain.begin, plain.end, plain.block = sm.n, sm.begin, sm.end, sm.block
    steps = {"local": lambda: plain.matvec_replicated("tt", fd, eta, out=out), "all_reduced": lambda: sm.matvec_replicated("tt", fd, eta, out=out)}
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:       # clocks primed as in the headline
      for _ in range(20):
        steps["all_reduced"]()
      torch.cuda.synchronize(dev)
    ref = None
    row = {"n_blobs": n, "allreduce_bytes": 24 * n}
    reps = 200 if n <= 20000 else 10
    for rnd in range(2):
      for name, step in steps.items():
        for _ in range(5):
          step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
          step()
        torch.cuda.synchronize(dev)
        row["ms_per_step_" + name] = round(1e3 * (time.perf_counter() - t0) / reps, 4)
        if ref is None:
          ref = out.clone()
        else:
          assert float(torch.linalg.norm(out - ref) / torch.linalg.norm(ref)) < 1e-13
    row["allreduce_overhead_us_per_step"] = round(1e3 * (row["ms_per_step_all_reduced"] - row["ms_per_step_local"]), 1)
    rows.append(row)
  dist.barrier()
  dist.destroy_process_group()
  print(json.dumps({"backend": "nccl (RCCL) through torch.distributed, one-rank group, always_exchange",
                    "what": "pair-shard product + fp64 all-reduce of u against the product alone; same stream order as an N-rank run",
                    "sizes": rows}), flush=True)


if __name__ == "__main__":
  main()
