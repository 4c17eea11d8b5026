"""Per-call cost of the single-process multi-device engine on ONE GPU (the device listed G times): what the hand-offs
(input event, per-shard partial / reduce events, the slice reduction, the final waits) add to the plain context, and
the host surface through both.  On a node the shards run on different devices; here they share one, so the sweep time
itself cannot shrink -- the difference to the plain context is the engine's overhead."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from rigidmultiblobswall_amd.multi import MultiContext
from bench import d2_cloud


def per_call(fn, reps):
  for _ in range(10): fn()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(reps): fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / reps * 1e6


for N, reps in ((1000, 500), (10000, 400), (100000, 10)):
  r, f, eta, a = d2_cloud(N)
  rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
  out = torch.empty(3 * N, dtype=torch.float64, device="cuda")
  ctx = MobilityContext(0); ctx.set_positions(rd, a, None, True)
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < 0.3:
    for _ in range(20): ctx.matvec_device("tt", fd, eta, out=out)
    torch.cuda.synchronize()
  base_d = per_call(lambda: ctx.matvec_device("tt", fd, eta, out=out), reps)
  base_h = per_call(lambda: ctx.matvec("tt", f, eta), reps)
  print("N=%7d  plain context: device entry %8.1f us   host entry %8.1f us" % (N, base_d, base_h), flush=True)
  for G in (1, 2, 4, 8):
    for nopeer, threads in ((0, 1), (0, 0), (1, 1)):
      os.environ["RMB_MULTI_NO_PEER"] = str(nopeer)
      os.environ["RMB_MULTI_THREADS"] = str(threads)
      m = MultiContext([0] * G); m.set_positions(rd, a, None, True)
      d = per_call(lambda: m.matvec_device("tt", fd, eta, out=out), reps)
      h = per_call(lambda: m.matvec("tt", f, eta), reps)
      m.set_option("deterministic", 2)
      dd = per_call(lambda: m.matvec_device("tt", fd, eta, out=out), reps)
      print("          engine G=%d %s: device entry %8.1f us (%+6.1f)   host entry %8.1f us (%+6.1f)   deterministic=2 device %8.1f us"
            % (G, ("staged " if nopeer else "peer   ") + ("workers" if m.get_option("threads") else "serial "), d, d - base_d, h, h - base_h, dd), flush=True)
      m.close()
  ctx.close()
