"""Where do the microseconds of one rank's pair shard go at N = 1e4?

For G in (1, 2, 4, 8): sweep time by HIP events (clocks primed), per-wave start / end stamps (ramp, duration, tail),
the same with skip_pairs = 1 (prologue + epilogue only, no pair arithmetic), and the end-to-end time of
sweep + finalize per call without events (what a rank really spends before its all-reduce)."""
import os, sys, time
os.environ.setdefault("RMB_DIAGNOSTICS", "1")   # skip_pairs / wave_clock exist in the diagnostics build only (librmb_mobility_diag.so)
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
r, f, eta, a = d2_cloud(N)
rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
out = torch.empty(3 * N, dtype=torch.float64, device="cuda")
ctx = MobilityContext(0)
ctx.set_positions(rd, a, None, wall=True)
# prime the clocks
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
  for _ in range(20):
    ctx.matvec_device("tt", fd, eta, out=out)
  torch.cuda.synchronize()

opts = [dict()]
if len(sys.argv) > 2:
  opts += [dict(kv.split("=") for kv in s.split(",")) for s in sys.argv[2:]]
for extra in opts:
  for k, v in extra.items():
    ctx.set_option(k, int(v))
  print("options", extra, flush=True)
  for G in (1, 2, 4, 8):
    sh = G // 2
    # end-to-end per call, no events
    ctx.set_option("timing", 0); ctx.set_option("wave_clock", 0); ctx.set_option("skip_pairs", 0)
    for _ in range(20): ctx.matvec_pairshard_device("tt", fd, eta, sh, G, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): ctx.matvec_pairshard_device("tt", fd, eta, sh, G, out=out)
    torch.cuda.synchronize(); e2e = (time.perf_counter() - t0) / 200 * 1e6
    res = {}
    for skip in (0, 1):
      ctx.set_option("timing", 1); ctx.set_option("skip_pairs", skip)
      for _ in range(10): ctx.matvec_pairshard_device("tt", fd, eta, sh, G, out=out)
      torch.cuda.synchronize(); ctx.timing_reset()
      for _ in range(50): ctx.matvec_pairshard_device("tt", fd, eta, sh, G, out=out)
      torch.cuda.synchronize()
      ev = float(np.mean(ctx.timing_collect(50))) * 1e3
      ctx.set_option("wave_clock", 1)
      ctx.matvec_pairshard_device("tt", fd, eta, sh, G, out=out); torch.cuda.synchronize()
      raw = ctx.wave_clock_collect(65536).copy()
      ctx.set_option("wave_clock", 0)
      raw[:, 1] &= 0xffffffffff; raw[:, 0] &= 0xffffffffff
      st = raw.astype(np.float64) * 0.01
      t_first = st[:, 0].min()
      start, end = st[:, 0] - t_first, st[:, 1] - t_first
      res[skip] = (ev, len(st), start.max(), np.median(end - start), (end - start).max(), end.max())
    ctx.set_option("skip_pairs", 0); ctx.set_option("timing", 0)
    wg = ctx.last_launch()["workgroups"]
    for skip in (0, 1):
      ev, nw, smax, dmed, dmax, emax = res[skip]
      print("G=%d skip_pairs=%d: sweep %.1f us by events | %d waves (%d workgroups) | last wave starts at %.1f us | duration med %.1f max %.1f us | "
            "last wave ends at %.1f us" % (G, skip, ev, nw, wg, smax, dmed, dmax, emax), flush=True)
    print("G=%d: sweep + finalize end to end %.1f us per call (ideal share of the primed full sweep: see G=1 / G)" % (G, e2e), flush=True)
  for k in extra:
    ctx.set_option(k, {"sym_min_steps": 64, "sym_oversub": 8, "sym_wps": 0, "sym_fine_steps": 0}.get(k, 0))
ctx.close()
