"""Why do the waves of a one-round pair-shard launch start over ~20 us when the chip starts 5120 empty waves in 3 us
(tools/ubench_dispatch.hip)?  Per-wave start / end stamps + placement (XCC, SE, CU, SIMD) of one G = 8 shard at 1e4."""
import os, sys, time
os.environ.setdefault("RMB_DIAGNOSTICS", "1")   # skip_pairs / wave_clock exist in the diagnostics build only (librmb_mobility_diag.so)
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
r, f, eta, a = d2_cloud(N)
rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
out = torch.empty(3 * N, dtype=torch.float64, device="cuda")
ctx = MobilityContext(0)
ctx.set_positions(rd, a, None, wall=True)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
  for _ in range(20):
    ctx.matvec_device("tt", fd, eta, out=out)
  torch.cuda.synchronize()
for opts in ({}, {"sym_wps": 4}, {"sym_wps": 3}, {"sym_pin": 0}, {"sym_wps": 2}):
  for k, v in opts.items():
    ctx.set_option(k, v)
  for skip in (0, 1):
    ctx.set_option("skip_pairs", skip)
    for _ in range(20): ctx.matvec_pairshard_device("tt", fd, eta, G // 2, G, out=out)
    ctx.set_option("wave_clock", 1)
    ctx.matvec_pairshard_device("tt", fd, eta, G // 2, G, out=out); torch.cuda.synchronize()
    raw = ctx.wave_clock_collect(65536).copy()
    ctx.set_option("wave_clock", 0)
    hw = (raw[:, 1] >> 40) & 0xffff; xcc = (raw[:, 1] >> 56) & 0xf
    raw[:, 1] &= 0xffffffffff; raw[:, 0] &= 0xffffffffff
    st = raw.astype(np.float64) * 0.01
    t_first = st[:, 0].min()
    start, end = st[:, 0] - t_first, st[:, 1] - t_first
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    sid = cuid * 4 + simd
    nw = len(st)
    print("options %s skip_pairs %d: %d waves on %d CUs / %d SIMDs; waves per SIMD histogram %s" %
          (opts, skip, nw, len(np.unique(cuid)), len(np.unique(sid)), np.bincount(np.unique(sid, return_counts=True)[1])))
    print("   start: p50 %.1f p90 %.1f max %.1f us | duration p50 %.1f p90 %.1f max %.1f | end max %.1f us" %
          (np.percentile(start, 50), np.percentile(start, 90), start.max(), np.percentile(end - start, 50),
           np.percentile(end - start, 90), (end - start).max(), end.max()))
    # start time against dispatch order (wave index = 4 * workgroup + wave)
    q = nw // 8
    print("   start by dispatch order (eighths of the grid, median us): %s" %
          " ".join("%.1f" % np.median(start[k * q:(k + 1) * q]) for k in range(8)))
    print("   duration by dispatch order (median us):                  %s" %
          " ".join("%.1f" % np.median((end - start)[k * q:(k + 1) * q]) for k in range(8)))
    # did a late starter take the slot of a wave that had ended on its SIMD?
    late = np.nonzero(start > 8.0)[0]
    reused = 0
    for w in late:
      same = (sid == sid[w])
      reused += int(np.any(end[same] <= start[w] + 0.05))
    print("   waves starting after 8 us: %d, of which %d start after another wave of their SIMD has ended" % (len(late), reused))
    # per XCC: first and last start
    print("   per XCC last start (us): %s" % " ".join("%.1f" % start[xcc == x].max() for x in range(8) if np.any(xcc == x)))
  ctx.set_option("skip_pairs", 0)
  for k in opts:
    ctx.set_option(k, {"sym_wps": 0, "sym_pin": 1}[k])
ctx.close()
