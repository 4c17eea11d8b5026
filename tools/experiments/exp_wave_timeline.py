"""Per-wave timeline of ONE symmetric wall-tt launch (where do the ~14 us between the 1e4-blob launch and
its issue-bound floor go?).  Diagnostics build (RMB_DIAGNOSTICS=1 -> librmb_mobility_diag.so, option "wave_clock"): every
wave stamps its wall-clock start / end + placement, sym2t_kernel also the shader-clock cycles it spends staging tiles.

  python tools/experiments/exp_wave_timeline.py [N] [key=value ...]      (context options, e.g. sym_oversub=2)

Prints: kernel time by HIP events (clocks primed), the ramp (when waves start), the tail (when they end), how many waves
are resident over time, per-wave duration statistics and the staging share."""
import os, sys, time
os.environ.setdefault("RMB_DIAGNOSTICS", "1")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
opts = dict(kv.split("=") for kv in sys.argv[2:])
r, f, eta, a = d2_cloud(N)
rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
out = torch.empty(3 * N, dtype=torch.float64, device="cuda")
ctx = MobilityContext(0)
for k, v in opts.items():
  ctx.set_option(k, int(v))
ctx.set_positions(rd, a, None, wall=True)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:
  for _ in range(20):
    ctx.matvec_device("tt", fd, eta, out=out)
  torch.cuda.synchronize()
ctx.set_option("timing", 1)
ctx.timing_reset()
for _ in range(200):
  ctx.matvec_device("tt", fd, eta, out=out)
torch.cuda.synchronize()
ev = np.asarray(ctx.timing_collect(200)) * 1e3
ctx.set_option("timing", 0)
path = ctx.get_option("last_path")
print("N = %d options %s: kernel by HIP events avg %.2f us median %.2f min %.2f (last_path %d, %d workgroups)" %
      (N, opts, ev.mean(), np.median(ev), ev.min(), path, ctx.last_launch()["workgroups"]), flush=True)
ctx.set_option("wave_clock", 1)
best = None
for rep in range(5):
  for _ in range(3):
    ctx.matvec_device("tt", fd, eta, out=out)
  torch.cuda.synchronize()
  raw = ctx.wave_clock_collect(1 << 20).copy()
  nw = len(raw) // 2 if path == 4 else len(raw)
  st = raw[:nw].copy(); ph = raw[nw:] if path == 4 else None
  place = (st[:, 1] >> 40) & 0xffffff
  st[:, 0] &= 0xffffffffff; st[:, 1] &= 0xffffffffff
  t = st.astype(np.float64) * 0.01            # us (100 MHz wall clock)
  t -= t[:, 0].min()
  span = t[:, 1].max()
  if best is None or span < best[0]:
    best = (span, t, ph, place)
span, t, ph, place = best
start, end = t[:, 0], t[:, 1]
dur = end - start
print("waves %d | span first start -> last end %.2f us (best of 5 stamped launches)" % (len(t), span))
print("ramp : wave starts  p1 %.2f  p25 %.2f  p50 %.2f  p75 %.2f  p99 %.2f  max %.2f us" % tuple(np.percentile(start, [1, 25, 50, 75, 99, 100])))
print("tail : wave ends    p1 %.2f  p25 %.2f  p50 %.2f  p75 %.2f  p99 %.2f  max %.2f us" % tuple(np.percentile(end, [1, 25, 50, 75, 99, 100])))
print("wave duration       min %.2f  p25 %.2f  p50 %.2f  p75 %.2f  max %.2f us" % tuple(np.percentile(dur, [0, 25, 50, 75, 100])))
# resident waves over time
grid = np.linspace(0, span, 41)
act = [(np.sum((start <= x) & (end > x))) for x in grid]
print("resident waves at t (us): " + " ".join("%.0f:%d" % (x, n_) for x, n_ in zip(grid[::2], act[::2])))
busy = dur.sum() / (len(t) * span)
print("sum of wave durations / (waves x span) = %.3f   (1 = every wave busy from first start to last end)" % busy)
# per SIMD: placement bits -> (xcc, se, sh, cu, simd)
simd = (place & 0xffff) >> 4 & 0x3; cu = (place >> 8) & 0xf; sh = (place >> 12) & 1; se = (place >> 13) & 7; xcc = (place >> 16) & 0xf
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
key_simd = key * 4 + simd
u, cnt = np.unique(key_simd, return_counts=True)
print("SIMDs used %d, waves per SIMD histogram %s; CUs used %d" % (len(u), dict(zip(*np.unique(cnt, return_counts=True))), len(np.unique(key))))
simd_end = np.array([end[key_simd == k].max() for k in u]); simd_start = np.array([start[key_simd == k].min() for k in u])
print("per SIMD: first wave starts p50 %.2f max %.2f | last wave ends p1 %.2f p50 %.2f max %.2f us" %
      (np.median(simd_start), simd_start.max(), np.percentile(simd_end, 1), np.median(simd_end), simd_end.max()))
if ph is not None:
  stage, total = ph[:, 0].astype(np.float64), ph[:, 1].astype(np.float64)
  ok = total > 0
  print("staging share of a wave's cycles: mean %.3f  p50 %.3f  max %.3f | staging cycles per wave mean %.0f (%.2f us at 2.4 GHz), total cycles mean %.0f" %
        ((stage[ok] / total[ok]).mean(), np.median(stage[ok] / total[ok]), (stage[ok] / total[ok]).max(), stage[ok].mean(), stage[ok].mean() / 2400.0, total[ok].mean()))
ctx.close()
