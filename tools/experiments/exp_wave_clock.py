"""Where does the symmetric kernel's wall time go at small N?  Per-wave start/end stamps."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud
for N in (10000, 24576):
  r, f, eta, a = d2_cloud(N)
  ctx = MobilityContext(0); ctx.set_option("timing", 1); ctx.set_option("wave_clock", 1)
  rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
  ctx.set_positions(rd, a, wall=True)
  for _ in range(5):
    ctx.matvec_device("tt", fd, eta)
  torch.cuda.synchronize(); ctx.timing_reset()
  ctx.matvec_device("tt", fd, eta); torch.cuda.synchronize()
  ms = ctx.timing_collect(1)[0]
  raw = ctx.wave_clock_collect()
  hw = (raw[:, 1] >> 40) & 0xffff
  xcc = (raw[:, 1] >> 56) & 0xf
  raw = raw.copy(); raw[:, 1] &= 0xffffffffff; raw[:, 0] &= 0xffffffffff
  st = raw.astype(np.float64) * 0.01   # 100 MHz ticks -> us
  simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
  cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
  uniq, cnt = np.unique(cuid, return_counts=True)
  print("   distinct CUs used: %d ; waves per CU: min %d med %d max %d ; histogram %s" % (len(uniq), cnt.min(), np.median(cnt), cnt.max(), np.bincount(cnt)))
  sid = cuid * 4 + simd
  u2, c2 = np.unique(sid, return_counts=True)
  print("   distinct SIMDs: %d ; waves per SIMD histogram %s" % (len(u2), np.bincount(c2)))
  t0 = st[:, 0].min()
  start, end = st[:, 0] - t0, st[:, 1] - t0
  dur = end - start
  print("N=%d kernel %.1f us (HIP events) | waves %d | start: max %.1f us | end: min %.1f med %.1f p90 %.1f max %.1f us | duration: min %.1f med %.1f max %.1f us" %
        (N, ms * 1e3, len(st), start.max(), end.min(), np.median(end), np.percentile(end, 90), end.max(), dur.min(), np.median(dur), dur.max()))
  # per CU (block = 4 waves): end of the block
  blk = end.reshape(-1, 4).max(axis=1)
  print("   per-workgroup end: min %.1f med %.1f max %.1f" % (blk.min(), np.median(blk), blk.max()))
  per_cu_end = {c: end[cuid == c].max() for c in uniq}
  for k in sorted(set(cnt)):
    sel = [per_cu_end[c] for c, n in zip(uniq, cnt) if n == k]
    print("   CUs with %2d waves: %3d CUs, last wave ends at med %.1f us (min %.1f max %.1f)" % (k, len(sel), np.median(sel), min(sel), max(sel)))
  ctx.close()
