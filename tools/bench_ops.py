"""Round-2 kernel table: the multi-block symmetric operations against what they replace, and the products that moved
onto the symmetric skeleton against their one-sided sweeps.  Device-resident vectors, HIP-event kernel time summed
per PRODUCT (a product made of several launches counts all of them), clocks primed first.

  python tools/bench_ops.py [N ...]      -> gpurun_out/r2_ops_table.json (RMB_AB_OUT names another file: profiles/r4_ops_table.json)
  RMB_AB_LIB=<other build of librmb_mobility.so> RMB_AB_OUT=<json>   time that build instead (same-box A/B of two
  builds: boxes differ by a few per cent, so a change is only priced against a baseline built from the previous
  commit and timed in the same call)
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rigidmultiblobswall_amd import _lib as _rmb_lib
if os.environ.get("RMB_AB_LIB"):
  _rmb_lib.LIB_PATH = os.path.abspath(os.environ["RMB_AB_LIB"])
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud

SIZES = [int(x) for x in sys.argv[1:]] or [10000, 100000]
rows = []


def timed(ctx, fn, reps, launches_per_product):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  ctx.timing_reset()
  for _ in range(reps):
    fn()
  torch.cuda.synchronize()
  if launches_per_product is None:          # however many bracketed launches a product takes (chunked passes)
    return float(np.sum(ctx.timing_collect(8192))) / reps
  ms = ctx.timing_collect(reps * launches_per_product)
  assert len(ms) == reps * launches_per_product, (len(ms), reps, launches_per_product)
  return float(np.sum(ms)) / reps


def row(N, name, path, ms, base_ms=None, note=None):
  d = dict(N=N, product=name, path=path, kernel_ms=round(ms, 4), gpairs_per_s=round(float(N) * N / (ms * 1e-3) / 1e9, 1))
  if base_ms is not None:
    d["vs_replaced"] = round(base_ms / ms, 3)
  if note:
    d["note"] = note
  rows.append(d)
  print(d, flush=True)


for N in SIZES:
  r, f, eta, a = d2_cloud(N)
  rng = np.random.RandomState(1)
  dev = lambda x: torch.as_tensor(np.ascontiguousarray(x).reshape(-1), device="cuda")
  rd, fd = dev(r), dev(f)
  vs = [dev(rng.randn(N, 3)) for _ in range(4)]
  td = vs[0]
  reps = 100 if N <= 20000 else 6
  ctx = MobilityContext(0)
  ctx.set_option("timing", 1)
  ctx.set_positions(rd, a, wall=True)
  # prime the clocks (tools/experiments/exp_prewarm.py: ~100 launches / 25 ms until the fp64 clock settles)
  for _ in range(300 if N <= 20000 else 3):
    ctx.matvec_device("tt", fd, eta)
  torch.cuda.synchronize()

  t = {}
  for kind in ("tt", "tr", "rt", "rr"):
    t[kind] = timed(ctx, lambda: ctx.matvec_device(kind, fd, eta), reps, 1)
    row(N, "wall " + kind, "sym2t_kernel (two target blobs per lane; sym_kernel / sym_coop_kernel with sym_two_targets = 0)", t[kind])
  ctx.set_option("precision", 32)
  row(N, "wall tt, single precision", "sym32_tt_kernel", timed(ctx, lambda: ctx.matvec_device("tt", fd, eta), reps, 1), t["tt"])
  ctx.set_option("precision", 64)
  ctx.set_option("symx_single", 1)
  row(N, "wall tt", "symx_kernel<OpSingle>", timed(ctx, lambda: ctx.matvec_device("tt", fd, eta), reps, 1), t["tt"])
  ctx.set_option("symx_single", 0)
  ctx.set_option("deterministic", 1)
  t["tt_sweep"] = timed(ctx, lambda: ctx.matvec_device("tt", fd, eta), reps, 1)
  row(N, "wall tt", "one-sided sweep (deterministic)", t["tt_sweep"])
  ctx.set_option("deterministic", 2)
  # the per-product time of mode 2 = sweep + ordered reduction (bracketed together), times the number of chunks
  det2 = timed(ctx, lambda: ctx.matvec_device("tt", fd, eta), reps, None)
  row(N, "wall tt", "deterministic symmetric (ordered reduction of per-unit partials)", det2, t["tt_sweep"])
  ctx.set_option("deterministic", 0)

  # fused row
  ctx.set_option("fused_symmetric", 2)
  two = timed(ctx, lambda: ctx.matvec_device("tt_tr", fd, eta, vec2=td), reps, 2)
  row(N, "wall tt+tr", "two symmetric passes (round 1)", two)
  ctx.set_option("fused_symmetric", 0)
  row(N, "wall tt+tr", "one-sided fused sweep", timed(ctx, lambda: ctx.matvec_device("tt_tr", fd, eta, vec2=td), reps, 1))
  ctx.set_option("fused_symmetric", 1)
  fused = timed(ctx, lambda: ctx.matvec_device("tt_tr", fd, eta, vec2=td), reps, 1)
  row(N, "wall tt+tr", "single symmetric pass (symx OpFusedRow)", fused, two)

  # grand product
  sep = two + t["rt"] + t["rr"]
  row(N, "wall grand [tt tr; rt rr]", "round 1: tt, tr, rt, rr passes (sum of the rows above)", sep)
  row(N, "wall grand [tt tr; rt rr]", "single symmetric pass (symx OpGrand)",
      timed(ctx, lambda: ctx.matvec_op_device("grand", (fd, td), eta), reps, 1), sep)
  row(N, "wall [tt; rt] f", "single symmetric pass (symx OpColumnF)",
      timed(ctx, lambda: ctx.matvec_op_device("force_column", (fd,), eta), reps, 1), t["tt"] + t["rt"])

  # single-precision twins of the multi-block operations (symx32_kernels.h)
  ctx.set_option("precision", 32)
  row(N, "wall tt+tr, single precision", "symx32_kernel<OpFusedRow32>",
      timed(ctx, lambda: ctx.matvec_device("tt_tr", fd, eta, vec2=td), reps, 1), fused)
  row(N, "wall grand [tt tr; rt rr], single precision", "symx32_kernel<OpGrand32>",
      timed(ctx, lambda: ctx.matvec_op_device("grand", (fd, td), eta), reps, 1))
  row(N, "wall rr, single precision", "symx32_kernel<OpSingle32>", timed(ctx, lambda: ctx.matvec_device("rr", fd, eta), reps, 1), t["rr"])
  ctx.set_option("precision", 64)

  # k vectors
  row(N, "wall tt x2", "sym2_kernel", timed(ctx, lambda: ctx.matvec2_device("tt", fd, td, eta), reps, 1), 2 * t["tt"])
  for k in (2, 3, 4):
    row(N, "wall tt x%d" % k, "symx OpTTk<%d>" % k,
        timed(ctx, lambda: ctx.matvec_op_device("tt_multi", [fd] + vs[:k - 1], eta), reps, 1), k * t["tt"])

  # in-plane
  for kind in ("tt", "tr"):
    ctx.set_option("deterministic", 1)
    sw = timed(ctx, lambda: ctx.matvec_device(kind, fd, eta, in_plane=True), reps, 1)
    ctx.set_option("deterministic", 0)
    row(N, "in-plane " + kind, "one-sided sweep", sw)
    row(N, "in-plane " + kind, "symmetric (symx OpSingle, in_plane)", timed(ctx, lambda: ctx.matvec_device(kind, fd, eta, in_plane=True), reps, 1), sw)

  # no wall: free surface, radii forces, forces
  ctx.set_positions(rd, a, wall=False)
  ctx.set_option("deterministic", 1)
  sw = timed(ctx, lambda: ctx.matvec_device("tt_free", fd, eta), reps, 1)
  ctx.set_option("deterministic", 0)
  row(N, "free surface tt", "one-sided sweep", sw)
  row(N, "free surface tt", "symmetric (symx OpFreeSurface)", timed(ctx, lambda: ctx.matvec_device("tt_free", fd, eta), reps, 1), sw)
  rad = dev(a * (0.5 + rng.rand(N)))
  ctx.set_option("deterministic", 1)
  sw = timed(ctx, lambda: ctx.blob_blob_force_radii_device(rad, 3.92, 0.1 * a), reps, 1)
  ctx.set_option("deterministic", 0)
  row(N, "forces, per-blob radii", "one-sided sweep", sw)
  row(N, "forces, per-blob radii", "symmetric (sym_force_kernel<RADII>)", timed(ctx, lambda: ctx.blob_blob_force_radii_device(rad, 3.92, 0.1 * a), reps, 1), sw)
  row(N, "forces", "symmetric", timed(ctx, lambda: ctx.blob_blob_force_device(3.92, 0.1 * a, a), reps, 1))
  # per-blob-radii mobility, sources == targets (the reference's radii_* modes), wall
  import ctypes
  from rigidmultiblobswall_amd import _lib
  lib = _lib.load()
  vp = lambda t: ctypes.c_void_p(t.data_ptr())
  rd2, rad2, outr = rd.clone(), rad.clone(), torch.empty_like(fd)
  L0 = np.zeros(3)
  Lp = ctypes.c_void_p(L0.ctypes.data)
  st = lambda tgt, radt: _lib.check(lib.rmb_mobility_source_target_device(ctx._h, N, vp(rd), vp(rad), N, vp(tgt), vp(radt), vp(fd), eta, Lp, 1, vp(outr)))
  sw = timed(ctx, lambda: st(rd2, rad2), reps, 1)
  row(N, "radii mobility, wall (sources == targets)", "one-sided source->target sweep", sw)
  row(N, "radii mobility, wall (sources == targets)", "symmetric (symx OpRadiiTT)", timed(ctx, lambda: st(rd, rad), reps, 1), sw)
  # Stokeslet pressure / Stokes double layer, N sources -> N targets (one-sided, atomic-free; aux_kernels.h)
  nrm, vv, ww = dev(rng.randn(N, 3)), dev(rng.randn(N, 3)), dev(0.1 + rng.rand(N))
  tg = dev(np.asarray(r) + 0.01)
  po, uo = torch.empty(N, dtype=torch.float64, device="cuda"), torch.empty(3 * N, dtype=torch.float64, device="cuda")
  for wall in (0, 1):
    row(N, "Stokeslet pressure, %s" % ("wall" if wall else "unbounded"), "aux_sweep_kernel",
        timed(ctx, lambda: _lib.check(lib.rmb_pressure_stokeslet_device(ctx._h, N, vp(rd), N, vp(tg), vp(fd), None, wall, vp(po))), reps, None))
    row(N, "Stokes double layer, %s" % ("wall" if wall else "unbounded"), "aux_sweep_kernel",
        timed(ctx, lambda: _lib.check(lib.rmb_double_layer_device(ctx._h, N, vp(rd), N, vp(tg), vp(nrm), vp(vv), vp(ww), wall, -1.0, vp(uo))), reps, None))
  row(N, "Stokes double layer, RPY blobs", "aux_sweep_kernel",
      timed(ctx, lambda: _lib.check(lib.rmb_double_layer_device(ctx._h, N, vp(rd), N, vp(tg), vp(nrm), vp(vv), vp(ww), 0, a, vp(uo))), reps, None))
  for kind in ("tt", "tr", "rt", "rr"):
    row(N, "no-wall " + kind, "sym_kernel", timed(ctx, lambda: ctx.matvec_device(kind, fd, eta), reps, 1))
  row(N, "no-wall grand", "single symmetric pass (symx OpGrand)", timed(ctx, lambda: ctx.matvec_op_device("grand", (fd, td), eta), reps, 1))
  ctx.close()

os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open(os.environ.get("RMB_AB_OUT", "gpurun_out/r2_ops_table.json"), "w"), indent=1)
