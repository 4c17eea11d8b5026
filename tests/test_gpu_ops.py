"""GPU parity of the multi-block symmetric operations (csrc/symx_kernels.h, rmb_matvec_op_device) and of the products
that moved onto the symmetric skeleton this round (in-plane, free surface, per-blob-radii forces), against the oracle.

Reference semantics: the fused row is mobility/mobility_pycuda.py:1266-1391 (K11) / :1394-1512 (K12); the grand product
is the four separate calls of quaternion_integrator/quaternion_integrator_rollers.py:1114-1121; in-plane
mobility_numba.py:291, :690; free surface :1770-1937; radii forces multi_bodies/forces_numba.py:73-122.
Tolerances as tests/test_gpu_parity.py (relative L2, 1e-12 D2 / 1e-10 D1).
"""
import os

import numpy as np
import pytest

from conftest import rel_err
from test_gpu_parity import d1_cloud, d2_cloud, TOL_D1, TOL_D2, TOL_SHARD

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
  import torch
  return torch


@pytest.fixture(scope="module")
def Ctx():
  from rigidmultiblobswall_amd import MobilityContext
  return MobilityContext


@pytest.fixture(scope="module")
def mob():
  from rigidmultiblobswall_amd import mobility
  return mobility


def _dev(torch, x):
  return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64).reshape(-1), device="cuda")


def _oracle_blocks(oracle, wall, r, f, t, eta, a, L=None, in_plane=False):
  kw = dict(periodic_length=np.zeros(3) if L is None else np.asarray(L, dtype=np.float64), in_plane=in_plane)
  W = lambda kind, v: oracle._wrapped(kind, int(wall), r, v, eta, a, **kw)
  return dict(tt_f=W("tt", f), tr_t=W("tr", t), rt_f=W("rt", f), rr_t=W("rr", t))


CASES = [
    # (N, wall, L, cloud)
    (130, True, None, "d2"), (1000, True, None, "d2"), (4097, True, None, "d2"), (1000, False, None, "d2"),
    (3000, True, None, "d1"), (700, True, (14.0, 16.0, 0.0), "d2"), (700, False, (9.0, 0.0, 11.0), "d2"),
]


@pytest.mark.parametrize("N,wall,L,cloud", CASES)
def test_ops_match_oracle_blocks(Ctx, oracle, torch_mod, N, wall, L, cloud):
  torch = torch_mod
  r, f, eta, a = (d2_cloud if cloud == "d2" else d1_cloud)(N, seed=N + 1)
  t = np.random.RandomState(N + 2).randn(N, 3)
  tol = TOL_D2 if cloud == "d2" else TOL_D1
  ref = _oracle_blocks(oracle, wall, r, f, t, eta, a, L)
  ctx = Ctx(0)
  ctx.set_positions(_dev(torch, r), a, L, wall=wall)
  fd, td = _dev(torch, f), _dev(torch, t)
  (u,) = ctx.matvec_op_device("velocity_from_force_torque", (fd, td), eta)
  assert ctx.last_launch()["chunks"] == 0            # symmetric path
  assert rel_err(u.cpu().numpy(), ref["tt_f"] + ref["tr_t"]) < tol
  u, w = ctx.matvec_op_device("grand", (fd, td), eta)
  assert rel_err(u.cpu().numpy(), ref["tt_f"] + ref["tr_t"]) < tol
  assert rel_err(w.cpu().numpy(), ref["rt_f"] + ref["rr_t"]) < tol
  u, w = ctx.matvec_op_device("force_column", (fd,), eta)
  assert rel_err(u.cpu().numpy(), ref["tt_f"]) < tol
  assert rel_err(w.cpu().numpy(), ref["rt_f"]) < tol
  # the same entry point as rmb_matvec_device(RMB_TT_TR) (single pass by default, two passes with option 2)
  u1 = ctx.matvec_device("tt_tr", fd, eta, vec2=td).cpu().numpy()
  ctx.set_option("fused_symmetric", 2)
  u2 = ctx.matvec_device("tt_tr", fd, eta, vec2=td).cpu().numpy()
  ctx.set_option("fused_symmetric", 0)
  u0 = ctx.matvec_device("tt_tr", fd, eta, vec2=td).cpu().numpy()
  for x in (u1, u2, u0):
    assert rel_err(x, ref["tt_f"] + ref["tr_t"]) < tol
  ctx.close()


@pytest.mark.parametrize("k", [1, 2, 3, 4])
@pytest.mark.parametrize("N,wall,L", [(1000, True, None), (257, False, None), (500, True, (12.0, 0.0, 0.0))])
def test_tt_multi(Ctx, oracle, torch_mod, k, N, wall, L):
  torch = torch_mod
  r, _, eta, a = d2_cloud(N, seed=5 * N + k)
  vs = [np.random.RandomState(100 * k + v).randn(N, 3) for v in range(k)]
  ctx = Ctx(0)
  ctx.set_positions(_dev(torch, r), a, L, wall=wall)
  outs = ctx.matvec_op_device("tt_multi", [_dev(torch, v) for v in vs], eta)
  assert len(outs) == k
  kw = dict(periodic_length=np.zeros(3) if L is None else np.asarray(L, dtype=np.float64))
  for v, o in zip(vs, outs):
    assert rel_err(o.cpu().numpy(), oracle._wrapped("tt", int(wall), r, v, eta, a, **kw)) < TOL_D2
  ctx.close()


@pytest.mark.parametrize("kind", ["tr", "rt", "rr"])
@pytest.mark.parametrize("k", [1, 2, 3, 4])
@pytest.mark.parametrize("wall,L", [(True, None), (False, None), (True, (0.0, 13.0, 0.0))])
def test_kind_multi(Ctx, oracle, torch_mod, kind, k, wall, L):
  """tr / rt / rr applied to k vectors in one pass (RMB_OP_TR_MULTI / RT / RR), incl. two pair shards."""
  torch = torch_mod
  N = 700
  r, _, eta, a = d2_cloud(N, seed=3 * k + 1)
  vs = [np.random.RandomState(10 * k + v).randn(N, 3) for v in range(k)]
  ctx = Ctx(0)
  ctx.set_positions(_dev(torch, r), a, L, wall=wall)
  dv = [_dev(torch, v) for v in vs]
  outs = ctx.matvec_op_device(kind + "_multi", dv, eta)
  parts = [ctx.matvec_op_device(kind + "_multi", dv, eta, shard=s, nshards=2) for s in range(2)]
  kw = dict(periodic_length=np.zeros(3) if L is None else np.asarray(L, dtype=np.float64))
  for c, (v, o) in enumerate(zip(vs, outs)):
    assert rel_err(o.cpu().numpy(), oracle._wrapped(kind, int(wall), r, v, eta, a, **kw)) < TOL_D2
    assert rel_err((parts[0][c] + parts[1][c]).cpu().numpy(), o.cpu().numpy()) < TOL_SHARD
  ctx.close()


@pytest.mark.parametrize("kind", ["tt", "tr", "rt", "rr"])
@pytest.mark.parametrize("wall,L", [(True, None), (False, None), (True, (14.0, 16.0, 0.0))])
def test_single_kinds_through_generic_skeleton(Ctx, oracle, torch_mod, kind, wall, L):
  """OpSingle<KIND> in symx_kernel == sym_kernel<KIND> (option symx_single): same pair function, other skeleton."""
  torch = torch_mod
  r, v, eta, a = d2_cloud(900, seed=17)
  ctx = Ctx(0)
  ctx.set_option("symx_single", 1)
  ctx.set_positions(_dev(torch, r), a, L, wall=wall)
  u = ctx.matvec_device(kind, _dev(torch, v), eta).cpu().numpy()
  assert ctx.last_launch()["chunks"] == 0
  kw = dict(periodic_length=np.zeros(3) if L is None else np.asarray(L, dtype=np.float64))
  assert rel_err(u, oracle._wrapped(kind, int(wall), r, v, eta, a, **kw)) < TOL_D2
  ctx.close()


@pytest.mark.parametrize("N", [128, 1000, 2500])
@pytest.mark.parametrize("L", [None, (13.0, 12.0, 0.0)])
def test_in_plane_free_surface_on_symmetric_path(Ctx, oracle, torch_mod, N, L):
  torch = torch_mod
  r, v, eta, a = d2_cloud(N, seed=N + 40)
  t = np.random.RandomState(N).randn(N, 3)
  kw = dict(periodic_length=np.zeros(3) if L is None else np.asarray(L, dtype=np.float64))
  ctx = Ctx(0)
  ctx.set_positions(_dev(torch, r), a, L, wall=True)
  for kind, ref_fn in (("tt", oracle.in_plane_mobility_trans_times_force_oracle),
                       ("tr", oracle.in_plane_mobility_trans_times_torque_oracle)):
    u = ctx.matvec_device(kind, _dev(torch, v), eta, in_plane=True).cpu().numpy()
    assert ctx.last_launch()["chunks"] == 0, "in-plane %s fell back to the one-sided sweep" % kind
    assert np.all(u.reshape(-1, 3)[:, 2] == 0.0)
    assert rel_err(u, ref_fn(r, v, eta, a, **kw)) < TOL_D2
  # fused in-plane row
  (u,) = ctx.matvec_op_device("velocity_from_force_torque", (_dev(torch, v), _dev(torch, t)), eta, in_plane=True)
  ref = (oracle.in_plane_mobility_trans_times_force_oracle(r, v, eta, a, **kw) +
         oracle.in_plane_mobility_trans_times_torque_oracle(r, t, eta, a, **kw))
  assert rel_err(u.cpu().numpy(), ref) < TOL_D2
  # free surface: raw heights
  ctx.set_positions(_dev(torch, r), a, L, wall=False)
  u = ctx.matvec_device("tt_free", _dev(torch, v), eta).cpu().numpy()
  assert ctx.last_launch()["chunks"] == 0, "free surface fell back to the one-sided sweep"
  assert rel_err(u, oracle.free_surface_mobility_trans_times_force_oracle(r, v, eta, a, **kw)) < TOL_D2
  ctx.set_option("deterministic", 1)
  u_sweep = ctx.matvec_device("tt_free", _dev(torch, v), eta).cpu().numpy()
  assert ctx.last_launch()["chunks"] >= 1
  assert rel_err(u, u_sweep) < TOL_D2
  ctx.close()


def test_free_surface_overlapping_blobs(Ctx, oracle, torch_mod):
  """Near-field branch of both the direct and the image RPY evaluation (blobs closer than 2a to each other and to
  their own / each other's mirror image)."""
  torch = torch_mod
  rng = np.random.RandomState(3)
  N, a, eta = 600, 0.4, 1.3
  r = rng.rand(N, 3) * np.array([4.0, 4.0, 1.0])
  r[:, 2] += 0.05
  f = rng.randn(N, 3)
  ctx = Ctx(0)
  ctx.set_positions(_dev(torch, r), a, None, wall=False)
  u = ctx.matvec_device("tt_free", _dev(torch, f), eta).cpu().numpy()
  assert ctx.last_launch()["chunks"] == 0
  assert rel_err(u, oracle.free_surface_mobility_trans_times_force_oracle(r, f, eta, a)) < TOL_D1
  ctx.close()


@pytest.mark.parametrize("N", [128, 777, 3000])
@pytest.mark.parametrize("L", [None, (6.0, 7.0, 0.0), (8.0, 8.0, 8.0)])
def test_radii_forces_on_symmetric_path(Ctx, oracle, torch_mod, N, L):
  torch = torch_mod
  rng = np.random.RandomState(N)
  r = rng.rand(N, 3) * (N / 40.0) ** (1.0 / 3.0) * 2.0
  radii = 0.1 + 0.4 * rng.rand(N)
  eps, b = 0.7, 0.13
  kw = dict(periodic_length=np.zeros(3) if L is None else np.asarray(L, dtype=np.float64), repulsion_strength=eps,
            debye_length=b)
  ctx = Ctx(0)
  ctx.set_positions(_dev(torch, r), 0.25, L, wall=False)
  F = ctx.blob_blob_force_radii_device(_dev(torch, radii), eps, b).cpu().numpy().reshape(N, 3)
  assert ctx.last_launch()["chunks"] == 0, "radii forces fell back to the one-sided sweep"
  assert rel_err(F, oracle.calc_blob_blob_forces_radii_oracle(r, radii, **kw)) < TOL_D2
  ctx.close()


@pytest.mark.parametrize("wall,z_special", [(True, 0.7), (False, 1.0)])
@pytest.mark.parametrize("path", ["sym", "sym2", "symx"])
def test_padding_sentinels_with_power_of_two_box(Ctx, oracle, torch_mod, wall, z_special, path):
  """Padded lanes of the last tile (N % 64 != 0) carry +-1e100 coordinates; wrapped by a power-of-two box they
  landed exactly on a real blob whose effective height is 1.0 (rsqrt(0) -> NaN).  a = 1: wall-clamped blobs
  (z <= a) have z_eff = 1.0; without a wall some blobs sit at z = 1.0 exactly."""
  torch = torch_mod
  N, a, eta = 200, 1.0, 1.0
  rng = np.random.RandomState(11)
  r = rng.rand(N, 3) * np.array([8.0, 8.0, 6.0])
  r[:, 2] += 1.2
  r[::5, 2] = z_special
  v = rng.randn(N, 3)
  v2 = rng.randn(N, 3)
  L = np.array([8.0, 8.0, 0.0])
  ctx = Ctx(0)
  ctx.set_positions(_dev(torch, r), a, L, wall=wall)
  ref = oracle._wrapped("tt", int(wall), r, v, eta, a, periodic_length=L)
  if path == "sym":
    u = ctx.matvec_device("tt", _dev(torch, v), eta).cpu().numpy()
  elif path == "sym2":
    ua, ub = ctx.matvec2_device("tt", _dev(torch, v), _dev(torch, v2), eta)
    u = ua.cpu().numpy()
    assert np.all(np.isfinite(ub.cpu().numpy()))
  else:
    u, w = ctx.matvec_op_device("force_column", (_dev(torch, v),), eta)
    assert np.all(np.isfinite(w.cpu().numpy()))
    u = u.cpu().numpy()
  assert ctx.last_launch()["chunks"] == 0
  assert np.all(np.isfinite(u))
  assert rel_err(u, ref) < TOL_D1
  ctx.close()


@pytest.mark.parametrize("op,n_in", [("velocity_from_force_torque", 2), ("grand", 2), ("force_column", 1), ("tt_multi", 3)])
@pytest.mark.parametrize("N,L", [(1000, None), (90, None), (24, None), (400, (11.0, 12.0, 0.0))])
def test_op_pair_shards_sum_to_full_product(Ctx, torch_mod, op, n_in, N, L):
  """What G ranks compute (pair shard g of G into a full-length partial) summed the way all_reduce will."""
  torch = torch_mod
  r, _, eta, a = d2_cloud(N, seed=N)
  vs = [_dev(torch, np.random.RandomState(7 + v).randn(N, 3)) for v in range(n_in)]
  ctx = Ctx(0)
  ctx.set_positions(_dev(torch, r), a, L, wall=True)
  full = ctx.matvec_op_device(op, vs, eta)
  for G in (2, 3, 8):
    parts = [ctx.matvec_op_device(op, vs, eta, shard=g, nshards=G) for g in range(G)]
    for c in range(len(full)):
      total = torch.stack([p[c] for p in parts]).sum(0)
      assert rel_err(total.cpu().numpy(), full[c].cpu().numpy()) < TOL_SHARD
  ctx.close()


def test_ops_fall_back_below_the_symmetric_threshold_and_in_deterministic_mode(Ctx, oracle, torch_mod):
  torch = torch_mod
  for N, det in ((40, 0), (600, 1)):
    r, f, eta, a = d2_cloud(N, seed=3)
    t = np.random.RandomState(4).randn(N, 3)
    ref = _oracle_blocks(oracle, True, r, f, t, eta, a)
    ctx = Ctx(0)
    ctx.set_option("deterministic", det)
    ctx.set_positions(_dev(torch, r), a, None, wall=True)
    u, w = ctx.matvec_op_device("grand", (_dev(torch, f), _dev(torch, t)), eta)
    assert ctx.last_launch()["chunks"] >= 1           # one-sided sweeps
    assert rel_err(u.cpu().numpy(), ref["tt_f"] + ref["tr_t"]) < TOL_D2
    assert rel_err(w.cpu().numpy(), ref["rt_f"] + ref["rr_t"]) < TOL_D2
    u, w = ctx.matvec_op_device("force_column", (_dev(torch, f),), eta)
    assert rel_err(u.cpu().numpy(), ref["tt_f"]) < TOL_D2 and rel_err(w.cpu().numpy(), ref["rt_f"]) < TOL_D2
    ctx.close()


def test_two_vector_pair_shards_for_small_suspensions(Ctx, torch_mod):
  """rmb_matvec2_pairshard_device refused nshards > 1 for n < 128 while the one-vector entry accepted it."""
  torch = torch_mod
  r, f, eta, a = d2_cloud(24, seed=1)
  g = np.random.RandomState(2).randn(24, 3)
  ctx = Ctx(0)
  ctx.set_positions(_dev(torch, r), a, None, wall=True)
  fa, fb = _dev(torch, f), _dev(torch, g)
  ua, ub = ctx.matvec2_device("tt", fa, fb, eta)
  pa = [ctx.matvec2_device("tt", fa, fb, eta, shard=s, nshards=2) for s in range(2)]
  assert rel_err((pa[0][0] + pa[1][0]).cpu().numpy(), ua.cpu().numpy()) < TOL_SHARD
  assert rel_err((pa[0][1] + pa[1][1]).cpu().numpy(), ub.cpu().numpy()) < TOL_SHARD
  ctx.close()


def test_stream_switch_is_ordered(Ctx, oracle, torch_mod):
  """A context follows torch's current stream; switching streams between calls must not let the new stream's sweep
  overtake the finalize / memset still queued on the old one (rmb_ctx_set_stream waits on an event)."""
  torch = torch_mod
  r, f, eta, a = d2_cloud(3000, seed=9)
  ref = oracle.single_wall_mobility_trans_times_force_oracle(r, f, eta, a)
  ctx = Ctx(0)
  rd, fd = _dev(torch, r), _dev(torch, f)
  ctx.set_positions(rd, a, None, wall=True)
  streams = [torch.cuda.Stream() for _ in range(3)]
  torch.cuda.synchronize()
  outs = []
  for it in range(12):
    with torch.cuda.stream(streams[it % 3]):
      outs.append(ctx.matvec_device("tt", fd, eta))
  torch.cuda.synchronize()
  for o in outs:
    assert rel_err(o.cpu().numpy(), ref) < TOL_D2
  ctx.close()


@pytest.mark.parametrize("N,wall,L", [(130, True, None), (1000, True, None), (4097, True, None), (1500, False, None),
                                       (700, True, (14.0, 16.0, 0.0))])
def test_deterministic_symmetric_mode_is_bit_reproducible_and_correct(Ctx, oracle, torch_mod, N, wall, L):
  """`deterministic = 2`: the symmetric pass with per-unit partials reduced in a fixed order.  Bit-identical across
  repeated launches and across fresh contexts (the default atomic path agrees only to rounding), equal to the oracle,
  for single kinds, the fused row, the grand product and the two-vector product; also with a workspace so small that
  the unit list is processed in many chunks."""
  torch = torch_mod
  r, f, eta, a = d2_cloud(N, seed=N + 7)
  t = np.random.RandomState(N + 8).randn(N, 3)
  ref = _oracle_blocks(oracle, wall, r, f, t, eta, a, L)
  fd, td = _dev(torch, f), _dev(torch, t)

  def run(workspace_mb=None):
    ctx = Ctx(0)
    ctx.set_option("deterministic", 2)
    if workspace_mb is not None:
      ctx.set_option("det_workspace_mb", workspace_mb)
    ctx.set_positions(_dev(torch, r), a, L, wall=wall)
    res = {}
    for kind in ("tt", "tr", "rt", "rr"):
      res[kind] = ctx.matvec_device(kind, fd, eta).cpu().numpy()
      assert ctx.last_launch()["chunks"] == 0                  # still the symmetric path
    res["fused"] = ctx.matvec_device("tt_tr", fd, eta, vec2=td).cpu().numpy()
    u, w = ctx.matvec_op_device("grand", (fd, td), eta)
    res["grand_u"], res["grand_w"] = u.cpu().numpy(), w.cpu().numpy()
    ua, ub = ctx.matvec2_device("tt", fd, td, eta)
    res["two_a"], res["two_b"] = ua.cpu().numpy(), ub.cpu().numpy()
    res["tt_again"] = ctx.matvec_device("tt", fd, eta).cpu().numpy()
    ctx.close()
    return res

  a1, a2 = run(), run()
  for k in a1:
    assert np.array_equal(a1[k], a2[k]), k                      # bit-identical across contexts
  assert np.array_equal(a1["tt"], a1["tt_again"])               # and across launches
  tol = TOL_D2
  assert rel_err(a1["tt"], ref["tt_f"]) < tol and rel_err(a1["tr"], oracle._wrapped("tr", int(wall), r, f, eta, a,
      periodic_length=np.zeros(3) if L is None else np.asarray(L, dtype=np.float64))) < tol
  assert rel_err(a1["fused"], ref["tt_f"] + ref["tr_t"]) < tol
  assert rel_err(a1["grand_u"], ref["tt_f"] + ref["tr_t"]) < tol and rel_err(a1["grand_w"], ref["rt_f"] + ref["rr_t"]) < tol
  assert rel_err(a1["two_a"], ref["tt_f"]) < tol
  # many chunks (1 MB workspace): same mathematics, still correct and reproducible
  b1, b2 = run(workspace_mb=1), run(workspace_mb=1)
  for k in b1:
    assert np.array_equal(b1[k], b2[k]), k
    assert rel_err(b1[k], a1[k]) < 1e-13, k


@pytest.mark.parametrize("N", [128, 700, 3000])
@pytest.mark.parametrize("wall", [1, 0])
@pytest.mark.parametrize("L", [None, (9.0, 10.0, 0.0)])
def test_radii_mobility_sources_equal_targets_is_symmetric_pass(Ctx, oracle, torch_mod, N, wall, L):
  """`radii_*` mobility modes call the source->target product with the same blobs on both sides
  (mobility/mobility.py:1369-1374); that operator is symmetric and runs on the symmetric skeleton (OpRadiiTT):
  overlapping blobs of very different radii (all three Zuk regimes), blobs below their own radius (per-blob clamp +
  B), pseudo-periodic images; against the oracle and against the one-sided source->target sweep."""
  import ctypes
  from rigidmultiblobswall_amd import _lib, mobility as mob
  torch = torch_mod
  rng = np.random.RandomState(N + wall)
  box = (N ** (1.0 / 3.0)) * 0.9
  r = rng.rand(N, 3) * box
  rad = 0.05 + 0.6 * rng.rand(N) ** 2
  f = rng.randn(N, 3)
  eta = 0.8
  Lv = np.zeros(3) if L is None else np.asarray(L, dtype=np.float64)
  pre = "single_wall" if wall else "no_wall"
  ref = getattr(oracle, pre + "_mobility_trans_times_force_source_target_oracle")(r, r, f, rad, rad, eta, periodic_length=Lv)
  # host surface: same arrays on both sides
  fn = getattr(mob, pre + "_mobility_trans_times_force_source_target_hip")
  u = mob.mobility_radii_trans_times_force(r, f, eta, 0.3, rad, fn, periodic_length=Lv)
  assert rel_err(u, ref) < TOL_D1, rel_err(u, ref)
  # equal contents in different arrays take the same path
  u_b = fn(r.copy(), r, f, rad.copy(), rad, eta, periodic_length=Lv)
  assert rel_err(u_b, u) < 1e-13
  # device entry: symmetric path (chunks == 0) when the pointers coincide, one-sided sweep otherwise; same numbers
  ctx = Ctx(0)
  lib = _lib.load()
  rd, radd, fd = _dev(torch, r), _dev(torch, rad), _dev(torch, f)
  rd2, radd2 = rd.clone(), radd.clone()
  out = torch.empty(3 * N, dtype=torch.float64, device="cuda")
  out2 = torch.empty_like(out)
  vp = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
  Lp = ctypes.c_void_p(Lv.ctypes.data)
  _lib.check(lib.rmb_mobility_source_target_device(ctx._h, N, vp(rd), vp(radd), N, vp(rd), vp(radd), vp(fd), eta, Lp, wall, vp(out)))
  assert ctx.last_launch()["chunks"] == 0
  _lib.check(lib.rmb_mobility_source_target_device(ctx._h, N, vp(rd), vp(radd), N, vp(rd2), vp(radd2), vp(fd), eta, Lp, wall, vp(out2)))
  assert ctx.last_launch()["chunks"] >= 1
  assert rel_err(out.cpu().numpy(), ref) < TOL_D1
  assert rel_err(out.cpu().numpy(), out2.cpu().numpy()) < 1e-12
  ctx.close()


# ---------------------------------------------------------------------------------------------
# single-precision mode of the tt product (the reference's `precision = 'single'` build)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("wall", [True, False])
@pytest.mark.parametrize("N", [128, 1000, 10000, 20011])
def test_single_precision_tt(Ctx, oracle, wall, N):
  """fp32 pair arithmetic, fp64 accumulation: relative L2 error 1e-5 against the fp64 oracle (measured ~2e-7 on the
  5 % cloud, ~1e-6 on the dense overlapping one); the deterministic sweep stays fp64."""
  r, f, eta, a = d1_cloud(N, seed=N) if N == 1000 else d2_cloud(N, seed=N)
  ctx = Ctx(0)
  try:
    ctx.set_positions(r, a, np.zeros(3), wall=wall)
    u64 = ctx.matvec("tt", f, eta)
    ctx.set_option("precision", 32)
    u32 = ctx.matvec("tt", f, eta)
    ref = getattr(oracle, ("single_wall" if wall else "no_wall") + "_mobility_trans_times_force_oracle")(r, f, eta, a)
    e32, e64 = rel_err(u32, ref), rel_err(u64, ref)
    assert np.all(np.isfinite(u32))
    assert e64 < 1e-10 and 1e-9 < e32 < 1e-5, (e32, e64)           # the fp32 kernel ran, and is single-precision accurate
    # pseudo-periodic domains and the deterministic sweep stay fp64 too
    ctx.set_option("deterministic", 1)
    assert rel_err(ctx.matvec("tt", f, eta), ref) < 1e-10
    ctx.set_option("deterministic", 0)
    ctx.set_option("precision", 64)
    assert rel_err(ctx.matvec("tt", f, eta), ref) < 1e-10
    with pytest.raises(Exception):
      ctx.set_option("precision", 16)
  finally:
    ctx.close()


def test_single_precision_switch_of_the_python_surface(mob, oracle):
  r, f, eta, a = d2_cloud(3000, seed=3)
  ref = oracle.single_wall_mobility_trans_times_force_oracle(r, f, eta, a)
  try:
    mob.precision = 'single'
    e = rel_err(mob.single_wall_mobility_trans_times_force_hip(r, f, eta, a), ref)
    assert 1e-9 < e < 1e-5, e
    mob.precision = 'half'
    with pytest.raises(ValueError):
      mob.single_wall_mobility_trans_times_force_hip(r, f, eta, a)
  finally:
    mob.precision = 'double'
  assert rel_err(mob.single_wall_mobility_trans_times_force_hip(r, f, eta, a), ref) < 1e-12


@pytest.mark.parametrize("wall", [True, False])
@pytest.mark.parametrize("N", [128, 1000, 6000])
def test_single_precision_other_products(Ctx, oracle, torch_mod, wall, N):
  """precision = 32 for tr / rt / rr, the fused row, the grand mobility, the force column, k-vector and in-plane products:
  single-precision accurate against the fp64 ORACLE (not against its own fp64 twin, with which it shares the algebra);
  pseudo-periodic domains stay fp64."""
  torch = torch_mod
  r, f, eta, a = d1_cloud(N, seed=N) if N == 1000 else d2_cloud(N, seed=N)
  t = np.random.RandomState(N + 7).randn(*f.shape)
  fd, td = _dev(torch, f), _dev(torch, t)
  W = lambda kind, v, ip=False: oracle._wrapped(kind, int(wall), r, v, eta, a, periodic_length=np.zeros(3), in_plane=ip)
  o = dict(tt_f=W("tt", f), tt_t=W("tt", t), tr_f=W("tr", f), tr_t=W("tr", t), rt_f=W("rt", f), rr_f=W("rr", f), rr_t=W("rr", t))
  ref = {"tr": o["tr_f"], "rt": o["rt_f"], "rr": o["rr_f"], "fused": o["tt_f"] + o["tr_t"],
         "grand_u": o["tt_f"] + o["tr_t"], "grand_w": o["rt_f"] + o["rr_t"], "col_u": o["tt_f"], "col_w": o["rt_f"],
         "tt3_a": o["tt_f"], "tt3_b": o["tt_t"], "tt3_c": o["tt_f"] + o["tt_t"], "rr2_a": o["rr_f"], "rr2_b": o["rr_t"],
         "mv2_a": o["tt_f"], "mv2_b": o["tt_t"]}
  if wall:
    ref["in_plane_tt"], ref["in_plane_tr"] = W("tt", f, True), W("tr", f, True)
  else:
    ref["free_surface"] = oracle.free_surface_mobility_trans_times_force_oracle(r, f, eta, a)
  ctx = Ctx(0)
  try:
    ctx.set_positions(r, a, np.zeros(3), wall=wall)

    def products():
      out = {k: ctx.matvec_device(k, fd, eta).cpu().numpy() for k in ("tr", "rt", "rr")}
      out["fused"] = ctx.matvec_device("tt_tr", fd, eta, vec2=td).cpu().numpy()
      g = ctx.matvec_op_device("grand", (fd, td), eta)
      out["grand_u"], out["grand_w"] = g[0].cpu().numpy(), g[1].cpu().numpy()
      c = ctx.matvec_op_device("force_column", (fd,), eta)
      out["col_u"], out["col_w"] = c[0].cpu().numpy(), c[1].cpu().numpy()
      m = ctx.matvec_op_device("tt_multi", (fd, td, fd + td), eta)
      out["tt3_a"], out["tt3_b"], out["tt3_c"] = (x.cpu().numpy() for x in m)
      m = ctx.matvec_op_device("rr_multi", (fd, td), eta)
      out["rr2_a"], out["rr2_b"] = (x.cpu().numpy() for x in m)
      m = ctx.matvec2_device("tt", fd, td, eta)
      out["mv2_a"], out["mv2_b"] = (x.cpu().numpy() for x in m)
      if wall:
        out["in_plane_tt"] = ctx.matvec_device("tt", fd, eta, in_plane=True).cpu().numpy()
        out["in_plane_tr"] = ctx.matvec_device("tr", fd, eta, in_plane=True).cpu().numpy()
      else:
        out["free_surface"] = ctx.matvec_device("tt_free", fd, eta).cpu().numpy()
      return out

    p64 = products()
    assert set(p64) == set(ref)
    for k in p64:
      assert rel_err(p64[k], ref[k]) < (TOL_D1 if N == 1000 else TOL_D2), k
    ctx.set_option("precision", 32)
    assert ctx.get_option("precision") == 32
    p32 = products()
    for k in p64:
      e = rel_err(p32[k], ref[k])
      assert np.all(np.isfinite(p32[k])) and 1e-9 < e < 2e-5, (k, e)      # the fp32 kernel ran and is single-precision accurate
    ctx.set_option("precision", 64)
    p64b = products()
    for k in p64:
      assert rel_err(p64b[k], p64[k]) < 1e-13, k
    # pseudo-periodic: the option is ignored (fp64)
    ctx.set_positions(r, a, np.array([0.0, 40.0 * a * (N / 1000.0) ** (1 / 3.0), 0.0]), wall=wall)
    ref = ctx.matvec_device("tr", fd, eta).cpu().numpy()
    ctx.set_option("precision", 32)
    assert rel_err(ctx.matvec_device("tr", fd, eta).cpu().numpy(), ref) < 1e-13
  finally:
    ctx.close()


@pytest.mark.parametrize("N", [200, 5000])
def test_single_precision_forces(Ctx, oracle, N):
  """precision = 32: the blob-blob forces in fp32 (what the reference's GPU force kernel computes in), uniform and
  per-blob radii; single-precision accurate against the fp64 oracle, fp64 again after switching back."""
  r, f, eta, a = d2_cloud(N, seed=N + 1)
  eps, b = 0.7, 0.15 * a
  rad = a * (0.6 + 0.8 * np.random.RandomState(N).rand(N))
  ref = oracle.calc_blob_blob_forces_oracle(r, repulsion_strength=eps, debye_length=b, blob_radius=a, periodic_length=np.zeros(3))
  ref_r = oracle.calc_blob_blob_forces_radii_oracle(r, rad, repulsion_strength=eps, debye_length=b, periodic_length=np.zeros(3))
  ctx = Ctx(0)
  try:
    ctx.set_positions(r, a, np.zeros(3), wall=False)
    ctx.set_option("precision", 32)
    F = ctx.blob_blob_force(eps, b, a)
    e = rel_err(F.reshape(-1), ref.reshape(-1))
    assert np.all(np.isfinite(F)) and 1e-9 < e < 1e-5, e
    Fr = ctx.blob_blob_force_radii(rad, eps, b)
    er = rel_err(Fr.reshape(-1), ref_r.reshape(-1))
    assert 1e-9 < er < 1e-5, er
    ctx.set_option("precision", 64)
    assert rel_err(ctx.blob_blob_force(eps, b, a).reshape(-1), ref.reshape(-1)) < 1e-12
  finally:
    ctx.close()


@pytest.mark.parametrize("N", [130, 3000])
def test_single_precision_pair_shards_sum_to_the_product(Ctx, torch_mod, N):
  """Multi-GPU layout in single precision: the shards of the unordered pairs (tt kernel, fused row, grand mobility) still
  partition the work exactly -- their sum is the unsharded single-precision product up to fp32 summation order."""
  torch = torch_mod
  r, f, eta, a = d2_cloud(N, seed=N + 2)
  t = np.random.RandomState(N).randn(*f.shape)
  fd, td = _dev(torch, f), _dev(torch, t)
  ctx = Ctx(0)
  try:
    ctx.set_positions(r, a, np.zeros(3), wall=True)
    ctx.set_option("precision", 32)
    G = 3
    whole = ctx.matvec_device("tt", fd, eta)
    parts = sum(ctx.matvec_pairshard_device("tt", fd, eta, g, G) for g in range(G))
    assert rel_err(parts.cpu().numpy(), whole.cpu().numpy()) < 1e-6
    for op, vecs in (("velocity_from_force_torque", (fd, td)), ("grand", (fd, td))):
      w = ctx.matvec_op_device(op, vecs, eta)
      p = [ctx.matvec_op_device(op, vecs, eta, shard=g, nshards=G) for g in range(G)]
      for c in range(len(w)):
        assert rel_err(sum(x[c] for x in p).cpu().numpy(), w[c].cpu().numpy()) < 1e-6, (op, c)
    # and they are single-precision results: close to, but not equal to, the double-precision product
    ctx.set_option("precision", 64)
    e = rel_err(whole.cpu().numpy(), ctx.matvec_device("tt", fd, eta).cpu().numpy())
    assert 1e-9 < e < 1e-5, e
  finally:
    ctx.close()


def test_single_precision_is_insensitive_to_the_size_of_the_domain(Ctx, oracle):
  """The float kernels subtract positions through a head / tail split of the fp64 coordinates, so a cloud sitting 4e5
  radii from the origin (where a float coordinate resolves 0.03 radii) is as accurate as the same cloud at the origin.
  (The reference's float build subtracts rounded coordinates and would be off by 10 % here.)"""
  N = 2000
  r, f, eta, a = d2_cloud(N, seed=21)
  shift = np.array([4.0e5 * a, -2.5e5 * a, 0.0])
  eps, b = 0.7, 0.15 * a
  for origin in (np.zeros(3), shift):
    rr = r + origin
    ctx = Ctx(0)
    try:
      ctx.set_positions(rr, a, np.zeros(3), wall=True)
      ctx.set_option("precision", 32)
      u = ctx.matvec("tt", f, eta)
      w = ctx.matvec("rr", f, eta)
      ref_u = oracle.single_wall_mobility_trans_times_force_oracle(rr, f, eta, a)
      ref_w = oracle.single_wall_mobility_rot_times_torque_oracle(rr, f, eta, a)
      assert rel_err(u, ref_u) < 2e-6 and rel_err(w, ref_w) < 2e-6, (origin, rel_err(u, ref_u), rel_err(w, ref_w))
      F = ctx.blob_blob_force(eps, b, a)
      ref_F = oracle.calc_blob_blob_forces_oracle(rr, repulsion_strength=eps, debye_length=b, blob_radius=a, periodic_length=np.zeros(3))
      assert rel_err(F.reshape(-1), ref_F.reshape(-1)) < 1e-5, (origin, rel_err(F.reshape(-1), ref_F.reshape(-1)))
    finally:
      ctx.close()


# ---------------------------------------------------------------------------------------------
# round 3: options that interact (advisor findings of round 2) and bit-reproducible pair shards
# ---------------------------------------------------------------------------------------------
def test_release_library_does_not_know_the_wrong_result_diagnostics(Ctx, torch_mod):
  """"skip_pairs" (results wrong by design) and "wave_clock" (changes which kernel runs) are not in the option table of the
  release library -- only of the diagnostics build tools/ load with RMB_DIAGNOSTICS=1: the boundary
  cannot be talked into a silent wrong answer."""
  torch = torch_mod
  from rigidmultiblobswall_amd._lib import RmbError
  r, f, eta, a = d2_cloud(1000, seed=5)
  fd = _dev(torch, f)
  ctx = Ctx(0)
  try:
    assert ctx.get_option("diagnostics_build") == 0
    ctx.set_positions(r, a, np.zeros(3), wall=True)
    ref = ctx.matvec_device("tt", fd, eta)
    for key in ("wave_clock", "skip_pairs"):
      with pytest.raises(RmbError, match="diagnostics build"):
        ctx.set_option(key, 1)
    u = ctx.matvec_device("tt", fd, eta)
    assert float((u - ref).abs().max()) <= 1e-12 * float(ref.abs().max())
    assert ctx.get_option("skip_pairs") == 0 and ctx.get_option("sym_oversub") == 8
    with pytest.raises(RmbError):
      ctx.get_option("no_such_option")
  finally:
    ctx.close()


def test_force_precision_option_pins_the_force_kernel(Ctx, oracle):
  """"force_precision": 0 follows "precision", 32 / 64 pin the blob-blob force kernel whatever the products run in."""
  N = 3000
  r, _, _, a = d2_cloud(N, seed=17)
  eps, b = 0.3, 0.2 * a
  ref = oracle.calc_blob_blob_forces_oracle(r, periodic_length=np.zeros(3), repulsion_strength=eps, debye_length=b, blob_radius=a)
  ctx = Ctx(0)
  try:
    ctx.set_positions(r, a, np.zeros(3), wall=False)
    err = {}
    for prec, fprec in ((64, 0), (32, 0), (32, 64), (64, 32), (64, 64)):
      ctx.set_option("precision", prec); ctx.set_option("force_precision", fprec)
      err[(prec, fprec)] = rel_err(ctx.blob_blob_force(eps, b, a), ref)
    assert err[(64, 0)] < 1e-12 and err[(32, 64)] < 1e-12 and err[(64, 64)] < 1e-12, err
    assert 1e-9 < err[(32, 0)] < 1e-4 and 1e-9 < err[(64, 32)] < 1e-4, err
    from rigidmultiblobswall_amd._lib import RmbError
    with pytest.raises(RmbError):
      ctx.set_option("force_precision", 16)
  finally:
    ctx.close()


@pytest.mark.parametrize("N,G", [(1000, 2), (4097, 3), (10000, 8), (100, 4)])
def test_pair_shards_honour_the_deterministic_symmetric_mode(Ctx, oracle, torch_mod, N, G):
  """deterministic = 2 on a pair shard: whole tile pairs per shard, ordered reduction -- every shard is bit-identical from
  launch to launch and from context to context, and the shards still sum to the product (single kinds, the two-vector
  product and a multi-block operation)."""
  torch = torch_mod
  r, f, eta, a = d2_cloud(N, seed=N + 3)
  t = np.random.RandomState(N).randn(*f.shape)
  fd, td = _dev(torch, f), _dev(torch, t)
  ref_tt = oracle.single_wall_mobility_trans_times_force_oracle(r, f, eta, a)
  ref_rr = oracle.single_wall_mobility_rot_times_torque_oracle(r, t, eta, a)
  runs = []
  for rep in range(2):
    ctx = Ctx(0)
    try:
      ctx.set_positions(r, a, np.zeros(3), wall=True)
      ctx.set_option("deterministic", 2)
      parts = []
      for g in range(G):
        one = [ctx.matvec_pairshard_device("tt", fd, eta, g, G).cpu().numpy(),
               ctx.matvec_pairshard_device("rr", td, eta, g, G).cpu().numpy()]
        one += [x.cpu().numpy() for x in ctx.matvec2_device("tt", fd, td, eta, shard=g, nshards=G)]
        one += [x.cpu().numpy() for x in ctx.matvec_op_device("grand", (fd, td), eta, shard=g, nshards=G)]
        again = ctx.matvec_pairshard_device("tt", fd, eta, g, G).cpu().numpy()
        assert np.array_equal(again, one[0])
        parts.append(one)
      runs.append(parts)
      # atomics are back when the option is cleared
      ctx.set_option("deterministic", 0)
      full = sum(ctx.matvec_pairshard_device("tt", fd, eta, g, G).cpu().numpy() for g in range(G))
      assert rel_err(full, ref_tt) < TOL_D2
    finally:
      ctx.close()
  for g in range(G):
    for x, y in zip(runs[0][g], runs[1][g]):
      assert np.array_equal(x, y)
  tot = [sum(runs[0][g][c] for g in range(G)) for c in range(6)]
  assert rel_err(tot[0], ref_tt) < TOL_D2 and rel_err(tot[1], ref_rr) < TOL_D2
  assert rel_err(tot[2], ref_tt) < TOL_D2
  assert rel_err(tot[3], oracle.single_wall_mobility_trans_times_force_oracle(r, t, eta, a)) < TOL_D2
  o_tr = oracle._wrapped("tr", 1, r, t, eta, a, periodic_length=np.zeros(3), in_plane=False)
  o_rt = oracle._wrapped("rt", 1, r, f, eta, a, periodic_length=np.zeros(3), in_plane=False)
  assert rel_err(tot[4], ref_tt + o_tr) < TOL_D2 and rel_err(tot[5], o_rt + ref_rr) < TOL_D2


def test_streams_that_are_destroyed_between_steps(Ctx, oracle, torch_mod):
  """C hosts with one stream per step: rmb_ctx_release_stream() before hipStreamDestroy, then the next stream is
  adopted without the context ever touching the dead handle (HIP does not validate stream handles).  Switching between
  two LIVE streams needs nothing from the caller."""
  import ctypes
  torch = torch_mod
  hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
  r, f, eta, a = d2_cloud(2000, seed=9)
  ref = oracle.single_wall_mobility_trans_times_force_oracle(r, f, eta, a)
  fd = _dev(torch, f)
  out = torch.empty_like(fd)
  torch.cuda.synchronize()
  ctx = Ctx(0)
  try:
    keep = []
    for step in range(3):
      s = ctypes.c_void_p()
      assert hip.hipStreamCreate(ctypes.byref(s)) == 0
      keep.append(ctypes.c_void_p())                 # a second live stream, so that the allocator cannot hand the
      assert hip.hipStreamCreate(ctypes.byref(keep[-1])) == 0   # address of the destroyed one straight back
      ctx.set_stream(s.value)
      if step == 0:
        ctx.set_positions(_dev(torch, r), a, np.zeros(3), wall=True)
      ctx.matvec_device("tt", fd, eta, out=out)
      ctx.release_stream()                          # waits for the product, forgets the handle
      assert rel_err(out.cpu().numpy(), ref) < TOL_D2
      assert hip.hipStreamDestroy(s) == 0
    # live -> live switches: fenced by the context itself
    for s in keep + [ctypes.c_void_p(0)]:
      out.zero_()
      torch.cuda.synchronize()
      ctx.set_stream(s.value)
      ctx.matvec_device("tt", fd, eta, out=out)
    ctx.synchronize()
    assert rel_err(out.cpu().numpy(), ref) < TOL_D2
    ctx.release_stream()
    for s in keep:
      assert hip.hipStreamDestroy(s) == 0
    out.zero_()
    ctx.matvec_device("tt", fd, eta, out=out)       # follows torch's current stream again
    torch.cuda.synchronize()
    assert rel_err(out.cpu().numpy(), ref) < TOL_D2
  finally:
    ctx.close()


@pytest.mark.parametrize("N,G,L", [(1000, 2, None), (4097, 3, None), (90, 4, None), (700, 2, (14.0, 16.0, 0.0))])
def test_force_and_free_surface_pair_shards_sum_to_the_full_result(Ctx, oracle, torch_mod, N, G, L):
  """rmb_blob_blob_force_pairshard_device and the RMB_TT_FREE_SURFACE pair shard: what the ranks of a G-GPU run each
  evaluate sums to the oracle's forces / free-surface velocities (also below the 128-blob threshold of the symmetric
  path and with pseudo-periodic images)."""
  torch = torch_mod
  r, f, eta, a = d2_cloud(N, seed=N + 11)
  eps, b = 0.4, 0.3 * a
  Lz = np.zeros(3) if L is None else np.asarray(L, dtype=np.float64)
  F_ref = oracle.calc_blob_blob_forces_oracle(r, periodic_length=Lz, repulsion_strength=eps, debye_length=b, blob_radius=a)
  fd = _dev(torch, f)
  ctx = Ctx(0)
  try:
    ctx.set_positions(r, a, Lz, wall=False)
    parts = [ctx.blob_blob_force_pairshard_device(eps, b, a, g, G).cpu().numpy() for g in range(G)]
    assert rel_err(sum(parts).reshape(-1, 3), F_ref) < 1e-12
    assert max(np.abs(p).max() for p in parts) > 0
    if L is None:
      u_ref = oracle.free_surface_mobility_trans_times_force_oracle(r, f, eta, a)
      u = sum(ctx.matvec_pairshard_device("tt_free", fd, eta, g, G).cpu().numpy() for g in range(G))
      assert rel_err(u, u_ref) < TOL_D2
    from rigidmultiblobswall_amd._lib import RmbError
    with pytest.raises(RmbError):
      ctx.blob_blob_force_pairshard_device(eps, b, a, G, G)
  finally:
    ctx.close()


@pytest.mark.parametrize("prec", [64, 32])
def test_force_tile_culling_changes_no_bit(Ctx, oracle, prec):
  """Tile pairs beyond the range of the exponential are skipped (`force_cull`): exp(-(r - 2a)/b) is exactly zero there in
  the kernel and in the reference, so the result is the same array with and without the culling -- here on a monolayer
  whose extent is 40x the force range (most tile pairs culled) and on a cloud within the range (none culled)."""
  rng = np.random.RandomState(4)
  a, b, eps = 0.5, 0.004, 0.3                 # range 2a + 750 b = 4 (float kernel: 2a + 110 b = 1.44)
  n = 6000
  side = int(np.ceil(np.sqrt(n)))
  ij = np.array([(i, j) for i in range(side) for j in range(side)][:n], dtype=np.float64)
  r_far = np.concatenate([ij * 2.05 * a + 0.02 * rng.rand(n, 2), a * (1.0 + rng.rand(n, 1))], axis=1)   # 160 x 160 units
  r_near = rng.rand(500, 3) * 3.0
  # pseudo-periodic in x and y (fp64 path only): the same monolayer with every roller moved by a random multiple of the
  # period, so that nearest images -- not raw separations -- decide what is in range
  Lp = np.array([side * 2.05 * a, side * 2.05 * a, 0.0])
  r_per = r_far.copy()
  r_per[:, 0] += Lp[0] * rng.randint(-2, 3, n)
  r_per[:, 1] += Lp[1] * rng.randint(-2, 3, n)
  cases = [(r_far, np.zeros(3)), (r_near, np.zeros(3))] + ([(r_per, Lp), (r_far, Lp)] if prec == 64 else [])
  for r, Lbox in cases:
    ctx = Ctx(0)
    try:
      ctx.set_positions(r, a, Lbox, wall=False)
      ctx.set_option("precision", prec)
      ctx.set_option("deterministic", 0)
      ctx.set_option("force_sort", 0)      # the culling alone: the caller's order on both sides (same summation order per tile)
      res = {}
      for cull in (1, 0, 1):
        ctx.set_option("force_cull", cull)
        res.setdefault(cull, []).append(ctx.blob_blob_force(eps, b, a))
      # atomics: run-to-run differences are at round-off, a culled pair contributes exactly 0 -> compare at round-off
      scale = np.abs(res[0][0]).max()
      assert np.abs(res[1][0] - res[0][0]).max() <= 1e-13 * scale and np.abs(res[1][1] - res[0][0]).max() <= 1e-13 * scale
      ref = oracle.calc_blob_blob_forces_oracle(r, periodic_length=Lbox, repulsion_strength=eps, debye_length=b, blob_radius=a)
      assert rel_err(res[1][0], ref) < (1e-12 if prec == 64 else 1e-4)
      # pair shards cull too and still sum to the forces
      parts = sum(ctx.blob_blob_force_pairshard_device(eps, b, a, g, 3).cpu().numpy() for g in range(3)).reshape(-1, 3)
      assert rel_err(parts, ref) < (1e-12 if prec == 64 else 1e-4)
    finally:
      ctx.close()


@pytest.mark.parametrize("wall", [True, False])
def test_single_precision_per_blob_radii_product(mob, oracle, wall):
  """precision = 'single' for the per-blob-radii mobility with sources == targets (how the `radii_*` modes call K13,
  mobility/mobility.py:1369-1374; the reference's float build covers it, mobility_pycuda.py:1841-2067): single-precision
  accurate against the oracle; sources != targets keep computing in fp64."""
  rng = np.random.RandomState(12)
  N = 3000
  r, f, eta, a = d2_cloud(N, seed=21)
  radii = a * (0.6 + 0.8 * rng.rand(N))
  name = "single_wall" if wall else "no_wall"
  fn_hip = getattr(mob, name + "_mobility_trans_times_force_source_target_hip")
  fn_ref = getattr(oracle, name + "_mobility_trans_times_force_source_target_oracle")
  ref = fn_ref(r, r, f, radii, radii, eta)
  try:
    e64 = rel_err(mob.mobility_radii_trans_times_force(r, f, eta, a, radii, fn_hip), ref)
    mob.precision = 'single'
    e32 = rel_err(mob.mobility_radii_trans_times_force(r, f, eta, a, radii, fn_hip), ref)
    assert e64 < 1e-12 and 1e-9 < e32 < 2e-5, (e64, e32)
    # sources != targets: one-sided fp64 kernel whatever the switch says
    tgt = r[:500] + 0.3 * a
    e_st = rel_err(fn_hip(r, tgt, f, radii, radii[:500], eta), fn_ref(r, tgt, f, radii, radii[:500], eta))
    assert e_st < 1e-12, e_st
  finally:
    mob.precision = 'double'
  assert rel_err(mob.mobility_radii_trans_times_force(r, f, eta, a, radii, fn_hip), ref) < 1e-12


@pytest.mark.parametrize("n", [130, 1000, 5000])
def test_workgroup_cooperative_kernel_matches_the_per_wave_kernel_and_the_oracle(oracle, n):
  """Option "sym_coop": the four waves of a workgroup share one staged tile and one flush per tile
  (csrc/sym_coop_kernels.h).  Default for launches below one resident round; forced here (2) for every kind, wall /
  no wall / pseudo-periodic, full products and pair shards, against the per-wave kernel (0) and the oracle."""
  import torch
  from rigidmultiblobswall_amd import MobilityContext
  rng = np.random.RandomState(n)
  a, eta = 0.37, 0.9
  box = (n * (4.0 / 3.0) * np.pi * a ** 3 / 0.05) ** (1.0 / 3.0)
  r = rng.rand(n, 3) * box
  r[:, 2] += 0.8 * a                                    # some blobs below z = a: B-damping on load and store
  v = rng.randn(n, 3)
  rd, vd = torch.as_tensor(r.reshape(-1), device="cuda"), torch.as_tensor(v.reshape(-1), device="cuda")
  ctx = MobilityContext(0)
  try:
    for wall, L in ((True, None), (False, None), (True, np.array([box, 0.0, 0.0]))):
      ctx.set_positions(rd, a, L, wall)
      for kind in ("tt", "tr", "rt", "rr"):
        ctx.set_option("sym_coop", 0)
        ref = ctx.matvec_device(kind, vd, eta).cpu().numpy()
        ctx.set_option("sym_coop", 2)
        got = ctx.matvec_device(kind, vd, eta).cpu().numpy()
        assert ctx.last_launch()["chunks"] == 0 or n < 128
        assert rel_err(got, ref) < 1e-13, (kind, wall, L, rel_err(got, ref))
        parts = sum(ctx.matvec_pairshard_device(kind, vd, eta, g, 3).cpu().numpy() for g in range(3))
        assert rel_err(parts, ref) < 1e-13, (kind, wall, "shards")
      if L is None:
        stem = {"tt": "trans_times_force", "rr": "rot_times_torque"}
        for kind in ("tt", "rr"):
          uo = getattr(oracle, ("single_wall" if wall else "no_wall") + "_mobility_" + stem[kind] + "_oracle")(r, v, eta, a)
          assert rel_err(ctx.matvec_device(kind, vd, eta).cpu().numpy(), uo) < 1e-12
    ctx.set_option("sym_coop", 1)
    ctx.set_positions(rd, a, None, True)
    ctx.matvec_device("tt", vd, eta)
    assert ctx.get_option("sym_coop") == 1
  finally:
    ctx.close()


@pytest.mark.parametrize("periodic", [False, True])
def test_force_culling_does_not_depend_on_the_order_of_the_blobs(oracle, periodic):
  """Option "force_sort": the blobs are sorted along a Morton curve on the device before the force kernel's tile
  culling (csrc/rmb_sort.hip), results go back to the caller's indices.  A monolayer listed in lattice order and the
  same monolayer listed in random order give the same forces (to rounding), equal to the oracle; the reference's own
  answer to the short range of the force is a tree (multi_bodies/forces_numba.py:141-271)."""
  import torch
  from rigidmultiblobswall_amd import MobilityContext
  rng = np.random.RandomState(9)
  a, b, eps = 0.5, 0.01, 0.4
  n = 9000
  side = int(np.ceil(np.sqrt(n)))
  ij = np.array([(i, j) for i in range(side) for j in range(side)][:n], dtype=np.float64)
  r = np.concatenate([ij * 2.02 * a + 0.03 * rng.rand(n, 2), a * (1.0 + 0.5 * rng.rand(n, 1))], axis=1)
  L = np.array([side * 2.02 * a, side * 2.02 * a, 0.0]) if periodic else np.zeros(3)
  if periodic:
    r[:, 0] += L[0] * rng.randint(-1, 2, n)              # positions need not lie in one cell
  perm = rng.permutation(n)
  ref = oracle.calc_blob_blob_forces_oracle(r, periodic_length=L, repulsion_strength=eps, debye_length=b, blob_radius=a)
  ctx = MobilityContext(0)
  try:
    out = {}
    for label, rr, sort in (("lattice", r, 1), ("random", r[perm], 1), ("random_unsorted", r[perm], 0)):
      ctx.set_option("force_sort", sort)
      ctx.set_positions(rr, a, L, wall=False)
      out[label] = ctx.blob_blob_force(eps, b, a)
    assert rel_err(out["lattice"], ref) < 1e-12
    assert rel_err(out["random"], ref[perm]) < 1e-12
    assert rel_err(out["random"], out["random_unsorted"]) < 1e-13
    # device entry, pair shards (full-length partials in the CALLER's order), and a new configuration on the same context
    ctx.set_option("force_sort", 1)
    rd = torch.as_tensor(r[perm].reshape(-1), device="cuda")
    ctx.set_positions(rd, a, L, wall=False)
    fd = ctx.blob_blob_force_device(eps, b, a).cpu().numpy().reshape(-1, 3)
    assert rel_err(fd, ref[perm]) < 1e-12
    parts = sum(ctx.blob_blob_force_pairshard_device(eps, b, a, g, 4).cpu().numpy() for g in range(4)).reshape(-1, 3)
    assert rel_err(parts, ref[perm]) < 1e-12
    r2 = r.copy(); r2[:, :2] *= 1.01
    ctx.set_positions(r2[perm], a, L * 1.01, wall=False)
    ref2 = oracle.calc_blob_blob_forces_oracle(r2, periodic_length=L * 1.01, repulsion_strength=eps, debye_length=b, blob_radius=a)
    assert rel_err(ctx.blob_blob_force(eps, b, a), ref2[perm]) < 1e-12
    # single-precision force kernel on the sorted copy
    ctx.set_option("force_precision", 32)
    if not periodic:
      assert rel_err(ctx.blob_blob_force(eps, b, a), ref2[perm]) < 1e-4
  finally:
    ctx.close()


@pytest.mark.parametrize("n", [200, 3000])
def test_cooperative_generic_skeleton_matches_the_per_wave_one(n):
  """symx_coop_kernel (csrc/symx_coop_kernels.h): every multi-block / multi-vector operation, forced cooperative
  (sym_coop = 2) against per wave (0): wall / no wall / pseudo-periodic, in-plane, pair shards, free surface."""
  import torch
  from rigidmultiblobswall_amd import MobilityContext
  rng = np.random.RandomState(n + 1)
  a, eta = 0.4, 1.1
  box = (n * (4.0 / 3.0) * np.pi * a ** 3 / 0.05) ** (1.0 / 3.0)
  r = rng.rand(n, 3) * box
  r[:, 2] += 0.9 * a
  rd = torch.as_tensor(r.reshape(-1), device="cuda")
  vs = [torch.as_tensor(rng.randn(3 * n), device="cuda") for _ in range(4)]
  ctx = MobilityContext(0)
  try:
    for wall, L in ((True, None), (False, None), (True, np.array([0.0, box, 0.0]))):
      ctx.set_positions(rd, a, L, wall)
      cases = [("velocity_from_force_torque", vs[:2], False), ("grand", vs[:2], False), ("force_column", vs[:1], False),
               ("tt_multi", vs[:2], False), ("tt_multi", vs[:3], False), ("rr_multi", vs[:4], False), ("tr_multi", vs[:1], True),
               ("velocity_from_force_torque", vs[:2], True)]
      for op, vecs, in_plane in cases:
        ctx.set_option("sym_coop", 0)
        ref = [o.cpu().numpy() for o in ctx.matvec_op_device(op, vecs, eta, in_plane=in_plane)]
        ctx.set_option("sym_coop", 2)
        got = [o.cpu().numpy() for o in ctx.matvec_op_device(op, vecs, eta, in_plane=in_plane)]
        assert ctx.get_option("last_path") == 3
        for x, y in zip(got, ref):
          assert rel_err(x, y) < 1e-13, (op, wall, L, in_plane)
        parts = [ctx.matvec_op_device(op, vecs, eta, in_plane=in_plane, shard=g, nshards=3) for g in range(3)]
        for c in range(len(ref)):
          assert rel_err(sum(p[c].cpu().numpy() for p in parts), ref[c]) < 1e-13, (op, "shards")
      if not wall:
        ctx.set_option("sym_coop", 0)
        ref = ctx.matvec_device("tt_free", vs[0], eta).cpu().numpy()
        ctx.set_option("sym_coop", 2)
        assert rel_err(ctx.matvec_device("tt_free", vs[0], eta).cpu().numpy(), ref) < 1e-13
  finally:
    ctx.close()


@pytest.mark.parametrize("N", [200, 256, 257, 449, 1000, 4097, 10000])
def test_two_target_blobs_per_lane_equal_the_one_target_kernels(N):
  """sym2t_kernel (context option "sym_two_targets"; a lane keeps blob `lane` of two tile rows, units are (row pair, tile))
  against the one-target symmetric kernels and the oracle: every kind, wall and no wall, odd and even tile counts, a
  partial last tile, whole products and pair shards (step ranges of the row-pair units), forced from the smallest launch."""
  import torch
  from rigidmultiblobswall_amd import MobilityContext
  from oracle import oracle
  rng = np.random.RandomState(N)
  a, eta = 0.4, 1.7
  r = rng.rand(N, 3) * (N / 0.05) ** (1.0 / 3.0) * a
  r[:, 2] += 0.2 * a                       # some blobs below z = a: clamp + B path
  f = rng.randn(N, 3)
  rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
  ctx = MobilityContext(0)
  try:
    for wall in (True, False):
      ctx.set_positions(rd, a, None, wall)
      for kind in ("tt", "tr", "rt", "rr"):
        ctx.set_option("sym_two_targets", 0)
        ref = ctx.matvec_device(kind, fd, eta).clone()
        assert ctx.get_option("last_path") in (1, 3)
        ctx.set_option("sym_two_targets", 2)
        got = ctx.matvec_device(kind, fd, eta).clone()
        assert ctx.get_option("last_path") == 4
        assert rel_err(got.cpu().numpy(), ref.cpu().numpy()) < 1e-13, (N, wall, kind)
        for G in (2, 5):
          tot = torch.zeros_like(ref)
          for g in range(G):
            tot += ctx.matvec_pairshard_device(kind, fd, eta, g, G)
          assert rel_err(tot.cpu().numpy(), ref.cpu().numpy()) < 1e-13, (N, wall, kind, G)
      if N <= 1000:
        ctx.set_option("sym_two_targets", 2)
        u = ctx.matvec_device("tt", fd, eta).cpu().numpy()
        fn = oracle.single_wall_mobility_trans_times_force_oracle if wall else oracle.no_wall_mobility_trans_times_force_oracle
        assert rel_err(u, fn(r, f, eta, a).reshape(-1)) < 1e-12, (N, wall)
    # the default: two targets from one resident round of workgroups on, the cooperative kernel below
    ctx.set_option("sym_two_targets", 1)
    ctx.set_positions(rd, a, None, True)
    ctx.matvec_device("tt", fd, eta)
    assert ctx.get_option("last_path") == (4 if N >= 10000 else 3)
    # periodic products: the two-target instance of the generic skeleton (round 5, symx2t_kernels.h) when forced, the
    # one-target kernels with the option off
    ctx.set_positions(rd, a, np.array([0.0, 9.0 * a * N ** (1 / 3.0), 0.0]), True)
    ctx.set_option("sym_two_targets", 2)
    u2 = ctx.matvec_device("tt", fd, eta).clone()
    assert ctx.get_option("last_path") == 4
    ctx.set_option("sym_two_targets", 0)
    u1 = ctx.matvec_device("tt", fd, eta)
    assert ctx.get_option("last_path") in (1, 3)
    assert rel_err(u2.cpu().numpy(), u1.cpu().numpy()) < 1e-13
  finally:
    ctx.close()


# ---------------------------------------------------------------------------------------------
# round 5: two target blobs per lane for the multi-block / multi-vector operations and the pseudo-periodic products
# (csrc/symx2t_kernels.h).  Reference semantics: fused row mobility_pycuda.py:1266-1391, grand product
# quaternion_integrator_rollers.py:1114-1121, image convention mobility_numba.py:170-197.
# ---------------------------------------------------------------------------------------------
TWO_T_CASES = [
    # (N, wall, L)       odd / even tile counts, a partial last tile, open and pseudo-periodic in one and two directions
    (257, True, None), (1000, True, None), (4097, True, None), (1000, False, None),
    (700, True, (14.0, 16.0, 0.0)), (449, True, (0.0, 13.0, 0.0)), (700, False, (9.0, 0.0, 11.0)),
]


@pytest.mark.parametrize("N,wall,L", TWO_T_CASES)
def test_two_targets_generic_operations_vs_oracle(Ctx, oracle, torch_mod, N, wall, L):
  """Every operation that has a two-targets-per-lane instance, forced from the smallest launch (sym_two_targets = 2),
  against the oracle's blocks: whole products, three pair shards summed, the in-plane masks."""
  torch = torch_mod
  r, f, eta, a = d2_cloud(N, seed=N + 7)
  r = r.copy(); r[:, 2] -= 0.3 * a                      # some blobs below z = a: clamp + B-damping with a wall
  t = np.random.RandomState(N + 8).randn(N, 3)
  ref = _oracle_blocks(oracle, wall, r, f, t, eta, a, L)
  ctx = Ctx(0)
  try:
    ctx.set_option("sym_two_targets", 2)
    ctx.set_positions(_dev(torch, r), a, L, wall=wall)
    fd, td = _dev(torch, f), _dev(torch, t)

    def check(op, vecs, want, in_plane=False, tol=TOL_D2):
      outs = ctx.matvec_op_device(op, vecs, eta, in_plane=in_plane)
      assert ctx.get_option("last_path") == 4, (op, ctx.get_option("last_path"))      # rmb::symx2t_kernel
      for o, w in zip(outs, want):
        assert rel_err(o.cpu().numpy(), w) < tol, (op, N, wall, L, in_plane, rel_err(o.cpu().numpy(), w))
      parts = [ctx.matvec_op_device(op, vecs, eta, in_plane=in_plane, shard=g, nshards=3) for g in range(3)]
      for c, w in enumerate(want):
        assert rel_err(sum(p_[c] for p_ in parts).cpu().numpy(), w) < tol, (op, "shards")

    check("velocity_from_force_torque", (fd, td), [ref["tt_f"] + ref["tr_t"]])
    check("grand", (fd, td), [ref["tt_f"] + ref["tr_t"], ref["rt_f"] + ref["rr_t"]])
    check("force_column", (fd,), [ref["tt_f"], ref["rt_f"]])
    kw = dict(periodic_length=np.zeros(3) if L is None else np.asarray(L, dtype=np.float64))
    for kind in ("tt", "tr", "rt", "rr"):
      check(kind + "_multi", (fd, td), [oracle._wrapped(kind, int(wall), r, v, eta, a, **kw) for v in (f, t)])
    if wall:
      refp = _oracle_blocks(oracle, wall, r, f, t, eta, a, L, in_plane=True)
      check("velocity_from_force_torque", (fd, td), [refp["tt_f"] + refp["tr_t"]], in_plane=True)
      check("tt_multi", (fd, td), [oracle._wrapped("tt", 1, r, v, eta, a, in_plane=True, **kw) for v in (f, t)], in_plane=True)
    if L is not None:
      # the pseudo-periodic single-vector products go to the same instances (sym_device hands them over)
      for kind in ("tt", "tr", "rt", "rr"):
        u = ctx.matvec_device(kind, fd, eta).cpu().numpy()
        assert ctx.get_option("last_path") == 4
        assert rel_err(u, oracle._wrapped(kind, int(wall), r, f, eta, a, **kw)) < TOL_D2, (kind, N, wall, L)
        tot = sum(ctx.matvec_pairshard_device(kind, fd, eta, g, 4) for g in range(4)).cpu().numpy()
        assert rel_err(tot, u) < TOL_SHARD, (kind, "shards")
    # and they equal the one-target kernels to rounding
    ctx.set_option("sym_two_targets", 0)
    one = [o.cpu().numpy() for o in ctx.matvec_op_device("grand", (fd, td), eta)]
    assert ctx.get_option("last_path") in (1, 3)
    ctx.set_option("sym_two_targets", 2)
    two = [o.cpu().numpy() for o in ctx.matvec_op_device("grand", (fd, td), eta)]
    for x, y in zip(two, one):
      assert rel_err(x, y) < 1e-13
  finally:
    ctx.close()


@pytest.mark.parametrize("periodic", [False, True], ids=["open", "periodic_xy"])
def test_two_targets_generic_operations_are_the_default_at_24576_blobs(Ctx, oracle, torch_mod, periodic):
  """configs[2] size, default options: fused row, grand, two-vector and (periodic) single-vector products run on the
  two-target instances (last_path 4) and match the oracle on a sample of targets incl. tile edges."""
  torch = torch_mod
  from test_gpu_parity import _edge_sample
  N = 24576
  r, f, eta, a = d2_cloud(N, seed=3)
  r = r.copy(); r[:, 2] -= 0.45 * a
  t = np.random.RandomState(4).randn(N, 3)
  box = (N * (4.0 / 3.0) * np.pi * a ** 3 / 0.05) ** (1.0 / 3.0)
  L = np.array([box, box, 0.0]) if periodic else None
  tg = _edge_sample(N, k=64)
  r_eff, bdiag, _ = oracle.wall_regularisation(r, a)

  def ref(kind, v):
    x = oracle.raw_matvec_targets(kind, 1, r_eff, v * bdiag[:, None], eta, a, tg, L=L)
    return (x.reshape(-1, 3) * bdiag[tg][:, None]).reshape(-1)

  pick = lambda o: o.cpu().numpy().reshape(-1, 3)[tg].reshape(-1)
  ctx = Ctx(0)
  try:
    ctx.set_positions(_dev(torch, r), a, L, wall=True)
    fd, td = _dev(torch, f), _dev(torch, t)
    u, w = ctx.matvec_op_device("grand", (fd, td), eta)
    assert ctx.get_option("last_path") == 4
    assert rel_err(pick(u), ref("tt", f) + ref("tr", t)) < TOL_D2 and rel_err(pick(w), ref("rt", f) + ref("rr", t)) < TOL_D2
    (u,) = ctx.matvec_op_device("velocity_from_force_torque", (fd, td), eta)
    assert ctx.get_option("last_path") == 4 and rel_err(pick(u), ref("tt", f) + ref("tr", t)) < TOL_D2
    ua, ub = ctx.matvec_op_device("tt_multi", (fd, td), eta)
    assert ctx.get_option("last_path") == 4
    assert rel_err(pick(ua), ref("tt", f)) < TOL_D2 and rel_err(pick(ub), ref("tt", t)) < TOL_D2
    if periodic:
      for kind in ("tt", "rr"):
        u = ctx.matvec_device(kind, fd, eta)
        assert ctx.get_option("last_path") == 4 and rel_err(pick(u), ref(kind, f)) < TOL_D2, kind
  finally:
    ctx.close()
