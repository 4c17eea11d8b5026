"""Stokeslet pressure and Stokes double layer, source -> target (mobility/mobility_numba.py:1332-1476, :1662-1766,
:2095-2168; wrappers mobility/mobility.py:1345-1366, :1376-1387, :1432-1442).

Host tests: the CPU oracle against tests/golden/g11_aux_operators.npz, which the reference's own functions produced
(oracle/gen_golden_aux_operators.py).  GPU tests: the HIP kernels through the C ABI against the same fixture and
against the oracle on larger clouds, including the chunked launch, the device-pointer entry points and the edge cases.
Tolerance: relative L2 1e-12 (fp64 sums of O(1e3) terms of both signs).
"""
import ctypes

import numpy as np
import pytest

from conftest import golden_files, load_golden, rel_err

TOL_ORACLE = 2e-14
TOL = 1e-12


@pytest.fixture(scope="module")
def g11():
  return load_golden(golden_files("g11_aux_operators.npz")[0])


def _dl_args(g, self_=False):
  return (g["source"], g["source"] if self_ else g["target"], g["normals"], g["vector"], g["weights"])


# ---------------------------------------------------------------------------------------------
# oracle vs the reference's output (runs without a GPU)
# ---------------------------------------------------------------------------------------------
def test_oracle_pressure_matches_reference(oracle, g11):
  g = g11
  assert rel_err(oracle.no_wall_pressure_Stokeslet_oracle(g["source"], g["target"], g["force"]), g["p_no_wall"]) < TOL_ORACLE
  for k in (0, 7, len(g["source"]) - 1):
    p = oracle.single_wall_pressure_Stokeslet_oracle(g["source"][k:k + 1], g["target"], g["force"][k:k + 1])
    assert rel_err(p, g["p_wall_single_%d" % k]) < TOL_ORACLE
  p = oracle.single_wall_pressure_Stokeslet_oracle(g["source"], g["target"], g["force"])
  assert rel_err(p, g["p_wall_superposed"]) < TOL_ORACLE
  # the reference's multi-source wall output is NOT the superposition (it rescales inside the source loop)
  assert rel_err(g["p_wall_as_written"], g["p_wall_superposed"]) > 0.5
  with pytest.raises(ValueError):
    oracle.no_wall_pressure_Stokeslet_oracle(g["source"], g["target"], g["force"], periodic_length=np.array([3.0, 0, 0]))


def test_oracle_double_layer_matches_reference(oracle, g11):
  g = g11
  a = float(g["blob_radius"])
  for self_, sfx in ((False, ""), (True, "_self")):
    args = _dl_args(g, self_)
    assert rel_err(oracle.double_layer_source_target_oracle(*args), g["dl_no_wall" + sfx]) < TOL_ORACLE
    assert rel_err(oracle.double_layer_source_target_oracle(*args, wall=1), g["dl_wall" + sfx]) < TOL_ORACLE
    assert rel_err(oracle.no_wall_double_layer_source_target_oracle(*args, a), g["dl_rpy" + sfx]) < TOL_ORACLE


def test_oracle_double_layer_of_a_constant_on_a_sphere(oracle):
  """Property the reference does not test: for a closed surface the double layer of a constant density is -c inside
  and 0 outside (Stokes double-layer identity with this kernel's -3/(4 pi) normalisation and outward normals)."""
  rng = np.random.RandomState(5)
  n = 4000
  x = rng.randn(n, 3)
  x /= np.linalg.norm(x, axis=1)[:, None]          # uniform points on the unit sphere, equal weights
  w = np.full(n, 4 * np.pi / n)
  c = np.array([0.3, -1.1, 0.7])
  v = np.tile(c, (n, 1))
  inside = np.array([[0.1, 0.05, -0.2]])
  outside = np.array([[2.5, 1.0, -3.0]])
  ui = oracle.double_layer_source_target_oracle(x, inside, x, v, w)
  uo = oracle.double_layer_source_target_oracle(x, outside, x, v, w)
  assert np.linalg.norm(uo) < 0.05 * np.linalg.norm(c)
  assert np.linalg.norm(np.abs(ui) - np.abs(c)) < 0.08 * np.linalg.norm(c)      # Monte-Carlo quadrature, 4000 points


# ---------------------------------------------------------------------------------------------
# HIP path
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def mob():
  from rigidmultiblobswall_amd import mobility
  return mobility


@pytest.mark.gpu
def test_hip_pressure_golden(mob, g11):
  g = g11
  p = mob.no_wall_pressure_Stokeslet_hip(g["source"], g["target"], g["force"])
  assert p.shape == (len(g["target"]),)
  assert rel_err(p, g["p_no_wall"]) < TOL
  for k in (0, 7, len(g["source"]) - 1):
    p = mob.single_wall_pressure_Stokeslet_hip(g["source"][k:k + 1], g["target"], g["force"][k:k + 1])
    assert rel_err(p, g["p_wall_single_%d" % k]) < TOL
  assert rel_err(mob.single_wall_pressure_Stokeslet_hip(g["source"], g["target"], g["force"]), g["p_wall_superposed"]) < TOL


@pytest.mark.gpu
def test_hip_double_layer_golden(mob, g11):
  g = g11
  a = float(g["blob_radius"])
  for self_, sfx in ((False, ""), (True, "_self")):
    args = _dl_args(g, self_)
    u = mob.double_layer_source_target_hip(*args)
    assert u.shape == (3 * len(args[1]),)
    assert rel_err(u, g["dl_no_wall" + sfx]) < TOL
    assert rel_err(mob.double_layer_source_target_hip(*args, wall=1), g["dl_wall" + sfx]) < TOL
    assert rel_err(mob.no_wall_double_layer_source_target_hip(*args, a), g["dl_rpy" + sfx]) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("ns,nt", [(1, 1), (3, 700), (5000, 64), (4000, 3000), (257, 129)])
def test_hip_aux_operators_vs_oracle(mob, oracle, ns, nt):
  rng = np.random.RandomState(7 * ns + nt)
  box = (max(ns, nt) ** (1.0 / 3.0)) * 1.5
  src = rng.rand(ns, 3) * box + np.array([0, 0, 0.1])
  tgt = rng.rand(nt, 3) * box + np.array([0, 0, 0.1])
  f = rng.randn(ns, 3)
  nrm = rng.randn(ns, 3)
  nrm /= np.linalg.norm(nrm, axis=1)[:, None]
  v = rng.randn(ns, 3)
  w = 0.05 + rng.rand(ns)
  for wall in (0, 1):
    pre = "single_wall" if wall else "no_wall"
    p = getattr(mob, pre + "_pressure_Stokeslet_hip")(src, tgt, f)
    ref = getattr(oracle, pre + "_pressure_Stokeslet_oracle")(src, tgt, f)
    assert np.all(np.isfinite(p)) and rel_err(p, ref) < TOL, (wall, rel_err(p, ref))
    u = mob.double_layer_source_target_hip(src, tgt, nrm, v, w, wall=wall)
    ref = oracle.double_layer_source_target_oracle(src, tgt, nrm, v, w, wall=wall)
    assert np.all(np.isfinite(u)) and rel_err(u, ref) < TOL, (wall, rel_err(u, ref))
  u = mob.no_wall_double_layer_source_target_hip(src, tgt, nrm, v, w, 0.21)
  assert rel_err(u, oracle.no_wall_double_layer_source_target_oracle(src, tgt, nrm, v, w, 0.21)) < TOL
  # on its own nodes: the diagonal is skipped, the wall image of a node is not
  if ns >= 3:
    for wall in (0, 1):
      u = mob.double_layer_source_target_hip(src, src, nrm, v, w, wall=wall)
      assert np.all(np.isfinite(u))
      assert rel_err(u, oracle.double_layer_source_target_oracle(src, src, nrm, v, w, wall=wall)) < TOL
    u = mob.no_wall_double_layer_source_target_hip(src, src, nrm, v, w, 0.21)
    assert rel_err(u, oracle.no_wall_double_layer_source_target_oracle(src, src, nrm, v, w, 0.21)) < TOL


@pytest.mark.gpu
def test_hip_aux_device_entry_points_chunks_and_edges(oracle):
  """Device-pointer variants on a context, forced source chunks (partials + fixed-order reduction: bit-identical
  run to run), empty inputs and the documented argument errors."""
  import torch
  from rigidmultiblobswall_amd import MobilityContext, _lib
  lib = _lib.load()
  rng = np.random.RandomState(3)
  ns, nt = 3001, 1777
  src, tgt = rng.rand(ns, 3) * 9 + np.array([0, 0, 0.2]), rng.rand(nt, 3) * 9 + np.array([0, 0, 0.2])
  f, nrm, v, w = rng.randn(ns, 3), rng.randn(ns, 3), rng.randn(ns, 3), 0.1 + rng.rand(ns)
  dev = lambda x: torch.as_tensor(np.ascontiguousarray(x).reshape(-1), device="cuda")   # noqa: E731
  vp = lambda t: ctypes.c_void_p(t.data_ptr())                                           # noqa: E731
  sd, td, fd, nd, vd, wd = (dev(x) for x in (src, tgt, f, nrm, v, w))
  ctx = MobilityContext(0)
  try:
    for chunks in (0, 1, 5):
      ctx.set_option("chunks", chunks)
      outs = []
      for rep in range(2):
        p = torch.empty(nt, dtype=torch.float64, device="cuda")
        u = torch.empty(3 * nt, dtype=torch.float64, device="cuda")
        _lib.check(lib.rmb_pressure_stokeslet_device(ctx._h, ns, vp(sd), nt, vp(td), vp(fd), None, 1, vp(p)))
        _lib.check(lib.rmb_double_layer_device(ctx._h, ns, vp(sd), nt, vp(td), vp(nd), vp(vd), vp(wd), 1, -1.0, vp(u)))
        torch.cuda.synchronize()
        outs.append((p.cpu().numpy(), u.cpu().numpy()))
      assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])   # atomic-free
      assert rel_err(outs[0][0], oracle.single_wall_pressure_Stokeslet_oracle(src, tgt, f)) < TOL
      assert rel_err(outs[0][1], oracle.double_layer_source_target_oracle(src, tgt, nrm, v, w, wall=1)) < TOL
    ctx.set_option("chunks", 0)
    # no sources: zero output; no targets: nothing to do
    p = torch.full((nt,), 7.0, dtype=torch.float64, device="cuda")
    _lib.check(lib.rmb_pressure_stokeslet_device(ctx._h, 0, None, nt, vp(td), None, None, 0, vp(p)))
    assert float(p.abs().max()) == 0.0
    _lib.check(lib.rmb_double_layer_device(ctx._h, ns, vp(sd), 0, None, vp(nd), vp(vd), vp(wd), 0, -1.0, None))
    # documented errors
    L = np.array([4.0, 0.0, 0.0])
    assert lib.rmb_pressure_stokeslet_device(ctx._h, ns, vp(sd), nt, vp(td), vp(fd), ctypes.c_void_p(L.ctypes.data), 0, vp(p)) != 0
    u = torch.empty(3 * nt, dtype=torch.float64, device="cuda")
    assert lib.rmb_double_layer_device(ctx._h, ns, vp(sd), nt, vp(td), vp(nd), vp(vd), vp(wd), 1, 0.3, vp(u)) != 0
    assert lib.rmb_double_layer_device(ctx._h, ns, vp(sd), nt, vp(td), None, vp(vd), vp(wd), 0, -1.0, vp(u)) != 0
  finally:
    ctx.close()


@pytest.mark.gpu
def test_hip_wrappers_reject_bad_shapes(mob):
  src, tgt = np.zeros((4, 3)), np.ones((2, 3))
  with pytest.raises(ValueError):
    mob.no_wall_pressure_Stokeslet_hip(src, tgt, np.zeros((3, 3)))
  with pytest.raises(ValueError):
    mob.double_layer_source_target_hip(src, tgt, np.zeros((4, 3)), np.zeros((4, 3)), np.zeros(3))
  from rigidmultiblobswall_amd._lib import RmbError
  with pytest.raises(RmbError):
    mob.no_wall_pressure_Stokeslet_hip(src + 1, tgt, np.zeros((4, 3)), periodic_length=np.array([0.0, 5.0, 0.0]))


# ---------------------------------------------------------------------------------------------
# small host-surface members of mobility/mobility.py
# ---------------------------------------------------------------------------------------------
def test_per_blob_radius_clamp_and_damping_match_reference(g11):
  from rigidmultiblobswall_amd import mobility as mob      # host-only helpers: no GPU needed
  g = g11
  assert np.array_equal(mob.shift_heights_different_radius(g["source"], g["radii"]), g["shift_heights_different_radius"])
  B, overlap = mob.damping_matrix_B_different_radius(g["source"], g["radii"])
  assert np.array_equal(B.diagonal(), g["B_different_radius_diag"]) and overlap == bool(g["B_different_radius_overlap"])
  assert bool(g["B_different_radius_overlap"])            # the fixture does contain blobs below their radius


@pytest.mark.gpu
def test_hip_dense_products_and_self_mobility_golden(mob, g11):
  g = g11
  f = g["force"].flatten()
  assert rel_err(mob.single_wall_fluid_mobility_product_hip(g["dense_src"], f, 0.9, 0.2), g["wall_dense_product"]) < TOL
  assert rel_err(mob.no_wall_fluid_mobility_product_hip(g["dense_src"], f, 0.9, 0.2), g["no_wall_dense_product"]) < TOL
  for k in range(3):
    M = mob.single_wall_self_mobility_with_rotation_hip(np.array([0.3, -0.2, float(g["self_6x6_h%d_height" % k])]), 1.3, 0.25)
    ref = g["self_6x6_h%d" % k]
    assert M.shape == (6, 6)
    assert np.abs(M - ref).max() < 1e-13 * np.abs(ref).max(), (k, np.abs(M - ref).max())
