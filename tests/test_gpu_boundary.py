"""The drop-in boundary as a maintainer of the reference would use it: the ctypes stub printed in INTEGRATION.md
section B is executed AS WRITTEN (extracted from the file) and its functions -- bound to rmb_mobility_oneshot /
rmb_forces_oneshot, the stateless entry points with the reference wrappers' arguments
(mobility/mobility.py:222-252, multi_bodies/forces_pycuda.py:148-180) -- are checked against the reference's own
outputs (tests/golden g1-g3, g5), including `periodic_length`."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, golden_files, load_golden, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def stub():
  text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
  sect = text[text.index("## B. "):text.index("## Build")]
  code = re.findall(r"```python\n(.*?)```", sect, re.S)[0]
  assert "rmb_mobility_oneshot" in code and "rmb_forces_oneshot" in code and "ctypes.CDLL('librmb_mobility.so')" in code
  # The stub opens the library by its bare name, as it would from an installed location.  Mapping the in-tree build
  # first (one HIP runtime per process, see _lib.py) lets the loader resolve that name through the SONAME.
  from rigidmultiblobswall_amd import _lib
  _lib.load()
  ns = {}
  exec(compile(code, "INTEGRATION.md#B", "exec"), ns)
  return ns


# D2-style fixtures (well-separated clouds): the 1e-12 bar of SURVEY 8(d); dense / contact clouds: 1e-10
def _tol(path):
  return 1e-12 if "wall_cloud" in path or "periodic" in path else 1e-10


@pytest.mark.parametrize("path", golden_files("g[123]_*.npz"), ids=lambda p: p.split("/")[-1][:-4])
def test_stub_mobility_products_match_reference_fixtures(stub, path):
  g = load_golden(path)
  r, v, eta, a, L = g["r_vectors"], g["vector"], float(g["eta"]), float(g["a"]), g["periodic_length"]
  checked = 0
  for key, fn in (("wall_tt", "single_wall_mobility_trans_times_force_hip"),
                  ("no_wall_tt", "no_wall_mobility_trans_times_force_hip"),
                  ("wall_rr", "single_wall_mobility_rot_times_torque_hip")):
    if key in g:
      u = stub[fn](r, v, eta, a, periodic_length=L, step=3, update_PC=1)     # extra kwargs are ignored
      assert u.shape == (3 * len(r),)
      assert rel_err(u, g[key]) < _tol(path), (key, rel_err(u, g[key]))
      checked += 1
  if "wall_tt" in g and "wall_tr" in g:
    u = stub["single_wall_mobility_trans_times_force_torque_hip"](r, v, v, eta, a, periodic_length=L)
    assert rel_err(u, g["wall_tt"] + g["wall_tr"]) < _tol(path)
    checked += 1
  assert checked >= 1


@pytest.mark.parametrize("path", golden_files("g5_*.npz"), ids=lambda p: p.split("/")[-1][:-4])
def test_stub_forces_match_reference_fixtures(stub, path):
  g = load_golden(path)
  if "radius_blobs" in g:
    pytest.skip("per-blob radii go through rmb_blob_blob_force_radii")
  F = stub["calc_blob_blob_forces_hip"](g["r_vectors"], periodic_length=g["periodic_length"],
                                        repulsion_strength=float(g["repulsion_strength"]),
                                        debye_length=float(g["debye_length"]), blob_radius=float(g["blob_radius"]))
  assert F.shape == g["force"].shape
  assert rel_err(F, g["force"]) < 1e-12


def test_stub_reports_errors_as_exceptions(stub):
  r = np.random.rand(4, 3) + 1.0
  with pytest.raises(RuntimeError):
    stub["single_wall_mobility_trans_times_force_hip"](r, r, -1.0, 0.1)      # eta <= 0 -> RMB_ERR_ARG


def test_stub_double_layer_and_pressure_match_reference_fixture(stub):
  g = load_golden(golden_files("g11_aux_operators.npz")[0])
  args = (g["source"], g["target"], g["normals"], g["vector"], g["weights"])
  assert rel_err(stub["double_layer_source_target_hip"](*args), g["dl_no_wall"]) < 1e-12
  assert rel_err(stub["double_layer_source_target_hip"](*args, wall=1), g["dl_wall"]) < 1e-12
  assert rel_err(stub["no_wall_pressure_Stokeslet_hip"](g["source"], g["target"], g["force"]), g["p_no_wall"]) < 1e-12


def test_node_stub_runs_the_multi_device_engine(stub, monkeypatch):
  """The second block of section B -- the same stub on the multi-device engine (rmb_multi_*), executed as written in
  the first block's namespace with RMB_DEVICES listing this box's GPU three times -- against the reference's fixture."""
  text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
  sect = text[text.index("## B. "):text.index("## Build")]
  code = re.findall(r"```python\n(.*?)```", sect, re.S)[1]
  assert "rmb_multi_create" in code and "rmb_multi_matvec" in code
  monkeypatch.setenv("RMB_DEVICES", "0,0,0")
  ns = dict(stub)
  exec(compile(code, "INTEGRATION.md#B-node", "exec"), ns)
  g = load_golden([p for p in golden_files("g[123]_*.npz") if "wall_tt" in load_golden(p)][0])
  r, v, eta, a, L = g["r_vectors"], g["vector"], float(g["eta"]), float(g["a"]), g["periodic_length"]
  u = ns["single_wall_mobility_trans_times_force_hip_node"](r, v, eta, a, periodic_length=L)
  assert rel_err(u, g["wall_tt"]) < 1e-10
  u1 = stub["single_wall_mobility_trans_times_force_hip"](r, v, eta, a, periodic_length=L)
  assert rel_err(u, u1) < 1e-13
  ns["_lib"].rmb_multi_destroy(ns["_engine"])
