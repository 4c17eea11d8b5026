"""utilities.py on the GPU: the reference script's outputs for the same decks (golden g10), and the velocity-field
grid evaluation against the oracle's source->target product."""
import os

import numpy as np
import pytest

from conftest import golden_files, load_golden, rel_err
from test_utilities_host import _run

pytestmark = pytest.mark.gpu

CASES = golden_files("g10_*.npz")


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[4:-4] for p in CASES])
def test_utilities_scheme_matches_reference_script(tmp_path, path):
  g = load_golden(path)
  read, out = _run(g, tmp_path, "cuda:0", None)
  for key in ("velocity", "force", "body_mobility", "body_slip_mobility"):
    if key in g:
      assert rel_err(np.loadtxt(read.output_name + "." + key + ".dat"), g[key]) < 1e-9, key


def test_velocity_field_grid_and_values(oracle, tmp_path):
  """plot_velocity_field (multi_bodies_utilities.py:74-186): cell-centred tracers, x the fast axis, tracer radius 0,
  values = source->target product of the blob forces."""
  from rigidmultiblobswall_amd import utilities
  rng = np.random.RandomState(2)
  a, eta = 0.25, 1.1
  r = rng.rand(40, 3) * 4 + np.array([0, 0, 0.5])
  lam = rng.randn(40, 3)
  grid = [-1.0, 5.0, 7, -2.0, 4.0, 5, 0.0, 6.0, 4]        # x0 x1 nx  y0 y1 ny  z0 z1 nz
  coor, vel = utilities.velocity_field(grid, r, lam, a, eta, tracer_radius=0.0, wall=True, output=str(tmp_path / "run"))
  assert coor.shape == (7 * 5 * 4, 3)
  dx = 6.0 / 7
  assert np.allclose(coor[0], [-1 + dx / 2, -2 + 0.6, 0.75]) and np.allclose(coor[1] - coor[0], [dx, 0, 0])
  assert np.allclose(coor[7] - coor[0], [0, 1.2, 0]) and np.allclose(coor[35] - coor[0], [0, 0, 1.5])
  ref = oracle.single_wall_mobility_trans_times_force_source_target_oracle(r, coor, lam, np.full(40, a), np.zeros(len(coor)), eta)
  assert rel_err(vel, ref) < 1e-12
  text = open(str(tmp_path / "run.velocity_field.vtk")).read().split("\n")
  assert text[3] == "DATASET RECTILINEAR_GRID" and text[4] == "DIMENSIONS 8 6 5"
  first = np.array(text[text.index("VECTORS velocity double") + 1].split(), dtype=float)
  assert np.allclose(first, vel[0], rtol=1e-10)
  # no-wall variant
  coor2, vel2 = utilities.velocity_field(grid, r, lam, a, eta, wall=False)
  ref2 = oracle.no_wall_mobility_trans_times_force_source_target_oracle(r, coor2, lam, np.full(40, a), np.zeros(len(coor2)), eta)
  assert rel_err(vel2, ref2) < 1e-12


def test_dense_builder_large_body_uses_many_workgroups(oracle):
  """body_dense_tt_kernel with the whole suspension as one body (the dense schemes): N = 600 blobs."""
  from rigidmultiblobswall_amd import mobility as mob
  rng = np.random.RandomState(5)
  a, eta = 0.2, 0.9
  r = rng.rand(600, 3) * 6 + np.array([0, 0, 0.3])
  M = mob.single_wall_fluid_mobility_hip(r, eta, a)
  f = rng.randn(1800)
  assert rel_err(M @ f, oracle.single_wall_mobility_trans_times_force_oracle(r, f, eta, a)) < 1e-12
