"""The CPU oracle against golden vectors produced by the REFERENCE's own Python
(oracle/gen_golden.py ran mobility/mobility_numba.py, the dense builders of mobility/mobility.py and
multi_bodies/forces_numba.py in the build container).  This is what pins the oracle."""
import numpy as np
import pytest

from conftest import KERNEL_KEYS, golden_files, load_golden, rel_err

TOL = 2e-14   # oracle restates the same arithmetic; only summation/rounding order differs


@pytest.mark.parametrize("path", golden_files("g[123]_*.npz"), ids=lambda p: p.split("/")[-1][:-4])
def test_mobility_kernels_match_reference(oracle, path):
  g = load_golden(path)
  r, v, eta, a, L = g["r_vectors"], g["vector"], float(g["eta"]), float(g["a"]), g["periodic_length"]
  checked = 0
  for key, stem in KERNEL_KEYS.items():
    if key not in g:
      continue
    u = getattr(oracle, stem + "_oracle")(r, v, eta, a, periodic_length=L)
    assert rel_err(u, g[key]) < TOL, (key, rel_err(u, g[key]))
    checked += 1
  assert checked >= 2
  # dense builders (mobility.py:967-1013, :1018-1116) agree with the matrix-free kernels
  if "dense_no_wall_tt" in g:
    u = oracle.no_wall_mobility_trans_times_force_oracle(r, v, eta, a)
    assert rel_err(u, g["dense_no_wall_tt"]) < 1e-13
  if "dense_wall_tt" in g:
    u = oracle.single_wall_mobility_trans_times_force_oracle(r, v, eta, a)
    assert rel_err(u, g["dense_wall_tt"]) < 1e-13


@pytest.mark.parametrize("path", golden_files("g5_*.npz"), ids=lambda p: p.split("/")[-1][:-4])
def test_blob_blob_forces_match_reference(oracle, path):
  g = load_golden(path)
  kw = dict(periodic_length=g["periodic_length"], repulsion_strength=float(g["repulsion_strength"]),
            debye_length=float(g["debye_length"]), blob_radius=float(g["blob_radius"]))
  if "radius_blobs" in g:       # forces_numba.py:125-137, one radius per blob
    F = oracle.calc_blob_blob_forces_radii_oracle(g["r_vectors"], g["radius_blobs"], **kw)
  else:
    F = oracle.calc_blob_blob_forces_oracle(g["r_vectors"], **kw)
  assert F.shape == g["force"].shape
  assert rel_err(F, g["force"]) < TOL
  if "force_tree" in g:         # forces_numba.py:142-271: the full sum minus pairs beyond 2 a + 30 b
    assert rel_err(F, g["force_tree"]) < 1e-12


def test_fast_flavour_agrees(oracle):
  """-O3 -ffast-math build (the timed cpu_baseline) stays within fp64 rounding of the strict one."""
  rng = np.random.RandomState(11)
  N, a, eta = 257, 0.5, 1.3
  r = rng.rand(N, 3) * 12 + np.array([0, 0, 0.55])
  f = rng.randn(N, 3)
  for kind in ("tt", "tr", "rt", "rr"):
    for wall in (0, 1):
      u0 = oracle.raw_matvec(kind, wall, r, f, eta, a)
      u1 = oracle.raw_matvec(kind, wall, r, f, eta, a, fast=True)
      assert rel_err(u1, u0) < 1e-13


def test_dense_matches_matvec(oracle):
  rng = np.random.RandomState(12)
  N, a, eta = 40, 0.3, 0.9
  r = rng.rand(N, 3) * 3 + np.array([0, 0, 0.31])
  f = rng.randn(3 * N)
  for kind in ("tt", "tr", "rt", "rr"):
    for wall in (0, 1):
      M = oracle.dense(kind, wall, r, eta, a)
      assert rel_err(M @ f, oracle.raw_matvec(kind, wall, r, f, eta, a)) < 1e-13


def test_source_target_matches_reference(oracle):
  """K13 (per-blob radii, N_src != N_trg): mobility/mobility.py:551-615 over mobility_numba.py:1480-1658."""
  from conftest import GOLDEN
  import os
  g = np.load(os.path.join(GOLDEN, "g4_source_target.npz"))
  for name in ("small", "mixed", "periodic"):
    args = [g[name + "_" + k] for k in ("source", "target", "force", "radius_source", "radius_target")]
    for wall, fn in ((1, oracle.single_wall_mobility_trans_times_force_source_target_oracle),
                     (0, oracle.no_wall_mobility_trans_times_force_source_target_oracle),
                     (2, oracle.free_surface_mobility_trans_times_force_source_target_oracle)):   # mobility_numba.py:1941
      u = fn(*args, float(g[name + "_eta"]), periodic_length=g[name + "_L"])
      assert rel_err(u, g["%s_wall%d" % (name, wall)]) < TOL
  c = "mixed"
  u = oracle.single_wall_mobility_trans_times_force_source_target_oracle(
      g[c + "_source"], g[c + "_source"], g[c + "_force"], g[c + "_radius_source"], g[c + "_radius_source"], float(g[c + "_eta"]))
  assert rel_err(u, g["mixed_radii_self_wall1"]) < TOL
