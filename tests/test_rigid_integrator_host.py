"""Rigid-multiblob time integrators (rigidmultiblobswall_amd/rigid_integrator.py) on CPU tensors with an
oracle-backed context, against trajectories the reference's own driver (multi_bodies.py) produced for the same
decks (tests/golden/g9_*, generator oracle/gen_golden_rigid_integrator.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_files, load_golden
from _oracle_ctx import OracleContext
from _rigid_common import replay, reference_counters

CASES = [p for p in golden_files("g9_rigid_*.npz") if "16shells" not in p]


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[9:-4] for p in CASES])
def test_deck_replay_matches_reference_driver(oracle, tmp_path, path):
  g = load_golden(path)
  integ, worst_x, worst_q = replay(g, tmp_path, "cpu", OracleContext(oracle))
  # solver tolerance of the decks is 1e-10 (GMRES and Lanczos); deterministic runs differ by that, stochastic ones by
  # the Lanczos tolerance amplified through sqrt(2 kT / dt)
  tol = 1e-7 if float(g["kT"]) == 0.0 else 1e-6
  assert worst_x < tol and worst_q < tol, (worst_x, worst_q)
  # the solvers take exactly as many iterations as the reference's (scipy GMRES(60) with the same right preconditioner,
  # the reference's Lanczos with the same stopping rule): its `.info` file records the totals of the run
  ref = reference_counters(g)
  assert integ.invalid_configuration_count == ref["invalid_configuration_count"] == 0
  assert integ.det_iterations_count == ref["deterministic_iterations_count"]
  assert integ.stoch_iterations_count == ref["stochastic_iterations_count"]


def test_quaternion_helpers_match_reference_formulas():
  from rigidmultiblobswall_amd.rigid import (quaternion_from_rotation_torch, quaternion_multiply_torch,
                                             quaternion_rotation_matrix_torch, quaternion_rotation_matrix)
  rng = np.random.RandomState(0)
  q = rng.randn(5, 4); q /= np.linalg.norm(q, axis=1)[:, None]
  phi = rng.randn(5, 3); phi[2] = 0.0
  R = quaternion_rotation_matrix_torch(torch.from_numpy(q)).numpy()
  assert np.abs(R - quaternion_rotation_matrix(q)).max() < 1e-15
  dq = quaternion_from_rotation_torch(torch.from_numpy(phi)).numpy()
  n = np.linalg.norm(phi, axis=1)
  assert np.allclose(dq[:, 0], np.cos(n / 2)) and np.allclose(dq[2], [1, 0, 0, 0])
  assert np.allclose(dq[0, 1:], np.sin(n[0] / 2) * phi[0] / n[0])
  # rotating by phi then composing equals multiplying the rotation matrices
  qq = quaternion_multiply_torch(torch.from_numpy(dq), torch.from_numpy(q))
  R2 = quaternion_rotation_matrix_torch(qq).numpy()
  Rd = quaternion_rotation_matrix_torch(torch.from_numpy(dq)).numpy()
  assert np.abs(R2 - Rd @ R).max() < 1e-14


def test_set_configuration_moves_bodies_and_keeps_preconditioner(oracle):
  from rigidmultiblobswall_amd.rigid import RigidSuspension, blob_positions
  g = load_golden(golden_files("g9_rigid_det_euler.npz")[0])
  refs = [g["vertex_boomerang"]] * 2 + [g["vertex_shell"]] * 3
  loc = np.concatenate([g["locations_boomerang"], g["locations_shell"]])
  quat = np.concatenate([g["quaternions_boomerang"], g["quaternions_shell"]])
  rs = RigidSuspension(refs, loc, quat, 0.25, 1.1, device="cpu", ctx=OracleContext(oracle))
  rs.build_preconditioner()
  K_pc = rs.groups[0].K_pc.clone()
  loc2 = loc + 0.1
  quat2 = np.roll(quat, 1, axis=0)
  rs.set_configuration(loc2, quat2)
  r = np.concatenate([blob_positions(c, l, q) for c, l, q in zip(refs, loc2, quat2)])
  assert np.abs(rs.r_vectors - r).max() < 1e-14
  assert torch.equal(rs.groups[0].K_pc, K_pc) and not torch.equal(rs.groups[0].K, K_pc)


def test_stochastic_forcing_has_the_covariance_square_root_property(oracle):
  """noise = P^-1 (P^T M P)^{1/2} z: applying the construction twice through its transpose gives M z, i.e.
  for G = P^-1 S (S symmetric), G G^T = M.  Checked against the dense oracle mobility on a small suspension."""
  from rigidmultiblobswall_amd.rigid import RigidSuspension
  g = load_golden(golden_files("g9_rigid_det_euler.npz")[0])
  refs = [g["vertex_shell"]] * 3
  rs = RigidSuspension(refs, g["locations_shell"], g["quaternions_shell"], 0.25, 1.1, device="cpu", ctx=OracleContext(oracle))
  n = 3 * rs.n_blobs
  G = np.empty((n, n))
  for k in range(n):
    e = torch.zeros(n, dtype=torch.float64); e[k] = 1.0
    G[:, k] = rs.stochastic_forcing(e, 1.0, tol=1e-12)[0].numpy()
  M = oracle.dense("tt", 1, rs.r_vectors, 1.1, 0.25)
  assert np.abs(G @ G.T - M).max() < 1e-8 * np.abs(M).max()


def test_warm_start_saves_iterations_and_keeps_the_trajectory(oracle, tmp_path):
  """warm_start = True seeds GMRES with the previous normalised solution (what the reference's `x0 = self.first_guess`
  intends): same trajectory to solver tolerance, fewer iterations in a deterministic run."""
  from rigidmultiblobswall_amd.read_input import ReadInput
  from rigidmultiblobswall_amd import rigid_integrator
  from _rigid_common import write_case
  g = load_golden(golden_files("g9_rigid_det_ab.npz")[0])
  read = ReadInput(write_case(g, str(tmp_path)))
  counts, finals = [], []
  for warm in (False, True):
    integ = rigid_integrator.integrator_from_input(read, device="cpu", ctx=OracleContext(oracle))
    integ.warm_start = warm
    for step in range(read.n_steps):
      integ.advance_time_step(read.dt, step=step)
    counts.append(integ.det_iterations_count)
    finals.append(torch.cat([integ.location.reshape(-1), integ.orientation.reshape(-1)]).numpy())
  assert counts[1] < counts[0], counts
  assert np.abs(finals[0] - finals[1]).max() < 1e-8


def test_restart_from_saved_clones_and_random_state(oracle, tmp_path):
  """initial_step > 0 restarts from <output>.<ID>.<step>.clones (read_input.py:139-144); `random_state` restores the
  pickled numpy state written at the start of a run (multi_bodies.py:1150-1162)."""
  import os
  import pickle
  from rigidmultiblobswall_amd.read_input import ReadInput
  from rigidmultiblobswall_amd import rigid_integrator, structures
  from _rigid_common import write_case
  g = load_golden(golden_files("g9_rigid_det_euler.npz")[0])
  deck = write_case(g, str(tmp_path))
  text = open(deck).read().replace("n_steps                                  3", "n_steps                                  4")
  open(deck, "w").write(text)
  read = ReadInput(deck)
  rigid_integrator.run(read, rigid_integrator.integrator_from_input(read, device="cpu", ctx=OracleContext(oracle)))
  straight = {ID: structures.read_clones_file(os.path.join(str(tmp_path), "run.%s.00000004.clones" % ID)) for ID in read.structures_ID}
  for ID in read.structures_ID:
    os.remove(os.path.join(str(tmp_path), "run.%s.00000004.clones" % ID))
  open(deck, "w").write(text + "initial_step 2\n")
  again = ReadInput(deck)
  assert all(s[1].endswith(".00000002.clones") for s in again.structures)
  rigid_integrator.run(again, rigid_integrator.integrator_from_input(again, device="cpu", ctx=OracleContext(oracle)))
  for ID in read.structures_ID:
    n, loc, quat = structures.read_clones_file(os.path.join(str(tmp_path), "run.%s.00000004.clones" % ID))
    assert np.abs(loc - straight[ID][1]).max() < 1e-12 and np.abs(quat - straight[ID][2]).max() < 1e-12
  # random state: saved at the start, reloadable
  rng = read.random_generator(save=True)
  first = rng.normal(0.0, 1.0, 5)
  assert os.path.exists(read.output_name + ".random_state")
  open(deck, "w").write(text + "random_state %s\n" % (read.output_name + ".random_state"))
  again = ReadInput(deck)
  assert np.array_equal(again.random_generator(save=False).normal(0.0, 1.0, 5), first)
  with open(read.output_name + ".random_state", "rb") as fh:
    assert pickle.load(fh)[0] == "MT19937"


def test_obstacles_are_refused_where_the_reference_refuses_them(tmp_path):
  from rigidmultiblobswall_amd.read_input import ReadInput
  g = load_golden(golden_files("g9_rigid_obstacle_det_euler.npz")[0])
  from _rigid_common import write_case
  deck = write_case(g, str(tmp_path))
  text = open(deck).read().replace("deterministic_forward_euler", "stochastic_traction_AB")
  open(deck, "w").write(text)
  with pytest.raises(ValueError):
    ReadInput(deck)
