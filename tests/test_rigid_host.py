"""Host logic of the rigid-multiblob layer (rows N1/N3 of SURVEY 8f) on CPU: geometry against the
reference's Body/Quaternion outputs (golden g7), the GMRES driver, the readers, and the whole
saddle-point solve with an ORACLE-backed stand-in for the device context."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err


@pytest.fixture(scope="module")
def g7():
  d = np.load(os.path.join(GOLDEN, "g7_rigid_suspension.npz"))
  return {k: d[k] for k in d.files}


def _refs(g7):
  return [g7["shell"] if s else g7["boomerang"] for s in g7["body_is_shell"]]


from _oracle_ctx import OracleContext as OracleCtx  # noqa: E402


def test_blob_positions_and_K_match_reference_body(g7):
  from rigidmultiblobswall_amd.rigid import RigidSuspension, blob_positions

  class Dummy(OracleCtx):
    def __init__(self):
      pass

    def set_positions(self, *a):
      pass

  refs = _refs(g7)
  r = np.concatenate([blob_positions(c, l, q) for c, l, q in zip(refs, g7["locations"], g7["quaternions"])])
  assert np.abs(r - g7["r_vectors"]).max() < 1e-14
  rs = RigidSuspension(refs, g7["locations"], g7["quaternions"], float(g7["a"]), float(g7["eta"]), device="cpu",
                       ctx=Dummy())
  assert np.abs(rs.r_vectors - g7["r_vectors"]).max() < 1e-14
  # K.U and K^T.lambda against the reference's dense K (body.calc_K_matrix)
  rng = np.random.RandomState(0)
  U = rng.randn(6 * rs.n_bodies)
  lam = rng.randn(3 * rs.n_blobs)
  assert rel_err(rs.K_times_U(torch.from_numpy(U)).numpy(), g7["K"] @ U) < 1e-14
  assert rel_err(rs.KT_times_lambda(torch.from_numpy(lam)).numpy(), g7["K"].T @ lam) < 1e-14


def test_gmres_driver_matches_direct_solve():
  from rigidmultiblobswall_amd.rigid import gmres_right_preconditioned
  rng = np.random.RandomState(1)
  n = 300
  A = np.eye(n) * 4 + rng.randn(n, n) * 0.1
  P = np.diag(1.0 / np.diag(A))
  b = rng.randn(n)
  b /= np.linalg.norm(b)
  At, Pt = torch.from_numpy(A), torch.from_numpy(P)
  for restart in (60, 7):
    x, info = gmres_right_preconditioned(lambda v: At @ v, lambda v: Pt @ v, torch.from_numpy(b), tol=1e-11,
                                         restart=restart, maxiter=1000)
    assert info["converged"]
    assert np.linalg.norm(A @ x.numpy() - b) <= 2e-11
    assert rel_err(x.numpy(), np.linalg.solve(A, b)) < 1e-9


def test_saddle_point_solve_matches_reference_direct_solve(g7, oracle):
  """The reference's system assembled from its own Body/K/dense-M pieces and solved directly (golden)
  vs our operator + block-diagonal preconditioner + GMRES (oracle-backed matvec on CPU)."""
  from rigidmultiblobswall_amd.rigid import RigidSuspension
  rs = RigidSuspension(_refs(g7), g7["locations"], g7["quaternions"], float(g7["a"]), float(g7["eta"]), device="cpu",
                       ctx=OracleCtx(oracle))
  U, lam, info = rs.solve_mobility_problem(slip=g7["slip"], force_torque=g7["force_torque"], tol=1e-10)
  assert info["converged"] and info["iterations"] < 60
  assert rel_err(U.reshape(-1), g7["velocities"]) < 1e-8
  assert rel_err(lam.reshape(-1), g7["lambda_blobs"]) < 1e-7


def test_config1_boomerang_body_mobility(oracle):
  """BASELINE.json configs[0] (multi_bodies/inputfile_body_mobility.dat): N = (K^T M^-1 K)^-1 of one
  15-blob boomerang; the preconditioner of a single body IS that matrix."""
  from rigidmultiblobswall_amd.rigid import RigidSuspension
  d = np.load(os.path.join(GOLDEN, "g7_boomerang_body_mobility.npz"))
  rs = RigidSuspension([d["reference_configuration"]], [d["location"]], [d["quaternion"]], float(d["a"]),
                       float(d["eta"]), device="cpu", ctx=OracleCtx(oracle))
  assert np.abs(rs.r_vectors - d["r_vectors"]).max() < 1e-14
  rs.build_preconditioner()
  Nb = rs.groups[0].Nbody[0].numpy()
  assert rel_err(Nb, d["body_mobility"]) < 1e-10
  assert abs(Nb[0, 0] - 0.1507443) < 1e-7          # the value quoted in SURVEY.md 8(c)
  # and the full solve with F applied reproduces N.F
  F = np.array([[0.3, -0.2, 1.0, 0.1, 0.05, -0.4]])
  U, _, info = rs.solve_mobility_problem(force_torque=F, tol=1e-11)
  assert rel_err(U.reshape(-1), d["body_mobility"] @ F.reshape(-1)) < 1e-9


def test_structure_readers_and_shell(tmp_path):
  from rigidmultiblobswall_amd import structures as st
  v = tmp_path / "s.vertex"
  v.write_text("# comment\n3  0.25\n0 0 0\n1.5 0 0  # inline\n\n0 2 0\n")
  c = tmp_path / "s.clones"
  c.write_text("2\n0 0 10 2 0 0 0\n1 2 3 0.5 0.5 0.5 0.5\nignored extra line 1 2 3 4 5 6 7\n")
  coor = st.read_vertex_file(str(v))
  n, loc, q = st.read_clones_file(str(c))
  assert coor.shape == (3, 3) and coor[1, 0] == 1.5
  assert n == 2 and loc.shape == (2, 3) and np.allclose(q[0], [1, 0, 0, 0]) and np.allclose(np.linalg.norm(q, axis=1), 1)
  s = st.icosahedron_shell(0.792079207921)
  assert s.shape == (12, 3) and np.allclose(np.linalg.norm(s, axis=1), 0.792079207921)
  assert abs(st.min_blob_separation(s) / 2 - 0.41642068286674966) < 1e-12   # blob radius the reference pairs with it
  loc, q, L = st.roller_monolayer(100, seed=1)
  d = np.linalg.norm(loc[:, None, :2] - loc[None, :, :2], axis=-1) + np.eye(100) * 10
  assert d.min() > 2 * 1.0155 * 0.9 and loc[:, 2].min() > 1.0155


@pytest.mark.parametrize("res", ["low", "mid"])
def test_pair_active_rods_pinned_reference_velocities(oracle, res):
  """The reference's only pinned known-answer for this path (multi_bodies/examples/pair_active_rods,
  README.md:37-44): body velocities of two active rods near a wall must match
  run_<res>_res.velocity.dat.reference to solver_tolerance (1e-8).  Oracle-backed matvec on CPU."""
  from rigidmultiblobswall_amd.rigid import RigidSuspension
  d = np.load(os.path.join(GOLDEN, "g7_pair_active_rods_%s.npz" % res))
  nb = len(d["locations"])
  rs = RigidSuspension([d["reference_configuration"]] * nb, d["locations"], d["quaternions"], float(d["a"]),
                       float(d["eta"]), device="cpu", ctx=OracleCtx(oracle))
  assert np.abs(rs.r_vectors - d["r_vectors"]).max() < 1e-13
  U, lam, info = rs.solve_mobility_problem(slip=d["slip"], force_torque=d["force_torque"], tol=1e-10)
  assert info["converged"]
  ref = d["velocities_reference"]
  # the reference file holds 12 significant digits of a solve converged to 1e-8
  assert np.abs(U - ref).max() < 2e-8 * np.abs(ref).max()
  assert abs(U[0, 5] - 3.88202665409) < 1e-7 if res == "low" else True


def test_body_mobility_from_resistance_matches_pinv():
  from rigidmultiblobswall_amd.rigid import _body_mobility_from_resistance
  rng = np.random.RandomState(4)
  B = rng.randn(7, 6, 6)
  A = torch.from_numpy(B @ B.transpose(0, 2, 1) + 0.1 * np.eye(6))
  assert torch.allclose(_body_mobility_from_resistance(A), torch.linalg.pinv(A), rtol=1e-9, atol=1e-12)
  # a rank-5 resistance (collinear rod: no resistance to spinning about its axis) must take the pseudo-inverse route
  v = rng.randn(6, 5)
  S = torch.from_numpy(np.stack([v @ v.T, B[0] @ B[0].T + np.eye(6)]))
  N = _body_mobility_from_resistance(S)
  assert torch.allclose(N, torch.linalg.pinv(S), rtol=1e-9, atol=1e-10)
  assert float(N[0].abs().max()) < 1e3


def test_lockstep_gmres_pair_equals_two_solves():
  """gmres_pair_right_preconditioned: each of the two solves sees exactly the iterates it would see alone (same
  solution, same iteration count), whatever the other one does -- including one finishing long before the other."""
  from rigidmultiblobswall_amd.rigid import gmres_right_preconditioned, gmres_pair_right_preconditioned
  rng = np.random.RandomState(21)
  n = 240
  M = np.eye(n) * 3 + rng.randn(n, n) * 0.12
  M[-3:, :-3] = 0.0
  M[:-3, -3:] = 0.0                                      # the last three unknowns form an invariant block
  A = torch.from_numpy(M)
  P = torch.diag(1.0 / torch.diag(A))
  b1 = torch.from_numpy(rng.randn(n))
  b2 = torch.zeros(n, dtype=torch.float64)
  b2[-3:] = torch.from_numpy(rng.randn(3))               # lives in that block: converges in <= 3 iterations
  calls = {"single": 0, "pair": 0}

  def op(x):
    calls["single"] += 1
    return A @ x

  def op2(x, y):
    calls["pair"] += 1
    return A @ x, A @ y
  for tol, restart in ((1e-10, 60), (1e-10, 7)):
    calls["single"] = calls["pair"] = 0
    (x1, i1), (x2, i2) = gmres_pair_right_preconditioned(op, op2, lambda v: P @ v, b1, b2, tol=tol, restart=restart)
    s1, j1 = gmres_right_preconditioned(lambda v: A @ v, lambda v: P @ v, b1, tol=tol, restart=restart)
    s2, j2 = gmres_right_preconditioned(lambda v: A @ v, lambda v: P @ v, b2, tol=tol, restart=restart)
    assert torch.equal(x1, s1) and torch.equal(x2, s2)
    assert i1["iterations"] == j1["iterations"] and i2["iterations"] == j2["iterations"] <= 3
    assert calls["pair"] == j2["iterations"] and calls["single"] >= j1["iterations"] - j2["iterations"]
    assert float(torch.linalg.norm(A @ x1 - b1) / torch.linalg.norm(b1)) < 1e-9
