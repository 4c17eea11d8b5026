"""Analytic properties lifted from the reference's unittest file mobility/mobility_test.py
(:101-143 symmetry + positive definiteness of RPY and wall mobility, zero mobility at the wall)
and SURVEY.md section 4 (g.M_tr f = f.M_rt g).  Checked on the oracle here (CPU) and on the HIP
path in test_gpu_parity.py."""
import numpy as np


def _cloud(seed, N, a, zmin):
  rng = np.random.RandomState(seed)
  r = rng.rand(N, 3) * (6 * a * N ** (1.0 / 3.0))
  r[:, 2] += zmin
  return r


def test_tt_rr_symmetric_positive_definite(oracle):
  a, eta = 0.4, 1.1
  r = _cloud(1, 30, a, 1.05 * a)
  for wall in (0, 1):
    for kind in ("tt", "rr"):
      M = oracle.dense(kind, wall, r, eta, a)
      assert np.abs(M - M.T).max() < 1e-13 * np.abs(M).max()
      assert np.linalg.eigvalsh(0.5 * (M + M.T)).min() > 0


def test_tr_is_transpose_of_rt(oracle):
  a, eta = 0.4, 1.1
  r = _cloud(2, 25, a, 1.05 * a)
  for wall in (0, 1):
    Mtr = oracle.dense("tr", wall, r, eta, a)
    Mrt = oracle.dense("rt", wall, r, eta, a)
    assert np.abs(Mtr - Mrt.T).max() < 1e-13 * np.abs(Mtr).max()


def test_grand_mobility_positive_definite(oracle):
  a, eta = 0.4, 1.1
  r = _cloud(3, 12, a, 1.2 * a)
  blocks = {k: oracle.dense(k, 1, r, eta, a) for k in ("tt", "tr", "rt", "rr")}
  G = np.block([[blocks["tt"], blocks["tr"]], [blocks["rt"], blocks["rr"]]])
  assert np.linalg.eigvalsh(0.5 * (G + G.T)).min() > 0


def test_mobility_vanishes_at_wall(oracle):
  """mobility_test.py:132-143 -- a blob with z -> 0 does not move (B-damping, mobility.py:67-84)."""
  a, eta = 0.5, 1.0
  r = _cloud(4, 20, a, 1.5 * a)
  r[0, 2] = 0.0
  f = np.random.RandomState(5).randn(20, 3)
  u = oracle.single_wall_mobility_trans_times_force_oracle(r, f, eta, a).reshape(-1, 3)
  assert np.abs(u[0]).max() == 0.0
  assert np.abs(u[1:]).max() > 0


def test_far_field_is_oseen(oracle):
  """mobility_test.py:70-82 -- finite size correction vanishes with distance."""
  a, eta = 0.01, 1.0
  r = np.array([[0, 0, 0.0], [30.0, 10.0, -5.0]])
  f = np.array([[0.3, -0.2, 0.9], [0, 0, 0]])
  u = oracle.no_wall_mobility_trans_times_force_oracle(r, f, eta, a).reshape(-1, 3)[1]
  d = r[1] - r[0]
  R = np.linalg.norm(d)
  oseen = (f[0] / R + d * np.dot(d, f[0]) / R ** 3) / (8 * np.pi * eta)
  assert np.linalg.norm(u - oseen) < 1e-6 * np.linalg.norm(oseen)
