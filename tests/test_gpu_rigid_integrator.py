"""Rigid-multiblob time integrators on the GPU: replay of the decks the reference's driver was run on
(tests/golden/g9_*), and size-independent checks on a larger suspension."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_files, load_golden, rel_err
from _rigid_common import replay, reference_counters

pytestmark = pytest.mark.gpu

CASES = golden_files("g9_rigid_*.npz")


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[9:-4] for p in CASES])
def test_deck_replay_matches_reference_driver(tmp_path, path):
  g = load_golden(path)
  integ, worst_x, worst_q = replay(g, tmp_path, "cuda:0", None)
  tol = 1e-7 if float(g["kT"]) == 0.0 else 1e-6
  assert worst_x < tol and worst_q < tol, (worst_x, worst_q)
  ref = reference_counters(g)
  assert integ.invalid_configuration_count == ref["invalid_configuration_count"] == 0
  assert abs(integ.det_iterations_count - ref["deterministic_iterations_count"]) <= 2     # atomics: round-off level
  assert integ.stoch_iterations_count == ref["stochastic_iterations_count"]
  integ.close()


def _suspension(nb, seed):
  from rigidmultiblobswall_amd import structures as st
  R = 1.0155
  shell = st.icosahedron_shell(0.792079207921 * R)
  a = st.min_blob_separation(shell) / 2
  loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=seed)
  return shell, a, loc, quat


def test_brownian_slip_trapz_step_on_a_large_suspension():
  """600 shells (7200 blobs), two stochastic_Slip_Trapz steps with the device generator: reproducible for a fixed
  seed (atomics => round-off only), quaternions stay unit, nothing is rejected, every solve converged."""
  from rigidmultiblobswall_amd.rigid_integrator import RigidIntegrator
  nb = 600
  shell, a, loc, quat = _suspension(nb, 3)
  out = []
  for _ in range(2):
    integ = RigidIntegrator([shell] * nb, loc, quat, "stochastic_Slip_Trapz", a, 0.957e-3, tolerance=1e-6, device="cuda:0", seed=9)
    integ.kT, integ.g = 0.0041419464, 0.0024892 * 12
    integ.repulsion_strength_wall, integ.debye_length_wall = 0.0165677856, 0.0656
    integ.repulsion_strength, integ.debye_length = 0.0165677856, 0.0656
    for step in range(2):
      integ.advance_time_step(0.01, step=step)
    assert integ.invalid_configuration_count == 0 and integ.stoch_iterations_count > 0
    assert integ.det_iterations_count < 2 * 3 * 40
    q = integ.orientation
    assert float((torch.linalg.norm(q, dim=1) - 1).abs().max()) < 1e-12
    out.append((integ.location.cpu().numpy(), q.cpu().numpy()))
    integ.close()
  assert np.abs(out[0][0] - out[1][0]).max() < 1e-8 and np.abs(out[0][1] - out[1][1]).max() < 1e-8
  assert np.abs(out[0][0] - loc).max() > 1e-5


def test_brownian_step_with_single_precision_products():
  """The integrator's precision switch: the same Brownian steps with the fp32 twin of the pair sweep.  At the schemes'
  solver tolerance (1e-4) the solvers take the same number of iterations and the trajectory moves by ~1e-6 of the
  displacement scale; the switch can be flipped back."""
  from rigidmultiblobswall_amd.rigid_integrator import RigidIntegrator
  nb = 400
  shell, a, loc, quat = _suspension(nb, 5)
  res = {}
  for prec in ("double", "single"):
    integ = RigidIntegrator([shell] * nb, loc, quat, "stochastic_Slip_Trapz", a, 0.957e-3, tolerance=1e-4, device="cuda:0", seed=4)
    integ.kT, integ.g = 0.0041419464, 0.0024892 * 12
    integ.repulsion_strength_wall, integ.debye_length_wall = 0.0165677856, 0.0656
    integ.repulsion_strength, integ.debye_length = 0.0165677856, 0.0656
    integ.precision = prec
    assert integ.precision == prec
    for step in range(2):
      integ.advance_time_step(0.01, step=step)
    res[prec] = (integ.location.cpu().numpy(), integ.det_iterations_count, integ.stoch_iterations_count,
                 integ.invalid_configuration_count)
    with pytest.raises(ValueError):
      integ.precision = "half"
    integ.precision = "double"
    integ.close()
  moved = np.abs(res["double"][0] - loc).max()
  assert res["single"][3] == 0 and res["double"][3] == 0
  assert abs(res["single"][1] - res["double"][1]) <= 2 and abs(res["single"][2] - res["double"][2]) <= 1
  diff = np.abs(res["single"][0] - res["double"][0]).max()
  assert 0 < diff < 1e-3 * moved, (diff, moved)          # fp32 products were used, and only perturb the step


def test_deterministic_step_equals_solve_plus_update():
  """deterministic_forward_euler == one RigidSuspension solve with the integrator's own force model + the
  quaternion update formula, on 300 shells."""
  from rigidmultiblobswall_amd.rigid_integrator import RigidIntegrator
  from rigidmultiblobswall_amd.rigid import RigidSuspension, quaternion_rotation_matrix
  nb = 300
  shell, a, loc, quat = _suspension(nb, 4)
  eta, dt = 0.957e-3, 0.02
  integ = RigidIntegrator([shell] * nb, loc, quat, "deterministic_forward_euler", a, eta, tolerance=1e-10, device="cuda:0")
  integ.g, integ.repulsion_strength_wall, integ.debye_length_wall = 0.03, 0.0165677856, 0.0656
  FT = integ.force_torque_calculator().cpu().numpy()
  integ.advance_time_step(dt, step=0)
  rs = RigidSuspension([shell] * nb, loc, quat, a, eta)
  U, lam, info = rs.solve_mobility_problem(force_torque=FT, tol=1e-10)
  assert rel_err(integ.location.cpu().numpy(), loc + dt * U[:, :3]) < 1e-12
  # orientation: R_new = R(omega dt) R_old
  w = U[:, 3:] * dt
  n = np.linalg.norm(w, axis=1)
  dq = np.concatenate([np.cos(n / 2)[:, None], np.sin(n / 2)[:, None] * w / n[:, None]], axis=1)
  R_new = quaternion_rotation_matrix(integ.orientation.cpu().numpy())
  R_ref = quaternion_rotation_matrix(dq) @ quaternion_rotation_matrix(quat)
  assert np.abs(R_new - R_ref).max() < 1e-9
  rs.close()
  integ.close()


@pytest.mark.parametrize("devices", [None, "0,0,0"])
def test_command_line_runs_a_reference_deck(tmp_path, devices):
  """`python -m rigidmultiblobswall_amd --input-file deck` = the reference's `python multi_bodies.py --input-file deck`:
  same deck, same output files (.clones per step, .bodies_info, .info with the same iteration totals).  With
  `--devices` the pair sweeps run on the single-process multi-device engine (here: this box's GPU listed three times)
  and the deck reproduces the same reference trajectory."""
  import subprocess
  import sys
  from conftest import ROOT
  from _rigid_common import write_case
  from rigidmultiblobswall_amd import structures
  g = load_golden([p for p in CASES if p.endswith("g9_rigid_stoch_slip_trapz.npz")][0])
  deck = write_case(g, str(tmp_path))
  res = subprocess.run([sys.executable, "-m", "rigidmultiblobswall_amd", "--input-file", deck] +
                       (["--devices", devices] if devices else []), cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
  for ID in [str(x) for x in g["IDs"]]:
    tl = g["trajectory_locations_" + ID]
    n, loc, quat = structures.read_clones_file(os.path.join(str(tmp_path), "run.%s.%08d.clones" % (ID, len(tl) - 1)))
    assert np.abs(loc - tl[-1]).max() < 1e-6 * np.abs(tl[-1] - tl[0]).max()
  info = open(os.path.join(str(tmp_path), "run.info")).read()
  ref = reference_counters(g)
  assert "stochastic_iterations_count    = %d" % ref["stochastic_iterations_count"] in info
  assert "num_blobs          %d" % (15 * 2 + 12 * 3) in open(os.path.join(str(tmp_path), "run.bodies_info")).read()


def test_gpu_stack_equals_oracle_backed_stack_on_a_mid_size_suspension(oracle):
  """250 shells (3000 blobs), one stochastic_Slip_Trapz step with the same numpy stream: the whole stack on the GPU
  (symmetric kernels, device GMRES / Lanczos) against the same integrator driven by the oracle-backed CPU context."""
  from rigidmultiblobswall_amd.rigid_integrator import RigidIntegrator
  from _oracle_ctx import OracleContext
  nb = 250
  shell, a, loc, quat = _suspension(nb, 6)
  res = []
  for device, ctx in (("cuda:0", None), ("cpu", OracleContext(oracle))):
    integ = RigidIntegrator([shell] * nb, loc, quat, "stochastic_Slip_Trapz", a, 0.957e-3, tolerance=1e-9, device=device,
                            ctx=ctx, rng=np.random.RandomState(12))
    integ.kT, integ.g = 0.0041419464, 0.03
    integ.repulsion_strength_wall, integ.debye_length_wall = 0.0165677856, 0.0656
    integ.repulsion_strength, integ.debye_length = 0.0165677856, 0.0656
    integ.advance_time_step(0.01, step=0)
    res.append((integ.location.cpu().numpy(), integ.orientation.cpu().numpy(), integ.det_iterations_count,
                integ.stoch_iterations_count))
    if ctx is None:
      integ.close()
  scale = np.abs(res[1][0] - loc).max()
  assert np.abs(res[0][0] - res[1][0]).max() < 1e-6 * scale
  assert np.abs(res[0][1] - res[1][1]).max() < 1e-7
  assert abs(res[0][2] - res[1][2]) <= 2 and res[0][3] == res[1][3]


def test_blob_forces_in_two_launches_equal_the_tensor_formula():
  """_blob_forces through the library (pair repulsion, then rmb_one_blob_force_device adding weight and wall repulsion to its
  z entries: multi_bodies_functions.py:153-188) against the tensor formula it replaces, to the last bit or two (torch divides
  by a scalar through its reciprocal) -- blobs below contact (h < a) and far above included -- with and without either
  repulsion, and on its own (overwrite)."""
  from rigidmultiblobswall_amd import MobilityContext
  from rigidmultiblobswall_amd.rigid_integrator import RigidIntegrator
  nb = 20
  shell, a, loc, quat = _suspension(nb, 6)
  loc = loc.copy()
  loc[0, 2] = 0.85            # lowest blobs of this shell end up below z = a
  loc[1, 2] = 9.0
  nat = RigidIntegrator([shell] * nb, loc, quat, "deterministic_forward_euler", a, 0.957e-3, tolerance=1e-6, device="cuda:0")
  ref = RigidIntegrator([shell] * nb, loc, quat, "deterministic_forward_euler", a, 0.957e-3, tolerance=1e-6, device="cuda:0")
  ref.susp.native_helpers = False
  try:
    r = nat.susp.r_dev.view(-1, 3)
    assert float(r[:, 2].min()) < a < float(r[:, 2].max())
    for g_, ew, eb in ((0.0024892 * 12, 0.0165677856, 0.0165677856), (0.03, 0.0, 0.0165677856), (0.03, 0.0165677856, 0.0), (0.0, 0.0, 0.0)):
      for it in (nat, ref):
        it.g, it.repulsion_strength_wall, it.debye_length_wall, it.repulsion_strength, it.debye_length = g_, ew, 0.0656, eb, 0.0656
      fa, fb = nat._blob_forces(r), ref._blob_forces(ref.susp.r_dev.view(-1, 3))
      assert fa.shape == fb.shape == (12 * nb, 3)
      assert float((fa - fb).abs().max()) <= 1e-15 * max(float(fb.abs().max()), 1e-300), (g_, ew, eb, float((fa - fb).abs().max()))
    ctx = MobilityContext(0)
    out = ctx.one_blob_force_device(r.contiguous(), a, 0.5, 0.2, 0.1)
    h = r[:, 2]
    want = -0.5 + torch.where(h > a, 2.0 * torch.exp(-(h - a) / 0.1), torch.full_like(h, 2.0))
    assert float((out.view(-1, 3)[:, 2] - want).abs().max()) <= 4e-16 * float(want.abs().max()) and float(out.view(-1, 3)[:, :2].abs().max()) == 0.0
    ctx.close()
  finally:
    nat.close(); ref.close()
