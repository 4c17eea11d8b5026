"""Roller time steppers (rigidmultiblobswall_amd/rollers.py) on CPU tensors with an oracle-backed context,
against trajectories recorded from the reference's QuaternionIntegratorRollers (tests/golden/g8_*, generator
oracle/gen_golden_rollers.py).  Same seeds => same random draws => same trajectory up to solver tolerance."""
import os

import numpy as np
import pytest

from conftest import golden_files, load_golden, rel_err
from _oracle_ctx import OracleContext
from _rollers_common import (integrator_from_golden, run_and_compare, replay_driven_steps, driven_factory,
                             check_driven_replay)

TRAJ = [p for p in golden_files("g8_rollers_*.npz") if "velocity_pieces" not in p and "prescribed" not in p]


@pytest.mark.parametrize("path", TRAJ, ids=[os.path.basename(p)[11:-4] for p in TRAJ])
def test_trajectory_matches_reference_integrator(oracle, path):
  g = load_golden(path)
  integ = integrator_from_golden(g, OracleContext(oracle), "cpu")
  worst = run_and_compare(g, integ)
  # deterministic schemes: round-off only; stochastic: Lanczos tolerance 1e-10 on the noise
  assert worst < (1e-11 if float(g["kT"]) == 0.0 else 1e-7), worst
  assert integ.wall_overlaps == int(g["wall_overlaps"])
  assert integ.invalid_configuration_count == int(g["invalid_configuration_count"])


def test_velocity_pieces(oracle):
  g = load_golden(golden_files("g8_rollers_velocity_pieces.npz")[0])
  integ = integrator_from_golden(g, OracleContext(oracle), "cpu")
  dt = float(g["dt"])
  v, t = integ.compute_deterministic_velocity_and_torque()
  assert rel_err(v.numpy(), g["det_velocity"]) < 1e-12 and rel_err(t.numpy(), g["det_torque"]) < 1e-14
  assert rel_err(integ.compute_stochastic_linear_velocity(dt).numpy(), g["stochastic_linear_velocity"]) < 1e-7
  assert rel_err(integ.compute_stochastic_velocity(dt).numpy(), g["stochastic_velocity_grand"]) < 1e-7
  assert rel_err(integ.compute_stochastic_linear_velocity_without_drift(dt).numpy(), g["stochastic_without_drift"]) < 1e-7
  assert rel_err(integ.compute_linear_thermal_drift().numpy(), g["thermal_drift"]) < 1e-6   # difference of two close products


def test_prescribed_kinematics_torque_solve(oracle):
  """free_kinematics == 'False': GMRES on M_rr against the dense direct solve of the same equations."""
  g = load_golden(golden_files("g8_rollers_prescribed_kinematics.npz")[0])
  integ = integrator_from_golden(g, OracleContext(oracle), "cpu")
  v, t = integ.compute_deterministic_velocity_and_torque()
  assert rel_err(t.numpy(), g["torque"]) < 1e-8 and rel_err(v.numpy(), g["velocity"]) < 1e-8
  assert 0 < integ.det_iterations_count < 60
  # warm start from the previous (normalised) torque: the second solve needs no iterations
  before = integ.det_iterations_count
  v2, t2 = integ.compute_deterministic_velocity_and_torque()
  assert integ.det_iterations_count - before <= 1 and rel_err(v2.numpy(), g["velocity"]) < 1e-8


def test_rejected_step_is_retried_and_counted(oracle):
  """A step that would put a roller below the wall is rejected (quaternion_integrator_rollers.py:137-151)
  and redone with fresh noise."""
  from rigidmultiblobswall_amd.rollers import RollersIntegrator
  r0 = np.array([[0.0, 0.0, 0.05], [3.0, 0.0, 2.0], [0.0, 3.0, 2.0]])
  integ = RollersIntegrator(r0, "stochastic_first_order_rollers", 0.4, 1.0, tolerance=1e-8, device="cpu",
                            ctx=OracleContext(oracle), rng=np.random.RandomState(3))
  integ.kT = 20.0
  integ.g = 1.0
  for _ in range(10):
    integ.advance_time_step(0.1)
  assert integ.invalid_configuration_count > 0
  assert float(integ.location[:, 2].min()) >= 0.0 and integ.wall_overlaps > 0


def test_device_generator_default_and_scheme_suffix(oracle):
  from rigidmultiblobswall_amd.rollers import RollersIntegrator
  r0 = np.array([[0.0, 0.0, 1.0], [2.0, 0.0, 1.5], [0.0, 2.0, 1.2], [2.0, 2.0, 2.0]])
  runs = []
  for _ in range(2):
    integ = RollersIntegrator(r0, "stochastic_adams_bashforth", 0.4, 1.0, device="cpu", ctx=OracleContext(oracle), seed=11)
    integ.kT = 0.01
    integ.advance_time_step(0.01)
    integ.advance_time_step(0.01)
    runs.append(integ.location.numpy().copy())
  assert np.array_equal(runs[0], runs[1]) and np.abs(runs[0] - r0).max() > 0
  with pytest.raises(AttributeError):
    RollersIntegrator(r0, "no_such_scheme", 0.4, 1.0, device="cpu", ctx=OracleContext(oracle)).advance_time_step(0.1)
  with pytest.raises(ValueError):
    RollersIntegrator(r0, "deterministic_forward_euler", 0.4, 1.0, domain="two_walls", device="cpu", ctx=OracleContext(oracle))


def _write_deck(tmp_path, r0, scheme="stochastic_adams_bashforth_rollers", save_clones="one_file_per_step", extra=""):
  (tmp_path / "blob.vertex").write_text("1\n0 0 0\n")
  with open(tmp_path / "rollers.clones", "w") as fh:
    fh.write("%d\n" % len(r0))
    for x in r0:
      fh.write("%.17g %.17g %.17g 1.0 0.0 0.0 0.0\n" % tuple(x))
  deck = tmp_path / "inputfile_rollers.dat"
  deck.write_text("""# same option names as multi_bodies/examples/rollers/inputfile_rollers.dat
scheme                                 %s
mobility_vector_prod_implementation    pycuda
blob_blob_force_implementation         pycuda
repulsion_strength                     0.0165677856
debye_length                           0.0656
dt                                     0.016
n_steps                                4
n_save                                 2
solver_tolerance                       1e-10
eta                                    1.0e-3
g                                      0.0024892
blob_radius                            0.656
kT                                     0.0041419464
omega_one_roller                       0.0   62.8   0.0
free_kinematics                        True
repulsion_strength_wall                0.0165677856
debye_length_wall                      0.0656
seed                                   7
save_clones                            %s
output_name                            %s
structure  blob.vertex rollers.clones
%s
""" % (scheme, save_clones, str(tmp_path / "run_rollers"), extra))
  return str(deck)


def test_input_deck_drives_the_roller_loop(oracle, tmp_path):
  """The reference's roller deck (examples/rollers/inputfile_rollers.dat option names) -> integrator -> time loop
  -> .clones output in the reference's text format; equals driving the class by hand with the same seed."""
  from rigidmultiblobswall_amd.read_input import ReadInput
  from rigidmultiblobswall_amd import rollers, structures
  rng = np.random.RandomState(0)
  r0 = np.concatenate([rng.rand(14, 2) * 8, 0.8 + rng.rand(14, 1)], axis=1)
  read = ReadInput(_write_deck(tmp_path, r0))
  assert read.free_kinematics == "True" and read.hydro_interactions == 1 and read.save_clones == "one_file_per_step"
  integ = rollers.integrator_from_input(read, device="cpu", ctx=OracleContext(oracle))
  assert integ.Nblobs == 14 and integ.kT == 0.0041419464 and integ.tolerance == 1e-10
  assert np.allclose(integ.omega_one_roller, [0, 62.8, 0]) and integ.repulsion_strength == 0.0165677856
  seen = []
  rollers.run(read, integ, callback=lambda step, it: seen.append(step))
  assert seen == [0, 1, 2, 3]
  # by hand
  ref = rollers.RollersIntegrator(r0, read.scheme, 0.656, 1.0e-3, tolerance=1e-10, device="cpu", ctx=OracleContext(oracle),
                                  rng=np.random.RandomState(7))
  ref.kT, ref.g = 0.0041419464, 0.0024892
  ref.repulsion_strength = ref.repulsion_strength_wall = 0.0165677856
  ref.debye_length = ref.debye_length_wall = 0.0656
  ref.omega_one_roller = np.array([0.0, 62.8, 0.0])
  saved = {0: r0.copy()}
  for step in range(4):
    ref.advance_time_step(0.016)
    saved[step + 1] = ref.location.numpy().copy()
  assert np.array_equal(integ.location.numpy(), saved[4])
  for step in (0, 2, 4):
    n, loc, quat = structures.read_clones_file(str(tmp_path / ("run_rollers.rollers.%08d.clones" % step)))
    assert n == 14 and np.abs(loc - saved[step]).max() < 1e-12 and np.allclose(quat, [1, 0, 0, 0])
  assert not os.path.exists(str(tmp_path / "run_rollers.rollers.00000001.clones"))


def test_input_deck_one_file_output_and_errors(oracle, tmp_path):
  from rigidmultiblobswall_amd.read_input import ReadInput
  from rigidmultiblobswall_amd import rollers
  r0 = np.array([[0.0, 0.0, 1.0], [3.0, 0.0, 1.5], [0.0, 3.0, 1.2]])
  read = ReadInput(_write_deck(tmp_path, r0, scheme="deterministic_forward_euler_rollers", save_clones="one_file"))
  integ = rollers.integrator_from_input(read, device="cpu", ctx=OracleContext(oracle))
  rollers.run(read, integ)
  lines = open(str(tmp_path / "run_rollers.rollers.config")).read().split("\n")
  assert lines[0] == "3" and lines[4] == "3" and lines[8] == "3" and len([l for l in lines if l]) == 12
  read.save_clones = "hdf5"
  with pytest.raises(ValueError):
    rollers.run(read, integ)
  (tmp_path / "blob.vertex").write_text("2\n0 0 0\n1 0 0\n")
  with pytest.raises(ValueError):
    rollers.integrator_from_input(read, device="cpu", ctx=OracleContext(oracle))


def test_a_step_rejected_over_and_over_raises_instead_of_spinning(oracle, capsys):
  """The reference redraws a rejected step without bound and prints one line per rejection
  (quaternion_integrator_rollers.py:287-302).  Here every rejection is reported too, and a step that fails
  `max_consecutive_retries` times in a row raises: at 2.6e5 rollers a too-large dt otherwise spins silently for minutes
  (the run of round 2 that was killed for writing nothing)."""
  from rigidmultiblobswall_amd.rollers import RollersIntegrator
  r0 = np.array([[0.0, 0.0, 0.3], [3.0, 0.0, 2.0], [0.0, 3.0, 2.0]])
  integ = RollersIntegrator(r0, "deterministic_forward_euler_rollers", 0.4, 1.0, device="cpu", ctx=OracleContext(oracle))
  integ.g = 50.0                      # gravity alone pushes roller 0 through the wall in one step, every time
  integ.max_consecutive_retries = 3
  with pytest.raises(RuntimeError, match="rejected 4 times in a row"):
    integ.advance_time_step(1.0)
  out = capsys.readouterr().out
  assert out.count("Invalid configuration") == 4 and "rejection 4 of this step" in out
  assert integ.invalid_configuration_count == 4 and np.array_equal(integ.location.numpy(), r0)
  # an accepted step resets the run length
  integ.g = 0.0
  integ.advance_time_step(0.01)
  assert integ.consecutive_rejections == 0
  integ.report_rejections = False
  integ.g = 50.0
  with pytest.raises(RuntimeError):
    integ.advance_time_step(1.0)
  assert capsys.readouterr().out == ""


def test_single_precision_needs_a_finite_difference_step_that_survives_it(oracle):
  """precision = 'single': products carry ~1e-6 relative error, so the random finite differences need
  rf_delta >= 1e-4 (doc/README.md:512-523: 1e-3 single, 1e-6 double)."""
  from rigidmultiblobswall_amd.rollers import RollersIntegrator
  r0 = np.array([[0.0, 0.0, 1.0], [3.0, 0.0, 1.5], [0.0, 3.0, 1.2]])
  integ = RollersIntegrator(r0, "stochastic_adams_bashforth_rollers", 0.4, 1.0, device="cpu", ctx=OracleContext(oracle), seed=1)
  integ.kT = 0.01
  integ.rf_delta = 1e-6
  with pytest.raises(ValueError, match="rf_delta >= 1e-4"):
    integ.precision = 'single'
  assert integ.precision == 'double'
  integ.rf_delta = 1e-3
  integ.precision = 'single'
  integ.advance_time_step(0.01)
  integ.rf_delta = 1e-6               # changed after the switch: caught at the next stochastic step
  with pytest.raises(ValueError):
    integ.advance_time_step(0.01)
  integ.precision = 'double'
  integ.advance_time_step(0.01)
  with pytest.raises(ValueError):
    integ.force_precision = 'half'
  integ.force_precision = 'double'
  assert integ.ctx.get_option("force_precision") == 64


# ---------------------------------------------------------------------------------------------
# configs[4]'s driven recipe in small, recorded from the reference's own integrator (g8_driven_dense_monolayer)
# ---------------------------------------------------------------------------------------------
def test_driven_dense_monolayer_steps_match_the_reference_integrator(oracle):
  g = load_golden(golden_files("g8_driven_dense_monolayer.npz")[0])
  n_steps = len(g["trajectory"]) - 1
  assert n_steps >= 40 and g["trajectory"].shape[1] >= 200
  # the record itself: the layer starts below its equilibrium height and rises (physical), the reference's own
  # whole-step rejections are in the record
  h = g["trajectory"][:, :, 2].mean(axis=1)
  assert h[-1] > h[0]
  steps = list(range(0, n_steps, 4)) + [1, n_steps - 1]   # every fourth step here, all of them on the GPU
  rejected_at = np.nonzero(np.diff(np.concatenate([[0], g["rejected_cumulative"]])))[0]
  if len(rejected_at):
    steps.append(int(rejected_at[0]))
  res = replay_driven_steps(g, driven_factory(g, lambda: OracleContext(oracle), "cpu"), steps)
  check_driven_replay(g, res, steps)
