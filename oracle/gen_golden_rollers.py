"""Golden trajectories of the reference's roller integrator.  Build-container only.

Runs /root/reference/quaternion_integrator/quaternion_integrator_rollers.py (`QuaternionIntegratorRollers`)
itself -- with the numba-backed mobility and force functions, interpreted under the identity stub of
gen_golden.py -- on small seeded suspensions and records the locations after every step.

Two accommodations, both so that the reference's OWN code runs unchanged; nothing is copied:
  * `numba` identity stub (as gen_golden.py);
  * quaternion_integrator/gmres.py cannot be imported under scipy 1.15 (it binds a private Fortran module
    that scipy removed), and quaternion_integrator_rollers.py:15-20 imports it at module level.  An EMPTY
    module object named `gmres` is registered so the class definition loads.  Nothing recorded here calls
    into it: only `solve_mobility_problem` of the articulated schemes does (:1496-1575).
The prescribed-kinematics branch (free_kinematics == 'False' with hydrodynamic interactions) calls
general_application_utils.gmres, which raises TypeError under scipy 1.15 (`tol` keyword).  That branch is
pinned instead by a dense direct solve of the same equations (:905-915) assembled from the reference's
mobility products applied to unit vectors.

Usage:  python oracle/gen_golden_rollers.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import os
import sys
import tempfile
import time
import types
import warnings
from functools import partial

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden  # noqa: E402  (STUB text only)


def load(ref):
  stub_dir = tempfile.mkdtemp(prefix="numba_stub_")
  with open(os.path.join(stub_dir, "numba.py"), "w") as fh:
    fh.write(gen_golden.STUB)
  sys.path.insert(0, stub_dir)
  sys.path.insert(0, os.path.join(ref, "quaternion_integrator"))   # `from quaternion import Quaternion`
  sys.path.insert(0, os.path.join(ref, "multi_bodies"))
  sys.path.insert(0, ref)
  sys.modules["gmres"] = types.ModuleType("gmres")                 # see header
  warnings.simplefilter("ignore")
  from quaternion_integrator import quaternion_integrator_rollers as qir
  from quaternion_integrator.quaternion import Quaternion
  from body import body
  import multi_bodies_functions as mbf
  return qir, Quaternion, body, mbf


def suspension(N, a, seed, spacing=2.6, height=(1.2, 3.0)):
  """Perturbed square lattice of rollers above the wall (in the spirit of create_random_suspension.py)."""
  rng = np.random.RandomState(seed)
  m = int(np.ceil(np.sqrt(N)))
  ij = np.array([(i, j) for i in range(m) for j in range(m)][:N], dtype=float)
  r = np.empty((N, 3))
  r[:, 0:2] = ij * spacing * a + 0.3 * a * rng.randn(N, 2)
  r[:, 2] = a * (height[0] + (height[1] - height[0]) * rng.rand(N))
  return r


def make_integrator(lib, r0, scheme, p):
  qir, Quaternion, body, mbf = lib
  bodies = [body.Body(np.copy(x), Quaternion([1.0, 0.0, 0.0, 0.0]), np.zeros((1, 3)), p["a"]) for x in r0]
  integ = qir.QuaternionIntegratorRollers(bodies, len(bodies), scheme, tolerance=p["tolerance"], domain=p["domain"],
                                          mobility_vector_prod_implementation="numba")
  L = np.asarray(p["periodic_length"], dtype=float)
  integ.calc_one_blob_forces = partial(mbf.calc_one_blob_forces, g=p["g"],
                                       repulsion_strength_wall=p["repulsion_strength_wall"],
                                       debye_length_wall=p["debye_length_wall"])
  integ.calc_blob_blob_forces = partial(mbf.set_blob_blob_forces("numba"), g=p["g"],
                                        repulsion_strength_wall=p["repulsion_strength_wall"],
                                        debye_length_wall=p["debye_length_wall"],
                                        repulsion_strength=p["repulsion_strength"], debye_length=p["debye_length"],
                                        periodic_length=L)
  integ.omega_one_roller = np.asarray(p["omega_one_roller"], dtype=float)
  integ.free_kinematics = p["free_kinematics"]
  integ.hydro_interactions = p["hydro_interactions"]
  integ.eta, integ.a, integ.kT = p["eta"], p["a"], p["kT"]
  integ.periodic_length = L
  integ.print_residual = False
  integ.preprocess = mbf.preprocess
  integ.postprocess = mbf.postprocess
  return integ


BASE = dict(a=0.4, eta=1.1, kT=0.0, g=0.8, repulsion_strength_wall=0.6, debye_length_wall=0.12,
            repulsion_strength=0.5, debye_length=0.1, periodic_length=(0.0, 0.0, 0.0), omega_one_roller=(0.0, 9.0, 0.0),
            free_kinematics="True", hydro_interactions=1, domain="single_wall", tolerance=1e-10, dt=0.01, seed=0)


def run_trajectory(lib, name, scheme, N, n_steps, out_dir, **over):
  p = dict(BASE)
  p.update(over)
  t0 = time.time()
  r0 = suspension(N, p["a"], seed=100 + N)
  integ = make_integrator(lib, r0, scheme, p)
  np.random.seed(p["seed"])
  traj = [r0.copy()]
  for _ in range(n_steps):
    integ.advance_time_step(p["dt"])
    traj.append(np.array([b.location for b in integ.bodies]))
  data = {k: (np.asarray(v) if not isinstance(v, str) else v) for k, v in p.items()}
  np.savez_compressed(os.path.join(out_dir, name + ".npz"), scheme=scheme, trajectory=np.array(traj),
                      wall_overlaps=integ.wall_overlaps, invalid_configuration_count=integ.invalid_configuration_count,
                      **data)
  print("  %-44s %-34s N=%-3d steps=%d  %.1fs" % (name, scheme, N, n_steps, time.time() - t0), flush=True)


def run_driven_monolayer(lib, out_dir, N=256, n_steps=48, only=False):
  """configs[4]'s driven recipe in small: a DENSE torque-driven monolayer (area fraction 0.4, heights 1.1-2 a: below
  the equilibrium height, so it also relaxes upwards) at the reference roller deck's own parameters
  (multi_bodies/examples/rollers/inputfile_rollers.dat: dt = 0.016, omega = 62.8 rad/s, a = 0.656, eta = 1e-3, kT,
  gravity and repulsions), stochastic Adams-Bashforth, >= 40 steps.  At this density and dt the contact repulsion is
  stiff (steps get rejected) and trajectories separate exponentially, so besides the trajectory every step records what
  is needed to REPLAY THAT STEP ALONE from the reference's own state: the numpy RNG state before the step, the previous
  deterministic velocity, the first-step flag and the counters."""
  p = dict(BASE)
  p.update(a=0.656, eta=1.0e-3, kT=0.0041419464, g=0.0024892, repulsion_strength_wall=0.0165677856,
           debye_length_wall=0.0656, repulsion_strength=0.0165677856, debye_length=0.0656,
           omega_one_roller=(0.0, 62.8, 0.0), tolerance=1e-6, dt=0.016, seed=21)
  t0 = time.time()
  a = p["a"]
  rng = np.random.RandomState(7)
  side = int(np.ceil(np.sqrt(N)))
  cell = np.sqrt(np.pi * a ** 2 / 0.4)
  ij = np.array([(i, j) for i in range(side) for j in range(side)][:N], dtype=np.float64)
  r0 = np.empty((N, 3))
  r0[:, :2] = (ij + 0.5) * cell + (rng.rand(N, 2) - 0.5) * (cell - 2.0 * a) * 0.9
  r0[:, 2] = a * (1.1 + 0.9 * rng.rand(N))
  integ = make_integrator(lib, r0, "stochastic_adams_bashforth_rollers", p)
  np.random.seed(p["seed"])
  traj, rng_keys, rng_pos, rng_gauss, prev_vel, first_flag, rejected, overlaps, lanczos = [r0.copy()], [], [], [], [], [], [], [], []
  for step in range(n_steps):
    st = np.random.get_state()
    rng_keys.append(np.asarray(st[1], dtype=np.uint32)); rng_pos.append(int(st[2])); rng_gauss.append((int(st[3]), float(st[4])))
    prev_vel.append(np.zeros(3 * N) if integ.velocities_previous_step is None else np.array(integ.velocities_previous_step))
    first_flag.append(bool(integ.first_step))
    integ.advance_time_step(p["dt"])
    traj.append(np.array([b.location for b in integ.bodies]))
    rejected.append(integ.invalid_configuration_count); overlaps.append(integ.wall_overlaps); lanczos.append(integ.stoch_iterations_count)
    print("    step %2d  %.0f s  mean height %.4f  min height %.4f  rejected %d  lanczos its %d" %
          (step + 1, time.time() - t0, traj[-1][:, 2].mean(), traj[-1][:, 2].min(), rejected[-1], lanczos[-1]), flush=True)
  data = {k: (np.asarray(v) if not isinstance(v, str) else v) for k, v in p.items()}
  np.savez_compressed(os.path.join(out_dir, "g8_driven_dense_monolayer.npz"), scheme="stochastic_adams_bashforth_rollers",
                      trajectory=np.array(traj), rng_keys=np.array(rng_keys), rng_pos=np.array(rng_pos),
                      rng_gauss=np.array(rng_gauss), velocities_previous_step=np.array(prev_vel),
                      first_step=np.array(first_flag), rejected_cumulative=np.array(rejected),
                      wall_overlaps_cumulative=np.array(overlaps), lanczos_iterations_cumulative=np.array(lanczos),
                      wall_overlaps=integ.wall_overlaps, invalid_configuration_count=integ.invalid_configuration_count, **data)
  print("  g8_driven_dense_monolayer N=%d steps=%d  %.1fs" % (N, n_steps, time.time() - t0), flush=True)


def run_velocity_pieces(lib, out_dir):
  """Single calls of the velocity builders (no time step): det + stochastic pieces at one configuration."""
  p = dict(BASE)
  p.update(kT=0.0041, seed=5)
  N = 12
  r0 = suspension(N, p["a"], seed=77)
  integ = make_integrator(lib, r0, "stochastic_adams_bashforth", p)
  np.random.seed(p["seed"])
  det_v, det_t = integ.compute_deterministic_velocity_and_torque()
  lin = integ.compute_stochastic_linear_velocity(p["dt"])
  grand = integ.compute_stochastic_velocity(p["dt"])
  nodrift = integ.compute_stochastic_linear_velocity_without_drift(p["dt"])
  drift = integ.compute_linear_thermal_drift()
  data = {k: (np.asarray(v) if not isinstance(v, str) else v) for k, v in p.items()}
  np.savez_compressed(os.path.join(out_dir, "g8_rollers_velocity_pieces.npz"), r_vectors=r0, det_velocity=det_v,
                      det_torque=det_t, stochastic_linear_velocity=lin, stochastic_velocity_grand=grand,
                      stochastic_without_drift=nodrift, thermal_drift=drift, **data)
  print("  g8_rollers_velocity_pieces", flush=True)


def run_prescribed_kinematics(lib, out_dir):
  """M_rr T = omega - M_rt F;  v = M_tt F + M_tr T  (quaternion_integrator_rollers.py:905-915), dense solve."""
  qir, Quaternion, body, mbf = lib
  p = dict(BASE)
  p.update(free_kinematics="False", omega_one_roller=(1.0, 7.0, -0.5))
  N = 14
  r0 = suspension(N, p["a"], seed=91)
  integ = make_integrator(lib, r0, "deterministic_forward_euler", p)
  L = integ.periodic_length
  force = integ.calc_one_blob_forces(r0, blob_radius=p["a"], blob_mass=1.0)
  force = force + integ.calc_blob_blob_forces(r0, blob_radius=p["a"])
  force = force.reshape(-1)
  eye = np.eye(3 * N)
  M_rr = np.array([integ.mobility_rot_times_torque(r0, e, p["eta"], p["a"], periodic_length=L) for e in eye]).T
  omega = np.tile(np.asarray(p["omega_one_roller"], dtype=float), N)
  rhs = omega - integ.mobility_rot_times_force(r0, force, p["eta"], p["a"], periodic_length=L)
  torque = np.linalg.solve(M_rr, rhs)
  velocity = integ.mobility_trans_times_force(r0, force, p["eta"], p["a"], periodic_length=L)
  velocity = velocity + integ.mobility_trans_times_torque(r0, torque, p["eta"], p["a"], periodic_length=L)
  data = {k: (np.asarray(v) if not isinstance(v, str) else v) for k, v in p.items()}
  np.savez_compressed(os.path.join(out_dir, "g8_rollers_prescribed_kinematics.npz"), r_vectors=r0, force=force,
                      torque=torque, velocity=velocity, **data)
  print("  g8_rollers_prescribed_kinematics", flush=True)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--ref", default="/root/reference")
  ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
  ap.add_argument("--only-driven", action="store_true", help="only the (slow) driven dense monolayer record")
  args = ap.parse_args()
  out_dir = os.path.abspath(args.out)
  lib = load(args.ref)
  if args.only_driven:
    return run_driven_monolayer(lib, out_dir)
  kT = 0.0041
  run_trajectory(lib, "g8_rollers_det_euler", "deterministic_forward_euler_rollers", 20, 3, out_dir)
  run_trajectory(lib, "g8_rollers_det_ab", "deterministic_adams_bashforth_rollers", 20, 4, out_dir)
  run_trajectory(lib, "g8_rollers_det_ab_periodic", "deterministic_adams_bashforth_rollers", 16, 3, out_dir,
                 periodic_length=(4.4, 4.4, 0.0))
  run_trajectory(lib, "g8_rollers_det_euler_no_wall", "deterministic_forward_euler_rollers", 16, 2, out_dir,
                 domain="no_wall", repulsion_strength_wall=0.0)
  run_trajectory(lib, "g8_rollers_det_euler_in_plane", "deterministic_forward_euler_rollers", 16, 2, out_dir,
                 domain="in_plane")
  run_trajectory(lib, "g8_rollers_stoch_first_order", "stochastic_first_order_rollers", 16, 2, out_dir, kT=kT, seed=1)
  run_trajectory(lib, "g8_rollers_stoch_ab", "stochastic_adams_bashforth_rollers", 16, 3, out_dir, kT=kT, seed=2)
  run_trajectory(lib, "g8_rollers_stoch_mid_point", "stochastic_mid_point_rollers", 12, 2, out_dir, kT=kT, seed=3)
  run_trajectory(lib, "g8_rollers_stoch_mid_point_v2", "stochastic_mid_point_version_2_rollers", 12, 2, out_dir, kT=kT,
                 seed=6)
  run_trajectory(lib, "g8_rollers_stoch_trapezoidal", "stochastic_trapezoidal_rollers", 12, 2, out_dir, kT=kT, seed=4)
  run_trajectory(lib, "g8_rollers_stoch_EM", "stochastic_EM_rollers", 12, 2, out_dir, kT=kT, seed=7)
  run_trajectory(lib, "g8_rollers_stoch_GDC", "stochastic_GDC_rollers", 12, 2, out_dir, kT=kT, seed=8)
  # no hydrodynamic interactions: analytic single-roller coefficients, both kinematics
  for fk in ("True", "False"):
    run_trajectory(lib, "g8_rollers_uncorrelated_ab_free_%s" % fk, "stochastic_adams_bashforth_rollers", 16, 3, out_dir,
                   kT=kT, seed=9, hydro_interactions=0, free_kinematics=fk)
    run_trajectory(lib, "g8_rollers_uncorrelated_GDC_free_%s" % fk, "stochastic_GDC_rollers", 16, 2, out_dir,
                   kT=kT, seed=10, hydro_interactions=0, free_kinematics=fk)
  # Brownian steps in the other domains, pseudo-periodic noise, prescribed kinematics without hydrodynamics
  run_trajectory(lib, "g8_rollers_stoch_ab_no_wall", "stochastic_adams_bashforth_rollers", 16, 2, out_dir, kT=kT, seed=13,
                 domain="no_wall", repulsion_strength_wall=0.0)
  run_trajectory(lib, "g8_rollers_stoch_first_order_in_plane", "stochastic_first_order_rollers", 16, 2, out_dir, kT=kT,
                 seed=14, domain="in_plane")
  run_trajectory(lib, "g8_rollers_stoch_mid_point_periodic", "stochastic_mid_point_rollers", 12, 2, out_dir, kT=kT, seed=15,
                 periodic_length=(4.0, 4.0, 0.0))
  run_trajectory(lib, "g8_rollers_uncorrelated_euler_free_False", "deterministic_forward_euler_rollers", 16, 2, out_dir,
                 hydro_interactions=0, free_kinematics="False")
  # larger cases: on the GPU these take the symmetric pair kernels (N >= 128)
  run_trajectory(lib, "g8_rollers_stoch_ab_N160", "stochastic_adams_bashforth_rollers", 160, 2, out_dir, kT=kT, seed=12)
  run_trajectory(lib, "g8_rollers_det_ab_periodic_N144", "deterministic_adams_bashforth_rollers", 144, 2, out_dir,
                 periodic_length=(12.6, 12.6, 0.0))
  run_velocity_pieces(lib, out_dir)
  run_prescribed_kinematics(lib, out_dir)
  run_driven_monolayer(lib, out_dir)


if __name__ == "__main__":
  main()
