"""Shared by the CPU (oracle-backed) and GPU roller tests: build a RollersIntegrator from a g8 fixture."""
import numpy as np


def integrator_from_golden(g, ctx, device, rng=True):
  from rigidmultiblobswall_amd.rollers import RollersIntegrator
  r0 = g["trajectory"][0] if "trajectory" in g else g["r_vectors"]
  scheme = str(g["scheme"]) if "scheme" in g else "deterministic_forward_euler"
  integ = RollersIntegrator(r0, scheme, float(g["a"]), float(g["eta"]), tolerance=float(g["tolerance"]),
                            domain=str(g["domain"]), device=device, ctx=ctx,
                            rng=np.random.RandomState(int(g["seed"])) if rng else None)
  integ.kT = float(g["kT"])
  integ.g = float(g["g"])
  integ.repulsion_strength_wall = float(g["repulsion_strength_wall"])
  integ.debye_length_wall = float(g["debye_length_wall"])
  integ.repulsion_strength = float(g["repulsion_strength"])
  integ.debye_length = float(g["debye_length"])
  integ.periodic_length = np.asarray(g["periodic_length"], dtype=np.float64)
  integ.omega_one_roller = np.asarray(g["omega_one_roller"], dtype=np.float64)
  integ.free_kinematics = str(g["free_kinematics"])
  integ.hydro_interactions = int(g["hydro_interactions"])
  return integ


def run_and_compare(g, integ):
  """Largest deviation from the reference trajectory, relative to the largest displacement of the run."""
  traj = g["trajectory"]
  scale = np.abs(traj[-1] - traj[0]).max()
  worst = 0.0
  for k in range(1, len(traj)):
    integ.advance_time_step(float(g["dt"]))
    worst = max(worst, np.abs(integ.location.cpu().numpy() - traj[k]).max() / scale)
  return worst
