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


# ---------------------------------------------------------------------------------------------
# configs[4]'s driven recipe in small, recorded from the reference's own integrator (g8_driven_dense_monolayer)
# ---------------------------------------------------------------------------------------------
def replay_driven_steps(g, integ_factory, steps):
  """Every listed step is replayed ALONE from the reference's own state (locations, previous deterministic velocity,
  first-step flag, numpy RNG state): the dense driven monolayer is chaotic, a free-running replay separates from the
  record at any tolerance, a single step does not.  Returns per step (deviation / largest displacement of the step,
  rejections of the step, Lanczos iterations of the step)."""
  import torch
  traj = g["trajectory"]
  out = []
  for k in steps:
    rng = np.random.RandomState(0)
    rng.set_state(("MT19937", g["rng_keys"][k], int(g["rng_pos"][k]), int(g["rng_gauss"][k][0]), float(g["rng_gauss"][k][1])))
    integ = integ_factory(traj[k], rng)
    integ.first_step = bool(g["first_step"][k])
    if not integ.first_step:
      integ.velocities_previous_step = torch.as_tensor(g["velocities_previous_step"][k], dtype=torch.float64, device=integ.device)
    integ.report_rejections = False
    integ.advance_time_step(float(g["dt"]))
    dev = np.abs(integ.location.cpu().numpy() - traj[k + 1]).max() / np.abs(traj[k + 1] - traj[k]).max()
    out.append((dev, integ.invalid_configuration_count, integ.stoch_iterations_count))
    integ.close()
  return out


def driven_factory(g, ctx_factory, device):
  from rigidmultiblobswall_amd.rollers import RollersIntegrator

  def make(r, rng):
    integ = RollersIntegrator(r, str(g["scheme"]), float(g["a"]), float(g["eta"]), tolerance=float(g["tolerance"]),
                              device=device, ctx=ctx_factory(), rng=rng)
    integ.kT, integ.g = float(g["kT"]), float(g["g"])
    integ.repulsion_strength_wall, integ.debye_length_wall = float(g["repulsion_strength_wall"]), float(g["debye_length_wall"])
    integ.repulsion_strength, integ.debye_length = float(g["repulsion_strength"]), float(g["debye_length"])
    integ.omega_one_roller = np.asarray(g["omega_one_roller"], dtype=np.float64)
    return integ
  return make


def check_driven_replay(g, res, steps, iteration_slack=0):
  """iteration_slack: the GPU products carry atomic-order round-off, so a Lanczos run sitting exactly on its stopping
  threshold may stop one iteration earlier or later than the reference; the noise then differs at the tolerance."""
  rej = np.concatenate([[0], g["rejected_cumulative"]])
  its = np.concatenate([[0], g["lanczos_iterations_cumulative"]])
  for (dev, n_rej, n_its), k in zip(res, steps):
    assert n_rej == rej[k + 1] - rej[k], (k, n_rej, rej[k + 1] - rej[k])      # rejected exactly when the reference rejects
    same = n_its == its[k + 1] - its[k]
    assert same or abs(n_its - (its[k + 1] - its[k])) <= iteration_slack, (k, n_its, its[k + 1] - its[k])
    assert dev < (1e-5 if same else 1e-4), (k, dev)                           # Lanczos tolerance of the record: 1e-6


