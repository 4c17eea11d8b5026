"""Input-deck reader.

Same plain-text format and option names as the reference's read_input/read_input.py:9-159:
`key value` lines, `#` comments, repeated `structure` / `obstacle` / `articulated` keys.  Table-driven:
every known option has a type and the reference's default; unknown options are kept in `.options`.
"""
import os

import numpy as np


def _vec(s):
  return np.array(s.split(), dtype=np.float64)


# option -> (attribute name, converter, default)       defaults as read_input.py:48-102
_OPTIONS = {
    "n_steps": ("n_steps", int, 0),
    "initial_step": ("initial_step", int, 0),
    "n_save": ("n_save", int, 1),
    "dt": ("dt", float, 0.0),
    "eta": ("eta", float, 1.0),
    "g": ("g", float, 1.0),
    "tilt_angle": ("theta", float, 0.0),
    "blob_radius": ("blob_radius", float, 1.0),
    "kT": ("kT", float, 1.0),
    "scheme": ("scheme", str, "deterministic_forward_euler"),
    "output_name": ("output_name", str, "run"),
    "seed": ("seed", str, None),
    "repulsion_strength_wall": ("repulsion_strength_wall", float, 1.0),
    "debye_length_wall": ("debye_length_wall", float, 1.0),
    "mobility_blobs_implementation": ("mobility_blobs_implementation", str, "python"),
    "mobility_vector_prod_implementation": ("mobility_vector_prod_implementation", str, "python"),
    "repulsion_strength": ("repulsion_strength", float, 1.0),
    "debye_length": ("debye_length", float, 1.0),
    "blob_blob_force_implementation": ("blob_blob_force_implementation", str, "None"),
    "body_body_force_torque_implementation": ("body_body_force_torque_implementation", str, "None"),
    "save_body_mobility": ("save_body_mobility", str, "False"),
    "save_blobs_mobility": ("save_blobs_mobility", str, "False"),
    "save_velocities": ("save_velocities", str, "False"),
    "slip_file": ("slip_file", str, None),
    "force_file": ("force_file", str, None),
    "velocity_file": ("velocity_file", str, None),
    "solver_tolerance": ("solver_tolerance", float, 1e-8),
    "rf_delta": ("rf_delta", float, 1e-3),
    "periodic_length": ("periodic_length", _vec, np.zeros(3)),
    "omega_one_roller": ("omega_one_roller", _vec, np.zeros(3)),
    "update_PC": ("update_PC", int, 1),
    "domain": ("domain", str, "single_wall"),
    "free_kinematics": ("free_kinematics", str, "True"),
    "hydro_interactions": ("hydro_interactions", int, 1),
    "n_relaxation": ("n_relaxation", int, 0),
    "tracer_radius": ("tracer_radius", float, 0.0),
    "random_state": ("random_state", str, None),
    "nonlinear_solver_tolerance": ("nonlinear_solver_tolerance", float, 1e-8),
    "save_clones": ("save_clones", str, "one_file_per_step"),
    "repulsion_strength_firm": ("repulsion_strength_firm", float, 0.0),
    "firm_delta": ("firm_delta", float, 1e-2),
    "Lub_Cut": ("Lub_Cut", float, 4.5),
    "zmin": ("zmin", float, 0.0),
    "zmax": ("zmax", float, 1e7),
    "domType": ("domType", str, "RPB"),
    "diffusion_coefficient": ("diffusion_coefficient", float, 1.0),
}


_NO_OBSTACLES = ("deterministic_forward_euler_dense_algebra", "stochastic_first_order_RFD", "stochastic_adams_bashforth",
                 "stochastic_first_order_RFD_dense_algebra", "stochastic_traction_EM", "Fixman", "stochastic_traction_AB",
                 "stochastic_Slip_Mid_DLA")


class ReadInput(object):
  def __init__(self, input_file):
    self.input_file = input_file
    self.options = {}
    counts = {"structure": 0, "obstacle": 0, "articulated": 0}
    with open(input_file, "r") as fh:
      for line in fh:
        line = line.split("#", 1)[0].strip()
        if not line:
          continue
        parts = line.split(None, 1)
        key, value = parts[0], (parts[1] if len(parts) > 1 else "")
        if key in counts:
          key, counts[parts[0]] = key + str(counts[key]), counts[key] + 1
        self.options[key] = value
    for key, (attr, conv, default) in _OPTIONS.items():
      raw = self.options.get(key)
      setattr(self, attr, conv(raw) if raw not in (None, "") else default)
    # [vertex_file, clones_file] per structure, then per obstacle (read_input.py:104-121)
    self.structures = [self.options["structure%d" % i].split() for i in range(counts["structure"])]
    self.structures += [self.options["obstacle%d" % i].split() for i in range(counts["obstacle"])]
    self.articulated = [self.options["articulated%d" % i].split() for i in range(counts["articulated"])]
    self.num_free_bodies = counts["structure"]
    self.structures_ID = [os.path.basename(s[1])[:-len(".clones")] for s in self.structures]
    self.num_obstacles = counts["obstacle"]
    # restart (read_input.py:139-144): with initial_step > 0 the bodies start from the .clones files the run saved
    if self.initial_step > 0:
      for k, struct in enumerate(self.structures):
        struct[1] = self.output_name + "." + self.structures_ID[k] + "." + str(self.initial_step).zfill(8) + ".clones"
    # schemes whose reference implementation ignores prescribed kinematics refuse obstacles (read_input.py:146-157)
    if self.num_obstacles > 0 and self.scheme in _NO_OBSTACLES:
      raise ValueError("Obstacles are not implemented for scheme: %s" % self.scheme)

  def random_generator(self, save=True):
    """numpy RandomState of the run as multi_bodies.py:1150-1162 sets it up: restored from the pickled state named by
    `random_state`, else seeded with `seed`, else seeded from OS entropy (the reference leaves numpy's global generator
    entropy-seeded: repeated launches of an unseeded deck form an ensemble); the state at the start of the run is ALWAYS
    pickled to `<output_name>.random_state` so that any run can be repeated or resumed."""
    import pickle
    if self.random_state is not None:
      rng = np.random.RandomState()
      with open(self.resolve(self.random_state), "rb") as fh:
        rng.set_state(pickle.load(fh))
    elif self.seed is not None:
      rng = np.random.RandomState(int(self.seed))
    else:
      rng = np.random.RandomState()
    if save:
      with open(self.output_name + ".random_state", "wb") as fh:
        pickle.dump(rng.get_state(), fh)
    return rng

  def resolve(self, path):
    """Structure paths are relative to the directory the deck is run from; fall back to the deck's own."""
    if os.path.isabs(path) or os.path.exists(path):
      return path
    return os.path.join(os.path.dirname(os.path.abspath(self.input_file)), path)
