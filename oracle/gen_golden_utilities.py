"""Golden outputs of the reference's one-shot driver multi_bodies/multi_bodies_utilities.py (schemes mobility,
resistance, body_mobility).  Build-container only; same accommodations as gen_golden_rigid_integrator.py (numba stub,
empty `gmres` placeholder module, scipy `tol` -> `rtol` keyword shim); the script itself runs unchanged via runpy.

Case g10_config1_body_mobility is BASELINE.json configs[0]: the reference's own deck
multi_bodies/inputfile_body_mobility.dat (options copied as data; structure paths made absolute and the output
redirected to a scratch directory, because the reference tree is read-only).

Usage:  python oracle/gen_golden_utilities.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import os
import runpy
import shutil
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden_rigid_integrator import prepare, random_quaternions  # noqa: E402


def run_reference(ref, work, deck_text):
  with open(os.path.join(work, "deck.dat"), "w") as fh:
    fh.write(deck_text)
  cwd, argv = os.getcwd(), sys.argv
  os.chdir(work)
  try:
    sys.argv = ["multi_bodies_utilities.py", "--input-file", "deck.dat"]
    # fresh module state per run: the preconditioner builders cache per-body blocks in function attributes
    for m in [m for m in sys.modules if m.startswith("multi_bodies")]:
      del sys.modules[m]
    runpy.run_path(os.path.join(ref, "multi_bodies", "multi_bodies_utilities.py"), run_name="__main__")
  finally:
    os.chdir(cwd)
    sys.argv = argv


def write_structure(work, ID, vertex, loc, quat, slip=None, keyword="structure"):
  with open(os.path.join(work, ID + ".vertex"), "w") as fh:
    fh.write("%d\n" % len(vertex))
    for x in vertex:
      fh.write("%.17g %.17g %.17g\n" % tuple(x))
  with open(os.path.join(work, ID + ".clones"), "w") as fh:
    fh.write("%d\n" % len(loc))
    for x, q in zip(loc, quat):
      fh.write("%.17g %.17g %.17g %.17g %.17g %.17g %.17g\n" % (tuple(x) + tuple(q)))
  line = "%s %s.vertex %s.clones" % (keyword, ID, ID)
  if slip is not None:
    with open(os.path.join(work, ID + ".slip"), "w") as fh:
      fh.write("%d\n" % len(slip))
      for x in slip:
        fh.write("%.17g %.17g %.17g\n" % tuple(x))
    line += " %s.slip" % ID
  return line


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--ref", default="/root/reference")
  ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
  args = ap.parse_args()
  out_dir = os.path.abspath(args.out)
  ref = args.ref
  prepare(ref)
  from read_input import read_vertex_file, read_clones_file
  S = os.path.join(ref, "multi_bodies", "Structures")
  boomerang = read_vertex_file.read_vertex_file(os.path.join(S, "boomerang_N_15.vertex"))[:, :3]
  shell = read_vertex_file.read_vertex_file(os.path.join(S, "shell_N_12_Rg_0.3960_Rh_0.5.vertex"))[:, :3]

  # --- configs[0]: the reference's own deck ----------------------------------------------------------
  work = tempfile.mkdtemp(prefix="ref_util_")
  n, loc, quat = read_clones_file.read_clones_file(os.path.join(S, "boomerang_N_15.clones"))
  line = write_structure(work, "boomerang_N_15", boomerang, loc, [q.entries for q in quat])
  deck = ("scheme                                   body_mobility\n"
          "mobility_blobs_implementation            python\n"
          "eta                                      1.0\n"
          "blob_radius                              0.25\n"
          "output_name                              run.body_mobility\n" + line + "\n")
  run_reference(ref, work, deck)
  np.savez_compressed(os.path.join(out_dir, "g10_config1_body_mobility.npz"), deck=deck, IDs=np.array(["boomerang_N_15"]),
                      vertex_boomerang_N_15=boomerang, locations_boomerang_N_15=np.array(loc),
                      quaternions_boomerang_N_15=np.array([q.entries for q in quat]),
                      body_mobility=np.loadtxt(os.path.join(work, "run.body_mobility.body_mobility.dat")),
                      body_slip_mobility=np.loadtxt(os.path.join(work, "run.body_mobility.body_slip_mobility.dat")))
  shutil.rmtree(work)
  print("  g10_config1_body_mobility", flush=True)

  # --- mixed suspension: mobility (force model + slip), mobility (force file), resistance, body_mobility ---------
  rng = np.random.RandomState(11)
  loc_b = np.array([[0.0, 0.0, 3.0], [3.5, 0.4, 3.3]])
  loc_s = np.array([[1.7 * (k % 2), 4.0 + 1.7 * (k // 2), 1.1 + 0.4 * rng.rand()] for k in range(3)])
  q_b, q_s = random_quaternions(rng, 2), random_quaternions(rng, 3)
  slip_b = rng.randn(15, 3) * 0.4
  common = ("mobility_blobs_implementation            python\n"
            "mobility_vector_prod_implementation      numba\n"
            "blob_blob_force_implementation           numba\n"
            "eta                                      1.1\n"
            "blob_radius                              0.25\n"
            "g                                        0.6\n"
            "solver_tolerance                         1e-11\n"
            "repulsion_strength                       0.3\n"
            "debye_length                             0.1\n"
            "repulsion_strength_wall                  0.4\n"
            "debye_length_wall                        0.1\n"
            "output_name                              run\n")
  FT = rng.randn(5, 6)
  U = rng.randn(5, 6)
  for name, scheme, extra, outputs in (
      ("g10_util_mobility_model_forces", "mobility", "", ("velocity", "force")),
      ("g10_util_mobility_force_file", "mobility", "force_file force.dat\n", ("velocity", "force")),
      ("g10_util_resistance", "resistance", "velocity_file velocity.dat\n", ("force",)),
      ("g10_util_body_mobility", "body_mobility", "", ("body_mobility", "body_slip_mobility"))):
    work = tempfile.mkdtemp(prefix="ref_util_")
    lines = [write_structure(work, "boomerang", boomerang, loc_b, q_b, slip=slip_b),
             write_structure(work, "shell", shell, loc_s, q_s)]
    np.savetxt(os.path.join(work, "force.dat"), FT)
    np.savetxt(os.path.join(work, "velocity.dat"), U)
    deck = "scheme                                   %s\n" % scheme + common + extra + "\n".join(lines) + "\n"
    run_reference(ref, work, deck)
    data = {k: np.loadtxt(os.path.join(work, "run.%s.dat" % k)) for k in outputs}
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), deck=deck, IDs=np.array(["boomerang", "shell"]),
                        vertex_boomerang=boomerang, vertex_shell=shell, locations_boomerang=loc_b, locations_shell=loc_s,
                        quaternions_boomerang=q_b, quaternions_shell=q_s, slip_boomerang=slip_b, force_file=FT,
                        velocity_file=U, **data)
    shutil.rmtree(work)
    print("  " + name, flush=True)


  # --- an obstacle (prescribed kinematics, U = 0) next to free shells: scheme mobility -------------------------------
  work = tempfile.mkdtemp(prefix="ref_util_")
  loc_o, q_o = np.array([[0.8, 2.0, 1.6]]), random_quaternions(rng, 1)
  lines = [write_structure(work, "shell", shell, loc_s, q_s),
           write_structure(work, "fixed", boomerang, loc_o, q_o, keyword="obstacle")]
  deck = "scheme                                   mobility\n" + common + "\n".join(lines) + "\n"
  run_reference(ref, work, deck)
  np.savez_compressed(os.path.join(out_dir, "g10_util_mobility_obstacle.npz"), deck=deck, IDs=np.array(["shell", "fixed"]),
                      vertex_shell=shell, vertex_fixed=boomerang, locations_shell=loc_s, locations_fixed=loc_o,
                      quaternions_shell=q_s, quaternions_fixed=q_o,
                      velocity=np.loadtxt(os.path.join(work, "run.velocity.dat")),
                      force=np.loadtxt(os.path.join(work, "run.force.dat")))
  shutil.rmtree(work)
  print("  g10_util_mobility_obstacle", flush=True)


if __name__ == "__main__":
  main()
