"""Golden trajectories of rigid-multiblob suspensions from the reference's OWN driver.  Build-container only.

Runs /root/reference/multi_bodies/multi_bodies.py (`python multi_bodies.py --input-file deck`) unchanged, via
runpy, on small decks written here, and records what it saved (`.clones` per step).  Three accommodations make
the reference's code runnable in this image; none replaces any of its logic and nothing is copied:
  * `numba` identity stub (as gen_golden.py): the numba kernels run interpreted;
  * an EMPTY module object named `gmres`: quaternion_integrator/gmres.py binds a private Fortran module that
    scipy 1.15 no longer has, and it is imported at module level; the driver's solves do not use it (they call
    general_application_utils.gmres -> scipy.sparse.linalg.gmres, quaternion_integrator_multi_bodies.py:1523);
  * scipy renamed gmres' keyword `tol` to `rtol` (the README pins scipy 1.10): scipy.sparse.linalg.gmres is
    wrapped to accept `tol` and forward it as `rtol` -- the same solver, the same tolerance.

Each fixture holds the deck text, the structure files' arrays, the initial clones and the saved trajectory.

Usage:  python oracle/gen_golden_rigid_integrator.py [--ref /root/reference] [--out tests/golden] [--only NAME]
"""
import argparse
import glob
import os
import runpy
import shutil
import sys
import tempfile
import time
import types
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden  # noqa: E402  (STUB text only)


def prepare(ref):
  stub_dir = tempfile.mkdtemp(prefix="numba_stub_")
  with open(os.path.join(stub_dir, "numba.py"), "w") as fh:
    fh.write(gen_golden.STUB)
  sys.path.insert(0, stub_dir)
  sys.path.insert(0, os.path.join(ref, "quaternion_integrator"))
  sys.path.insert(0, os.path.join(ref, "multi_bodies"))
  sys.path.insert(0, ref)
  sys.modules["gmres"] = types.ModuleType("gmres")
  import scipy.sparse.linalg as spla
  original = spla.gmres

  def gmres_tol_keyword(A, b, x0=None, tol=1e-5, atol=0.0, **kw):
    return original(A, b, x0=x0, rtol=tol, atol=atol, **kw)
  spla.gmres = gmres_tol_keyword
  warnings.simplefilter("ignore")


DECK = """scheme                                   {scheme}
mobility_blobs_implementation            {mobility_blobs}
mobility_vector_prod_implementation      {mobility_vector_prod}
blob_blob_force_implementation           numba
body_body_force_torque_implementation    None
domain                                   {domain}
eta                                      1.1
blob_radius                              {a}
g                                        0.6
kT                                       {kT}
solver_tolerance                         1e-10
rf_delta                                 1e-3
repulsion_strength                       0.3
debye_length                             0.1
repulsion_strength_wall                  0.4
debye_length_wall                        0.1
dt                                       {dt}
n_steps                                  {n_steps}
n_save                                   1
update_PC                                {update_PC}
seed                                     {seed}
save_clones                              one_file_per_step
output_name                              run
{structures}
"""


def read_clones(path):
  with open(path) as fh:
    rows = [l.split() for l in fh if l.strip()]
  n = int(rows[0][0])
  d = np.array(rows[1:n + 1], dtype=np.float64)
  return d[:, 0:3], d[:, 3:7]


def random_quaternions(rng, n):
  q = rng.randn(n, 4)
  return q / np.linalg.norm(q, axis=1)[:, None]


def case(ref, out_dir, name, scheme, bodies, n_steps, kT=0.0, dt=0.01, seed=1, domain="single_wall", update_PC=1,
         a=0.25, slip=None):
  """bodies: list of (ID, vertex array, locations, quaternions)."""
  t0 = time.time()
  work = tempfile.mkdtemp(prefix="ref_run_")
  lines = []
  data = {}
  obstacles = []
  for entry in bodies:
    ID, vertex, loc, quat = entry[:4]
    with open(os.path.join(work, ID + ".vertex"), "w") as fh:
      fh.write("%d\n" % len(vertex))
      for x in vertex:
        fh.write("%.17g %.17g %.17g\n" % tuple(x))
    with open(os.path.join(work, ID + ".clones"), "w") as fh:
      fh.write("%d\n" % len(loc))
      for x, q in zip(loc, quat):
        fh.write("%.17g %.17g %.17g %.17g %.17g %.17g %.17g\n" % (tuple(x) + tuple(q)))
    keyword = "obstacle" if len(entry) > 4 and entry[4] == "obstacle" else "structure"
    if keyword == "obstacle":
      obstacles.append(ID)
    line = "%s %s.vertex %s.clones" % (keyword, ID, ID)
    if slip is not None and ID in slip:
      with open(os.path.join(work, ID + ".slip"), "w") as fh:
        fh.write("%d\n" % len(slip[ID]))
        for x in slip[ID]:
          fh.write("%.17g %.17g %.17g\n" % tuple(x))
      line += " %s.slip" % ID
      data["slip_" + ID] = np.asarray(slip[ID])
    lines.append(line)
    data["vertex_" + ID] = np.asarray(vertex)
    data["locations_" + ID] = np.asarray(loc)
    data["quaternions_" + ID] = np.asarray(quat)
  wall = domain == "single_wall"
  deck = DECK.format(scheme=scheme, mobility_blobs="python" if wall else "python_no_wall",
                     mobility_vector_prod="numba" if wall else "numba_no_wall", domain=domain, a=a, kT=kT, dt=dt,
                     n_steps=n_steps, update_PC=update_PC, seed=seed, structures="\n".join(lines))
  with open(os.path.join(work, "deck.dat"), "w") as fh:
    fh.write(deck)
  cwd = os.getcwd()
  argv = sys.argv
  os.chdir(work)
  try:
    sys.argv = ["multi_bodies.py", "--input-file", "deck.dat"]
    # a fresh module namespace per run: the preconditioner builders keep state in function attributes
    for m in [m for m in sys.modules if m.startswith("multi_bodies")]:
      del sys.modules[m]
    runpy.run_path(os.path.join(ref, "multi_bodies", "multi_bodies.py"), run_name="__main__")
  finally:
    os.chdir(cwd)
    sys.argv = argv
  for ID in [e[0] for e in bodies]:
    files = sorted(glob.glob(os.path.join(work, "run.%s.*.clones" % ID)))
    assert len(files) == n_steps + 1, files
    traj = [read_clones(f) for f in files]
    data["trajectory_locations_" + ID] = np.array([t[0] for t in traj])
    data["trajectory_quaternions_" + ID] = np.array([t[1] for t in traj])
  with open(os.path.join(work, "run.info")) as fh:
    info = fh.read()
  np.savez_compressed(os.path.join(out_dir, name + ".npz"), deck=deck, IDs=np.array([b[0] for b in bodies]),
                      scheme=scheme, n_steps=n_steps, seed=seed, kT=kT, info=info, obstacles=np.array(obstacles), **data)
  shutil.rmtree(work)
  print("  %-40s %-30s steps=%d  %.1fs" % (name, scheme, n_steps, time.time() - t0), flush=True)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--ref", default="/root/reference")
  ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
  ap.add_argument("--only", default=None)
  args = ap.parse_args()
  out_dir = os.path.abspath(args.out)
  prepare(args.ref)
  S = os.path.join(args.ref, "multi_bodies", "Structures")
  from read_input import read_vertex_file     # the reference's own reader
  boomerang = read_vertex_file.read_vertex_file(os.path.join(S, "boomerang_N_15.vertex"))[:, :3]
  shell = read_vertex_file.read_vertex_file(os.path.join(S, "shell_N_12_Rg_0.3960_Rh_0.5.vertex"))[:, :3]
  a = 0.25
  rng = np.random.RandomState(42)

  def mixed(nb_boom, nb_shell):
    out = []
    if nb_boom:
      loc = np.array([[3.5 * k, 0.4 * k, 3.0 + 0.3 * k] for k in range(nb_boom)])
      out.append(("boomerang", boomerang, loc, random_quaternions(rng, nb_boom)))
    if nb_shell:
      m = int(np.ceil(np.sqrt(nb_shell)))
      loc = np.array([[1.7 * (k % m), 4.0 + 1.7 * (k // m), 1.1 + 0.4 * rng.rand()] for k in range(nb_shell)])
      out.append(("shell", shell, loc, random_quaternions(rng, nb_shell)))
    return out

  def with_obstacle(nb_shell):
    """free shells next to a fixed boomerang-shaped obstacle (an `obstacle` line: prescribed kinematics, U = 0)"""
    free = mixed(0, nb_shell)
    obst = ("fixed", boomerang, np.array([[0.8, 2.0, 1.6]]), random_quaternions(rng, 1), "obstacle")
    return free + [obst]

  kT = 0.0041
  cases = [
      ("g9_rigid_det_euler", "deterministic_forward_euler", mixed(2, 3), 3, {}),
      ("g9_rigid_det_ab", "deterministic_adams_bashforth", mixed(2, 3), 4, {}),
      ("g9_rigid_det_ab_update_pc2", "deterministic_adams_bashforth", mixed(1, 3), 4, dict(update_PC=2)),
      ("g9_rigid_det_midpoint", "deterministic_midpoint", mixed(2, 3), 2, {}),
      ("g9_rigid_det_euler_no_wall", "deterministic_forward_euler", mixed(2, 2), 2, dict(domain="no_wall")),
      ("g9_rigid_det_euler_slip", "deterministic_forward_euler", mixed(2, 0), 3,
       dict(slip={"boomerang": np.random.RandomState(7).randn(15, 3) * 0.5})),
      ("g9_rigid_stoch_first_order_RFD", "stochastic_first_order_RFD", mixed(1, 3), 2, dict(kT=kT, seed=2)),
      ("g9_rigid_stoch_ab", "stochastic_adams_bashforth", mixed(1, 3), 3, dict(kT=kT, seed=3)),
      ("g9_rigid_stoch_slip_trapz", "stochastic_Slip_Trapz", mixed(2, 3), 3, dict(kT=kT, seed=4)),
      ("g9_rigid_stoch_slip_mid", "stochastic_Slip_Mid", mixed(2, 3), 2, dict(kT=kT, seed=5)),
      ("g9_rigid_stoch_EM", "stochastic_EM", mixed(1, 3), 2, dict(kT=kT, seed=7)),
      ("g9_rigid_stoch_traction_EM", "stochastic_traction_EM", mixed(1, 3), 2, dict(kT=kT, seed=8)),
      ("g9_rigid_stoch_traction_AB", "stochastic_traction_AB", mixed(1, 3), 3, dict(kT=kT, seed=9)),
      ("g9_rigid_stoch_GDC_RFD", "stochastic_GDC_RFD", mixed(1, 3), 2, dict(kT=kT, seed=11)),
      ("g9_rigid_obstacle_det_euler", "deterministic_forward_euler", with_obstacle(3), 3, {}),
      ("g9_rigid_obstacle_slip_trapz", "stochastic_Slip_Trapz", with_obstacle(3), 2, dict(kT=kT, seed=10)),
      ("g9_rigid_stoch_slip_trapz_16shells", "stochastic_Slip_Trapz", mixed(0, 16), 2, dict(kT=kT, seed=6)),
  ]
  # Round 5: the dense-algebra schemes (quaternion_integrator_multi_bodies.py:110, :552, :738, :1346).  Fixman and the RFD
  # scheme take N^{1/2} W as V S^{1/2} W from numpy's eigendecomposition (stochastic_forcing_eig), which depends on the
  # eigenvector basis: only boomerangs at generic positions there (distinct eigenvalues of the body mobility; a shell's
  # near-degenerate pairs would make the trajectory depend on round-off).  Slip_Mid_DLA uses the symmetric square root.
  cases += [
      ("g9_rigid_dense_det_euler", "deterministic_forward_euler_dense_algebra", mixed(2, 3), 3, {}),
      ("g9_rigid_dense_stoch_RFD", "stochastic_first_order_RFD_dense_algebra", mixed(3, 0), 2, dict(kT=kT, seed=21)),
      ("g9_rigid_dense_Fixman", "Fixman", mixed(3, 0), 2, dict(kT=kT, seed=22)),
      ("g9_rigid_dense_slip_mid_DLA", "stochastic_Slip_Mid_DLA", mixed(2, 3), 2, dict(kT=kT, seed=23)),
  ]
  # a larger deterministic case (480 blobs: several tiles of the symmetric kernel, chunked sweeps), one step
  cases.append(("g9_rigid_det_euler_40shells", "deterministic_forward_euler", mixed(0, 40), 1, {}))
  cases.append(("g9_rigid_stoch_slip_trapz_40shells", "stochastic_Slip_Trapz", mixed(0, 40), 1, dict(kT=kT, seed=12)))
  for name, scheme, bodies, n_steps, kw in cases:
    if args.only and args.only != name:
      continue
    case(args.ref, out_dir, name, scheme, bodies, n_steps, a=a, **kw)
  # Round 5: the reference's 42-blob shell (Structures/shell_N_42_Rg_0_8913_Rh_1.vertex; its timing harness uses the 12 /
  # 42 / 162 family, examples/Mobility_Prod_Timing/Multiblob.inputfile:52-55) -- bodies of more than 16 blobs, whose
  # per-body preconditioner factors (multi_bodies.py:752-903) the build keeps in LDS one matrix at a time.  Blob radius =
  # half the smallest blob separation (Mobility_Prod_Timing/main.py:121-126).
  shell42 = read_vertex_file.read_vertex_file(os.path.join(S, "shell_N_42_Rg_0_8913_Rh_1.vertex"))[:, :3]
  d = np.linalg.norm(shell42[:, None, :] - shell42[None, :, :], axis=2)
  a42 = float(np.min(d[d > 0]) / 2)
  rng42 = np.random.RandomState(4242)
  loc42 = np.array([[2.6 * (k % 3) + 0.2 * rng42.rand(), 2.6 * (k // 3) + 0.2 * rng42.rand(), 1.3 + 0.5 * rng42.rand()] for k in range(6)])
  bodies42 = [("shell42", shell42, loc42, random_quaternions(rng42, 6))]
  for name, scheme, n_steps, kw in (("g9_rigid_det_euler_42blob_shells", "deterministic_forward_euler", 2, {}),
                                    ("g9_rigid_stoch_slip_trapz_42blob_shells", "stochastic_Slip_Trapz", 1, dict(kT=kT, seed=13))):
    if args.only and args.only != name:
      continue
    case(args.ref, out_dir, name, scheme, bodies42, n_steps, a=a42, **kw)


if __name__ == "__main__":
  main()
