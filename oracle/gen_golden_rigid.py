"""Golden vectors for the rigid-multiblob mobility problem, from the REFERENCE's own pieces.

The reference driver cannot be imported here (scipy 1.15 removed the private Fortran GMRES it binds,
quaternion_integrator/gmres.py:5-9), but its building blocks can: body.Body (blob coordinates,
K matrix), quaternion.Quaternion, the structure readers and the dense blob mobility
mobility.single_wall_fluid_mobility.  The saddle-point system of
multi_bodies/multi_bodies.py:424-471
    |  M   -K | |lambda|   | slip |
    | -K^T  0 | |  U   | = |  -F  |
is assembled from those pieces and solved DIRECTLY (numpy.linalg.solve); a GMRES solve of the same
system to tolerance tol must reproduce U to ~tol.  Also stores the 6x6 body mobility of
the config-1 boomerang (multi_bodies/inputfile_body_mobility.dat, multi_bodies_utilities.py:583-605).

Build-container only; writes tests/golden/g7_*.npz (inputs + reference outputs, no reference source).
"""
import os
import sys
import warnings

import numpy as np

REF = "/root/reference"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import load_reference  # noqa: E402  (numba stub + sys.path)


def main():
  out_dir = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
  mob, _ = load_reference(REF)
  warnings.simplefilter("ignore")
  from body import body as ref_body
  from quaternion_integrator.quaternion import Quaternion
  from read_input import read_vertex_file, read_clones_file

  S = os.path.join(REF, "multi_bodies", "Structures")
  shell = read_vertex_file.read_vertex_file(os.path.join(S, "shell_N_12_Rg_0_7921_Rh_1.vertex"))
  boom = read_vertex_file.read_vertex_file(os.path.join(S, "boomerang_N_15.vertex"))

  # ---- config 1: single boomerang body mobility (inputfile_body_mobility.dat: eta 1, a 0.25) ----
  nb, locs, oris = read_clones_file.read_clones_file(os.path.join(S, "boomerang_N_15.clones"))
  eta, a = 1.0, 0.25
  b = ref_body.Body(locs[0], oris[0], boom, a)
  r = b.get_r_vectors()
  M = mob.single_wall_fluid_mobility(r, eta, a)
  K = b.calc_K_matrix()
  Nbody = np.linalg.pinv(K.T @ np.linalg.inv(M) @ K)
  np.savez_compressed(os.path.join(out_dir, "g7_boomerang_body_mobility.npz"), reference_configuration=boom,
                      location=locs[0], quaternion=oris[0].entries, eta=eta, a=a, r_vectors=r, K=K,
                      body_mobility=Nbody)
  print("  boomerang N[0,0] = %.7f" % Nbody[0, 0])

  # ---- mixed suspension: 6 shells + 2 boomerangs above the wall, random orientations ----
  rng = np.random.RandomState(77)
  eta, a = 0.9, 0.3
  refs, locations, quats, bodies = [], [], [], []
  for k in range(8):
    ref = shell if k < 6 else boom
    loc = np.array([3.5 * (k % 3) + rng.rand(), 3.5 * (k // 3) + rng.rand(), 2.2 + 1.5 * rng.rand()])
    q = rng.randn(4)
    q /= np.linalg.norm(q)
    bodies.append(ref_body.Body(loc, Quaternion(q), ref, a))
    refs.append(ref); locations.append(loc); quats.append(q)
  r = np.concatenate([bb.get_r_vectors() for bb in bodies])
  assert r[:, 2].min() > a
  N = len(r)
  M = mob.single_wall_fluid_mobility(r, eta, a)
  K = np.zeros((3 * N, 6 * len(bodies)))
  off = 0
  for k, bb in enumerate(bodies):
    K[3 * off:3 * (off + bb.Nblobs), 6 * k:6 * k + 6] = bb.calc_K_matrix()
    off += bb.Nblobs
  slip = 0.1 * rng.randn(N, 3)
  FT = rng.randn(len(bodies), 6)
  A = np.block([[M, -K], [-K.T, np.zeros((6 * len(bodies), 6 * len(bodies)))]])
  rhs = np.concatenate([slip.reshape(-1), -FT.reshape(-1)])
  sol = np.linalg.solve(A, rhs)
  lam, U = sol[:3 * N], sol[3 * N:]
  print("  suspension: N=%d residual=%.2e" % (N, np.linalg.norm(A @ sol - rhs) / np.linalg.norm(rhs)))
  np.savez_compressed(os.path.join(out_dir, "g7_rigid_suspension.npz"), shell=shell, boomerang=boom,
                      body_is_shell=np.array([1] * 6 + [0] * 2), locations=np.array(locations),
                      quaternions=np.array(quats), eta=eta, a=a, r_vectors=r, K=K, slip=slip, force_torque=FT,
                      lambda_blobs=lam, velocities=U)




def pair_active_rods():
  """The reference's only pinned known-answer on this path: multi_bodies/examples/pair_active_rods
  (README.md:37-44: velocities must match run_*_res.velocity.dat.reference to solver_tolerance 1e-8).
  Inputs (structure, clones, slip from the example's slip_function.py, forces from force_*.dat) are
  evaluated with the reference's own code; the expected output is the reference's data file."""
  out_dir = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
  load_reference(REF)
  warnings.simplefilter("ignore")
  from body import body as ref_body
  from read_input import read_vertex_file, read_clones_file
  ex = os.path.join(REF, "multi_bodies", "examples", "pair_active_rods")
  # slip_function.py mixes tabs and spaces (Python 2 era; TabError under Python 3): run it with tabs
  # expanded the way Python 2 read them.  Nothing of it is written to disk.
  import types
  slip_function = types.ModuleType("slip_function")
  src = open(os.path.join(ex, "slip_function.py")).read().expandtabs(8)
  exec(compile(src, "slip_function.py", "exec"), slip_function.__dict__)
  S = os.path.join(REF, "multi_bodies", "Structures")
  cases = {"low": ("Cylinder_N_14_Lg_1_9295_Rg_0_18323", 0.183228708092682),
           "mid": ("Cylinder_N_86_Lg_1_9384_Rg_0_1484", None),
           "high": ("Cylinder_N_324_Lg_2_0299_Rg_0_1554", None)}
  for res, (name, a) in cases.items():
    deck = os.path.join(ex, "inputfile_%s_resolution.dat" % res)
    opts = {}
    for line in open(deck):
      line = line.split("#", 1)[0].strip()
      if line:
        k, v = line.split(None, 1)
        opts[k] = v
    a = float(opts["blob_radius"])
    eta = float(opts["eta"])
    ref_conf = read_vertex_file.read_vertex_file(os.path.join(S, name + ".vertex"))
    nb, locs, oris = read_clones_file.read_clones_file(os.path.join(S, name + ".clones"))
    bodies = [ref_body.Body(locs[k], oris[k], ref_conf, a) for k in range(nb)]
    slip = np.concatenate([slip_function.slip_extensile_rod(b) for b in bodies])
    FT = np.loadtxt(os.path.join(ex, "force_%s_resolution.dat" % res)).reshape(nb, 6)
    expected = np.loadtxt(os.path.join(ex, "run_%s_res.velocity.dat.reference" % res)).reshape(nb, 6)
    r = np.concatenate([b.get_r_vectors() for b in bodies])
    print("  rods %s: %d blobs, z_min/a = %.3f" % (res, len(r), r[:, 2].min() / a))
    np.savez_compressed(os.path.join(out_dir, "g7_pair_active_rods_%s.npz" % res), reference_configuration=ref_conf,
                        locations=locs, quaternions=np.array([o.entries for o in oris]), eta=eta, a=a, slip=slip,
                        force_torque=FT, r_vectors=r, velocities_reference=expected,
                        solver_tolerance=float(opts["solver_tolerance"]))


if __name__ == "__main__":
  main()
  pair_active_rods()
