"""Generate golden vectors from the REFERENCE's own Python.  Build-container only.

Runs /root/reference/mobility/mobility.py (+ mobility_numba.py,
multi_bodies/forces_numba.py) on seeded inputs and writes small .npz fixtures
to tests/golden/.  The reference needs `numba`, which this image lacks; its
kernels are plain Python under the decorator, so we put an identity stub for
`numba.njit` / `numba.prange` on sys.path (written to a temp dir at run time)
and the kernels run interpreted.  Nothing from the reference is copied: the
fixtures hold inputs and the reference's outputs only.

Usage:  python oracle/gen_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import os
import sys
import tempfile
import time
import warnings

import numpy as np

STUB = '''
def njit(*args, **kwargs):
  if len(args) == 1 and callable(args[0]) and not kwargs:
    return args[0]
  def deco(fn):
    return fn
  return deco
jit = njit
prange = range
'''


def load_reference(ref):
  stub_dir = tempfile.mkdtemp(prefix="numba_stub_")
  with open(os.path.join(stub_dir, "numba.py"), "w") as fh:
    fh.write(STUB)
  sys.path.insert(0, stub_dir)
  sys.path.insert(0, ref)
  sys.path.insert(0, os.path.join(ref, "mobility"))
  sys.path.insert(0, os.path.join(ref, "multi_bodies"))
  warnings.simplefilter("ignore")
  import mobility as mob          # /root/reference/mobility/mobility.py
  import forces_numba             # /root/reference/multi_bodies/forces_numba.py
  return mob, forces_numba


KERNELS = [
    # name in fixture, reference function name, takes wall clamp?
    ("no_wall_tt", "no_wall_mobility_trans_times_force_numba"),
    ("wall_tt", "single_wall_mobility_trans_times_force_numba"),
    ("no_wall_tr", "no_wall_mobility_trans_times_torque_numba"),
    ("wall_tr", "single_wall_mobility_trans_times_torque_numba"),
    ("no_wall_rt", "no_wall_mobility_rot_times_force_numba"),
    ("wall_rt", "single_wall_mobility_rot_times_force_numba"),
    ("no_wall_rr", "no_wall_mobility_rot_times_torque_numba"),
    ("wall_rr", "single_wall_mobility_rot_times_torque_numba"),
    ("in_plane_tt", "in_plane_mobility_trans_times_force_numba"),
    ("in_plane_tr", "in_plane_mobility_trans_times_torque_numba"),
    ("free_surface_tt", "free_surface_mobility_trans_times_force_numba"),
]


def run_case(mob, name, r, v, eta, a, L, out_dir, kernels=None, dense=False):
  t0 = time.time()
  data = dict(r_vectors=r, vector=v, eta=eta, a=a, periodic_length=L)
  for key, fn in KERNELS:
    if kernels is not None and key not in kernels:
      continue
    data[key] = getattr(mob, fn)(r, v, eta, a, periodic_length=L)
  if dense:
    # dense builders (mobility.py:967-1013, :1018-1116) -- non-periodic only.
    # single_wall_fluid_mobility has no height clamp; only use it when no blob overlaps the wall
    data["dense_no_wall_tt"] = mob.rotne_prager_tensor(r, eta, a) @ v.flatten()
    if np.all(r[:, 2] > a):
      data["dense_wall_tt"] = mob.single_wall_fluid_mobility(r, eta, a) @ v.flatten()
  np.savez_compressed(os.path.join(out_dir, name + ".npz"), **data)
  print("  %-28s N=%-5d %.1fs" % (name, len(r), time.time() - t0), flush=True)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--ref", default="/root/reference")
  ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
  args = ap.parse_args()
  out_dir = os.path.abspath(args.out)
  os.makedirs(out_dir, exist_ok=True)
  mob, forces_numba = load_reference(args.ref)
  zero = np.zeros(3)

  # G1: mobility/test_blobs.py:31-44 distribution (dense, overlapping, ~20% below z=a)
  eta, a = 7.0, 0.13
  for N, seed in ((2, 1), (64, 2), (300, 3)):
    rng = np.random.RandomState(seed)
    r = 5 * a * rng.rand(N, 3)
    f = rng.randn(N, 3)
    run_case(mob, "g1_test_blobs_N%d" % N, r, f, eta, a, zero, out_dir, dense=True)
  rng = np.random.RandomState(4)
  N = 1000
  s = 1.0
  r = s * 5 * a * rng.rand(N, 3)
  f = rng.randn(N, 3)
  run_case(mob, "g1_test_blobs_N1000", r, f, eta, a, zero, out_dir, kernels=("no_wall_tt", "wall_tt"))

  # G2: pseudo-periodic in x,y (and one fully periodic no-wall case)
  rng = np.random.RandomState(5)
  N = 64
  Lxy = np.array([1.7, 2.3, 0.0])
  r = rng.rand(N, 3) * np.array([1.7, 2.3, 1.0])
  f = rng.randn(N, 3)
  run_case(mob, "g2_periodic_xy_N64", r, f, eta, a, Lxy, out_dir)
  Lx = np.array([1.1, 0.0, 0.0])
  run_case(mob, "g2_periodic_x_N64", r, f, eta, a, Lx, out_dir)
  Lxyz = np.array([1.7, 2.3, 1.9])
  run_case(mob, "g2_periodic_xyz_N24", r[:24], f[:24], eta, a, Lxyz, out_dir,
           kernels=("no_wall_tt", "no_wall_tr", "no_wall_rt", "no_wall_rr", "free_surface_tt"))

  # G3a: well-separated wall cloud (D2-like: 5% volume fraction, z in [1.1a, 1.1a+Lbox))
  rng = np.random.RandomState(6)
  N, a3, eta3 = 200, 0.5, 1.0
  Lbox = (N * (4.0 / 3.0) * np.pi * a3**3 / 0.05) ** (1.0 / 3.0)
  r = rng.rand(N, 3) * Lbox
  r[:, 2] += 1.1 * a3
  f = rng.randn(N, 3)
  run_case(mob, "g3_wall_cloud_N200", r, f, eta3, a3, zero, out_dir, dense=True)
  # G3b: near-contact / overlapping pairs straddling r = 2a, some blobs below the wall plane z<a, one at z<0
  rng = np.random.RandomState(7)
  N = 48
  r = rng.rand(N, 3) * np.array([2.0, 2.0, 1.5]) * a3 * 3
  r[0, 2] = -0.2 * a3
  r[1] = r[2] + np.array([2.0 * a3, 0, 0])       # exactly r = 2a
  r[3] = r[4] + np.array([0, 1e-3 * a3, 0])       # nearly coincident
  f = rng.randn(N, 3)
  run_case(mob, "g3_contact_N48", r, f, eta3, a3, zero, out_dir)

  # G5: blob-blob forces (multi_bodies/test_force.py:29-34 parameters)
  rng = np.random.RandomState(8)
  N, af, b, eps = 100, 0.13, 0.01, 3.92
  r = 10 * rng.rand(N, 3) * af * 3
  for nm, L in (("g5_forces_N100", np.zeros(3)), ("g5_forces_periodic_N100", np.array([2.5, 3.0, 0.0]))):
    F = forces_numba.calc_blob_blob_forces_numba(r, periodic_length=L, repulsion_strength=eps,
                                                 debye_length=b, blob_radius=af)
    # the k-d tree variant (forces_numba.py:142-271, `blob_blob_force_implementation tree_numba`): pairs beyond
    # 2 a + 30 b are dropped, so it differs from the full sum by e^-30 of a contact force at most
    F_tree = forces_numba.calc_blob_blob_forces_tree_numba(r, periodic_length=L, repulsion_strength=eps,
                                                           debye_length=b, blob_radius=af)
    np.savez_compressed(os.path.join(out_dir, nm + ".npz"), r_vectors=r, periodic_length=L,
                        repulsion_strength=eps, debye_length=b, blob_radius=af, force=F, force_tree=F_tree)
    print("  %s" % nm, flush=True)
  # one radius per blob (forces_numba.py:125-137)
  radii = af * (0.5 + rng.rand(N))
  for nm, L in (("g5_forces_radii_N100", np.zeros(3)), ("g5_forces_radii_periodic_N100", np.array([2.5, 3.0, 0.0]))):
    F = forces_numba.calc_blob_blob_forces_radii_numba(r, radii, periodic_length=L, repulsion_strength=eps,
                                                       debye_length=b, blob_radius=af)
    np.savez_compressed(os.path.join(out_dir, nm + ".npz"), r_vectors=r, radius_blobs=radii, periodic_length=L,
                        repulsion_strength=eps, debye_length=b, blob_radius=af, force=F)
    print("  %s" % nm, flush=True)


if __name__ == "__main__":
  main()
