"""Golden vectors for the remaining source->target operators of mobility/mobility_numba.py, from the reference's own
wrappers (mobility/mobility.py:1345-1366 pressure, :1376-1387 and :1432-1442 double layer) under the numba stub.

Pressure above a wall: the reference rescales its running sum by 1/(4 pi) inside the source loop
(mobility_numba.py:1474), so a multi-source call is not a sum over sources.  The fixture therefore holds
  * `p_wall_single_<k>`: the reference's output for ONE source at a time (where the rescaling is harmless), and
  * `p_wall_superposed`: the sum of those single-source outputs -- the linear operator the routine documents;
the reference's multi-source output is stored too (`p_wall_as_written`) for the record, not as a target.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import load_reference  # noqa: E402


def main():
  out_dir = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
  mob, _ = load_reference("/root/reference")
  rng = np.random.RandomState(1332)
  ns, nt = 23, 31
  src = rng.rand(ns, 3) * np.array([3.0, 3.0, 2.0]) + np.array([0, 0, 0.05])
  tgt = rng.rand(nt, 3) * np.array([3.0, 3.0, 2.0]) + np.array([0, 0, 0.05])
  force = rng.randn(ns, 3)
  normals = rng.randn(ns, 3)
  normals /= np.linalg.norm(normals, axis=1)[:, None]
  vector = rng.randn(ns, 3)
  weights = 0.1 + rng.rand(ns)
  out = dict(source=src, target=tgt, force=force, normals=normals, vector=vector, weights=weights, blob_radius=0.17)
  # pressure
  out["p_no_wall"] = np.asarray(mob.no_wall_pressure_Stokeslet_numba(src, tgt, force))
  singles = [np.asarray(mob.single_wall_pressure_Stokeslet_numba(src[k:k + 1], tgt, force[k:k + 1])) for k in range(ns)]
  for k in (0, 7, ns - 1):
    out["p_wall_single_%d" % k] = singles[k]
  out["p_wall_superposed"] = np.sum(singles, axis=0)
  out["p_wall_as_written"] = np.asarray(mob.single_wall_pressure_Stokeslet_numba(src, tgt, force))
  # double layer: unbounded, above a wall, RPY-regularised; plus the operator on its own nodes (diagonal skipped /
  # image self-interaction kept)
  out["dl_no_wall"] = np.asarray(mob.double_layer_source_target_numba(src, tgt, normals, vector, weights))
  out["dl_wall"] = np.asarray(mob.double_layer_source_target_numba(src, tgt, normals, vector, weights, wall=1))
  out["dl_rpy"] = np.asarray(mob.no_wall_double_layer_source_target_numba(src, tgt, normals, vector, weights, 0.17))
  out["dl_no_wall_self"] = np.asarray(mob.double_layer_source_target_numba(src, src, normals, vector, weights))
  out["dl_wall_self"] = np.asarray(mob.double_layer_source_target_numba(src, src, normals, vector, weights, wall=1))
  out["dl_rpy_self"] = np.asarray(mob.no_wall_double_layer_source_target_numba(src, src, normals, vector, weights, 0.17))
  # small host-surface members of mobility.py: per-blob-radius clamp / damping (:87-119), dense products (:711-736),
  # 6 x 6 self mobility of a sphere above the wall (:739-772)
  rad = 0.05 + 0.3 * rng.rand(ns)
  out["radii"] = rad
  out["shift_heights_different_radius"] = mob.shift_heights_different_radius(src, rad)
  B, overlap = mob.damping_matrix_B_different_radius(src, rad)
  out["B_different_radius_diag"] = B.diagonal()
  out["B_different_radius_overlap"] = np.array(overlap)
  hi = src + np.array([0, 0, 0.4])          # every blob above z = a = 0.2
  out["dense_src"] = hi
  out["wall_dense_product"] = mob.single_wall_fluid_mobility_product(hi, force.flatten(), 0.9, 0.2)
  out["no_wall_dense_product"] = mob.no_wall_fluid_mobility_product(hi, force.flatten(), 0.9, 0.2)
  for k, h in enumerate((1.05, 1.7, 6.0)):
    out["self_6x6_h%d" % k] = mob.single_wall_self_mobility_with_rotation(np.array([0.3, -0.2, h * 0.25]), 1.3, 0.25)
    out["self_6x6_h%d_height" % k] = np.array(h * 0.25)
  np.savez_compressed(os.path.join(out_dir, "g11_aux_operators.npz"), **out)
  print("g11_aux_operators.npz written: %d arrays" % len(out))


if __name__ == "__main__":
  main()
