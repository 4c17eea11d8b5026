"""Golden vectors for the source->target kernels with per-blob radii (K13), from the reference's
mobility/mobility.py:551-615 wrappers over mobility_numba.py:1480-1658 (numba stub, interpreted).
Inputs follow mobility/test_source_target.py:12-24 (N_src = 10, N_trg = 12, eta = 0.13, radii ~ 0.97)
plus a larger mixed-radius cloud with overlaps, wall-overlapping blobs and a pseudo-periodic case."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import load_reference  # noqa: E402


def main():
  out_dir = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
  mob, _ = load_reference("/root/reference")
  cases = {}
  rng = np.random.RandomState(91)
  # test_source_target.py-like
  ns, nt, eta = 10, 12, 0.13
  cases["small"] = dict(source=rng.rand(ns, 3) * 5 + np.array([0, 0, 1.0]), target=rng.rand(nt, 3) * 5 + np.array([0, 0, 1.0]),
                        force=rng.randn(ns, 3), radius_source=0.97 * (1 + 0.1 * rng.rand(ns)),
                        radius_target=0.97 * (1 + 0.1 * rng.rand(nt)), eta=eta, L=np.zeros(3))
  # mixed radii, overlapping pairs (all three regimes of the unbounded part), blobs below z = radius
  ns, nt = 70, 55
  src = rng.rand(ns, 3) * np.array([4.0, 4.0, 2.5])
  tgt = rng.rand(nt, 3) * np.array([4.0, 4.0, 2.5])
  tgt[:5] = src[:5] + 1e-3          # target inside a source blob
  tgt[5] = src[5]                    # coincident
  cases["mixed"] = dict(source=src, target=tgt, force=rng.randn(ns, 3), radius_source=0.1 + 0.6 * rng.rand(ns),
                        radius_target=0.05 + 0.7 * rng.rand(nt), eta=1.7, L=np.zeros(3))
  cases["periodic"] = dict(cases["mixed"], L=np.array([4.5, 5.0, 0.0]))
  out = {}
  for name, c in cases.items():
    for wall, fn in ((1, mob.single_wall_mobility_trans_times_force_source_target_numba),
                     (0, mob.no_wall_mobility_trans_times_force_source_target_numba),
                     (2, mob.free_surface_mobility_trans_times_force_source_target_numba)):   # stress-free surface
      u = fn(c["source"], c["target"], c["force"], c["radius_source"], c["radius_target"], c["eta"], periodic_length=c["L"])
      out["%s_wall%d" % (name, wall)] = np.asarray(u).reshape(-1)
    for k, v in c.items():
      out["%s_%s" % (name, k)] = v
    print("  source_target %s done" % name, flush=True)
  # radii_* mode: same set as source and target (mobility.py:1369-1374)
  c = cases["mixed"]
  u = mob.mobility_radii_trans_times_force(c["source"], c["force"], c["eta"], 0.3, c["radius_source"],
                                           mob.single_wall_mobility_trans_times_force_source_target_numba)
  out["mixed_radii_self_wall1"] = np.asarray(u).reshape(-1)
  np.savez_compressed(os.path.join(out_dir, "g4_source_target.npz"), **out)


if __name__ == "__main__":
  main()
