"""Golden vectors for the Lanczos stochastic forcing, from the REFERENCE's
stochastic_forcing/stochastic_forcing.py (:112-264 Lanczos, :7-60 dense eigen-decomposition variant).
Build-container only; writes tests/golden/g6_lanczos.npz."""
import os
import sys
import warnings

import numpy as np

REF = "/root/reference"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import load_reference  # noqa: E402


def main():
  out_dir = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
  mob, _ = load_reference(REF)
  warnings.simplefilter("ignore")
  from stochastic_forcing import stochastic_forcing as sf
  rng = np.random.RandomState(66)
  N, eta, a = 60, 1.1, 0.35
  r = rng.rand(N, 3) * 6.0
  r[:, 2] += 1.05 * a
  M = mob.single_wall_fluid_mobility(r, eta, a)
  z = rng.randn(3 * N)
  out = dict(r_vectors=r, eta=eta, a=a, z=z)
  for tol in (1e-6, 1e-10):
    noise, its = sf.stochastic_forcing_lanczos(factor=0.7, tolerance=tol, dim=3 * N, mobility=M, z=z)
    out["noise_tol%g" % tol] = noise
    out["iterations_tol%g" % tol] = its
    print("  tol %g: %d iterations" % (tol, its))
  # exact M^{1/2} z via eigen-decomposition (same algebra as stochastic_forcing_eig_symm, with our z)
  w, Q = np.linalg.eigh(M)
  out["noise_exact"] = 0.7 * (Q @ (np.sqrt(np.maximum(w, 0)) * (Q.T @ z)))
  print("  lanczos(1e-10) vs exact:", np.linalg.norm(out["noise_tol1e-10"] - out["noise_exact"]) / np.linalg.norm(out["noise_exact"]))
  np.savez_compressed(os.path.join(out_dir, "g6_lanczos.npz"), **out)


if __name__ == "__main__":
  main()
