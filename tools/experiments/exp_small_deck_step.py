"""Whole time steps of the rigid-multiblob integrators on small decks (the reference's usual sizes): ms per step with
the round's solver-loop work on (library helpers + captured Arnoldi iterations) and off (torch operations, eager loop).
  python tools/experiments/exp_small_deck_step.py [n_bodies ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import structures as st
from rigidmultiblobswall_amd.rigid_integrator import RigidIntegrator
R, eta = 1.0155, 0.957e-3
shell = st.icosahedron_shell(0.792079207921 * R)
a = st.min_blob_separation(shell) / 2
for nb in [int(x) for x in sys.argv[1:]] or [8, 64, 256]:
  loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=5)
  for scheme, tol, steps in (("deterministic_forward_euler", 1e-8, 40), ("deterministic_adams_bashforth", 1e-8, 40),
                             ("stochastic_first_order_RFD", 1e-6, 12), ("stochastic_Slip_Trapz", 1e-6, 12)):
    row = []
    for on in (False, True):
      os.environ["RMB_GMRES_GRAPH"] = "" if on else "0"
      os.environ["RMB_NATIVE_HELPERS"] = "" if on else "0"
      integ = RigidIntegrator([shell] * nb, loc, quat, scheme, a, eta, tolerance=tol, device="cuda:0", seed=9)
      integ.kT, integ.g = 0.0041419464, 0.0024892 * 12
      integ.repulsion_strength_wall, integ.debye_length_wall = 0.0165677856, 0.0656
      integ.repulsion_strength, integ.debye_length = 0.0165677856, 0.0656
      for step in range(4): integ.advance_time_step(0.002, step=step)
      torch.cuda.synchronize(); t0 = time.perf_counter()
      for step in range(4, 4 + steps): integ.advance_time_step(0.002, step=step)
      torch.cuda.synchronize()
      row.append(1e3 * (time.perf_counter() - t0) / steps)
      its = (integ.det_iterations_count, integ.stoch_iterations_count)
      integ.close()
    print("bodies %4d %-30s torch ops + eager loop %8.3f ms/step | helpers + graphs %8.3f ms/step (x%.2f)   iterations det %d stoch %d"
          % (nb, scheme, row[0], row[1], row[0] / row[1], its[0], its[1]), flush=True)
