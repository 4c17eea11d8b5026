"""Workload for `rocprofv3 --kernel-trace --stats`: time steps of the rigid-multiblob integrators on 64 and 256 shells with
the defaults (library helper kernels, captured Arnoldi iterations) -- which kernels a small-deck step is made of.
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_small_deck -- python3 tools/profile_small_deck.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rigidmultiblobswall_amd import structures as st
from rigidmultiblobswall_amd.rigid_integrator import RigidIntegrator
R, eta = 1.0155, 0.957e-3
shell = st.icosahedron_shell(0.792079207921 * R)
a = st.min_blob_separation(shell) / 2
for nb, scheme, tol, steps in ((64, "deterministic_adams_bashforth", 1e-8, 60), (256, "deterministic_adams_bashforth", 1e-8, 60),
                               (64, "stochastic_Slip_Trapz", 1e-6, 20)):
  loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=5)
  integ = RigidIntegrator([shell] * nb, loc, quat, scheme, a, eta, tolerance=tol, device="cuda:0", seed=9)
  integ.kT, integ.g = 0.0041419464, 0.0024892 * 12
  integ.repulsion_strength_wall, integ.debye_length_wall = 0.0165677856, 0.0656
  integ.repulsion_strength, integ.debye_length = 0.0165677856, 0.0656
  for step in range(4): integ.advance_time_step(0.002, step=step)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for step in range(4, 4 + steps): integ.advance_time_step(0.002, step=step)
  torch.cuda.synchronize()
  print("bodies %d %s: %.3f ms per step (%d steps; iterations det %d stoch %d)" %
        (nb, scheme, 1e3 * (time.perf_counter() - t0) / steps, steps, integ.det_iterations_count, integ.stoch_iterations_count), flush=True)
  integ.close()
