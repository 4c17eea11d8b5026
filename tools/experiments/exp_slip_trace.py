"""Kernel trace target: stochastic_Slip_Trapz steps of argv[1] 12-blob shells (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import structures as st
from rigidmultiblobswall_amd.rigid_integrator import RigidIntegrator
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
R, eta_s = 1.0155, 0.957e-3
shell = st.icosahedron_shell(0.792079207921 * R)
a_s = st.min_blob_separation(shell) / 2
loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=5)
integ = RigidIntegrator([shell] * nb, loc, quat, "stochastic_Slip_Trapz", a_s, eta_s, tolerance=1e-6, device="cuda:0", seed=9)
integ.kT, integ.g = 0.0041419464, 0.0024892 * 12
integ.repulsion_strength_wall, integ.debye_length_wall = 0.0165677856, 0.0656
integ.repulsion_strength, integ.debye_length = 0.0165677856, 0.0656
for step in range(5): integ.advance_time_step(0.002, step=step)
torch.cuda.synchronize(); t0 = time.perf_counter()
for step in range(5, 45): integ.advance_time_step(0.002, step=step)
torch.cuda.synchronize()
print("stochastic_Slip_Trapz, %d shells: %.3f ms per step (%.1f GMRES, %.1f Lanczos iterations per step)" % (nb, (time.perf_counter() - t0) / 40 * 1e3,
      integ.det_iterations_count / 45.0, integ.stoch_iterations_count / 45.0))
