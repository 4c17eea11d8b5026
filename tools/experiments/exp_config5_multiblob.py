"""usage: exp_config5_multiblob.py [bodies] [steps] [tol] [rfd_tol|-] [lockstep 0/1] [warm_start 0/1] [lockstep_width] [single|double]
configs[4] recipe with rigid multiblobs: 21845 shells x 12 blobs = 262140 blobs, stochastic_Slip_Trapz steps
(physical parameters of examples/Spectral_Multiblob_Roller/inputfile_2048_rollers.dat).  Prints per-step timing."""
import math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rigidmultiblobswall_amd import structures as st
from rigidmultiblobswall_amd.rigid_integrator import RigidIntegrator

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 21845
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4
R, eta = 1.0155, 0.957e-3
shell = st.icosahedron_shell(0.792079207921 * R)
a = st.min_blob_separation(shell) / 2
loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=5)
t0 = time.perf_counter()
integ = RigidIntegrator([shell] * nb, loc, quat, "stochastic_Slip_Trapz", a, eta, tolerance=tol, device="cuda:0", seed=1)
integ.kT, integ.g = 0.0040749841, 0.0303 / 12
integ.repulsion_strength_wall = integ.repulsion_strength = 0.0326
integ.debye_length_wall = integ.debye_length = 0.0406
torque = 8 * math.pi * eta * R ** 3 * 62.8
FT = torch.zeros((nb, 6), dtype=torch.float64, device="cuda:0"); FT[:, 4] = torque
integ.external_force_torque = lambda it: FT
if len(sys.argv) > 4 and sys.argv[4] != "-":
  integ.rfd_solve_tolerance = float(sys.argv[4])
if len(sys.argv) > 5:
  integ.lockstep_solves = bool(int(sys.argv[5]))
if len(sys.argv) > 6:
  integ.warm_start = bool(int(sys.argv[6]))
if len(sys.argv) > 7:
  integ.susp.lockstep_width = int(sys.argv[7])
if len(sys.argv) > 8:
  integ.precision = sys.argv[8]
torch.cuda.synchronize()
print("setup %.2f s, blobs %d" % (time.perf_counter() - t0, integ.Nblobs), flush=True)
for step in range(steps):
  d0, s0, m0, p0 = integ.det_iterations_count, integ.stoch_iterations_count, integ.susp.matvec_count, integ.susp.sweep_count
  t0 = time.perf_counter()
  integ.advance_time_step(0.01, step=step)
  torch.cuda.synchronize()
  print("step %d: %.3f s, gmres its %d, lanczos its %d, M.v products %d in %d passes over the pairs, rejected %d" %
        (step, time.perf_counter() - t0, integ.det_iterations_count - d0, integ.stoch_iterations_count - s0,
         integ.susp.matvec_count - m0, integ.susp.sweep_count - p0, integ.invalid_configuration_count), flush=True)
