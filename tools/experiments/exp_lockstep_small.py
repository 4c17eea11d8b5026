"""stochastic_Slip_Trapz steps on small decks: solves advanced in lockstep (k-vector passes, Python coroutines) against
sequential solves through the one-call Arnoldi step."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import structures as st
from rigidmultiblobswall_amd.rigid_integrator import RigidIntegrator
R, eta_s = 1.0155, 0.957e-3
shell = st.icosahedron_shell(0.792079207921 * R)
a_s = st.min_blob_separation(shell) / 2
for nb in [int(x) for x in sys.argv[1:]] or [16, 64, 256, 1024, 2048]:
  loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=5)
  row = {}
  for lock in (True, False, True, False):
    integ = RigidIntegrator([shell] * nb, loc, quat, "stochastic_Slip_Trapz", a_s, eta_s, tolerance=1e-6, device="cuda:0", seed=9)
    integ.kT, integ.g = 0.0041419464, 0.0024892 * 12
    integ.repulsion_strength_wall, integ.debye_length_wall = 0.0165677856, 0.0656
    integ.repulsion_strength, integ.debye_length = 0.0165677856, 0.0656
    integ.lockstep_solves = lock
    for step in range(3): integ.advance_time_step(0.002, step=step)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10 if nb <= 256 else 4
    for step in range(3, 3 + n): integ.advance_time_step(0.002, step=step)
    torch.cuda.synchronize()
    row.setdefault(lock, []).append(((time.perf_counter() - t0) / n * 1e3, integ.det_iterations_count / (3 + n), integ.stoch_iterations_count / (3 + n), integ.susp.sweep_count))
    integ.close()
  f = lambda v: "%.3f ms (%.1f GMRES + %.1f Lanczos its, %d passes)" % (min(x[0] for x in v), v[0][1], v[0][2], v[0][3])
  print("bodies %5d (%6d blobs): lockstep %s | sequential %s" % (nb, 12 * nb, f(row[True]), f(row[False])), flush=True)
