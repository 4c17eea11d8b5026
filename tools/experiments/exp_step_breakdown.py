"""Where a deterministic_forward_euler step of a small deck goes: each piece timed with a device synchronisation around it."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import structures as st
from rigidmultiblobswall_amd.rigid_integrator import RigidIntegrator
R, eta = 1.0155, 0.957e-3
shell = st.icosahedron_shell(0.792079207921 * R)
a = st.min_blob_separation(shell) / 2
for nb in [int(x) for x in sys.argv[1:]] or [64, 2048]:
  loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=5)
  integ = RigidIntegrator([shell] * nb, loc, quat, "deterministic_forward_euler", a, eta, tolerance=1e-8, device="cuda:0", seed=9)
  integ.g = 0.0024892 * 12
  integ.repulsion_strength_wall, integ.debye_length_wall = 0.0165677856, 0.0656
  integ.repulsion_strength, integ.debye_length = 0.0165677856, 0.0656
  for step in range(4): integ.advance_time_step(0.002, step=step)
  acc = {}
  def timed(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return out
  reps = 20
  for k in range(reps):
    timed("set_configuration", lambda: integ._move(integ.location, integ.orientation))
    timed("build_preconditioner", lambda: integ.susp.build_preconditioner())
    rhs = timed("forces + rhs", lambda: integ._assemble_rhs())
    sol = timed("solve", lambda: integ.susp.solve(rhs, tol=1e-8)[0])
    new = timed("advance (quaternion update)", lambda: integ._advance(integ.location, integ.orientation, sol[3 * integ.Nblobs:], 0.002))
    ok = timed("valid (blob heights)", lambda: integ._valid(*new))
    timed("accept (set_configuration)", lambda: integ._accept(*new))
  print("bodies %d:" % nb, "  ".join("%s %.3f ms" % (k, 1e3 * v / reps) for k, v in acc.items()), " | sum %.3f ms" % (1e3 * sum(acc.values()) / reps), flush=True)
  integ.close()
