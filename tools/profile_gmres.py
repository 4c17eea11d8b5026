"""Breakdown of the config-3 rigid solve (2048 x 12-blob shells): matvec vs preconditioner vs K products vs GMRES bookkeeping."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rigidmultiblobswall_amd import structures as st
from rigidmultiblobswall_amd.rigid import RigidSuspension
R, eta = 1.0155, 0.957e-3
shell = st.icosahedron_shell(0.792079207921 * R)
a = st.min_blob_separation(shell) / 2
n_bodies = 2048
loc, q, _ = st.roller_monolayer(n_bodies, radius=R, seed=5)
FT = np.zeros((n_bodies, 6)); FT[:, 2] = -0.05; FT[:, 4] = 1.0
rs = RigidSuspension([shell] * n_bodies, loc, q, a, eta)
def timed(fn, *args, reps=20):
  for _ in range(3): fn(*args)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(reps): out = fn(*args)
  torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
t0 = time.perf_counter(); rs.build_preconditioner(); torch.cuda.synchronize(); t_pc_build = (time.perf_counter() - t0) * 1e3
x = torch.randn(rs.size, dtype=torch.float64, device="cuda")
n3 = 3 * rs.n_blobs
print("build_preconditioner   %.3f ms (once per configuration)" % t_pc_build)
print("M.lambda (HIP matvec)  %.3f ms" % timed(rs.mobility_times_lambda, x[:n3].contiguous()))
print("K.U                    %.3f ms" % timed(rs.K_times_U, x[n3:]))
print("K^T.lambda             %.3f ms" % timed(rs.KT_times_lambda, x[:n3]))
print("apply_operator         %.3f ms" % timed(rs.apply_operator, x))
print("apply_preconditioner   %.3f ms" % timed(rs.apply_preconditioner, x))
# host bookkeeping of GMRES one iteration behind the device (round 3) against the synchronous loop of rounds 1-2
for lag in (False, True, False, True):
  rs.gmres_lag = lag
  rs.solve_mobility_problem(force_torque=FT, tol=1e-8)
  ts = []
  for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    U, lam, info = rs.solve_mobility_problem(force_torque=FT, tol=1e-8)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
  print("solve, host bookkeeping %s: %d iterations (%d sweeps discarded), residual %.2e, %.2f ms (min of 5; all: %s), %.3f ms/iteration" %
        ("one iteration late" if lag else "synchronous       ", info["iterations"], info.get("discarded_sweeps", 0), info["residual"],
         min(ts), " ".join("%.2f" % t for t in ts), min(ts) / info["iterations"]))
