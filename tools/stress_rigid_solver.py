import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rigidmultiblobswall_amd import structures as st, MobilityContext
from rigidmultiblobswall_amd.rigid import RigidSuspension
R, eta = 1.0155, 0.957e-3
shell = st.icosahedron_shell(0.792079207921 * R)
a = st.min_blob_separation(shell) / 2
n_bodies = 2048
loc, q, _ = st.roller_monolayer(n_bodies, radius=R, seed=5)
FT = np.zeros((n_bodies, 6)); FT[:, 2] = -0.05; FT[:, 4] = 1.0
for mode in ("sym", "det", "sym+sync"):
  bad = 0
  for rep in range(8):
    rs = RigidSuspension([shell] * n_bodies, loc, q, a, eta)
    if mode == "det":
      rs.ctx.set_option("deterministic", 1)
    if mode == "sym+sync":
      orig = rs.mobility_times_lambda
      def synced(lam, orig=orig):
        torch.cuda.synchronize(); u = orig(lam); torch.cuda.synchronize(); return u
      rs.mobility_times_lambda = synced
    U, lam, info = rs.solve_mobility_problem(force_torque=FT, tol=1e-8, maxiter=200)
    print(mode, rep, info["iterations"], "%.2e" % info["residual"], flush=True)
    bad += (not info["converged"])
    rs.close()
  print(mode, "failures:", bad, flush=True)
# raw repeatability of the matvec with torch churn in between
ctx = MobilityContext(0)
r = np.concatenate([shell @ np.eye(3) + l for l in loc])
ctx.set_positions(torch.as_tensor(r.reshape(-1), device="cuda"), a, wall=True)
v = torch.randn(3 * len(r), dtype=torch.float64, device="cuda")
ref = ctx.matvec_device("tt", v, eta).clone()
torch.cuda.synchronize()
worst = 0.0
for k in range(200):
  junk = [torch.randn(1 + (k * 7919) % 50000, device="cuda", dtype=torch.float64) * 2 for _ in range(3)]
  vv = v * 1.0
  u = ctx.matvec_device("tt", vv, eta)
  del vv, junk
  e = float((u - ref).abs().max() / ref.abs().max())
  worst = max(worst, e)
print("matvec repeat worst rel diff:", worst)
