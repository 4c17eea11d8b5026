"""Small decks (the reference's usual sizes): time of one GMRES mobility solve per iteration against the time of the
blob product inside it -- how launch-bound is the solver loop of rigid.py when the O(N^2) sweep takes ~10 us?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import structures as st
from rigidmultiblobswall_amd.rigid import RigidSuspension
dev = torch.device("cuda:0")
R, eta3 = 1.0155, 0.957e-3
shell = st.icosahedron_shell(0.792079207921 * R)
a3 = st.min_blob_separation(shell) / 2
for nb in [int(x) for x in sys.argv[1:]] or [2, 8, 64, 256, 1024]:
  loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=5)
  FT = np.zeros((nb, 6)); FT[:, 2] = -0.05; FT[:, 4] = 1.0
  rs = RigidSuspension([shell] * nb, loc, quat, a3, eta3, device=dev)
  for _ in range(3): rs.solve_mobility_problem(force_torque=FT, tol=1e-8)
  torch.cuda.synchronize(dev)
  reps = 20
  t0 = time.perf_counter()
  for _ in range(reps): U, lam, info = rs.solve_mobility_problem(force_torque=FT, tol=1e-8)
  torch.cuda.synchronize(dev)
  ms = 1e3 * (time.perf_counter() - t0) / reps
  # the blob product alone, resident, back to back
  v = torch.randn(3 * rs.n_blobs, dtype=torch.float64, device=dev)
  ctx = rs.ctx if hasattr(rs, "ctx") else rs.backend.ctx
  for _ in range(20): ctx.matvec_device("tt", v, eta3)
  torch.cuda.synchronize(dev); t0 = time.perf_counter()
  for _ in range(200): ctx.matvec_device("tt", v, eta3)
  torch.cuda.synchronize(dev)
  mv = 1e6 * (time.perf_counter() - t0) / 200
  print("bodies %5d blobs %6d: %8.3f ms per solve, %3d iterations -> %7.1f us per iteration; blob product alone %6.1f us"
        % (nb, rs.n_blobs, ms, info["iterations"], 1e3 * ms / max(info["iterations"], 1), mv), flush=True)
  rs.close()
