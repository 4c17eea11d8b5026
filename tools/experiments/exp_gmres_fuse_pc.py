"""Same-run A/B of "gmres_fuse_pc" (normalisation + next step's preconditioner in one launch) and "gmres_fuse_dots" (operator
finish + first Gram-Schmidt dots in one launch) on small-deck solves."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import structures as st
from rigidmultiblobswall_amd.rigid import RigidSuspension
R, eta3 = 1.0155, 0.957e-3
shell42 = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "g9_rigid_det_euler_42blob_shells.npz"))["vertex_shell42"]
for nb, shell in ((16, st.icosahedron_shell(0.792079207921 * R)), (64, st.icosahedron_shell(0.792079207921 * R)), (256, st.icosahedron_shell(0.792079207921 * R)), (24, shell42)):
  a3 = st.min_blob_separation(shell) / 2
  loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=5)
  FT = np.zeros((nb, 6)); FT[:, 2] = -0.05; FT[:, 4] = 1.0
  rs = RigidSuspension([shell] * nb, loc, quat, a3, eta3, device=torch.device("cuda:0"))
  rhs = rs.prescribe(torch.cat([torch.zeros(3 * rs.n_blobs, dtype=torch.float64, device="cuda"), -torch.as_tensor(FT.reshape(-1), device="cuda")]))
  sols = {}
  modes = ((0, 0), (1, 0), (1, 1))
  res = {m: [] for m in modes}
  for rnd in range(6):
    for flag in (modes if rnd % 2 == 0 else modes[::-1]):
      rs.ctx.set_option("gmres_fuse_pc", flag[0]); rs.ctx.set_option("gmres_fuse_dots", flag[1])
      for _ in range(20): rs.solve(rhs, tol=1e-8)
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for _ in range(100): sol, info = rs.solve(rhs, tol=1e-8)
      torch.cuda.synchronize()
      res[flag].append((time.perf_counter() - t0) / 100 * 1e3)
      sols[flag] = (sol.clone(), info["iterations"])
  d = max(float((sols[m][0] - sols[(0, 0)][0]).abs().max() / sols[(0, 0)][0].abs().max()) for m in modes)
  print("bodies %4d x %d blobs: separate %.3f ms, +pc %.3f ms, +pc +dots %.3f ms per solve (%s iterations), solutions differ by %.1e"
        % (nb, shell.shape[0], np.median(res[(0, 0)]), np.median(res[(1, 0)]), np.median(res[(1, 1)]), "/".join(str(sols[m][1]) for m in modes), d), flush=True)
