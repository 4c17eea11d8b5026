"""Time of RigidSuspension.build_preconditioner (dense body mobilities + the per-body factor kernel) on small decks."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import _lib
if os.environ.get("RMB_AB_LIB"):             # another build of the library (same-box A/B)
  _lib.LIB_PATH = os.path.abspath(os.environ["RMB_AB_LIB"])
from rigidmultiblobswall_amd import structures as st
from rigidmultiblobswall_amd.rigid import RigidSuspension
R, eta3 = 1.0155, 0.957e-3
shell42 = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "g9_rigid_det_euler_42blob_shells.npz"))["vertex_shell42"]
boom = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "g9_rigid_det_euler.npz"))["vertex_boomerang"]
for nb, shell in ((64, st.icosahedron_shell(0.792079207921 * R)), (256, st.icosahedron_shell(0.792079207921 * R)), (64, boom), (32, shell42), (256, shell42)):
  a3 = st.min_blob_separation(shell) / 2
  loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=5)
  rs = RigidSuspension([shell] * nb, loc, quat, a3, eta3, device=torch.device("cuda:0"))
  for _ in range(20): rs.build_preconditioner()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(200): rs.build_preconditioner()
  torch.cuda.synchronize()
  g = rs.groups[0]
  chk = float(torch.linalg.norm(torch.bmm(g.Nbody, torch.linalg.inv(g.Nbody)) - torch.eye(6, dtype=torch.float64, device="cuda")))
  print("bodies %4d x %2d blobs: build_preconditioner %.1f us   (N N^-1 - I: %.1e, |N| %.6e)" % (nb, shell.shape[0], (time.perf_counter() - t0) / 200 * 1e6, chk,
        float(torch.linalg.norm(g.Nbody))), flush=True)
