"""A few graphed GMRES solves of a small deck (to be run under rocprofv3 --kernel-trace; tools/experiments/gmres_nodes.sh prints the
node timeline of the last solve: kernel, duration, gap to the previous kernel)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import structures as st
from rigidmultiblobswall_amd.rigid import RigidSuspension
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R, eta3 = 1.0155, 0.957e-3
shell = st.icosahedron_shell(0.792079207921 * R)
a3 = st.min_blob_separation(shell) / 2
loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=5)
FT = np.zeros((nb, 6)); FT[:, 2] = -0.05; FT[:, 4] = 1.0
rs = RigidSuspension([shell] * nb, loc, quat, a3, eta3, device=torch.device("cuda:0"))
for k in range(8):
  U, lam, info = rs.solve_mobility_problem(force_torque=FT, tol=1e-8)
torch.cuda.synchronize()
print("bodies %d iterations %d replays %s" % (nb, info["iterations"], info.get("graph_replays")))
rs.close()
