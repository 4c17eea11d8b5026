"""Kernel trace target: GMRES solves of argv[1] 12-blob shells (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import structures as st
from rigidmultiblobswall_amd.rigid import RigidSuspension
R, eta3 = 1.0155, 0.957e-3
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
shell = st.icosahedron_shell(0.792079207921 * R)
a3 = st.min_blob_separation(shell) / 2
loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=5)
FT = np.zeros((nb, 6)); FT[:, 2] = -0.05; FT[:, 4] = 1.0
rs = RigidSuspension([shell] * nb, loc, quat, a3, eta3, device=torch.device("cuda:0"))
rhs = rs.prescribe(torch.cat([torch.zeros(3 * rs.n_blobs, dtype=torch.float64, device="cuda"), -torch.as_tensor(FT.reshape(-1), device="cuda")]))
for _ in range(30): rs.solve(rhs, tol=1e-8)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): sol, info = rs.solve(rhs, tol=1e-8)
torch.cuda.synchronize()
print("bodies %d x 12: %.3f ms per solve, %d iterations, launch %s" % (nb, (time.perf_counter() - t0) / 100 * 1e3, info["iterations"], rs.ctx.last_launch()))
