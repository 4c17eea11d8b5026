"""Where the HOST time of a captured-iteration GMRES solve goes (cProfile over repeated solves of a 64-body deck)."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import structures as st
from rigidmultiblobswall_amd.rigid import RigidSuspension
R, eta3 = 1.0155, 0.957e-3
shell = st.icosahedron_shell(0.792079207921 * R)
a3 = st.min_blob_separation(shell) / 2
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=5)
rs = RigidSuspension([shell] * nb, loc, quat, a3, eta3, device="cuda:0")
rhs = torch.randn(rs.size, dtype=torch.float64, device="cuda:0")
for _ in range(5): rs.solve(rhs, tol=1e-8)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): x, info = rs.solve(rhs, tol=1e-8)
torch.cuda.synchronize()
print("ms per solve %.3f, iterations %d, replays %s" % (1e3 * (time.perf_counter() - t0) / 100, info["iterations"], info.get("graph_replays")))
pr = cProfile.Profile(); pr.enable()
for _ in range(100): rs.solve(rhs, tol=1e-8)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
rs.close()
