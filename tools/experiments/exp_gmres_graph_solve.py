"""Whole GMRES mobility solves with the captured Arnoldi iterations (RigidSuspension.gmres_graph) against the eager
loop: ms per solve over a sequence of solves with moving bodies, as a time integrator issues them."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import structures as st
from rigidmultiblobswall_amd.rigid import RigidSuspension
dev = torch.device("cuda:0")
R, eta3 = 1.0155, 0.957e-3
shell = st.icosahedron_shell(0.792079207921 * R)
a3 = st.min_blob_separation(shell) / 2
for nb in [int(x) for x in sys.argv[1:]] or [2, 8, 64, 256, 512, 1024]:
  loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=5)
  FT = np.zeros((nb, 6)); FT[:, 2] = -0.05; FT[:, 4] = 1.0
  row = []
  sols = {}
  for mode in (False, True):
    rs = RigidSuspension([shell] * nb, loc, quat, a3, eta3, device=dev)
    rs.gmres_graph = mode
    rng = np.random.RandomState(1)
    l = loc.copy()
    for _ in range(4): U, lam, info = rs.solve_mobility_problem(force_torque=FT, tol=1e-8)
    torch.cuda.synchronize(dev)
    reps = 30
    t0 = time.perf_counter()
    for k in range(reps):
      l = l + 1e-3 * rng.randn(*l.shape) * np.array([1, 1, 0.1])
      rs.set_configuration(l, quat); rs.build_preconditioner()
      U, lam, info = rs.solve_mobility_problem(force_torque=FT, tol=1e-8)
    torch.cuda.synchronize(dev)
    ms_step = 1e3 * (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for k in range(reps):
      U, lam, info = rs.solve_mobility_problem(force_torque=FT, tol=1e-8)
    torch.cuda.synchronize(dev)
    ms_solve = 1e3 * (time.perf_counter() - t0) / reps
    sols[mode] = U
    row.append((ms_step, ms_solve, info["iterations"], info.get("graph_replays")))
    rs.close()
  (se, ve, it, _), (sg, vg, it2, rep) = row
  print("bodies %5d blobs %6d  %2d iterations: solve eager %7.3f ms  graphed %7.3f ms (x%.2f) | move + preconditioner + solve eager %7.3f ms graphed %7.3f ms (x%.2f) | replays %s, U rel diff %.1e"
        % (nb, 12 * nb, it, ve, vg, ve / vg, se, sg, se / sg, rep, np.linalg.norm(sols[True] - sols[False]) / np.linalg.norm(sols[False])), flush=True)
