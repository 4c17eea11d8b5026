"""The library's O(N) helpers against the torch operations at the sizes where the sweep dominates: GMRES mobility solves
of 1024 / 2048 / 4096 shells (eager loop; graphs are off above 4096 blobs), same box, alternating."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import structures as st
from rigidmultiblobswall_amd.rigid import RigidSuspension
dev = torch.device("cuda:0")
R, eta3 = 1.0155, 0.957e-3
shell = st.icosahedron_shell(0.792079207921 * R)
a3 = st.min_blob_separation(shell) / 2
for nb in [int(x) for x in sys.argv[1:]] or [1024, 2048, 4096]:
  loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=5)
  FT = np.zeros((nb, 6)); FT[:, 2] = -0.05; FT[:, 4] = 1.0
  rs = {}
  for native in (False, True):
    s = RigidSuspension([shell] * nb, loc, quat, a3, eta3, device=dev)
    s.native_helpers = native
    s.set_configuration(loc, quat); s.build_preconditioner()
    for _ in range(3): s.solve_mobility_problem(force_torque=FT, tol=1e-8)
    rs[native] = s
  res = {False: [], True: []}
  for rnd in range(3):
    for native in (False, True):
      torch.cuda.synchronize(); t0 = time.perf_counter()
      for _ in range(5): U, lam, info = rs[native].solve_mobility_problem(force_torque=FT, tol=1e-8)
      torch.cuda.synchronize()
      res[native].append(1e3 * (time.perf_counter() - t0) / 5)
  v = torch.randn(3 * 12 * nb, dtype=torch.float64, device=dev)
  for _ in range(5): rs[True].ctx.matvec_device("tt", v, eta3)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(20): rs[True].ctx.matvec_device("tt", v, eta3)
  torch.cuda.synchronize()
  mv = 1e3 * (time.perf_counter() - t0) / 20
  print("bodies %5d (%d iterations, sweep %.3f ms): torch ops %s ms | helpers %s ms per solve" %
        (nb, info["iterations"], mv, " ".join("%.2f" % x for x in res[False]), " ".join("%.2f" % x for x in res[True])), flush=True)
  for s in rs.values(): s.close()
