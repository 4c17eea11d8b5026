"""Where does the library's lagged Lanczos loop (one discarded pair sweep per forcing) stop paying against the generic loop (one
host wait per iteration)?  RigidSuspension.stochastic_forcing at growing shell counts, native_lanczos forced on / off."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import structures as st
from rigidmultiblobswall_amd.rigid import RigidSuspension
R, eta3 = 1.0155, 0.957e-3
shell = st.icosahedron_shell(0.792079207921 * R)
a3 = st.min_blob_separation(shell) / 2
for nb in (512, 1024, 1366, 2048, 4096):
  loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=5)
  rs = RigidSuspension([shell] * nb, loc, quat, a3, eta3, device=torch.device("cuda:0"))
  rs.build_preconditioner()
  z = torch.randn(3 * rs.n_blobs, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
  res = {}
  for rnd in range(4):
    for mode in ((True, False) if rnd % 2 == 0 else (False, True)):
      rs.native_lanczos = mode
      for _ in range(3): rs.stochastic_forcing(z, 1.0, tol=1e-6)
      torch.cuda.synchronize(); t0 = time.perf_counter()
      for _ in range(10): noise, its = rs.stochastic_forcing(z, 1.0, tol=1e-6)
      torch.cuda.synchronize()
      res.setdefault(mode, []).append((time.perf_counter() - t0) / 10 * 1e3)
  print("shells %5d (%6d blobs), %d iterations: library loop %.3f ms, generic loop %.3f ms per forcing" % (nb, rs.n_blobs, its, np.median(res[True]), np.median(res[False])), flush=True)
  rs.close()
