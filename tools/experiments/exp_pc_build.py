"""Where the preconditioner build spends its time (per-body dense M, Cholesky, inverse, (K^T M^-1 K)^+)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import structures as st
from rigidmultiblobswall_amd.rigid import RigidSuspension
R, eta = 1.0155, 0.957e-3
shell = st.icosahedron_shell(0.792079207921 * R)
a = st.min_blob_separation(shell) / 2
for nb in (2048, 21845):
  loc, q, _ = st.roller_monolayer(nb, radius=R, seed=5)
  rs = RigidSuspension([shell] * nb, loc, q, a, eta)
  rs.build_preconditioner()
  def t(fn, reps=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3, out
  g = rs.groups[0]
  print("bodies %d" % nb)
  ms, _ = t(rs.build_preconditioner); print("  build_preconditioner total %.2f ms" % ms)
  ms, Mb = t(lambda: rs.ctx.body_mobility_dense_device(g.first_blob, g.n_b, rs.eta)); print("  dense blocks   %.2f ms" % ms)
  ms, L = t(lambda: torch.linalg.cholesky(Mb)); print("  cholesky       %.2f ms" % ms)
  ms, Mi = t(lambda: torch.cholesky_inverse(L)); print("  chol inverse   %.2f ms" % ms)
  ms, A = t(lambda: torch.bmm(g.K.transpose(1, 2), torch.bmm(Mi, g.K))); print("  K^T Minv K     %.2f ms" % ms)
  ms, _ = t(lambda: torch.linalg.pinv(A)); print("  pinv 6x6       %.2f ms" % ms)
  ms, _ = t(lambda: torch.linalg.inv(A)); print("  inv 6x6        %.2f ms" % ms)
  ms, _ = t(lambda: torch.linalg.pinv(A, hermitian=True)); print("  pinv hermitian %.2f ms" % ms)
  ms, _ = t(lambda: rs.set_configuration(rs.location, rs.orientation)); print("  set_configuration %.2f ms" % ms)
  rs.close()
