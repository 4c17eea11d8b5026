import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rigidmultiblobswall_amd import structures as st
from rigidmultiblobswall_amd.rigid import RigidSuspension
R, eta = 1.0155, 0.957e-3
shell = st.icosahedron_shell(0.792079207921 * R)
a = st.min_blob_separation(shell) / 2
n_bodies = 2048
loc, q, _ = st.roller_monolayer(n_bodies, radius=R, seed=5)
rs = RigidSuspension([shell] * n_bodies, loc, q, a, eta)
rs.ctx.set_option("deterministic", 1)
g = rs.groups[0]
def mx(a, b): return float((a - b).abs().max())
M0 = rs.ctx.body_mobility_dense_device(g.first_blob, g.n_b, eta).clone(); torch.cuda.synchronize()
for k in range(5):
  M1 = rs.ctx.body_mobility_dense_device(g.first_blob, g.n_b, eta)
  print("dense repeat", k, mx(M0, M1))
print("dense symmetric?", float((M0 - M0.transpose(1, 2)).abs().max()), "min diag", float(torch.diagonal(M0, dim1=1, dim2=2).min()))
L0 = torch.linalg.cholesky(M0); torch.cuda.synchronize()
for k in range(3):
  print("chol repeat", k, mx(L0, torch.linalg.cholesky(M0)))
print("chol recon", float((L0 @ L0.transpose(1, 2) - M0).abs().max()))
rs.build_preconditioner(); torch.cuda.synchronize()
x = torch.randn(rs.size, dtype=torch.float64, device="cuda")
p0 = rs.apply_preconditioner(x).clone(); torch.cuda.synchronize()
for k in range(5):
  print("pc repeat", k, mx(p0, rs.apply_preconditioner(x)))
a0 = rs.apply_operator(x).clone(); torch.cuda.synchronize()
for k in range(5):
  print("op repeat", k, mx(a0, rs.apply_operator(x)))
# consistency: A(P(x)) repeated with fresh temporaries
c0 = rs.apply_operator(rs.apply_preconditioner(x)).clone(); torch.cuda.synchronize()
for k in range(10):
  junk = torch.randn(100000 + k * 1000, device="cuda")
  print("AP repeat", k, mx(c0, rs.apply_operator(rs.apply_preconditioner(x))))
# cholesky_solve repeat
slip = torch.randn(n_bodies, 36, 1, dtype=torch.float64, device="cuda")
s0 = torch.cholesky_solve(slip, g.Lchol).clone(); torch.cuda.synchronize()
for k in range(5):
  t1 = torch.cholesky_solve(slip, g.Lchol); t2 = torch.cholesky_solve(slip * 2, g.Lchol)
  print("potrs back-to-back", k, mx(s0, t1), mx(2 * s0, t2))
print("potrs residual", float((M0 @ s0 - slip).abs().max()))
