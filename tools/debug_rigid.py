import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rigidmultiblobswall_amd.rigid import RigidSuspension
from oracle import oracle
d = np.load("tests/golden/g7_rigid_suspension.npz")
refs = [d["shell"] if s else d["boomerang"] for s in d["body_is_shell"]]
eta, a = float(d["eta"]), float(d["a"])
rs = RigidSuspension(refs, d["locations"], d["quaternions"], a, eta)
N = rs.n_blobs
M = oracle.dense("tt", 1, d["r_vectors"], eta, a)
K = d["K"]
A = np.block([[M, -K], [-K.T, np.zeros((K.shape[1], K.shape[1]))]])
rng = np.random.RandomState(0)
x = rng.randn(A.shape[0])
y = rs.apply_operator(torch.as_tensor(x, device="cuda")).cpu().numpy()
print("operator err", np.linalg.norm(y - A @ x) / np.linalg.norm(A @ x))
lam = rng.randn(3 * N)
u = rs.mobility_times_lambda(torch.as_tensor(lam, device="cuda")).cpu().numpy()
print("matvec err", np.linalg.norm(u - M @ lam) / np.linalg.norm(M @ lam))
rs.build_preconditioner()
z = rs.apply_preconditioner(torch.as_tensor(x, device="cuda")).cpu().numpy()
# dense PC reference
zr = np.zeros_like(x)
off = 0
for k in range(rs.n_bodies):
  nb = len(refs[k]); sl = slice(3 * off, 3 * (off + nb))
  Mb = M[sl, sl]; Kb = K[sl, 6 * k:6 * k + 6]
  Nb = np.linalg.inv(Kb.T @ np.linalg.solve(Mb, Kb))
  slip = x[sl]; F = x[3 * N + 6 * k:3 * N + 6 * k + 6]
  Lt = np.linalg.solve(Mb, slip); Y = Nb @ (-F - Kb.T @ Lt)
  zr[sl] = np.linalg.solve(Mb, slip + Kb @ Y); zr[3 * N + 6 * k:3 * N + 6 * k + 6] = Y
  off += nb
print("pc err", np.linalg.norm(z - zr) / np.linalg.norm(zr))
U, lam2, info = rs.solve_mobility_problem(slip=d["slip"], force_torque=d["force_torque"], tol=1e-10)
print(info["iterations"], info["residual"], info["history"][-3:])
sol = np.concatenate([lam2.reshape(-1), U.reshape(-1)])
rhs = np.concatenate([d["slip"].reshape(-1), -d["force_torque"].reshape(-1)])
print("true residual (dense A)", np.linalg.norm(A @ sol - rhs) / np.linalg.norm(rhs))
print("U err", np.linalg.norm(U.reshape(-1) - d["velocities"]) / np.linalg.norm(d["velocities"]))
print("current_stream handle:", torch.cuda.current_stream().cuda_stream, "default:", torch.cuda.default_stream().cuda_stream)
orig = rs.mobility_times_lambda
def synced(lam):
  torch.cuda.synchronize()
  u = orig(lam)
  torch.cuda.synchronize()
  return u
rs.mobility_times_lambda = synced
U, lam2, info = rs.solve_mobility_problem(slip=d["slip"], force_torque=d["force_torque"], tol=1e-10)
print("with syncs: U err", np.linalg.norm(U.reshape(-1) - d["velocities"]) / np.linalg.norm(d["velocities"]), info["iterations"])
