import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rigidmultiblobswall_amd.rigid import gmres_right_preconditioned
rng = np.random.RandomState(1)
n = 354
A = np.eye(n) * 4 + rng.randn(n, n) * 0.1
P = np.diag(1.0 / np.diag(A))
b = rng.randn(n); b /= np.linalg.norm(b)
for dev in ("cpu", "cuda"):
  At, Pt, bt = torch.as_tensor(A, device=dev), torch.as_tensor(P, device=dev), torch.as_tensor(b, device=dev)
  x, info = gmres_right_preconditioned(lambda v: At @ v, lambda v: Pt @ v, bt, tol=1e-11)
  print(dev, info["iterations"], info["residual"], np.linalg.norm(A @ x.cpu().numpy() - b))
# pieces on cuda
V = torch.empty((61, n), dtype=torch.float64, device="cuda")
V[0] = torch.as_tensor(b, device="cuda")
w = torch.as_tensor(rng.randn(n), device="cuda")
Vj = V[:1]
h = Vj @ w
print("h", h.cpu().numpy(), b @ w.cpu().numpy())
w2 = w - Vj.t() @ h
print("orth", float(torch.dot(w2, V[0])))
c = torch.cat([h + h, torch.linalg.norm(w2).reshape(1)]).cpu().numpy()
print(c)
