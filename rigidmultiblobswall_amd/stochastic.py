"""Lanczos stochastic forcing  factor * M^{1/2} z, device resident.

Same contract as stochastic_forcing/stochastic_forcing.py:112-264 (`stochastic_forcing_lanczos`): one
M.v product per iteration, stop when the relative change of the noise estimate drops below
`tolerance` (:239-255), return (noise, iterations); optional `L_mult` applied to the result.

Differences in HOW:
  * the Krylov basis lives in one pre-allocated device matrix (the reference grows it with
    np.concatenate every iteration, :237) and is re-orthogonalised with two classical Gram-Schmidt
    passes = two GEMVs (the reference loops over all previous rows in Python, :233-234);
  * the noise estimate V^T (Q sqrt(L) Q^T e_1) |z| factor is NOT formed every iteration: with an
    orthonormal basis, |noise_i - noise_{i-1}| equals the norm of the difference of the small
    coefficient vectors, so the convergence test is O(i) host work on the (tiny) tridiagonal
    eigen-problem and the 3N-vector is assembled once at the end (the reference does a dense `eigh`
    plus a V^T product per iteration, :215-229);
  * per iteration two scalars (h_ii, h_i+1,i) cross PCIe.
"""
import numpy as np
import torch


def _noise_coefficients(h_diag, h_sup, k, scale):
  """Small symmetric tridiagonal eigenproblem on the host (size k): noise = V[:k]^T coef."""
  H = np.diag(h_diag) + np.diag(h_sup[:k - 1], -1) + np.diag(h_sup[:k - 1], 1)
  lam, Q = np.linalg.eigh(H)
  return Q @ (np.sqrt(np.maximum(lam, 0.0)) * Q[0, :]) * scale


def _plain_step(V, w, i, h_diag, h_sup, sync, v_norm, factor):
  """One Lanczos step as separate tensor operations (any device, any process group): three-term recurrence, the two
  scalars to the host, the small eigenproblem, full re-orthogonalisation.  Returns (new basis vector, coef, k)."""
  if i > 0:
    w = w - h_sup[i - 1] * V[i - 1]
  hd = torch.dot(w, V[i])
  w = w - hd * V[i]
  hs = torch.linalg.norm(w)
  pair = torch.stack([hd, hs])
  if sync is not None:           # multi-rank replicated loop: every rank acts on rank 0's coefficients
    sync(pair)
  hd_f, hs_f = (float(x) for x in pair.cpu())
  h_diag.append(hd_f)
  h_sup.append(hs_f)
  if hs_f > 0:
    w = w / hs_f
  else:
    w = torch.zeros_like(w)
    w[0] = 1.0
  k = i + 1
  coef = _noise_coefficients(h_diag, h_sup, k, v_norm * factor)
  # full re-orthogonalisation of the new basis vector (two classical Gram-Schmidt passes)
  Vk = V[:k]
  w = w - Vk.t() @ (Vk @ w)
  w = w - Vk.t() @ (Vk @ w)
  return w, coef, k


def _lanczos_steps(factor, tolerance, max_iter, dim, z, print_residual, device, sync, ortho=None):
  """The Lanczos iteration as a coroutine: YIELDS every vector it needs the mobility applied to and receives the
  product back, so that one driver can run a single forcing or advance two of them in lockstep on a two-vector
  product.  Returns (noise, iterations) before `L_mult`.
  ortho: optional fused orthogonalisation ortho(V, rows, w, col, v_next) (MobilityContext.krylov_orthogonalize_device:
  two classical Gram-Schmidt passes of w against V[:rows], col[:rows] = the coefficients, col[rows] = |w|,
  v_next = w / |w|, four launches).  With a basis that is orthonormal to rounding the coefficients of all rows but the
  last two vanish to rounding, so col[i] and col[i + 1] ARE the three-term recurrence's h_ii and h_i+1,i; the iteration is
  then product + 4 launches + one 16-byte transfer instead of ~15 launches."""
  cap = min(max_iter + 2, 64)
  if sync is not None or device.type != "cuda":
    ortho = None
  V = torch.empty((cap, dim), dtype=torch.float64, device=device)
  col = torch.zeros(cap + 1, dtype=torch.float64, device=device) if ortho is not None else None
  v_norm = float(torch.linalg.norm(z))
  V[0] = z / v_norm
  h_diag, h_sup = [], []
  coef_old = None
  coef = None
  its = max_iter
  for i in range(max_iter + 1):
    w = (yield V[i]).reshape(-1)
    if i + 2 > cap:                # room for V[i + 1] before anything writes it
      cap = min(2 * cap, max_iter + 2)
      Vn = torch.empty((cap, dim), dtype=torch.float64, device=device)
      Vn[:i + 1] = V[:i + 1]
      V = Vn
      if col is not None:
        col = torch.zeros(cap + 1, dtype=torch.float64, device=device)
    if ortho is not None and i + 1 <= 256:     # the fused step takes up to 256 basis vectors
      ortho(V, i + 1, w if w.is_contiguous() else w.contiguous(), col, V[i + 1])
      hd_f, hs_f = col[i:i + 2].tolist()
      broke = not (hs_f > 0 and np.isfinite(hs_f))
      h_diag.append(hd_f)
      h_sup.append(0.0 if broke else hs_f)
      k = i + 1
      coef = _noise_coefficients(h_diag, h_sup, k, v_norm * factor)
      if broke:                    # exact breakdown (V[i + 1] holds 0 / 0): continue from e_0, orthogonalised, as the plain step does
        w = torch.zeros(dim, dtype=torch.float64, device=device)
        w[0] = 1.0
        Vk = V[:k]
        w = w - Vk.t() @ (Vk @ w)
        w = w - Vk.t() @ (Vk @ w)
        V[k] = w
    else:
      w, coef, k = _plain_step(V, w, i, h_diag, h_sup, sync, v_norm, factor)
      V[k] = w
    if i > 0:
      old = np.concatenate([coef_old, [0.0]])
      old_norm = np.linalg.norm(old)
      diff = np.linalg.norm(coef - old)
      if print_residual:
        if i == 1:
          print('lanczos =  0 1')
        print('lanczos = ', i, diff / old_norm)
      if diff / max(old_norm, np.finfo(float).eps) < tolerance:
        its = i
        break
    coef_old = coef
  k = len(coef)
  noise = V[:k].t() @ torch.as_tensor(coef, dtype=torch.float64, device=device)
  return noise, its


def _prepare(z, dim, device, mobility):
  if z is not None and dim is None:
    dim = int(z.numel() if isinstance(z, torch.Tensor) else np.size(z))
  if isinstance(z, torch.Tensor) and device is None:
    device = z.device
  if device is None:
    device = mobility.device if isinstance(mobility, torch.Tensor) else torch.device("cpu")
  device = torch.device(device)
  if z is None:
    z = torch.randn(dim, dtype=torch.float64, device=device)
  z = torch.as_tensor(z, dtype=torch.float64, device=device).reshape(-1).clone()
  return z, dim, device


def stochastic_forcing_lanczos(factor=1.0, tolerance=1e-6, max_iter=1000, dim=None, mobility=None,
                               mobility_mult=None, L_mult=None, z=None, print_residual=False, device=None, sync=None,
                               ortho=None):
  """mobility_mult: callable(torch tensor (dim,)) -> torch tensor (dim,) on the same device
  (e.g. lambda v: ctx.matvec_device('tt', v, eta)); or `mobility` = dense torch/numpy matrix.
  z: numpy array or torch tensor; drawn from N(0,1) when None.  Returns (noise tensor, iterations)."""
  if z is not None and dim is None:
    dim = int(z.numel() if isinstance(z, torch.Tensor) else np.size(z))
  if factor == 0.0:
    dev = z.device if isinstance(z, torch.Tensor) and device is None else (device if device is not None else
                                                                            (mobility.device if isinstance(mobility, torch.Tensor) else "cpu"))
    return torch.zeros(dim, dtype=torch.float64, device=torch.device(dev)), 0
  z, dim, device = _prepare(z, dim, device, mobility)
  if mobility is not None:
    Mt = torch.as_tensor(mobility, dtype=torch.float64, device=device)
    mobility_mult = lambda v: Mt @ v  # noqa: E731
  steps = _lanczos_steps(factor, tolerance, max_iter, dim, z, print_residual, device, sync, ortho=ortho)
  try:
    request = next(steps)
    while True:
      request = steps.send(mobility_mult(request))
  except StopIteration as done:
    noise, its = done.value
  if L_mult is not None:
    noise = L_mult(noise).reshape(-1)
  return noise, its


def stochastic_forcing_lanczos_pair(factors, zs, mobility_mult, mobility_mult2, tolerance=1e-6, max_iter=1000, L_mult=None,
                                    print_residual=False, device=None, sync=None, ortho=None):
  """Two forcings factor_k M^{1/2} z_k with the SAME mobility advanced in lockstep: while both run, each iteration hands
  its two product requests to mobility_mult2(u, v) -> (M u, M v) (one pass over the pairs, rmb_matvec2_device).  Each
  forcing sees exactly the iterates it would see alone.  Returns ((noise_a, its_a), (noise_b, its_b))."""
  z0, dim, device = _prepare(zs[0], None, device, None)
  z1, _, _ = _prepare(zs[1], None, device, None)
  gens = [_lanczos_steps(f, tolerance, max_iter, dim, z, print_residual, device, sync, ortho=ortho) for f, z in zip(factors, (z0, z1))]
  requests, results = [None, None], [None, None]
  for k in (0, 1):
    requests[k] = next(gens[k])
  while results[0] is None or results[1] is None:
    if results[0] is None and results[1] is None:
      answers = mobility_mult2(requests[0], requests[1])
    else:
      k = 0 if results[0] is None else 1
      answers = [None, None]
      answers[k] = mobility_mult(requests[k])
    for k in (0, 1):
      if results[k] is None:
        try:
          requests[k] = gens[k].send(answers[k])
        except StopIteration as done:
          results[k] = done.value
  out = []
  for noise, its in results:
    out.append((L_mult(noise).reshape(-1) if L_mult is not None else noise, its))
  return out[0], out[1]


# ---- dense forcings (stochastic_forcing/stochastic_forcing.py:7-109): O(n^3), for small systems and as checks --------
def _dense(mobility, z, device):
  M = torch.as_tensor(np.asarray(mobility, dtype=np.float64) if not isinstance(mobility, torch.Tensor) else mobility,
                      dtype=torch.float64, device=device)
  if z is None:
    z = torch.randn(M.shape[0], dtype=torch.float64, device=M.device)
  return M, torch.as_tensor(z, dtype=torch.float64, device=M.device).reshape(-1)


def stochastic_forcing_eig(mobility, factor=1.0, z=None, device=None):
  """factor V S^{1/2} z with M = V S V^T (negative eigenvalues clipped to 0), :7-41."""
  M, z = _dense(mobility, z, device)
  lam, V = torch.linalg.eigh(M)
  return factor * (V @ (torch.sqrt(torch.clamp(lam, min=0.0)) * z))


def stochastic_forcing_eig_symm(mobility, factor=1.0, z=None, device=None):
  """factor V S^{1/2} V^T z -- the symmetric square root, what Lanczos converges to, :44-82."""
  M, z = _dense(mobility, z, device)
  lam, V = torch.linalg.eigh(M)
  return factor * (V @ (torch.sqrt(torch.clamp(lam, min=0.0)) * (V.t() @ z)))


def stochastic_forcing_cholesky(mobility, factor=1.0, z=None, device=None):
  """factor L z with M = L L^T, :85-109."""
  M, z = _dense(mobility, z, device)
  return factor * (torch.linalg.cholesky(M) @ z)
