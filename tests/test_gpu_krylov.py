"""The O(N) helpers of the rigid-body solve (csrc/rmb_krylov.hip) against plain numpy / torch fp64: the batched
two-by-two block product (preconditioner, K and K^T products) and the fused two-pass Gram-Schmidt of an Arnoldi step."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nb,r1,c1,r2,c2", [(1, 1, 1, 1, 1), (5, 36, 36, 6, 6), (300, 36, 36, 6, 6), (7, 150, 150, 6, 6), (3, 700, 640, 9, 5),
                                            (4, 5, 0, 3, 2), (2048, 36, 36, 6, 6)])
def test_block_apply_matches_batched_products(nb, r1, c1, r2, c2):
  import torch
  from rigidmultiblobswall_amd import MobilityContext
  g = torch.Generator(device="cpu").manual_seed(nb + r1)
  rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64).cuda()
  ctx = MobilityContext(0)
  try:
    A11, A12, A21, A22 = rnd(nb, r1, c1), rnd(nb, r1, c2), rnd(nb, r2, c1), rnd(nb, r2, c2)
    x1, x2 = rnd(nb, c1), rnd(nb, c2)
    y1, y2 = rnd(nb, r1), rnd(nb, r2)
    y1_0, y2_0 = y1.clone(), y2.clone()
    mv = lambda A, x: torch.bmm(A, x.unsqueeze(-1)).squeeze(-1)
    # overwrite (beta = 0: the old contents, here with a NaN planted, must not be read)
    y1[0, 0] = float("nan")
    ctx.block_apply_device(A11, A12, A21, A22, x1, x2, y1, y2)
    assert rel_err(y1.cpu().numpy(), (mv(A11, x1) + mv(A12, x2)).cpu().numpy()) < 1e-13
    assert rel_err(y2.cpu().numpy(), (mv(A21, x1) + mv(A22, x2)).cpu().numpy()) < 1e-13
    # accumulate with alpha, beta; absent blocks; a transposed block (the operator's use: top -= K U, bottom = -K^T lambda)
    if c1 == r1:
      K = rnd(nb, r1, c2)
      KT_shape_ok = (r2 == c2)
      y1.copy_(y1_0); y2.copy_(y2_0)
      if KT_shape_ok:
        ctx.block_apply_device(None, K, K, None, x1, x2, y1, y2, alpha=-1.0, beta1=1.0, transpose=(False, False, True, False))
        assert rel_err(y1.cpu().numpy(), (y1_0 - mv(K, x2)).cpu().numpy()) < 1e-13
        assert rel_err(y2.cpu().numpy(), (-mv(K.transpose(1, 2), x1)).cpu().numpy()) < 1e-13
    y1.copy_(y1_0); y2.copy_(y2_0)
    ctx.block_apply_device(A11, None, None, A22, x1, x2, y1, y2, alpha=0.5, beta1=2.0, beta2=-1.0)
    assert rel_err(y1.cpu().numpy(), (2.0 * y1_0 + 0.5 * mv(A11, x1)).cpu().numpy()) < 1e-13
    assert rel_err(y2.cpu().numpy(), (-y2_0 + 0.5 * mv(A22, x2)).cpu().numpy()) < 1e-13
    # strided blocks: a slice of a larger tensor
    big = rnd(nb, r1 + 3, c1 + 2)
    sub = big[:, 1:1 + r1, 2:2 + c1]
    ctx.block_apply_device(sub, None, None, None, x1, x2, y1, y2)
    assert rel_err(y1.cpu().numpy(), mv(sub, x1).cpu().numpy()) < 1e-13 if c1 > 0 else float(y1.abs().max()) == 0.0
    assert float(y2.abs().max()) == 0.0
  finally:
    ctx.close()


@pytest.mark.parametrize("n,rows", [(1, 1), (5, 3), (1024, 1), (1025, 7), (4608, 9), (4608, 60), (86016, 19), (300001, 61), (70000, 256)])
def test_fused_gram_schmidt_matches_two_classical_passes(n, rows):
  import torch
  from rigidmultiblobswall_amd import MobilityContext
  g = torch.Generator(device="cpu").manual_seed(n + rows)
  m = rows + 2
  rows_eff = min(rows, n)           # an orthonormal set of more than n vectors does not exist
  Q, _ = torch.linalg.qr(torch.randn(n, rows_eff, generator=g, dtype=torch.float64))
  V = torch.zeros((m, n), dtype=torch.float64)
  V[:rows_eff] = Q.t()
  V = V.cuda()
  w0 = torch.randn(n, generator=g, dtype=torch.float64).cuda()
  ctx = MobilityContext(0)
  try:
    w, col = w0.clone(), torch.zeros(m + 1, dtype=torch.float64, device="cuda")
    ctx.krylov_orthogonalize_device(V, rows, w, col, V[rows])
    Vj = V[:rows].clone()
    h = Vj @ w0
    w1 = w0 - Vj.t() @ h
    h2 = Vj @ w1
    w2 = w1 - Vj.t() @ h2
    nrm = float(torch.linalg.vector_norm(w2))
    assert rel_err(col[:rows].cpu().numpy(), (h + h2).cpu().numpy()) < 1e-12
    if rows_eff < n:
      assert abs(float(col[rows]) - nrm) <= 1e-12 * max(nrm, 1.0)
      assert rel_err(w.cpu().numpy(), w2.cpu().numpy()) < 1e-9
      assert rel_err(V[rows].cpu().numpy(), (w2 / nrm).cpu().numpy()) < 1e-9
      # what the solver needs of it: orthogonal to the basis to rounding, unit length
      assert float((Vj @ V[rows]).abs().max()) < 1e-12 and abs(float(torch.linalg.vector_norm(V[rows])) - 1.0) < 1e-13
    # fixed-order reductions: bit-reproducible
    w_b, col_b = w0.clone(), torch.zeros_like(col)
    vb = torch.empty(n, dtype=torch.float64, device="cuda")
    ctx.krylov_orthogonalize_device(V, rows, w_b, col_b, vb)
    assert torch.equal(w_b, w) and torch.equal(col_b[:rows + 1], col[:rows + 1])
  finally:
    ctx.close()


def test_helper_argument_checks():
  import torch
  from rigidmultiblobswall_amd import MobilityContext, _lib
  ctx = MobilityContext(0)
  try:
    V = torch.zeros((300, 16), dtype=torch.float64, device="cuda")
    w = torch.zeros(16, dtype=torch.float64, device="cuda")
    col = torch.zeros(301, dtype=torch.float64, device="cuda")
    with pytest.raises(_lib.RmbError):
      ctx.krylov_orthogonalize_device(V, 257, w, col, V[299])          # more rows than one call takes
    x1 = torch.zeros((2, 9000), dtype=torch.float64, device="cuda")
    x2 = torch.zeros((2, 1), dtype=torch.float64, device="cuda")
    y = torch.zeros((2, 1), dtype=torch.float64, device="cuda")
    with pytest.raises(_lib.RmbError):
      ctx.block_apply_device(None, None, None, None, x1, x2, y, y.clone())   # operand of one entry does not fit LDS
  finally:
    ctx.close()
