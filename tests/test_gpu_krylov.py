"""The O(N) helpers of the rigid-body solve (csrc/rmb_krylov.hip) against plain numpy / torch fp64: the batched
two-by-two block product (preconditioner, K and K^T products) and the fused two-pass Gram-Schmidt of an Arnoldi step."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nb,r1,c1,r2,c2", [(1, 1, 1, 1, 1), (5, 36, 36, 6, 6), (300, 36, 36, 6, 6), (7, 150, 150, 6, 6), (3, 700, 640, 9, 5),
                                            (4, 5, 0, 3, 2), (2048, 36, 36, 6, 6), (32, 126, 126, 6, 6), (5, 96, 96, 6, 6), (5, 97, 97, 6, 6), (3, 129, 257, 5, 130)])
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


@pytest.mark.parametrize("n,rows", [(1, 1), (5, 3), (1024, 1), (1025, 7), (4608, 9), (4608, 60), (86016, 19), (300001, 61), (70000, 256),
                                    (2688, 17), (6144, 61), (6144, 256), (6145, 61)])
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
    # the column once more in page-locked, device-mapped host memory: readable after one stream wait, no copy command
    from rigidmultiblobswall_amd.context import MappedHostArray
    mapped = MappedHostArray((3, m + 1))
    try:
      w_c, col_c = w0.clone(), torch.zeros_like(col)
      ctx.krylov_orthogonalize_device(V, rows, w_c, col_c, vb, mapped.dev_ptr + 8 * (m + 1))      # row 1 of the buffer
      torch.cuda.synchronize()
      assert np.array_equal(mapped.array[1, :rows + 1], col[:rows + 1].cpu().numpy())
      assert not mapped.array[0].any() and not mapped.array[2].any() and torch.equal(w_c, w)
    finally:
      mapped.close()
  finally:
    ctx.close()


@pytest.mark.parametrize("nb,n_b,wall,L", [(40, 12, True, None), (7, 15, False, None), (3, 42, True, None), (64, 12, True, (60.0, 60.0, 0.0)),
                                           (5, 12, True, None), (300, 12, True, None)])
def test_rigid_operator_in_one_call_equals_product_and_block_products(nb, n_b, wall, L):
  """rmb_rigid_operator_device = [M_tt lambda - K U; -K^T lambda] (multi_bodies.py:424-471): the pair sweep + ONE finishing
  launch with the symmetric kernels, the three-launch fallback below 128 blobs and with periodic images -- against the
  product and numpy K products, twice in a row (the accumulators are back to zero), and against the oracle's M."""
  import torch
  from rigidmultiblobswall_amd import MobilityContext
  from oracle import oracle
  rng = np.random.RandomState(nb + n_b)
  N = nb * n_b
  a, eta = 0.3, 1.3
  r = rng.rand(N, 3) * (N / 0.05) ** (1.0 / 3.0) * a
  r[:, 2] += 0.5 * a
  K = rng.randn(nb, 3 * n_b, 6)
  x = rng.randn(3 * N + 6 * nb)
  dev = lambda v: torch.as_tensor(np.ascontiguousarray(v), device="cuda")
  ctx = MobilityContext(0)
  try:
    ctx.set_positions(dev(r.reshape(-1)), a, None if L is None else np.array(L), wall)
    lam, U = x[:3 * N], x[3 * N:].reshape(nb, 6)
    Mlam = ctx.matvec_device("tt", dev(lam), eta).cpu().numpy()
    want = np.concatenate([Mlam - np.einsum("bij,bj->bi", K, U).reshape(-1),
                           -np.einsum("bij,bi->bj", K, lam.reshape(nb, 3 * n_b)).reshape(-1)])
    for rep in range(2):
      out = torch.full((3 * N + 6 * nb,), float("nan"), dtype=torch.float64, device="cuda")
      ctx.rigid_operator_device(dev(K), dev(x), eta, out)
      assert rel_err(out.cpu().numpy(), want) < 1e-13, (rep, rel_err(out.cpu().numpy(), want))
    assert rel_err(ctx.matvec_device("tt", dev(lam), eta).cpu().numpy(), Mlam) < 1e-13      # the plain product still finds clean accumulators
    kw = {} if L is None else dict(periodic_length=np.array(L))
    fn = oracle.single_wall_mobility_trans_times_force_oracle if wall else oracle.no_wall_mobility_trans_times_force_oracle
    assert rel_err(out.cpu().numpy()[:3 * N] + np.einsum("bij,bj->bi", K, U).reshape(-1), fn(r, lam.reshape(-1, 3), eta, a, **kw)) < 1e-11
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


@pytest.mark.parametrize("nb,n_b", [(1, 1), (3, 2), (5, 12), (300, 12), (7, 16), (2048, 12), (4, 7)])
def test_rigid_configuration_kernel_matches_the_body_formulas(nb, n_b):
  """Blob coordinates, body-frame offsets and K of body/body.py:64-115 in one launch, against the numpy formulas
  (rigid.blob_positions / quaternion_rotation_matrix, the rot matrix written out)."""
  import torch
  from rigidmultiblobswall_amd import MobilityContext
  from rigidmultiblobswall_amd.rigid import blob_positions
  rng = np.random.RandomState(nb + n_b)
  ref = rng.randn(nb, n_b, 3)
  loc = rng.randn(nb, 3) * 3
  quat = rng.randn(nb, 4)
  quat /= np.linalg.norm(quat, axis=1, keepdims=True)
  dev = lambda x: torch.as_tensor(x, device="cuda")
  r = torch.empty((nb * n_b, 3), dtype=torch.float64, device="cuda")
  rel = torch.empty((nb, n_b, 3), dtype=torch.float64, device="cuda")
  K = torch.full((nb, 3 * n_b, 6), float("nan"), dtype=torch.float64, device="cuda")       # every entry must be written
  ctx = MobilityContext(0)
  try:
    ctx.rigid_configuration_device(dev(ref), dev(loc), dev(quat), r, rel, K)
    r_ref = np.concatenate([blob_positions(ref[b], loc[b], quat[b]) for b in range(nb)])
    assert np.abs(r.cpu().numpy() - r_ref).max() < 1e-13
    rel_ref = r_ref.reshape(nb, n_b, 3) - loc[:, None, :]
    assert np.abs(rel.cpu().numpy() - rel_ref).max() < 1e-13
    K_ref = np.zeros((nb, n_b, 3, 6))
    K_ref[:, :, 0, 0] = K_ref[:, :, 1, 1] = K_ref[:, :, 2, 2] = 1.0
    K_ref[:, :, 0, 4] = rel_ref[:, :, 2];  K_ref[:, :, 0, 5] = -rel_ref[:, :, 1]
    K_ref[:, :, 1, 3] = -rel_ref[:, :, 2]; K_ref[:, :, 1, 5] = rel_ref[:, :, 0]
    K_ref[:, :, 2, 3] = rel_ref[:, :, 1];  K_ref[:, :, 2, 4] = -rel_ref[:, :, 0]
    assert np.abs(K.cpu().numpy() - K_ref.reshape(nb, 3 * n_b, 6)).max() < 1e-13
    # outputs that are not wanted may be left out
    r2 = torch.empty_like(r)
    ctx.rigid_configuration_device(dev(ref), dev(loc), dev(quat), r2)
    assert torch.equal(r2, r)
  finally:
    ctx.close()


@pytest.mark.parametrize("nb,n_b", [(1, 3), (5, 12), (300, 12), (9, 16), (2048, 12), (6, 4),
                                    # round 5: 17 .. 42 blobs per body, ONE n x n matrix in LDS, everything in place
                                    (3, 17), (7, 30), (2, 42), (300, 42)])
def test_rigid_preconditioner_kernel_matches_dense_linear_algebra(nb, n_b):
  """Per-body Cholesky factor, its inverse, M^-1, N = (K^T M^-1 K)^-1 and the four blocks of [[M, -K], [-K^T, 0]]^-1
  (multi_bodies.py:516-560) against numpy on every body -- the blocks checked by what they must satisfy."""
  import torch
  from rigidmultiblobswall_amd import MobilityContext
  rng = np.random.RandomState(nb * 7 + n_b)
  n = 3 * n_b
  G = rng.randn(nb, n, n)
  Mb = G @ G.transpose(0, 2, 1) / n + 0.5 * np.eye(n)          # SPD, condition number O(10)
  rel = rng.randn(nb, n_b, 3)
  K = np.zeros((nb, n_b, 3, 6))
  K[:, :, 0, 0] = K[:, :, 1, 1] = K[:, :, 2, 2] = 1.0
  K[:, :, 0, 4] = rel[:, :, 2];  K[:, :, 0, 5] = -rel[:, :, 1]
  K[:, :, 1, 3] = -rel[:, :, 2]; K[:, :, 1, 5] = rel[:, :, 0]
  K[:, :, 2, 3] = rel[:, :, 1];  K[:, :, 2, 4] = -rel[:, :, 0]
  K = K.reshape(nb, n, 6)
  dev = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda")
  mk = lambda *s: torch.full(s, float("nan"), dtype=torch.float64, device="cuda")
  out = dict(Lchol=mk(nb, n, n), Linv=mk(nb, n, n), Minv=mk(nb, n, n), Nbody=mk(nb, 6, 6), A11=mk(nb, n, n), A12=mk(nb, n, 6),
             A21=mk(nb, 6, n), A22=mk(nb, 6, 6))
  info = torch.ones(1, dtype=torch.int32, device="cuda")
  ctx = MobilityContext(0)
  try:
    # a slightly unsymmetric input: the kernel symmetrises on load, like the torch path
    skew = 1e-13 * rng.randn(nb, n, n)
    ctx.rigid_preconditioner_device(dev(Mb + skew - skew.transpose(0, 2, 1)), dev(K), *out.values(), info)
    assert int(info) == 0
    o = {k: v.cpu().numpy() for k, v in out.items()}
    L = np.linalg.cholesky(Mb)
    assert np.abs(o["Lchol"] - L).max() < 1e-12 and np.abs(np.triu(o["Lchol"], 1)).max() == 0.0
    eye = np.eye(n)
    assert np.abs(o["Linv"] @ L - eye).max() < 1e-11
    assert np.abs(o["Minv"] @ Mb - eye).max() < 1e-10 and np.abs(o["Minv"] - o["Minv"].transpose(0, 2, 1)).max() == 0.0
    Rres = K.transpose(0, 2, 1) @ np.linalg.solve(Mb, K)
    assert np.abs(o["Nbody"] @ Rres - np.eye(6)).max() < 1e-9 and np.abs(o["Nbody"] - o["Nbody"].transpose(0, 2, 1)).max() == 0.0
    # the blocks ARE the inverse of the saddle-point matrix of one body
    S = np.zeros((nb, n + 6, n + 6))
    S[:, :n, :n] = Mb; S[:, :n, n:] = -K; S[:, n:, :n] = -K.transpose(0, 2, 1)
    P = np.zeros_like(S)
    P[:, :n, :n] = o["A11"]; P[:, :n, n:] = o["A12"]; P[:, n:, :n] = o["A21"]; P[:, n:, n:] = o["A22"]
    assert np.abs(P @ S - np.eye(n + 6)).max() < 1e-8
    assert np.abs(o["A21"] - o["A12"].transpose(0, 2, 1)).max() == 0.0
    # a body whose resistance has no inverse (all blobs at the tracking point: K^T M^-1 K is rank 3): flagged
    K_bad = K.copy(); K_bad[0, :, 3:] = 0.0
    ctx.rigid_preconditioner_device(dev(Mb), dev(K_bad), *out.values(), info)
    assert int(info) == 1
    # and a matrix that is not positive definite
    Mb_bad = Mb.copy(); Mb_bad[-1] = -Mb_bad[-1]
    ctx.rigid_preconditioner_device(dev(Mb_bad), dev(K), *out.values(), info)
    assert int(info) == 1
    ctx.rigid_preconditioner_device(dev(Mb), dev(K), *out.values(), info)
    assert int(info) == 0                                        # the flag is reset by every call
  finally:
    ctx.close()


def test_native_preconditioner_equals_the_torch_build_and_single_blobs_take_the_pseudo_inverse():
  """RigidSuspension.build_preconditioner through the kernel against native_helpers = False (batched torch.linalg): same
  factors and blocks to rounding; a suspension of single-blob bodies (rank-3 resistance) is flagged by the kernel once
  and keeps the reference's pseudo-inverse route."""
  import torch
  from rigidmultiblobswall_amd import structures as st
  from rigidmultiblobswall_amd.rigid import RigidSuspension
  R, eta = 1.0155, 0.957e-3
  shell = st.icosahedron_shell(0.792079207921 * R)
  a = st.min_blob_separation(shell) / 2
  nb = 50
  loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=3)
  nat = RigidSuspension([shell] * nb, loc, quat, a, eta, device="cuda:0")
  ref = RigidSuspension([shell] * nb, loc, quat, a, eta, device="cuda:0")
  ref.native_helpers = False
  try:
    ref.set_configuration(loc, quat)
    assert rel_err(nat.r_vectors, ref.r_vectors) < 1e-14
    assert rel_err(nat.groups[0].K.cpu().numpy(), ref.groups[0].K.cpu().numpy()) < 1e-14
    nat.build_preconditioner(); ref.build_preconditioner()
    assert nat.groups[0].Linv is not None and ref.groups[0].Linv is None
    for name, tol in (("Lchol", 1e-12), ("Minv", 1e-10), ("Nbody", 1e-10), ("A11", 1e-10), ("A12", 1e-10), ("A21", 1e-10), ("A22", 1e-10)):
      assert rel_err(getattr(nat.groups[0], name).cpu().numpy(), getattr(ref.groups[0], name).cpu().numpy()) < tol, name
    ref._stochastic_factors()
    assert rel_err(nat.groups[0].Linv.cpu().numpy(), ref.groups[0].Linv.cpu().numpy()) < 1e-11
  finally:
    nat.close(); ref.close()
  one = np.zeros((1, 3))
  loc1 = np.random.RandomState(0).rand(30, 3) * 8 + np.array([0, 0, 1.5])
  quat1 = np.tile([1.0, 0, 0, 0], (30, 1))
  s1 = RigidSuspension([one] * 30, loc1, quat1, 0.5, 1.0, device="cuda:0")
  s2 = RigidSuspension([one] * 30, loc1, quat1, 0.5, 1.0, device="cuda:0")
  s2.native_helpers = False
  try:
    s1.build_preconditioner(); s2.build_preconditioner()
    assert len(s1._native_pc_rejected) == 1
    assert rel_err(s1.groups[0].Nbody.cpu().numpy(), s2.groups[0].Nbody.cpu().numpy()) < 1e-12
    FT = np.zeros((30, 6)); FT[:, 2] = -1.0
    U1, _, i1 = s1.solve_mobility_problem(force_torque=FT, tol=1e-10)
    U2, _, i2 = s2.solve_mobility_problem(force_torque=FT, tol=1e-10)
    assert i1["converged"] and rel_err(U1[:, :3], U2[:, :3]) < 1e-8
  finally:
    s1.close(); s2.close()


def test_native_preconditioner_for_the_references_42_blob_shells():
  """Bodies of 42 blobs (multi_bodies/Structures/shell_N_42_Rg_0_8913_Rh_1.vertex through the g9 fixture): the per-body
  factors come from the in-place LDS kernel (Linv is only set by the native path), equal the batched torch.linalg build, and
  the preconditioned solve takes the same iterations either way."""
  import os
  import torch
  from conftest import GOLDEN, load_golden
  from rigidmultiblobswall_amd.rigid import RigidSuspension
  g = load_golden(os.path.join(GOLDEN, "g9_rigid_det_euler_42blob_shells.npz"))
  shell, loc, quat = g["vertex_shell42"], g["locations_shell42"], g["quaternions_shell42"]
  assert len(shell) == 42
  d = np.linalg.norm(shell[:, None] - shell[None], axis=2)
  a, eta = float(np.min(d[d > 0]) / 2), 1.1
  # 6 bodies as the fixture, and 60 of them on a lattice (more workgroups than one per CU is not needed for the check)
  loc60 = np.array([[2.6 * (k % 8), 2.6 * (k // 8), 1.4] for k in range(60)])
  quat60 = np.tile(quat, (10, 1))
  for L_, Q_ in ((loc, quat), (loc60, quat60)):
    nb = len(L_)
    nat = RigidSuspension([shell] * nb, L_, Q_, a, eta, device="cuda:0")
    ref = RigidSuspension([shell] * nb, L_, Q_, a, eta, device="cuda:0")
    ref.native_helpers = False
    try:
      nat.build_preconditioner(); ref.build_preconditioner()
      assert nat.groups[0].Linv is not None and ref.groups[0].Linv is None          # the kernel ran, not torch.linalg
      assert not getattr(nat, "_native_pc_rejected", ())
      for name, tol in (("Lchol", 1e-12), ("Minv", 1e-9), ("Nbody", 1e-9), ("A11", 1e-9), ("A12", 1e-9), ("A21", 1e-9), ("A22", 1e-9)):
        assert rel_err(getattr(nat.groups[0], name).cpu().numpy(), getattr(ref.groups[0], name).cpu().numpy()) < tol, name
      FT = np.zeros((nb, 6)); FT[:, 2] = -1.0; FT[:, 4] = 0.3
      U1, _, i1 = nat.solve_mobility_problem(force_torque=FT, tol=1e-9)
      U2, _, i2 = ref.solve_mobility_problem(force_torque=FT, tol=1e-9)
      assert i1["converged"] and abs(i1["iterations"] - i2["iterations"]) <= 1 and rel_err(U1, U2) < 1e-7
    finally:
      nat.close(); ref.close()


def test_rigid_advance_kernel_matches_the_quaternion_update():
  """x + v dt and quaternion(omega dt) * q (quaternion_integrator_multi_bodies.py:86-91) in one launch against the torch
  formulas of rigid.py, with a scalar step, a per-body step, and a body that does not rotate (|omega| = 0)."""
  import torch
  from rigidmultiblobswall_amd import MobilityContext
  from rigidmultiblobswall_amd.rigid import quaternion_from_rotation_torch, quaternion_multiply_torch
  g = torch.Generator().manual_seed(4)
  nb = 777
  loc = torch.randn(nb, 3, generator=g, dtype=torch.float64).cuda()
  quat = torch.randn(nb, 4, generator=g, dtype=torch.float64)
  quat = (quat / quat.norm(dim=1, keepdim=True)).cuda()
  U = torch.randn(nb, 6, generator=g, dtype=torch.float64).cuda()
  U[5, 3:] = 0.0
  ctx = MobilityContext(0)
  try:
    for dt in (0.013, torch.rand(nb, 1, generator=g, dtype=torch.float64).cuda()):
      l2, q2 = ctx.rigid_advance_device(loc, quat, U, dt)
      l_ref = loc + U[:, :3] * dt
      q_ref = quaternion_multiply_torch(quaternion_from_rotation_torch(U[:, 3:] * dt), quat)
      assert float((l2 - l_ref).abs().max()) < 1e-14 and float((q2 - q_ref).abs().max()) < 1e-14
      assert torch.equal(q2[5], quat[5])
      assert float((q2.norm(dim=1) - 1).abs().max()) < 1e-14
  finally:
    ctx.close()


def test_lanczos_with_the_fused_step_equals_the_plain_recurrence():
  """stochastic_forcing_lanczos with ortho = the fused orthogonalisation against the plain three-term recurrence + full
  re-orthogonalisation: same iteration counts, the same M^{1/2} z to rounding, equal to the dense symmetric square root;
  and an exact breakdown (z an eigenvector: |w| = 0 after one step) is survived."""
  import torch
  from rigidmultiblobswall_amd import MobilityContext
  from rigidmultiblobswall_amd.stochastic import stochastic_forcing_lanczos, stochastic_forcing_eig_symm
  rng = np.random.RandomState(3)
  n = 900
  Q, _ = np.linalg.qr(rng.randn(n, n))
  lam = 10.0 ** rng.uniform(-1.5, 1.0, n)
  M = torch.as_tensor((Q * lam) @ Q.T, device="cuda")
  ctx = MobilityContext(0)
  try:
    for tol in (1e-3, 1e-8):
      z = torch.as_tensor(rng.randn(n), device="cuda")
      a, ia = stochastic_forcing_lanczos(factor=0.7, tolerance=tol, mobility_mult=lambda v: M @ v, z=z)
      b, ib = stochastic_forcing_lanczos(factor=0.7, tolerance=tol, mobility_mult=lambda v: M @ v, z=z,
                                         ortho=ctx.krylov_orthogonalize_device)
      assert ia == ib, (tol, ia, ib)
      assert rel_err(b.cpu().numpy(), a.cpu().numpy()) < 1e-10
      if tol < 1e-6:
        ref = stochastic_forcing_eig_symm(M, factor=0.7, z=z)
        assert rel_err(b.cpu().numpy(), ref.cpu().numpy()) < 1e-6
    z = torch.as_tensor(Q[:, 5].copy(), device="cuda")        # eigenvector: the Krylov space is one-dimensional
    b, ib = stochastic_forcing_lanczos(factor=1.0, tolerance=1e-8, mobility_mult=lambda v: M @ v, z=z, ortho=ctx.krylov_orthogonalize_device)
    assert np.all(np.isfinite(b.cpu().numpy())) and rel_err(b.cpu().numpy(), np.sqrt(lam[5]) * Q[:, 5]) < 1e-7
  finally:
    ctx.close()
