"""GPU tests of the rigid-multiblob layer: dense per-body blocks vs the oracle, the
golden saddle-point solve, and BASELINE.json configs[2] (2048 rollers x 12-blob shells, full GMRES
mobility solve on 1 MI355X) through size-independent checks."""
import os
import time

import numpy as np
import pytest

from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu


def test_body_dense_blocks_match_oracle(oracle):
  import torch
  from rigidmultiblobswall_amd import MobilityContext
  rng = np.random.RandomState(3)
  n_b, nb, a, eta = 7, 5, 0.3, 1.3
  r = rng.rand(nb * n_b, 3) * 2.5 + np.array([0, 0, 0.1])      # some blobs below z = a: clamp + B path
  for wall in (True, False):
    ctx = MobilityContext(0)
    ctx.set_positions(torch.as_tensor(r.reshape(-1), device="cuda"), a, wall=wall)
    first = torch.arange(0, nb * n_b, n_b, device="cuda", dtype=torch.int64)
    M = ctx.body_mobility_dense_device(first, n_b, eta).cpu().numpy()
    for k in range(nb):
      rk = r[k * n_b:(k + 1) * n_b]
      if wall:
        r_eff, b, _ = oracle.wall_regularisation(rk, a)
        B = np.repeat(b, 3)
        ref = B[:, None] * oracle.dense("tt", 1, r_eff, eta, a) * B[None, :]
      else:
        ref = oracle.dense("tt", 0, rk, eta, a)
      assert rel_err(M[k], ref) < 1e-13
    ctx.close()


def test_golden_saddle_point_solve_on_gpu():
  from rigidmultiblobswall_amd.rigid import RigidSuspension
  d = np.load(os.path.join(GOLDEN, "g7_rigid_suspension.npz"))
  refs = [d["shell"] if s else d["boomerang"] for s in d["body_is_shell"]]
  rs = RigidSuspension(refs, d["locations"], d["quaternions"], float(d["a"]), float(d["eta"]))
  U, lam, info = rs.solve_mobility_problem(slip=d["slip"], force_torque=d["force_torque"], tol=1e-10)
  assert info["converged"] and info["iterations"] < 60
  assert rel_err(U.reshape(-1), d["velocities"]) < 1e-8
  assert rel_err(lam.reshape(-1), d["lambda_blobs"]) < 1e-7
  rs.close()


def test_config1_boomerang_on_gpu():
  from rigidmultiblobswall_amd.rigid import RigidSuspension
  d = np.load(os.path.join(GOLDEN, "g7_boomerang_body_mobility.npz"))
  rs = RigidSuspension([d["reference_configuration"]], [d["location"]], [d["quaternion"]], float(d["a"]), float(d["eta"]))
  rs.build_preconditioner()
  assert rel_err(rs.groups[0].Nbody[0].cpu().numpy(), d["body_mobility"]) < 1e-10
  rs.close()


@pytest.mark.parametrize("n_bodies", [30, 2048])
def test_config3_roller_shells_gmres(oracle, n_bodies):
  """configs[2]: 12-blob shells (R = 1.0155, a = half the blob separation), D3 monolayer, constant torque
  about y + gravity-like force; GMRES tol 1e-8."""
  import torch
  from rigidmultiblobswall_amd import structures as st
  from rigidmultiblobswall_amd.rigid import RigidSuspension
  R, eta = 1.0155, 0.957e-3
  shell = st.icosahedron_shell(0.792079207921 * R)
  a = st.min_blob_separation(shell) / 2
  loc, q, _ = st.roller_monolayer(n_bodies, radius=R, seed=5)
  rs = RigidSuspension([shell] * n_bodies, loc, q, a, eta)
  assert rs.n_blobs == 12 * n_bodies
  FT = np.zeros((n_bodies, 6))
  FT[:, 2] = -0.05
  FT[:, 4] = 1.0
  t0 = time.time()
  U, lam, info = rs.solve_mobility_problem(force_torque=FT, tol=1e-8)
  torch.cuda.synchronize()
  dt = time.time() - t0
  print("config3 n_bodies=%d blobs=%d iterations=%d residual=%.2e matvecs=%d time=%.3fs" %
        (n_bodies, rs.n_blobs, info["iterations"], info["residual"], rs.matvec_count, dt))
  assert info["converged"] and info["iterations"] <= 120
  # independent residual of the saddle-point system
  x = torch.as_tensor(np.concatenate([lam.reshape(-1), U.reshape(-1)]), device="cuda")
  rhs = torch.as_tensor(np.concatenate([np.zeros(3 * rs.n_blobs), -FT.reshape(-1)]), device="cuda")
  res = float(torch.linalg.norm(rs.apply_operator(x) - rhs) / torch.linalg.norm(rhs))
  assert res < 5e-8
  # force / torque balance: K^T lambda = F
  assert rel_err(rs.KT_times_lambda(x[:3 * rs.n_blobs]).cpu().numpy(), FT.reshape(-1)) < 1e-6
  if n_bodies <= 30:
    # small case: dense direct solve with the oracle's M as ground truth
    N = rs.n_blobs
    r_eff, bdiag, _ = oracle.wall_regularisation(rs.r_vectors, a)   # lowest blobs sit below z = a
    B = np.repeat(bdiag, 3)
    M = B[:, None] * oracle.dense("tt", 1, r_eff, eta, a) * B[None, :]
    K = np.zeros((3 * N, 6 * n_bodies))
    for k in range(n_bodies):
      K[36 * k:36 * (k + 1), 6 * k:6 * k + 6] = rs.groups[0].K[k].cpu().numpy()
    A = np.block([[M, -K], [-K.T, np.zeros((6 * n_bodies, 6 * n_bodies))]])
    sol = np.linalg.solve(A, rhs.cpu().numpy())
    assert rel_err(U.reshape(-1), sol[3 * N:]) < 1e-6
  else:
    # rollers driven by a torque about +y translate along +x on average (the roller instability's base flow)
    assert U[:, 0].mean() > 0
  rs.close()


def test_lanczos_golden_on_gpu():
  """Reference Lanczos output (golden g6) with the HIP matvec as mobility_mult, vectors on the device."""
  import torch
  from rigidmultiblobswall_amd import MobilityContext
  from rigidmultiblobswall_amd.stochastic import stochastic_forcing_lanczos
  d = np.load(os.path.join(GOLDEN, "g6_lanczos.npz"))
  eta, a = float(d["eta"]), float(d["a"])
  ctx = MobilityContext(0)
  ctx.set_positions(torch.as_tensor(d["r_vectors"].reshape(-1), device="cuda"), a, wall=True)
  for tol in (1e-6, 1e-10):
    noise, its = stochastic_forcing_lanczos(factor=0.7, tolerance=tol, z=torch.as_tensor(d["z"], device="cuda"),
                                            mobility_mult=lambda v: ctx.matvec_device("tt", v.contiguous(), eta))
    assert abs(its - int(d["iterations_tol%g" % tol])) <= 1
    assert rel_err(noise.cpu().numpy(), d["noise_tol%g" % tol]) < 20 * tol
  ctx.close()


@pytest.mark.parametrize("N", [20000, 262144])
def test_lanczos_large_covariance_identity(N):
  """configs[4]-sized Brownian forcing (2.6e5 blobs): |M^1/2 z|^2 == z.M.z, the defining property."""
  import torch
  from rigidmultiblobswall_amd import MobilityContext
  from test_gpu_parity import d2_cloud
  r, _, eta, a = d2_cloud(N, seed=60)
  ctx = MobilityContext(0)
  ctx.set_positions(torch.as_tensor(r.reshape(-1), device="cuda"), a, wall=True)
  from rigidmultiblobswall_amd.stochastic import stochastic_forcing_lanczos
  z = torch.randn(3 * N, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
  mult = lambda v: ctx.matvec_device("tt", v.contiguous(), eta)  # noqa: E731
  t0 = time.time()
  tol = 1e-6 if N <= 20000 else 1e-3      # iteration count grows with the spread of M's spectrum (N-dependent)
  noise, its = stochastic_forcing_lanczos(factor=1.0, tolerance=tol, mobility_mult=mult, z=z)
  torch.cuda.synchronize()
  dt = time.time() - t0
  zMz = float(torch.dot(z, mult(z)))
  nn = float(torch.dot(noise, noise))
  print("lanczos N=%d iterations=%d time=%.3fs  |M^1/2 z|^2 / z.M.z - 1 = %.2e" % (N, its, dt, nn / zMz - 1))
  assert its < 300 and abs(nn / zMz - 1) < 10 * tol
  ctx.close()


@pytest.mark.parametrize("res", ["low", "mid", "high"])
def test_pair_active_rods_pinned_reference_velocities_gpu(res):
  """multi_bodies/examples/pair_active_rods/run_<res>_res.velocity.dat.reference, through the HIP matvec."""
  from rigidmultiblobswall_amd.rigid import RigidSuspension
  d = np.load(os.path.join(GOLDEN, "g7_pair_active_rods_%s.npz" % res))
  nb = len(d["locations"])
  rs = RigidSuspension([d["reference_configuration"]] * nb, d["locations"], d["quaternions"], float(d["a"]), float(d["eta"]))
  U, lam, info = rs.solve_mobility_problem(slip=d["slip"], force_torque=d["force_torque"], tol=1e-10)
  ref = d["velocities_reference"]
  print("rods %s: %d blobs, %d iterations, max |U - U_ref| / max|U_ref| = %.2e, omega_z = %.9f (ref %.9f)" %
        (res, rs.n_blobs, info["iterations"], np.abs(U - ref).max() / np.abs(ref).max(), U[0, 5], ref[0, 5]))
  assert info["converged"]
  assert np.abs(U - ref).max() < 2e-8 * np.abs(ref).max()
  rs.close()


def test_config3_solve_is_repeatable():
  """Round 1 met run-to-run differences in this solve and traced them (A/B only) to back-to-back batched
  torch.cholesky_solve calls on this torch/ROCm build; the preconditioner has applied explicit inverses through batched
  GEMMs since.  Guard: two independent builds + solves of the config-3 system (fresh RigidSuspension each, deterministic
  pair sweep so that atomics ordering is not a variable) must agree to 1e-10 in U and use the same iteration count.
  If this ever fails: diagnose from that failure (tools/stress_repeatability.py isolates dense blocks, Cholesky,
  preconditioner and operator), do not re-run."""
  import torch
  from rigidmultiblobswall_amd import structures as st
  from rigidmultiblobswall_amd.rigid import RigidSuspension
  R, eta, n_bodies = 1.0155, 0.957e-3, 2048
  shell = st.icosahedron_shell(0.792079207921 * R)
  a = st.min_blob_separation(shell) / 2
  loc, q, _ = st.roller_monolayer(n_bodies, radius=R, seed=5)
  FT = np.zeros((n_bodies, 6))
  FT[:, 2] = -0.05
  FT[:, 4] = 1.0
  runs = []
  for _ in range(2):
    rs = RigidSuspension([shell] * n_bodies, loc, q, a, eta)
    rs.ctx.set_option("deterministic", 1)
    U, lam, info = rs.solve_mobility_problem(force_torque=FT, tol=1e-8)
    torch.cuda.synchronize()
    runs.append((U.copy(), lam.copy(), info["iterations"]))
    rs.close()
  assert runs[0][2] == runs[1][2]
  assert rel_err(runs[1][0], runs[0][0]) < 1e-10
  assert rel_err(runs[1][1], runs[0][1]) < 1e-10


def test_mixed_precision_solve_meets_the_same_tolerance():
  """Iterative refinement with the fp32 pair sweep inside: the fp64 residual meets the tolerance of the plain solve
  and the body velocities agree to that tolerance; the option leaves the context in double precision."""
  import torch
  from rigidmultiblobswall_amd import structures as st
  from rigidmultiblobswall_amd.rigid import RigidSuspension
  R, eta = 1.0155, 0.957e-3
  shell = st.icosahedron_shell(0.792079207921 * R)
  a = st.min_blob_separation(shell) / 2
  nb = 300
  loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=11)
  FT = np.zeros((nb, 6)); FT[:, 2] = -0.05; FT[:, 4] = 1.0
  rs = RigidSuspension([shell] * nb, loc, quat, a, eta)
  try:
    # the fp64 product BEFORE any mixed solve is the yardstick for "the option did not leak"
    x = torch.randn(3 * rs.n_blobs, dtype=torch.float64, device="cuda")
    y64 = rs.mobility_times_lambda(x).clone()
    U, lam, info = rs.solve_mobility_problem(force_torque=FT, tol=1e-8)
    U2, lam2, info2 = rs.solve_mobility_problem(force_torque=FT, tol=1e-8, mixed_precision=True)
    assert info["converged"] and info2["converged"]
    assert info2["residual"] <= 1e-8 and 1 <= info2["outer_iterations"] <= 4
    assert np.linalg.norm(U2 - U) < 1e-6 * np.linalg.norm(U)
    assert np.linalg.norm(lam2 - lam) < 1e-5 * np.linalg.norm(lam)
    # back in fp64 without anybody resetting the option: the product agrees with the one taken before to rounding
    assert rs.ctx.get_option("precision") == 64
    y_after = rs.mobility_times_lambda(x)
    assert float(torch.linalg.norm(y_after - y64)) <= 1e-13 * float(torch.linalg.norm(y64))
    # a caller that had selected single precision gets it back (RigidIntegrator.precision = 'single' + mixed solves)
    rs.ctx.set_option("precision", 32)
    y32 = rs.mobility_times_lambda(x).clone()
    e32 = float(torch.linalg.norm(y32 - y64)) / float(torch.linalg.norm(y64))
    assert 1e-9 < e32 < 1e-4, e32
    U3, _, info3 = rs.solve_mobility_problem(force_torque=FT, tol=1e-8, mixed_precision=True)
    assert info3["converged"] and info3["residual"] <= 1e-8         # the refinement's residuals ran in fp64 all the same
    assert np.linalg.norm(U3 - U) < 1e-6 * np.linalg.norm(U)
    assert rs.ctx.get_option("precision") == 32
    assert float(torch.linalg.norm(rs.mobility_times_lambda(x) - y32)) <= 1e-12 * float(torch.linalg.norm(y32))
    rs.ctx.set_option("precision", 64)
  finally:
    rs.close()


def test_lagged_gmres_bookkeeping_equals_the_synchronous_loop():
  """The host side of GMRES (Givens rotations, convergence test) runs one iteration behind the device by default on a
  GPU (`lag`): iterates, stopping rule and iteration counts must be those of the synchronous loop
  (general_application_utils.py:514-635) -- with restarts, with an initial guess, when the solve converges exactly at
  a restart boundary, and on a solve that converges in very few iterations (where the loop turns synchronous early)."""
  import torch
  from rigidmultiblobswall_amd.rigid import gmres_right_preconditioned
  dev = torch.device("cuda:0")
  rng = np.random.RandomState(5)
  n = 400
  Q, _ = np.linalg.qr(rng.randn(n, n))
  eig = np.concatenate([np.linspace(1.0, 3.0, n - 30), 10.0 ** rng.uniform(-2, 1.5, 30)])
  A_h = (Q * eig) @ Q.T + 0.05 * rng.randn(n, n) / np.sqrt(n)          # non-symmetric, a few outlying eigenvalues
  A = torch.as_tensor(A_h, device=dev)
  Pinv = torch.as_tensor(np.diag(1.0 / np.diag(A_h)), device=dev)
  b = torch.as_tensor(rng.randn(n), device=dev)
  x0 = torch.as_tensor(0.1 * rng.randn(n), device=dev)
  op = lambda v: A @ v
  pc = lambda v: Pinv @ v

  def both(strict=True, **kw):
    """strict: one restart cycle -- same iterates to rounding.  Otherwise (many short cycles: restarted GMRES amplifies
    rounding from cycle to cycle) the first two cycles are compared tightly and the end result at the solver's level."""
    out = []
    for lag in (False, True):
      x, info = gmres_right_preconditioned(op, pc, b, lag=lag, **kw)
      torch.cuda.synchronize()
      out.append((x.cpu().numpy(), info))
    (xs, i_s), (xl, i_l) = out
    assert i_l["converged"] == i_s["converged"]
    assert i_l["discarded_sweeps"] <= 1 and i_s["discarded_sweeps"] == 0
    if strict:
      assert i_l["iterations"] == i_s["iterations"], (kw, i_s["iterations"], i_l["iterations"])
      assert np.allclose(i_l["history"], i_s["history"], rtol=1e-7, atol=0)
      # same iterates up to the rounding of w / |w| (device scalar) against w * (1 / |w|) (host scalar), amplified by
      # the conditioning of the system (~3e3 here)
      assert rel_err(xl, xs) < 1e-8, rel_err(xl, xs)
    else:
      # (a residual recomputed at a restart close to convergence, b - A x with |b - A x| ~ 1e-10 |b|, is itself only
      # accurate to ~1e-3 relative: compare where the residual is still well above the rounding of x)
      head = min(2 * kw["restart"], len(i_s["history"]), len(i_l["history"]))
      hs_, hl_ = np.array(i_s["history"][:head]), np.array(i_l["history"][:head])
      big = hs_ > 1e-6
      assert np.allclose(hl_[big], hs_[big], rtol=1e-6, atol=0)
      assert abs(i_l["iterations"] - i_s["iterations"]) <= max(2, i_s["iterations"] // 50), (kw, i_s["iterations"], i_l["iterations"])
      assert rel_err(xl, xs) < 1e-6, rel_err(xl, xs)
    return i_s

  full = both(tol=1e-10, restart=200)
  assert full["converged"] and full["iterations"] > 20
  both(strict=False, tol=1e-10, restart=7)                    # many restart cycles
  both(strict=False, tol=1e-10, restart=7, x0=x0)             # warm start + restarts
  both(tol=1e-10, restart=200, x0=x0)                         # warm start, one cycle
  both(tol=1e-10, restart=full["iterations"])                 # converges exactly when the cycle ends
  both(strict=False, tol=1e-10, restart=full["iterations"] - 1)   # one iteration into the next cycle
  both(tol=1e-1, restart=60)                                  # a handful of iterations
  both(tol=1e-10, restart=60, maxiter=9)                      # stops on the iteration cap, not converged


def _shell_suspension(nb, seed=5, **kw):
  import torch
  from rigidmultiblobswall_amd import structures as st
  from rigidmultiblobswall_amd.rigid import RigidSuspension
  R, eta = 1.0155, 0.957e-3
  shell = st.icosahedron_shell(0.792079207921 * R)
  a = st.min_blob_separation(shell) / 2
  loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=seed)
  return RigidSuspension([shell] * nb, loc, quat, a, eta, device=torch.device("cuda:0"), **kw), loc, quat


def test_captured_arnoldi_iterations_equal_the_eager_loop():
  """Small systems replay one captured hipGraph per Arnoldi iteration from the third solve on (rigid._ArnoldiGraphs).
  The iterates are those of the eager loop -- same kernels, same order: iteration counts and residual histories equal,
  solutions equal to rounding -- while bodies move, the preconditioner is rebuilt (K and the blocks are rewritten in
  place under the graphs), the right-hand side changes, with restarts and with an initial guess; a context option that
  changes the launches drops the graphs."""
  import torch
  nb = 40
  eager, loc, quat = _shell_suspension(nb)
  graphed, _, _ = _shell_suspension(nb)
  eager.gmres_graph, graphed.gmres_graph = False, True
  rng = np.random.RandomState(2)
  n3 = 3 * eager.n_blobs
  try:
    replays = []
    for step in range(7):
      if step in (2, 4, 5):           # move the bodies; rebuild the preconditioner on some of the moves only
        loc = loc + 0.02 * rng.randn(*loc.shape) * np.array([1.0, 1.0, 0.2])
        for s in (eager, graphed):
          s.set_configuration(loc, quat)
          if step != 5:
            s.build_preconditioner()
      rhs = torch.as_tensor(rng.randn(eager.size), device="cuda:0")
      kw = dict(tol=1e-9, restart=60)
      if step == 3:
        kw = dict(tol=1e-9, restart=60, x0=torch.as_tensor(1e-3 * rng.randn(eager.size), device="cuda:0"))
      if step == 6:
        kw = dict(tol=1e-9, restart=7)      # another restart length: a second workspace, eager again at first
      m0 = (eager.matvec_count, graphed.matvec_count)
      xe, ie = eager.solve(rhs, **kw)
      xg, ig = graphed.solve(rhs, **kw)
      torch.cuda.synchronize()
      assert "graph_replays" in ig and "graph_replays" not in ie
      replays.append(ig["graph_replays"])
      assert ig["converged"] and ie["converged"]
      if kw["restart"] == 60:
        assert ig["iterations"] == ie["iterations"], (step, ie["iterations"], ig["iterations"])
        assert np.allclose(ig["history"], ie["history"], rtol=1e-6, atol=0), step
        assert eager.matvec_count - m0[0] == graphed.matvec_count - m0[1], step      # replays are counted as products
      assert rel_err(xg.cpu().numpy(), xe.cpu().numpy()) < 1e-7, (step, rel_err(xg.cpu().numpy(), xe.cpu().numpy()))
      # and it IS a solution: true residual of the graphed result by the eager operator
      res = float(torch.linalg.norm(eager.apply_operator(xg) - rhs) / torch.linalg.norm(rhs))
      assert res < 5e-9, (step, res)
    assert replays[0] == 0 and replays[1] == 0 and all(r > 0 for r in replays[2:6]), replays
    ws = graphed._arnoldi_ws
    # a launch-changing option of the context: the signature differs, the graphs go, the next solves are eager again
    graphed.ctx.set_option("deterministic", 2)
    rhs = torch.as_tensor(rng.randn(eager.size), device="cuda:0")
    xg, ig = graphed.solve(rhs, tol=1e-9)
    xe, ie = eager.solve(rhs, tol=1e-9)
    assert ig["graph_replays"] == 0 and rel_err(xg.cpu().numpy(), xe.cpu().numpy()) < 1e-7
    graphed.ctx.set_option("deterministic", 0)
  finally:
    eager.close()
    graphed.close()


def test_captured_iterations_survive_longer_solves_and_a_shared_context():
  """the captured graphs hold the addresses of the library's internal buffers by value.  (i) The
  Gram-Schmidt scratch used to grow with the iteration index, so a solve that reached a higher index than the captured
  ones freed the buffer under them: capture with short solves, run a longer one, replay a short one -- the scratch must
  not have moved (buffers_signature constant) and the results must equal the eager loop.  (ii) A context shared by two
  suspensions of different sizes (n -> n' -> n) moves the packed positions / accumulators: the stale graphs are dropped
  before a replay (stale_drops) instead of being replayed into freed memory."""
  import torch
  from rigidmultiblobswall_amd import MobilityContext
  nb = 40
  eager, loc, quat = _shell_suspension(nb)
  graphed, _, _ = _shell_suspension(nb)
  eager.gmres_graph, graphed.gmres_graph = False, True
  rng = np.random.RandomState(4)

  def both(tol, label, sus=(None, None)):
    e, g = sus if sus[0] is not None else (eager, graphed)
    rhs = torch.as_tensor(rng.randn(e.size), device="cuda:0")
    xe, ie = e.solve(rhs, tol=tol, restart=60)
    xg, ig = g.solve(rhs, tol=tol, restart=60)
    torch.cuda.synchronize()
    assert ig["iterations"] == ie["iterations"], (label, ie["iterations"], ig["iterations"])
    assert rel_err(xg.cpu().numpy(), xe.cpu().numpy()) < 1e-7, (label, rel_err(xg.cpu().numpy(), xe.cpu().numpy()))
    res = float(torch.linalg.norm(e.apply_operator(xg) - rhs) / torch.linalg.norm(rhs))
    assert res < 5 * tol, (label, res)
    return ig

  try:
    short = [both(1e-2, "short %d" % k) for k in range(4)]          # captures the first few iteration indices
    assert short[-1]["graph_replays"] > 0
    sig = graphed.ctx.buffers_signature()
    k_short = short[-1]["iterations"]
    long_ = both(1e-10, "long")                                     # reaches indices never seen before: eager there
    assert long_["iterations"] >= k_short + 5, (k_short, long_["iterations"])
    assert graphed.ctx.buffers_signature() == sig                   # nothing the graphs point at has moved
    again = both(1e-2, "short again")                               # replays the graphs captured BEFORE the long solve
    assert again["graph_replays"] > 0 and graphed._arnoldi_ws.stale_drops == 0
    both(1e-10, "long again")
    both(1e-10, "long, captured now")
    assert graphed.ctx.buffers_signature() == sig
  finally:
    eager.close()
    graphed.close()

  # (ii) one context, two suspensions: 20 shells, then 60 shells (every per-blob buffer grows), then the 20 again
  ctx = MobilityContext(0)
  small_e, _, _ = _shell_suspension(20, seed=3)
  big_e, _, _ = _shell_suspension(60, seed=4)
  small_g, loc_s, quat_s = _shell_suspension(20, seed=3, ctx=ctx)
  big_g, loc_b, quat_b = _shell_suspension(60, seed=4, ctx=ctx)
  small_e.gmres_graph = big_e.gmres_graph = False
  small_g.gmres_graph = big_g.gmres_graph = True
  try:
    small_g.set_configuration(loc_s, quat_s)          # the shared context holds the LAST configuration bound: rebind
    for k in range(4):
      ig = both(1e-8, "small %d" % k, (small_e, small_g))
    assert ig["graph_replays"] > 0
    sig_small = ctx.buffers_signature()
    big_g.set_configuration(loc_b, quat_b)
    for k in range(4):
      ig = both(1e-8, "big %d" % k, (big_e, big_g))
    assert ig["graph_replays"] > 0 and ctx.buffers_signature() != sig_small
    small_g.set_configuration(loc_s, quat_s)
    ig = both(1e-8, "small after big", (small_e, small_g))
    assert small_g._arnoldi_ws.stale_drops == 1 and ig["graph_replays"] == 0      # noticed BEFORE the first replay
    for k in range(3):                                                            # eager, eager, captured afresh
      ig = both(1e-8, "small after big, %d" % k, (small_e, small_g))
    assert ig["graph_replays"] > 0 and small_g._arnoldi_ws.stale_drops == 1
    # and the big one again: nothing moved since its captures (buffers only grow), its graphs are still good
    big_g.set_configuration(loc_b, quat_b)
    ig = both(1e-8, "big after small", (big_e, big_g))
    assert ig["graph_replays"] > 0 and big_g._arnoldi_ws.stale_drops == 0
  finally:
    for s_ in (small_e, big_e, small_g, big_g):
      s_.close()
    ctx.close()


def test_native_gmres_loop_equals_the_python_loop():
  """rmb_rigid_gmres_device (the whole right-preconditioned GMRES in one library call: rotations and convergence test in
  C, one iteration behind the device) against rigid.py's loop over rmb_rigid_arnoldi_step_device: same iteration counts,
  residual histories and solutions -- full cycle, several restarts, an iteration cap, a loose and a tight tolerance --
  and the result solves the system (true residual by the separate operator)."""
  import torch
  nb = 40
  nat, loc, quat = _shell_suspension(nb)
  pyl, _, _ = _shell_suspension(nb)
  pyl.native_gmres = False
  rng = np.random.RandomState(5)
  try:
    for kw in (dict(tol=1e-9, restart=60), dict(tol=1e-9, restart=5), dict(tol=1e-2, restart=60), dict(tol=1e-12, restart=7),
               dict(tol=1e-10, restart=60, maxiter=9), dict(tol=1e-10, restart=4, maxiter=10)):
      rhs = torch.as_tensor(rng.randn(nat.size), device="cuda:0")
      m0 = (nat.matvec_count, pyl.matvec_count)
      xn, inn = nat.solve(rhs, **kw)
      xp, ip = pyl.solve(rhs, **kw)
      assert inn.get("native_gmres") and "native_gmres" not in ip and ip.get("native_steps", 0) > 0
      assert inn["iterations"] == ip["iterations"], (kw, inn["iterations"], ip["iterations"])
      assert inn["converged"] == ip["converged"] and len(inn["history"]) == inn["iterations"]
      assert np.allclose(inn["history"], ip["history"], rtol=1e-6, atol=1e-13), kw       # (round-off level near 1e-12)
      assert rel_err(xn.cpu().numpy(), xp.cpu().numpy()) < 1e-8, (kw, rel_err(xn.cpu().numpy(), xp.cpu().numpy()))
      assert nat.matvec_count - m0[0] == pyl.matvec_count - m0[1], kw       # same number of operator applications
      if inn["converged"]:
        true_res = float(torch.linalg.norm(pyl.apply_operator(xn) - rhs) / torch.linalg.norm(rhs))
        assert true_res < 5 * kw["tol"] + 1e-13, (kw, true_res)
    # a zero right-hand side and moved bodies (the blocks are rewritten in place; the library reads them afresh)
    loc2 = loc + 0.03 * rng.randn(*loc.shape) * np.array([1.0, 1.0, 0.2])
    for s_ in (nat, pyl):
      s_.set_configuration(loc2, quat); s_.build_preconditioner()
    rhs = torch.as_tensor(rng.randn(nat.size), device="cuda:0")
    xn, inn = nat.solve(rhs, tol=1e-9)
    xp, ip = pyl.solve(rhs, tol=1e-9)
    assert inn["iterations"] == ip["iterations"] and rel_err(xn.cpu().numpy(), xp.cpu().numpy()) < 1e-8
    x0, i0 = nat.solve(torch.zeros(nat.size, dtype=torch.float64, device="cuda:0"), tol=1e-9)
    assert i0["iterations"] == 0 and float(x0.abs().max()) == 0.0
    # the fused launch (normalisation + next step's preconditioner) off: same arithmetic in one launch more
    xf, inf = nat.solve(rhs, tol=1e-9, restart=7)
    for opts in ({"gmres_fuse_pc": 0}, {"gmres_fuse_dots": 0}, {"krylov_low_sync": 0}, {"krylov_low_sync": 0, "gmres_fuse_pc": 0}):
      # (fuse_pc off: separate normalisation and preconditioner launches, and with them the separate dots; fuse_dots off: the
      #  operator's finishing launch without the first pass's dots)
      for key, val in opts.items():
        nat.ctx.set_option(key, val)
      xs, ins = nat.solve(rhs, tol=1e-9, restart=7)
      for key in opts:
        nat.ctx.set_option(key, 1)
      assert ins["iterations"] == inf["iterations"] and rel_err(xs.cpu().numpy(), xf.cpu().numpy()) < 1e-9, opts
      assert np.allclose(ins["history"], inf["history"], rtol=1e-6, atol=1e-13), opts
  finally:
    nat.close(); pyl.close()


def test_native_loops_cover_the_references_42_blob_shells():
  """Bodies of 33 .. 64 blobs (the reference's shell_N_42 structures) take the library's block kernels with wave = row
  (csrc/rmb_krylov.hip two_by_two_rows) and therefore the one-call GMRES and the native Lanczos loop, instead of batched
  GEMMs under the Python loops: same iteration counts and answers as that path (native_products_max_blobs = 32 restores it)."""
  import os
  import torch
  from rigidmultiblobswall_amd import structures as st
  from rigidmultiblobswall_amd.rigid import RigidSuspension
  shell = np.load(os.path.join(os.path.dirname(__file__), "golden", "g9_rigid_det_euler_42blob_shells.npz"))["vertex_shell42"]
  assert shell.shape == (42, 3)
  R, eta, nb = 1.0155, 0.957e-3, 24
  a = st.min_blob_separation(shell) / 2
  loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=9)
  nat = RigidSuspension([shell] * nb, loc, quat, a, eta, device=torch.device("cuda:0"))
  ref = RigidSuspension([shell] * nb, loc, quat, a, eta, device=torch.device("cuda:0"))
  ref.native_products_max_blobs = 32
  rng = np.random.RandomState(3)
  try:
    assert nat._native_products() is nat.ctx and ref._native_products() is None
    for kw in (dict(tol=1e-9, restart=60), dict(tol=1e-6, restart=6)):
      rhs = torch.as_tensor(rng.randn(nat.size), device="cuda:0")
      xn, inn = nat.solve(rhs, **kw)
      xr, ir = ref.solve(rhs, **kw)
      assert inn.get("native_gmres") and not ir.get("native_gmres") and not ir.get("native_steps")
      assert inn["iterations"] == ir["iterations"], (kw, inn["iterations"], ir["iterations"])
      assert np.allclose(inn["history"], ir["history"], rtol=1e-6, atol=1e-13), kw
      assert rel_err(xn.cpu().numpy(), xr.cpu().numpy()) < 1e-8
      true_res = float(torch.linalg.norm(ref.apply_operator(xn) - rhs) / torch.linalg.norm(rhs))
      assert true_res < 5 * kw["tol"], (kw, true_res)
    x = torch.as_tensor(rng.randn(nat.size), device="cuda:0")
    assert rel_err(nat.apply_operator(x).cpu().numpy(), ref.apply_operator(x).cpu().numpy()) < 1e-13
    assert rel_err(nat.apply_preconditioner(x).cpu().numpy(), ref.apply_preconditioner(x).cpu().numpy()) < 1e-11
    z = torch.as_tensor(rng.randn(3 * nat.n_blobs), device="cuda:0")
    fa, ia = nat.stochastic_forcing(z, 1.3, tol=1e-8)
    fb, ib = ref.stochastic_forcing(z, 1.3, tol=1e-8)
    assert ia == ib and rel_err(fa.cpu().numpy(), fb.cpu().numpy()) < 1e-10, (ia, ib)
    assert nat.lanczos_native_loop_calls == 1 and ref.lanczos_native_loop_calls == 0
  finally:
    nat.close(); ref.close()


def test_native_lanczos_loop_equals_the_generic_one():
  """RigidSuspension.stochastic_forcing three ways: the whole loop inside the library (rmb_rigid_lanczos_device: tridiagonal
  eigen-solve and stopping rule in C, one iteration behind the device), one rmb_rigid_lanczos_step_device call per iteration
  under the Python loop (coefficients through mapped memory), and the generic coroutine loop: same iteration count, same
  noise to rounding, at several tolerances, with an iteration cap; a workspace with too few basis rows hands the forcing
  back to the generic loop from either native path."""
  import torch
  nat, _, _ = _shell_suspension(40)
  stp, _, _ = _shell_suspension(40)
  gen, _, _ = _shell_suspension(40)
  stp.native_lanczos_loop = False
  gen.native_lanczos = False
  rng = np.random.RandomState(11)
  try:
    for tol, factor in ((1e-3, 1.0), (1e-6, 0.37), (1e-10, 2.5)):
      z = torch.as_tensor(rng.randn(3 * nat.n_blobs), device="cuda:0")
      m0, m1, calls = nat.matvec_count, stp.matvec_count, nat.lanczos_native_loop_calls
      a, ia = nat.stochastic_forcing(z, factor, tol=tol)
      s_, is_ = stp.stochastic_forcing(z, factor, tol=tol)
      b, ib = gen.stochastic_forcing(z, factor, tol=tol)
      assert ia == ib == is_ and ia >= 2, (tol, ia, is_, ib)
      assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-11, (tol, rel_err(a.cpu().numpy(), b.cpu().numpy()))
      assert rel_err(s_.cpu().numpy(), b.cpu().numpy()) < 1e-11
      assert rel_err(a.cpu().numpy(), s_.cpu().numpy()) < 1e-12            # same device work, eigen-solvers differ in rounding
      assert nat.matvec_count - m0 in (ia + 1, ia + 2)          # its + 1 products as the generic loop, + the discarded one
      assert stp.matvec_count - m1 in (ia + 1, ia + 2)
      assert nat.lanczos_native_loop_calls == calls + 1 and stp.lanczos_native_loop_calls == 0
      assert stp._lanczos_ws is not None and getattr(nat, "_lanczos_ws", None) is None
    # the defining identity on the library's loop alone: with w = L^-1 noise = (P^T M P)^{1/2} z, |w|^2 = (P z).M.(P z), P = L^-T
    z = torch.as_tensor(rng.randn(3 * nat.n_blobs), device="cuda:0")
    a, ia = nat.stochastic_forcing(z, 1.0, tol=1e-10)
    w = nat._blockdiag(a, "Linv")
    Pz = nat._blockdiag(z, "Linv", transpose=True)
    zMz = float(torch.dot(Pz, nat.ctx.matvec_device("tt", Pz.contiguous(), nat.eta)))
    assert abs(float(torch.dot(w, w)) / zMz - 1.0) < 1e-8, float(torch.dot(w, w)) / zMz - 1.0
    # the fused launches of the library's step off (finalize + L^-1, normalisation + next L^-T): same arithmetic
    a0, i0 = nat.stochastic_forcing(z, 1.0, tol=1e-10)
    for keys in (("lanczos_fuse_finish", "gmres_fuse_pc"), ("gmres_fuse_dots",), ("gmres_fuse_pc",), ("krylov_low_sync",), ("krylov_low_sync", "gmres_fuse_pc")):
      for key in keys:
        nat.ctx.set_option(key, 0)
      a1, i1 = nat.stochastic_forcing(z, 1.0, tol=1e-10)
      for key in keys:
        nat.ctx.set_option(key, 1)
      assert i0 == i1 == ia and rel_err(a0.cpu().numpy(), a1.cpu().numpy()) < 1e-13, keys
    assert rel_err(a0.cpu().numpy(), a.cpu().numpy()) < 1e-13
    # too few basis rows: both native paths hand the forcing back to the generic loop
    for s in (nat, stp):
      s.lanczos_native_rows = 3
    z = torch.as_tensor(rng.randn(3 * nat.n_blobs), device="cuda:0")
    a, ia = nat.stochastic_forcing(z, 1.0, tol=1e-8)
    s_, is_ = stp.stochastic_forcing(z, 1.0, tol=1e-8)
    b, ib = gen.stochastic_forcing(z, 1.0, tol=1e-8)
    assert ia == ib == is_ and ia > 3 and rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-11 and rel_err(s_.cpu().numpy(), b.cpu().numpy()) < 1e-11
  finally:
    nat.close(); stp.close(); gen.close()


def test_captured_arnoldi_iterations_with_mixed_shapes_and_prescribed_bodies():
  """The general operator path (two body shapes: gathers / scatters, torch.cat; an obstacle with prescribed kinematics)
  under the captured iterations, against the eager loop; and the automatic switch: on below gmres_graph_max_blobs for a
  plain context, off above it and with RMB_GMRES_GRAPH=0."""
  import torch
  from rigidmultiblobswall_amd.rigid import RigidSuspension
  g = np.load(os.path.join(GOLDEN, "g9_rigid_det_euler.npz"))
  refs = [g["vertex_boomerang"]] * 2 + [g["vertex_shell"]] * 3
  loc = np.concatenate([g["locations_boomerang"], g["locations_shell"]])
  quat = np.concatenate([g["quaternions_boomerang"], g["quaternions_shell"]])
  presc = np.array([False, False, True, False, False])
  mk = lambda: RigidSuspension(refs, loc, quat, 0.25, 1.1, device=torch.device("cuda:0"), prescribed=presc)
  eager, graphed = mk(), mk()
  eager.gmres_graph = False
  rng = np.random.RandomState(8)
  try:
    assert graphed.gmres_graph is None and graphed._arnoldi_graphs(60) is not None       # automatic: small system, plain context
    graphed.gmres_graph_max_blobs = 3
    assert graphed._arnoldi_graphs(60) is None
    graphed.gmres_graph_max_blobs = 4096
    os.environ["RMB_GMRES_GRAPH"] = "0"
    try:
      assert graphed._arnoldi_graphs(60) is None
    finally:
      del os.environ["RMB_GMRES_GRAPH"]
    for step in range(5):
      rhs = graphed.prescribe(torch.as_tensor(rng.randn(eager.size), device="cuda:0"))
      xe, ie = eager.solve(rhs, tol=1e-10)
      xg, ig = graphed.solve(rhs, tol=1e-10)
      assert ig["iterations"] == ie["iterations"] and rel_err(xg.cpu().numpy(), xe.cpu().numpy()) < 1e-7
      assert (ig["graph_replays"] > 0) == (step >= 2), (step, ig["graph_replays"])
  finally:
    eager.close()
    graphed.close()


def test_capture_survives_garbage_from_an_earlier_suspension():
  """A suspension that is dropped WITHOUT close() leaves its captured graphs to the cyclic collector; if that runs while
  another suspension captures, hipGraphDestroy is refused mid-capture and the process aborts in a destructor (seen in
  tools/experiments/exp_small_deck_step.py).  Captures therefore run with the collector paused."""
  import gc
  import torch
  rng = np.random.RandomState(0)
  for round_ in range(3):
    s, _, _ = _shell_suspension(6, seed=round_)
    s.gmres_graph = True
    s._self_cycle = s                       # make it cyclic garbage on purpose, graphs alive, never closed
    for k in range(4):
      x, info = s.solve(torch.as_tensor(rng.randn(s.size), device="cuda:0"), tol=1e-9)
    assert info["graph_replays"] > 0
    del s
    gc.set_threshold(1)                     # collect at every opportunity during the next round's captures
  gc.set_threshold(700, 10, 10)
  gc.collect()
  torch.cuda.synchronize()
