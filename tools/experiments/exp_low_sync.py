"""Same-run A/B of "krylov_low_sync" (second Gram-Schmidt update + norm by Pythagoras + normalisation in one launch) on small-deck
GMRES solves, preconditioned Lanczos forcings and the roller schemes' plain Lanczos forcing."""
import math, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import structures as st
from rigidmultiblobswall_amd.rigid import RigidSuspension
from rigidmultiblobswall_amd.rollers import RollersIntegrator
R, eta3 = 1.0155, 0.957e-3
shell42 = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "g9_rigid_det_euler_42blob_shells.npz"))["vertex_shell42"]

def ab(ctx, fn, reps):
  res, out = {0: [], 1: []}, {}
  for rnd in range(6):
    for flag in ((0, 1) if rnd % 2 == 0 else (1, 0)):
      ctx.set_option("krylov_low_sync", flag)
      for _ in range(10): fn()
      torch.cuda.synchronize(); t0 = time.perf_counter()
      for _ in range(reps): r = fn()
      torch.cuda.synchronize()
      res[flag].append((time.perf_counter() - t0) / reps * 1e3)
      out[flag] = r
  ctx.set_option("krylov_low_sync", 1)
  return np.median(res[0]), np.median(res[1]), out

for nb, shell in ((16, st.icosahedron_shell(0.792079207921 * R)), (64, st.icosahedron_shell(0.792079207921 * R)), (256, st.icosahedron_shell(0.792079207921 * R)), (24, shell42)):
  a3 = st.min_blob_separation(shell) / 2
  loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=5)
  FT = np.zeros((nb, 6)); FT[:, 2] = -0.05; FT[:, 4] = 1.0
  rs = RigidSuspension([shell] * nb, loc, quat, a3, eta3, device=torch.device("cuda:0"))
  rhs = rs.prescribe(torch.cat([torch.zeros(3 * rs.n_blobs, dtype=torch.float64, device="cuda"), -torch.as_tensor(FT.reshape(-1), device="cuda")]))
  z = torch.randn(3 * rs.n_blobs, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
  t0, t1, o = ab(rs.ctx, lambda: rs.solve(rhs, tol=1e-8), 100)
  d = float((o[0][0] - o[1][0]).abs().max() / o[0][0].abs().max())
  print("bodies %4d x %2d blobs: GMRES solve   separate %.3f ms, low-sync %.3f ms (%d / %d iterations, solutions differ by %.1e)" % (nb, shell.shape[0], t0, t1, o[0][1]["iterations"], o[1][1]["iterations"], d), flush=True)
  t0, t1, o = ab(rs.ctx, lambda: rs.stochastic_forcing(z, 1.0, tol=1e-6), 100)
  d = float((o[0][0] - o[1][0]).abs().max() / o[0][0].abs().max())
  print("bodies %4d x %2d blobs: Lanczos forcing separate %.3f ms, low-sync %.3f ms (%d / %d iterations, noise differs by %.1e)" % (nb, shell.shape[0], t0, t1, o[0][1], o[1][1], d), flush=True)
  rs.close()
a5 = 0.656
for n5 in (1000, 4096):
  loc5, _, _ = st.roller_monolayer(n5, radius=a5, seed=7)
  integ = RollersIntegrator(loc5, "stochastic_adams_bashforth_rollers", a5, 1.0e-3, tolerance=1e-6, device="cuda:0", seed=11)
  integ.kT = 0.0041419464
  integ._bind(integ.location)
  z = torch.randn(3 * n5, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
  def forcing():
    i0 = integ.stoch_iterations_count
    noise = integ._lanczos(lambda v: integ._product("tt", v), 3 * n5, z, 0.016, product="tt")
    return noise, integ.stoch_iterations_count - i0
  t0, t1, o = ab(integ.ctx, forcing, 30)
  d = float((o[0][0] - o[1][0]).abs().max() / o[0][0].abs().max())
  print("%5d rollers: plain Lanczos forcing separate %.3f ms, low-sync %.3f ms (%d / %d iterations, noise differs by %.1e)" % (n5, t0, t1, o[0][1], o[1][1], d), flush=True)
  integ.close()
