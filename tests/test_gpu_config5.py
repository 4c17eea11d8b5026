"""BASELINE.json configs[4] at full size under pytest: ONE Brownian time step of the 2.6e5-blob suspension --
forces kernel + mobility products + Lanczos M^{1/2} z (+ 3 GMRES rigid solves for the multiblob variant) + random
finite differences -- once for single-blob rollers (quaternion_integrator_rollers.py:251-302) and once for 21 845
twelve-blob shells under stochastic_Slip_Trapz (quaternion_integrator_multi_bodies.py:925-1045; main loop
multi_bodies.py:1511).  Checks that do not depend on the size: no rejected step, solver iteration counts inside stated
bounds, the forces and the fused product of the step's configuration against the oracle on a sample of targets, and
the whole step equal (<= 1e-10 of the displacement) to the same step through a 2-shard stand-in of the multi-GPU path
(tests/_shard_standin.py: both ranks' launches on this GPU, summed as the all-reduce would)."""
import math

import numpy as np
import pytest
import torch

from conftest import rel_err
from _shard_standin import replicated_standin

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sample(n, k=48, seed=1):
  tg = np.random.RandomState(seed).choice(n, k, replace=False)
  tg[:4] = (0, 63, n - 1, n - 64)         # tile edges and the partial last tile
  return np.unique(tg)


def test_config5_rollers_one_brownian_step(oracle):
  from rigidmultiblobswall_amd import structures as st
  from rigidmultiblobswall_amd.rollers import RollersIntegrator
  n, a, eta, dt = 262144, 0.656, 1.0e-3, 0.016
  loc, _, _ = st.roller_monolayer(n, radius=a, seed=7)
  eps, b = 0.0165677856, 0.0656

  def make(ctx):
    integ = RollersIntegrator(loc, "stochastic_adams_bashforth_rollers", a, eta, tolerance=1e-3, device=DEV, ctx=ctx, seed=11)
    integ.kT, integ.g = 0.0041419464, 0.0024892
    integ.repulsion_strength = integ.repulsion_strength_wall = eps
    integ.debye_length = integ.debye_length_wall = b
    integ.omega_one_roller = np.array([0.0, 62.8, 0.0])
    return integ

  one = make(None)
  # the step's own building blocks at the start configuration against the oracle, on a sample of targets
  tg = _sample(n)
  F = one.calc_blob_blob_forces(one.location).cpu().numpy()
  F_ref = oracle.calc_blob_blob_forces_targets_oracle(loc, tg, repulsion_strength=eps, debye_length=b, blob_radius=a)
  assert rel_err(F[tg], F_ref) < 1e-12
  force = (one.calc_one_blob_forces(one.location) + one.calc_blob_blob_forces(one.location)).reshape(-1)
  torque = one.get_torque()
  v, _ = one.compute_deterministic_velocity_and_torque()
  r_eff, bdiag, _ = oracle.wall_regularisation(loc, a)
  f_h, t_h = force.cpu().numpy().reshape(-1, 3) * bdiag[:, None], torque.cpu().numpy().reshape(-1, 3) * bdiag[:, None]
  ref = oracle.raw_matvec_targets("tt", 1, r_eff, f_h, eta, a, tg) + oracle.raw_matvec_targets("tr", 1, r_eff, t_h, eta, a, tg)
  ref = (ref.reshape(-1, 3) * bdiag[tg][:, None]).reshape(-1)
  assert rel_err(v.cpu().numpy().reshape(-1, 3)[tg].reshape(-1), ref) < 1e-12
  # one Brownian step (the first one: forward Euler + noise + drift)
  p0 = one.mobility_products
  one.advance_time_step(dt)
  assert one.invalid_configuration_count == 0
  assert 3 <= one.stoch_iterations_count <= 12, one.stoch_iterations_count          # Lanczos at the deck's 1e-3
  assert one.mobility_products - p0 <= one.stoch_iterations_count + 4                 # + fused product + 2 RFD (+ 1 spare)
  moved = float((one.location - torch.as_tensor(loc, device=DEV)).abs().max())
  assert 1e-3 * a < moved < 3 * a       # 1 sigma of the Brownian kick is 0.15 a per component; the max is over 7.9e5 draws
  # the same step with every pair sweep divided over two shards, as two GPUs would run it
  ctx2, backend = replicated_standin(torch.device(DEV), 2)
  two = make(ctx2)
  two.advance_time_step(dt)
  assert backend.launches >= 2 * (two.stoch_iterations_count + 3)
  assert two.invalid_configuration_count == 0 and two.stoch_iterations_count == one.stoch_iterations_count
  assert float((two.location - one.location).abs().max()) <= 1e-10 * moved
  one.close(); ctx2.close()


def test_config5_multiblob_one_slip_trapz_step(oracle):
  from rigidmultiblobswall_amd import structures as st
  from rigidmultiblobswall_amd.rigid_integrator import RigidIntegrator
  R, eta, nb, dt = 1.0155, 0.957e-3, 21845, 0.01
  shell = st.icosahedron_shell(0.792079207921 * R)
  a = st.min_blob_separation(shell) / 2
  loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=5)
  eps, b = 0.0326, 0.0406

  def make(ctx):
    ri = RigidIntegrator([shell] * nb, loc, quat, "stochastic_Slip_Trapz", a, eta, tolerance=1e-4, device=DEV, ctx=ctx, seed=1)
    ri.kT, ri.g = 0.0040749841, 0.0303 / 12
    ri.repulsion_strength_wall = ri.repulsion_strength = eps
    ri.debye_length_wall = ri.debye_length = b
    FT = torch.zeros((nb, 6), dtype=torch.float64, device=DEV)
    FT[:, 4] = 8 * math.pi * eta * R ** 3 * 62.8
    ri.external_force_torque = lambda it: FT
    return ri

  one = make(None)
  n = one.Nblobs
  assert n == 262140
  # blob-level building blocks of the step against the oracle on a sample of targets
  r_blobs = one.susp.blob_positions_device(one.location, one.orientation)[0].cpu().numpy()
  tg = _sample(n)
  lam = torch.randn(3 * n, dtype=torch.float64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
  u = one.susp.mobility_times_lambda(lam).cpu().numpy()
  r_eff, bdiag, _ = oracle.wall_regularisation(r_blobs, a)
  ref = oracle.raw_matvec_targets("tt", 1, r_eff, lam.cpu().numpy().reshape(-1, 3) * bdiag[:, None], eta, a, tg)
  assert rel_err(u.reshape(-1, 3)[tg].reshape(-1), (ref.reshape(-1, 3) * bdiag[tg][:, None]).reshape(-1)) < 1e-12
  one.susp.ctx.set_positions(torch.as_tensor(r_blobs.reshape(-1), device=DEV), a, None, wall=False)
  F = one.susp.ctx.blob_blob_force_device(eps, b, a).cpu().numpy().reshape(-1, 3)
  assert rel_err(F[tg], oracle.calc_blob_blob_forces_targets_oracle(r_blobs, tg, repulsion_strength=eps, debye_length=b,
                                                                    blob_radius=a)) < 1e-12
  one.susp.set_configuration(one.location, one.orientation)
  # one Brownian step: 3 rigid solves + preconditioned Lanczos + 2 RFD products + forces
  one.advance_time_step(dt, step=0)
  assert one.invalid_configuration_count == 0
  assert 30 <= one.det_iterations_count <= 75, one.det_iterations_count         # 3 GMRES solves at 1e-4 (49 in the bench)
  assert 5 <= one.stoch_iterations_count <= 16, one.stoch_iterations_count       # preconditioned Lanczos (9 in the bench)
  moved = float((one.location - torch.as_tensor(loc, device=DEV)).abs().max())
  assert 0 < moved < R
  ctx2, backend = replicated_standin(torch.device(DEV), 2)
  two = make(ctx2)
  two.advance_time_step(dt, step=0)
  assert two.invalid_configuration_count == 0
  # solver tolerance 1e-4 and atomic-order round-off: a solve may stop one iteration apart, the step then differs at
  # the level of the tolerance; with equal counts it agrees to round-off
  same = (two.det_iterations_count, two.stoch_iterations_count) == (one.det_iterations_count, one.stoch_iterations_count)
  assert abs(two.det_iterations_count - one.det_iterations_count) <= 2 and two.stoch_iterations_count == one.stoch_iterations_count
  diff = float((two.location - one.location).abs().max())
  assert diff <= (1e-10 if same else 1e-3) * moved, (diff, moved, same)
  one.close(); two.close(); ctx2.close()
