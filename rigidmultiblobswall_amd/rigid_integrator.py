"""Device-resident time integrators for suspensions of rigid multiblobs (the callers of the callers of the
hot path: BASELINE.json configs[4] = "GMRES + Lanczos M^{1/2} z + forces kernel" per time step).

Mirror of quaternion_integrator/quaternion_integrator_multi_bodies.py (`QuaternionIntegrator`) for free
rigid bodies: same scheme names, attribute names and random-draw order, so a run of the reference driver with
`seed s` in the deck and a run of this class with `rng=np.random.RandomState(s)` walk the same trajectory up
to solver tolerance (tests/golden/g9_*, recorded from the reference's own multi_bodies.py).

  deterministic_forward_euler (:75)   deterministic_adams_bashforth (:142)   deterministic_midpoint (:188)
  stochastic_EM (:262)                stochastic_first_order_RFD (:326)      stochastic_adams_bashforth (:431)
  stochastic_traction_EM (:626)       stochastic_traction_AB (:803)
  stochastic_Slip_Trapz (:925)        stochastic_GDC_RFD (:1048)             stochastic_Slip_Mid (:1214)

Where things live: locations / quaternions are two tensors in HBM; a step is a sequence of
`RigidSuspension.set_configuration` (batched rotation + K rebuild + position pack), saddle-point solves
(`RigidSuspension.solve`: HIP pair sweeps + batched GEMMs), preconditioned Lanczos
(`RigidSuspension.stochastic_forcing`) and the blob force kernel; the reference loops over Python `Body`
objects for every one of those (e.g. :86-91, :1003-1007).

Not built: articulated bodies / constraints, prescribed kinematics (obstacles), the dense-algebra variants
(`*_dense_algebra`, `Fixman`, `*_DLA`: O(N^3) teaching versions).
"""
import math

import numpy as np
import torch

from .rigid import (RigidSuspension, quaternion_from_rotation_torch, quaternion_multiply_torch,
                    quaternion_rotation_matrix_torch)


def seeded_generator(device, seed, ctx=None):
  """Device generator for a stepper that was not handed a numpy stream.  `seed=None` means what it means in the
  reference (numpy left entropy-seeded, multi_bodies.py:1154-1161): a fresh seed from the OS, so that repeated launches
  form an ensemble; an explicit `seed=` is reproducible.  On several ranks the seed of rank 0 is the one every rank
  uses (replicated steppers must draw identical numbers).  Returns (generator, seed used)."""
  gen = torch.Generator(device=device)
  if seed is None:
    seed = int(np.random.SeedSequence().generate_state(1, dtype=np.uint64)[0] >> 1)
    sync = getattr(ctx, "sync_scalars", None)
    if sync is not None:
      t = torch.tensor([seed], dtype=torch.int64, device=device)
      seed = int(sync(t).item())
  gen.manual_seed(int(seed))
  return gen, int(seed)


def replicate_rng(rng, read, ctx, device):
  """A deck without `seed` / `random_state` gives every process its own entropy-seeded numpy stream; replicated
  steppers (ReplicatedContext) need ONE stream: re-seed from a number rank 0 draws."""
  sync = getattr(ctx, "sync_scalars", None)
  if rng is None or sync is None or read.seed is not None or read.random_state is not None:
    return rng
  t = torch.tensor([int(rng.randint(0, 2 ** 31 - 1))], dtype=torch.int64, device=device)
  return np.random.RandomState(int(sync(t).item()))


def lab_frame_slip(susp, slip_body_frame):
  """Active slip given per blob in the body frame (.slip files), rotated to the bodies' current orientation:
  slip_lab = R(q) slip_body (multi_bodies_functions.py:123-140).  (Nblobs, 3) tensor -> (3 Nblobs,)."""
  out = torch.empty(3 * susp.n_blobs, dtype=torch.float64, device=susp.device)
  R = quaternion_rotation_matrix_torch(susp.orientation)
  for g in susp.groups:
    sb = slip_body_frame[g.blob_idx.reshape(-1)].view(len(g.body_idx), g.n_b, 3)
    out[g.blob_idx3.reshape(-1)] = torch.bmm(sb, R[g.body_idx].transpose(1, 2)).reshape(-1)
  return out


class RigidIntegrator(object):
  """reference_configurations: one (n_b, 3) array per body; locations (nb, 3); quaternions (nb, 4)."""

  def __init__(self, reference_configurations, locations, quaternions, scheme, a, eta, tolerance=None,
               domain="single_wall", periodic_length=None, device="cuda:0", ctx=None, rng=None, seed=None,
               prescribed=None, prescribed_velocity=None):
    if domain not in ("single_wall", "no_wall"):
      raise ValueError("domain must be single_wall or no_wall")
    self.device = torch.device(device)
    self.scheme = scheme
    self.a, self.eta = float(a), float(eta)
    self.domain = domain
    self.periodic_length = np.zeros(3) if periodic_length is None else np.asarray(periodic_length, dtype=np.float64)
    self.susp = RigidSuspension(reference_configurations, locations, quaternions, a, eta, wall=(domain == "single_wall"),
                                periodic_length=self.periodic_length, device=device, ctx=ctx, prescribed=prescribed,
                                prescribed_velocity=prescribed_velocity)
    self.location = self.susp.location.clone()
    self.orientation = self.susp.orientation.clone()
    self.Nblobs, self.Nbodies = self.susp.n_blobs, self.susp.n_bodies
    # body_length = largest blob-blob distance + 2a (body/body.py:218-231), one value per body
    lengths = np.empty(self.Nbodies)
    cache = {}
    for k, ref in enumerate(reference_configurations):
      ref = np.asarray(ref, dtype=np.float64).reshape(len(ref), -1)[:, :3]
      key = ref.tobytes()
      if key not in cache:
        if len(ref) > 2000:
          cache[key] = 10.0
        else:
          d = np.linalg.norm(ref[:, None, :] - ref[None, :, :], axis=-1)
          cache[key] = float(d.max()) + 2 * self.a
      lengths[k] = cache[key]
    self.body_length = torch.as_tensor(lengths, device=self.device)
    # state and counters, names as quaternion_integrator_multi_bodies.py:41-55
    self.velocities_previous_step = None
    self.first_step = True
    self._precision = 'double'
    self.kT = 0.0
    self.tolerance = 1e-08 if tolerance is None else float(tolerance)
    self.rf_delta = 1e-03
    self.invalid_configuration_count = 0
    self.det_iterations_count = 0
    self.stoch_iterations_count = 0
    self.update_PC = 1
    self.warm_start = False
    self.first_guess = None
    # Tolerance of the rigid solve that only produces the random-finite-difference displacement W_RFD of the Slip schemes
    # (None = solver tolerance, as the reference).  That solve sets the direction of a finite difference whose result is
    # a kT-order correction, so a loose value (1e-2) cuts ~20 % of the pair sweeps of a step without visible bias.
    self.rfd_solve_tolerance = None
    # Slip schemes: advance everything that shares the mobility of time level n (M W_slip, the Lanczos forcing(s), the RFD
    # solve) in lockstep, one k-vector pass over the blob pairs per round.
    # Numerically neutral: each solve sees exactly its own GMRES iterates.  None = automatic: from `lockstep_min_blobs`
    # blobs on, where a pass over the pairs costs more than the host work of a solver iteration; below, the solves run one
    # after the other through the library's GMRES / Lanczos loops (tools/experiments/exp_lockstep_small.py, end of round 5,
    # lockstep against sequential per stochastic_Slip_Trapz step: 512 shells 7.11 / 5.75 ms, 768: 10.59 / 9.34, 1024: level,
    # 1536: 25.9 / 27.9, 2048: 41.2 / 46.9, 4096: 151 / 174).
    self.lockstep_solves = None
    self.lockstep_min_blobs = 12500
    self.print_residual = False
    self.max_retries = 1000              # total rejected configurations over the life of the integrator
    self.max_consecutive_retries = 20    # in a row (one step, or its midpoint / predictor stages)
    self.consecutive_rejections = 0
    self.report_rejections = True
    self._pc_built = False
    # force model of multi_bodies_functions.py (gravity + wall repulsion per blob, blob-blob repulsion)
    self.g = 0.0
    self.blob_mass = 1.0
    self.repulsion_strength_wall = 0.0
    self.debye_length_wall = 1.0
    self.repulsion_strength = 0.0
    self.debye_length = 1.0
    # hooks, as the reference's attributes of the same names (tensors in / out)
    self.calc_slip = None                       # callable(integrator) -> (Nblobs, 3) tensor
    self.slip_body_frame = None                 # (Nblobs, 3) tensor: constant active slip in the body frame (.slip files)
    self.calc_blob_forces = self._blob_forces   # callable(r (N,3) tensor) -> (N,3) tensor
    self.external_force_torque = None           # callable(integrator) -> (nb, 6) tensor added to the blob-derived one
    self.preprocess = lambda integrator: None
    self.postprocess = lambda integrator: None
    self.rng = rng
    self._gen = None
    if rng is None:
      self._gen, self.seed = seeded_generator(self.device, seed, ctx)


  @property
  def precision(self):
    """'double' (default) or 'single' -- the reference GPU module's precision switch (mobility_pycuda.py:7-19) for the
    blob mobility products of this integrator: with 'single' every M_tt pass with open boundaries (single-vector and k-vector lockstep
    passes) and the blob-blob forces run the fp32 twins of the pair sweeps (csrc/sym32_kernels.h, symx32_kernels.h: fp32
    pair arithmetic, fp64 accumulation, ~1e-6 relative); pseudo-periodic domains and the O(N) rigid algebra stay fp64.  Meant for the
    Brownian schemes at their loose solver tolerances (1e-3 ... 1e-4), where the product error is two orders below
    the tolerance; the solvers' stopping rules are unchanged.  The random finite differences divide a difference of two
    products by rf_delta: with ~1e-6-accurate products rf_delta must be >= 1e-4 (doc/README.md:512-523 uses 1e-3 for the
    reference's single-precision build, 1e-6 for double) -- the setter (for a stochastic scheme with kT > 0) and every stochastic step raise ValueError otherwise."""
    return self._precision

  @precision.setter
  def precision(self, value):
    if value not in ('single', 'double'):
      raise ValueError("precision must be 'single' or 'double'")
    if value == 'single' and self.kT > 0.0 and self.scheme.startswith("stochastic"):
      # only the stochastic schemes form random finite differences (the per-step check covers later changes)
      from .rollers import _check_rfd_delta_for_single_precision
      _check_rfd_delta_for_single_precision(self.rf_delta)
    self._precision = value
    self.susp.ctx.set_option("precision", 32 if value == 'single' else 64)
  def close(self):
    self.susp.close()

  # ---- plumbing -------------------------------------------------------------------------------------
  def _normal(self, n):
    if self.rng is not None:
      return torch.as_tensor(self.rng.normal(0.0, 1.0, n), dtype=torch.float64, device=self.device)
    return torch.randn(n, dtype=torch.float64, device=self.device, generator=self._gen)

  def _move(self, location, orientation):
    self.susp.set_configuration(location, orientation)

  def _advance(self, location, orientation, velocities, dt):
    """x + v dt and quaternion(omega dt) * q (quaternion_integrator_multi_bodies.py:86-91).  dt may be a per-body
    column for the translation (the RFD displacement is scaled by the body length)."""
    U = velocities.view(-1, 6)
    per_body = isinstance(dt, torch.Tensor)
    helper = self.susp._native_blocks()
    if helper is not None and (not per_body or dt.numel() == U.shape[0]):
      return helper.rigid_advance_device(location.contiguous(), orientation.contiguous(), U.contiguous(), dt)   # one launch
    return location + U[:, 0:3] * dt, quaternion_multiply_torch(quaternion_from_rotation_torch(U[:, 3:6] * dt), orientation)

  def _valid(self, location, orientation):
    """body.check_function (body/body.py:118-140): no blob below the wall plane."""
    if self.domain != "single_wall":
      return True
    r, _ = self.susp.blob_positions_device(location, orientation)
    if bool(torch.any(r[:, 2] < 0.0)):
      # the reference prints 'Invalid configuration' and redraws without bound (body.check_function + the while True of
      # every scheme); here every rejection is reported and a step that fails too often in a row raises
      self.invalid_configuration_count += 1
      self.consecutive_rejections += 1
      if self.report_rejections:
        print("Invalid configuration (rejection %d in a row, %d in total, %d blobs)" %
              (self.consecutive_rejections, self.invalid_configuration_count, r.shape[0]), flush=True)
      if self.consecutive_rejections > self.max_consecutive_retries:
        raise RuntimeError("rigid integrator: %d configurations rejected in a row (a blob below the wall plane in every "
                           "draw): the time step is too large for this configuration" % self.consecutive_rejections)
      if self.invalid_configuration_count > self.max_retries:
        raise RuntimeError("rigid integrator: more than %d rejected configurations" % self.max_retries)
      return False
    return True

  def _refresh_preconditioner(self, step):
    """multi_bodies.py:502: rebuild when step % update_PC == 0 or nothing has been built yet."""
    if (not self._pc_built) or step is None or (step % self.update_PC == 0):
      self.susp.build_preconditioner()
      self._pc_built = True

  # ---- forces ---------------------------------------------------------------------------------------
  def _blob_forces(self, r):
    """One-blob forces (multi_bodies_functions.py:153-188) + blob-blob repulsion (forces_numba.py:12-55)."""
    helper = self.susp._native_blocks()
    if helper is not None and helper is self.susp.ctx and r.is_contiguous() and r.dtype == torch.float64:
      # two launches: the pair repulsion, then weight + wall repulsion added to its z entries (rmb_one_blob_force_device)
      f = None
      if self.repulsion_strength != 0.0:
        helper.set_positions(r.view(-1), self.a, self.periodic_length, False)   # true heights, no clamp
        f = helper.blob_blob_force_device(self.repulsion_strength, self.debye_length, self.a)
        helper.set_positions(self.susp.r_dev, self.a, self.periodic_length, self.susp.wall)  # back to the mobility view
      return helper.one_blob_force_device(r, self.a, self.g * self.blob_mass, self.repulsion_strength_wall, self.debye_length_wall,
                                          out=f).view(-1, 3)
    f = torch.zeros_like(r)
    f[:, 2] = -self.g * self.blob_mass
    if self.repulsion_strength_wall != 0.0:
      h = r[:, 2]
      e = self.repulsion_strength_wall / self.debye_length_wall
      f[:, 2] += torch.where(h > self.a, e * torch.exp(-(h - self.a) / self.debye_length_wall), torch.full_like(h, e))
    if self.repulsion_strength != 0.0:
      ctx = self.susp.ctx
      ctx.set_positions(r.contiguous().view(-1), self.a, self.periodic_length, False)   # true heights, no clamp
      f = f + ctx.blob_blob_force_device(self.repulsion_strength, self.debye_length, self.a).view(-1, 3)
      ctx.set_positions(self.susp.r_dev, self.a, self.periodic_length, self.susp.wall)  # back to the mobility view
    return f

  def force_torque_calculator(self):
    """(nb, 6) force and torque on every body from the blob forces: F = sum f, T = sum rel x f = K^T f
    (multi_bodies_functions.py:411-445), plus the optional external hook."""
    f = self.calc_blob_forces(self.susp.r_dev.view(-1, 3))
    FT = self.susp.KT_times_lambda(f.reshape(-1)).view(-1, 6)
    if self.external_force_torque is not None:
      FT = FT + self.external_force_torque(self)
    return FT

  def _slip(self):
    if self.calc_slip is not None:
      return self.calc_slip(self).reshape(-1)
    if self.slip_body_frame is not None:
      return lab_frame_slip(self.susp, self.slip_body_frame)
    return torch.zeros(3 * self.Nblobs, dtype=torch.float64, device=self.device)

  # ---- the rigid solve ------------------------------------------------------------------------------
  def _assemble_rhs(self, RHS=None, noise=None, noise_FT=None):
    n3 = 3 * self.Nblobs
    if RHS is None:
      FT = self.force_torque_calculator()
      if noise_FT is not None:
        FT = FT + noise_FT.view(-1, 6)
      RHS = self.susp.prescribe(torch.cat([self._slip(), -FT.reshape(-1)]))
    else:
      RHS = RHS.clone()
    if noise is not None:
      RHS[:n3] -= noise
    return RHS

  def solve_mobility_problem(self, RHS=None, noise=None, noise_FT=None, guess=False, tolerance=None):
    """[M -K; -K^T 0][lambda; U] = [slip - noise; -(F + noise_FT)] at the bound configuration
    (quaternion_integrator_multi_bodies.py:1441-1547).  Returns the full solution tensor.
    guess=True marks the calls the reference makes with `x0 = self.first_guess, save_first_guess = True`.  There the
    guess is silently dropped (general_application_utils.gmres passes x0=None on the right-preconditioned path, :627), so
    by default it is not used here either and the iteration counts equal the reference's; with `warm_start = True` the
    previous (normalised) solution does seed GMRES, which pays in slowly varying deterministic runs."""
    RHS = self._assemble_rhs(RHS, noise, noise_FT)
    x0 = None
    if guess and self.warm_start and self.first_guess is not None:
      x0 = self.first_guess * float(torch.linalg.norm(RHS))
    sol, info = self.susp.solve(RHS, tol=self.tolerance if tolerance is None else tolerance, restart=60, maxiter=1000,
                                x0=x0)
    self.det_iterations_count += info["iterations"]
    if guess and info.get("rhs_norm", 0.0) > 0:
      self.first_guess = sol / info["rhs_norm"]
    return self.susp.impose_prescribed_velocity(sol)

  def solve_mobility_problem_pair(self, first, second):
    """Two rigid solves at the same configuration advanced in lockstep (RigidSuspension.solve_pair); `first` / `second`
    are the keyword arguments one would give solve_mobility_problem (RHS, noise, noise_FT, guess)."""
    sols = []
    rhs = [self._assemble_rhs(kw.get("RHS"), kw.get("noise"), kw.get("noise_FT")) for kw in (first, second)]
    for kw, (sol, info) in zip((first, second), self.susp.solve_pair(rhs[0], rhs[1], tol=self.tolerance, restart=60,
                                                                    maxiter=1000)):
      self.det_iterations_count += info["iterations"]
      if kw.get("guess") and info.get("rhs_norm", 0.0) > 0:
        self.first_guess = sol / info["rhs_norm"]
      sols.append(self.susp.impose_prescribed_velocity(sol))
    return sols

  def _velocities(self, sol):
    return sol[3 * self.Nblobs:]

  def _noise(self, z, factor):
    noise, its = self.susp.stochastic_forcing(z, factor, tol=self.tolerance, print_residual=self.print_residual)
    self.stoch_iterations_count += its
    return noise

  def advance_time_step(self, dt, *args, **kwargs):
    if self._precision == 'single' and self.kT > 0.0 and self.scheme.startswith("stochastic"):
      from .rollers import _check_rfd_delta_for_single_precision
      _check_rfd_delta_for_single_precision(self.rf_delta)     # rf_delta may have been set after `precision`
    return getattr(self, self.scheme)(dt, *args, **kwargs)

  def _accept(self, location, orientation):
    self.consecutive_rejections = 0
    self.location, self.orientation = location, orientation
    self._move(location, orientation)

  # ---- deterministic schemes ------------------------------------------------------------------------
  def deterministic_forward_euler(self, dt, *args, **kwargs):
    while True:
      self.preprocess(self)
      self._move(self.location, self.orientation)
      self._refresh_preconditioner(kwargs.get("step"))
      U = self._velocities(self.solve_mobility_problem(guess=True))
      new = self._advance(self.location, self.orientation, U, dt)
      self.postprocess(self)
      if self._valid(*new):
        return self._accept(*new)

  def deterministic_adams_bashforth(self, dt, *args, **kwargs):
    while True:
      self.preprocess(self)
      self._move(self.location, self.orientation)
      self._refresh_preconditioner(kwargs.get("step"))
      U = self._velocities(self.solve_mobility_problem(guess=True))
      if self.first_step is False:
        new = self._advance(self.location, self.orientation, 1.5 * U - 0.5 * self.velocities_previous_step, dt)
      else:
        new = self._advance(self.location, self.orientation, U, dt)
      self.postprocess(self)
      if self._valid(*new):
        self.first_step = False
        self.velocities_previous_step = U
        return self._accept(*new)

  def deterministic_midpoint(self, dt, *args, **kwargs):
    while True:
      self.preprocess(self)
      old = (self.location, self.orientation)
      self._move(*old)
      self._refresh_preconditioner(kwargs.get("step"))
      U = self._velocities(self.solve_mobility_problem(guess=True))
      mid = self._advance(old[0], old[1], U, 0.5 * dt)
      if not self._valid(*mid):
        continue
      self._move(*mid)
      U_mid = self._velocities(self.solve_mobility_problem(guess=True))
      new = self._advance(old[0], old[1], U_mid, dt)
      self.postprocess(self)
      if self._valid(*new):
        return self._accept(*new)
      self._move(*old)

  # ---- stochastic schemes ---------------------------------------------------------------------------
  def _rfd_drift_velocity(self, old, rfd_noise):
    """Thermal drift of the first-order RFD schemes (:371-407): solve with the bodies displaced by -delta/2 W, then
    correct that solution at +delta/2 W with one more solve on the residual.  Returns the drift velocities
    (to be scaled by kT / delta)."""
    n3 = 3 * self.Nblobs
    W = rfd_noise.view(-1, 6)
    Lb = self.body_length.unsqueeze(1)
    force_rfd = W.clone()
    force_rfd[:, 0:3] /= Lb
    rhs = torch.cat([torch.zeros(n3, dtype=torch.float64, device=self.device), -force_rfd.reshape(-1)])
    half = self.rf_delta * 0.5
    minus = (old[0] + W[:, 0:3] * (-half * Lb),
             quaternion_multiply_torch(quaternion_from_rotation_torch(W[:, 3:6] * (-half)), old[1]))
    self._move(*minus)
    sol = self.solve_mobility_problem(RHS=rhs)
    plus = (old[0] + W[:, 0:3] * (half * Lb),
            quaternion_multiply_torch(quaternion_from_rotation_torch(W[:, 3:6] * half), old[1]))
    self._move(*plus)
    sol = self.solve_mobility_problem(RHS=rhs - self.susp.apply_operator(sol))
    return self._velocities(sol)

  def stochastic_first_order_RFD(self, dt, *args, **kwargs):
    while True:
      self.preprocess(self)
      old = (self.location, self.orientation)
      rfd_noise = self._normal(6 * self.Nbodies)
      self._move(*old)
      self._refresh_preconditioner(kwargs.get("step"))
      noise = self._noise(self._normal(3 * self.Nblobs), math.sqrt(2 * self.kT / dt))
      U = self._velocities(self.solve_mobility_problem(noise=noise, guess=True)).clone()
      U = U + (self.kT / self.rf_delta) * self._rfd_drift_velocity(old, rfd_noise)
      new = self._advance(old[0], old[1], U, dt)
      self.postprocess(self)
      if self._valid(*new):
        return self._accept(*new)
      self._move(*old)

  def stochastic_adams_bashforth(self, dt, *args, **kwargs):
    while True:
      self.preprocess(self)
      old = (self.location, self.orientation)
      rfd_noise = self._normal(6 * self.Nbodies)
      self._move(*old)
      self._refresh_preconditioner(kwargs.get("step"))
      noise = self._noise(self._normal(3 * self.Nblobs), math.sqrt(2 * self.kT / dt))
      zero = torch.zeros(self.susp.size, dtype=torch.float64, device=self.device)
      U_stoch = self._velocities(self.solve_mobility_problem(RHS=zero, noise=noise)).clone()
      U_det = self._velocities(self.solve_mobility_problem(guess=True)).clone()
      U_stoch = U_stoch + (self.kT / self.rf_delta) * self._rfd_drift_velocity(old, rfd_noise)
      if self.first_step is False:
        new = self._advance(old[0], old[1], 1.5 * U_det - 0.5 * self.velocities_previous_step + U_stoch, dt)
      else:
        new = self._advance(old[0], old[1], U_det + U_stoch, dt)
      self.postprocess(self)
      if self._valid(*new):
        self.first_step = False
        self.velocities_previous_step = U_det
        return self._accept(*new)
      self._move(*old)

  def stochastic_EM(self, dt, *args, **kwargs):
    """Euler-Maruyama without drift term (:262-323): Brownian slip + one rigid solve."""
    while True:
      self.preprocess(self)
      old = (self.location, self.orientation)
      self._move(*old)
      self._refresh_preconditioner(kwargs.get("step"))
      noise = self._noise(self._normal(3 * self.Nblobs), math.sqrt(2 * self.kT / dt))
      U = self._velocities(self.solve_mobility_problem(noise=noise, guess=True))
      new = self._advance(old[0], old[1], U, dt)
      self.postprocess(self)
      if self._valid(*new):
        return self._accept(*new)

  def _traction_scheme(self, dt, adams_bashforth, step):
    """stochastic_traction_EM (:626-735) and stochastic_traction_AB (:803-922): the thermal drift comes from a random
    finite difference of M, K and K^T applied to the constraint forces / velocities of a rigid solve with random
    force-torque W (2-3 rigid solves + 1 Lanczos + 2 blob products + 4 K products per step)."""
    n3 = 3 * self.Nblobs
    Lb = self.body_length.unsqueeze(1)
    while True:
      self.preprocess(self)
      old = (self.location, self.orientation)
      rfd_noise = self._normal(6 * self.Nbodies).view(-1, 6)
      W = rfd_noise.clone()
      W[:, 0:3] *= self.kT / Lb
      W[:, 3:6] *= self.kT
      self._move(*old)
      self._refresh_preconditioner(step)
      rhs = torch.cat([torch.zeros(n3, dtype=torch.float64, device=self.device), -W.reshape(-1)])
      sol = self.solve_mobility_problem(RHS=rhs)
      U_RFD, Lam = sol[n3:].contiguous(), sol[:n3].contiguous()
      MxLam = self.susp.mobility_times_lambda(Lam)
      KTxLam = self.susp.KT_times_lambda(Lam)
      KxU = self.susp.K_times_U(U_RFD)
      self._move(old[0] + rfd_noise[:, 0:3] * (self.rf_delta * Lb),
                 quaternion_multiply_torch(quaternion_from_rotation_torch(rfd_noise[:, 3:6] * self.rf_delta), old[1]))
      DxM = self.susp.mobility_times_lambda(Lam) - MxLam
      DxKT = self.susp.KT_times_lambda(Lam) - KTxLam
      DxK = self.susp.K_times_U(U_RFD) - KxU
      self._move(*old)
      slip_noise = self._noise(self._normal(n3), math.sqrt(2.0 * self.kT / dt))
      rand_force = (-1.0 / self.rf_delta) * DxKT
      if not adams_bashforth:
        rand_slip = slip_noise + (1.0 / self.rf_delta) * (DxM - DxK)
        U = self._velocities(self.solve_mobility_problem(noise=rand_slip, noise_FT=rand_force, guess=True))
      else:
        rand_slip = (1.0 / self.rf_delta) * (DxM - DxK)
        U_new = self._velocities(self.solve_mobility_problem(noise=rand_slip, noise_FT=rand_force, guess=True)).clone()
        rhs = torch.cat([-slip_noise, torch.zeros(6 * self.Nbodies, dtype=torch.float64, device=self.device)])
        U_noise = self._velocities(self.solve_mobility_problem(RHS=rhs))
        if self.first_step is False:
          U = 1.5 * U_new + U_noise - 0.5 * self.velocities_previous_step
        else:
          U = U_new + U_noise
      new = self._advance(old[0], old[1], U, dt)
      self.postprocess(self)
      if self._valid(*new):
        if adams_bashforth:
          self.first_step = False
          self.velocities_previous_step = U_new
        return self._accept(*new)

  def stochastic_traction_EM(self, dt, *args, **kwargs):
    return self._traction_scheme(dt, False, kwargs.get("step"))

  def stochastic_traction_AB(self, dt, *args, **kwargs):
    return self._traction_scheme(dt, True, kwargs.get("step"))

  def _identity_unconstrained_velocity(self, slip):
    """U = (K^T K)^+ K^T (-slip) per body: the unconstrained mobility problem with M = I and no force
    (multi_bodies.py:693-711, `block_diagonal_preconditioner_identity`), at the bound configuration."""
    out = torch.empty((self.Nbodies, 6), dtype=torch.float64, device=self.device)
    for g in self.susp.groups:
      KtK = torch.bmm(g.K.transpose(1, 2), g.K)
      rhs = -torch.bmm(g.K.transpose(1, 2), self.susp._blobs_of(slip, g).unsqueeze(-1))
      self.susp._put_bodies(out, g, torch.bmm(torch.linalg.pinv(KtK), rhs))
    return out.reshape(-1)

  def stochastic_GDC_RFD(self, dt, *args, **kwargs):
    """Generalised drift-corrector scheme (:1048-1211): Brownian velocities of the unconstrained M = I problem at q^n
    and at a randomly displaced configuration give div(U) by a random finite difference; the mid-point rigid solve
    is then advanced with the step corrected by (1 + dt/2 div U).  1 rigid solve + 3 Lanczos per step."""
    n3 = 3 * self.Nblobs
    step = kwargs.get("step")
    factor = math.sqrt(2 * self.kT / dt)
    Lb = self.body_length.unsqueeze(1)
    while True:
      self.preprocess(self)
      old = (self.location, self.orientation)
      W = self._normal(n3)
      self._move(*old)
      self._refresh_preconditioner(step)
      U_n = self._identity_unconstrained_velocity(-self._noise(W, factor))
      WRFD = self._normal(6 * self.Nbodies).view(-1, 6)
      self._move(old[0] + self.rf_delta * Lb * WRFD[:, 0:3],
                 quaternion_multiply_torch(quaternion_from_rotation_torch(self.rf_delta * WRFD[:, 3:6]), old[1]))
      self._refresh_preconditioner(step)
      U_rfd = self._identity_unconstrained_velocity(-self._noise(W, factor))
      dU = (U_rfd - U_n).view(-1, 6)
      div = float(((dU[:, 0:3] * WRFD[:, 0:3]).sum(dim=1) / (self.rf_delta * self.body_length)).sum() +
                  (dU[:, 3:6] * WRFD[:, 3:6]).sum() / self.rf_delta)
      mid = self._advance(old[0], old[1], U_n, 0.5 * dt)
      if not self._valid(*mid):
        self._move(*old)
        continue
      self._move(*mid)
      self._refresh_preconditioner(step)
      U_mid = self._velocities(self.solve_mobility_problem(noise=self._noise(W, factor), guess=True))
      new = self._advance(old[0], old[1], U_mid, dt * (1.0 + 0.5 * dt * div))
      self.postprocess(self)
      if self._valid(*new):
        return self._accept(*new)
      self._move(*old)

  def _slip_scheme(self, dt, trapezoidal, step):
    """Shared body of stochastic_Slip_Trapz (:925-1045) and stochastic_Slip_Mid (:1214-1343): predictor with the
    Brownian slip, random finite difference on M and K^T along the displacement a rigid solve of W_slip produces,
    corrector with the drift folded into slip and force."""
    n3 = 3 * self.Nblobs
    while True:
      self.preprocess(self)
      old = (self.location, self.orientation)
      W1 = self._normal(n3)
      W_slip = self._normal(n3)
      Wcor = None if trapezoidal else W1 + self._normal(n3)
      self._move(*old)
      KTxW = self.susp.KT_times_lambda(W_slip)
      self._refresh_preconditioner(step)
      rhs = torch.cat([-W_slip, torch.zeros(6 * self.Nbodies, dtype=torch.float64, device=self.device)])
      f1 = math.sqrt(2 * self.kT / dt) if trapezoidal else math.sqrt(4 * self.kT / dt)
      rfd_tol = self.tolerance if self.rfd_solve_tolerance is None else self.rfd_solve_tolerance
      if self.lockstep_solves if self.lockstep_solves is not None else self.Nblobs >= self.lockstep_min_blobs:
        # Everything at time level n that needs the mobility M(q^n) and does not depend on another result: the product
        # M W_slip, the Brownian forcing(s) (Lanczos) and the RFD solve advance together, one k-vector pass over the
        # pairs per round.  The Brownian-slip solve needs the forcing and follows; the corrector solve and the second
        # RFD product use other configurations (other mobilities) and cannot join.
        tasks = [self.susp.product_task(W_slip), self.susp.solve_task(rhs, tol=rfd_tol)]
        if self.kT > 0.0:
          tasks.append(self.susp.forcing_task(W1, f1, tol=self.tolerance, print_residual=self.print_residual))
          if not trapezoidal:
            tasks.append(self.susp.forcing_task(Wcor, math.sqrt(self.kT / dt), tol=self.tolerance,
                                                print_residual=self.print_residual))
        res = self.susp.run_lockstep(tasks)
        MxW, (sol_rfd, info_rfd) = res[0], res[1]
        self.det_iterations_count += info_rfd["iterations"]
        W_RFD = self._velocities(self.susp.impose_prescribed_velocity(sol_rfd))
        zero = torch.zeros(n3, dtype=torch.float64, device=self.device)
        noise_W1 = res[2][0] if self.kT > 0.0 else zero
        noise_Wcor = (res[3][0] if self.kT > 0.0 else zero) if not trapezoidal else None
        self.stoch_iterations_count += sum(r[1] for r in res[2:])
        U_1 = self._velocities(self.solve_mobility_problem(noise=noise_W1, guess=True)).clone()
      else:
        MxW = self.susp.mobility_times_lambda(W_slip)
        noise_W1 = self._noise(W1, f1)
        if not trapezoidal:
          noise_Wcor = self._noise(Wcor, math.sqrt(self.kT / dt))
        U_1 = self._velocities(self.solve_mobility_problem(noise=noise_W1, guess=True)).clone()
        W_RFD = self._velocities(self.solve_mobility_problem(RHS=rhs, tolerance=self.rfd_solve_tolerance))
      self._move(*self._advance(old[0], old[1], W_RFD, self.rf_delta))
      M_rfdxW = self.susp.mobility_times_lambda(W_slip)
      KT_rfdxW = self.susp.KT_times_lambda(W_slip)
      if trapezoidal:
        rand_slip_cor = noise_W1 + (2.0 * self.kT / self.rf_delta) * (M_rfdxW - MxW)
        rand_force_cor = -2.0 * (self.kT / self.rf_delta) * (KT_rfdxW - KTxW)
        predictor = self._advance(old[0], old[1], U_1, dt)
      else:
        rand_slip_cor = noise_Wcor + (self.kT / self.rf_delta) * (M_rfdxW - MxW)
        rand_force_cor = -1.0 * (self.kT / self.rf_delta) * (KT_rfdxW - KTxW)
        predictor = self._advance(old[0], old[1], U_1, 0.5 * dt)
      if not self._valid(*predictor):
        self._move(*old)
        continue
      self._move(*predictor)
      U_2 = self._velocities(self.solve_mobility_problem(noise=rand_slip_cor, noise_FT=rand_force_cor, guess=True))
      U_new = 0.5 * (U_1 + U_2) if trapezoidal else U_2
      new = self._advance(old[0], old[1], U_new, dt)
      self.postprocess(self)
      if self._valid(*new):
        return self._accept(*new)
      self._move(*old)

  def stochastic_Slip_Trapz(self, dt, *args, **kwargs):
    """3 rigid solves + 1 Lanczos + 2 blob mobility products + 2 K^T products per step."""
    return self._slip_scheme(dt, True, kwargs.get("step"))

  def stochastic_Slip_Mid(self, dt, *args, **kwargs):
    """3 rigid solves + 2 Lanczos + 2 blob mobility products + 2 K^T products per step."""
    return self._slip_scheme(dt, False, kwargs.get("step"))


# ---- driver: reference input deck -> integrator -> time loop ------------------------------------------
  # ---- dense-algebra schemes --------------------------------------------------------------------------------------
  # quaternion_integrator_multi_bodies.py:110-139 (deterministic_forward_euler_dense_algebra), :552-623
  # (stochastic_first_order_RFD_dense_algebra), :738-800 (Fixman), :1346-1438 (stochastic_Slip_Mid_DLA), on the dense
  # solves :1550-1635.  O(N^3) by definition, meant for a few bodies: the blob mobility is built densely on the device
  # (body_dense_tt_kernel, the whole suspension as one "body") and factorised there with torch.linalg.  The reference takes
  # N^{1/2} W from numpy's eigendecomposition as V S^{1/2} W (stochastic_forcing_eig, stochastic_forcing.py:7-41) -- NOT
  # the symmetric square root, so the result depends on LAPACK's eigenvector signs; the 6 n_bodies x 6 n_bodies matrix
  # goes to the host and through the same numpy routine so that a run with the reference's seed follows its trajectory.
  # The draws come in the reference's order.
  def _pinv(self, A):
    return torch.linalg.pinv(A, rtol=1e-14, hermitian=False)

  def _dense_solve(self):
    """(U, N, M, R, K) of solve_mobility_problem_DLA (:1592-1635): N = pinv(K^T M^-1 K), U = N (F + K^T M^-1 slip)...
    with the sign convention of the callers below: dense_algebra (:1550-1589) SUBTRACTS the slip term, DLA adds it."""
    rs = self.susp
    M = rs.dense_blob_mobility()
    R = torch.linalg.inv(M)
    K = rs.dense_K()
    N = self._pinv(K.t() @ R @ K)
    FT = self.force_torque_calculator().reshape(-1)
    force_slip = K.t() @ (R @ self._slip())
    return FT, force_slip, N, M, R, K

  def solve_mobility_problem_dense_algebra(self):
    """(velocities, body mobility) as :1550-1589: U = N (F - K^T M^-1 slip)."""
    FT, force_slip, N, M, R, K = self._dense_solve()
    return N @ (FT - force_slip), N

  def solve_mobility_problem_DLA(self):
    """(velocities, N, M, M^-1, K) as :1592-1635: U = N (F + K^T M^-1 slip)."""
    FT, force_slip, N, M, R, K = self._dense_solve()
    return N @ (FT + force_slip), N, M, R, K

  def _eig_forcing(self, mobility, factor, z=None):
    """factor V S^{1/2} z with numpy.linalg.eigh on the host, non-positive eigenvalues dropped (stochastic_forcing.py:7-41);
    z = None draws it HERE, after whatever the caller drew before (the reference's order)."""
    vals, vecs = np.linalg.eigh(mobility.cpu().numpy())
    root = np.sqrt(np.where(vals > 0, vals, 0.0))
    zz = self._normal(len(vals)) if z is None else z
    return torch.as_tensor(factor * (vecs @ (root * zz.cpu().numpy())), device=self.device)

  def _eig_symm_forcing(self, mobility, z):
    """V S^{1/2} V^T z (stochastic_forcing_eig_symm, stochastic_forcing.py:44-81): independent of the eigenvector basis, so
    it stays on the device."""
    vals, vecs = torch.linalg.eigh(mobility)
    return vecs @ (torch.sqrt(torch.clamp(vals, min=0.0)) * (vecs.t() @ z))

  def deterministic_forward_euler_dense_algebra(self, dt, *args, **kwargs):
    while True:
      self.preprocess(self)
      self._move(self.location, self.orientation)
      U, _ = self.solve_mobility_problem_dense_algebra()
      new = self._advance(self.location, self.orientation, U, dt)
      self.postprocess(self)
      if self._valid(*new):
        return self._accept(*new)

  def stochastic_first_order_RFD_dense_algebra(self, dt, *args, **kwargs):
    while True:
      self.preprocess(self)
      old = (self.location, self.orientation)
      self._move(*old)
      U, N = self.solve_mobility_problem_dense_algebra()
      rfd_noise = self._normal(6 * self.Nbodies)
      U = U + self._eig_forcing(N, math.sqrt(2 * self.kT / dt))
      W = rfd_noise.view(-1, 6)
      Lb = self.body_length.unsqueeze(1)
      force_rfd = W.clone()
      force_rfd[:, 0:3] /= Lb
      moved = (old[0] + W[:, 0:3] * (self.rf_delta * Lb),
               quaternion_multiply_torch(quaternion_from_rotation_torch(W[:, 3:6] * self.rf_delta), old[1]))
      self._move(*moved)
      rs = self.susp
      K = rs.dense_K()
      N_new = self._pinv(K.t() @ torch.linalg.inv(rs.dense_blob_mobility()) @ K)
      U = U + (self.kT / self.rf_delta) * ((N_new - N) @ force_rfd.reshape(-1))
      new = self._advance(old[0], old[1], U, dt)
      self.postprocess(self)
      if self._valid(*new):
        return self._accept(*new)
      self._move(*old)

  def Fixman(self, dt, *args, **kwargs):
    while True:
      self.preprocess(self)
      old = (self.location, self.orientation)
      self._move(*old)
      U_mid, N = self.solve_mobility_problem_dense_algebra()
      W1 = self._normal(6 * self.Nbodies)
      W_cor = W1 + self._normal(6 * self.Nbodies)
      Nhalf_W1 = self._eig_forcing(N, math.sqrt(4 * self.kT / dt), z=W1)
      Nhalf_Wcor = self._eig_forcing(N, math.sqrt(self.kT / dt), z=W_cor)
      Ninvhalf_cor = self._pinv(N) @ Nhalf_Wcor
      mid = self._advance(old[0], old[1], U_mid + Nhalf_W1, 0.5 * dt)
      if not self._valid(*mid):
        continue
      self._move(*mid)
      U_new, N_mid = self.solve_mobility_problem_dense_algebra()
      U_new = U_new + N_mid @ Ninvhalf_cor
      new = self._advance(old[0], old[1], U_new, dt)
      self.postprocess(self)
      if self._valid(*new):
        return self._accept(*new)
      self._move(*old)

  def stochastic_Slip_Mid_DLA(self, dt, *args, **kwargs):
    n3 = 3 * self.Nblobs
    while True:
      self.preprocess(self)
      old = (self.location, self.orientation)
      self._move(*old)
      U_mid, N_mid, M_mid, R_mid, K_mid = self.solve_mobility_problem_DLA()
      W1 = self._normal(n3)
      W_slip = self._normal(n3)
      Wcor = W1 + self._normal(n3)
      W_RFD = (N_mid @ (K_mid.t() @ (R_mid @ W_slip))).view(-1, 6)
      MxW_slip = M_mid @ W_slip
      KTxW_slip = K_mid.t() @ W_slip
      Mhalf_W1 = self._eig_symm_forcing(M_mid, W1)
      Mhalf_Wcor = self._eig_symm_forcing(M_mid, Wcor)
      U_mid = U_mid + math.sqrt(4 * self.kT / dt) * (N_mid @ (K_mid.t() @ (R_mid @ Mhalf_W1)))
      # random finite difference: bodies displaced by rf_delta W_RFD (no body-length scaling in this scheme, :1390-1393)
      rfd = (old[0] + W_RFD[:, 0:3] * self.rf_delta,
             quaternion_multiply_torch(quaternion_from_rotation_torch(W_RFD[:, 3:6] * self.rf_delta), old[1]))
      self._move(*rfd)
      DxM = self.susp.dense_blob_mobility() @ W_slip - MxW_slip
      DxKT = self.susp.dense_K().t() @ W_slip - KTxW_slip
      mid = self._advance(old[0], old[1], U_mid, 0.5 * dt)
      if not self._valid(*mid):
        continue
      self._move(*mid)
      U_new, N_new, M_new, R_new, K_new = self.solve_mobility_problem_DLA()
      RHS_cor = -(self.kT / self.rf_delta) * DxKT + K_new.t() @ (R_new @ (math.sqrt(self.kT / dt) * Mhalf_Wcor +
                                                                        (self.kT / self.rf_delta) * DxM))
      U_new = U_new + N_new @ RHS_cor
      new = self._advance(old[0], old[1], U_new, dt)
      self.postprocess(self)
      if self._valid(*new):
        return self._accept(*new)
      self._move(*old)


def bodies_from_input(read):
  """Bodies of a deck as multi_bodies/multi_bodies.py:1160-1212 creates them: every `structure` line = vertex file +
  clones file (+ optional .slip file with one body-frame slip per blob).  Returns a dict with one reference
  configuration and one body-frame slip array per body, stacked locations / quaternions, and bodies per structure."""
  from . import deck_modes
  from . import structures as st
  refs, locs, quats, slips, body_types = [], [], [], [], []
  any_slip = False
  if read.articulated:
    raise ValueError("articulated bodies are not supported")
  prescribed = []
  for sid, structure in enumerate(read.structures):
    ref = deck_modes.uniform_vertices(st.read_vertex_file(read.resolve(structure[0])), read.blob_radius, structure[0])
    n, loc, quat = st.read_clones_file(read.resolve(structure[1]))
    slip = None
    for extra in structure[2:]:
      if extra.endswith(".slip"):
        slip = st.read_slip_file(read.resolve(extra))[:len(ref)]
        any_slip = True
    for k in range(n):
      refs.append(ref)
      slips.append(slip if slip is not None else np.zeros((len(ref), 3)))
      prescribed.append(sid >= read.num_free_bodies)      # `obstacle` lines follow the `structure` lines
    locs.append(loc)
    quats.append(quat)
    body_types.append(n)
  if not refs:
    raise ValueError("input deck lists no structure")
  return dict(refs=refs, locations=np.concatenate(locs), quaternions=np.concatenate(quats),
              slips=np.concatenate(slips) if any_slip else None, body_types=body_types,
              structures_ID=list(read.structures_ID), prescribed=np.array(prescribed, dtype=bool))


def integrator_from_input(read, device="cuda:0", ctx=None, rng=None):
  """Integrator wired from a ReadInput deck as multi_bodies/multi_bodies.py:1319-1393 wires QuaternionIntegrator."""
  from . import deck_modes
  deck_modes.validate(read, uses_dense_blocks=True)     # ValueError for modes this engine does not run
  b = bodies_from_input(read)
  refs, body_types, any_slip = b["refs"], b["body_types"], b["slips"] is not None
  if rng is None:
    rng = read.random_generator(save=False)
  rng = replicate_rng(rng, read, ctx, device)
  integ = RigidIntegrator(refs, b["locations"], b["quaternions"], read.scheme, read.blob_radius, read.eta,
                          tolerance=read.solver_tolerance, domain=read.domain, periodic_length=read.periodic_length,
                          device=device, ctx=ctx, rng=rng, prescribed=b["prescribed"])
  integ.kT = read.kT
  integ.rf_delta = read.rf_delta
  integ.update_PC = read.update_PC
  integ.g = read.g
  integ.repulsion_strength_wall = read.repulsion_strength_wall
  integ.debye_length_wall = read.debye_length_wall
  if read.blob_blob_force_implementation != "None":
    integ.repulsion_strength = read.repulsion_strength
    integ.debye_length = read.debye_length
  if any_slip:
    integ.slip_body_frame = torch.as_tensor(b["slips"], device=integ.device)
  integ.body_types = body_types
  integ.structures_ID = b["structures_ID"]
  return integ


def _write_clones(fh, locations, quaternions):
  fh.write(str(len(locations)) + "\n")
  for x, q in zip(locations, quaternions):
    fh.write("%s %s %s %s %s %s %s\n" % (x[0], x[1], x[2], q[0], q[1], q[2], q[3]))


def run(read, integrator, output_name=None, n_steps=None, callback=None):
  """Time loop of multi_bodies.py:1412-1530: `.clones` per saved step or one `.config` per structure, reference text
  format (location + quaternion per body)."""
  output_name = read.output_name if output_name is None else output_name
  n_steps = read.n_steps if n_steps is None else n_steps
  if read.save_clones not in ("one_file_per_step", "one_file"):
    raise ValueError('save_clones = %s is not implemented; use "one_file_per_step" or "one_file"' % read.save_clones)
  files = None
  if read.save_clones == "one_file":
    files = [open(output_name + "." + ID + ".config", "w") for ID in integrator.structures_ID]

  def save(step):
    loc, quat = integrator.location.cpu().numpy(), integrator.orientation.cpu().numpy()
    offset = 0
    for i, ID in enumerate(integrator.structures_ID):
      sl = slice(offset, offset + integrator.body_types[i])
      offset += integrator.body_types[i]
      if files is not None:
        _write_clones(files[i], loc[sl], quat[sl])
      else:
        with open(output_name + "." + ID + "." + str(step).zfill(8) + ".clones", "w") as fh:
          _write_clones(fh, loc[sl], quat[sl])

  try:
    step = read.initial_step - 1
    for step in range(read.initial_step, n_steps):
      if step % read.n_save == 0 and step >= 0:
        save(step)
      integrator.advance_time_step(read.dt, step=step)
      if callback is not None:
        callback(step, integrator)
    if (step + 1) % read.n_save == 0 and step >= 0:
      save(step + 1)
  finally:
    if files is not None:
      for fh in files:
        fh.close()
  return integrator
