"""Device-resident time steppers for single-blob rollers (SURVEY.md section 8(f), row N4 tail; the
config-5 recipe of section 3.3).

Mirror of quaternion_integrator/quaternion_integrator_rollers.py (`QuaternionIntegratorRollers`): same
scheme names, same attribute names (eta, a, kT, tolerance, rf_delta, omega_one_roller, free_kinematics,
hydro_interactions, periodic_length, domain, counters), same order of random draws, so a run of the
reference with `np.random.seed(s)` and a run of this class with `rng=np.random.RandomState(s)` walk the
same trajectory (to solver tolerance).  What differs is WHERE things live:

  * the reference keeps one Python `Body` per roller and loops over them for every update (:132-151);
    here the locations are ONE (N,3) float64 tensor in HBM and every update is one fused tensor op;
  * forces, the four mobility products, the Lanczos basis and the GMRES basis stay on the device; per
    step only a handful of scalars (norms, the validity flag) cross PCIe;
  * `M_tt F + M_tr T` is one fused pair sweep (kind tt_tr, K11), which the reference has but leaves
    commented out (:71, :1116);
  * the random finite difference and all Krylov products reuse the same MobilityContext, re-packing the
    positions (an O(N) kernel) when they move.

Schemes: deterministic_forward_euler (:119), stochastic_first_order (:154),
deterministic_adams_bashforth (:199), stochastic_adams_bashforth (:251), stochastic_EM (:304),
stochastic_GDC (:369), stochastic_mid_point (:495), stochastic_mid_point_version_2 (:577),
stochastic_trapezoidal (:659).  The articulated schemes (:737-902) need the constraint solver and are
out of scope (constraints are not on the path).
"""
import math
import os

import numpy as np
import torch

from .context import MobilityContext
from .rigid import gmres_right_preconditioned
from .stochastic import stochastic_forcing_lanczos


def _check_rfd_delta_for_single_precision(rf_delta):
  """precision = 'single': a random finite difference (M(q + delta W) - M(q)) W / delta of products that carry ~1e-6
  relative error is noise unless delta is large enough (doc/README.md:512-523: 1e-3 for single, 1e-6 for double)."""
  if not (float(rf_delta) >= 1e-4):
    raise ValueError("precision = 'single' needs rf_delta >= 1e-4 (got %g): the random finite differences divide the "
                     "difference of two ~1e-6-accurate products by rf_delta (doc/README.md:512-523 uses 1e-3 for the "
                     "single-precision build)" % float(rf_delta))


class RollersIntegrator(object):
  """locations: (N,3) array-like or tensor.  `scheme` may carry the reference's `_rollers` suffix."""

  def __init__(self, locations, scheme, a, eta, tolerance=None, domain="single_wall", device="cuda:0", ctx=None,
               rng=None, seed=None):
    self.device = torch.device(device)
    self.location = torch.as_tensor(np.asarray(locations, dtype=np.float64) if not isinstance(locations, torch.Tensor)
                                    else locations, dtype=torch.float64, device=self.device).reshape(-1, 3).clone()
    self.Nblobs = self.location.shape[0]
    self.scheme = scheme
    self.a, self.eta = float(a), float(eta)
    self.domain = domain
    if domain not in ("single_wall", "no_wall", "in_plane"):
      raise ValueError("domain must be single_wall, no_wall or in_plane")
    # state and counters, names as quaternion_integrator_rollers.py:36-58
    self.velocities_previous_step = None
    self.deterministic_torque_previous_step = None
    self.first_step = True
    self.kT = 0.0
    self.tolerance = 1e-08 if tolerance is None else float(tolerance)
    self.rf_delta = 1e-03
    self.invalid_configuration_count = 0
    self.wall_overlaps = 0
    self.omega_one_roller = np.zeros(3)
    self.free_kinematics = "True"
    self.hydro_interactions = 1
    self.det_iterations_count = 0
    self.stoch_iterations_count = 0
    self.periodic_length = np.zeros(3)
    self.print_residual = False
    self.max_retries = 1000              # total rejected steps over the life of the integrator
    self.max_consecutive_retries = 20    # rejections of ONE step in a row: beyond this the step size is wrong, not the draw
    self.consecutive_rejections = 0
    self.report_rejections = True        # one line per rejected step, like the reference's print('Invalid configuration')
    # force parameters: multi_bodies.py:1326-1336 binds these with functools.partial
    self.g = 0.0
    self.blob_mass = 1.0
    self.repulsion_strength_wall = 0.0
    self.debye_length_wall = 1.0
    self.repulsion_strength = 0.0
    self.debye_length = 1.0
    # replaceable hooks, as the reference's attributes of the same names; tensors (N,3) in and out
    self.calc_one_blob_forces = self._one_blob_forces
    self.calc_blob_blob_forces = self._blob_blob_forces
    self.preprocess = lambda integrator: None
    self.postprocess = lambda integrator: None
    # random numbers: a numpy RandomState-like object (host draws, reference order) or a device generator
    self.rng = rng
    self._gen = None
    if rng is None:
      from .rigid_integrator import seeded_generator
      self._gen, self.seed = seeded_generator(self.device, seed, ctx)   # seed=None: fresh OS entropy, as the reference
    self.ctx = ctx if ctx is not None else MobilityContext(self.device.index or 0)
    self._own_ctx = ctx is None
    self.mobility_products = 0
    self._precision = 'double'
    self._force_precision = 'follow'

  @property
  def precision(self):
    """'double' (default) or 'single': the reference GPU module's precision switch (mobility_pycuda.py:7-19) for the
    products of this stepper -- M_tt F + M_tr T, the four blocks and the 6N grand mobility run their fp32 twins
    (csrc/sym32_kernels.h, symx32_kernels.h: fp32 pair arithmetic, fp64 accumulation, ~1e-6 relative) with open
    boundaries, and so do the blob-blob forces (the reference's GPU force kernel is always single precision; pin them
    with `force_precision = 'double'`); pseudo-periodic domains and the stepper's own algebra stay fp64.
    The random finite differences divide a difference of two products by rf_delta, so with ~1e-6-accurate products
    rf_delta must be >= 1e-4 (the reference's advice for its float build, doc/README.md:512-523): the setter (for a stochastic scheme
    with kT > 0) and every stochastic step raise otherwise."""
    return self._precision

  @precision.setter
  def precision(self, value):
    if value not in ('single', 'double'):
      raise ValueError("precision must be 'single' or 'double'")
    if value == 'single' and self.kT > 0.0 and self.scheme.startswith("stochastic"):
      # only the stochastic schemes form random finite differences; a deterministic scheme (or kT = 0) with the
      # reference's double-precision rf_delta = 1e-6 may select single precision (the per-step check covers later changes)
      _check_rfd_delta_for_single_precision(self.rf_delta)
    self._precision = value
    self.ctx.set_option("precision", 32 if value == 'single' else 64)

  @property
  def force_precision(self):
    """'follow' (default: the blob-blob forces use whatever `precision` says, i.e. the reference GPU force kernel's own
    float arithmetic with precision = 'single', forces_pycuda.py:14-21), 'single' or 'double' (pinned) -- context
    option "force_precision".  'double' with precision = 'single' is the products-only single-precision mode."""
    return self._force_precision

  @force_precision.setter
  def force_precision(self, value):
    if value not in ('follow', 'single', 'double'):
      raise ValueError("force_precision must be 'follow', 'single' or 'double'")
    self._force_precision = value
    self.ctx.set_option("force_precision", {'follow': 0, 'single': 32, 'double': 64}[value])

  def close(self):
    if self._own_ctx:
      self.ctx.close()

  # ---- plumbing -------------------------------------------------------------------------------------
  def _randn(self, n):
    if self.rng is not None:
      return torch.as_tensor(self.rng.randn(n), dtype=torch.float64, device=self.device)
    return torch.randn(n, dtype=torch.float64, device=self.device, generator=self._gen)

  def _bind(self, r, wall=None):
    """Make r the configuration the context's products refer to (clamp + B happen on the device)."""
    if wall is None:
      wall = self.domain != "no_wall"
    self.ctx.set_positions(r.contiguous().view(-1), self.a, self.periodic_length, wall)

  def _product(self, kind, vec, vec2=None):
    self.mobility_products += 1
    in_plane = self.domain == "in_plane"
    if in_plane and kind in ("rt", "rr"):
      raise ValueError("domain in_plane has no rot products (quaternion_integrator_rollers.py:85-91)")
    if vec2 is not None:    # M_tt F + M_tr T in one pass over the pairs (in-plane too)
      return self.ctx.matvec_op_device("velocity_from_force_torque", (vec.contiguous(), vec2.contiguous()), self.eta,
                                       in_plane=in_plane)[0]
    return self.ctx.matvec_device(kind, vec.contiguous(), self.eta, in_plane=in_plane)

  def _products_of_one_vector(self, kinds, vec):
    """Several blocks applied to the same vector on the bound configuration.  ("rt", "tt") -- the pair the random
    finite difference of the 6N drift needs (:1138-1160) -- is one pass with the shared pair geometry."""
    if tuple(kinds) == ("rt", "tt") and self.domain != "in_plane":
      self.mobility_products += 1
      u, w = self.ctx.matvec_op_device("force_column", (vec.contiguous(),), self.eta)
      return [w, u]
    return [self._product(k, vec) for k in kinds]

  def mobility_trans_times_force(self, r, force):
    self._bind(r)
    return self._product("tt", force)

  def mobility_trans_times_torque(self, r, torque):
    self._bind(r)
    return self._product("tr", torque)

  def mobility_rot_times_force(self, r, force):
    self._bind(r)
    return self._product("rt", force)

  def mobility_rot_times_torque(self, r, torque):
    self._bind(r)
    return self._product("rr", torque)

  # ---- forces (multi_bodies_functions.py:153-188 and forces_numba.py:12-55) ---------------------------
  def _one_blob_forces(self, r):
    """Gravity + wall repulsion  f_z = -g m + (e_w/b_w) exp(-(h-a)/b_w) for h > a, e_w/b_w below."""
    f = torch.zeros_like(r)
    f[:, 2] = -self.g * self.blob_mass
    if self.repulsion_strength_wall != 0.0:
      h = r[:, 2]
      e = self.repulsion_strength_wall / self.debye_length_wall
      f[:, 2] += torch.where(h > self.a, e * torch.exp(-(h - self.a) / self.debye_length_wall),
                             torch.full_like(h, e))
    return f

  def _blob_blob_forces(self, r):
    if self.repulsion_strength == 0.0:
      return torch.zeros_like(r)
    self._bind(r, wall=False)    # forces act on the true heights, not the clamped ones
    return self.ctx.blob_blob_force_device(self.repulsion_strength, self.debye_length, self.a).view(-1, 3)

  def get_omega_one_roller(self):
    return np.asarray(self.omega_one_roller, dtype=np.float64)

  def get_torque(self):
    """Free kinematics: constant torque 8 pi eta a^3 omega on every roller (:1586-1595)."""
    t = torch.as_tensor(self.get_omega_one_roller() * (8.0 * math.pi * self.eta * self.a ** 3), device=self.device)
    return t.repeat(self.Nblobs)

  # ---- deterministic part ---------------------------------------------------------------------------
  def compute_deterministic_velocity_and_torque(self):
    """v = M_tt F + M_tr T with T prescribed (free kinematics) or solved from M_rr T = omega - M_rt F
    (:905-982).  Returns (velocity (3N,), torque (3N,))."""
    r = self.location
    force = (self.calc_one_blob_forces(r) + self.calc_blob_blob_forces(r)).reshape(-1)
    if self.free_kinematics == "False":
      omega = torch.as_tensor(self.get_omega_one_roller(), device=self.device).repeat(self.Nblobs)
      self._bind(r)
      rhs = omega - self._product("rt", force)
      nrm = float(torch.linalg.norm(rhs))
      if nrm > 0:
        rhs = rhs / nrm
      sol, info = gmres_right_preconditioned(lambda x: self._product("rr", x), lambda x: x, rhs, tol=self.tolerance,
                                             restart=20, maxiter=1000, x0=self.deterministic_torque_previous_step,
                                             sync=getattr(self.ctx, "sync_scalars", None), ortho=self._ortho())
      self.det_iterations_count += info["iterations"]
      self.deterministic_torque_previous_step = sol
      torque = sol * nrm if nrm > 0 else sol
    else:
      torque = self.get_torque()
      self._bind(r)
    if bool(torch.any(torque != 0)):
      velocity = self._product("tt", force, vec2=torque)
    else:
      velocity = self._product("tt", force)
    return velocity, torque

  def _self_mobility_coefficients(self, r):
    """Swan-Brady single-blob coefficients with the max(h/a,1) and damping artefacts (:1027-1047)."""
    h_over_a = r[:, 2] / self.a
    h = torch.clamp(h_over_a, min=1.0)
    damping = torch.where(h_over_a < 0.0, torch.zeros_like(h), torch.where(h_over_a <= 1.0, h_over_a,
                                                                           torch.ones_like(h)))
    f_tt = 1.0 / (6 * math.pi * self.eta * self.a)
    f_rr = 1.0 / (6 * math.pi * self.eta * self.a ** 3)
    f_rt = 1.0 / (6 * math.pi * self.eta * self.a ** 2)
    c = dict(h=h, damping=damping)
    c["mu_rt_para"] = f_rt * (3 / (32 * h ** 4)) * damping
    c["mu_tt_perp"] = f_tt * (1 - 9 / (8 * h) + 1 / (2 * h ** 3) - 1 / (8 * h ** 5)) * damping
    c["mu_tt_para"] = f_tt * (1 - 9 / (16 * h) + 2 / (16 * h ** 3) - 1 / (16 * h ** 5)) * damping
    c["dmu_tt_perp"] = f_tt * (9 / (8 * h ** 2) - 3 / (2 * h ** 4) + 5 / (8 * h ** 6)) * damping
    c["mu_rr_perp"] = f_rr * (3.0 / 4 - 3 / (32 * h ** 3)) * damping
    c["mu_rr_para"] = f_rr * (3.0 / 4 - 15 / (64 * h ** 3)) * damping
    return c

  def compute_deterministic_velocity_and_torque_uncorrelated(self):
    """No hydrodynamic interactions: every roller sees only the wall (:985-1079); O(N) tensor ops."""
    r = self.location
    force = self.calc_one_blob_forces(r) + self.calc_blob_blob_forces(r)
    c = self._self_mobility_coefficients(r)
    if self.free_kinematics == "False":
      omega = torch.as_tensor(self.get_omega_one_roller(), device=self.device).expand(self.Nblobs, 3)
      torque = torch.empty_like(r)
      torque[:, 0] = (omega[:, 0] + c["mu_rt_para"] * force[:, 1]) / c["mu_rr_para"]
      torque[:, 1] = (omega[:, 1] - c["mu_rt_para"] * force[:, 0]) / c["mu_rr_para"]
      torque[:, 2] = omega[:, 2] / c["mu_rr_perp"]
    else:
      torque = self.get_torque().view(-1, 3)
    velocity = torch.empty_like(r)
    velocity[:, 0] = c["mu_tt_para"] * force[:, 0] + c["mu_rt_para"] * torque[:, 1]
    velocity[:, 1] = c["mu_tt_para"] * force[:, 1] - c["mu_rt_para"] * torque[:, 0]
    velocity[:, 2] = c["mu_tt_perp"] * force[:, 2]
    return velocity.reshape(-1), torque.reshape(-1)

  # ---- stochastic part ------------------------------------------------------------------------------
  fused_gram_schmidt = None      # None = automatic (a plain single-GPU context), False = tensor operations

  def _ortho(self):
    """The library's fused Gram-Schmidt step (rmb_krylov_orthogonalize_device: four launches and one 16-byte transfer per
    Lanczos iteration instead of ~15 tensor operations), as RigidSuspension._ortho hands it to the same loop."""
    if self.fused_gram_schmidt is False or type(self.ctx) is not MobilityContext or torch.device(self.device).type != "cuda":
      return None
    return self.ctx.krylov_orthogonalize_device

  native_lanczos = None              # None = automatic, False = never: the whole loop inside the library (rmb_lanczos_device)
  lanczos_native_rows = 128          # basis rows of the library's loop; a forcing that needs more falls back to the generic loop
  lanczos_native_max_blobs = 32768   # above, the iteration the lagged loop discards (a whole pair sweep, 1.4 ms here) costs more than
                                     # the host waits it saves (~70 us per iteration, 40-50 iterations per forcing)
  lanczos_native_loop_calls = 0

  def _lanczos(self, mult, dim, z, dt, product=None):
    """factor M^{1/2} z; product ("tt" / "grand") names the operator `mult` applies when the library has it as ONE call."""
    if (product is not None and self.native_lanczos is not False and os.environ.get("RMB_NATIVE_LANCZOS", "") != "0" and self.kT > 0.0
        and self._ortho() is not None and not self.print_residual and self.Nblobs <= self.lanczos_native_max_blobs
        and (product == "tt" or self.domain != "in_plane")):
      zt = torch.as_tensor(z, dtype=torch.float64, device=self.device).reshape(-1).contiguous()
      noise, its, products = self.ctx.lanczos_device(product, zt, math.sqrt(2 * self.kT / dt), self.tolerance, 1000,
                                                     self.lanczos_native_rows, self.eta, in_plane=self.domain == "in_plane")
      self.mobility_products += products
      self.lanczos_native_loop_calls += 1
      if noise is not None:
        self.stoch_iterations_count += its
        return noise
    noise, its = stochastic_forcing_lanczos(factor=math.sqrt(2 * self.kT / dt), tolerance=self.tolerance, dim=dim,
                                            mobility_mult=mult, z=z, print_residual=self.print_residual,
                                            device=self.device, sync=getattr(self.ctx, "sync_scalars", None), ortho=self._ortho())
    self.stoch_iterations_count += its
    return noise

  def _random_finite_difference(self, kinds):
    """(M(q + d/2 W) - M(q - d/2 W)) W for each product kind in `kinds`, one draw of W (:1138-1160)."""
    r = self.location
    dx = self._randn(3 * self.Nblobs)
    half = dx.view(-1, 3) * (self.rf_delta * self.a * 0.5)
    self._bind(r + half)
    plus = self._products_of_one_vector(kinds, dx)
    self._bind(r - half)
    return [p - m for p, m in zip(plus, self._products_of_one_vector(kinds, dx))]

  def compute_stochastic_linear_velocity(self, dt):
    """sqrt(2kT/dt) M_tt^{1/2} W + kT div(M_tt) by Lanczos + random finite difference (:1203-1260)."""
    z = self._randn(3 * self.Nblobs)
    self._bind(self.location)
    noise = self._lanczos(lambda v: self._product("tt", v), 3 * self.Nblobs, z, dt, product="tt")
    if self.kT > 0.0 and self.domain != "no_wall":
      (div_M_tt,) = self._random_finite_difference(("tt",))
      return noise + (self.kT / (self.rf_delta * self.a)) * div_M_tt
    return noise

  def compute_stochastic_linear_velocity_without_drift(self, dt):
    """sqrt(2kT/dt) M_tt^{1/2} W (:1315-1353)."""
    z = self._randn(3 * self.Nblobs)
    self._bind(self.location)
    return self._lanczos(lambda v: self._product("tt", v), 3 * self.Nblobs, z, dt, product="tt")

  def compute_linear_thermal_drift(self):
    """kT div(M_tt) by random finite difference (:1404-1434); zero without wall or at kT = 0."""
    if self.kT > 0.0 and self.domain != "no_wall":
      (div_M_tt,) = self._random_finite_difference(("tt",))
      return (self.kT / (self.rf_delta * self.a)) * div_M_tt
    return torch.zeros(3 * self.Nblobs, dtype=torch.float64, device=self.device)

  def grand_mobility(self, force_torque):
    """[v; w] = [[M_tt, M_tr], [M_rt, M_rr]] [F; T] on the bound configuration.  The reference applies the four
    blocks as four products per Lanczos iteration (:1114-1121); here all four come from ONE pass over the pairs
    (rmb_matvec_op_device, RMB_OP_GRAND: shared differences, inverse square roots and wall factors)."""
    n3 = 3 * self.Nblobs
    F, T = force_torque[:n3].contiguous(), force_torque[n3:].contiguous()
    self.mobility_products += 1
    out = torch.empty(2 * n3, dtype=torch.float64, device=force_torque.device)
    self.ctx.matvec_op_device("grand", (F, T), self.eta, outs=(out[:n3], out[n3:]))
    return out

  def compute_stochastic_velocity(self, dt):
    """Noise from the 6N grand mobility, stochastic torque solve for prescribed kinematics (:1082-1200)."""
    n3 = 3 * self.Nblobs
    z = self._randn(2 * n3)
    self._bind(self.location)
    noise = self._lanczos(self.grand_mobility, 2 * n3, z, dt, product="grand")
    if self.kT > 0.0 and self.domain != "no_wall":
      div_M_rt, div_M_tt = self._random_finite_difference(("rt", "tt"))
    else:
      div_M_rt = torch.zeros(n3, dtype=torch.float64, device=self.device)
      div_M_tt = torch.zeros(n3, dtype=torch.float64, device=self.device)
    self._bind(self.location)
    scale = self.kT / (self.rf_delta * self.a)
    if self.free_kinematics == "False":
      rhs = -noise[n3:] - div_M_rt * scale
      nrm = float(torch.linalg.norm(rhs))
      if nrm > 0:
        rhs = rhs / nrm
      sol, info = gmres_right_preconditioned(lambda x: self._product("rr", x), lambda x: x, rhs, tol=self.tolerance,
                                             restart=20, maxiter=1000, sync=getattr(self.ctx, "sync_scalars", None), ortho=self._ortho())
      self.det_iterations_count += info["iterations"]
      torque = sol * nrm if nrm > 0 else sol
      v_stoch = self._product("tr", torque)
    else:
      v_stoch = torch.zeros(n3, dtype=torch.float64, device=self.device)
    return v_stoch + noise[:n3] + scale * div_M_tt

  def compute_stochastic_linear_velocity_uncorrelated(self, dt):
    """Independent rollers: analytic M^{1/2} and drift (:1263-1312)."""
    z = self._randn(3 * self.Nblobs).view(-1, 3)
    c = self._self_mobility_coefficients(self.location)
    fd = math.sqrt(2 * self.kT / dt)
    v = torch.empty_like(z)
    v[:, 0:2] = fd * torch.sqrt(c["mu_tt_para"]).unsqueeze(1) * z[:, 0:2]
    v[:, 2] = fd * torch.sqrt(c["mu_tt_perp"]) * z[:, 2] + self.kT * c["dmu_tt_perp"]
    return v.reshape(-1)

  def compute_stochastic_linear_velocity_without_drift_uncorrelated(self, z, dt):
    """(:1356-1401)"""
    z = z.view(-1, 3)
    c = self._self_mobility_coefficients(self.location)
    fd = math.sqrt(2 * self.kT / dt)
    v = torch.empty_like(z)
    v[:, 0:2] = fd * torch.sqrt(c["mu_tt_para"]).unsqueeze(1) * z[:, 0:2]
    v[:, 2] = fd * torch.sqrt(c["mu_tt_perp"]) * z[:, 2]
    return v.reshape(-1)

  # ---- step bookkeeping -----------------------------------------------------------------------------
  def _valid(self, r):
    """A configuration is rejected when any roller is below the wall plane (:137-142)."""
    if self.domain != "single_wall":
      return True
    return not bool(torch.any(r[:, 2] < 0.0))

  def _accept(self, r_new):
    self.location = r_new
    self.consecutive_rejections = 0
    if self.domain == "single_wall":
      self.wall_overlaps += int(torch.count_nonzero(r_new[:, 2] < self.a))

  def _rejected(self):
    """A whole step is redrawn when ANY roller ends below the wall plane (:287-302; the reference prints 'Invalid
    configuration' and loops without bound).  With N rollers the chance that one of N independent draws fails grows
    with N, so a step size that a small suspension tolerates can reject nearly every draw of a large one: the
    rejections are reported as they happen and a step that fails `max_consecutive_retries` times in a row raises
    instead of spinning silently."""
    self.invalid_configuration_count += 1
    self.consecutive_rejections += 1
    if self.report_rejections:
      print("Invalid configuration (rejection %d of this step, %d in total, %d rollers)" %
            (self.consecutive_rejections, self.invalid_configuration_count, self.Nblobs), flush=True)
    if self.consecutive_rejections > self.max_consecutive_retries:
      raise RuntimeError("rollers: one time step was rejected %d times in a row (a roller below the wall plane in every "
                         "draw): the time step is too large for this configuration" % self.consecutive_rejections)
    if self.invalid_configuration_count > self.max_retries:
      raise RuntimeError("rollers: more than %d rejected steps" % self.max_retries)

  def _det(self):
    if self.hydro_interactions == 1:
      return self.compute_deterministic_velocity_and_torque()
    return self.compute_deterministic_velocity_and_torque_uncorrelated()

  def advance_time_step(self, dt, *args, **kwargs):
    if self._precision == 'single' and self.kT > 0.0 and self.scheme.startswith("stochastic"):
      _check_rfd_delta_for_single_precision(self.rf_delta)     # rf_delta may have been set after `precision`
    return getattr(self, self.scheme.replace("_rollers", ""))(dt, *args, **kwargs)

  # ---- schemes --------------------------------------------------------------------------------------
  def deterministic_forward_euler(self, dt, *args, **kwargs):
    while True:
      det_velocity, _ = self._det()
      r_new = self.location + dt * det_velocity.view(-1, 3)
      if self._valid(r_new):
        return self._accept(r_new)
      self._rejected()

  def stochastic_first_order(self, dt, *args, **kwargs):
    while True:
      det_velocity, _ = self._det()
      if self.hydro_interactions == 1:
        stoch_velocity = self.compute_stochastic_linear_velocity(dt)
      else:
        stoch_velocity = self.compute_stochastic_linear_velocity_uncorrelated(dt)
      r_new = self.location + dt * (det_velocity + stoch_velocity).view(-1, 3)
      if self._valid(r_new):
        return self._accept(r_new)
      self._rejected()

  def deterministic_adams_bashforth(self, dt, *args, **kwargs):
    while True:
      det_velocity, _ = self._det()
      if self.first_step is False and self.velocities_previous_step is not None:
        velocity = 1.5 * det_velocity - 0.5 * self.velocities_previous_step
      else:
        velocity = det_velocity
        self.first_step = False     # the reference clears it here even if the step is then rejected (:221) and would
                                    # fail on the retry (None previous velocities); the retry is forward Euler again
      r_new = self.location + dt * velocity.view(-1, 3)
      if self._valid(r_new):
        self.velocities_previous_step = det_velocity
        return self._accept(r_new)
      self._rejected()

  def stochastic_adams_bashforth(self, dt, *args, **kwargs):
    """The config-5 recipe: one fused det sweep + Lanczos on M_tt + 2 RFD sweeps per step."""
    while True:
      det_velocity, _ = self._det()
      if self.hydro_interactions == 1:
        stoch_velocity = self.compute_stochastic_linear_velocity(dt)
      else:
        stoch_velocity = self.compute_stochastic_linear_velocity_uncorrelated(dt)
      if self.first_step is False and self.velocities_previous_step is not None:
        velocity = 1.5 * det_velocity - 0.5 * self.velocities_previous_step + stoch_velocity
      else:
        velocity = det_velocity + stoch_velocity
        self.first_step = False
      r_new = self.location + dt * velocity.view(-1, 3)
      if self._valid(r_new):
        self.velocities_previous_step = det_velocity
        return self._accept(r_new)
      self._rejected()

  def stochastic_EM(self, dt, *args, **kwargs):
    while True:
      self.preprocess(self)
      W = self._randn(3 * self.Nblobs)
      det_velocity, _ = self._det()
      if self.hydro_interactions == 1:
        stoch_velocity = self.compute_stochastic_linear_velocity_without_drift(dt)
      else:
        stoch_velocity = self.compute_stochastic_linear_velocity_without_drift_uncorrelated(W, dt)
      r_new = self.location + dt * (det_velocity + stoch_velocity).view(-1, 3)
      self.postprocess(self)
      if self._valid(r_new):
        self.first_step = False
        self.velocities_previous_step = det_velocity
        return self._accept(r_new)
      self._rejected()

  def stochastic_GDC(self, dt, *args, **kwargs):
    while True:
      self.preprocess(self)
      r_old = self.location
      W = self._randn(3 * self.Nblobs)
      hydro = self.hydro_interactions == 1

      def brownian():
        if hydro:
          return self.compute_stochastic_linear_velocity_without_drift(dt)
        return self.compute_stochastic_linear_velocity_without_drift_uncorrelated(W, dt)

      stoch_n = brownian()
      # divergence of the Brownian velocity by a finite difference along z only (:403-421)
      shift = torch.zeros(3, dtype=torch.float64, device=self.device)
      shift[2] = self.rf_delta * self.a
      self.location = r_old + shift
      stoch_fd = brownian()
      dz = (stoch_fd.view(-1, 3)[:, 2] - stoch_n.view(-1, 3)[:, 2]) / (self.rf_delta * self.a)
      correction = (1.0 + 0.5 * dt * dz.sum()) if hydro else (1.0 + 0.5 * dt * dz).unsqueeze(1)
      self.location = r_old + stoch_n.view(-1, 3) * (dt / 2.0)
      if not self._valid(self.location):
        self.location = r_old
        self._rejected()
        continue
      det_velocity, _ = self._det()
      velocities_mid = (det_velocity + brownian()).view(-1, 3)
      r_new = r_old + velocities_mid * dt * correction
      self.location = r_old
      self.postprocess(self)
      if self._valid(r_new):
        self.first_step = False
        self.velocities_previous_step = det_velocity
        return self._accept(r_new)
      self._rejected()

  def _two_stage(self, dt, version):
    while True:
      r_old = self.location
      drift = self.compute_linear_thermal_drift()
      det_1, _ = self.compute_deterministic_velocity_and_torque()
      if version == "trapezoidal":
        stoch_1 = self.compute_stochastic_linear_velocity_without_drift(dt)
        stoch_2 = None
        first = dt
      else:
        stoch_1 = self.compute_stochastic_linear_velocity_without_drift(0.5 * dt)
        stoch_2 = self.compute_stochastic_linear_velocity_without_drift(0.5 * dt) if version == "mid_point_2" else None
        first = 0.5 * dt
      self.location = r_old + first * (det_1 + stoch_1).view(-1, 3)
      if not self._valid(self.location):
        self.location = r_old
        self._rejected()
        continue
      det_2, _ = self.compute_deterministic_velocity_and_torque()
      if version == "trapezoidal":
        velocity = 0.5 * (det_1 + det_2) + drift + stoch_1
      else:
        if stoch_2 is None:
          stoch_2 = self.compute_stochastic_linear_velocity_without_drift(0.5 * dt)
        velocity = det_2 + drift + (stoch_1 + stoch_2) * 0.5
      r_new = r_old + dt * velocity.view(-1, 3)
      if not self._valid(r_new):
        self.location = r_old
        self._rejected()
        continue
      return self._accept(r_new)

  def stochastic_mid_point(self, dt, *args, **kwargs):
    """Predictor to t + dt/2 with W_1, corrector with W_1 and a second noise at the mid point (:495-574)."""
    return self._two_stage(dt, "mid_point")

  def stochastic_mid_point_version_2(self, dt, *args, **kwargs):
    """As above with both noises generated at q^n (:577-656)."""
    return self._two_stage(dt, "mid_point_2")

  def stochastic_trapezoidal(self, dt, *args, **kwargs):
    """Predictor-corrector average of the deterministic velocity, one noise (:659-734)."""
    return self._two_stage(dt, "trapezoidal")


# ---- driver: the reference's input deck -> integrator -> time loop ------------------------------------
def integrator_from_input(read, device="cuda:0", ctx=None, rng=None):
  """Wire a RollersIntegrator from a ReadInput deck exactly as multi_bodies/multi_bodies.py:1322-1339 and
  :1379-1390 wire QuaternionIntegratorRollers.  Every `structure` must be a one-blob vertex file (the
  reference's Structures/blob.vertex); its .clones file lists the rollers.  With `seed` in the deck the
  random numbers are numpy's stream for that seed (the reference calls np.random.seed, :1157-1158)."""
  from . import deck_modes
  from . import structures as st
  deck_modes.validate(read, uses_dense_blocks=False)    # ValueError for modes this engine does not run
  locations = []
  body_types = []
  for vertex_file, clones_file in [s[:2] for s in read.structures]:
    ref = deck_modes.uniform_vertices(st.read_vertex_file(read.resolve(vertex_file)), read.blob_radius, vertex_file)
    if len(ref) != 1:
      raise ValueError("%s has %d blobs: the roller schemes are for single-blob bodies (use RigidSuspension)" %
                       (vertex_file, len(ref)))
    _n, loc, _quat = st.read_clones_file(read.resolve(clones_file))
    locations.append(np.asarray(loc, dtype=np.float64).reshape(-1, 3) + ref[0])
    body_types.append(len(locations[-1]))
  if not locations:
    raise ValueError("input deck lists no structure")
  if rng is None:
    rng = read.random_generator(save=False)
  from .rigid_integrator import replicate_rng
  rng = replicate_rng(rng, read, ctx, device)
  integ = RollersIntegrator(np.concatenate(locations), read.scheme, read.blob_radius, read.eta,
                            tolerance=read.solver_tolerance, domain=read.domain, device=device, ctx=ctx, rng=rng)
  integ.kT = read.kT
  integ.rf_delta = read.rf_delta
  integ.g = read.g
  integ.repulsion_strength_wall = read.repulsion_strength_wall
  integ.debye_length_wall = read.debye_length_wall
  if read.blob_blob_force_implementation != "None":
    integ.repulsion_strength = read.repulsion_strength
    integ.debye_length = read.debye_length
  integ.periodic_length = np.asarray(read.periodic_length, dtype=np.float64)
  integ.omega_one_roller = np.asarray(read.omega_one_roller, dtype=np.float64)
  integ.free_kinematics = read.free_kinematics
  integ.hydro_interactions = read.hydro_interactions
  integ.body_types = body_types
  integ.structures_ID = list(read.structures_ID)
  return integ


def _write_clones(fh, locations):
  fh.write(str(len(locations)) + "\n")
  for x in locations:
    fh.write("%s %s %s %s %s %s %s\n" % (x[0], x[1], x[2], 1.0, 0.0, 0.0, 0.0))


def run(read, integrator, output_name=None, n_steps=None, callback=None):
  """Time loop of multi_bodies.py:1412-1530 for rollers: every n_save steps (and after the last one) the
  locations go to `<output_name>.<ID>.<step>.clones` (save_clones one_file_per_step) or are appended to
  `<output_name>.<ID>.config` (one_file), in the reference's text format; orientation is the identity
  (rollers do not track it).  Returns the integrator."""
  output_name = read.output_name if output_name is None else output_name
  n_steps = read.n_steps if n_steps is None else n_steps
  if read.save_clones not in ("one_file_per_step", "one_file"):
    raise ValueError('save_clones = %s is not implemented; use "one_file_per_step" or "one_file"' % read.save_clones)
  files = None
  if read.save_clones == "one_file":
    files = [open(output_name + "." + ID + ".config", "w") for ID in integrator.structures_ID]

  def save(step):
    loc = integrator.location.cpu().numpy()
    offset = 0
    for i, ID in enumerate(integrator.structures_ID):
      block = loc[offset:offset + integrator.body_types[i]]
      offset += integrator.body_types[i]
      if files is not None:
        _write_clones(files[i], block)
      else:
        with open(output_name + "." + ID + "." + str(step).zfill(8) + ".clones", "w") as fh:
          _write_clones(fh, block)

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
