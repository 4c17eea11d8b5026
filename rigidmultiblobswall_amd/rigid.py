"""Device-resident rigid-multiblob mobility solve.

The caller side of the hot path: what multi_bodies/multi_bodies.py:424-471 (`linear_operator_rigid`),
:474-618 (block-diagonal preconditioner) and
quaternion_integrator/quaternion_integrator_multi_bodies.py:1441-1547 (`solve_mobility_problem`) do
with numpy + scipy on the host, here with every vector resident in HBM:

    |  M   -K | |lambda|   | slip |          lambda : constraint forces on the blobs (3 N_blobs)
    | -K^T  0 | |  U   | = |  -F  |          U      : (v, omega) of each body      (6 N_bodies)

  * M.lambda is the HIP pair sweep (rmb_matvec_device, symmetric kernel) -- the only O(N^2) piece;
  * K.U / K^T.lambda are batched (3 n_b x 6) products per body shape (torch.bmm);
  * the preconditioner solves each body alone: dense per-body blob mobility built on the device
    (rmb_body_mobility_dense_device), batched Cholesky -> explicit M_b^-1, N = (K^T M_b^-1 K)^-1
    (multi_bodies.py:516-531), applied with batched GEMMs only (:548-560);
  * right-preconditioned restarted GMRES(60), relative tolerance on the true residual, RHS normalised
    to 1 (general_application_utils.py:514-627, quaternion_integrator_multi_bodies.py:1518-1537).
    Krylov vectors stay on the device; per iteration only the new Hessenberg column crosses PCIe.

PyTorch is used for device memory and batched small dense algebra only.
"""
import gc
import math
import os

import numpy as np
import torch

from .context import MobilityContext


def quaternion_rotation_matrix(q):
  """Rotation matrix of a unit quaternion (s, p) -- same convention as
  quaternion_integrator/quaternion.py:41-51.  q: (..., 4) -> (..., 3, 3)."""
  q = np.asarray(q, dtype=np.float64)
  s, p0, p1, p2 = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
  d = s * s - 0.5
  R = np.empty(q.shape[:-1] + (3, 3))
  R[..., 0, 0] = p0 * p0 + d;       R[..., 0, 1] = p0 * p1 - s * p2;  R[..., 0, 2] = p0 * p2 + s * p1
  R[..., 1, 0] = p1 * p0 + s * p2;  R[..., 1, 1] = p1 * p1 + d;       R[..., 1, 2] = p1 * p2 - s * p0
  R[..., 2, 0] = p2 * p0 - s * p1;  R[..., 2, 1] = p2 * p1 + s * p0;  R[..., 2, 2] = p2 * p2 + d
  return 2.0 * R


def blob_positions(reference_configuration, location, quaternion):
  """r = ref . R(q)^T + location   (body/body.py:64-78)."""
  R = quaternion_rotation_matrix(quaternion)
  return np.asarray(reference_configuration) @ R.T + np.asarray(location)


def quaternion_rotation_matrix_torch(q):
  """Batched rotation matrices, same formula as above, on whatever device q lives.  (n,4) -> (n,3,3)."""
  s, p0, p1, p2 = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
  d = s * s - 0.5
  R = torch.stack([p0 * p0 + d, p0 * p1 - s * p2, p0 * p2 + s * p1,
                   p1 * p0 + s * p2, p1 * p1 + d, p1 * p2 - s * p0,
                   p2 * p0 - s * p1, p2 * p1 + s * p0, p2 * p2 + d], dim=1)
  return 2.0 * R.view(-1, 3, 3)


def quaternion_from_rotation_torch(phi):
  """Quaternion of the rotation vector phi = omega dt (quaternion_integrator/quaternion.py:17-27).  (n,3) -> (n,4)."""
  n = torch.linalg.norm(phi, dim=1, keepdim=True)
  safe = torch.where(n > 0, n, torch.ones_like(n))
  return torch.cat([torch.cos(0.5 * n), torch.where(n > 0, torch.sin(0.5 * n) / safe, torch.zeros_like(n)) * phi], dim=1)


def quaternion_multiply_torch(q, r):
  """q * r with q the LEFT quaternion (quaternion.py:30-39).  (n,4),(n,4) -> (n,4)."""
  qs, qp = q[:, 0:1], q[:, 1:4]
  rs, rp = r[:, 0:1], r[:, 1:4]
  return torch.cat([qs * rs - (qp * rp).sum(dim=1, keepdim=True), qs * rp + rs * qp + torch.cross(qp, rp, dim=1)], dim=1)


def _body_mobility_from_resistance(A):
  """(K^T M^-1 K)^+ per body (multi_bodies.py:531 uses pinv).  For bodies with a full-rank 6x6 resistance -- everything
  but single blobs and collinear rods -- the plain batched inverse is the pseudo-inverse and is ~100x cheaper than the
  batched SVD behind torch.linalg.pinv (measured: 0.14 ms vs 34.6 ms for 21845 bodies); rank-deficient groups are
  detected by the residual of the inverse and take the pinv route."""
  inv, info = torch.linalg.inv_ex(A)
  eye = torch.eye(6, dtype=A.dtype, device=A.device)
  ok = bool((info == 0).all()) and bool(torch.isfinite(inv).all()) and \
      float((torch.bmm(A, inv) - eye).abs().max()) < 1e-8
  if not ok:
    return torch.linalg.pinv(A)
  return 0.5 * (inv + inv.transpose(1, 2))


class _Group(object):
  """All bodies that share one reference configuration size n_b."""
  __slots__ = ("n_b", "body_idx", "first_blob", "blob_idx", "blob_idx3", "ref", "rel", "K", "K_pc", "Minv", "Nbody", "Lchol",
               "Linv", "A11", "A12", "A21", "A22")


class RigidSuspension(object):
  """Rigid bodies made of blobs, on one GPU.

  reference_configurations: list with one (n_b, 3) array per body (bodies may differ in shape);
  locations (n_bodies, 3); quaternions (n_bodies, 4) as (s, p1, p2, p3).  Blobs are numbered body after
  body, as multi_bodies.py:194-204 does.  The configuration lives on the device: `set_configuration`
  moves every body (rotation matrices, blob coordinates and K are batched tensor ops), which is what the
  time integrators (rigid_integrator.py) call between solves.
  """

  def __init__(self, reference_configurations, locations, quaternions, a, eta, wall=True, periodic_length=None,
               device="cuda:0", ctx=None, prescribed=None, prescribed_velocity=None):
    """prescribed: optional bool per body -- bodies with prescribed kinematics (the reference's `obstacle` structures,
    multi_bodies.py:1201-1203): their velocity is given (prescribed_velocity, default 0) and the unknown in their U slot
    is the force-torque that holds them (multi_bodies.py:457-462, :561-571)."""
    self.device = torch.device(device)
    self.a, self.eta, self.wall = float(a), float(eta), bool(wall)
    self.L = np.zeros(3) if periodic_length is None else np.asarray(periodic_length, dtype=np.float64)
    refs = [np.asarray(c, dtype=np.float64).reshape(len(c), -1)[:, :3] for c in reference_configurations]
    self.n_bodies = len(refs)
    sizes = np.array([len(c) for c in refs])
    first = np.concatenate([[0], np.cumsum(sizes)])
    self.n_blobs = int(first[-1])
    self.first_blob = first[:-1]
    self.body_sizes = sizes
    self.ctx = ctx if ctx is not None else MobilityContext(self.device.index or 0)
    self._own_ctx = ctx is None
    # groups of equal n_b
    self.groups = []
    for n_b in sorted(set(sizes.tolist())):
      g = _Group()
      g.n_b = int(n_b)
      idx = np.nonzero(sizes == n_b)[0]
      g.body_idx = torch.as_tensor(idx, device=self.device, dtype=torch.int64)
      g.first_blob = torch.as_tensor(first[idx], device=self.device, dtype=torch.int64)
      blob = first[idx][:, None] + np.arange(n_b)[None, :]                       # (nb_g, n_b)
      comp = (3 * blob[:, :, None] + np.arange(3)[None, None, :]).reshape(len(idx), 3 * n_b)
      g.blob_idx = torch.as_tensor(blob, device=self.device, dtype=torch.int64)
      g.blob_idx3 = torch.as_tensor(comp, device=self.device, dtype=torch.int64)
      g.ref = torch.as_tensor(np.array([refs[k] for k in idx]), device=self.device)      # (nb_g, n_b, 3)
      g.rel = g.K = g.K_pc = g.Minv = g.Nbody = g.Lchol = g.Linv = g.A11 = g.A12 = g.A21 = g.A22 = None
      self.groups.append(g)
    self.size = 3 * self.n_blobs + 6 * self.n_bodies
    self.matvec_count = 0       # M.v products requested (a two-vector pass counts two)
    self.matvec2_count = 0      # of which pairs served by one two-vector pass
    self.sweep_count = 0        # passes over the blob pairs actually launched (a k-vector pass counts once)
    self.lockstep_width = 4     # vectors per pass in lockstep products (1 = one pass per vector)
    self.free = None            # (n_bodies, 1) 1.0 = free body, 0.0 = prescribed kinematics; None = all free
    self.prescribed_velocity = None
    if prescribed is not None and np.any(prescribed):
      self.free = torch.as_tensor(1.0 - np.asarray(prescribed, dtype=np.float64).reshape(-1, 1), device=self.device)
      pv = np.zeros((self.n_bodies, 6)) if prescribed_velocity is None else np.asarray(prescribed_velocity, dtype=np.float64)
      self.prescribed_velocity = torch.as_tensor(pv.reshape(self.n_bodies, 6), device=self.device) * (1.0 - self.free)
    self.set_configuration(locations, quaternions)

  # ---- configuration --------------------------------------------------------------------------------
  def _as_dev(self, x, cols):
    if isinstance(x, torch.Tensor):
      return x.to(device=self.device, dtype=torch.float64).reshape(-1, cols)
    return torch.as_tensor(np.asarray(x, dtype=np.float64).reshape(-1, cols), device=self.device)

  def blob_positions_device(self, locations, quaternions):
    """(n_blobs, 3) blob coordinates of a configuration, r = ref . R(q)^T + location (body/body.py:64-78);
    also returns the per-group body-frame offsets.  Does not touch the bound configuration."""
    loc, quat = self._as_dev(locations, 3), self._as_dev(quaternions, 4)
    r = torch.empty((self.n_blobs, 3), dtype=torch.float64, device=self.device)
    if len(self.groups) == 1 and self._native_blocks():
      rel = torch.empty((self.n_bodies, self.groups[0].n_b, 3), dtype=torch.float64, device=self.device)
      self._native_blocks().rigid_configuration_device(self.groups[0].ref, loc.contiguous(), quat.contiguous(), r, rel)
      return r, [rel]
    rels = []
    for g in self.groups:
      R = quaternion_rotation_matrix_torch(quat[g.body_idx])
      rel = torch.bmm(g.ref, R.transpose(1, 2))
      r[g.blob_idx.reshape(-1)] = (rel + loc[g.body_idx].unsqueeze(1)).reshape(-1, 3)
      rels.append(rel)
    return r, rels

  def set_configuration(self, locations, quaternions):
    """Move the bodies: new blob coordinates, new K = [I, rot] with rot x = -(rel x x) (body/body.py:81-115), and the
    context's packed positions.  The preconditioner is NOT touched (the reference also keeps the one built at time
    level n for the other solves of a step); call build_preconditioner() to refresh it."""
    self.location, self.orientation = self._as_dev(locations, 3).clone(), self._as_dev(quaternions, 4).clone()
    if len(self.groups) == 1 and self._native_blocks():
      # one launch: blob coordinates, body-frame offsets and K (rmb_rigid_configuration_device), into storage that stays
      g = self.groups[0]
      if g.K is None or getattr(self, "_r_buf", None) is None:
        self._r_buf = torch.empty((self.n_blobs, 3), dtype=torch.float64, device=self.device)
        g.rel = torch.empty((self.n_bodies, g.n_b, 3), dtype=torch.float64, device=self.device)
        g.K = torch.empty((self.n_bodies, 3 * g.n_b, 6), dtype=torch.float64, device=self.device)
      self._native_blocks().rigid_configuration_device(g.ref, self.location, self.orientation, self._r_buf, g.rel, g.K)
      self.r_dev = self._r_buf.view(-1)
      self.ctx.set_positions(self.r_dev, self.a, self.L, self.wall)
      return
    r, rels = self.blob_positions_device(self.location, self.orientation)
    for g, rel in zip(self.groups, rels):
      g.rel = rel
      # K keeps its storage from one configuration to the next (captured Arnoldi iterations refer to it, _ArnoldiGraphs)
      if g.K is None:
        g.K = torch.zeros((rel.shape[0], 3 * g.n_b, 6), dtype=torch.float64, device=self.device)
        K = g.K.view(rel.shape[0], g.n_b, 3, 6)
        K[:, :, 0, 0] = 1.0; K[:, :, 1, 1] = 1.0; K[:, :, 2, 2] = 1.0
      K = g.K.view(rel.shape[0], g.n_b, 3, 6)
      K[:, :, 0, 4] = rel[:, :, 2];  K[:, :, 0, 5] = -rel[:, :, 1]
      K[:, :, 1, 3] = -rel[:, :, 2]; K[:, :, 1, 5] = rel[:, :, 0]
      K[:, :, 2, 3] = rel[:, :, 1];  K[:, :, 2, 4] = -rel[:, :, 0]
    self.r_dev = r.reshape(-1)
    self.ctx.set_positions(self.r_dev, self.a, self.L, self.wall)

  # ---- dense operators of the whole suspension (the reference's "dense algebra" schemes and one-shot utilities) ----
  def dense_blob_mobility(self):
    """(3N, 3N) blob mobility at the bound configuration: body_dense_tt_kernel with the whole suspension as one "body"
    (the reference calls self.mobility_blobs(r_vectors, eta, a), multi_bodies.py:207-230), symmetrised."""
    first = torch.zeros(1, dtype=torch.int64, device=self.device)
    M = self.ctx.body_mobility_dense_device(first, self.n_blobs, self.eta)[0]
    return 0.5 * (M + M.t())

  def dense_K(self):
    """(3N, 6 n_bodies) block-diagonal geometric matrix (multi_bodies.py:300-324)."""
    K = torch.zeros((3 * self.n_blobs, 6 * self.n_bodies), dtype=torch.float64, device=self.device)
    for g in self.groups:
      for k, body in enumerate(g.body_idx.tolist()):
        K[g.blob_idx3[k], 6 * body:6 * body + 6] = g.K[k]
    return K

  @property
  def r_vectors(self):
    return self.r_dev.detach().cpu().numpy().reshape(-1, 3)

  def close(self):
    ws = getattr(self, "_arnoldi_ws", None)
    if ws is not None:
      ws.release()
      self._arnoldi_ws = None
    ns = getattr(self, "_arnoldi_native", None)
    if ns is not None:
      ns.close()
      self._arnoldi_native = None
    lw = getattr(self, "_lanczos_ws", None)
    if lw is not None:
      lw["mapped"].close()
      self._lanczos_ws = None
    if self._own_ctx:
      self.ctx.close()

  # ---- per-group views: with one body shape (the common case) blobs of a group are the whole vector in order,
  #      so gathers / scatters are reshapes and cost no kernel -----------------------------------------
  def _blobs_of(self, x, g):
    """(3 n_blobs,) -> (nb_g, 3 n_b)"""
    if len(self.groups) == 1:
      return x.reshape(self.n_bodies, 3 * g.n_b)
    return x[g.blob_idx3.reshape(-1)].view(len(g.body_idx), 3 * g.n_b)

  def _put_blobs(self, out, g, values):
    if len(self.groups) == 1:
      out.view(self.n_bodies, 3 * g.n_b).copy_(values.reshape(self.n_bodies, 3 * g.n_b))
    else:
      out[g.blob_idx3.reshape(-1)] = values.reshape(-1)

  def _bodies_of(self, U, g):
    """(n_bodies, 6) -> (nb_g, 6)"""
    return U if len(self.groups) == 1 else U[g.body_idx]

  def _put_bodies(self, out, g, values):
    if len(self.groups) == 1:
      out.copy_(values.reshape(self.n_bodies, -1))
    else:
      out[g.body_idx] = values.reshape(len(g.body_idx), -1)

  # ---- pieces of the operator -------------------------------------------------------------------
  def mobility_times_lambda(self, lam):
    self.matvec_count += 1
    self.sweep_count += 1
    return self.ctx.matvec_device("tt", lam.contiguous(), self.eta)

  def mobility_times_lambdas(self, lams):
    """M applied to several blob vectors with as few passes over the pairs as possible: up to `lockstep_width` vectors
    share one pass (rmb_matvec_op_device, RMB_OP_TT_MULTI: the vector-independent part of every pair -- differences,
    both inverse square roots, RPY and wall coefficients -- is evaluated once)."""
    lams = [l.contiguous() for l in lams]
    out = []
    width = max(1, int(self.lockstep_width))
    for lo in range(0, len(lams), width):
      chunk = lams[lo:lo + width]
      self.matvec_count += len(chunk)
      self.sweep_count += 1
      if len(chunk) == 1 or not hasattr(self.ctx, "matvec_op_device"):
        self.sweep_count += len(chunk) - 1
        out.extend(self.ctx.matvec_device("tt", v, self.eta) for v in chunk)
      else:
        if len(chunk) == 2:
          self.matvec2_count += 1
        out.extend(self.ctx.matvec_op_device("tt_multi", chunk, self.eta))
    return out

  def K_times_U(self, U):
    """U (6 n_bodies,) -> (3 n_blobs,)   (multi_bodies.py:327-349)."""
    U = U.view(self.n_bodies, 6)
    out = torch.empty(3 * self.n_blobs, dtype=torch.float64, device=self.device)
    for g in self.groups:
      self._put_blobs(out, g, torch.bmm(g.K, self._bodies_of(U, g).unsqueeze(-1)))
    return out

  def KT_times_lambda(self, lam):
    """lambda (3 n_blobs,) -> (6 n_bodies,)   (multi_bodies.py:352-375)."""
    out = torch.empty((self.n_bodies, 6), dtype=torch.float64, device=self.device)
    for g in self.groups:
      self._put_bodies(out, g, torch.bmm(g.K.transpose(1, 2), self._blobs_of(lam, g).unsqueeze(-1)))
    return out.reshape(-1)

  def apply_operator(self, x):
    """[lambda; U] -> [M lambda - K U; -K^T lambda]   (multi_bodies.py:424-471, no constraints)."""
    n3 = 3 * self.n_blobs
    lam, U = x[:n3], x[n3:]
    if self.free is None and len(self.groups) == 1:
      # one body shape: the sweep writes straight into the result and the two K products are one batched GEMM each
      # (5 kernels per application instead of 11)
      g = self.groups[0]
      res = torch.empty_like(x)
      top = res[:n3]
      self.matvec_count += 1
      self.sweep_count += 1
      helper = self._native_blocks()
      if helper is not None and helper is self.ctx and x.is_contiguous() and g.n_b <= 256 and self.fused_operator:
        # sweep + ONE finishing launch (self term, scaling, - K U, -K^T lambda): rmb_rigid_operator_device
        return helper.rigid_operator_device(g.K, x, self.eta, res)
      lam = lam.contiguous()
      r = self.ctx.matvec_device("tt", lam, self.eta, out=top)
      if r.data_ptr() != top.data_ptr():       # contexts that do not write in place (test stand-ins)
        top.copy_(r)
      if self._native_products():
        # top -= K U and bottom = -K^T lambda in one launch (rmb_block_apply_device; K^T = K with exchanged strides)
        self._native_blocks().block_apply_device(None, g.K, g.K, None, lam.view(self.n_bodies, 3 * g.n_b), U.reshape(self.n_bodies, 6),
                                    top.view(self.n_bodies, 3 * g.n_b), res[n3:].view(self.n_bodies, 6), alpha=-1.0, beta1=1.0,
                                    transpose=(False, False, True, False))
        return res
      top.view(self.n_bodies, 3 * g.n_b, 1).baddbmm_(g.K, U.reshape(self.n_bodies, 6, 1), alpha=-1.0)
      bot = res[n3:].view(self.n_bodies, 6, 1)
      torch.baddbmm(bot, g.K.transpose(1, 2), lam.reshape(self.n_bodies, 3 * g.n_b, 1), beta=0.0, alpha=-1.0, out=bot)
      return res
    if self.free is None:
      top = self.mobility_times_lambda(lam) - self.K_times_U(U)
      return torch.cat([top, -self.KT_times_lambda(lam)])
    # prescribed bodies: their slot holds F, which enters only the force balance  -K^T lambda + F = 0
    U = U.view(self.n_bodies, 6)
    top = self.mobility_times_lambda(lam) - self.K_times_U((U * self.free).reshape(-1))
    bottom = -self.KT_times_lambda(lam).view(self.n_bodies, 6) + U * (1.0 - self.free)
    return torch.cat([top, bottom.reshape(-1)])

  def apply_operator2(self, xa, xb):
    """The operator applied to two vectors with ONE pass over the blob pairs (rmb_matvec2_device); the K products
    stay per vector (they are O(N)).  Falls back to two applications when the fast path does not apply."""
    if self.free is not None or len(self.groups) != 1 or not hasattr(self.ctx, "matvec2_device"):
      return self.apply_operator(xa), self.apply_operator(xb)
    n3 = 3 * self.n_blobs
    g = self.groups[0]
    res = torch.empty((2, self.size), dtype=torch.float64, device=self.device)
    self.matvec_count += 2
    self.matvec2_count += 1
    self.sweep_count += 1
    ra, rb = self.ctx.matvec2_device("tt", xa[:n3].contiguous(), xb[:n3].contiguous(), self.eta,
                                     out_a=res[0, :n3], out_b=res[1, :n3])
    for row, r, x in ((res[0], ra, xa), (res[1], rb, xb)):
      top = row[:n3]
      if r.data_ptr() != top.data_ptr():
        top.copy_(r)
      if self._native_products() and x.is_contiguous():
        self._native_blocks().block_apply_device(None, g.K, g.K, None, x[:n3].view(self.n_bodies, 3 * g.n_b), x[n3:].view(self.n_bodies, 6),
                                    top.view(self.n_bodies, 3 * g.n_b), row[n3:].view(self.n_bodies, 6), alpha=-1.0, beta1=1.0,
                                    transpose=(False, False, True, False))
        continue
      top.view(self.n_bodies, 3 * g.n_b, 1).baddbmm_(g.K, x[n3:].reshape(self.n_bodies, 6, 1), alpha=-1.0)
      bot = row[n3:].view(self.n_bodies, 6, 1)
      torch.baddbmm(bot, g.K.transpose(1, 2), x[:n3].reshape(self.n_bodies, 3 * g.n_b, 1), beta=0.0, alpha=-1.0, out=bot)
    return res[0], res[1]

  def prescribe(self, rhs):
    """RHS of a system whose bodies partly have prescribed kinematics (quaternion_integrator_multi_bodies.py:1478-1487):
    slip += K U_prescribed on their blobs, and their force rows are zero."""
    if self.free is None:
      return rhs
    n3 = 3 * self.n_blobs
    rhs = rhs.clone()
    rhs[:n3] += self.K_times_U(self.prescribed_velocity.reshape(-1))
    rhs[n3:] = (rhs[n3:].view(self.n_bodies, 6) * self.free).reshape(-1)
    return rhs

  def impose_prescribed_velocity(self, sol):
    """After a solve the velocity slots of prescribed bodies carry their known velocity (:1541-1544)."""
    if self.free is None:
      return sol
    n3 = 3 * self.n_blobs
    sol = sol.clone()
    sol[n3:] = (sol[n3:].view(self.n_bodies, 6) * self.free + self.prescribed_velocity).reshape(-1)
    return sol

  # ---- block-diagonal preconditioner ------------------------------------------------------------
  def build_preconditioner(self):
    """Per body: M_b (dense, device kernel), Cholesky, N_b = (K^T M_b^-1 K)^-1 (multi_bodies.py:516-531)."""
    for g in self.groups:
      Mb = self.ctx.body_mobility_dense_device(g.first_blob, g.n_b, self.eta)
      if self._native_pc(g, Mb):
        continue
      Mb = 0.5 * (Mb + Mb.transpose(1, 2))
      g.Lchol = torch.linalg.cholesky(Mb)
      # explicit inverse, as the reference stores it (mobility_inv_blobs, multi_bodies.py:524): applying
      # the preconditioner is then pure batched GEMM.  (Back-to-back batched cholesky_solve calls inside
      # the Krylov loop were observed to race on ROCm 7.0 torch; bmm does not.)
      g.Minv = torch.cholesky_inverse(g.Lchol)
      g.Minv = 0.5 * (g.Minv + g.Minv.transpose(1, 2))
      if g.K_pc is not None and g.K_pc.shape == g.K.shape:     # the K of the configuration the preconditioner was built at
        g.K_pc.copy_(g.K)
      else:
        g.K_pc = g.K.clone()
      g.Linv = None
      g.Nbody = _body_mobility_from_resistance(torch.bmm(g.K.transpose(1, 2), torch.bmm(g.Minv, g.K)))
      # The preconditioner is linear in (slip, F): [lambda; U] = [[A11, A12], [A21, A22]] [slip; F] with
      #   A12 = -M^-1 K N,  A11 = M^-1 + A12 K^T M^-1,  A21 = A12^T,  A22 = -N      (multi_bodies.py:548-560 expanded),
      # so applying it is four batched GEMMs.  Prescribed bodies (:561-571): lambda = M^-1 slip, slot = K^T M^-1 slip.
      MinvK = torch.bmm(g.Minv, g.K)
      A12 = -torch.bmm(MinvK, g.Nbody)
      A11 = g.Minv + torch.bmm(A12, MinvK.transpose(1, 2))
      A21 = A12.transpose(1, 2)
      A22 = -g.Nbody
      if self.free is not None:
        fr = self._bodies_of(self.free, g).unsqueeze(-1)
        A11 = fr * A11 + (1.0 - fr) * g.Minv
        A12 = fr * A12
        A21 = fr * A21 + (1.0 - fr) * MinvK.transpose(1, 2)
        A22 = fr * A22
      # the four blocks keep their storage from one build to the next (captured Arnoldi iterations refer to them)
      for name, new in (("A11", A11), ("A12", A12), ("A21", A21), ("A22", A22)):
        old = getattr(g, name)
        if old is not None and old.shape == new.shape:
          old.copy_(new)
        else:
          setattr(g, name, new.contiguous())
    if self.device.type == "cuda":
      torch.cuda.synchronize(self.device)
    return self

  def _native_pc(self, g, Mb):
    """The factors and blocks of group g in one launch (rmb_rigid_preconditioner_device, up to 42 blobs per body -- the
    reference's 12- and 42-blob shells -- all bodies free).  False = not applicable, or a body's 6 x 6 resistance has no accurate inverse (single blobs, collinear
    rods: the pseudo-inverse route of the torch path below, remembered for the group)."""
    if (self.free is not None or 3 * g.n_b > 128 or not self._native_blocks()
        or any(g is r for r in getattr(self, "_native_pc_rejected", ()))):
      return False
    nb, n = Mb.shape[0], 3 * g.n_b
    shapes = dict(Lchol=(nb, n, n), Linv=(nb, n, n), Minv=(nb, n, n), Nbody=(nb, 6, 6), A11=(nb, n, n), A12=(nb, n, 6),
                  A21=(nb, 6, n), A22=(nb, 6, 6))
    fresh = {}
    for name, shape in shapes.items():
      cur = getattr(g, name)
      fresh[name] = cur if (cur is not None and tuple(cur.shape) == shape and cur.is_contiguous()) else \
          torch.empty(shape, dtype=torch.float64, device=self.device)
    info = getattr(self, "_pc_info", None)
    if info is None:
      info = self._pc_info = torch.zeros(1, dtype=torch.int32, device=self.device)
    self._native_blocks().rigid_preconditioner_device(Mb.contiguous(), g.K, *[fresh[k] for k in ("Lchol", "Linv", "Minv", "Nbody", "A11", "A12", "A21", "A22")],
                                         info)
    if int(info) != 0:
      self._native_pc_rejected = tuple(getattr(self, "_native_pc_rejected", ())) + (g,)
      return False
    for name, t in fresh.items():
      setattr(g, name, t)
    if g.K_pc is not None and g.K_pc.shape == g.K.shape:
      g.K_pc.copy_(g.K)
    else:
      g.K_pc = g.K.clone()
    return True

  def apply_preconditioner(self, x):
    """Solve every body alone (multi_bodies.py:548-560):
       Lt = M^-1 slip;  Y = N (-F - K^T Lt);  lambda = M^-1 (slip + K Y);  U = Y
    applied through the blocks A11..A22 assembled by build_preconditioner."""
    n3 = 3 * self.n_blobs
    out = torch.empty_like(x)
    F = x[n3:].view(self.n_bodies, 6)
    outU = out[n3:].view(self.n_bodies, 6)
    if len(self.groups) == 1:
      g = self.groups[0]
      slip = x[:n3].reshape(self.n_bodies, 3 * g.n_b, 1)
      lam = out[:n3].view(self.n_bodies, 3 * g.n_b, 1)
      if self._native_products() and x.is_contiguous():
        self._native_blocks().block_apply_device(g.A11, g.A12, g.A21, g.A22, x[:n3].view(self.n_bodies, 3 * g.n_b), F,
                                    out[:n3].view(self.n_bodies, 3 * g.n_b), outU)      # the four blocks in one launch
        return out
      U = outU.unsqueeze(-1)
      torch.bmm(g.A11, slip, out=lam)
      lam.baddbmm_(g.A12, F.unsqueeze(-1))
      torch.bmm(g.A21, slip, out=U)
      U.baddbmm_(g.A22, F.unsqueeze(-1))
      return out
    for g in self.groups:
      slip = self._blobs_of(x[:n3], g).unsqueeze(-1)
      Fg = self._bodies_of(F, g).unsqueeze(-1)
      self._put_blobs(out[:n3], g, torch.baddbmm(torch.bmm(g.A12, Fg), g.A11, slip))
      self._put_bodies(outU, g, torch.baddbmm(torch.bmm(g.A22, Fg), g.A21, slip))
    return out

  # ---- solve ------------------------------------------------------------------------------------
  def solve(self, rhs, tol=1e-8, restart=60, maxiter=1000, x0=None):
    """Device-level solve of [M -K; -K^T 0] x = rhs at the bound configuration with the stored preconditioner
    (built on first use).  RHS normalised to 1 before GMRES (quaternion_integrator_multi_bodies.py:1518-1521).
    x0: optional initial guess in the units of x.  Returns (x tensor, info)."""
    if self.groups[0].Lchol is None:
      self.build_preconditioner()
    if (x0 is None and self.native_gmres is not False and os.environ.get("RMB_NATIVE_GMRES", "") != "0"
        and self._native_step_applies(restart)):
      # The whole loop inside the library (rmb_rigid_gmres_device): the norm of the right-hand side, per iteration one step
      # call and one event, the Givens rotations and the convergence test in C one iteration behind the device -- the
      # Python loop below costs 49 us of host time per iteration against 32-36 us of GPU time on a small deck
      # (profiles/r5_gmres_step.txt).
      g = self.groups[0]
      sol, info = self.ctx.rigid_gmres_device(g.A11, g.A12, g.A21, g.A22, g.K, rhs.contiguous(), tol, restart, maxiter, self.eta)
      if info["rhs_norm"] == 0.0:
        return sol, dict(iterations=0, residual=0.0, converged=True, history=[])
      self.matvec_count += info["operator_applications"]
      self.sweep_count += info["operator_applications"]
      info["native_gmres"] = True
      return sol, info
    nrm = float(torch.linalg.norm(rhs))
    if nrm == 0.0:
      return torch.zeros_like(rhs), dict(iterations=0, residual=0.0, converged=True, history=[])
    ortho = self._ortho(restart)
    ns = self._native_arnoldi(restart)
    if ns is not None:
      # One C call per Arnoldi iteration enqueues all of its launches (rmb_rigid_arnoldi_step_device: preconditioner blocks,
      # pair sweep, finishing launch with the K products, fused Gram-Schmidt that also stores the Hessenberg column into
      # mapped host memory): 7 launches from one host call, no graph, no copy command.
      ns.bind(self)
      sol, info = gmres_right_preconditioned(self.apply_operator, self.apply_preconditioner, rhs / nrm, tol=tol,
                                             restart=restart, maxiter=maxiter, x0=None if x0 is None else x0 / nrm,
                                             ws=ns, on_replay=self._count_operator, ortho=ortho)
      info["rhs_norm"] = nrm
      info["native_steps"] = ns.steps_this_solve
      return sol * nrm, info
    ws = self._arnoldi_graphs(restart)
    if ws is None:
      sol, info = gmres_right_preconditioned(self.apply_operator, self.apply_preconditioner, rhs / nrm, tol=tol,
                                             restart=restart, maxiter=maxiter, x0=None if x0 is None else x0 / nrm,
                                             sync=getattr(self.ctx, "sync_scalars", None), lag=getattr(self, "gmres_lag", None),
                                             ortho=ortho)
      info["rhs_norm"] = nrm
      return sol * nrm, info
    # Small systems: the device side of every Arnoldi iteration (preconditioner, operator, Gram-Schmidt, normalisation,
    # column to page-locked memory: ~16 launches) is one captured hipGraph per iteration index, replayed from the third
    # solve on.  The whole solve runs on the workspace's stream (a capture cannot happen on the default stream, and the
    # context must already enqueue on the capturing stream when a capture begins).
    cur = torch.cuda.current_stream(self.device)
    ws.stream.wait_stream(cur)
    with torch.cuda.stream(ws.stream):
      self.ctx._follow_torch_stream()
      ws.begin_solve()
      sol, info = gmres_right_preconditioned(self.apply_operator, self.apply_preconditioner, rhs / nrm, tol=tol,
                                             restart=restart, maxiter=maxiter, x0=None if x0 is None else x0 / nrm,
                                             ws=ws, on_replay=self._count_operator, ortho=ortho)
      sol = sol * nrm
    cur.wait_stream(ws.stream)
    sol.record_stream(cur)
    info["rhs_norm"] = nrm
    info["graph_replays"] = ws.replays_this_solve
    return sol, info

  fused_operator = True       # False: the product and the K products as separate launches (A/B, tests)
  native_step = None          # None = automatic, False = never: one C call per Arnoldi iteration (_ArnoldiNative)
  native_gmres = None         # None = automatic, False = never: the whole GMRES loop in one C call (rmb_rigid_gmres_device)

  def _native_step_applies(self, restart):
    want = self.native_step
    if os.environ.get("RMB_NATIVE_STEP", "") == "0":
      want = False
    return not (want is False or self.gmres_graph is True or self.free is not None or len(self.groups) != 1 or self.device.type != "cuda"
                or type(self.ctx) is not MobilityContext or self._native_products() is not self.ctx or not self.fused_operator
                or getattr(self, "gmres_lag", None) is False or not (0 < restart < 256))

  def _native_arnoldi(self, restart):
    """The one-call-per-iteration workspace for solve(), or None.  Applies to what rmb_rigid_arnoldi_step_device covers: one
    body shape of at most `native_products_max_blobs` blobs, all bodies free, a plain single-GPU context, the host bookkeeping one iteration late.
    `gmres_graph = True` (forced captured iterations) and `native_step = False` / RMB_NATIVE_STEP=0 turn it off."""
    if not self._native_step_applies(restart):
      return None
    ns = getattr(self, "_arnoldi_native", None)
    if ns is None or ns.m != restart or ns.n != self.size:
      if ns is not None:
        ns.close()
      ns = self._arnoldi_native = _ArnoldiNative(self.size, restart, self.device)
    return ns

  # `native_helpers`: the O(N) pieces between two sweeps (K / K^T products, the preconditioner's four blocks, the
  # Gram-Schmidt of an Arnoldi step) as the library's own kernels (csrc/rmb_krylov.hip) instead of batched-GEMM / GEMV
  # launches: None = automatic (a plain single-GPU context), False = the torch operations.
  native_helpers = None

  def _native_blocks(self):
    """The MobilityContext the helper kernels are launched through, or None (torch operations).  A plain context is its
    own helper; a facade over several ranks (distributed.ReplicatedContext) hands out the rank's context: the helpers
    are rank-local O(N) work on replicated vectors with fixed-order reductions, so every rank computes the same bits."""
    want = self.native_helpers
    if want is None:
      want = os.environ.get("RMB_NATIVE_HELPERS", "") != "0"
    if not want or self.device.type != "cuda":
      return None
    if type(self.ctx) is MobilityContext:
      return self.ctx
    h = getattr(self.ctx, "helper_context", None)
    return h if type(h) is MobilityContext else None

  def _native_products(self):
    """_native_blocks() for the batched block products (one workgroup per body: thread = row up to 32 blobs per body,
    wave = row with coalesced loads above, csrc/rmb_krylov.hip two_by_two_rows), rocBLAS batched GEMM beyond
    `native_products_max_blobs` per body, where one CU per body no longer carries the block's bytes."""
    return self._native_blocks() if max(g.n_b for g in self.groups) <= self.native_products_max_blobs else None

  native_products_max_blobs = 64

  def _ortho(self, restart):
    """The fused Gram-Schmidt step for _gmres_steps, or None (torch operations)."""
    h = self._native_blocks()
    return h.krylov_orthogonalize_device if h is not None and restart < 256 else None

  def _count_operator(self):
    self.matvec_count += 1
    self.sweep_count += 1

  # `gmres_graph`: None = automatic (on for a plain single-GPU context up to `gmres_graph_max_blobs` blobs, where the
  # iteration is launch-bound: profiles/r4_gmres_graph.txt), True / False = forced.  RMB_GMRES_GRAPH=0 turns it off.
  gmres_graph = None
  gmres_graph_max_blobs = 4096

  def _arnoldi_graphs(self, restart):
    """The captured-iteration workspace for solve(), or None when the plain loop is to run."""
    want = self.gmres_graph
    if os.environ.get("RMB_GMRES_GRAPH", "") == "0":
      want = False
    if want is None:
      want = self.n_blobs <= self.gmres_graph_max_blobs
    if (not want or self.device.type != "cuda" or type(self.ctx) is not MobilityContext
        or getattr(self, "gmres_lag", None) is False or self.ctx.get_option("timing") != 0):
      return None
    ws = getattr(self, "_arnoldi_ws", None)
    if ws is None or ws.m != restart:
      ws = self._arnoldi_ws = _ArnoldiGraphs(self.size, restart, self.device)
    ws.buffers = getattr(self.ctx, "buffers_signature", None)
    if self._ortho(restart) is not None and os.environ.get("RMB_MAPPED_COLUMNS", "") != "0":
      ws.use_mapped_columns()
    ptr = lambda t: None if t is None else t.data_ptr()
    ws.bind((self.ctx.launch_signature(), self.eta, self._native_blocks() is not None, ptr(self.free), ptr(self.prescribed_velocity),
             tuple(tuple(ptr(t) for t in (g.K, g.A11, g.A12, g.A21, g.A22)) for g in self.groups)))
    return ws

  def solve_mixed_precision(self, rhs, tol=1e-8, inner_tol=3e-5, restart=60, maxiter=1000, max_outer=8):
    """The same saddle-point solve by iterative refinement with a single-precision inner operator -- MI355X issues
    fp32 1.6x faster than fp64 and the fp32 twin of the pair sweep (csrc/sym32_kernels.h, context option "precision")
    runs at 1.5-1.6x the fp64 one.  Outer loop in fp64: r = rhs - A x with the fp64 operator; inner loop: the SAME
    right-preconditioned GMRES (same preconditioner) on A_32 dx = r to the loose relative tolerance `inner_tol`;
    x += dx.  The returned residual is the true fp64 one, so the result meets `tol` exactly as `solve` does; what
    differs from the reference's flow is the number of Krylov restarts (one per outer step), not the stopping rule.
    Where the fp32 kernel does not apply (pseudo-periodic domain, fewer than 128 blobs) the inner products fall back to
    fp64 and this is plain restarted GMRES.  Not used by default: `solve` is the reference's algorithm.
    Measured (2048 shells x 12 blobs, tol 1e-8, tools/experiments/exp_mixed_precision_solve.py): 15.8 ms against 18.5 ms -- 21 fp32
    sweeps + 3 fp64 ones instead of 19 fp64 sweeps; at this size the per-iteration host work limits the gain to 1.17x."""
    if self.groups[0].Lchol is None:
      self.build_preconditioner()
    nrm = float(torch.linalg.norm(rhs))
    if nrm == 0.0:
      return torch.zeros_like(rhs), dict(iterations=0, residual=0.0, converged=True, history=[], outer_iterations=0)
    b = rhs / nrm
    x = torch.zeros_like(b)
    r = b
    res, its, outer, history = 1.0, 0, 0, []
    sync = getattr(self.ctx, "sync_scalars", None)
    # the caller's precision (RigidIntegrator.precision = 'single' sets 32) is what the context goes back to afterwards
    get = getattr(self.ctx, "get_option", None)
    previous = (get("precision") if get is not None else None) or 64
    try:
      while res > tol and outer < max_outer and its < maxiter:
        self.ctx.set_option("precision", 32)
        dx, info = gmres_right_preconditioned(self.apply_operator, self.apply_preconditioner, r / res,
                                              tol=max(inner_tol, 0.25 * tol / res), restart=restart, maxiter=maxiter - its,
                                              sync=sync, ortho=self._ortho(restart))
        self.ctx.set_option("precision", 64)
        its += info["iterations"]
        history.extend(h * res for h in info["history"])
        x = x + dx * res
        r = b - self.apply_operator(x)                    # fp64
        t = torch.linalg.vector_norm(r).reshape(1)
        if sync is not None:
          sync(t)
        res = float(t)
        outer += 1
    finally:
      self.ctx.set_option("precision", previous)
    return x * nrm, dict(iterations=its, residual=res, converged=bool(res <= tol), history=history, outer_iterations=outer,
                         rhs_norm=nrm)

  def solve_pair(self, rhs_a, rhs_b, tol=1e-8, restart=60, maxiter=1000):
    """Two solves with the same operator and preconditioner advanced in lockstep (gmres_pair_right_preconditioned): each
    sees exactly its own GMRES iterates, but while both run every iteration costs one two-vector pair sweep instead of
    two sweeps.  Returns ((x_a, info_a), (x_b, info_b))."""
    if self.groups[0].Lchol is None:
      self.build_preconditioner()
    na, nb_ = float(torch.linalg.norm(rhs_a)), float(torch.linalg.norm(rhs_b))
    if na == 0.0 or nb_ == 0.0:
      return self.solve(rhs_a, tol, restart, maxiter), self.solve(rhs_b, tol, restart, maxiter)
    (xa, ia), (xb, ib) = gmres_pair_right_preconditioned(self.apply_operator, self.apply_operator2, self.apply_preconditioner,
                                                         rhs_a / na, rhs_b / nb_, tol=tol, restart=restart, maxiter=maxiter,
                                                         sync=getattr(self.ctx, "sync_scalars", None), ortho=self._ortho(restart))
    ia["rhs_norm"], ib["rhs_norm"] = na, nb_
    return (xa * na, ia), (xb * nb_, ib)

  # ---- lockstep tasks: coroutines that yield blob vectors and receive M . vector ----------------------
  def operator_from_product(self, x, Mlam):
    """[M lambda - K U; -K^T lambda] given the blob product M lambda (the O(N) rest of apply_operator)."""
    n3 = 3 * self.n_blobs
    lam, U = x[:n3], x[n3:]
    if self.free is None and len(self.groups) == 1 and self._native_products() and x.is_contiguous():
      g = self.groups[0]
      res = torch.empty_like(x)
      res[:n3].copy_(Mlam)
      self._native_blocks().block_apply_device(None, g.K, g.K, None, lam.view(self.n_bodies, 3 * g.n_b), U.view(self.n_bodies, 6),
                                  res[:n3].view(self.n_bodies, 3 * g.n_b), res[n3:].view(self.n_bodies, 6), alpha=-1.0, beta1=1.0,
                                  transpose=(False, False, True, False))
      return res
    if self.free is None:
      return torch.cat([Mlam - self.K_times_U(U), -self.KT_times_lambda(lam)])
    U = U.view(self.n_bodies, 6)
    top = Mlam - self.K_times_U((U * self.free).reshape(-1))
    bottom = -self.KT_times_lambda(lam).view(self.n_bodies, 6) + U * (1.0 - self.free)
    return torch.cat([top, bottom.reshape(-1)])

  def solve_task(self, rhs, tol=1e-8, restart=60, maxiter=1000, x0=None):
    """`solve` as a lockstep task (run_lockstep): yields the blob part of every vector GMRES needs the operator applied
    to, receives M . that vector, returns (x, info)."""
    if self.groups[0].Lchol is None:
      self.build_preconditioner()
    nrm = float(torch.linalg.norm(rhs))
    if nrm == 0.0:
      return torch.zeros_like(rhs), dict(iterations=0, residual=0.0, converged=True, history=[], rhs_norm=0.0)
    n3 = 3 * self.n_blobs
    steps = _gmres_steps(self.apply_preconditioner, rhs / nrm, tol, restart, maxiter, None if x0 is None else x0 / nrm,
                         getattr(self.ctx, "sync_scalars", None), ortho=self._ortho(restart))
    try:
      y = next(steps)
      while True:
        Mlam = yield y[:n3]
        y = steps.send(self.operator_from_product(y, Mlam))
    except StopIteration as done:
      sol, info = done.value
    info["rhs_norm"] = nrm
    return sol * nrm, info

  def forcing_task(self, z, factor, tol=1e-8, print_residual=False):
    """`stochastic_forcing` (preconditioned Lanczos, P = blockdiag(L_b^-T)) as a lockstep task: returns (noise, its)."""
    from .stochastic import _lanczos_steps, _prepare
    if self.groups[0].Lchol is None:
      self.build_preconditioner()
    self._stochastic_factors()
    z, dim, _ = _prepare(z, None, self.device, None)
    steps = _lanczos_steps(factor, tol, 1000, dim, z, print_residual, self.device, getattr(self.ctx, "sync_scalars", None),
                           ortho=self._ortho(0))
    try:
      w = next(steps)
      while True:
        Mx = yield self._blockdiag(w, "Linv", transpose=True)
        w = steps.send(self._blockdiag(Mx, "Linv"))
    except StopIteration as done:
      noise, its = done.value
    return self._blockdiag(noise, "Lchol").reshape(-1), its

  @staticmethod
  def product_task(v):
    """One product M . v as a lockstep task."""
    res = yield v
    return res

  def run_lockstep(self, tasks):
    """Advance independent tasks that all need products with the mobility of the BOUND configuration (GMRES solves,
    Lanczos recursions, single products): every round collects one request per running task and serves them with one
    k-vector pass over the pairs (mobility_times_lambdas).  Each task sees exactly the iterates it would see alone.
    Returns the tasks' return values, in order."""
    n = len(tasks)
    requests, results, running = [None] * n, [None] * n, [True] * n
    for k, t in enumerate(tasks):
      try:
        requests[k] = next(t)
      except StopIteration as done:
        results[k], running[k] = done.value, False
    while any(running):
      act = [k for k in range(n) if running[k]]
      answers = self.mobility_times_lambdas([requests[k] for k in act])
      for k, ans in zip(act, answers):
        try:
          requests[k] = tasks[k].send(ans)
        except StopIteration as done:
          results[k], running[k] = done.value, False
    return results

  def solve_mobility_problem(self, slip=None, force_torque=None, tol=1e-8, restart=60, maxiter=1000, x0=None,
                             mixed_precision=False):
    """Returns (velocities (n_bodies, 6), lambda (n_blobs, 3), info).  RHS = [slip, -F]
    (quaternion_integrator_multi_bodies.py:1458-1475).  mixed_precision = True: `solve_mixed_precision` (same
    tolerance on the true fp64 residual, fp32 inner products)."""
    n3 = 3 * self.n_blobs
    rhs = torch.zeros(self.size, dtype=torch.float64, device=self.device)
    if slip is not None:
      rhs[:n3] = torch.as_tensor(np.asarray(slip, dtype=np.float64).reshape(-1), device=self.device)
    if force_torque is not None:
      rhs[n3:] = -torch.as_tensor(np.asarray(force_torque, dtype=np.float64).reshape(-1), device=self.device)
    if mixed_precision:
      sol, info = self.solve_mixed_precision(rhs, tol=tol, restart=restart, maxiter=maxiter)
    else:
      sol, info = self.solve(rhs, tol=tol, restart=restart, maxiter=maxiter)
    return sol[n3:].view(-1, 6).cpu().numpy(), sol[:n3].view(-1, 3).cpu().numpy(), info

  # ---- Brownian forcing with the block-diagonal stochastic preconditioner ----------------------------
  def _stochastic_factors(self):
    """Per body M_b = L L^T (the Cholesky factor of the preconditioner build) and L^-1, explicit, so applying them is
    batched GEMM (multi_bodies.py:516-531 stores P = L^-T and P_inv = L the same way)."""
    for g in self.groups:
      if g.Linv is None:
        eye = torch.eye(3 * g.n_b, dtype=torch.float64, device=self.device).expand(len(g.body_idx), -1, -1)
        g.Linv = torch.linalg.solve_triangular(g.Lchol, eye, upper=False).contiguous()    # (the library's step kernels address it by strides of a dense batch)

  def _blockdiag(self, x, which, transpose=False):
    out = torch.empty_like(x)
    if len(self.groups) == 1 and self._native_products() and x.is_contiguous():
      g = self.groups[0]
      A = g.Linv if which == "Linv" else g.Lchol
      none = x.new_empty((self.n_bodies, 0))
      self._native_blocks().block_apply_device(A, None, None, None, x.view(self.n_bodies, 3 * g.n_b), none, out.view(self.n_bodies, 3 * g.n_b), none,
                                  transpose=(bool(transpose), False, False, False))      # one launch, no gather / scatter
      return out
    for g in self.groups:
      A = g.Linv if which == "Linv" else g.Lchol
      if transpose:
        A = A.transpose(1, 2)
      self._put_blobs(out, g, torch.bmm(A, self._blobs_of(x, g).unsqueeze(-1)))
    return out

  def _pc_mobility(self):
    """w -> P^T M P w and the two-vector form (u, v) -> (P^T M P u, P^T M P v), P = blockdiag(L_b^-T)."""
    def one(w):
      return self._blockdiag(self.mobility_times_lambda(self._blockdiag(w, "Linv", transpose=True)), "Linv")

    def two(u, v):
      if not hasattr(self.ctx, "matvec2_device"):
        return one(u), one(v)
      self.matvec_count += 2
      self.matvec2_count += 1
      self.sweep_count += 1
      a, b = self.ctx.matvec2_device("tt", self._blockdiag(u, "Linv", transpose=True),
                                     self._blockdiag(v, "Linv", transpose=True), self.eta)
      return self._blockdiag(a, "Linv"), self._blockdiag(b, "Linv")
    return one, two

  def stochastic_forcing(self, z, factor, tol=1e-8, print_residual=False):
    """factor * P^-1 (P^T M P)^{1/2} z with P = blockdiag(L_b^-T): the preconditioned Lanczos of
    quaternion_integrator_multi_bodies.py:966-973 / multi_bodies.py:590-614 (covariance factor^2 M; needs O(1)
    iterations per decade because each body's own block is the identity).  Uses the preconditioner of the
    configuration it was built at, and the mobility of the bound configuration.  Returns (noise, iterations)."""
    from .stochastic import stochastic_forcing_lanczos
    if self.groups[0].Lchol is None:
      self.build_preconditioner()
    self._stochastic_factors()
    native = self._lanczos_native(z, factor, tol, print_residual)
    if native is not None:
      return native
    one, _ = self._pc_mobility()
    return stochastic_forcing_lanczos(factor=factor, tolerance=tol, dim=3 * self.n_blobs, mobility_mult=one,
                                      L_mult=lambda x: self._blockdiag(x, "Lchol"), z=z, print_residual=print_residual,
                                      device=self.device, sync=getattr(self.ctx, "sync_scalars", None), ortho=self._ortho(0))

  native_lanczos = None       # None = automatic, False = never: the library's Lanczos step / loop (_lanczos_native)
  native_lanczos_loop = None  # None = automatic, False = one C call per iteration under a Python loop instead of rmb_rigid_lanczos_device
  lanczos_native_loop_calls = 0
  lanczos_native_rows = 96    # basis rows of the native loop; a forcing that needs more falls back to the generic loop
  lanczos_native_max_blobs = 20000  # above, the iteration it discards at the end (a whole pair sweep) costs more than the host waits it
                                    # saves: 12 288 blobs 5.33 -> 4.55 ms per forcing, 16 392: 8.23 -> 7.63, 24 576: level (exp_lanczos_threshold.py)

  def _lanczos_native(self, z, factor, tol, print_residual, max_iter=1000):
    """The preconditioned Lanczos forcing inside the library: the whole loop in one call (rmb_rigid_lanczos_device, the
    default), or -- `native_lanczos_loop = False`, RMB_NATIVE_LANCZOS_LOOP=0, print_residual -- the loop below with
    ONE library call per iteration (rmb_rigid_lanczos_step_device: two block
    launches around the pair sweep + the fused Gram-Schmidt, which also stores h_ii and h_{i+1,i} into mapped host
    memory), the host side (the small tridiagonal eigenproblem and the reference's stopping rule,
    stochastic_forcing.py:239-255) running ONE ITERATION LATE -- so the device never waits for numpy.  Same iterates,
    coefficients and iteration count as stochastic.stochastic_forcing_lanczos; the price is one discarded iteration at
    the end.  Returns (noise, iterations), or None when it does not apply (several body shapes, more than 32 blobs per
    body, a facade context, an exact breakdown, more iterations than `lanczos_native_rows`) -- the caller then runs the
    generic loop from the start."""
    from .stochastic import _noise_coefficients
    want = self.native_lanczos
    if os.environ.get("RMB_NATIVE_LANCZOS", "") == "0":
      want = False
    if (want is False or factor == 0.0 or len(self.groups) != 1 or self.device.type != "cuda" or type(self.ctx) is not MobilityContext
        or self._native_products() is not self.ctx or (want is None and self.n_blobs > self.lanczos_native_max_blobs)):
      return None
    g = self.groups[0]
    n3, cap = 3 * self.n_blobs, int(self.lanczos_native_rows)
    if not g.Linv.is_contiguous():
      g.Linv = g.Linv.contiguous()
    loop = self.native_lanczos_loop
    if os.environ.get("RMB_NATIVE_LANCZOS_LOOP", "") == "0":
      loop = False
    if loop is not False and not print_residual and 2 <= cap <= 254 and g.Lchol.is_contiguous():
      # the loop itself inside the library (rmb_rigid_lanczos_device): no Python between the iterations
      zt = torch.as_tensor(z, dtype=torch.float64, device=self.device).reshape(-1).contiguous()
      noise, its, products = self.ctx.rigid_lanczos_device(g.Linv, g.Lchol, zt, factor, tol, max_iter, cap, self.eta)
      self.matvec_count += products
      self.sweep_count += products
      self.lanczos_native_loop_calls += 1
      return None if noise is None else (noise, its)
    ws = getattr(self, "_lanczos_ws", None)
    if ws is None or ws["n3"] != n3 or ws["cap"] != cap:
      from .context import MappedHostArray
      ws = self._lanczos_ws = dict(n3=n3, cap=cap, V=torch.empty((cap + 1, n3), dtype=torch.float64, device=self.device),
                                   col=torch.zeros((cap, cap + 2), dtype=torch.float64, device=self.device),
                                   y=torch.empty(n3, dtype=torch.float64, device=self.device),
                                   w=torch.empty(n3, dtype=torch.float64, device=self.device), mapped=MappedHostArray((cap, cap + 2)),
                                   events=[torch.cuda.Event(), torch.cuda.Event()])
    V, host = ws["V"], ws["mapped"].array
    z = torch.as_tensor(z, dtype=torch.float64, device=self.device).reshape(-1)
    v_norm = float(torch.linalg.norm(z))
    V[0] = z / v_norm
    ctx = self.ctx
    ctx._follow_torch_stream()
    stream = torch.cuda.current_stream(self.device)
    fn = ctx._lib.rmb_rigid_lanczos_step_device
    if not g.Linv.is_contiguous():
      g.Linv = g.Linv.contiguous()
    head = (ctx._h, g.K.shape[0], g.n_b, g.Linv.data_ptr(), V.data_ptr(), V.stride(0))
    tail = (float(self.eta), ws["y"].data_ptr(), ws["w"].data_ptr())
    col_ptr, mapped_ptr, row = ws["col"].data_ptr(), ws["mapped"].dev_ptr, 8 * (cap + 2)
    if not g.Linv.is_contiguous():
      g.Linv = g.Linv.contiguous()
    h_diag, h_sup = [], []
    coef_old, coef, its, done = None, None, None, False

    def enqueue(i):
      rc = fn(*head, i, *tail, col_ptr + i * row, mapped_ptr + i * row)
      if rc != 0:
        from . import _lib
        _lib.check(rc)
      self.matvec_count += 1
      self.sweep_count += 1
      ws["events"][i & 1].record(stream)

    def finish(i):
      """Host side of iteration i (its two coefficients are in mapped memory once its event has completed).  True = stop."""
      nonlocal coef_old, coef, its
      ws["events"][i & 1].synchronize()
      hd_f, hs_f = float(host[i, i]), float(host[i, i + 1])
      if not (hs_f > 0 and np.isfinite(hs_f)):
        return None                          # exact breakdown: rare; the generic loop handles it
      h_diag.append(hd_f)
      h_sup.append(hs_f)
      coef = _noise_coefficients(h_diag, h_sup, i + 1, v_norm * factor)
      if i > 0:
        old = np.concatenate([coef_old, [0.0]])
        old_norm = np.linalg.norm(old)
        diff = np.linalg.norm(coef - old)
        if print_residual:
          if i == 1:
            print('lanczos =  0 1')
          print('lanczos = ', i, diff / old_norm)
        if diff / max(old_norm, np.finfo(float).eps) < tol:
          its = i
          return True
      coef_old = coef
      return False

    enqueue(0)
    i = 0
    while True:
      nxt = i + 1
      if nxt < cap and nxt <= max_iter:
        enqueue(nxt)                         # the device goes on while the host looks at iteration i
      stop = finish(i)
      if stop is None:
        torch.cuda.synchronize(self.device)
        return None
      if stop:
        break
      if nxt >= cap or nxt > max_iter:
        if nxt > max_iter:
          its = max_iter
          break
        torch.cuda.synchronize(self.device)  # more basis rows needed than the workspace holds: generic loop
        return None
      i = nxt
    k = len(coef)
    noise = V[:k].t() @ torch.as_tensor(coef, dtype=torch.float64, device=self.device)
    return self._blockdiag(noise, "Lchol").reshape(-1), its

  def stochastic_forcing_pair(self, z_a, factor_a, z_b, factor_b, tol=1e-8, print_residual=False):
    """Two forcings with the same mobility in lockstep (stochastic_forcing_lanczos_pair): one two-vector pair sweep per
    iteration while both run.  Returns ((noise_a, its_a), (noise_b, its_b))."""
    from .stochastic import stochastic_forcing_lanczos_pair
    if self.groups[0].Lchol is None:
      self.build_preconditioner()
    self._stochastic_factors()
    one, two = self._pc_mobility()
    return stochastic_forcing_lanczos_pair((factor_a, factor_b), (z_a, z_b), one, two, tolerance=tol,
                                           L_mult=lambda x: self._blockdiag(x, "Lchol"), print_residual=print_residual,
                                           device=self.device, sync=getattr(self.ctx, "sync_scalars", None), ortho=self._ortho(0))

# page-locked staging rows for the Hessenberg columns of running solves (allocated once, handed out per solve)
_pinned_pool = []


class _ArnoldiNative(object):
  """Static workspace of GMRES(restart) whose device side of an iteration is ONE call into the library
  (MobilityContext.rigid_arnoldi_step_device).  Same interface as _ArnoldiGraphs towards _gmres_steps: V, cols, host_cols
  (here page-locked memory mapped into the device's address space: the Gram-Schmidt kernel stores the column there
  itself) and run(j, ...)."""

  def __init__(self, n, restart, device):
    from .context import MappedHostArray
    self.n, self.m, self.device = int(n), int(restart), device
    self.V = torch.zeros((self.m + 1, self.n), dtype=torch.float64, device=device)
    self.cols = torch.zeros((self.m, self.m + 2), dtype=torch.float64, device=device)
    self.z = torch.empty(self.n, dtype=torch.float64, device=device)
    self.w = torch.empty(self.n, dtype=torch.float64, device=device)
    self.mapped = MappedHostArray((self.m, self.m + 2))
    self.host_cols = self.mapped.array
    self.owner = None
    self.steps = self.steps_this_solve = 0

  def bind(self, owner):
    """Once per solve: everything of the step call that does not change from one iteration to the next, as plain integers
    (building sixteen ctypes objects per iteration costs more host time than the GPU needs for the iteration)."""
    self.owner = owner
    self.steps_this_solve = 0
    ctx, g = owner.ctx, owner.groups[0]
    for t in (g.A11, g.A12, g.A21, g.A22, g.K):
      assert t.is_contiguous()
    ctx._follow_torch_stream()                   # the solve stays on the stream that is current now
    self._fn = ctx._lib.rmb_rigid_arnoldi_step_device
    self._head = (ctx._h, g.K.shape[0], g.K.shape[1] // 3, g.A11.data_ptr(), g.A12.data_ptr(), g.A21.data_ptr(), g.A22.data_ptr(),
                  g.K.data_ptr(), self.V.data_ptr(), self.V.stride(0))
    self._tail = (float(owner.eta), self.z.data_ptr(), self.w.data_ptr())
    self._cols_ptr, self._mapped_ptr, self._row = self.cols.data_ptr(), self.mapped.dev_ptr, 8 * (self.m + 2)

  def run(self, j, body, on_replay=None):
    rc = self._fn(*self._head, j, *self._tail, self._cols_ptr + j * self._row, self._mapped_ptr + j * self._row)
    if rc != 0:
      from . import _lib
      _lib.check(rc)
    if on_replay is not None:
      on_replay()
    self.steps += 1
    self.steps_this_solve += 1

  def close(self):
    self.owner = None
    if self.mapped is not None:
      self.host_cols = None
      self.mapped.close()
      self.mapped = None


class _ArnoldiGraphs(object):
  """Static workspace of GMRES(restart) on one system size and, per iteration index j, a captured hipGraph of everything
  the DEVICE does in that iteration.  On systems of a few thousand blobs an iteration is ~16 small launches whose
  enqueueing costs more host time than they take to run (tools/experiments/exp_small_deck_gmres.py: ~200 us per
  iteration around a 10-20 us blob product); replaying a graph is one call.

  An index j runs eagerly the first time it is met, is captured once `capture_after` solves have been seen, and is
  replayed from then on.  The graphs hold pointers: to this workspace, to the operator's K and preconditioner blocks
  (rewritten in place by set_configuration / build_preconditioner), to the context's packed positions and accumulators.
  bind() drops them whenever the signature the owner hands over changes."""
  capture_after = 2

  def __init__(self, n, restart, device):
    self.n, self.m, self.device = int(n), int(restart), device
    self.V = torch.zeros((self.m + 1, self.n), dtype=torch.float64, device=device)
    self.cols = torch.zeros((self.m, self.m + 2), dtype=torch.float64, device=device)
    self.host_cols = torch.zeros((self.m, self.m + 2), dtype=torch.float64).pin_memory()
    # The fused Gram-Schmidt kernel can store the new Hessenberg column straight into page-locked memory that is mapped
    # into the device's address space (context.MappedHostArray): one graph node (the copy) less per iteration.  Set up by
    # the owner when its context has the entry point; host_cols then IS that memory (a numpy array).
    self.mapped_cols = None
    self.stream = torch.cuda.Stream(device)
    self.graphs, self.seen, self.signature = {}, set(), None
    self.solves = self.captures = self.replays = self.replays_this_solve = 0
    self.buffers = None               # callable: the context's buffers_signature() (set by the owner), checked before a replay
    self.buffers_at_capture = None
    self.stale_drops = 0

  def bind(self, signature):
    if signature != self.signature:
      self.release()
      self.signature = signature

  def use_mapped_columns(self):
    if self.mapped_cols is None:
      from .context import MappedHostArray
      self.mapped_cols = MappedHostArray((self.m, self.m + 2))
      self.host_cols = self.mapped_cols.array

  def col_mapped_ptr(self, j):
    """Device address of row j of the mapped column buffer, or 0."""
    return 0 if self.mapped_cols is None else self.mapped_cols.dev_ptr + 8 * j * (self.m + 2)

  def release(self):
    """Destroy the graphs now (a safe point: nothing is capturing) rather than whenever the collector finds them.  A graph
    of the previous solve may still be executing -- the lagged bookkeeping leaves its last, discarded iteration in flight,
    and a stale-buffer drop happens in the middle of a solve: wait for the device first, destroying an executing
    hipGraphExec is not something to rely on (an intermittent hang of a slip-scheme test on the box was traced to here)."""
    if self.graphs and self.device.type == "cuda":
      torch.cuda.synchronize(self.device)
    self.graphs.clear()
    self.seen.clear()
    self.solves = 0

  def begin_solve(self):
    self.solves += 1
    self.replays_this_solve = 0

  def run(self, j, body, on_replay=None):
    g = self.graphs.get(j)
    if g is not None and self.buffers is not None and self.buffers() != self.buffers_at_capture:
      # The graph holds the addresses of the context's internal buffers by value, and one of them has moved since the
      # capture (another, larger suspension used the shared context; a product grew a scratch buffer): every graph is
      # stale.  Same treatment as a changed signature in bind(): drop them, run eagerly again, capture afresh after
      # `capture_after` further solves.
      self.release()
      self.solves = 1            # this solve is the first of the new series
      self.stale_drops += 1
      g = None
    if g is None:
      if j not in self.seen or self.solves <= self.capture_after:
        body()                                   # eager: also warms every library call of this iteration's shapes
        self.seen.add(j)
        return
      g = torch.cuda.CUDAGraph()
      torch.cuda.synchronize(self.device)
      # No cyclic garbage collection while the stream is capturing: collecting a dead CUDAGraph (another suspension's,
      # say) calls hipGraphDestroy, which HIP refuses during a capture -- and the refusal surfaces in a destructor.
      gc_was_on = gc.isenabled()
      gc.disable()
      try:
        g.capture_begin(capture_error_mode="thread_local")
        try:
          body()                                 # enqueues nothing: recorded into the graph (the owner counts it)
        finally:
          g.capture_end()
      finally:
        if gc_was_on:
          gc.enable()
      if self.buffers is not None:
        now = self.buffers()
        if self.graphs and now != self.buffers_at_capture:    # the eager warm-ups have sized everything: never expected
          self.graphs.clear()
          self.stale_drops += 1
        self.buffers_at_capture = now
      self.graphs[j] = g
      self.captures += 1
    elif on_replay is not None:
      on_replay()
    g.replay()
    self.replays += 1
    self.replays_this_solve += 1


def _pinned_columns(rows, cols):
  for k, t in enumerate(_pinned_pool):
    if t.shape[0] >= rows and t.shape[1] >= cols:
      return _pinned_pool.pop(k)
  return torch.empty((max(rows, 62), max(cols, 63)), dtype=torch.float64).pin_memory()


def _gmres_steps(Minv, b, tol, restart, maxiter, x0, sync, lag=None, ws=None, A=None, on_replay=None, ortho=None):
  """GMRES(restart) on A.Minv written as a coroutine: it YIELDS every vector it needs the operator applied to and
  receives A(vector) back, so one driver can serve a single solve (gmres_right_preconditioned) or advance two solves
  in lockstep and hand both requests to a two-vector operator (gmres_pair_right_preconditioned).  Returns (x, info).

  On a GPU the host side of an iteration (Givens rotations on the new Hessenberg column, the convergence test) runs ONE
  ITERATION LATE (`lag`, default on for CUDA tensors): the column is normalised on the device, copied to page-locked
  memory asynchronously, and read only after the NEXT iteration's preconditioner + operator + Gram-Schmidt have been
  enqueued -- the device never waits for the host between sweeps.  The iterates, the stopping rule and the iteration
  count are those of the plain loop; what the lag can cost is one discarded sweep when the solve converges earlier than
  its own history predicts, so the loop turns synchronous as soon as the last observed reduction rate says the next
  column may meet the tolerance (normally the last two or three iterations)."""
  dev = b.device
  n = b.numel()
  if lag is None:
    lag = dev.type == "cuda"
  if ws is not None:        # captured iterations (_ArnoldiGraphs): static buffers, the operator applied inside the step
    assert A is not None and sync is None and dev.type == "cuda" and ws.n == n and ws.m == restart
    lag = True
  if not lag or sync is not None:      # the fused Gram-Schmidt normalises on the device right away
    ortho = None

  def host_norm(v):
    t = torch.linalg.vector_norm(v).reshape(1)
    if sync is not None:
      sync(t)
    return float(t)

  bnorm = host_norm(b)
  y = torch.zeros(n, dtype=torch.float64, device=dev)
  if x0 is not None:
    b = b - (yield x0)
    beta = host_norm(b)
  else:
    beta = bnorm
  r = b.clone()
  its = 0
  res = beta / bnorm if bnorm > 0 else 0.0
  history = []
  wasted = 0
  host_cols = ws.host_cols if ws is not None else (_pinned_columns(restart + 1, restart + 2) if lag else None)
  events = [torch.cuda.Event(), torch.cuda.Event()] if lag else None
  # the stream the iterations are enqueued on: looked up once (a solve does not change streams; the lookup is 4 us of the
  # ~45 us of host time an iteration of a small deck costs)
  ev_stream = torch.cuda.current_stream(dev) if lag else None
  try:
    while its < maxiter and res > tol:
      m = min(restart, maxiter - its)
      if ws is not None:
        V, cols = ws.V, ws.cols
      else:
        V = torch.empty((m + 1, n), dtype=torch.float64, device=dev)
        cols = torch.empty((m, m + 2), dtype=torch.float64, device=dev)    # row j = column j of H, then |w_j|
      V[0] = r / beta
      H = np.zeros((m + 1, m))
      cs, sn = [0.0] * m, [0.0] * m          # plain Python floats: the rotations below are a scalar recurrence, and numpy
      g = [0.0] * (m + 1)                    # scalars cost ~10x a float operation (it is host time between two sweeps)
      g[0] = beta
      k_used = 0
      prev_res = None

      def finish(j):
        """Host side of iteration j: read its column, rotate, test.  True = stop after this column."""
        nonlocal its, k_used, res, prev_res
        if lag:
          events[j & 1].synchronize()
          col = host_cols[j, :j + 2].tolist()
        else:
          col = cols[j, :j + 2].tolist()                              # the one host transfer of the iteration
        w_norm = col[-1]
        last_norm[0] = w_norm
        for i in range(j):                                           # previous rotations
          t = cs[i] * col[i] + sn[i] * col[i + 1]
          col[i + 1] = -sn[i] * col[i] + cs[i] * col[i + 1]
          col[i] = t
        d = math.hypot(col[j], col[j + 1])
        cs[j], sn[j] = (col[j] / d, col[j + 1] / d) if d > 0 else (1.0, 0.0)
        col[j] = d
        col[j + 1] = 0.0
        H[:j + 2, j] = col
        g[j + 1] = -sn[j] * g[j]
        g[j] = cs[j] * g[j]
        its += 1
        k_used = j + 1
        prev_res, res = res, abs(g[j + 1]) / bnorm
        history.append(res)
        return res <= tol or w_norm == 0 or not math.isfinite(w_norm)

      def may_defer():
        """Whether the pending column can wait until the next iteration has been enqueued: not when the last
        observed reduction rate says it may already meet the tolerance."""
        rate = min(1.0, res / prev_res) if prev_res else 1.0
        return res * rate > 20.0 * tol

      def orthogonalise(j, w):
        """Two passes of classical Gram-Schmidt against V[0..j]; the new Hessenberg column and |w| go to cols[j]."""
        Vj = V[:j + 1]
        h = Vj @ w
        w = torch.addmv(w, Vj.t(), h, alpha=-1.0)
        h2 = Vj @ w
        w = torch.addmv(w, Vj.t(), h2, alpha=-1.0)
        torch.add(h, h2, out=cols[j, :j + 1])
        torch.linalg.vector_norm(w, out=cols[j, j + 1])
        return w

      pending, stop, last_norm = None, False, [0.0]
      for j in range(m):
        if pending is not None and not may_defer():
          stop, pending = finish(pending), None
          if stop:
            break
        if ws is not None:
          def device_side(j=j):
            w = A(Minv(V[j]))
            mapped = ws.col_mapped_ptr(j) if ortho is not None else 0
            if ortho is not None:
              if mapped:
                ortho(V, j + 1, w, cols[j], V[j + 1], mapped)       # the kernel stores the column into host memory itself
              else:
                ortho(V, j + 1, w, cols[j], V[j + 1])
            else:
              torch.div(orthogonalise(j, w), cols[j, j + 1], out=V[j + 1])
            if not mapped:
              host_cols[j, :j + 2].copy_(cols[j, :j + 2], non_blocking=True)
          ws.run(j, device_side, on_replay)
        else:
          w = yield Minv(V[j])
          if ortho is not None:                                      # both passes, column, |w| and V[j + 1] in four launches
            ortho(V, j + 1, w if w.is_contiguous() else w.contiguous(), cols[j], V[j + 1])
          else:
            w = orthogonalise(j, w)
          if sync is not None:                                       # multi-rank: all ranks act on rank 0's numbers
            sync(cols[j, :j + 2])
        if lag:
          if ws is None:
            if ortho is None:
              torch.div(w, cols[j, j + 1], out=V[j + 1])             # normalised on the device: no host value needed
            host_cols[j, :j + 2].copy_(cols[j, :j + 2], non_blocking=True)
          # fence on the stream the copy was enqueued on: the current stream of the VECTORS' device, which need not
          # be the process's current device (a suspension built on cuda:1 while cuda:0 is current)
          events[j & 1].record(ev_stream)
          if pending is not None:
            stop, pending = finish(pending), None
            if stop:
              wasted += 1                                            # iteration j was enqueued for nothing
              break
          pending = j
        else:
          stop = finish(j)
          if last_norm[0] > 0:
            torch.mul(w, 1.0 / last_norm[0], out=V[j + 1])
          if stop:
            break
      if pending is not None and not stop:
        finish(pending)
      coef = np.linalg.solve(np.triu(H[:k_used, :k_used]), np.array(g[:k_used])) if k_used > 0 else np.zeros(0)
      y = y + V[:k_used].t() @ torch.as_tensor(coef, device=dev)
      if res > tol and its < maxiter:                                # restart: true residual
        r = b - (yield Minv(y))
        beta = host_norm(r)
        res = beta / bnorm
  finally:
    if host_cols is not None and ws is None:
      _pinned_pool.append(host_cols)
  x = Minv(y)
  if x0 is not None:
    x = x + x0
  return x, dict(iterations=its, residual=res, converged=bool(res <= tol), history=history, discarded_sweeps=wasted)


def gmres_right_preconditioned(A, Minv, b, tol=1e-8, restart=60, maxiter=1000, x0=None, sync=None, lag=None, ws=None,
                               on_replay=None, ortho=None):
  """Solve A x = b with x = x0 + Minv y, GMRES(restart) on A.Minv (general_application_utils.py:608-627).
  Stops when |b - A x| <= tol |b| (scipy `tol`, atol = 0) or after `maxiter` INNER iterations in total -- not restart
  cycles: scipy (and the reference's call, maxiter=1000 with restart=60) counts cycles, i.e. up to 60 000 inner
  iterations; the solves here converge in tens of iterations, so the cap only differs in how soon a diverging solve
  gives up.
  Arnoldi with two passes of classical Gram-Schmidt (one device GEMV each); Givens rotations on the host, on a GPU one
  iteration behind the device (`lag`, see _gmres_steps; None = on for CUDA tensors).
  x0: optional initial guess (the roller torque solve warm-starts from the previous step,
  quaternion_integrator_rollers.py:961); the Krylov space is then built on the residual b - A x0."""
  steps = _gmres_steps(Minv, b, tol, restart, maxiter, x0, sync, lag, ws=ws, A=A if ws is not None else None, on_replay=on_replay,
                       ortho=ortho)
  try:
    request = next(steps)
    while True:
      request = steps.send(A(request))
  except StopIteration as done:
    return done.value


def gmres_pair_right_preconditioned(A, A2, Minv, b_a, b_b, tol=1e-8, restart=60, maxiter=1000, sync=None, ortho=None):
  """Two independent solves A x_a = b_a, A x_b = b_b advanced in lockstep: while both are running, each iteration
  hands its two operator requests to A2(u, v) -> (A u, A v) -- one pass over the blob pairs with two vectors
  (rmb_matvec2_device) instead of two.  Every solve sees exactly the iterates it would see alone.
  Returns ((x_a, info_a), (x_b, info_b))."""
  gens = [_gmres_steps(Minv, b, tol, restart, maxiter, None, sync, ortho=ortho) for b in (b_a, b_b)]
  requests, results = [None, None], [None, None]
  for k in (0, 1):
    try:
      requests[k] = next(gens[k])
    except StopIteration as done:
      results[k] = done.value
  while results[0] is None or results[1] is None:
    if results[0] is None and results[1] is None:
      answers = A2(requests[0], requests[1])
    else:
      k = 0 if results[0] is None else 1
      answers = [None, None]
      answers[k] = A(requests[k])
    for k in (0, 1):
      if results[k] is None:
        try:
          requests[k] = gens[k].send(answers[k])
        except StopIteration as done:
          results[k] = done.value
  return results[0], results[1]
