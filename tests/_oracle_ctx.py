"""CPU stand-in for MobilityContext backed by the oracle -- TEST infrastructure only (lets the host logic of
the callers run here without a GPU; the product never imports it)."""
import numpy as np
import torch


class OracleContext(object):
  def __init__(self, oracle):
    self.o = oracle
    self.n_set_positions = 0

  def set_stream(self, s):
    pass

  def set_option(self, key, value):      # remembered, without effect on the CPU stand-in
    self.options = dict(getattr(self, "options", {}), **{key: int(value)})

  def get_option(self, key):
    return getattr(self, "options", {}).get(key, {"precision": 64}.get(key, 0))

  def set_positions(self, r, a, L=None, wall=True):
    self.r = (r.detach().cpu().numpy() if isinstance(r, torch.Tensor) else np.asarray(r)).reshape(-1, 3).copy()
    self.a, self.wall = float(a), bool(wall)
    self.L = np.zeros(3) if L is None else np.asarray(L, dtype=np.float64)
    self.n = len(self.r)
    self.n_set_positions += 1

  def _wrapped(self, kind, v, eta, in_plane):
    return self.o._wrapped(kind, int(self.wall), self.r, v.detach().cpu().numpy(), eta, self.a, in_plane=in_plane,
                           periodic_length=self.L)

  def matvec_device(self, kind, vec, eta, vec2=None, in_plane=False, out=None):
    if kind == "tt_tr":
      u = self._wrapped("tt", vec, eta, in_plane) + self._wrapped("tr", vec2, eta, in_plane)
    else:
      u = self._wrapped(kind, vec, eta, in_plane)
    return torch.from_numpy(u)

  def matvec_op_device(self, op, vecs, eta, in_plane=False, outs=None, shard=0, nshards=1):
    mv = lambda kind, v, v2=None: self.matvec_device(kind, v, eta, vec2=v2, in_plane=in_plane)
    if op == "velocity_from_force_torque":
      res = (mv("tt_tr", vecs[0], vecs[1]),)
    elif op == "grand":
      res = (mv("tt_tr", vecs[0], vecs[1]), mv("rt", vecs[0]) + mv("rr", vecs[1]))
    elif op == "force_column":
      res = (mv("tt", vecs[0]), mv("rt", vecs[0]))
    else:
      res = tuple(mv(op[:2], v) for v in vecs)      # "tt_multi", "tr_multi", "rt_multi", "rr_multi"
    if outs is not None:
      for o, r in zip(outs, res):
        o.copy_(r)
      return tuple(outs)
    return res

  def matvec2_device(self, kind, vec_a, vec_b, eta, out_a=None, out_b=None, shard=0, nshards=1):
    return self.matvec_device(kind, vec_a, eta), self.matvec_device(kind, vec_b, eta)

  def blob_blob_force_device(self, eps, b, a, out=None, device=None):
    F = self.o.calc_blob_blob_forces_oracle(self.r, periodic_length=self.L, repulsion_strength=eps, debye_length=b,
                                            blob_radius=a)
    return torch.from_numpy(np.ascontiguousarray(F).reshape(-1))

  def body_mobility_dense_device(self, first_blob, n_b, eta, out=None):
    """Per-body dense blocks with the same wall regularisation as the products (height clamp + B on both sides), which
    is what body_dense_tt_kernel builds; equal to the reference's unclamped single_wall_fluid_mobility whenever every
    blob sits above z = a."""
    blocks = []
    for f in first_blob.tolist():
      rk = self.r[f:f + n_b]
      if self.wall:
        r_eff, b, _ = self.o.wall_regularisation(rk, self.a)
        B = np.repeat(b, 3)
        blocks.append(B[:, None] * self.o.dense("tt", 1, r_eff, eta, self.a) * B[None, :])
      else:
        blocks.append(self.o.dense("tt", 0, rk, eta, self.a))
    return torch.from_numpy(np.array(blocks))

  def close(self):
    pass
