"""Single-process multi-device context: the whole node behind one call.

The reference's callers are one Python process calling a module-level function (multi_bodies/multi_bodies.py:233-287
selects it, :445 / :599 call it), so the sharding has to sit behind that call: `MultiContext(devices)` wraps the C
engine `rmb_multi_*` (include/rmb_mobility.h, csrc/rmb_multi.hip) -- one shard context + stream per listed device, pair
shard g of G on device g, fixed-order slice reduction through peer-mapped reads -- and has the shape of
`MobilityContext` (set_positions / matvec / matvec_device / matvec_op_device / matvec2_device / blob_blob_force...), so
`mobility.py`, `forces.py` and the device-resident callers (RigidSuspension, Lanczos, the steppers) take either.
Vectors and results of the device entry points live on devices[0].  The reference has no counterpart (single device).
"""
import ctypes

import numpy as np

from . import _lib
from .context import MobilityContext, _as_f64, _is_torch_cuda, _ptr


class MultiContext(object):
  """G pair shards on G devices in one process; the same device may be listed several times (one-GPU rehearsal)."""

  def __init__(self, devices):
    self._lib = _lib.load()
    devices = [int(d) for d in devices]
    if not devices:
      raise ValueError("MultiContext needs at least one device")
    arr = (ctypes.c_int * len(devices))(*devices)
    h = ctypes.c_void_p()
    _lib.check(self._lib.rmb_multi_create(arr, len(devices), ctypes.byref(h)))
    self._h = h
    self.devices = devices
    self.device = devices[0]          # where the vectors of the device entry points live
    self.n = 0
    self.n_targets = 0
    self.target_range = (0, 0)
    self._keepalive = None
    self._stream_handle = None
    self._user_stream = False
    self._aux = None                  # plain context on devices[0] for what the engine does not shard
    self._aux_args = None
    self._aux_stale = True

  def close(self):
    if getattr(self, "_aux", None) is not None:
      self._aux.close()
      self._aux = None
    if getattr(self, "_h", None) is not None and self._h.value:
      self._lib.rmb_multi_destroy(self._h)
      self._h = ctypes.c_void_p()

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass

  @property
  def n_shards(self):
    return int(self._lib.rmb_multi_n_shards(self._h))

  # --- configuration ---------------------------------------------------------------------------
  def set_stream(self, stream_ptr):
    """Pin the engine's primary stream (on devices[0]); otherwise device calls follow torch's current stream there."""
    _lib.check(self._lib.rmb_multi_set_stream(self._h, ctypes.c_void_p(int(stream_ptr) if stream_ptr else 0)))
    self._user_stream = True

  def release_stream(self):
    """The engine never touches a previous primary stream again (rmb_multi_set_stream), so releasing is bookkeeping:
    wait for the engine's work and follow torch's current stream from the next call on."""
    self.synchronize()
    self._stream_handle = None
    self._user_stream = False

  def _follow_torch_stream(self):
    if self._user_stream:
      return
    from .context import _current_raw_stream
    h = _current_raw_stream(self.device)
    if h != self._stream_handle:
      _lib.check(self._lib.rmb_multi_set_stream(self._h, ctypes.c_void_p(h)))
      self._stream_handle = h

  def set_option(self, key, value):
    """"reduce" (0 fixed-order slices, 1 RCCL) is the engine's own; every other key goes to all shard contexts."""
    _lib.check(self._lib.rmb_multi_set_option(self._h, key.encode(), int(value)))
    if self._aux is not None and key != "reduce":
      self._aux.set_option(key, value)

  def get_option(self, key):
    v = ctypes.c_long()
    _lib.check(self._lib.rmb_multi_get_option(self._h, key.encode(), ctypes.byref(v)))
    return int(v.value)

  def shard_context(self, shard):
    """Borrowed handle of one shard's context (timing, last launch): not to be closed, not to be given a stream."""
    h = ctypes.c_void_p()
    _lib.check(self._lib.rmb_multi_shard_ctx(self._h, int(shard), ctypes.byref(h)))
    return h

  def set_positions(self, r_vectors, a, periodic_length=None, wall=True):
    L = _as_f64(np.zeros(3) if periodic_length is None else periodic_length, 3)
    if _is_torch_cuda(r_vectors):
      r = r_vectors.contiguous().view(-1)
      if r.device.index != self.device:
        raise ValueError("device-resident positions must live on devices[0] = cuda:%d" % self.device)
      n = r.numel() // 3
      self._follow_torch_stream()
      _lib.check(self._lib.rmb_multi_set_positions_device(self._h, ctypes.c_void_p(r.data_ptr()), n, float(a), _ptr(L),
                                                          int(bool(wall))))
      self._keepalive = r
    else:
      r = _as_f64(r_vectors)
      n = r.size // 3
      _lib.check(self._lib.rmb_multi_set_positions(self._h, _ptr(r), n, float(a), _ptr(L), int(bool(wall))))
    self.n = n
    self.n_targets = n
    self.target_range = (0, n)
    self._aux_args = (r_vectors if _is_torch_cuda(r_vectors) else r.copy(), float(a), L.copy(), bool(wall))
    self._aux_stale = True

  def set_target_range(self, begin, end):
    if (int(begin), int(end)) != (0, self.n):
      raise ValueError("the multi-device engine always produces all n targets (it shards pairs, not targets)")

  @property
  def helper_context(self):
    """A plain context on devices[0] for the rank-local O(N) helper kernels of the callers (rigid.py: block products,
    fused Gram-Schmidt, per-body geometry and factors).  They need no positions and run on torch's current stream of
    devices[0], where the callers' vectors live and against which the engine orders every product."""
    if self._aux is None:
      self._aux = MobilityContext(self.device)
    return self._aux

  def _auxiliary(self):
    """Plain context on devices[0] with the same configuration, for the O(n) / per-body work the engine does not shard."""
    if self._aux is None:
      self._aux = MobilityContext(self.device)
    if self._aux_stale:
      r, a, L, wall = self._aux_args
      self._aux.set_positions(r, a, L, wall)
      self._aux_stale = False
    return self._aux

  # --- products --------------------------------------------------------------------------------
  def matvec(self, kind, vec, eta, vec2=None, in_plane=False):
    """Host path: numpy in, new numpy (3n,) out; synchronous."""
    k = _lib.KINDS[kind] if isinstance(kind, str) else int(kind)
    v = _as_f64(vec, 3 * self.n)
    v2 = _as_f64(vec2, 3 * self.n) if vec2 is not None else None
    out = np.empty(3 * self.n)
    _lib.check(self._lib.rmb_multi_matvec(self._h, k, int(bool(in_plane)), _ptr(v), _ptr(v2) if v2 is not None else None,
                                          float(eta), _ptr(out)))
    return out

  def _check_vec(self, v, what="vec"):
    if not _is_torch_cuda(v) or v.numel() != 3 * self.n or not v.is_contiguous() or v.device.index != self.device:
      raise ValueError("%s must be a contiguous CUDA float64 tensor with 3*n entries on cuda:%d" % (what, self.device))

  def matvec_device(self, kind, vec, eta, vec2=None, in_plane=False, out=None):
    import torch
    k = _lib.KINDS[kind] if isinstance(kind, str) else int(kind)
    self._check_vec(vec)
    if vec2 is not None:
      self._check_vec(vec2, "vec2")
    if out is None:
      out = torch.empty(3 * self.n, dtype=torch.float64, device=vec.device)
    else:
      self._check_vec(out, "out")
    self._follow_torch_stream()
    _lib.check(self._lib.rmb_multi_matvec_device(self._h, k, int(bool(in_plane)), ctypes.c_void_p(vec.data_ptr()),
                                                 ctypes.c_void_p(vec2.data_ptr()) if vec2 is not None else None,
                                                 float(eta), ctypes.c_void_p(out.data_ptr())))
    return out

  def matvec_op_device(self, op, vecs, eta, in_plane=False, outs=None, shard=0, nshards=1):
    import torch
    if nshards != 1:
      raise ValueError("the multi-device engine shards internally: nshards must be 1")
    code, n_in, n_out = _lib.OPS[op]
    if n_in is None:
      n_in = n_out = len(vecs)
    if len(vecs) != n_in:
      raise ValueError("%s takes %d input vectors" % (op, n_in))
    for v in vecs:
      self._check_vec(v, "vectors")
    if outs is None:
      outs = [torch.empty(3 * self.n, dtype=torch.float64, device=vecs[0].device) for _ in range(n_out)]
    outs = list(outs)
    if len(outs) != n_out:
      raise ValueError("%s produces %d output vectors" % (op, n_out))
    for o in outs:
      self._check_vec(o, "outputs")
    ins_p = (ctypes.c_void_p * n_in)(*[v.data_ptr() for v in vecs])
    outs_p = (ctypes.c_void_p * n_out)(*[o.data_ptr() for o in outs])
    self._follow_torch_stream()
    _lib.check(self._lib.rmb_multi_matvec_op_device(self._h, code, int(bool(in_plane)), n_in,
                                                    ctypes.cast(ins_p, ctypes.c_void_p), n_out,
                                                    ctypes.cast(outs_p, ctypes.c_void_p), float(eta)))
    return tuple(outs)

  def matvec2_device(self, kind, vec_a, vec_b, eta, out_a=None, out_b=None, shard=0, nshards=1):
    k = kind if isinstance(kind, str) else {v: n for n, v in _lib.KINDS.items()}[int(kind)]
    outs = None if out_a is None and out_b is None else [out_a, out_b]
    if outs is not None and (out_a is None or out_b is None):
      import torch
      outs = [o if o is not None else torch.empty(3 * self.n, dtype=torch.float64, device=vec_a.device) for o in outs]
    r = self.matvec_op_device(k + "_multi", (vec_a, vec_b), eta, outs=outs, nshards=nshards)
    return r[0], r[1]

  def body_mobility_dense_device(self, first_blob, n_b, eta, out=None):
    """O(n_bodies n_b^2): not worth sharding; runs on devices[0]."""
    return self._auxiliary().body_mobility_dense_device(first_blob, n_b, eta, out=out)

  def blob_blob_force(self, repulsion_strength, debye_length, blob_radius):
    out = np.empty(3 * self.n)
    _lib.check(self._lib.rmb_multi_blob_blob_force(self._h, float(repulsion_strength), float(debye_length),
                                                   float(blob_radius), _ptr(out)))
    return out.reshape(self.n, 3)

  def blob_blob_force_device(self, repulsion_strength, debye_length, blob_radius, out=None, device=None):
    import torch
    if out is None:
      out = torch.empty(3 * self.n, dtype=torch.float64, device=device or ("cuda:%d" % self.device))
    else:
      self._check_vec(out, "out")
    self._follow_torch_stream()
    _lib.check(self._lib.rmb_multi_blob_blob_force_device(self._h, float(repulsion_strength), float(debye_length),
                                                          float(blob_radius), ctypes.c_void_p(out.data_ptr())))
    return out

  def blob_blob_force_radii(self, radius_blobs, repulsion_strength, debye_length):
    return self._auxiliary().blob_blob_force_radii(radius_blobs, repulsion_strength, debye_length)

  def blob_blob_force_radii_device(self, radius_blobs, repulsion_strength, debye_length, out=None):
    return self._auxiliary().blob_blob_force_radii_device(radius_blobs, repulsion_strength, debye_length, out=out)

  # --- measurement -----------------------------------------------------------------------------
  def timing_collect(self, max_n=8192, shard=0):
    """Sampled sweep durations (ms) of ONE shard's context (option "timing" is forwarded to all of them)."""
    buf = (ctypes.c_double * max_n)()
    n = self._lib.rmb_timing_collect(self.shard_context(shard), buf, max_n)
    if n < 0:
      _lib.check(n)
    return np.array(buf[:n])

  def timing_reset(self):
    for g in range(self.n_shards):
      _lib.check(self._lib.rmb_timing_reset(self.shard_context(g)))

  def last_launch(self, shard=0):
    t, c, w = ctypes.c_long(), ctypes.c_long(), ctypes.c_long()
    _lib.check(self._lib.rmb_last_launch(self.shard_context(shard), ctypes.byref(t), ctypes.byref(c), ctypes.byref(w)))
    return dict(tiles=t.value, chunks=c.value, workgroups=w.value)

  def ubench_fp64_issue(self, launches=40):
    return self._auxiliary().ubench_fp64_issue(launches)

  def synchronize(self):
    _lib.check(self._lib.rmb_multi_synchronize(self._h))
