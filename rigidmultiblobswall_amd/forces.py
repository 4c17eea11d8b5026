"""Blob-blob pair forces -- the `calc_blob_blob_forces_<impl>` surface on MI355X.

Replaces multi_bodies/forces_pycuda.py:148-180 (float32 CUDA) and its CPU twins
(multi_bodies/forces_numba.py:58-71) behind the same call:
  calc_blob_blob_forces_hip(r_vectors, periodic_length=L, repulsion_strength=eps,
                            debye_length=b, blob_radius=a) -> ndarray (N,3)
selected in the reference by `blob_blob_force_implementation`
(multi_bodies/multi_bodies_functions.py:249-278).  fp64 here; minimal image only.
"""
import numpy as np

from .context import MobilityContext

_ctx = None
_mctx = None


def _context(n=0):
  """Own contexts (the forces use raw positions, the products clamped ones: sharing would re-pack on every call), on
  the devices mobility.py is configured with (mobility.set_devices / RMB_DEVICES / RMB_DEVICE)."""
  global _ctx, _mctx
  from . import mobility
  devs = mobility.devices()
  if len(devs) > 1 and n >= mobility.multi_min_blobs:
    if _mctx is None or _mctx.devices != devs:
      if _mctx is not None:
        _mctx.close()
      from .multi import MultiContext
      _mctx = MultiContext(devs)
    return _mctx
  if _ctx is None or _ctx.device != devs[0]:
    if _ctx is not None:
      _ctx.close()
    _ctx = MobilityContext(devs[0])
  return _ctx


def reset():
  """Drop the module-level contexts (frees device memory)."""
  global _ctx, _mctx
  for c in (_ctx, _mctx):
    if c is not None:
      c.close()
  _ctx = _mctx = None


def calc_blob_blob_forces_hip(r_vectors, *args, **kwargs):
  L = kwargs.get('periodic_length')
  eps = kwargs.get('repulsion_strength')
  b = kwargs.get('debye_length')
  a = kwargs.get('blob_radius')
  if L is None:
    L = np.zeros(3)
  ctx = _context(np.asarray(r_vectors).size // 3)
  # wall=False: raw positions, no height clamp (the reference passes r_vectors untouched)
  ctx.set_positions(r_vectors, a, L, wall=False)
  return ctx.blob_blob_force(eps, b, a)


def calc_blob_blob_forces_tree_hip(r_vectors, *args, **kwargs):
  """`tree_numba` twin (multi_bodies/forces_numba.py:142-271): the reference builds a k-d tree and keeps the pairs within
  2 a + 30 b.  The force kernel here skips whole tile pairs whose bounding boxes are further apart than the exponential
  reaches in double precision (after a device Morton sort of the blobs), which changes no bit of the full sum -- so this
  is `calc_blob_blob_forces_hip`, and it differs from the reference's truncated sum by e^-30 of a contact force per
  dropped pair (4e-14 relative on the goldens)."""
  return calc_blob_blob_forces_hip(r_vectors, *args, **kwargs)


def calc_blob_blob_forces_radii_hip(r_vectors, radius_blobs, *args, **kwargs):
  """`radii_numba` twin (multi_bodies/forces_numba.py:125-137): every blob has its own radius and two blobs touch at
  r = a_i + a_j.  Same keyword arguments as above (`blob_radius` is ignored, as in the reference kernel)."""
  L = kwargs.get('periodic_length')
  if L is None:
    L = np.zeros(3)
  ctx = _context(0)      # per-blob radii: one device
  ctx.set_positions(r_vectors, 1.0, L, wall=False)
  return ctx.blob_blob_force_radii(radius_blobs, kwargs.get('repulsion_strength'), kwargs.get('debye_length'))
