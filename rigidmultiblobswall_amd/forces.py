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


def _context():
  global _ctx
  if _ctx is None:
    _ctx = MobilityContext(0)
  return _ctx


def calc_blob_blob_forces_hip(r_vectors, *args, **kwargs):
  L = kwargs.get('periodic_length')
  eps = kwargs.get('repulsion_strength')
  b = kwargs.get('debye_length')
  a = kwargs.get('blob_radius')
  if L is None:
    L = np.zeros(3)
  ctx = _context()
  # wall=False: raw positions, no height clamp (the reference passes r_vectors untouched)
  ctx.set_positions(r_vectors, a, L, wall=False)
  return ctx.blob_blob_force(eps, b, a)


def calc_blob_blob_forces_radii_hip(r_vectors, radius_blobs, *args, **kwargs):
  """`radii_numba` twin (multi_bodies/forces_numba.py:125-137): every blob has its own radius and two blobs touch at
  r = a_i + a_j.  Same keyword arguments as above (`blob_radius` is ignored, as in the reference kernel)."""
  L = kwargs.get('periodic_length')
  if L is None:
    L = np.zeros(3)
  ctx = _context()
  ctx.set_positions(r_vectors, 1.0, L, wall=False)
  return ctx.blob_blob_force_radii(radius_blobs, kwargs.get('repulsion_strength'), kwargs.get('debye_length'))
