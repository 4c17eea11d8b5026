"""What a reference input deck asks for, mapped onto what this engine runs -- or a ValueError.

The reference picks its backends by strings (multi_bodies/multi_bodies.py:207-287 `set_mobility_blobs`,
`set_mobility_vector_prod`; multi_bodies_functions.py:249-278 `set_blob_blob_forces`, :348-356
`set_body_body_forces_torques`).  The backend name itself (python / C++ / numba / pycuda / hip) does not change the
physics and is ignored; the SUFFIX does (`_no_wall`, `_free_surface`, `radii_*`), and so do per-blob radii in a
4-column .vertex file.  A deck that asks for a mode the time steppers here do not run must fail loudly: silently
running it with a wall, a uniform radius or without the body-body forces would be different physics.
"""
import numpy as np

_BACKENDS = ("python", "C++", "numba", "pycuda", "hip", "cpp")


def _split(impl):
  """-> (backend, suffix) for strings like 'pycuda_no_wall', 'numba', 'C++_free_surface'."""
  for b in sorted(_BACKENDS, key=len, reverse=True):
    if impl == b:
      return b, ""
    if impl.startswith(b + "_"):
      return b, impl[len(b) + 1:]
  return None, impl


def hydrodynamic_mode(impl, option):
  """'single_wall' | 'no_wall' from a mobility implementation string; ValueError for anything else."""
  if "radii" in impl:
    raise ValueError("%s %r: per-blob radii are served by the source-target products (dispatch.set_mobility_vector_prod), "
                     "not by the time steppers, which assume one blob radius" % (option, impl))
  backend, suffix = _split(impl)
  if backend is None:
    raise ValueError("%s %r: unknown implementation string" % (option, impl))
  if suffix == "":
    return "single_wall"
  if suffix == "no_wall":
    return "no_wall"
  if suffix == "free_surface":
    raise ValueError("%s %r: the time steppers do not run above a free surface (the product itself exists: "
                     "free_surface_mobility_trans_times_force_hip)" % (option, impl))
  raise ValueError("%s %r: unknown implementation suffix %r" % (option, impl, suffix))


def validate(read, uses_dense_blocks=True):
  """Checks every implementation option of a ReadInput deck against `domain`; returns the hydrodynamic mode
  ('single_wall', 'no_wall' or 'in_plane' as given by `domain`)."""
  domain = read.domain
  if domain not in ("single_wall", "no_wall", "in_plane"):
    raise ValueError("domain %r: expected single_wall, no_wall or in_plane" % (domain,))
  modes = [("mobility_vector_prod_implementation", hydrodynamic_mode(read.mobility_vector_prod_implementation,
                                                                      "mobility_vector_prod_implementation"))]
  if uses_dense_blocks:
    modes.append(("mobility_blobs_implementation", hydrodynamic_mode(read.mobility_blobs_implementation,
                                                                     "mobility_blobs_implementation")))
  want = "no_wall" if domain == "no_wall" else "single_wall"
  for option, mode in modes:
    if mode != want:
      raise ValueError("%s %r is a %s implementation but the deck says `domain %s`: the reference would mix an unbounded "
                       "mobility with wall checks (or the reverse); state the intended one"
                       % (option, getattr(read, option), mode, domain))
  ff = read.blob_blob_force_implementation
  if "radii" in ff:
    raise ValueError("blob_blob_force_implementation %r needs per-blob radii; the time steppers assume one blob radius "
                     "(dispatch.set_blob_blob_forces serves the product)" % (ff,))
  if ff != "None" and _split(ff)[0] is None and ff != "tree_numba":
    raise ValueError("blob_blob_force_implementation %r: unknown implementation string" % (ff,))
  bb = read.body_body_force_torque_implementation
  if bb != "None":
    raise ValueError("body_body_force_torque_implementation %r: body-body forces (multi_bodies_functions.py:359-395, a Yukawa "
                     "potential between body centres) are not built; only `None`" % (bb,))
  return domain


def uniform_vertices(coor, blob_radius, path):
  """(n,3) coordinates of a .vertex array; a 4th column (per-blob radius) must equal the deck's blob_radius."""
  coor = np.asarray(coor, dtype=np.float64)
  if coor.shape[1] > 3:
    rad = coor[:, 3]
    if not np.allclose(rad, blob_radius, rtol=1e-12, atol=0.0):
      raise ValueError("%s lists per-blob radii (%.6g .. %.6g) that differ from blob_radius %.6g: blobs of different "
                       "radii (the reference's radii_* modes) are not run by the time steppers"
                       % (path, rad.min(), rad.max(), blob_radius))
  return coor[:, :3]
