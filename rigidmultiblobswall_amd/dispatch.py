"""String dispatch of the blob-level backends.

The reference selects implementations by strings from the input deck:
  mobility_vector_prod_implementation -> multi_bodies/multi_bodies.py:233-287
  blob_blob_force_implementation      -> multi_bodies/multi_bodies_functions.py:249-278
  mobility_blobs_implementation       -> multi_bodies/multi_bodies.py:207-230
These functions add the `hip*` strings and return callables with the reference's signatures.  With
`accept_reference_gpu_names=True` the reference's own GPU strings (`pycuda`, `pycuda_no_wall`) are
served by the HIP engine too, so an existing deck runs unchanged.
"""
import numpy as np

from . import forces as _forces
from . import mobility as _mob


def _radius_blobs(kwargs):
  """Per-blob radii for the `radii_*` modes: `radius_blobs=` directly, or concatenated from `bodies=` objects that
  carry `.blobs_radius` (multi_bodies.py:279-284)."""
  if kwargs.get("radius_blobs") is not None:
    return np.asarray(kwargs["radius_blobs"], dtype=np.float64).reshape(-1)
  bodies = kwargs.get("bodies")
  if bodies is None:
    raise ValueError("radii_* implementations need radius_blobs= or bodies= (with .blobs_radius)")
  return np.concatenate([np.asarray(b.blobs_radius, dtype=np.float64).reshape(-1) for b in bodies])


def set_mobility_vector_prod(implementation, accept_reference_gpu_names=False, *args, **kwargs):
  table = {
      "hip": _mob.single_wall_mobility_trans_times_force_hip,
      "hip_no_wall": _mob.no_wall_mobility_trans_times_force_hip,
      "hip_in_plane": _mob.in_plane_mobility_trans_times_force_hip,
      "hip_free_surface": _mob.free_surface_mobility_trans_times_force_hip,
  }
  radii = {"radii_hip": _mob.single_wall_mobility_trans_times_force_source_target_hip,
           "radii_hip_no_wall": _mob.no_wall_mobility_trans_times_force_source_target_hip,
           "radii_hip_free_surface": _mob.free_surface_mobility_trans_times_force_source_target_hip}
  if accept_reference_gpu_names:
    table["pycuda"] = table["hip"]
    table["pycuda_no_wall"] = table["hip_no_wall"]
    table["pycuda_free_surface"] = table["hip_free_surface"]
    radii["radii_pycuda"] = radii["radii_hip"]
  if implementation in radii:
    # blobs of different radii: sources == targets through the source->target kernel (multi_bodies.py:266-286)
    from functools import partial
    return partial(_mob.mobility_radii_trans_times_force, radius_blobs=_radius_blobs(kwargs), function=radii[implementation])
  if implementation not in table:
    raise ValueError("mobility_vector_prod_implementation %r is not served by the HIP engine (known: %s)" %
                     (implementation, ", ".join(sorted(table))))
  return table[implementation]


def _zero_forces(r_vectors, *args, **kwargs):
  return np.zeros((np.size(r_vectors) // 3, 3))


def set_blob_blob_forces(implementation, accept_reference_gpu_names=False, *args, **kwargs):
  table = {"None": _zero_forces, "hip": _forces.calc_blob_blob_forces_hip, "tree_hip": _forces.calc_blob_blob_forces_tree_hip}
  if accept_reference_gpu_names:
    table["pycuda"] = table["hip"]
  if implementation == "radii_hip":
    # one radius per blob (multi_bodies_functions.py:270-277)
    from functools import partial
    return partial(_forces.calc_blob_blob_forces_radii_hip, radius_blobs=_radius_blobs(kwargs))
  if implementation not in table:
    raise ValueError("blob_blob_force_implementation %r is not served by the HIP engine" % (implementation,))
  return table[implementation]


def set_mobility_blobs(implementation, *args, **kwargs):
  """Dense (3N x 3N) blob mobility builders f(r_vectors, eta, a) (mobility.py:1018-1116, :967-1013)."""
  table = {"hip": _mob.single_wall_fluid_mobility_hip, "hip_no_wall": _mob.rotne_prager_tensor_hip}
  if implementation not in table:
    raise ValueError("mobility_blobs_implementation %r is not served by the HIP engine" % (implementation,))
  return table[implementation]
