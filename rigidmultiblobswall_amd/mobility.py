"""Blob-level mobility products -- the `mobility/mobility.py` plugin surface on MI355X.

Same names, argument meaning and return layout as the reference's per-backend wrappers
(`X_mobility_{trans,rot}_times_{force,torque}_<impl>(r_vectors, vec, eta, a, *args, **kwargs)
-> ndarray(3N)`, mobility/mobility.py:187-615 and :1119-1341) with `<impl> = hip`.  A maintainer
adds one `elif` per string in multi_bodies/multi_bodies.py:233-287.

Differences in HOW, not WHAT:
  * the height clamp and the B-damping (shift_heights / damping_matrix_B, mobility.py:52-84; the
    latter an interpreted loop over N per call) run on the device, fused into position packing,
    source staging and the output store;
  * positions are uploaded once and stay resident while the caller keeps passing the same array
    contents (GMRES and Lanczos call with fixed r_vectors: multi_bodies.py:445, :599);
  * only `periodic_length` is read from kwargs; any other kwarg / positional extra is ignored, as
    in the reference (callers pass step=, update_PC= ...).
There is no CPU fallback: without the HIP library or a GPU these functions raise.
"""
import os

import numpy as np
import scipy.sparse

from . import _lib
from .context import MobilityContext

_ctx = None       # one-device context on devices()[0]
_mctx = None      # multi-device engine over devices() (created when it is first needed)
_cached = {}      # id of the bound context -> (r copy, a, L tuple, wall)
_devices = None   # set_devices() / set_device(); None = environment

# The reference's GPU module has a source-level precision switch (`precision = 'single' | 'double'`,
# mobility/mobility_pycuda.py:7-19).  Same switch here, settable at run time: with 'single' the translation <- force
# products with open boundaries (wall / no wall) run the fp32 twin of the symmetric kernel (csrc/sym32_kernels.h,
# ~1e-6 relative accuracy, 1.5-1.6x faster); every other product keeps running in fp64.
precision = 'double'

# With more than one device the products of this module (and forces.calc_blob_blob_forces_hip) run on the
# single-process multi-device engine (multi.MultiContext: pair shards + fixed-order slice reduction over xGMI) once a
# configuration has at least this many blobs; smaller ones stay on devices()[0], where one launch is already shorter
# than the cross-device hand-offs.  Environment: RMB_MULTI_MIN_BLOBS.
multi_min_blobs = int(os.environ.get("RMB_MULTI_MIN_BLOBS", "20000"))


def devices():
  """Devices the module-level products use: set_devices() / set_device(), else RMB_DEVICES="0,1,2,3", else
  RMB_DEVICE="2", else [0]."""
  if _devices is not None:
    return list(_devices)
  env = os.environ.get("RMB_DEVICES", "").strip()
  if env:
    return [int(x) for x in env.replace(";", ",").split(",") if x.strip() != ""]
  env = os.environ.get("RMB_DEVICE", "").strip()
  if env:
    return [int(env)]
  return [0]


def set_devices(device_list):
  """Use these devices (indices into the visible ones) for every product of this module and of forces.py from now
  on: one entry = that GPU; several = the whole list behind every call, in this one process (a reference-side
  `elif implementation == 'hip'` then uses the node).  None = back to the environment / device 0."""
  global _devices
  reset()
  _devices = None if device_list is None else [int(d) for d in device_list]
  if _devices is not None and not _devices:
    raise ValueError("set_devices needs at least one device")
  # the stateless C entry points (source -> target, pressure, double layer) run on the library's default context
  _lib.check(_lib.load().rmb_default_ctx_set_device(-1 if _devices is None else _devices[0]))


def set_device(index):
  """Use this one device (see set_devices)."""
  set_devices([index])


def _context(n=0):
  """The context a configuration of n blobs is bound to: the engine over all devices when there are several and the
  configuration is large enough, the one-device context otherwise."""
  global _ctx, _mctx
  devs = devices()
  if len(devs) > 1 and n >= multi_min_blobs:
    if _mctx is None:
      from .multi import MultiContext
      _mctx = MultiContext(devs)
    return _mctx
  if _ctx is None:
    _ctx = MobilityContext(devs[0])
  return _ctx


def active_devices(n):
  """Devices a product on n blobs runs on right now (bench.py reports it as host_surface.devices)."""
  devs = devices()
  return devs if (len(devs) > 1 and n >= multi_min_blobs) else devs[:1]


def reset():
  """Drop the module-level contexts (frees device memory)."""
  global _ctx, _mctx
  if _ctx is not None:
    _ctx.close()
  if _mctx is not None:
    _mctx.close()
  _ctx = None
  _mctx = None
  _cached.clear()


_NO_PERIODICITY = (0.0, 0.0, 0.0)


def _bind_positions(r_vectors, a, L, wall):
  # (this wrapper is ~15 us of a 190 us call at 1e4 blobs: nothing is converted or sent twice)
  r = np.ascontiguousarray(r_vectors, dtype=np.float64).reshape(-1)
  Lt = _NO_PERIODICITY if L is None else tuple(float(x) for x in np.asarray(L, dtype=np.float64).reshape(3))
  ctx = _context(r.size // 3)
  if precision not in ('single', 'double'):
    raise ValueError("mobility.precision must be 'single' or 'double'")
  want = 32 if precision == 'single' else 64
  known = getattr(ctx, "_options_set", None)                   # every option set through the context wrapper
  if known is None or known.get("precision") != want:         # one C call less per product while it does not change
    ctx.set_option("precision", want)
  c = _cached.get(id(ctx))
  if (c is not None and c[1] == float(a) and c[2] == Lt and c[3] == bool(wall) and c[0].size == r.size
      and np.array_equal(c[0], r)):
    if ctx.target_range != (0, ctx.n):
      ctx.set_target_range(0, ctx.n)
    return ctx
  ctx.set_positions(r, a, Lt, wall)
  _cached[id(ctx)] = (r.copy(), float(a), Lt, bool(wall))
  return ctx


def _product(kind, wall, in_plane, r_vectors, vec, eta, a, kwargs, vec2=None):
  ctx = _bind_positions(r_vectors, a, kwargs.get('periodic_length'), wall)
  return ctx.matvec(kind, vec, eta, vec2=vec2, in_plane=in_plane)


# ---------------------------------------------------------------------------------------------
# Wall-overlap regularisation helpers (same returns as mobility/mobility.py:52-84), vectorised.
# The *_hip products do NOT call these -- the device does the same arithmetic -- they are kept
# for callers that use them directly (e.g. the per-body preconditioner builders).
# ---------------------------------------------------------------------------------------------
def shift_heights(r_vectors, blob_radius, *args, **kwargs):
  '''z_effective = maximum(z, blob_radius); returns a copy (mobility/mobility.py:52-64).'''
  r_effective = np.copy(r_vectors)
  r_effective[r_vectors[:, 2] <= blob_radius, 2] = blob_radius
  return r_effective


def damping_matrix_B(r_vectors, blob_radius, *args, **kwargs):
  '''(sparse diagonal B, overlap flag); B_ii = z_i/a for z_i < a else 1 (mobility/mobility.py:67-84).'''
  r = np.asarray(r_vectors).reshape(-1, 3)
  z = r[:, 2]
  below = z < blob_radius
  b = np.where(below, z / blob_radius, 1.0)
  B = np.repeat(b, 3)
  overlap = bool(np.any(below))
  return (scipy.sparse.dia_matrix((B, 0), shape=(B.size, B.size)), overlap)


def shift_heights_different_radius(r_vectors, blob_radius, *args, **kwargs):
  '''z_effective = maximum(z, blob_radius[k]) per blob; returns a copy (mobility/mobility.py:87-99).'''
  r_effective = np.copy(r_vectors)
  rad = np.asarray(blob_radius, dtype=np.float64).reshape(-1)
  low = ~(r_effective[:, 2] > rad)
  r_effective[low, 2] = rad[low]
  return r_effective


def damping_matrix_B_different_radius(r_vectors, blob_radius, *args, **kwargs):
  '''(sparse diagonal B, overlap flag); B_ii = z_i/a_i for z_i < a_i else 1 (mobility/mobility.py:102-119).'''
  r = np.asarray(r_vectors).reshape(-1, 3)
  rad = np.asarray(blob_radius, dtype=np.float64).reshape(-1)
  below = r[:, 2] < rad
  B = np.repeat(np.where(below, r[:, 2] / rad, 1.0), 3)
  return (scipy.sparse.dia_matrix((B, 0), shape=(B.size, B.size)), bool(np.any(below)))


# ---------------------------------------------------------------------------------------------
# translation <- force
# ---------------------------------------------------------------------------------------------
def single_wall_mobility_trans_times_force_hip(r_vectors, force, eta, a, *args, **kwargs):
  '''u = B M_tt(z_eff) B f above a no-slip wall (mobility/mobility.py:222-252, :1132-1163).'''
  return _product('tt', True, False, r_vectors, force, eta, a, kwargs)


def no_wall_mobility_trans_times_force_hip(r_vectors, force, eta, a, *args, **kwargs):
  '''u = M_tt f, unbounded RPY (mobility/mobility.py:288-297, :1119-1129).'''
  return _product('tt', False, False, r_vectors, force, eta, a, kwargs)


def in_plane_mobility_trans_times_force_hip(r_vectors, force, eta, a, *args, **kwargs):
  '''x,y rows/columns only of the wall M_tt (mobility/mobility.py:255-285, :1166-1197).'''
  return _product('tt', True, True, r_vectors, force, eta, a, kwargs)


# ---------------------------------------------------------------------------------------------
# translation <- torque
# ---------------------------------------------------------------------------------------------
def single_wall_mobility_trans_times_torque_hip(r_vectors, torque, eta, a, *args, **kwargs):
  '''u = B M_tr(z_eff) B tau (mobility/mobility.py:427-452, :1213-1233).'''
  return _product('tr', True, False, r_vectors, torque, eta, a, kwargs)


def no_wall_mobility_trans_times_torque_hip(r_vectors, torque, eta, a, *args, **kwargs):
  '''u = M_tr tau, unbounded (mobility/mobility.py:482-491, :1200-1210).'''
  return _product('tr', False, False, r_vectors, torque, eta, a, kwargs)


def in_plane_mobility_trans_times_torque_hip(r_vectors, torque, eta, a, *args, **kwargs):
  '''in-plane wall M_tr (mobility/mobility.py:454-479, :1235-1255).'''
  return _product('tr', True, True, r_vectors, torque, eta, a, kwargs)


# ---------------------------------------------------------------------------------------------
# rotation <- force, rotation <- torque
# ---------------------------------------------------------------------------------------------
def single_wall_mobility_rot_times_force_hip(r_vectors, force, eta, a, *args, **kwargs):
  '''w = B M_rt(z_eff) B f (mobility/mobility.py:300-325, :1273-1298).'''
  return _product('rt', True, False, r_vectors, force, eta, a, kwargs)


def no_wall_mobility_rot_times_force_hip(r_vectors, force, eta, a, *args, **kwargs):
  '''w = M_rt f, unbounded (mobility/mobility.py:328-339, :1258-1270).'''
  return _product('rt', False, False, r_vectors, force, eta, a, kwargs)


def single_wall_mobility_rot_times_torque_hip(r_vectors, torque, eta, a, *args, **kwargs):
  '''w = B M_rr(z_eff) B tau (mobility/mobility.py:342-367, :1316-1341).'''
  return _product('rr', True, False, r_vectors, torque, eta, a, kwargs)


def no_wall_mobility_rot_times_torque_hip(r_vectors, torque, eta, a, *args, **kwargs):
  '''w = M_rr tau, unbounded (mobility/mobility.py:370-381, :1301-1313).'''
  return _product('rr', False, False, r_vectors, torque, eta, a, kwargs)


# ---------------------------------------------------------------------------------------------
# fused translation <- (force, torque)   (GPU-only in the reference too)
# ---------------------------------------------------------------------------------------------
def single_wall_mobility_trans_times_force_torque_hip(r_vectors, force, torque, eta, a, *args, **kwargs):
  '''u = B (M_tt B f + M_tr B tau) in one sweep (mobility/mobility.py:384-410).'''
  return _product('tt_tr', True, False, r_vectors, force, eta, a, kwargs, vec2=torque)


def no_wall_mobility_trans_times_force_torque_hip(r_vectors, force, torque, eta, a, *args, **kwargs):
  '''u = M_tt f + M_tr tau, unbounded (mobility/mobility.py:413-424).'''
  return _product('tt_tr', False, False, r_vectors, force, eta, a, kwargs, vec2=torque)


def free_surface_mobility_trans_times_force_hip(r_vectors, force, eta, a, *args, **kwargs):
  '''u = M f above a stress-free surface at z = 0: RPY plus the mirrored image blob, no height clamp
  (mobility/mobility.py:533-548, :1390-1406).'''
  return _product('tt_free', False, False, r_vectors, force, eta, a, kwargs)


# ---------------------------------------------------------------------------------------------
# source -> target products with per-blob radii (velocity fields, `radii_*` mobility modes)
# ---------------------------------------------------------------------------------------------
def _source_target(source, target, force, radius_source, radius_target, eta, wall, kwargs):
  import ctypes
  L = np.ascontiguousarray(kwargs.get('periodic_length', np.zeros(3)), dtype=np.float64).reshape(3)
  src = np.ascontiguousarray(source, dtype=np.float64).reshape(-1)
  tgt = np.ascontiguousarray(target, dtype=np.float64).reshape(-1)
  f = np.ascontiguousarray(force, dtype=np.float64).reshape(-1)
  ns, nt = src.size // 3, tgt.size // 3
  rs = np.ascontiguousarray(np.broadcast_to(np.asarray(radius_source, dtype=np.float64).reshape(-1), (ns,)))
  rt = np.ascontiguousarray(np.broadcast_to(np.asarray(radius_target, dtype=np.float64).reshape(-1), (nt,)))
  if f.size != 3 * ns:
    raise ValueError("force must have 3*N_source entries")
  out = np.empty(3 * nt)
  p = lambda x: ctypes.c_void_p(x.ctypes.data)  # noqa: E731
  if precision not in ('single', 'double'):
    raise ValueError("mobility.precision must be 'single' or 'double'")
  # the stateless entry point runs on the library's default context: hand it the module's precision switch (only the
  # sources == targets case has a single-precision twin; everything else computes in fp64 whatever it says)
  _lib.check(_lib.load().rmb_default_ctx_set_option(b"precision", 32 if precision == 'single' else 64))
  _lib.check(_lib.load().rmb_mobility_source_target(ns, p(src), p(rs), nt, p(tgt), p(rt), p(f), float(eta), p(L),
                                                    int(wall), p(out)))
  return out


def single_wall_mobility_trans_times_force_source_target_hip(source, target, force, radius_source, radius_target, eta,
                                                             *args, **kwargs):
  '''Velocity of targets (radii radius_target) due to forces on sources (radii radius_source) above a wall;
  per-blob height clamp and B-damping on both sides (mobility/mobility.py:494-530, :551-590).'''
  return _source_target(source, target, force, radius_source, radius_target, eta, True, kwargs)


def no_wall_mobility_trans_times_force_source_target_hip(source, target, force, radius_source, radius_target, eta,
                                                         *args, **kwargs):
  '''Same in an unbounded domain (mobility/mobility.py:593-615).'''
  return _source_target(source, target, force, radius_source, radius_target, eta, False, kwargs)


def mobility_vector_product_source_target_one_wall_hip(source, target, force, radius_source, radius_target, eta, *args, **kwargs):
  '''The reference's pure-Python twin of the same product (mobility/mobility.py:830-902,
  `mobility_vector_product_source_target_one_wall`; equal to its numba sibling to rounding, checked when the goldens were
  generated): served by the same kernel.'''
  return _source_target(source, target, force, radius_source, radius_target, eta, True, kwargs)


def mobility_vector_product_source_target_unbounded_hip(source, target, force, radius_source, radius_target, eta, *args, **kwargs):
  '''mobility/mobility.py:905-960 (`mobility_vector_product_source_target_unbounded`, Zuk et al. 2014): unbounded twin.'''
  return _source_target(source, target, force, radius_source, radius_target, eta, False, kwargs)


def free_surface_mobility_trans_times_force_source_target_hip(source, target, force, radius_source, radius_target, eta,
                                                              *args, **kwargs):
  '''Same below a stress-free surface at z = 0 (mobility/mobility.py:1409-1429, kernel mobility_numba.py:1941-2091):
  no height clamp, mirror image with the z column negated.'''
  return _source_target(source, target, force, radius_source, radius_target, eta, 2, kwargs)


def mobility_radii_trans_times_force(r_vectors, force, eta, a, radius_blobs, function, *args, **kwargs):
  '''M.f for blobs with different radii: sources == targets (mobility/mobility.py:1369-1374).'''
  return function(r_vectors, r_vectors, force, radius_blobs, radius_blobs, eta, *args, **kwargs)


# ---------------------------------------------------------------------------------------------
# Stokeslet pressure and Stokes double layer, source -> target
# ---------------------------------------------------------------------------------------------
def _flat(x, what, n=None):
  x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
  if n is not None and x.size != n:
    raise ValueError("%s must have %d entries" % (what, n))
  return x


def _pressure(source, target, force, wall, kwargs):
  import ctypes
  L = np.ascontiguousarray(kwargs.get('periodic_length', np.zeros(3)), dtype=np.float64).reshape(3)
  src, tgt = _flat(source, "source"), _flat(target, "target")
  ns, nt = src.size // 3, tgt.size // 3
  f = _flat(force, "force", 3 * ns)
  out = np.empty(nt)
  p = lambda x: ctypes.c_void_p(x.ctypes.data)  # noqa: E731
  _lib.check(_lib.load().rmb_pressure_stokeslet(ns, p(src), nt, p(tgt), p(f), p(L), int(wall), p(out)))
  return out


def no_wall_pressure_Stokeslet_hip(source, target, force, *args, **kwargs):
  '''Pressure at the targets created by Stokeslets at the sources, unbounded (mobility/mobility.py:1345-1354;
  kernel mobility_numba.py:1332-1396).  periodic_length must be zero (see include/rmb_mobility.h).'''
  return _pressure(source, target, force, 0, kwargs)


def single_wall_pressure_Stokeslet_hip(source, target, force, *args, **kwargs):
  '''Same above a no-slip wall (Blake's image system; mobility/mobility.py:1357-1366, kernel mobility_numba.py:1399-1476).
  The 1/(4 pi) factor is applied once: the reference rescales its running sum inside the source loop, which equals
  this result for one source only.'''
  return _pressure(source, target, force, 1, kwargs)


def _double_layer(source, target, normals, vector, weights, wall, blob_radius):
  import ctypes
  src, tgt = _flat(source, "source"), _flat(target, "target")
  ns, nt = src.size // 3, tgt.size // 3
  n, v, w = _flat(normals, "normals", 3 * ns), _flat(vector, "vector", 3 * ns), _flat(weights, "weights", ns)
  out = np.empty(3 * nt)
  p = lambda x: ctypes.c_void_p(x.ctypes.data)  # noqa: E731
  _lib.check(_lib.load().rmb_double_layer(ns, p(src), nt, p(tgt), p(n), p(v), p(w), int(wall), float(blob_radius), p(out)))
  return out


def double_layer_source_target_hip(source, target, normals, vector, weights, *args, **kwargs):
  '''Stokes double-layer operator times a vector, diagonal terms zero; kwarg wall = 1 adds the image system of a
  no-slip wall (mobility/mobility.py:1376-1387, kernel mobility_numba.py:1662-1766).'''
  return _double_layer(source, target, normals, vector, weights, 1 if kwargs.get('wall', 0) else 0, -1.0)


def no_wall_double_layer_source_target_hip(source, target, normals, vector, weights, blob_radius, *args, **kwargs):
  '''RPY-regularised double layer, unbounded (mobility/mobility.py:1432-1442, kernel mobility_numba.py:2095-2168).'''
  return _double_layer(source, target, normals, vector, weights, 0, float(blob_radius))


# ---------------------------------------------------------------------------------------------
# dense builders (used per body by the preconditioner / body_mobility scheme in the reference)
# ---------------------------------------------------------------------------------------------
def _dense(r_vectors, eta, a, wall):
  import torch
  r = np.ascontiguousarray(r_vectors, dtype=np.float64).reshape(-1, 3)
  n = len(r)
  dev = "cuda:%d" % devices()[0]
  ctx = MobilityContext(devices()[0])
  try:
    ctx.set_positions(torch.as_tensor(r.reshape(-1), device=dev), a, None, wall)
    first = torch.zeros(1, dtype=torch.int64, device=dev)
    M = ctx.body_mobility_dense_device(first, n, eta)[0].cpu().numpy()
  finally:
    ctx.close()
  return M


def single_wall_fluid_mobility_hip(r_vectors, eta, a, *args, **kwargs):
  '''Dense 3N x 3N wall mobility B M(z_eff) B (mobility/mobility.py:1018-1116 builds M(z) without the
  clamp; identical whenever every blob has z > a).'''
  return _dense(r_vectors, eta, a, True)


def rotne_prager_tensor_hip(r_vectors, eta, a, *args, **kwargs):
  '''Dense 3N x 3N unbounded RPY mobility (mobility/mobility.py:967-1013).'''
  return _dense(r_vectors, eta, a, False)


def single_wall_fluid_mobility_product_hip(r_vectors, vector, eta, a, *args, **kwargs):
  '''Dense wall mobility times a vector (mobility/mobility.py:711-724: `single_wall_fluid_mobility(...) @ vector`, no
  pseudo-PBC).  The dense matrix carries no height clamp in the reference; for blobs with z > a -- the only case in
  which that matrix is a valid mobility -- this is the matrix-free product, which is what runs.'''
  return _product('tt', True, False, r_vectors, vector, eta, a, {})


def no_wall_fluid_mobility_product_hip(r_vectors, vector, eta, a, *args, **kwargs):
  '''Dense RPY mobility times a vector (mobility/mobility.py:727-736), evaluated matrix-free.'''
  return _product('tt', False, False, r_vectors, vector, eta, a, {})


def single_wall_self_mobility_with_rotation_hip(location, eta, a, *args, **kwargs):
  '''6 x 6 self mobility of one sphere of radius a at height location[2] above the wall, force and torque to velocity
  and angular velocity (Swan & Brady; mobility/mobility.py:739-772).  Built from the device kernels' own self terms:
  the six columns are the grand-mobility product [[M_tt, M_tr], [M_rt, M_rr]] of a single blob on the unit vectors.
  Sign of the coupling blocks: this legacy routine of the reference writes them with epsilon(2, l, m), the OPPOSITE sign
  to its own product kernels (mobility_numba.py:646-679, :1035-1066: a torque +y above the wall drives the sphere towards
  +x, like a rolling wheel; the routine returns -x).  The function returns what the reference's routine returns, so the
  two off-diagonal blocks of the device result are negated; the products themselves are untouched.'''
  import torch
  r = np.ascontiguousarray(location, dtype=np.float64).reshape(1, 3)
  dev = "cuda:%d" % devices()[0]
  ctx = MobilityContext(devices()[0])
  try:
    ctx.set_positions(torch.as_tensor(r.reshape(-1), device=dev), a, None, True)
    M = np.empty((6, 6))
    for k in range(6):
      e = np.zeros(6)
      e[k] = 1.0
      f, t = torch.as_tensor(e[:3].copy(), device=dev), torch.as_tensor(e[3:].copy(), device=dev)
      u, w = ctx.matvec_op_device("grand", (f, t), eta)
      M[:3, k] = u.cpu().numpy()
      M[3:, k] = w.cpu().numpy()
  finally:
    ctx.close()
  M[:3, 3:] *= -1.0
  M[3:, :3] *= -1.0
  return M
