"""CPU oracle loader -- TEST INFRASTRUCTURE ONLY.

ctypes front-end for ``oracle/liboracle_mobility.so`` (see the header of
``oracle_mobility.c``).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product
package ``rigidmultiblobswall_amd`` never does.

The ``*_oracle`` functions reproduce the reference's Python wrapper semantics
(``mobility/mobility.py:1119-1341`` for the numba family: height clamp +
B-damping around the wall kernels, ``periodic_length`` kwarg, flat ``(3N,)``
return) so that parity tests can call oracle and HIP path with the same
arguments.  Parity status: pinned by ``tests/golden/*.npz`` (generated from the
reference's own Python by ``oracle/gen_golden.py``).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

KIND = {"tt": 0, "tr": 1, "rt": 2, "rr": 3}

_dp = ctypes.POINTER(ctypes.c_double)
_lp = ctypes.POINTER(ctypes.c_long)


def build(force=False):
  """Compile the oracle shared objects (gcc, a few seconds)."""
  need = force or not all(
      os.path.exists(os.path.join(_HERE, n))
      for n in ("liboracle_mobility.so", "liboracle_mobility_fast.so"))
  if need:
    subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))


def _lib(fast=False):
  name = "liboracle_mobility_fast.so" if fast else "liboracle_mobility.so"
  if name not in _LIBS:
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
      build()
    lib = ctypes.CDLL(path)
    lib.oracle_mobility_matvec.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long,
                                           _dp, _dp, ctypes.c_double, ctypes.c_double, _dp, _dp]
    lib.oracle_mobility_matvec.restype = ctypes.c_int
    lib.oracle_mobility_matvec_targets.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_long, _dp, _dp,
                                                   ctypes.c_double, ctypes.c_double, _dp, ctypes.c_long,
                                                   _lp, _dp]
    lib.oracle_mobility_matvec_targets.restype = ctypes.c_int
    lib.oracle_mobility_dense.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_long, _dp,
                                          ctypes.c_double, ctypes.c_double, _dp]
    lib.oracle_mobility_dense.restype = ctypes.c_int
    lib.oracle_blob_blob_force.argtypes = [ctypes.c_long, _dp, _dp, ctypes.c_double, ctypes.c_double,
                                           ctypes.c_double, _dp]
    lib.oracle_blob_blob_force.restype = ctypes.c_int
    lib.oracle_blob_blob_force_targets.argtypes = [ctypes.c_long, _dp, _dp, ctypes.c_double, ctypes.c_double,
                                                   ctypes.c_double, ctypes.c_long, _lp, _dp]
    lib.oracle_blob_blob_force_targets.restype = ctypes.c_int
    lib.oracle_blob_blob_force_radii.argtypes = [ctypes.c_long, _dp, _dp, _dp, ctypes.c_double, ctypes.c_double, _dp]
    lib.oracle_blob_blob_force_radii.restype = ctypes.c_int
    lib.oracle_wall_regularisation.argtypes = [ctypes.c_long, _dp, ctypes.c_double, _dp, _dp,
                                               ctypes.POINTER(ctypes.c_int)]
    lib.oracle_wall_regularisation.restype = ctypes.c_int
    lib.oracle_free_surface_matvec.argtypes = [ctypes.c_long, _dp, _dp, ctypes.c_double, ctypes.c_double, _dp, _dp]
    lib.oracle_free_surface_matvec.restype = ctypes.c_int
    lib.oracle_source_target_matvec.argtypes = [ctypes.c_long, _dp, _dp, ctypes.c_long, _dp, _dp, _dp, ctypes.c_double,
                                                _dp, ctypes.c_int, _dp]
    lib.oracle_source_target_matvec.restype = ctypes.c_int
    lib.oracle_num_threads.restype = ctypes.c_int
    lib.oracle_pressure_stokeslet.argtypes = [ctypes.c_long, _dp, ctypes.c_long, _dp, _dp, ctypes.c_int, _dp]
    lib.oracle_double_layer.argtypes = [ctypes.c_long, _dp, ctypes.c_long, _dp, _dp, _dp, _dp, ctypes.c_int, ctypes.c_double, _dp]
    lib.oracle_set_num_threads.argtypes = [ctypes.c_int]
    lib.oracle_set_num_threads.restype = None
    _LIBS[name] = lib
  return _LIBS[name]


def _c(x):
  return np.ascontiguousarray(x, dtype=np.float64)


def _p(x):
  return x.ctypes.data_as(_dp)


def num_threads():
  return int(_lib().oracle_num_threads())


def usable_cpus():
  """CPUs this process may really use: scheduler affinity capped by the cgroup CPU quota (v2 cpu.max, v1 cfs quota)."""
  n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
  try:
    with open("/sys/fs/cgroup/cpu.max") as fh:
      quota, period = fh.read().split()[:2]
    if quota != "max":
      n = min(n, max(1, int(round(float(quota) / float(period)))))
  except (OSError, ValueError):
    try:
      with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
        quota = int(fh.read())
      with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
        period = int(fh.read())
      if quota > 0:
        n = min(n, max(1, int(round(quota / period))))
    except (OSError, ValueError):
      pass
  return n


def set_num_threads(n):
  for fast in (False, True):
    _lib(fast).oracle_set_num_threads(int(n))


def raw_matvec(kind, wall, r, v, eta, a, L=None, in_plane=False, fast=False):
  """Bare kernel: no height clamp, no B (== mobility_numba.* functions)."""
  r = _c(r).reshape(-1)
  v = _c(v).reshape(-1)
  N = r.size // 3
  L = _c(np.zeros(3) if L is None else L).reshape(3)
  out = np.zeros(3 * N)
  rc = _lib(fast).oracle_mobility_matvec(KIND[kind], int(wall), int(in_plane), N, _p(r), _p(v),
                                         float(eta), float(a), _p(L), _p(out))
  if rc != 0:
    raise RuntimeError("oracle_mobility_matvec failed: %d" % rc)
  return out


def raw_matvec_targets(kind, wall, r, v, eta, a, targets, L=None):
  r = _c(r).reshape(-1)
  v = _c(v).reshape(-1)
  N = r.size // 3
  L = _c(np.zeros(3) if L is None else L).reshape(3)
  t = np.ascontiguousarray(targets, dtype=np.int64)
  out = np.zeros(3 * t.size)
  rc = _lib().oracle_mobility_matvec_targets(KIND[kind], int(wall), N, _p(r), _p(v), float(eta),
                                             float(a), _p(L), t.size,
                                             t.ctypes.data_as(_lp), _p(out))
  if rc != 0:
    raise RuntimeError("oracle_mobility_matvec_targets failed: %d" % rc)
  return out


def dense(kind, wall, r, eta, a):
  r = _c(r).reshape(-1)
  N = r.size // 3
  M = np.zeros((3 * N, 3 * N))
  rc = _lib().oracle_mobility_dense(KIND[kind], int(wall), N, _p(r), float(eta), float(a), _p(M))
  if rc != 0:
    raise RuntimeError("oracle_mobility_dense failed: %d" % rc)
  return M


def wall_regularisation(r, a):
  """(r_eff, bdiag, overlap) as mobility/mobility.py:52-84."""
  r = _c(r).reshape(-1)
  N = r.size // 3
  r_eff = np.empty(3 * N)
  b = np.empty(N)
  ov = ctypes.c_int(0)
  _lib().oracle_wall_regularisation(N, _p(r), float(a), _p(r_eff), _p(b), ctypes.byref(ov))
  return r_eff, b, bool(ov.value)


def _wrapped(kind, wall, r_vectors, vec, eta, a, in_plane=False, fast=False, **kwargs):
  L = kwargs.get("periodic_length", np.zeros(3))
  if not wall:
    return raw_matvec(kind, 0, r_vectors, vec, eta, a, L, fast=fast)
  r_eff, b, overlap = wall_regularisation(r_vectors, a)
  v = _c(vec).reshape(-1)
  B = np.repeat(b, 3)
  if overlap:
    v = B * v
  out = raw_matvec(kind, 1, r_eff, v, eta, a, L, in_plane=in_plane, fast=fast)
  if overlap:
    out = B * out
  return out


# --- reference-named surface (mobility/mobility.py:1119-1341) ----------------
def no_wall_mobility_trans_times_force_oracle(r, f, eta, a, *args, **kw):
  return _wrapped("tt", 0, r, f, eta, a, **kw)


def single_wall_mobility_trans_times_force_oracle(r, f, eta, a, *args, **kw):
  return _wrapped("tt", 1, r, f, eta, a, **kw)


def in_plane_mobility_trans_times_force_oracle(r, f, eta, a, *args, **kw):
  return _wrapped("tt", 1, r, f, eta, a, in_plane=True, **kw)


def no_wall_mobility_trans_times_torque_oracle(r, t, eta, a, *args, **kw):
  return _wrapped("tr", 0, r, t, eta, a, **kw)


def single_wall_mobility_trans_times_torque_oracle(r, t, eta, a, *args, **kw):
  return _wrapped("tr", 1, r, t, eta, a, **kw)


def in_plane_mobility_trans_times_torque_oracle(r, t, eta, a, *args, **kw):
  return _wrapped("tr", 1, r, t, eta, a, in_plane=True, **kw)


def no_wall_mobility_rot_times_force_oracle(r, f, eta, a, *args, **kw):
  return _wrapped("rt", 0, r, f, eta, a, **kw)


def single_wall_mobility_rot_times_force_oracle(r, f, eta, a, *args, **kw):
  return _wrapped("rt", 1, r, f, eta, a, **kw)


def no_wall_mobility_rot_times_torque_oracle(r, t, eta, a, *args, **kw):
  return _wrapped("rr", 0, r, t, eta, a, **kw)


def single_wall_mobility_rot_times_torque_oracle(r, t, eta, a, *args, **kw):
  return _wrapped("rr", 1, r, t, eta, a, **kw)


def single_wall_mobility_trans_times_force_torque_oracle(r, f, t, eta, a, *args, **kw):
  """Fused K11 has no numba twin; its value is M_tt f + M_tr tau
  (mobility/mobility_pycuda.py:1351-1388)."""
  return _wrapped("tt", 1, r, f, eta, a, **kw) + _wrapped("tr", 1, r, t, eta, a, **kw)


def no_wall_mobility_trans_times_force_torque_oracle(r, f, t, eta, a, *args, **kw):
  return _wrapped("tt", 0, r, f, eta, a, **kw) + _wrapped("tr", 0, r, t, eta, a, **kw)


def free_surface_mobility_trans_times_force_oracle(r, f, eta, a, *args, **kw):
  """mobility/mobility.py:1390-1406 (no height clamp)."""
  L = _c(kw.get("periodic_length", np.zeros(3))).reshape(3)
  r = _c(r).reshape(-1)
  f = _c(f).reshape(-1)
  out = np.zeros(r.size)
  rc = _lib(kw.get("fast", False)).oracle_free_surface_matvec(r.size // 3, _p(r), _p(f), float(eta), float(a), _p(L), _p(out))
  if rc != 0:
    raise RuntimeError("oracle_free_surface_matvec failed: %d" % rc)
  return out


def _source_target(source, target, force, radius_source, radius_target, eta, wall, **kw):
  """Wrapper semantics of mobility/mobility.py:551-615: per-blob height clamp + B on both sides when wall."""
  L = _c(kw.get("periodic_length", np.zeros(3))).reshape(3)
  src = _c(source).reshape(-1, 3).copy()
  tgt = _c(target).reshape(-1, 3).copy()
  rs = _c(radius_source).reshape(-1)
  rt = _c(radius_target).reshape(-1)
  f = _c(force).reshape(-1, 3).copy()
  bt = np.ones(len(tgt))
  if wall == 1:
    # damping_matrix_B_different_radius, mobility.py:102-119 (tracers have radius 0: the quotient is never selected)
    bs = np.where(src[:, 2] < rs, src[:, 2] / np.where(rs > 0, rs, 1.0), 1.0)
    bt = np.where(tgt[:, 2] < rt, tgt[:, 2] / np.where(rt > 0, rt, 1.0), 1.0)
    src[:, 2] = np.where(src[:, 2] > rs, src[:, 2], rs)      # shift_heights_different_radius, mobility.py:87-99
    tgt[:, 2] = np.where(tgt[:, 2] > rt, tgt[:, 2], rt)
    f = f * bs[:, None]
  out = np.zeros(3 * len(tgt))
  src, tgt, f = _c(src).reshape(-1), _c(tgt).reshape(-1), _c(f).reshape(-1)
  rc = _lib().oracle_source_target_matvec(len(rs), _p(src), _p(rs), len(rt), _p(tgt), _p(rt), _p(f), float(eta), _p(L),
                                          int(wall), _p(out))
  if rc != 0:
    raise RuntimeError("oracle_source_target_matvec failed: %d" % rc)
  return (out.reshape(-1, 3) * bt[:, None]).reshape(-1)


def single_wall_mobility_trans_times_force_source_target_oracle(source, target, force, radius_source, radius_target, eta,
                                                                *args, **kw):
  return _source_target(source, target, force, radius_source, radius_target, eta, 1, **kw)


def no_wall_mobility_trans_times_force_source_target_oracle(source, target, force, radius_source, radius_target, eta,
                                                            *args, **kw):
  return _source_target(source, target, force, radius_source, radius_target, eta, 0, **kw)


def free_surface_mobility_trans_times_force_source_target_oracle(source, target, force, radius_source, radius_target, eta,
                                                                 *args, **kw):
  return _source_target(source, target, force, radius_source, radius_target, eta, 2, **kw)


def calc_blob_blob_forces_oracle(r_vectors, *args, **kwargs):
  """multi_bodies/forces_numba.py:58-71; returns (N,3)."""
  L = _c(kwargs.get("periodic_length")).reshape(3)
  eps = float(kwargs.get("repulsion_strength"))
  b = float(kwargs.get("debye_length"))
  a = float(kwargs.get("blob_radius"))
  r = _c(r_vectors).reshape(-1)
  N = r.size // 3
  out = np.zeros(3 * N)
  rc = _lib(kwargs.get("_fast", False)).oracle_blob_blob_force(N, _p(r), _p(L), eps, b, a, _p(out))
  if rc != 0:
    raise RuntimeError("oracle_blob_blob_force failed: %d" % rc)
  return out.reshape(N, 3)


def calc_blob_blob_forces_targets_oracle(r_vectors, targets, *args, **kwargs):
  """The rows `targets` of calc_blob_blob_forces_oracle (all blobs as sources): full-size spot checks."""
  L = _c(kwargs.get("periodic_length", np.zeros(3))).reshape(3)
  r = _c(r_vectors).reshape(-1)
  tg = np.ascontiguousarray(targets, dtype=np.int64)
  out = np.zeros(3 * tg.size)
  rc = _lib().oracle_blob_blob_force_targets(r.size // 3, _p(r), _p(L), float(kwargs.get("repulsion_strength")),
                                             float(kwargs.get("debye_length")), float(kwargs.get("blob_radius")), tg.size,
                                             tg.ctypes.data_as(_lp), _p(out))
  if rc != 0:
    raise RuntimeError("oracle_blob_blob_force_targets failed: %d" % rc)
  return out.reshape(-1, 3)


def calc_blob_blob_forces_radii_oracle(r_vectors, radius_blobs, *args, **kwargs):
  """multi_bodies/forces_numba.py:125-137 (`radii_numba`): one radius per blob, contact distance a_i + a_j."""
  r = _c(r_vectors).reshape(-1)
  N = r.size // 3
  rad = _c(radius_blobs).reshape(-1)
  L = _c(kwargs.get("periodic_length", np.zeros(3)) if kwargs.get("periodic_length") is not None else np.zeros(3)).reshape(3)
  out = np.zeros(3 * N)
  rc = _lib().oracle_blob_blob_force_radii(N, _p(r), _p(rad), _p(L), float(kwargs.get("repulsion_strength")),
                                           float(kwargs.get("debye_length")), _p(out))
  if rc != 0:
    raise RuntimeError("oracle_blob_blob_force_radii failed: %d" % rc)
  return out.reshape(N, 3)


# ---- Stokeslet pressure / Stokes double layer, source -> target (mobility/mobility.py:1345-1366, :1376-1387, :1432-1442)
def _pressure(source, target, force, wall, **kw):
  L = np.asarray(kw.get("periodic_length", np.zeros(3)), dtype=np.float64)
  if np.any(L > 0):
    raise ValueError("pressure: only periodic_length = 0 is restated (the reference divides by the unwrapped distance)")
  src, tgt, f = _c(np.asarray(source, dtype=np.float64).reshape(-1)), _c(np.asarray(target, dtype=np.float64).reshape(-1)), \
      _c(np.asarray(force, dtype=np.float64).reshape(-1))
  p = np.empty(tgt.size // 3)
  _lib().oracle_pressure_stokeslet(src.size // 3, _p(src), tgt.size // 3, _p(tgt), _p(f), int(wall), _p(p))
  return p


def no_wall_pressure_Stokeslet_oracle(source, target, force, *args, **kw):
  return _pressure(source, target, force, 0, **kw)


def single_wall_pressure_Stokeslet_oracle(source, target, force, *args, **kw):
  return _pressure(source, target, force, 1, **kw)


def _double_layer(source, target, normals, vector, weights, wall, blob_radius):
  src, tgt = _c(np.asarray(source, dtype=np.float64).reshape(-1)), _c(np.asarray(target, dtype=np.float64).reshape(-1))
  n, v, w = _c(np.asarray(normals, dtype=np.float64).reshape(-1)), _c(np.asarray(vector, dtype=np.float64).reshape(-1)), \
      _c(np.asarray(weights, dtype=np.float64).reshape(-1))
  u = np.empty(tgt.size)
  _lib().oracle_double_layer(src.size // 3, _p(src), tgt.size // 3, _p(tgt), _p(n), _p(v), _p(w), int(wall), float(blob_radius), _p(u))
  return u


def double_layer_source_target_oracle(source, target, normals, vector, weights, *args, **kw):
  return _double_layer(source, target, normals, vector, weights, 1 if kw.get("wall", 0) else 0, -1.0)


def no_wall_double_layer_source_target_oracle(source, target, normals, vector, weights, blob_radius, *args, **kw):
  return _double_layer(source, target, normals, vector, weights, 0, blob_radius)
