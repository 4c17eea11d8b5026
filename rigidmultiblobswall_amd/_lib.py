"""ctypes binding of librmb_mobility.so (C ABI declared in include/rmb_mobility.h).

There is NO CPU fallback: if the HIP library is missing or no device is visible every compute
call raises.  (The reference probes its backends and silently skips the missing ones,
mobility/mobility.py:9-50; a silent fallback here would void the parity claims.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RMB_DIAGNOSTICS=1 (tools/ only): the diagnostics build, whose option table knows "wave_clock" and "skip_pairs" -- the
# release library refuses them (they change which kernel runs / make results wrong by design).
DIAGNOSTICS = os.environ.get("RMB_DIAGNOSTICS", "0") not in ("", "0")
LIB_PATH = os.path.join(_HERE, "librmb_mobility_diag.so" if DIAGNOSTICS else "librmb_mobility.so")

_dp = ctypes.POINTER(ctypes.c_double)
_vp = ctypes.c_void_p
_lp = ctypes.POINTER(ctypes.c_long)

KIND_TT, KIND_TR, KIND_RT, KIND_RR, KIND_TT_TR, KIND_TT_FREE = 0, 1, 2, 3, 4, 5
OP_VELOCITY_FROM_FORCE_TORQUE, OP_GRAND, OP_FORCE_COLUMN, OP_TT_MULTI, OP_TR_MULTI, OP_RT_MULTI, OP_RR_MULTI = range(7)
# name -> (rmb_op, inputs, outputs); the "*_multi" operations take 1..4 vectors
OPS = {"velocity_from_force_torque": (OP_VELOCITY_FROM_FORCE_TORQUE, 2, 1), "grand": (OP_GRAND, 2, 2),
       "force_column": (OP_FORCE_COLUMN, 1, 2), "tt_multi": (OP_TT_MULTI, None, None), "tr_multi": (OP_TR_MULTI, None, None),
       "rt_multi": (OP_RT_MULTI, None, None), "rr_multi": (OP_RR_MULTI, None, None)}
KINDS = {"tt": KIND_TT, "tr": KIND_TR, "rt": KIND_RT, "rr": KIND_RR, "tt_tr": KIND_TT_TR, "tt_free": KIND_TT_FREE}

# every symbol include/rmb_mobility.h declares: (restype, argtypes)
SYMBOLS = {
    "rmb_version": (ctypes.c_char_p, []),
    "rmb_last_error": (ctypes.c_char_p, []),
    "rmb_device_count": (ctypes.c_int, []),
    "rmb_ctx_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_vp)]),
    "rmb_ctx_destroy": (ctypes.c_int, [_vp]),
    "rmb_ctx_set_stream": (ctypes.c_int, [_vp, _vp]),
    "rmb_ctx_release_stream": (ctypes.c_int, [_vp]),
    "rmb_ctx_set_option": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_long]),
    "rmb_ctx_get_option": (ctypes.c_int, [_vp, ctypes.c_char_p, _lp]),
    "rmb_set_positions": (ctypes.c_int, [_vp, _vp, ctypes.c_long, ctypes.c_double, _vp, ctypes.c_int]),
    "rmb_set_positions_device": (ctypes.c_int, [_vp, _vp, ctypes.c_long, ctypes.c_double, _vp, ctypes.c_int]),
    "rmb_set_target_range": (ctypes.c_int, [_vp, ctypes.c_long, ctypes.c_long]),
    "rmb_matvec": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_double, _vp]),
    "rmb_matvec_device": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_double, _vp]),
    "rmb_matvec2_device": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, ctypes.c_double, _vp, _vp]),
    "rmb_matvec2_pairshard_device": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, ctypes.c_double, _vp, _vp, ctypes.c_long,
                                                    ctypes.c_long]),
    "rmb_matvec_pairshard_device": (ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.c_double, _vp, ctypes.c_long,
                                                   ctypes.c_long]),
    "rmb_body_mobility_dense_device": (ctypes.c_int, [_vp, _vp, ctypes.c_long, ctypes.c_int, ctypes.c_double, _vp]),
    "rmb_rigid_configuration_device": (ctypes.c_int, [_vp, ctypes.c_long, ctypes.c_long, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rmb_rigid_advance_device": (ctypes.c_int, [_vp, ctypes.c_long, _vp, _vp, _vp, ctypes.c_double, _vp, _vp, _vp]),
    "rmb_rigid_preconditioner_device": (ctypes.c_int, [_vp, ctypes.c_long, ctypes.c_long] + [_vp] * 11),
    "rmb_block_apply_device": (ctypes.c_int, [_vp, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_long, _vp, _vp,
                                              _vp, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double, _vp, ctypes.c_double, _vp]),
    "rmb_krylov_orthogonalize_device": (ctypes.c_int, [_vp, ctypes.c_long, ctypes.c_long, _vp, ctypes.c_long, _vp, _vp, _vp]),
    "rmb_krylov_orthogonalize2_device": (ctypes.c_int, [_vp, ctypes.c_long, ctypes.c_long, _vp, ctypes.c_long, _vp, _vp, _vp, _vp]),
    "rmb_host_mapped_alloc": (ctypes.c_int, [ctypes.c_size_t, ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    "rmb_host_mapped_free": (ctypes.c_int, [_vp]),
    "rmb_rigid_operator_device": (ctypes.c_int, [_vp, ctypes.c_long, ctypes.c_long, _vp, _vp, ctypes.c_double, _vp]),
    "rmb_rigid_gmres_device": (ctypes.c_int, [_vp, ctypes.c_long, ctypes.c_long, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_double, ctypes.c_long,
                                             ctypes.c_long, ctypes.c_double, _vp, _lp, _dp, _lp, _lp, _dp, ctypes.c_long, _dp]),
    "rmb_rigid_lanczos_device": (ctypes.c_int, [_vp, ctypes.c_long, ctypes.c_long, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double, ctypes.c_long,
                                               ctypes.c_long, ctypes.c_double, _vp, _lp, _lp, ctypes.POINTER(ctypes.c_int)]),
    "rmb_lanczos_device": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_double, ctypes.c_double, ctypes.c_long, ctypes.c_long,
                                         ctypes.c_double, _vp, _lp, _lp, ctypes.POINTER(ctypes.c_int)]),
    "rmb_lanczos_noise_coefficients": (ctypes.c_int, [ctypes.c_long, _dp, _dp, ctypes.c_double, _dp]),
    "rmb_rigid_lanczos_step_device": (ctypes.c_int, [_vp, ctypes.c_long, ctypes.c_long, _vp, _vp, ctypes.c_long, ctypes.c_long, ctypes.c_double,
                                                    _vp, _vp, _vp, _vp]),
    "rmb_rigid_arnoldi_step_device": (ctypes.c_int, [_vp, ctypes.c_long, ctypes.c_long, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_long,
                                                    ctypes.c_long, ctypes.c_double, _vp, _vp, _vp, _vp]),
    "rmb_matvec_op_device": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, _vp,
                                            ctypes.c_double]),
    "rmb_matvec_op_pairshard_device": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, _vp,
                                                      ctypes.c_double, ctypes.c_long, ctypes.c_long]),
    "rmb_blob_blob_force": (ctypes.c_int, [_vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, _vp]),
    "rmb_blob_blob_force_device": (ctypes.c_int, [_vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, _vp]),
    "rmb_one_blob_force_device": (ctypes.c_int, [_vp, ctypes.c_long, _vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                                ctypes.c_int, _vp]),
    "rmb_blob_blob_force_pairshard_device": (ctypes.c_int, [_vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, _vp,
                                                            ctypes.c_long, ctypes.c_long]),
    "rmb_blob_blob_force_radii": (ctypes.c_int, [_vp, _vp, ctypes.c_double, ctypes.c_double, _vp]),
    "rmb_blob_blob_force_radii_device": (ctypes.c_int, [_vp, _vp, ctypes.c_double, ctypes.c_double, _vp]),
    "rmb_timing_collect": (ctypes.c_int, [_vp, _dp, ctypes.c_int]),
    "rmb_timing_reset": (ctypes.c_int, [_vp]),
    "rmb_ubench_fp64_issue": (ctypes.c_int, [_vp, ctypes.c_int, _dp]),
    "rmb_wave_clock_collect": (ctypes.c_int, [_vp, _vp, ctypes.c_long]),
    "rmb_last_host_timing": (ctypes.c_int, [_vp, _dp]),
    "rmb_last_launch": (ctypes.c_int, [_vp, _lp, _lp, _lp]),
    "rmb_ctx_synchronize": (ctypes.c_int, [_vp]),
    "rmb_default_ctx_set_option": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_long]),
    "rmb_default_ctx_set_device": (ctypes.c_int, [ctypes.c_int]),
    "rmb_multi_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(_vp)]),
    "rmb_multi_destroy": (ctypes.c_int, [_vp]),
    "rmb_multi_n_shards": (ctypes.c_int, [_vp]),
    "rmb_multi_shard_ctx": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(_vp)]),
    "rmb_multi_set_stream": (ctypes.c_int, [_vp, _vp]),
    "rmb_multi_set_option": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_long]),
    "rmb_multi_get_option": (ctypes.c_int, [_vp, ctypes.c_char_p, _lp]),
    "rmb_multi_set_positions": (ctypes.c_int, [_vp, _vp, ctypes.c_long, ctypes.c_double, _vp, ctypes.c_int]),
    "rmb_multi_set_positions_device": (ctypes.c_int, [_vp, _vp, ctypes.c_long, ctypes.c_double, _vp, ctypes.c_int]),
    "rmb_multi_matvec": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_double, _vp]),
    "rmb_multi_matvec_device": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_double, _vp]),
    "rmb_multi_matvec_op_device": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, _vp,
                                                  ctypes.c_double]),
    "rmb_multi_blob_blob_force": (ctypes.c_int, [_vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, _vp]),
    "rmb_multi_blob_blob_force_device": (ctypes.c_int, [_vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, _vp]),
    "rmb_multi_synchronize": (ctypes.c_int, [_vp]),
    "rmb_mobility_oneshot": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long, _vp, _vp, _vp,
                                            ctypes.c_double, ctypes.c_double, _vp, _vp]),
    "rmb_mobility_source_target": (ctypes.c_int, [ctypes.c_long, _vp, _vp, ctypes.c_long, _vp, _vp, _vp, ctypes.c_double, _vp,
                                                  ctypes.c_int, _vp]),
    "rmb_mobility_source_target_device": (ctypes.c_int, [_vp, ctypes.c_long, _vp, _vp, ctypes.c_long, _vp, _vp, _vp,
                                                         ctypes.c_double, _vp, ctypes.c_int, _vp]),
    "rmb_forces_oneshot": (ctypes.c_int, [ctypes.c_long, _vp, _vp, ctypes.c_double, ctypes.c_double,
                                          ctypes.c_double, _vp]),
    "rmb_pressure_stokeslet": (ctypes.c_int, [ctypes.c_long, _vp, ctypes.c_long, _vp, _vp, _vp, ctypes.c_int, _vp]),
    "rmb_pressure_stokeslet_device": (ctypes.c_int, [_vp, ctypes.c_long, _vp, ctypes.c_long, _vp, _vp, _vp, ctypes.c_int, _vp]),
    "rmb_double_layer": (ctypes.c_int, [ctypes.c_long, _vp, ctypes.c_long, _vp, _vp, _vp, _vp, ctypes.c_int,
                                        ctypes.c_double, _vp]),
    "rmb_double_layer_device": (ctypes.c_int, [_vp, ctypes.c_long, _vp, ctypes.c_long, _vp, _vp, _vp, _vp, ctypes.c_int,
                                               ctypes.c_double, _vp]),
}



class Block(ctypes.Structure):
  """rmb_block: p[b batch_stride + row row_stride + col col_stride] (strides in doubles)."""
  _fields_ = [("p", ctypes.c_void_p), ("batch_stride", ctypes.c_long), ("row_stride", ctypes.c_long), ("col_stride", ctypes.c_long)]


_lib = None


class RmbError(RuntimeError):
  pass


def _preload_torch_hip_runtime():
  """One HIP runtime per process.  The PyTorch wheel bundles its own libamdhip64.so (+ HSA) and asks
  for it by file name, librmb_mobility.so asks for the SONAME libamdhip64.so.7: if ours pulls in
  /opt/rocm's copy first, a later `import torch` loads a SECOND runtime, which then finds no GPU.
  So when a torch installation is present, map its runtime first (without importing torch);
  librmb_mobility.so and torch then share it.  Without torch the system runtime is used."""
  import importlib.util
  try:
    spec = importlib.util.find_spec("torch")
  except (ImportError, ValueError):
    spec = None
  if spec is None or not spec.submodule_search_locations:
    return
  cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
  if os.path.exists(cand):
    try:
      ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except OSError:
      pass


def load():
  """Load the shared library (no GPU needed for loading / symbol checks)."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise RmbError(
          "HIP extension %s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
          "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    _preload_torch_hip_runtime()
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
      fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
      fn.restype = res
      fn.argtypes = args
    _lib = lib
  return _lib


def check(rc):
  if rc != 0:
    msg = load().rmb_last_error()
    raise RmbError("librmb_mobility error %d: %s" % (rc, msg.decode() if msg else "?"))


def device_count():
  return int(load().rmb_device_count())
