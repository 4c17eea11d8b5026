"""MI355X-native blob-mobility engine: RPY + single-wall M.f products and blob-blob forces.

Layout (only what the hot path needs):
  csrc/         HIP kernels (pair_ops.h, matvec_kernels.h) + C ABI (rmb_capi.hip) -> librmb_mobility.so
  _lib.py       ctypes binding of include/rmb_mobility.h (no CPU fallback)
  context.py    persistent per-GPU context (resident positions, host + device entry points)
  mobility.py   the reference's mobility/mobility.py function surface, `<impl> = hip`
  forces.py     calc_blob_blob_forces_hip
  distributed.py  target sharding + all-gather of sources over torch.distributed (RCCL)
"""
from . import _lib  # noqa: F401
from .context import MobilityContext  # noqa: F401
from . import mobility  # noqa: F401
from . import forces  # noqa: F401

__all__ = ["MobilityContext", "mobility", "forces"]
