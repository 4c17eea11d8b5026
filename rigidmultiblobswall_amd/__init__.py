"""MI355X-native blob-mobility engine: RPY + single-wall M.f products and blob-blob forces.

Layout (only what the hot path needs):
  csrc/         HIP kernels (*_kernels.h, pair_*.h) + C ABI in separately compiled units (rmb_internal.h lists them)
                -> librmb_mobility.so
  _lib.py       ctypes binding of include/rmb_mobility.h (no CPU fallback)
  context.py    persistent per-GPU context (resident positions, host + device entry points)
  mobility.py   the reference's mobility/mobility.py function surface, `<impl> = hip`
  forces.py     calc_blob_blob_forces_hip
  distributed.py  pair / target sharding over torch.distributed (RCCL) + ReplicatedContext for the callers
callers built on the path, each mirroring the reference module of the same role:
  rigid.py             RigidSuspension: saddle-point operator, block-diagonal preconditioner, GMRES
  rigid_integrator.py  RigidIntegrator: deterministic / Brownian schemes for rigid multiblobs, deck driver
  rollers.py           RollersIntegrator: single-blob roller schemes, deck driver
  stochastic.py        Lanczos M^{1/2} z (+ dense forcings)
  utilities.py         one-shot mobility / resistance / body_mobility problems, velocity field
  read_input.py, structures.py, dispatch.py   decks, .vertex / .clones / .slip files, backend strings
  __main__.py          python -m rigidmultiblobswall_amd --input-file deck
"""
from . import _lib  # noqa: F401
from .context import MobilityContext  # noqa: F401
from . import mobility  # noqa: F401
from . import forces  # noqa: F401

__all__ = ["MobilityContext", "mobility", "forces"]
