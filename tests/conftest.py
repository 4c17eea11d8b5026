import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
  config.addinivalue_line("markers", "bench_harness: self-tests of bench.py's launcher / watchdog (collected last)")


# Collection order of the GPU run.  The driver runs `pytest -m gpu -x`: whatever comes first decides what a failure can
# hide.  Product parity against the oracle / the reference's fixtures goes first, the callers built on the path next, the
# multi-rank rehearsals after them and the self-tests of bench.py's harness (marker bench_harness) LAST -- a harness test
# must never again stand in front of the parity tests (round 4: one of them stopped the driver's run at test 34 of 593).
_ORDER = ["test_gpu_parity", "test_gpu_ops", "test_gpu_boundary", "test_aux_operators", "test_gpu_physics", "test_gpu_utilities",
          "test_gpu_krylov", "test_gpu_rigid", "test_gpu_rigid_integrator", "test_gpu_rollers", "test_gpu_multi",
          "test_gpu_config5", "test_gpu_distributed"]


def pytest_collection_modifyitems(config, items):
  def key(item):
    mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    harness = item.get_closest_marker("bench_harness") is not None
    rank = _ORDER.index(mod) if mod in _ORDER else len(_ORDER) - 1      # unknown modules: before the multi-rank ones
    return (2 if harness else (1 if item.get_closest_marker("gpu") is not None else 0), rank)
  items.sort(key=key)       # stable: the order inside a module is kept


def golden_files(pattern):
  return sorted(glob.glob(os.path.join(GOLDEN, pattern)))


def load_golden(path):
  d = np.load(path)
  return {k: d[k] for k in d.files}


def rel_err(u, ref):
  u = np.asarray(u).reshape(-1)
  ref = np.asarray(ref).reshape(-1)
  nrm = np.linalg.norm(ref)
  return np.linalg.norm(u - ref) / (nrm if nrm > 0 else 1.0)


# fixture key -> (kind, wall, in_plane) and the reference-named function stem
KERNEL_KEYS = {
    "no_wall_tt": "no_wall_mobility_trans_times_force",
    "wall_tt": "single_wall_mobility_trans_times_force",
    "in_plane_tt": "in_plane_mobility_trans_times_force",
    "no_wall_tr": "no_wall_mobility_trans_times_torque",
    "wall_tr": "single_wall_mobility_trans_times_torque",
    "in_plane_tr": "in_plane_mobility_trans_times_torque",
    "no_wall_rt": "no_wall_mobility_rot_times_force",
    "wall_rt": "single_wall_mobility_rot_times_force",
    "no_wall_rr": "no_wall_mobility_rot_times_torque",
    "wall_rr": "single_wall_mobility_rot_times_torque",
    "free_surface_tt": "free_surface_mobility_trans_times_force",
}


@pytest.fixture(scope="session")
def oracle():
  from oracle import oracle as o
  o.build()
  return o
