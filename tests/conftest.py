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
