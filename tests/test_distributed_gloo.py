"""N>1 path on CPU: 2 and 3 ranks over gloo, sharded result == single-process result."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_matches_single_process(tmp_path, world):
  env = dict(os.environ, OMP_NUM_THREADS="2", MASTER_ADDR="127.0.0.1")
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
         os.path.join(ROOT, "tests", "_dist_worker.py"), str(tmp_path)]
  res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
  assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
  for case in range(3):
    d = np.load(os.path.join(str(tmp_path), "case%d.npz" % case))
    assert np.abs(d["u_full"] - d["ref_rr"]).max() <= 1e-13 * max(1.0, np.abs(d["ref_rr"]).max())
    assert np.abs(d["u_local0"] - d["ref_tt_local0"]).max() <= 1e-13 * max(1.0, np.abs(d["ref_tt_local0"]).max())
    if case == 0:
      for name in ("g8_rollers_stoch_ab", "g8_rollers_det_ab_periodic"):
        final = np.load(os.path.join(str(tmp_path), name + "_final.npy"))
        ref = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))["trajectory"][-1]
        assert np.abs(final - ref).max() < 1e-7
