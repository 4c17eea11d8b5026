"""Worker for test_distributed_gloo.py: world_size ranks over gloo on CPU.  The compute backend is
an ORACLE-backed stand-in (tests may use the oracle); what is under test is the sharding, the
block-padded all-gather and the target-range bookkeeping of rigidmultiblobswall_amd.distributed."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle  # noqa: E402
from rigidmultiblobswall_amd.distributed import ShardedMobility, partition  # noqa: E402


class OracleBackend(object):
  def set_positions(self, r_full, a, L, wall):
    self.r = r_full.cpu().numpy().reshape(-1, 3).copy()
    self.a, self.L, self.wall = a, np.asarray(L), wall
    self.range = (0, len(self.r))

  def set_target_range(self, b, e):
    self.range = (b, e)

  def matvec(self, kind, v_full, eta, vec2_full=None, in_plane=False, out=None):
    names = {"tt": "trans_times_force", "tr": "trans_times_torque", "rt": "rot_times_force", "rr": "rot_times_torque"}
    pre = "single_wall" if self.wall else "no_wall"
    fn = getattr(oracle, "%s_mobility_%s_oracle" % (pre, names[kind]))
    u = fn(self.r, v_full.cpu().numpy(), eta, self.a, periodic_length=self.L)
    b, e = self.range
    return torch.from_numpy(u[3 * b:3 * e].copy())


def main():
  dist.init_process_group("gloo")
  rank, world = dist.get_rank(), dist.get_world_size()
  out_dir = sys.argv[1]
  for case, N in enumerate((103, 64, 3)):
    rng = np.random.RandomState(100 + case)   # same on every rank
    a, eta = 0.3, 1.7
    r = rng.rand(N, 3) * 4 + np.array([0, 0, 0.2])
    v = rng.randn(N, 3)
    sm = ShardedMobility(OracleBackend(), device="cpu")
    b, e, _ = partition(N, world, rank)
    sm.set_local_positions(r[b:e], N, a, wall=True)
    u_local = sm.matvec_local("tt", v[b:e].reshape(-1), eta)
    assert u_local.numel() == 3 * (e - b)
    u_full = sm.matvec("rr", v.reshape(-1), eta)
    if rank == 0:
      ref_tt = oracle.single_wall_mobility_trans_times_force_oracle(r, v, eta, a)
      ref_rr = oracle.single_wall_mobility_rot_times_torque_oracle(r, v, eta, a)
      np.savez(os.path.join(out_dir, "case%d.npz" % case), u_full=u_full.numpy(), ref_rr=ref_rr,
               u_local0=u_local.numpy(), ref_tt_local0=ref_tt[3 * b:3 * e])
    # every rank's local block must equal the matching slice of the single-process result
    ref_tt = oracle.single_wall_mobility_trans_times_force_oracle(r, v, eta, a)
    err = np.abs(u_local.numpy() - ref_tt[3 * b:3 * e]).max() if e > b else 0.0
    t = torch.tensor([err], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() < 1e-13, t.item()
  dist.destroy_process_group()


if __name__ == "__main__":
  main()
