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
    if kind == "tt_tr":
      u = getattr(oracle, pre + "_mobility_trans_times_force_torque_oracle")(
          self.r, v_full.cpu().numpy(), vec2_full.cpu().numpy(), eta, self.a, periodic_length=self.L)
    else:
      fn = getattr(oracle, "%s_mobility_%s_oracle" % (pre, names[kind]))
      u = fn(self.r, v_full.cpu().numpy(), eta, self.a, periodic_length=self.L)
    b, e = self.range
    return torch.from_numpy(u[3 * b:3 * e].copy())

  def blob_blob_force(self, eps, b, a, out=None):
    F = oracle.calc_blob_blob_forces_oracle(self.r, periodic_length=self.L, repulsion_strength=eps, debye_length=b,
                                            blob_radius=a)
    lo, hi = self.range
    return torch.from_numpy(np.ascontiguousarray(F[lo:hi]).reshape(-1))

  def body_mobility_dense(self, first_blob, n_b, eta, out=None):
    M = [oracle.dense("tt", int(self.wall), self.r[f:f + n_b], eta, self.a) for f in first_blob.tolist()]
    return torch.from_numpy(np.array(M))

  # pair-shard stand-in: "shard g" = the contribution of source block g to all targets (self terms of
  # block g included).  Like the HIP kernel's slices of unordered pairs, the shards sum to M.v.
  def supports_pairshard(self, kind, periodic):
    return kind in ("tt", "tr") and not periodic

  def matvec_pairshard(self, kind, v_full, eta, shard, nshards, out=None):
    n = len(self.r)
    b, e, _ = partition(n, nshards, shard)
    v = np.zeros(3 * n)
    v[3 * b:3 * e] = v_full.cpu().numpy()[3 * b:3 * e]
    pre = "single_wall" if self.wall else "no_wall"
    name = {"tt": "trans_times_force", "tr": "trans_times_torque"}[kind]
    u = getattr(oracle, "%s_mobility_%s_oracle" % (pre, name))(self.r, v, eta, self.a, periodic_length=self.L)
    return torch.from_numpy(u.copy())


def _matvec2_pairshard(self, kind, va, vb, eta, shard, nshards, out_a=None, out_b=None):
  """Two-vector pair shard of the stand-in: both partials, written into the caller's (stacked) buffers."""
  a = self.matvec_pairshard(kind, va, eta, shard, nshards)
  b = self.matvec_pairshard(kind, vb, eta, shard, nshards)
  out_a.copy_(a)
  out_b.copy_(b)
  return out_a, out_b


OracleBackend.matvec2_pairshard = _matvec2_pairshard


def rollers_replicated(rank, world, out_dir):
  """A replicated time stepper over sharded sweeps: RollersIntegrator on a ReplicatedContext must walk the
  reference trajectory (golden g8) on every rank, and all ranks must hold identical locations."""
  from rigidmultiblobswall_amd.distributed import ReplicatedContext
  sys.path.insert(0, os.path.join(ROOT, "tests"))
  from _rollers_common import integrator_from_golden, run_and_compare
  for name in ("g8_rollers_stoch_ab", "g8_rollers_det_ab_periodic"):
    d = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    g = {k: d[k] for k in d.files}
    ctx = ReplicatedContext(ShardedMobility(OracleBackend(), device="cpu"))
    integ = integrator_from_golden(g, ctx, "cpu")
    worst = run_and_compare(g, integ)
    assert worst < (1e-11 if float(g["kT"]) == 0.0 else 1e-7), (name, worst)
    mine = integ.location.clone()
    hi = mine.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert torch.equal(mine, hi), "ranks diverged"
    if rank == 0:
      np.save(os.path.join(out_dir, name + "_final.npy"), mine.numpy())


def rigid_replicated(rank, world, out_dir):
  """RigidIntegrator (GMRES + preconditioned Lanczos + forces) on a ReplicatedContext: the deck the reference's
  driver was run on (golden g9) must replay on every rank, bit-identically across ranks."""
  import tempfile
  from rigidmultiblobswall_amd.distributed import ReplicatedContext
  sys.path.insert(0, os.path.join(ROOT, "tests"))
  from _rigid_common import replay
  d = np.load(os.path.join(ROOT, "tests", "golden", "g9_rigid_stoch_slip_trapz.npz"))
  g = {k: d[k] for k in d.files}
  ctx = ReplicatedContext(ShardedMobility(OracleBackend(), device="cpu"))
  integ, worst_x, worst_q = replay(g, tempfile.mkdtemp(prefix="rank%d_" % rank), "cpu", ctx)
  assert worst_x < 1e-6 and worst_q < 1e-6, (worst_x, worst_q)
  mine = torch.cat([integ.location.reshape(-1), integ.orientation.reshape(-1)])
  hi = mine.clone()
  dist.all_reduce(hi, op=dist.ReduceOp.MAX)
  assert torch.equal(mine, hi), "ranks diverged"


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
    # replicated-vector API: tt goes through pair sharding + all-reduce, rr through target sharding + all-gather
    ref_tt_all = oracle.single_wall_mobility_trans_times_force_oracle(r, v, eta, a)
    ref_rr_all = oracle.single_wall_mobility_rot_times_torque_oracle(r, v, eta, a)
    u_rep_tt = sm.matvec_replicated("tt", v.reshape(-1), eta).numpy()
    u_rep_rr = sm.matvec_replicated("rr", v.reshape(-1), eta).numpy()
    assert u_rep_tt.shape == (3 * N,) and u_rep_rr.shape == (3 * N,)
    assert np.abs(u_rep_tt - ref_tt_all).max() <= 1e-13 * max(1.0, np.abs(ref_tt_all).max())
    assert np.abs(u_rep_rr - ref_rr_all).max() <= 1e-13 * max(1.0, np.abs(ref_rr_all).max())
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
  rollers_replicated(rank, world, out_dir)
  rigid_replicated(rank, world, out_dir)
  dist.destroy_process_group()


if __name__ == "__main__":
  main()
