"""Worker of tests/test_gpu_distributed.py: G ranks share cuda:0 over gloo (RCCL refuses several ranks on one device)
and run the REAL per-rank HIP launches of both decompositions -- pair shard `rank` of `world` + all-reduce, target
block + all-gather -- through ShardedMobility / ReplicatedContext; every rank checks the result against the
single-context product computed on the same device.  Exit code != 0 on any mismatch."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from rigidmultiblobswall_amd import MobilityContext            # noqa: E402
from rigidmultiblobswall_amd.distributed import HipBackend, ShardedMobility, ReplicatedContext, partition  # noqa: E402

TOL = 1e-13


def rel(a, b):
  return float(torch.linalg.norm(a - b) / torch.linalg.norm(b))


def main():
  # RMB_DIST_BACKEND=nccl with ONE rank: the same collectives go through RCCL (always_exchange: a one-rank group skips
  # them by default); several ranks on one device need gloo
  backend = os.environ.get("RMB_DIST_BACKEND", "gloo")
  dev = torch.device("cuda:0")
  torch.cuda.set_device(dev)
  if backend == "nccl":
    dist.init_process_group("nccl", device_id=dev)
  else:
    dist.init_process_group(backend)
  rank, world = dist.get_rank(), dist.get_world_size()
  sm = ShardedMobility(HipBackend(dev), device=dev, always_exchange=backend == "nccl")
  assert sm.exchange
  rc = ReplicatedContext(sm)
  single = MobilityContext(0)
  rng = np.random.RandomState(5)           # same numbers on every rank
  checked = 0
  for N, L, wall in ((1000, None, True), (333, (11.0, 12.0, 0.0), True), (90, None, True), (500, (0.0, 9.0, 10.0), False)):
    a, eta = 0.4, 1.3
    r = rng.rand(N, 3) * (N / 0.08) ** (1.0 / 3.0) * a * 1.6
    r[:, 2] += 1.1 * a
    vs = [torch.as_tensor(rng.randn(3 * N), device=dev) for _ in range(3)]
    rd = torch.as_tensor(r.reshape(-1), device=dev)
    single.set_positions(rd, a, L, wall=wall)
    # --- pair shard + all-reduce (replicated vectors) ---
    rc.set_positions(rd, a, L, wall)
    for kind in ("tt", "tr", "rt", "rr"):
      u = rc.matvec_device(kind, vs[0], eta)
      assert rel(u, single.matvec_device(kind, vs[0], eta)) < TOL, (N, kind)
      checked += 1
    u = rc.matvec_device("tt_tr", vs[0], eta, vec2=vs[1])
    assert rel(u, single.matvec_device("tt_tr", vs[0], eta, vec2=vs[1])) < TOL
    for op, ins in (("grand", vs[:2]), ("force_column", vs[:1]), ("tt_multi", vs[:3]), ("velocity_from_force_torque", vs[:2])):
      got = rc.matvec_op_device(op, ins, eta)
      ref = single.matvec_op_device(op, ins, eta)
      for g, s in zip(got, ref):
        assert rel(g, s) < TOL, (N, op)
        checked += 1
    # bit-reproducible multi-rank products: every rank's shard is a fixed-order reduction of whole tile pairs
    rc.set_option("deterministic", 2)
    for kind in ("tt", "rr"):
      u1 = rc.matvec_device(kind, vs[0], eta).clone()
      u2 = rc.matvec_device(kind, vs[0], eta)
      assert torch.equal(u1, u2), (N, kind, "not bit-reproducible")
      assert rel(u1, single.matvec_device(kind, vs[0], eta)) < TOL, (N, kind)
    g1 = [x.clone() for x in rc.matvec_op_device("grand", vs[:2], eta)]
    g2 = rc.matvec_op_device("grand", vs[:2], eta)
    assert all(torch.equal(x, y) for x, y in zip(g1, g2))
    rc.set_option("deterministic", 0)
    checked += 3
    ua, ub = rc.matvec2_device("tt", vs[0], vs[1], eta)
    sa, sb = single.matvec2_device("tt", vs[0], vs[1], eta)
    assert rel(ua, sa) < TOL and rel(ub, sb) < TOL
    # in-plane: target shard + all-gather behind the replicated facade
    if wall:
      u = rc.matvec_device("tt", vs[0], eta, in_plane=True)
      assert rel(u, single.matvec_device("tt", vs[0], eta, in_plane=True)) < 1e-12
    # --- target shard + all-gather of the source blocks (north_star's layout) ---
    b, e, _ = partition(N, world, rank)
    sm.set_local_positions(rd[3 * b:3 * e], N, a, L, wall)
    u_loc = sm.matvec_local("tt", vs[0][3 * b:3 * e], eta)
    ref = single.matvec_device("tt", vs[0], eta)
    if e > b:
      assert rel(u_loc, ref[3 * b:3 * e]) < 1e-12, (N, "target shard")
    checked += 1
    # forces: pair shard of the symmetric force kernel + all-reduce; free surface: pair shard + all-reduce
    single.set_positions(rd, a, L, wall=False)
    rc.set_positions(rd, a, L, False)
    F = rc.blob_blob_force_device(0.7, 0.15, a)
    assert rel(F, single.blob_blob_force_device(0.7, 0.15, a)) < 1e-12
    # with "deterministic" on, multi-rank forces are bit-reproducible: own target block (one-sided sweep) + all-gather
    for mode in (1, 2):
      rc.set_option("deterministic", mode)
      F1 = rc.blob_blob_force_device(0.7, 0.15, a).clone()
      F2 = rc.blob_blob_force_device(0.7, 0.15, a)
      assert torch.equal(F1, F2), (N, mode, "forces not bit-reproducible")
      assert rel(F1.view(-1), F.view(-1)) < 1e-12
    rc.set_option("deterministic", 0)
    checked += 2
    u = rc.matvec_device("tt_free", vs[0], eta)
    assert rel(u, single.matvec_device("tt_free", vs[0], eta)) < 1e-12
    checked += 2
  torch.cuda.synchronize()
  dist.barrier()
  if rank == 0:
    print("gpu dist worker: %d checks on %d ranks ok (%s)" % (checked, world, dist.get_backend()), flush=True)
  single.close()
  dist.destroy_process_group()


if __name__ == "__main__":
  main()
