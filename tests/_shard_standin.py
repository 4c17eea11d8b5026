"""In-process stand-in for a G-rank run: the per-rank launches of every rank executed one after the other on ONE
MobilityContext and combined the way the collectives would (sum of the pair-shard partials = all-reduce; concatenated
target blocks = all-gather).  TEST infrastructure: lets a full-size step run through exactly the kernels and launch
geometries of a G-GPU run on the single GPU of the test box.  (tests/test_gpu_distributed.py runs real ranks over gloo
at small sizes; this covers BASELINE configs[4] sizes, where G processes sharing one card would be the bottleneck.)"""
import torch

from rigidmultiblobswall_amd.distributed import HipBackend, ShardedMobility, ReplicatedContext, partition


class SequentialShardsBackend(HipBackend):
  def __init__(self, device, nshards):
    HipBackend.__init__(self, device)
    self.G = int(nshards)
    self.launches = 0

  def matvec_pairshard(self, kind, v_full, eta, shard, nshards, out=None):
    tot = None
    for g in range(self.G):
      part = self.ctx.matvec_pairshard_device(kind, v_full, eta, g, self.G)
      tot = part if tot is None else tot.add_(part)
      self.launches += 1
    if out is not None:
      out.copy_(tot)
      return out
    return tot

  def matvec_op_pairshard(self, op, vecs, eta, shard, nshards, in_plane=False, outs=None):
    tot = None
    for g in range(self.G):
      parts = self.ctx.matvec_op_device(op, vecs, eta, in_plane=in_plane, shard=g, nshards=self.G)
      tot = list(parts) if tot is None else [t.add_(p) for t, p in zip(tot, parts)]
      self.launches += 1
    if outs is not None:
      for o, t in zip(outs, tot):
        o.copy_(t)
      return tuple(outs)
    return tuple(tot)

  def matvec2_pairshard(self, kind, va, vb, eta, shard, nshards, out_a=None, out_b=None):
    ta = tb = None
    for g in range(self.G):
      pa, pb = self.ctx.matvec2_device(kind, va, vb, eta, shard=g, nshards=self.G)
      ta, tb = (pa, pb) if ta is None else (ta.add_(pa), tb.add_(pb))
      self.launches += 1
    if out_a is not None:
      out_a.copy_(ta); ta = out_a
    if out_b is not None:
      out_b.copy_(tb); tb = out_b
    return ta, tb

  def _over_target_blocks(self, fn):
    n = self.ctx.n
    blocks = []
    for g in range(self.G):
      b, e, _ = partition(n, self.G, g)
      self.ctx.set_target_range(b, e)
      if e > b:
        blocks.append(fn().clone())
      self.launches += 1
    self.ctx.set_target_range(0, n)
    return torch.cat(blocks)

  def blob_blob_force_pairshard(self, eps, b, a, shard, nshards, out=None):
    tot = None
    for g in range(self.G):
      part = self.ctx.blob_blob_force_pairshard_device(eps, b, a, g, self.G, device=self.device)
      tot = part if tot is None else tot.add_(part)
      self.launches += 1
    return tot

  def blob_blob_force(self, eps, b, a, out=None):
    return self._over_target_blocks(lambda: self.ctx.blob_blob_force_device(eps, b, a, device=self.device))

  def matvec(self, kind, v_full, eta, vec2_full=None, in_plane=False, out=None):
    return self._over_target_blocks(lambda: self.ctx.matvec_device(kind, v_full, eta, vec2=vec2_full, in_plane=in_plane))

  def set_target_range(self, begin, end):
    pass          # the stand-in plays every rank in turn


def replicated_standin(device, nshards):
  backend = SequentialShardsBackend(device, nshards)
  return ReplicatedContext(ShardedMobility(backend, device=device)), backend
