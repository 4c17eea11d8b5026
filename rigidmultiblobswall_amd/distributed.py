"""Blob mobility over several GPUs, one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI).

The reference is single-device; this layer has no counterpart there.  Its contract is "equal to
the 1-GPU result to rounding".  (One process driving several GPUs -- the reference's own call shape -- is
multi.MultiContext / rmb_multi_*.)  Two decompositions:

  * SYMMETRIC PAIR SHARDING (default of the replicated API: matvec_replicated, matvec_op_replicated,
    matvec2_replicated, blob_blob_force_replicated -- every product of the surface, open or pseudo-periodic): every
    rank holds all positions and the full source vector(s); rank g evaluates the g-th slice of the unordered blob
    pairs once each (rmb_matvec_pairshard_device & co.) into a full-length partial; ONE all-reduce (sum) of 24 N bytes
    per output vector completes the product on every rank.  Keeps the 2x arithmetic saving of the symmetric kernel at
    any G.  Bit-reproducible per rank with the context option "deterministic" = 2 (forces: own target block + all-gather).
  * TARGET SHARDING (the block-distributed API: set_local_positions / matvec_local, north_star's layout): blob index
    range [0, N) is cut into G contiguous blocks of ceil(N/G); rank g OWNS block g: it holds r_local, v_local and
    produces u_local.  Positions are all-gathered once per configuration, the source vector (24 N bytes in total)
    before each matvec; outputs need no reduction.  The per-rank kernel is the one-sided sweep restricted to the
    rank's targets (rmb_set_target_range): every ordered pair of its rows, 1.6x the work per pair of the default.

The compute backend is injected (`backend=`): the product default is the HIP context (`HipBackend`); the CPU
test-suite injects an oracle-backed one to exercise partitioning and collectives under gloo.  There is no implicit
CPU fallback.
"""
import numpy as np
import torch
import torch.distributed as dist

from .context import MobilityContext


def partition(n, world_size, rank):
  """Contiguous block partition: (begin, end, block) with block = ceil(n / world_size)."""
  block = (n + world_size - 1) // world_size if world_size > 0 else n
  begin = min(rank * block, n)
  end = min((rank + 1) * block, n)
  return begin, end, block


class HipBackend(object):
  """Per-rank compute on one MI355X through the C ABI."""

  def __init__(self, device):
    self.device = torch.device(device)
    idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
    self.ctx = MobilityContext(idx)   # device-path calls follow torch's current stream (context._follow_torch_stream)

  def set_positions(self, r_full, a, L, wall):
    self.ctx.set_positions(r_full, a, L, wall)

  def set_target_range(self, begin, end):
    self.ctx.set_target_range(begin, end)

  def matvec(self, kind, v_full, eta, vec2_full=None, in_plane=False, out=None):
    return self.ctx.matvec_device(kind, v_full, eta, vec2=vec2_full, in_plane=in_plane, out=out)

  def blob_blob_force(self, eps, b, a, out=None):
    return self.ctx.blob_blob_force_device(eps, b, a, out=out, device=self.device)

  def supports_pairshard(self, kind, periodic):
    return kind in ("tt", "tr", "rt", "rr", "tt_free")

  def blob_blob_force_pairshard(self, eps, b, a, shard, nshards, out=None):
    return self.ctx.blob_blob_force_pairshard_device(eps, b, a, shard, nshards, out=out, device=self.device)

  def matvec_pairshard(self, kind, v_full, eta, shard, nshards, out=None):
    return self.ctx.matvec_pairshard_device(kind, v_full, eta, shard, nshards, out=out)

  def body_mobility_dense(self, first_blob, n_b, eta, out=None):
    return self.ctx.body_mobility_dense_device(first_blob, n_b, eta, out=out)

  def matvec_op_pairshard(self, op, vecs, eta, shard, nshards, in_plane=False, outs=None):
    return self.ctx.matvec_op_device(op, vecs, eta, in_plane=in_plane, outs=outs, shard=shard, nshards=nshards)

  def matvec2_pairshard(self, kind, va, vb, eta, shard, nshards, out_a=None, out_b=None):
    return self.ctx.matvec2_device(kind, va, vb, eta, out_a=out_a, out_b=out_b, shard=shard, nshards=nshards)


class ShardedMobility(object):
  """M.v over the ranks of a process group: unordered pairs sharded + all-reduce (replicated API) or targets sharded +
  all-gather (block-distributed API); see the module docstring."""

  def __init__(self, backend, group=None, device=None, always_exchange=False):
    """always_exchange: issue the collectives of the multi-rank path in a one-rank group too (they are skipped there by
    default).  A one-GPU box can then run the all-reduce / all-gather / broadcast through RCCL exactly as a node does."""
    self.backend = backend
    self.group = group
    self.rank = dist.get_rank(group) if dist.is_initialized() else 0
    self.world = dist.get_world_size(group) if dist.is_initialized() else 1
    self.exchange = self.world > 1 or (bool(always_exchange) and dist.is_initialized())
    self.device = torch.device(device) if device is not None else torch.device("cpu")
    self.n = 0
    self.begin = self.end = self.block = 0
    self._gather_buf = None
    self._gather_buf2 = None
    self._periodic = False

  # -- helpers ----------------------------------------------------------------------------------
  def _all_gather_blocks(self, local_flat, buf):
    """local_flat: (3*n_local,) -> view (3*n,) of the gathered, block-padded buffer."""
    width = 3 * self.block
    if not self.exchange:  # nothing to exchange: the local block IS the full vector
      return local_flat, buf
    if buf is None or buf.numel() != width * self.world:
      buf = torch.empty(width * self.world, dtype=torch.float64, device=self.device)
    send = local_flat
    if local_flat.numel() != width:  # last (short or empty) block: pad so every rank sends `width`
      send = torch.zeros(width, dtype=torch.float64, device=self.device)
      send[:local_flat.numel()].copy_(local_flat)
    dist.all_gather_into_tensor(buf, send.contiguous(), group=self.group)
    # all blocks but the last are full, so the first 3n entries are the blobs in global order
    return buf[:3 * self.n], buf

  def _to_dev(self, x):
    if isinstance(x, torch.Tensor):
      return x.to(device=self.device, dtype=torch.float64).contiguous().view(-1)
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64).reshape(-1), device=self.device)

  # -- API --------------------------------------------------------------------------------------
  def set_local_positions(self, r_local, n_total, a, periodic_length=None, wall=True):
    """Each rank passes the positions of the blobs it owns (partition(n_total, world, rank))."""
    self.n = int(n_total)
    self.begin, self.end, self.block = partition(self.n, self.world, self.rank)
    r_local = self._to_dev(r_local)
    if r_local.numel() != 3 * (self.end - self.begin):
      raise ValueError("rank %d owns blobs [%d,%d) but got %d coordinates" %
                       (self.rank, self.begin, self.end, r_local.numel()))
    r_full, _ = self._all_gather_blocks(r_local, None)
    L = np.zeros(3) if periodic_length is None else np.asarray(periodic_length, dtype=np.float64)
    self._periodic = bool(np.any(L > 0))
    self.backend.set_positions(r_full.clone(), a, L, wall)
    self.backend.set_target_range(self.begin, self.end)

  def set_positions(self, r_full, a, periodic_length=None, wall=True):
    """Convenience: every rank passes the full (N,3) array; it keeps only its own block."""
    r = self._to_dev(r_full)
    n = r.numel() // 3
    b, e, _ = partition(n, self.world, self.rank)
    self.set_local_positions(r[3 * b:3 * e], n, a, periodic_length, wall)

  def matvec_local(self, kind, v_local, eta, vec2_local=None, in_plane=False, out=None):
    """v_local: this rank's block of the source vector -> this rank's block of M.v (device tensor)."""
    v_local = self._to_dev(v_local)
    v_full, self._gather_buf = self._all_gather_blocks(v_local, self._gather_buf)
    v2_full = None
    if vec2_local is not None:
      v2_full, self._gather_buf2 = self._all_gather_blocks(self._to_dev(vec2_local), self._gather_buf2)
    return self.backend.matvec(kind, v_full, eta, vec2_full=v2_full, in_plane=in_plane, out=out)

  def matvec(self, kind, v_full, eta, vec2_full=None, in_plane=False):
    """Full-vector convenience (every rank passes and receives all 3N entries): shards the input,
    runs matvec_local, all-gathers the output.  Used by host-surface callers and the tests."""
    v = self._to_dev(v_full)
    v2 = self._to_dev(vec2_full) if vec2_full is not None else None
    lo, hi = 3 * self.begin, 3 * self.end
    u_local = self.matvec_local(kind, v[lo:hi], eta, None if v2 is None else v2[lo:hi], in_plane)
    u_full, _ = self._all_gather_blocks(u_local.view(-1), None)
    return u_full.clone()

  def matvec_replicated(self, kind, v_full, eta, vec2_full=None, in_plane=False, out=None):
    """Every rank holds the full source vector and receives the full product (the layout a replicated
    Krylov loop wants: its dot products / axpys on 3N-vectors are negligible next to the O(N^2) sweep).
    Symmetric pair sharding for every kind the backend shards (tt / tr / rt / rr / free surface, the fused tt+tr row and
    the in-plane products; open or pseudo-periodic): rank g evaluates the g-th slice of the unordered pairs once each
    into a full-length partial, then ONE all-reduce (sum) of 24 N bytes.  Backends without the pair-shard entry points:
    target sharding on the rank's block, then all-gather of the blocks."""
    v = self._to_dev(v_full)
    periodic = bool(self._periodic)
    if (vec2_full is None and not in_plane and hasattr(self.backend, "supports_pairshard")
        and self.backend.supports_pairshard(kind, periodic)):
      part = self.backend.matvec_pairshard(kind, v, eta, self.rank, self.world, out=out)
      if self.exchange:
        dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)
      return part
    if (vec2_full is None and in_plane and kind in ("tt", "tr") and hasattr(self.backend, "matvec_op_pairshard")):
      # in-plane products are a row / column mask of the symmetric matrix: pair shard of the one-vector operation
      part = out if out is not None else torch.empty(3 * self.n, dtype=torch.float64, device=self.device)
      self.backend.matvec_op_pairshard(kind + "_multi", (v,), eta, self.rank, self.world, in_plane=True, outs=[part])
      if self.exchange:
        dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)
      return part
    v2 = self._to_dev(vec2_full) if vec2_full is not None else None
    if kind == "tt_tr" and v2 is not None and hasattr(self.backend, "matvec_op_pairshard"):
      # fused M_tt f + M_tr tau: one pass over this rank's pair shard for both blocks, one all-reduce
      part = out if out is not None else torch.empty(3 * self.n, dtype=torch.float64, device=self.device)
      self.backend.matvec_op_pairshard("velocity_from_force_torque", (v, v2), eta, self.rank, self.world, in_plane=in_plane,
                                       outs=[part])
      if self.exchange:
        dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)
      return part
    if (kind == "tt_tr" and v2 is not None and not in_plane and self.exchange
        and hasattr(self.backend, "supports_pairshard") and self.backend.supports_pairshard("tt", periodic)
        and self.backend.supports_pairshard("tr", periodic)):
      # fused M_tt f + M_tr tau: two pair-sharded symmetric passes, ONE all-reduce of the summed partials
      part = self.backend.matvec_pairshard("tt", v, eta, self.rank, self.world, out=out)
      part += self.backend.matvec_pairshard("tr", v2, eta, self.rank, self.world)
      dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)
      return part
    u_local = self.backend.matvec(kind, v, eta, vec2_full=v2, in_plane=in_plane)
    u_full, _ = self._all_gather_blocks(u_local.view(-1), None)
    return u_full if not self.exchange else u_full.clone()

  def blob_blob_force_local(self, eps, b, a):
    return self.backend.blob_blob_force(eps, b, a)

  # -- replicated layout: every rank holds full vectors (Krylov loops, time steppers) ---------------
  def set_replicated_positions(self, r_full, a, periodic_length=None, wall=True):
    """Every rank already holds all N positions (a replicated time stepper): no exchange at all."""
    r = self._to_dev(r_full)
    self.n = r.numel() // 3
    self.begin, self.end, self.block = partition(self.n, self.world, self.rank)
    L = np.zeros(3) if periodic_length is None else np.asarray(periodic_length, dtype=np.float64)
    self._periodic = bool(np.any(L > 0))
    self.backend.set_positions(r, a, L, wall)
    self.backend.set_target_range(self.begin, self.end)

  def matvec2_replicated(self, kind, va_full, vb_full, eta):
    """Two tt products with replicated vectors: one pass over this rank's pair shard with both vectors, then ONE
    all-reduce of the stacked partials (2 x 24 N bytes).  Backends without the two-vector kernel run two products."""
    va, vb = self._to_dev(va_full), self._to_dev(vb_full)
    if hasattr(self.backend, "matvec2_pairshard") and self.backend.supports_pairshard(kind, bool(self._periodic)):
      both = torch.empty((2, va.numel()), dtype=torch.float64, device=self.device)
      self.backend.matvec2_pairshard(kind, va, vb, eta, self.rank, self.world, out_a=both[0], out_b=both[1])
      if self.exchange:
        dist.all_reduce(both, op=dist.ReduceOp.SUM, group=self.group)
      return both[0], both[1]
    return self.matvec_replicated(kind, va, eta), self.matvec_replicated(kind, vb, eta)

  def matvec_op_replicated(self, op, vecs, eta, in_plane=False):
    """Multi-block operation (context.matvec_op_device: "velocity_from_force_torque", "grand", "force_column",
    "tt_multi" / "tr_multi" / "rt_multi" / "rr_multi") with replicated vectors: one pass over this rank's pair shard for all blocks, then ONE all-reduce of the
    stacked partial outputs.  Backends without the multi-block kernel compose it from single products."""
    vecs = [self._to_dev(v) for v in vecs]
    if hasattr(self.backend, "matvec_op_pairshard"):
      from ._lib import OPS
      n_out = OPS[op][2] if OPS[op][2] is not None else len(vecs)
      stacked = torch.empty((n_out, 3 * self.n), dtype=torch.float64, device=self.device)
      self.backend.matvec_op_pairshard(op, vecs, eta, self.rank, self.world, in_plane=in_plane,
                                       outs=[stacked[c] for c in range(n_out)])
      if self.exchange:
        dist.all_reduce(stacked, op=dist.ReduceOp.SUM, group=self.group)
      return tuple(stacked[c] for c in range(n_out))
    mv = lambda kind, v, v2=None: self.matvec_replicated(kind, v, eta, vec2_full=v2, in_plane=in_plane)
    if op == "velocity_from_force_torque":
      return (mv("tt_tr", vecs[0], vecs[1]),)
    if op == "grand":
      return (mv("tt_tr", vecs[0], vecs[1]), mv("rt", vecs[0]) + mv("rr", vecs[1]))
    if op == "force_column":
      return (mv("tt", vecs[0]), mv("rt", vecs[0]))
    if op.endswith("_multi") and op[:2] in ("tt", "tr", "rt", "rr"):
      return tuple(mv(op[:2], v) for v in vecs)
    raise ValueError("unknown operation %r" % (op,))

  def blob_blob_force_replicated(self, eps, b, a):
    """Forces on ALL blobs on every rank: pair shard of the symmetric force kernel (F_ji = -F_ij, each unordered pair
    once) + one all-reduce; with the "deterministic" option, and for backends without the pair shard, every rank sweeps
    its own target block (atomic-free, bit-reproducible) and the blocks are all-gathered."""
    # The pair shard flushes with atomics.  With the context's "deterministic" option on (1 or 2) the forces stay
    # bit-reproducible across runs: own target block with the one-sided sweep (fixed summation order) + all-gather.
    ctx = getattr(self.backend, "ctx", None)
    deterministic = bool(ctx.get_option("deterministic")) if ctx is not None and hasattr(ctx, "get_option") else False
    if hasattr(self.backend, "blob_blob_force_pairshard") and not deterministic:
      part = self.backend.blob_blob_force_pairshard(eps, b, a, self.rank, self.world)
      if self.exchange:
        dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)
      return part
    f_local = self.backend.blob_blob_force(eps, b, a)
    f_full, _ = self._all_gather_blocks(f_local.view(-1), None)
    return f_full if not self.exchange else f_full.clone()


class ReplicatedContext(object):
  """MobilityContext-shaped facade over ShardedMobility (set_positions / matvec_device /
  blob_blob_force_device / body_mobility_dense_device with FULL vectors on every rank), so the
  device-resident callers written against one GPU -- RigidSuspension (rigid.py), stochastic_forcing_lanczos
  (stochastic.py), RollersIntegrator (rollers.py) -- run unchanged on G GPUs: every rank executes the same
  O(N) host/Krylov logic on replicated vectors (identical on all ranks because all-reduce / all-gather
  return the same bits everywhere) and only the O(N^2) pair sweeps are divided.  Random numbers must come
  from identically seeded generators on every rank."""

  def __init__(self, sharded):
    self.sm = sharded
    self.n = 0

  @property
  def helper_context(self):
    """The rank's own MobilityContext, for the rank-local O(N) helper kernels of the callers (rigid.py: block products,
    fused Gram-Schmidt, per-body geometry and factors); None for backends without one."""
    return getattr(self.sm.backend, "ctx", None)

  def set_stream(self, stream_ptr):
    ctx = getattr(self.sm.backend, "ctx", None)
    if ctx is not None:
      ctx.set_stream(stream_ptr)

  def set_option(self, key, value):
    """Context options of the rank's own MobilityContext ("precision", "deterministic", ...); backends without one
    (the oracle stand-in of the host tests) ignore them."""
    ctx = getattr(self.sm.backend, "ctx", None)
    if ctx is not None:
      ctx.set_option(key, value)

  def get_option(self, key):
    """Current value of a context option, or None for backends without a MobilityContext."""
    ctx = getattr(self.sm.backend, "ctx", None)
    return ctx.get_option(key) if ctx is not None else None

  def sync_scalars(self, t):
    """Make rank 0's copy of a small control tensor (Hessenberg column, Lanczos coefficients) the one every rank acts
    on.  The replicated Krylov loops branch on such scalars; identical hardware and identical inputs already give
    identical values, this broadcast (a few bytes per iteration) turns that into a guarantee, so no rank can leave a
    loop one iteration early and strand the others in a collective."""
    if self.sm.exchange:
      dist.broadcast(t, src=dist.get_global_rank(self.sm.group, 0) if self.sm.group is not None else 0, group=self.sm.group)
    return t

  def set_positions(self, r_vectors, a, periodic_length=None, wall=True):
    self.sm.set_replicated_positions(r_vectors, a, periodic_length, wall)
    self.n = self.sm.n

  def matvec_device(self, kind, vec, eta, vec2=None, in_plane=False, out=None):
    return self.sm.matvec_replicated(kind, vec, eta, vec2_full=vec2, in_plane=in_plane, out=out)

  def matvec_op_device(self, op, vecs, eta, in_plane=False, outs=None, shard=0, nshards=1):
    res = self.sm.matvec_op_replicated(op, vecs, eta, in_plane=in_plane)
    if outs is not None:
      for o, r in zip(outs, res):
        o.copy_(r)
      return tuple(outs)
    return res

  def matvec2_device(self, kind, vec_a, vec_b, eta, out_a=None, out_b=None, shard=0, nshards=1):
    a, b = self.sm.matvec2_replicated(kind, vec_a, vec_b, eta)
    if out_a is not None:
      out_a.copy_(a); a = out_a
    if out_b is not None:
      out_b.copy_(b); b = out_b
    return a, b

  def blob_blob_force_device(self, repulsion_strength, debye_length, blob_radius, out=None, device=None):
    return self.sm.blob_blob_force_replicated(repulsion_strength, debye_length, blob_radius)

  def body_mobility_dense_device(self, first_blob, n_b, eta, out=None):
    # O(n_bodies n_b^2): replicated, not worth an exchange
    return self.sm.backend.body_mobility_dense(first_blob, n_b, eta, out=out)

  def close(self):
    ctx = getattr(self.sm.backend, "ctx", None)
    if ctx is not None:
      ctx.close()
