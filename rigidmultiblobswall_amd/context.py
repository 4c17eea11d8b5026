"""Persistent device context: positions resident across the matvecs of a solve.

Semantic list of SURVEY.md section 8(b): create/destroy, set_positions (upload + height clamp + B
on device), matvec(kind), blob_blob_forces, timing.  The reference has no such object -- every
pycuda call re-allocates and re-uploads (mobility/mobility_pycuda.py:2249-2266).
"""
import ctypes

import numpy as np

from . import _lib


def _as_f64(x, n3=None):
  x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
  if n3 is not None and x.size != n3:
    raise ValueError("expected %d values, got %d" % (n3, x.size))
  return x


def _ptr(x):
  return ctypes.c_void_p(x.ctypes.data)


class MobilityContext(object):
  """One context = one GPU.  Host (numpy) and device (torch tensor) entry points."""

  def __init__(self, device=0):
    self._lib = _lib.load()
    h = ctypes.c_void_p()
    _lib.check(self._lib.rmb_ctx_create(int(device), ctypes.byref(h)))
    self._h = h
    self.device = int(device)
    self.n = 0
    self.n_targets = 0
    self.target_range = (0, 0)
    self._keepalive = None
    self._stream_handle = None  # hipStream_t the C context currently enqueues on
    self._user_stream = False   # True once set_stream() pinned a caller-owned stream
    self._options_set = {}      # every option set through this wrapper (launch_signature)
    self._geometry = None       # (n, a, L, wall) of the bound configuration

  def close(self):
    if getattr(self, "_h", None) is not None and self._h.value:
      self._lib.rmb_ctx_destroy(self._h)
      self._h = ctypes.c_void_p()

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass

  # --- configuration ---------------------------------------------------------------------------
  def set_stream(self, stream_ptr):
    """Pin the context to a caller-owned hipStream_t (otherwise device-path calls follow torch's current stream)."""
    _lib.check(self._lib.rmb_ctx_set_stream(self._h, ctypes.c_void_p(int(stream_ptr) if stream_ptr else 0)))
    self._user_stream = True

  def release_stream(self):
    """Call BEFORE destroying a stream this context is bound to: waits for the context's work on it and forgets the
    handle (rmb_ctx_release_stream).  The next device call follows torch's current stream again unless set_stream()
    pins another one."""
    _lib.check(self._lib.rmb_ctx_release_stream(self._h))
    self._stream_handle = 0
    self._user_stream = False

  # Ordering of the device path against PyTorch: every *_device call is enqueued on the stream that is
  # torch's CURRENT stream at call time (handle passed through the C ABI), so it is ordered after the
  # torch kernels that produced its inputs and before those that consume its outputs -- no events, no
  # host synchronisation.  (Measured alternative: a private stream + wait_stream fences costs ~24 us per
  # call, 11 % of a 1e4-blob matvec.)
  def _follow_torch_stream(self):
    if self._user_stream:
      return
    h = _current_raw_stream(self.device)
    if h != self._stream_handle:
      _lib.check(self._lib.rmb_ctx_set_stream(self._h, ctypes.c_void_p(h)))
      self._stream_handle = h

  def set_option(self, key, value):
    _lib.check(self._lib.rmb_ctx_set_option(self._h, key.encode(), int(value)))
    self._options_set[key] = int(value)

  def launch_signature(self):
    """Everything that decides WHICH launches a product of this context turns into, as far as this wrapper knows: the
    number of blobs, radius, box and wall of the bound configuration and every option set through set_option().  A
    captured hipGraph of device-path calls stays valid while this is unchanged (positions may move: the packed
    coordinates are rewritten in place) -- rigid.py keys its captured Arnoldi iterations on it."""
    return (self._geometry, tuple(self.target_range), tuple(sorted(self._options_set.items())))

  def buffers_signature(self):
    """Hash of the addresses of every device buffer the library owns for this context (read-only option
    "buffers_signature").  A captured graph holds those addresses by value: it may be replayed only while this is what it
    was at capture time (a buffer that grows is freed and reallocated)."""
    return self.get_option("buffers_signature")

  def get_option(self, key):
    v = ctypes.c_long()
    _lib.check(self._lib.rmb_ctx_get_option(self._h, key.encode(), ctypes.byref(v)))
    return int(v.value)

  def set_positions(self, r_vectors, a, periodic_length=None, wall=True):
    """r_vectors: numpy (N,3)/(3N,) or a CUDA torch float64 tensor (stays on device)."""
    L = _as_f64(np.zeros(3) if periodic_length is None else periodic_length, 3)
    if _is_torch_cuda(r_vectors):
      r = r_vectors.contiguous().view(-1)
      n = r.numel() // 3
      self._follow_torch_stream()
      _lib.check(self._lib.rmb_set_positions_device(self._h, ctypes.c_void_p(r.data_ptr()), n, float(a), _ptr(L),
                                                    int(bool(wall))))
      self._keepalive = r
    else:
      r = _as_f64(r_vectors)
      n = r.size // 3
      _lib.check(self._lib.rmb_set_positions(self._h, _ptr(r), n, float(a), _ptr(L), int(bool(wall))))
    self.n = n
    self.n_targets = n
    self.target_range = (0, n)
    self._geometry = (int(n), float(a), tuple(float(x) for x in L), bool(wall))

  def set_target_range(self, begin, end):
    _lib.check(self._lib.rmb_set_target_range(self._h, int(begin), int(end)))
    self.target_range = (int(begin), int(end))
    self.n_targets = int(end) - int(begin)

  # --- products --------------------------------------------------------------------------------
  def matvec(self, kind, vec, eta, vec2=None, in_plane=False):
    """Host path: numpy in, new numpy (3*n_targets,) out; synchronous."""
    k = _lib.KINDS[kind] if isinstance(kind, str) else int(kind)
    v = _as_f64(vec, 3 * self.n)
    v2 = _as_f64(vec2, 3 * self.n) if vec2 is not None else None
    out = np.empty(3 * self.n_targets)
    _lib.check(self._lib.rmb_matvec(self._h, k, int(bool(in_plane)), _ptr(v), _ptr(v2) if v2 is not None else None,
                                    float(eta), _ptr(out)))
    return out

  def matvec_device(self, kind, vec, eta, vec2=None, in_plane=False, out=None):
    """Device path: CUDA float64 torch tensors; asynchronous on the context's stream."""
    import torch
    k = _lib.KINDS[kind] if isinstance(kind, str) else int(kind)
    if not _is_torch_cuda(vec) or vec.numel() != 3 * self.n or not vec.is_contiguous():
      raise ValueError("vec must be a contiguous CUDA float64 tensor with 3*n entries")
    if vec2 is not None and (not _is_torch_cuda(vec2) or vec2.numel() != 3 * self.n or not vec2.is_contiguous()):
      raise ValueError("vec2 must be a contiguous CUDA float64 tensor with 3*n entries")
    if out is None:
      out = torch.empty(3 * self.n_targets, dtype=torch.float64, device=vec.device)
    elif not _is_torch_cuda(out) or out.numel() != 3 * self.n_targets or not out.is_contiguous():
      raise ValueError("out must be a contiguous CUDA float64 tensor with 3*n_targets entries")
    self._follow_torch_stream()
    _lib.check(self._lib.rmb_matvec_device(self._h, k, int(bool(in_plane)), ctypes.c_void_p(vec.data_ptr()),
                                           ctypes.c_void_p(vec2.data_ptr()) if vec2 is not None else None,
                                           float(eta), ctypes.c_void_p(out.data_ptr())))
    return out

  def matvec_pairshard_device(self, kind, vec, eta, shard, nshards, out=None):
    """Contribution of pair-shard `shard` of `nshards` to ALL targets (3n entries): tt / tr / rt / rr / tt_free."""
    import torch
    k = _lib.KINDS[kind] if isinstance(kind, str) else int(kind)
    if not _is_torch_cuda(vec) or vec.numel() != 3 * self.n or not vec.is_contiguous():
      raise ValueError("vec must be a contiguous CUDA float64 tensor with 3*n entries")
    if out is None:
      out = torch.empty(3 * self.n, dtype=torch.float64, device=vec.device)
    elif not _is_torch_cuda(out) or out.numel() != 3 * self.n or not out.is_contiguous():
      raise ValueError("out must be a contiguous CUDA float64 tensor with 3*n entries")
    self._follow_torch_stream()
    _lib.check(self._lib.rmb_matvec_pairshard_device(self._h, k, ctypes.c_void_p(vec.data_ptr()), float(eta),
                                                     ctypes.c_void_p(out.data_ptr()), int(shard), int(nshards)))
    return out

  def matvec2_device(self, kind, vec_a, vec_b, eta, out_a=None, out_b=None, shard=0, nshards=1):
    """Two tt products in one pass over the pairs: returns (M vec_a, M vec_b) (rmb_matvec2_device; with nshards > 1
    the contribution of one pair shard, to be summed over shards).  out_a / out_b: optional contiguous 3n tensors."""
    import torch
    k = _lib.KINDS[kind] if isinstance(kind, str) else int(kind)
    for v in (vec_a, vec_b):
      if not _is_torch_cuda(v) or v.numel() != 3 * self.n or not v.is_contiguous():
        raise ValueError("vectors must be contiguous CUDA float64 tensors with 3*n entries")
    outs = []
    for o in (out_a, out_b):
      if o is None:
        o = torch.empty(3 * self.n, dtype=torch.float64, device=vec_a.device)
      elif not _is_torch_cuda(o) or o.numel() != 3 * self.n or not o.is_contiguous():
        raise ValueError("outputs must be contiguous CUDA float64 tensors with 3*n entries")
      outs.append(o)
    self._follow_torch_stream()
    _lib.check(self._lib.rmb_matvec2_pairshard_device(self._h, k, ctypes.c_void_p(vec_a.data_ptr()),
                                                      ctypes.c_void_p(vec_b.data_ptr()), float(eta),
                                                      ctypes.c_void_p(outs[0].data_ptr()),
                                                      ctypes.c_void_p(outs[1].data_ptr()), int(shard), int(nshards)))
    return outs[0], outs[1]

  def matvec_op_device(self, op, vecs, eta, in_plane=False, outs=None, shard=0, nshards=1):
    """Several blocks of the grand mobility from one pass over the pairs (rmb_matvec_op_device):
      "velocity_from_force_torque"  (f, tau) -> (M_tt f + M_tr tau,)
      "grand"                       (f, tau) -> (M_tt f + M_tr tau, M_rt f + M_rr tau)
      "force_column"                (f,)     -> (M_tt f, M_rt f)
      "tt_multi" (tr_ / rt_ / rr_)  k vectors -> the block applied to each (k = 1..4)
    vecs / outs: sequences of contiguous CUDA float64 tensors with 3n (outs: 3*n_targets) entries; returns the tuple
    of outputs.  With nshards > 1: the contribution of one pair shard to all n targets (to be summed over shards)."""
    import torch
    code, n_in, n_out = _lib.OPS[op]
    if n_in is None:
      n_in = n_out = len(vecs)
    if len(vecs) != n_in:
      raise ValueError("%s takes %d input vectors" % (op, n_in))
    for v in vecs:
      if not _is_torch_cuda(v) or v.numel() != 3 * self.n or not v.is_contiguous():
        raise ValueError("vectors must be contiguous CUDA float64 tensors with 3*n entries")
    n_res = 3 * (self.n if nshards > 1 else self.n_targets)
    if outs is None:
      outs = [torch.empty(n_res, dtype=torch.float64, device=vecs[0].device) for _ in range(n_out)]
    outs = list(outs)
    if len(outs) != n_out:
      raise ValueError("%s produces %d output vectors" % (op, n_out))
    for o in outs:
      if not _is_torch_cuda(o) or o.numel() != n_res or not o.is_contiguous():
        raise ValueError("outputs must be contiguous CUDA float64 tensors with %d entries" % n_res)
    ins_p = (ctypes.c_void_p * n_in)(*[v.data_ptr() for v in vecs])
    outs_p = (ctypes.c_void_p * n_out)(*[o.data_ptr() for o in outs])
    self._follow_torch_stream()
    _lib.check(self._lib.rmb_matvec_op_pairshard_device(self._h, code, int(bool(in_plane)), n_in,
                                                        ctypes.cast(ins_p, ctypes.c_void_p), n_out,
                                                        ctypes.cast(outs_p, ctypes.c_void_p), float(eta), int(shard),
                                                        int(nshards)))
    return tuple(outs)

  def body_mobility_dense_device(self, first_blob, n_b, eta, out=None):
    """Dense (3 n_b x 3 n_b) tt mobility of every listed body (int64 CUDA tensor of first-blob indices)."""
    import torch
    if not (isinstance(first_blob, torch.Tensor) and first_blob.is_cuda and first_blob.dtype == torch.int64):
      raise ValueError("first_blob must be a CUDA int64 tensor")
    first_blob = first_blob.contiguous()
    nb = first_blob.numel()
    if out is None:
      out = torch.empty((nb, 3 * n_b, 3 * n_b), dtype=torch.float64, device=first_blob.device)
    self._follow_torch_stream()
    _lib.check(self._lib.rmb_body_mobility_dense_device(self._h, ctypes.c_void_p(first_blob.data_ptr()), nb, int(n_b),
                                                        float(eta), ctypes.c_void_p(out.data_ptr())))
    return out

  # --- O(N) helpers of the rigid-body solve (csrc/rmb_krylov.hip) -------------------------------
  @staticmethod
  def _block(t, transpose=False):
    """rmb_block of a (batch, rows, cols) CUDA float64 tensor (any strides), or of its transpose per batch entry."""
    if t is None:
      return None
    assert t.dim() == 3 and t.dtype.itemsize == 8
    bs, rs, cs = t.stride()
    return _lib.Block(t.data_ptr(), bs, cs if transpose else rs, rs if transpose else cs)

  def block_apply_device(self, a11, a12, a21, a22, x1, x2, y1, y2, alpha=1.0, beta1=0.0, beta2=0.0, transpose=(False,) * 4):
    """y1_b = beta1 y1_b + alpha (A11_b x1_b + A12_b x2_b), y2_b likewise with A21, A22, for every batch entry b, in one
    launch (rmb_block_apply_device).  a..: (batch, rows, cols) tensors or None (zero block); x1 (batch, c1), x2 (batch, c2),
    y1 (batch, r1), y2 (batch, r2) contiguous views.  transpose[k]: use the k-th block transposed."""
    nb, c1, c2, r1, r2 = x1.shape[0], x1.shape[1], x2.shape[1], y1.shape[1], y2.shape[1]
    blocks = [self._block(t, tr) for t, tr in zip((a11, a12, a21, a22), transpose)]
    refs = [ctypes.byref(b) if b is not None else None for b in blocks]
    assert x1.is_contiguous() and x2.is_contiguous() and y1.is_contiguous() and y2.is_contiguous()
    self._follow_torch_stream()
    _lib.check(self._lib.rmb_block_apply_device(self._h, nb, r1, c1, r2, c2, refs[0], refs[1], refs[2], refs[3],
                                                ctypes.c_void_p(x1.data_ptr()), ctypes.c_void_p(x2.data_ptr()), float(alpha),
                                                float(beta1), ctypes.c_void_p(y1.data_ptr()), float(beta2),
                                                ctypes.c_void_p(y2.data_ptr())))

  def rigid_configuration_device(self, ref, loc, quat, r, rel=None, K=None):
    """Blob coordinates, body-frame offsets and K of every body in one launch (rmb_rigid_configuration_device).
    ref (nb, n_b, 3), loc (nb, 3), quat (nb, 4); outputs r (nb n_b, 3), rel (nb, n_b, 3), K (nb, 3 n_b, 6), contiguous."""
    nb, n_b = ref.shape[0], ref.shape[1]
    for t in (ref, loc, quat, r, rel, K):
      assert t is None or (t.is_contiguous() and t.dtype.itemsize == 8)
    assert loc.numel() == 3 * nb and quat.numel() == 4 * nb and r.numel() == 3 * nb * n_b
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    self._follow_torch_stream()
    _lib.check(self._lib.rmb_rigid_configuration_device(self._h, nb, n_b, p(ref), p(loc), p(quat), p(r), p(rel), p(K)))

  def rigid_advance_device(self, loc, quat, U, dt):
    """(loc + U[:, :3] dt, quaternion(U[:, 3:] dt) * quat) as new tensors, one launch (rmb_rigid_advance_device).
    dt: a float, or a per-body (nb,) / (nb, 1) tensor."""
    import torch
    nb = loc.shape[0]
    U = U.reshape(nb, 6)
    assert loc.is_contiguous() and quat.is_contiguous() and U.is_contiguous()
    per_body = dt.reshape(-1).contiguous() if isinstance(dt, torch.Tensor) else None
    assert per_body is None or per_body.numel() == nb
    loc_out, quat_out = torch.empty_like(loc), torch.empty_like(quat)
    self._follow_torch_stream()
    _lib.check(self._lib.rmb_rigid_advance_device(self._h, nb, ctypes.c_void_p(loc.data_ptr()), ctypes.c_void_p(quat.data_ptr()),
                                                  ctypes.c_void_p(U.data_ptr()), 0.0 if per_body is not None else float(dt),
                                                  ctypes.c_void_p(per_body.data_ptr()) if per_body is not None else None,
                                                  ctypes.c_void_p(loc_out.data_ptr()), ctypes.c_void_p(quat_out.data_ptr())))
    return loc_out, quat_out

  def rigid_preconditioner_device(self, Mb, K, Lchol, Linv, Minv, Nbody, A11, A12, A21, A22, info):
    """Per-body Cholesky factor, inverses and the preconditioner's four blocks in one launch
    (rmb_rigid_preconditioner_device).  Mb (nb, n, n), K (nb, n, 6), outputs contiguous; info: int32 tensor of one entry."""
    nb, n = Mb.shape[0], Mb.shape[1]
    outs = (Lchol, Linv, Minv, Nbody, A11, A12, A21, A22)
    assert all(t.is_contiguous() for t in (Mb, K) + outs) and n % 3 == 0 and info.dtype.itemsize == 4
    self._follow_torch_stream()
    _lib.check(self._lib.rmb_rigid_preconditioner_device(self._h, nb, n // 3, ctypes.c_void_p(Mb.data_ptr()),
                                                         ctypes.c_void_p(K.data_ptr()), *[ctypes.c_void_p(t.data_ptr()) for t in outs],
                                                         ctypes.c_void_p(info.data_ptr())))

  def krylov_orthogonalize_device(self, V, rows, w, col, v_next, col_mapped=0):
    """Two classical Gram-Schmidt passes of w against V[:rows] (row-major (m, n) tensor), in place; col[:rows] = the
    coefficients, col[rows] = |w|, v_next = w / |w| (rmb_krylov_orthogonalize2_device).  col_mapped: device address of
    rows + 1 doubles of mapped host memory (MappedHostArray.dev_ptr + offset) that receive the column too, or 0."""
    n = w.numel()
    assert V.stride(1) == 1 and V.shape[1] == n and w.is_contiguous() and col.is_contiguous() and v_next.is_contiguous()
    assert col.numel() >= rows + 1 and v_next.numel() == n
    self._follow_torch_stream()
    _lib.check(self._lib.rmb_krylov_orthogonalize2_device(self._h, n, int(rows), ctypes.c_void_p(V.data_ptr()), V.stride(0),
                                                          ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(col.data_ptr()),
                                                          ctypes.c_void_p(v_next.data_ptr()),
                                                          ctypes.c_void_p(col_mapped) if col_mapped else None))

  def rigid_arnoldi_step_device(self, A11, A12, A21, A22, K, V, j, eta, z, w, col, col_mapped=0):
    """One Arnoldi step of the rigid-body GMRES enqueued by one call (rmb_rigid_arnoldi_step_device): z = P^-1 V[j],
    w = A z, Gram-Schmidt against V[:j + 1], column -> col (and the mapped host address), V[j + 1] = w / |w|."""
    nb, n_b = K.shape[0], K.shape[1] // 3
    for t in (A11, A12, A21, A22, K, z, w, col):
      assert t.is_contiguous()
    assert V.stride(1) == 1 and V.shape[1] == z.numel() == w.numel() == 3 * nb * n_b + 6 * nb and col.numel() >= j + 2
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    self._follow_torch_stream()
    _lib.check(self._lib.rmb_rigid_arnoldi_step_device(self._h, nb, n_b, p(A11), p(A12), p(A21), p(A22), p(K), p(V), V.stride(0), int(j),
                                                       float(eta), p(z), p(w), p(col), ctypes.c_void_p(col_mapped) if col_mapped else None))

  def rigid_gmres_device(self, A11, A12, A21, A22, K, b, tol, restart, maxiter, eta):
    """The whole right-preconditioned GMRES of the rigid-body problem in one library call (rmb_rigid_gmres_device); b is
    the RAW right-hand side (scaled to unit norm inside, the solution scaled back).  Returns (x tensor, info dict as
    rigid.gmres_right_preconditioned, plus rhs_norm)."""
    import torch
    nb, n_b = K.shape[0], K.shape[1] // 3
    for t in (A11, A12, A21, A22, K, b):
      assert t.is_contiguous()
    assert b.numel() == 3 * nb * n_b + 6 * nb
    x = torch.empty_like(b)
    its, disc, prod = ctypes.c_long(0), ctypes.c_long(0), ctypes.c_long(0)
    res, nrm = ctypes.c_double(0.0), ctypes.c_double(0.0)
    cap = int(maxiter) + 1
    hist = (ctypes.c_double * cap)()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    self._follow_torch_stream()
    _lib.check(self._lib.rmb_rigid_gmres_device(self._h, nb, n_b, p(A11), p(A12), p(A21), p(A22), p(K), p(b), float(tol), int(restart),
                                                int(maxiter), float(eta), p(x), ctypes.byref(its), ctypes.byref(res), ctypes.byref(disc),
                                                ctypes.byref(prod), hist, cap, ctypes.byref(nrm)))
    k = min(its.value, cap)
    return x, dict(iterations=int(its.value), residual=float(res.value), converged=bool(res.value <= tol), history=list(hist[:k]),
                   discarded_sweeps=int(disc.value), operator_applications=int(prod.value), rhs_norm=float(nrm.value))

  def rigid_lanczos_device(self, Linv, Lchol, z, factor, tol, max_iter, max_rows, eta):
    """The whole preconditioned Lanczos forcing in one library call (rmb_rigid_lanczos_device).  Returns (noise, iterations,
    products) or (None, status, products) when the library hands the forcing back (breakdown, more basis rows needed)."""
    import torch
    nb, n_b = Linv.shape[0], Linv.shape[1] // 3
    assert Linv.is_contiguous() and Lchol.is_contiguous() and z.is_contiguous() and z.numel() == 3 * nb * n_b
    noise = torch.empty_like(z)
    its, prod, status = ctypes.c_long(0), ctypes.c_long(0), ctypes.c_int(0)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    self._follow_torch_stream()
    _lib.check(self._lib.rmb_rigid_lanczos_device(self._h, nb, n_b, p(Linv), p(Lchol), p(z), float(factor), float(tol), int(max_iter),
                                                  int(max_rows), float(eta), p(noise), ctypes.byref(its), ctypes.byref(prod),
                                                  ctypes.byref(status)))
    if status.value != 0:
      return None, int(status.value), int(prod.value)
    return noise, int(its.value), int(prod.value)

  def lanczos_device(self, product, z, factor, tol, max_iter, max_rows, eta, in_plane=False):
    """factor * M^{1/2} z in one library call (rmb_lanczos_device): product "tt" (3n entries; in_plane: its in-plane variant) or
    "grand" (6n entries: the grand mobility on [z_f; z_tau]).  Returns (noise, iterations, products), or (None, status,
    products) when the library hands the forcing back (breakdown, more basis rows needed)."""
    import torch
    code = {"tt": 0, "grand": 1}[product]
    assert _is_torch_cuda(z) and z.is_contiguous() and z.numel() == (6 if code else 3) * self.n
    noise = torch.empty_like(z)
    its, prod, status = ctypes.c_long(0), ctypes.c_long(0), ctypes.c_int(0)
    self._follow_torch_stream()
    _lib.check(self._lib.rmb_lanczos_device(self._h, code, int(bool(in_plane)), ctypes.c_void_p(z.data_ptr()), float(factor), float(tol),
                                            int(max_iter), int(max_rows), float(eta), ctypes.c_void_p(noise.data_ptr()), ctypes.byref(its),
                                            ctypes.byref(prod), ctypes.byref(status)))
    if status.value != 0:
      return None, int(status.value), int(prod.value)
    return noise, int(its.value), int(prod.value)

  def rigid_operator_device(self, K, x, eta, out):
    """out = [M_tt lambda - K U; -K^T lambda] for x = [lambda; U] on the resident configuration (all bodies free, one body
    shape; rmb_rigid_operator_device): the pair sweep + one finishing launch.  K (n_bodies, 3 n_b, 6) contiguous."""
    nb, n_b = K.shape[0], K.shape[1] // 3
    assert K.is_contiguous() and x.is_contiguous() and out.is_contiguous() and x.numel() == 3 * nb * n_b + 6 * nb == out.numel()
    self._follow_torch_stream()
    _lib.check(self._lib.rmb_rigid_operator_device(self._h, nb, n_b, ctypes.c_void_p(K.data_ptr()), ctypes.c_void_p(x.data_ptr()),
                                                   float(eta), ctypes.c_void_p(out.data_ptr())))
    return out

  def blob_blob_force(self, repulsion_strength, debye_length, blob_radius):
    out = np.empty(3 * self.n_targets)
    _lib.check(self._lib.rmb_blob_blob_force(self._h, float(repulsion_strength), float(debye_length),
                                             float(blob_radius), _ptr(out)))
    return out.reshape(self.n_targets, 3)

  def blob_blob_force_radii(self, radius_blobs, repulsion_strength, debye_length):
    """One radius per blob (contact distance a_i + a_j, forces_numba.py:73-137); host arrays."""
    rad = _as_f64(radius_blobs, self.n)
    out = np.empty(3 * self.n_targets)
    _lib.check(self._lib.rmb_blob_blob_force_radii(self._h, _ptr(rad), float(repulsion_strength), float(debye_length),
                                                   _ptr(out)))
    return out.reshape(self.n_targets, 3)

  def blob_blob_force_radii_device(self, radius_blobs, repulsion_strength, debye_length, out=None):
    import torch
    if not _is_torch_cuda(radius_blobs) or radius_blobs.numel() != self.n or not radius_blobs.is_contiguous():
      raise ValueError("radius_blobs must be a contiguous CUDA float64 tensor with n entries")
    if out is None:
      out = torch.empty(3 * self.n_targets, dtype=torch.float64, device=radius_blobs.device)
    self._follow_torch_stream()
    _lib.check(self._lib.rmb_blob_blob_force_radii_device(self._h, ctypes.c_void_p(radius_blobs.data_ptr()),
                                                          float(repulsion_strength), float(debye_length),
                                                          ctypes.c_void_p(out.data_ptr())))
    return out

  def blob_blob_force_device(self, repulsion_strength, debye_length, blob_radius, out=None, device=None):
    import torch
    if out is None:
      out = torch.empty(3 * self.n_targets, dtype=torch.float64, device=device or ("cuda:%d" % self.device))
    self._follow_torch_stream()
    _lib.check(self._lib.rmb_blob_blob_force_device(self._h, float(repulsion_strength), float(debye_length),
                                                    float(blob_radius), ctypes.c_void_p(out.data_ptr())))
    return out

  def one_blob_force_device(self, r, blob_radius, weight, eps_wall, debye_wall, out=None):
    """(0, 0, -weight + wall repulsion) per blob on the caller's coordinates r (n x 3 CUDA tensor; rmb_one_blob_force_device);
    `out` given = accumulate into it, None = a new tensor."""
    import torch
    assert _is_torch_cuda(r) and r.is_contiguous() and r.dtype == torch.float64
    n = r.numel() // 3
    acc = out is not None
    if out is None:
      out = torch.empty(3 * n, dtype=torch.float64, device=r.device)
    assert out.is_contiguous() and out.numel() == 3 * n
    self._follow_torch_stream()
    _lib.check(self._lib.rmb_one_blob_force_device(self._h, n, ctypes.c_void_p(r.data_ptr()), float(blob_radius), float(weight), float(eps_wall),
                                                   float(debye_wall) if eps_wall != 0.0 else 1.0, 1 if acc else 0, ctypes.c_void_p(out.data_ptr())))
    return out

  def blob_blob_force_pairshard_device(self, repulsion_strength, debye_length, blob_radius, shard, nshards, out=None, device=None):
    """Contribution of pair shard `shard` of `nshards` to the forces on ALL blobs (3n entries; sum over shards = forces)."""
    import torch
    if out is None:
      out = torch.empty(3 * self.n, dtype=torch.float64, device=device or ("cuda:%d" % self.device))
    elif not _is_torch_cuda(out) or out.numel() != 3 * self.n or not out.is_contiguous():
      raise ValueError("out must be a contiguous CUDA float64 tensor with 3*n entries")
    self._follow_torch_stream()
    _lib.check(self._lib.rmb_blob_blob_force_pairshard_device(self._h, float(repulsion_strength), float(debye_length),
                                                              float(blob_radius), ctypes.c_void_p(out.data_ptr()), int(shard),
                                                              int(nshards)))
    return out

  # --- measurement -----------------------------------------------------------------------------
  def timing_collect(self, max_n=8192):
    buf = (ctypes.c_double * max_n)()
    n = self._lib.rmb_timing_collect(self._h, buf, max_n)
    if n < 0:
      _lib.check(n)
    return np.array(buf[:n])

  def wave_clock_collect(self, max_waves=8192):
    """(n_waves, 2) array of start/end wall-clock stamps (100 MHz ticks) of the last symmetric launch."""
    buf = np.zeros((max_waves, 2), dtype=np.int64)
    n = self._lib.rmb_wave_clock_collect(self._h, ctypes.c_void_p(buf.ctypes.data), max_waves)
    if n < 0:
      _lib.check(n)
    return buf[:n]

  def ubench_fp64_issue(self, launches=40):
    """G wave-instructions/s of independent v_fma_f64 on this chip right now (rmb_ubench_fp64_issue)."""
    out = ctypes.c_double()
    self._follow_torch_stream()
    _lib.check(self._lib.rmb_ubench_fp64_issue(self._h, int(launches), ctypes.byref(out)))
    return float(out.value)

  def last_host_timing(self):
    """Host wall clock (us) of the last matvec() on this context: upload, launch, wait + download, whole C call."""
    buf = (ctypes.c_double * 4)()
    _lib.check(self._lib.rmb_last_host_timing(self._h, buf))
    return dict(upload_us=buf[0], launch_us=buf[1], wait_and_download_us=buf[2], c_call_us=buf[3])

  def timing_reset(self):
    _lib.check(self._lib.rmb_timing_reset(self._h))

  def last_launch(self):
    t, c, w = ctypes.c_long(), ctypes.c_long(), ctypes.c_long()
    _lib.check(self._lib.rmb_last_launch(self._h, ctypes.byref(t), ctypes.byref(c), ctypes.byref(w)))
    return dict(tiles=t.value, chunks=c.value, workgroups=w.value)

  def synchronize(self):
    _lib.check(self._lib.rmb_ctx_synchronize(self._h))


_raw_stream = None


def _current_raw_stream(device_index):
  """hipStream_t of torch's current stream on `device_index`, as an int.  Every *_device call asks for it, so the cheap
  accessor is used when this torch has it (0.3 us; torch.cuda.current_stream() builds a Stream object: 3.5 us, a third
  of a small product's launch cost)."""
  global _raw_stream
  if _raw_stream is None:
    import torch
    fast = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if fast is not None:
      _raw_stream = fast
    else:
      _raw_stream = lambda idx: torch.cuda.current_stream(torch.device("cuda", idx)).cuda_stream   # noqa: E731
  return _raw_stream(device_index)


def _is_torch_cuda(x):
  try:
    import torch
  except ImportError:
    return False
  return isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float64


class MappedHostArray(object):
  """A float64 numpy array in page-locked host memory that is mapped into the device's address space
  (rmb_host_mapped_alloc): kernels store into `dev_ptr`, the host reads `array` after one event / stream wait -- no copy
  command.  Freed with close() or when collected."""

  def __init__(self, shape):
    self._lib = _lib.load()
    n = int(np.prod(shape))
    h, d = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.check(self._lib.rmb_host_mapped_alloc(ctypes.c_size_t(8 * n), ctypes.byref(h), ctypes.byref(d)))
    self._host = h
    self.dev_ptr = int(d.value)
    self.array = np.ctypeslib.as_array((ctypes.c_double * n).from_address(h.value)).reshape(shape)

  def close(self):
    if self._host is not None:
      self.array = None
      self._lib.rmb_host_mapped_free(self._host)
      self._host = None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass
