"""One Arnoldi iteration of the rigid-body GMRES (preconditioner + operator + two Gram-Schmidt passes + normalise +
column to pinned memory) as eager torch launches against a captured hipGraph replayed -- small decks, where the
iteration is launch-bound (tools/experiments/exp_small_deck_gmres.py)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import structures as st
from rigidmultiblobswall_amd.rigid import RigidSuspension
dev = torch.device("cuda:0")
R, eta3 = 1.0155, 0.957e-3
shell = st.icosahedron_shell(0.792079207921 * R)
a3 = st.min_blob_separation(shell) / 2
for nb in [int(x) for x in sys.argv[1:]] or [8, 64, 256, 1024]:
  loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=5)
  rs = RigidSuspension([shell] * nb, loc, quat, a3, eta3, device=dev)
  rs.build_preconditioner()
  n, m, j = rs.size, 30, 8
  V = torch.randn((m + 1, n), dtype=torch.float64, device=dev)
  V /= torch.linalg.vector_norm(V, dim=1, keepdim=True)
  cols = torch.zeros((m, m + 2), dtype=torch.float64, device=dev)
  host = torch.empty((m, m + 2), dtype=torch.float64).pin_memory()

  native = os.environ.get("RMB_NATIVE_HELPERS", "") != "0"

  def body():
    w = rs.apply_operator(rs.apply_preconditioner(V[j]))
    if native:
      rs.ctx.krylov_orthogonalize_device(V, j + 1, w, cols[j], V[j + 1])
      host[j, :j + 2].copy_(cols[j, :j + 2], non_blocking=True)
      return
    Vj = V[:j + 1]
    h = Vj @ w
    w = torch.addmv(w, Vj.t(), h, alpha=-1.0)
    h2 = Vj @ w
    w = torch.addmv(w, Vj.t(), h2, alpha=-1.0)
    torch.add(h, h2, out=cols[j, :j + 1])
    torch.linalg.vector_norm(w, out=cols[j, j + 1])
    torch.div(w, cols[j, j + 1], out=V[j + 1])
    host[j, :j + 2].copy_(cols[j, :j + 2], non_blocking=True)

  def timed(fn, reps=300):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / reps

  t_eager = timed(body)
  ref = V[j + 1].clone(); refc = cols[j].clone()
  s = torch.cuda.Stream()
  with torch.cuda.stream(s):
    for _ in range(3): body()                      # the context follows torch's stream; libraries warm
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
      body()
    V[j + 1].zero_(); cols[j].zero_()
    t_graph = timed(g.replay)
    torch.cuda.synchronize()
    err = float(torch.linalg.norm(V[j + 1] - ref) / torch.linalg.norm(ref)), float(torch.linalg.norm(cols[j] - refc) / torch.linalg.norm(refc))
  mv = timed(lambda: rs.ctx.matvec_device("tt", V[0][:3 * rs.n_blobs], eta3))
  print("bodies %5d blobs %6d: iteration eager %7.1f us | graph replay %7.1f us (x%.2f) | blob product alone %6.1f us | replay vs eager rel diff %.1e %.1e"
        % (nb, rs.n_blobs, t_eager, t_graph, t_eager / t_graph, mv, err[0], err[1]), flush=True)
  rs.close()
