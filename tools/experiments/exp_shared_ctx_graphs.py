import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_gpu_rigid import _shell_suspension
from rigidmultiblobswall_amd import MobilityContext
rng = np.random.RandomState(4)
ctx = MobilityContext(0)
small_e, _, _ = _shell_suspension(20, seed=3)
big_e, _, _ = _shell_suspension(60, seed=4)
small_g, loc_s, quat_s = _shell_suspension(20, seed=3, ctx=ctx)
big_g, loc_b, quat_b = _shell_suspension(60, seed=4, ctx=ctx)
small_e.gmres_graph = big_e.gmres_graph = False
small_g.gmres_graph = big_g.gmres_graph = True
def both(e, g, label):
  rhs = torch.as_tensor(rng.randn(e.size), device="cuda:0")
  xe, ie = e.solve(rhs, tol=1e-8, restart=60)
  xg, ig = g.solve(rhs, tol=1e-8, restart=60)
  torch.cuda.synchronize()
  ws = g._arnoldi_ws
  print(label, ie["iterations"], ig["iterations"], "replays", ig["graph_replays"], "drops", ws.stale_drops, "captures", ws.captures,
        "hist diff", max(abs(a - b) / b for a, b in zip(ie["history"], ig["history"])), "sig", ctx.buffers_signature() % 100000, flush=True)
small_g.set_configuration(loc_s, quat_s)
for k in range(4): both(small_e, small_g, "small%d" % k)
big_g.set_configuration(loc_b, quat_b)
for k in range(4): both(big_e, big_g, "big%d" % k)
small_g.set_configuration(loc_s, quat_s)
for k in range(4): both(small_e, small_g, "small again %d" % k)
big_g.set_configuration(loc_b, quat_b)
for k in range(4): both(big_e, big_g, "big again %d" % k)
