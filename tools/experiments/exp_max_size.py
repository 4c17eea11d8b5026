"""Beyond BASELINE's largest size: one wall tt product at N blobs (default 3 000 000: 46 875 tiles, 3.5e10 rotation steps --
past 2^32 -- 4.5e12 pair evaluations) against the oracle on a sample of targets incl. tile edges and the last tile, and the
symmetry check  g.(M f) = f.(M g)  over the whole vector.   python tools/experiments/exp_max_size.py [N]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud
from oracle import oracle            # the checker (tools only; the product never imports it)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000000
oracle.build()
r, f, eta, a = d2_cloud(N)
r = r.copy(); r[:, 2] -= 0.45 * a
rng = np.random.RandomState(7)
g = rng.randn(N, 3)
tg = rng.choice(N, 40, replace=False)
tg[:10] = [0, 63, 64, N - 1, N - 64, N - 65, (N // 2) // 64 * 64, (N // 2) // 64 * 64 + 63, 2 ** 21, 2 ** 21 - 1]
tg = np.unique(tg[tg < N])
ctx = MobilityContext(0)
rd = torch.as_tensor(r.reshape(-1), device="cuda")
ctx.set_positions(rd, a, None, True)
out = {}
for name, v in (("f", f), ("g", g)):
  vd = torch.as_tensor(v.reshape(-1), device="cuda")
  torch.cuda.synchronize(); t0 = time.perf_counter()
  u = ctx.matvec_device("tt", vd, eta)
  torch.cuda.synchronize()
  print("N = %d: M.%s in %.2f s (last_path %d)" % (N, name, time.perf_counter() - t0, ctx.get_option("last_path")), flush=True)
  out[name] = u.cpu().numpy()
r_eff, bdiag, _ = oracle.wall_regularisation(r, a)
t0 = time.perf_counter()
ref = oracle.raw_matvec_targets("tt", 1, r_eff, f * bdiag[:, None], eta, a, tg).reshape(-1, 3) * bdiag[tg][:, None]
got = out["f"].reshape(-1, 3)[tg]
err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
print("oracle on %d targets (%.1f s): relative error %.2e" % (len(tg), time.perf_counter() - t0, err), flush=True)
s1, s2 = float(np.dot(g.reshape(-1), out["f"])), float(np.dot(f.reshape(-1), out["g"]))
print("symmetry g.(M f) = %.15e, f.(M g) = %.15e, relative difference %.2e" % (s1, s2, abs(s1 - s2) / abs(s1)))
assert err < 1e-12 and abs(s1 - s2) / abs(s1) < 1e-11 and np.isfinite(out["f"]).all()
print("MAX SIZE OK")
