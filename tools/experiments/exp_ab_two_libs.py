"""Same-box A/B of TWO BUILDS of the library on the headline products: alternating child processes (one library each, RMB_AB_LIB),
every one priming the clocks first; HIP-event kernel time.   python tools/experiments/exp_ab_two_libs.py <other .so> [sizes]"""
import json, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
  import time
  import numpy as np, torch
  sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
  from rigidmultiblobswall_amd import _lib as L
  if os.environ.get("RMB_AB_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["RMB_AB_LIB"])
  from rigidmultiblobswall_amd import MobilityContext
  from bench import d2_cloud
  out = {}
  ctx = MobilityContext(0)
  for N in [int(x) for x in sys.argv[2].split(",")]:
    r, f, eta, a = d2_cloud(N)
    rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
    o = torch.empty(3 * N, dtype=torch.float64, device="cuda")
    for wall in (True, False):
      ctx.set_positions(rd, a, None, wall)
      for kind in ("tt", "rr", "tr"):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.25:
          for _ in range(10): ctx.matvec_device(kind, fd, eta, out=o)
          torch.cuda.synchronize()
        reps = 200 if N <= 30000 else 12
        ctx.set_option("timing", 1); ctx.timing_reset()
        for _ in range(reps): ctx.matvec_device(kind, fd, eta, out=o)
        torch.cuda.synchronize()
        out["%d %s %s" % (N, "wall" if wall else "open", kind)] = float(np.median(ctx.timing_collect(reps))) * 1e3
        ctx.set_option("timing", 0)
  print(json.dumps(out))
  sys.exit(0)
other = sys.argv[1]
sizes = sys.argv[2] if len(sys.argv) > 2 else "10000,100000"
res = {"new": [], "old": []}
for rnd in range(3):
  for tag in (("new", "old") if rnd % 2 == 0 else ("old", "new")):
    env = dict(os.environ)
    if tag == "old":
      env["RMB_AB_LIB"] = other
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", sizes], env=env, capture_output=True, text=True, timeout=600)
    rows = [l for l in p.stdout.split("\n") if l.startswith("{")]
    if not rows:
      print(p.stderr[-2000:]); sys.exit(1)
    res[tag].append(json.loads(rows[-1]))
import numpy as np
for k in res["new"][0]:
  new = np.median([r[k] for r in res["new"]]); old = np.median([r[k] for r in res["old"]])
  print("%-22s other build %10.2f us   this build %10.2f us   x %.4f" % (k, old, new, old / new), flush=True)
