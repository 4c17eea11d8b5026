"""Does a hipGraph shrink the per-step overhead outside the sweep kernel (finalize launch + gaps, ~8 us of 199 us at
1e4 blobs)?  A/B: 200 steps issued as plain stream launches vs one captured graph of [sweep, finalize] replayed 200
times vs a captured graph of 20 steps replayed 10 times.  torch.cuda.CUDAGraph capture works because the context
enqueues on torch's current stream."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
r, f, eta, a = d2_cloud(N)
rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
out = torch.empty_like(fd)
ctx = MobilityContext(0)
ctx.set_positions(rd, a, wall=True)
for _ in range(1500):
  ctx.matvec_device("tt", fd, eta, out=out)
torch.cuda.synchronize()
ref = out.clone()


def timed(fn, reps):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(reps):
    fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / reps

s = torch.cuda.Stream()
with torch.cuda.stream(s):
  ctx.matvec_device("tt", fd, eta, out=out)       # switch the context to this stream before capturing
  torch.cuda.synchronize()
  g1 = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g1, stream=s):
    ctx.matvec_device("tt", fd, eta, out=out)
  g20 = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g20, stream=s):
    for _ in range(20):
      ctx.matvec_device("tt", fd, eta, out=out)
  for rep in range(3):
    t_plain = timed(lambda: ctx.matvec_device("tt", fd, eta, out=out), 200)
    t_g1 = timed(g1.replay, 200)
    t_g20 = timed(g20.replay, 10) / 20
    print("per step: plain launches %.2f us | graph of 1 step %.2f us | graph of 20 steps %.2f us" %
          (1e6 * t_plain, 1e6 * t_g1, 1e6 * t_g20), flush=True)
  out.zero_(); g20.replay(); torch.cuda.synchronize()
  print("graph result rel err vs plain:", float(torch.linalg.norm(out - ref) / torch.linalg.norm(ref)))
ctx.close()
