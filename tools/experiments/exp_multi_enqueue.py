"""Host-side cost of ENQUEUEING one product on the multi-device engine (time for the call to return, the GPU far
behind): serial issue from the calling thread vs one worker thread per shard.  Independent of how many physical devices
there are -- on a node this is the delay before the last device starts."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from rigidmultiblobswall_amd.multi import MultiContext
from bench import d2_cloud

N = 60000
r, f, eta, a = d2_cloud(N)
rd = torch.as_tensor(r.reshape(-1), device="cuda"); fd = torch.as_tensor(f.reshape(-1), device="cuda")
out = torch.empty(3 * N, dtype=torch.float64, device="cuda")


def enqueue_us(fn, calls=12, reps=5):
  best = 1e9
  for _ in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls): fn()
    best = min(best, (time.perf_counter() - t0) / calls * 1e6)
    torch.cuda.synchronize()
  return best


ctx = MobilityContext(0); ctx.set_positions(rd, a, None, True)
print("plain context: %.1f us per enqueued product" % enqueue_us(lambda: ctx.matvec_device("tt", fd, eta, out=out)), flush=True)
for G in (1, 2, 4, 8):
  for threads in (1, 0):
    os.environ["RMB_MULTI_THREADS"] = str(threads)
    m = MultiContext([0] * G); m.set_positions(rd, a, None, True)
    print("engine G=%d %s: %.1f us per enqueued product" % (G, "workers" if m.get_option("threads") else "serial ",
                                                            enqueue_us(lambda: m.matvec_device("tt", fd, eta, out=out))), flush=True)
    m.close()
ctx.close()
