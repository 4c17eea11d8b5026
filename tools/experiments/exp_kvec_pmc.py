"""Workload for PMC passes on the k-vector symmetric passes: wall tt on k = 1, 2, 3, 4 vectors at N blobs (a few launches
each).  Run under rocprofv3 --pmc ...; kernel names tell the k."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import MobilityContext
from bench import d2_cloud
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
r, f, eta, a = d2_cloud(N)
rd = torch.as_tensor(r.reshape(-1), device="cuda")
vs = [torch.as_tensor(np.random.RandomState(k).randn(3 * N), device="cuda") for k in range(4)]
ctx = MobilityContext(0); ctx.set_positions(rd, a, None, True)
ctx.set_option("symx_single", 1)          # k = 1 on the same generic skeleton
for k in (1, 2, 3, 4):
  for _ in range(4):
    ctx.matvec_op_device("tt_multi", vs[:k], eta)
torch.cuda.synchronize()
ctx.close()
