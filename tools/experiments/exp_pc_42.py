"""Per-body preconditioner factors for the reference's 42-blob shells: the in-place LDS kernel (rmb_rigid_preconditioner_device,
17 .. 42 blobs per body, round 5) against the batched torch.linalg build, and a whole deterministic time step either way."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd.rigid import RigidSuspension
from rigidmultiblobswall_amd.rigid_integrator import RigidIntegrator

g = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "g9_rigid_det_euler_42blob_shells.npz"))
shell = g["vertex_shell42"]
d = np.linalg.norm(shell[:, None] - shell[None], axis=2)
a, eta = float(np.min(d[d > 0]) / 2), 0.957e-3
for nb in (16, 64, 256, 1024):
  m = int(np.ceil(np.sqrt(nb)))
  loc = np.array([[2.6 * (k % m), 2.6 * (k // m), 1.4] for k in range(nb)])
  quat = np.tile([1.0, 0, 0, 0], (nb, 1))
  row = {}
  for native in (True, False):
    s = RigidSuspension([shell] * nb, loc, quat, a, eta, device="cuda:0")
    s.native_helpers = native
    for _ in range(3):
      s.build_preconditioner()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
      s.build_preconditioner()
    torch.cuda.synchronize()
    row[native] = (time.perf_counter() - t0) / 10 * 1e3
    s.close()
  steps = {}
  for native in (True, False):
    integ = RigidIntegrator([shell] * nb, loc, quat, "deterministic_adams_bashforth", a, eta, tolerance=1e-8, device="cuda:0", seed=1)
    integ.susp.native_helpers = native
    integ.g = 0.01; integ.repulsion_strength_wall, integ.debye_length_wall = 0.0165677856, 0.0656
    for st_ in range(4):
      integ.advance_time_step(0.002, step=st_)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for st_ in range(4, 14):
      integ.advance_time_step(0.002, step=st_)
    torch.cuda.synchronize()
    steps[native] = ((time.perf_counter() - t0) / 10 * 1e3, integ.det_iterations_count / 14.0)
    integ.close()
  print("%5d shells x 42 blobs (%6d blobs): build_preconditioner %8.3f ms (LDS kernel) vs %8.3f ms (torch.linalg)  x %.2f | "
        "deterministic_adams_bashforth step %8.3f vs %8.3f ms (%.1f GMRES iterations per step)" %
        (nb, 42 * nb, row[True], row[False], row[False] / row[True], steps[True][0], steps[False][0], steps[True][1]), flush=True)
