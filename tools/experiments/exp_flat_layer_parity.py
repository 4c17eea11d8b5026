import sys, numpy as np
sys.path.insert(0, '.')
from oracle import oracle
from rigidmultiblobswall_amd import mobility as mob
oracle.build()
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
for N, span, zmax in ((3000, 3000.0, 1.2), (3000, 30000.0, 1.02), (3000, 300.0, 5.0), (2000, 1e6, 1.0)):
  rng = np.random.RandomState(N + int(span))
  a, eta = 0.5, 1.0
  r = np.column_stack([rng.rand(N) * span * a, rng.rand(N) * span * a, a * (1.0 + (zmax - 1.0) * rng.rand(N))])
  f = rng.randn(N, 3)
  for nm in ("trans_times_force", "trans_times_torque", "rot_times_force", "rot_times_torque"):
    u = getattr(mob, "single_wall_mobility_" + nm + "_hip")(r, f, eta, a)
    ref = getattr(oracle, "single_wall_mobility_" + nm + "_oracle")(r, f, eta, a)
    # also error relative to the size of the cancelling parts: the unbounded product
    ub = getattr(oracle, "no_wall_mobility_" + nm + "_oracle")(r, f, eta, a)
    print(N, span, zmax, nm, "rel err %.2e   |u_wall|/|u_nowall| = %.2e" % (rel(u, ref), np.linalg.norm(ref) / np.linalg.norm(ub)))
