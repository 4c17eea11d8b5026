"""cProfile of Brownian Adams-Bashforth steps of 1000 rollers: where the host time of a 2 ms step goes."""
import cProfile, pstats, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import structures as st
from rigidmultiblobswall_amd.rollers import RollersIntegrator
n5, a5 = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 0.656
loc5, _, _ = st.roller_monolayer(n5, radius=a5, seed=7)
integ = RollersIntegrator(loc5, "stochastic_adams_bashforth_rollers", a5, 1.0e-3, tolerance=1e-6, device="cuda:0", seed=11)
integ.kT, integ.g = 0.0041419464, 0.0024892
integ.repulsion_strength = integ.repulsion_strength_wall = 0.0165677856
integ.debye_length = integ.debye_length_wall = 0.0656
integ.omega_one_roller = np.array([0.0, 62.8, 0.0])
integ.report_rejections = False
for _ in range(5): integ.advance_time_step(0.016)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40): integ.advance_time_step(0.016)
torch.cuda.synchronize()
print("%d rollers: %.3f ms per step" % (n5, (time.perf_counter() - t0) / 40 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(40): integ.advance_time_step(0.016)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(26)
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
