"""Small roller decks (single-blob rollers, quaternion_integrator_rollers.py schemes): Brownian Adams-Bashforth steps with the
library's fused Gram-Schmidt in the Lanczos / GMRES loops (RollersIntegrator.fused_gram_schmidt) and with the whole Lanczos loop
inside the library (rmb_lanczos_device, the default) against the tensor operations; same seed, same-process A/B."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rigidmultiblobswall_amd import structures as st
from rigidmultiblobswall_amd.rollers import RollersIntegrator
a5 = 0.656
cases = ((1000, "stochastic_adams_bashforth_rollers"), (4096, "stochastic_adams_bashforth_rollers"), (16384, "stochastic_adams_bashforth_rollers"),
                   (1000, "stochastic_first_order_rollers"), (1000, "deterministic_adams_bashforth_rollers"))
if len(sys.argv) > 1:
  cases = tuple((int(x), "stochastic_adams_bashforth_rollers") for x in sys.argv[1].split(","))
for n5, scheme in cases:
  loc5, _, _ = st.roller_monolayer(n5, radius=a5, seed=7)
  res = {}
  for fused in (False, None, "loop", False, None, "loop"):
    integ = RollersIntegrator(loc5, scheme, a5, 1.0e-3, tolerance=1e-6, device="cuda:0", seed=11)
    integ.fused_gram_schmidt = None if fused == "loop" else fused
    integ.native_lanczos = None if fused == "loop" else False
    integ.kT, integ.g = 0.0041419464, 0.0024892
    integ.repulsion_strength = integ.repulsion_strength_wall = 0.0165677856
    integ.debye_length = integ.debye_length_wall = 0.0656
    integ.omega_one_roller = np.array([0.0, 62.8, 0.0])
    integ.report_rejections = False
    for _ in range(3): integ.advance_time_step(0.016)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    l0, d0 = integ.stoch_iterations_count, integ.det_iterations_count
    for _ in range(20): integ.advance_time_step(0.016)
    torch.cuda.synchronize()
    res[fused] = ((time.perf_counter() - t0) / 20 * 1e3, integ.location.clone(), (integ.stoch_iterations_count - l0) / 20.0, (integ.det_iterations_count - d0) / 20.0)
    integ.close()
  d = max(float((res[False][1] - res[m][1]).abs().max()) for m in (None, "loop"))
  print("%6d rollers, %-38s: tensor operations %.3f ms per step, fused Gram-Schmidt %.3f ms, the library's loop %.3f ms (Lanczos %.1f / %.1f / %.1f, GMRES %.1f iterations per step), positions differ by %.1e"
        % (n5, scheme, res[False][0], res[None][0], res["loop"][0], res[False][2], res[None][2], res["loop"][2], res[False][3], d), flush=True)
