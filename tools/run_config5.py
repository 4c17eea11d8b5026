"""BASELINE.json configs[4] as written: a 2.6e5-blob Brownian suspension advanced for 100 time steps
(multi_bodies/multi_bodies.py:1511 main loop; per step: blob-blob forces kernel, M_tt F + M_tr T, Lanczos
M^{1/2} z, random-finite-difference drift; quaternion_integrator/quaternion_integrator_rollers.py:251-302).

  python tools/run_config5.py [driven|equilibrium|multiblob] [N] [steps] [dt|-] [single|double]

  driven       262 144 torque-driven rollers in a dense monolayer (the recipe bench.py times for 2 steps), physical
               parameters of multi_bodies/examples/rollers/inputfile_rollers.dat
  equilibrium  the same stepper on non-driven rollers started in equilibrium: after 100 steps the mean height must
               still be the Gibbs-Boltzmann one (check of tests/test_gpu_physics.py at full size)
  multiblob    21 845 twelve-blob shells (262 140 blobs), stochastic_Slip_Trapz: 3 GMRES solves + preconditioned
               Lanczos + forces kernel per step
Prints a progress line every step (and every rejected step, from the integrator) and one JSON record at the end (-> profiles/).
"""
import json
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "experiments"))   # exp_equilibrium
from rigidmultiblobswall_amd import structures as st
from rigidmultiblobswall_amd.rollers import RollersIntegrator

mode = sys.argv[1] if len(sys.argv) > 1 else "driven"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dt_arg = float(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4] != "-" else None
precision = sys.argv[5] if len(sys.argv) > 5 else "double"
dev = "cuda:0"
rec = {"mode": mode, "steps": steps}

if mode in ("driven", "equilibrium"):
  import exp_equilibrium as E
  a, eta, dt = E.a, E.eta, (0.016 if dt_arg is None else dt_arg)
  m1, var, h, cdf = E.analytic_moments()
  rng = np.random.RandomState(7)
  if mode == "driven":
    loc, _, _ = st.roller_monolayer(N, radius=a, seed=7)
  else:
    side = int(math.ceil(math.sqrt(N)))
    ij = np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 2)[:N].astype(float)
    loc = np.empty((N, 3))
    loc[:, :2] = ij * 5.0 * a
    loc[:, 2] = np.interp(rng.rand(N), cdf, h)
  integ = RollersIntegrator(loc, "stochastic_adams_bashforth_rollers", a, eta, tolerance=1e-3, device=dev, seed=11)
  integ.precision = precision
  rec["precision_of_the_mobility_products"] = precision
  integ.max_retries = 200                # in total; a single step rejected max_consecutive_retries (20) times in a row raises
  integ.kT, integ.g = E.kT, E.mg
  integ.repulsion_strength = integ.repulsion_strength_wall = E.ew
  integ.debye_length = integ.debye_length_wall = E.bw
  if mode == "driven":
    integ.omega_one_roller = np.array([0.0, 62.8, 0.0])
  heights = []
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for step in range(steps):
    integ.advance_time_step(dt)
    heights.append(float(integ.location[:, 2].mean()))
    torch.cuda.synchronize()      # a line every step: a step takes >= 1 s here, and a silent run is a killed run
    print("step %3d  %.1f s  mean height %.4f  products %d  lanczos its %d  rejected %d" %
          (step + 1, time.perf_counter() - t0, heights[-1], integ.mobility_products, integ.stoch_iterations_count,
           integ.invalid_configuration_count), flush=True)
  torch.cuda.synchronize()
  wall = time.perf_counter() - t0
  tail = heights[steps // 4:]
  rec.update(rollers=N, scheme=integ.scheme, dt=dt, lanczos_tolerance=1e-3, seconds=round(wall, 2),
             steps_per_s=round(steps / wall, 4), s_per_step=round(wall / steps, 4),
             pair_sweeps_per_step=integ.mobility_products / steps, lanczos_iterations_per_step=integ.stoch_iterations_count / steps,
             rejected_steps=integ.invalid_configuration_count, mean_height_last_three_quarters=float(np.mean(tail)),
             analytic_equilibrium_mean_height=m1)
  if mode == "equilibrium":
    rec["mean_height_over_analytic"] = float(np.mean(tail) / m1)
    rec["equilibrium_check_passed"] = bool(abs(np.mean(tail) / m1 - 1.0) < 0.01 and integ.invalid_configuration_count == 0)
else:
  from rigidmultiblobswall_amd.rigid_integrator import RigidIntegrator
  R, eta, nb = 1.0155, 0.957e-3, (N if N < 100000 else 21845)
  shell = st.icosahedron_shell(0.792079207921 * R)
  a = st.min_blob_separation(shell) / 2
  loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=5)
  ri = RigidIntegrator([shell] * nb, loc, quat, "stochastic_Slip_Trapz", a, eta, tolerance=1e-4, device=dev, seed=1)
  ri.kT, ri.g = 0.0040749841, 0.0303 / 12
  ri.repulsion_strength_wall = ri.repulsion_strength = 0.0326
  ri.debye_length_wall = ri.debye_length = 0.0406
  FT = torch.zeros((nb, 6), dtype=torch.float64, device=dev)
  FT[:, 4] = 8 * math.pi * eta * R ** 3 * 62.8
  ri.external_force_torque = lambda it: FT
  ri.precision = precision
  rec["precision_of_the_mobility_products"] = precision
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for step in range(steps):
    ri.advance_time_step(0.01, step=step)
    torch.cuda.synchronize()
    print("step %3d  %.1f s  mean height %.4f  gmres %d  lanczos %d  sweeps %d  rejected %d" %
          (step + 1, time.perf_counter() - t0, float(ri.location[:, 2].mean()), ri.det_iterations_count,
           ri.stoch_iterations_count, ri.susp.matvec_count, ri.invalid_configuration_count), flush=True)
  torch.cuda.synchronize()
  wall = time.perf_counter() - t0
  rec.update(bodies=nb, blobs=ri.Nblobs, scheme=ri.scheme, dt=0.01, solver_tolerance=1e-4, seconds=round(wall, 2),
             steps_per_s=round(steps / wall, 4), s_per_step=round(wall / steps, 4),
             gmres_iterations_per_step=ri.det_iterations_count / steps, lanczos_iterations_per_step=ri.stoch_iterations_count / steps,
             pair_sweeps_per_step=ri.susp.matvec_count / steps, rejected_steps=ri.invalid_configuration_count,
             mean_height_final=float(ri.location[:, 2].mean()))
print(json.dumps(rec), flush=True)
