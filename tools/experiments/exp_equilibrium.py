"""Physics check of the Brownian roller steppers: non-driven rollers above the wall must keep the Gibbs-Boltzmann height
distribution P(h) ~ exp(-U(h)/kT), U = m g h + wall repulsion, whatever the hydrodynamic interactions -- which only
holds if the stochastic drift kT div(M) (random finite difference) and the noise amplitude are right."""
import math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rigidmultiblobswall_amd.rollers import RollersIntegrator

a, eta, kT, mg, ew, bw = 0.656, 1.0e-3, 0.0041419464, 0.0024892, 0.0165677856, 0.0656


def potential(h):
  return mg * h + np.where(h > a, ew * np.exp(-(h - a) / bw), ew + ew * (a - h) / bw)


def analytic_moments():
  h = np.linspace(0.0, 40.0, 400001)
  w = np.exp(-(potential(h) - potential(h).min()) / kT)
  Z = np.trapezoid(w, h)
  m1 = np.trapezoid(w * h, h) / Z
  m2 = np.trapezoid(w * h * h, h) / Z
  return m1, m2 - m1 * m1, h, np.cumsum(w) / np.sum(w)


def main(N=2048, steps=200, dt=0.016, scheme="stochastic_adams_bashforth_rollers", seed=3, spacing=5.0, drift=True,
         precision="double"):
  m1, var, h, cdf = analytic_moments()
  rng = np.random.RandomState(seed)
  side = int(math.ceil(math.sqrt(N)))
  ij = np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 2)[:N].astype(float)
  r0 = np.empty((N, 3))
  r0[:, :2] = ij * spacing * a
  r0[:, 2] = np.interp(rng.rand(N), cdf, h)             # start in equilibrium
  integ = RollersIntegrator(r0, scheme, a, eta, tolerance=1e-4, device="cuda:0", seed=seed)
  integ.kT, integ.g, integ.repulsion_strength_wall, integ.debye_length_wall = kT, mg, ew, bw
  integ.precision = precision
  if not drift:
    # drop the random-finite-difference term kT div(M): what an integrator without the stochastic drift would do
    integ._random_finite_difference = lambda kinds: [torch.zeros(3 * N, dtype=torch.float64, device="cuda:0") for _ in kinds]
    integ.compute_linear_thermal_drift = lambda: torch.zeros(3 * N, dtype=torch.float64, device="cuda:0")
  sums = np.zeros(2)
  count = 0
  for step in range(steps):
    integ.advance_time_step(dt)
    if step >= steps // 4:
      z = integ.location[:, 2]
      sums += [float(z.mean()), float((z * z).mean())]
      count += 1
  mean = sums[0] / count
  variance = sums[1] / count - mean * mean
  integ.close()
  return dict(analytic_mean=m1, analytic_var=var, mean=mean, var=variance, rejected=integ.invalid_configuration_count)


if __name__ == "__main__":
  for scheme in ("stochastic_adams_bashforth_rollers", "stochastic_mid_point_rollers", "stochastic_trapezoidal_rollers"):
    print(scheme, main(scheme=scheme), flush=True)
  print("no drift term:", main(drift=False), flush=True)
