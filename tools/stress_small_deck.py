"""Soak of the small-deck path (helper kernels + captured Arnoldi iterations, preconditioner rebuilt every step): thousands
of integrator steps, the state checked for finiteness / unit quaternions, the solver for convergence, and the captured
graphs for being used.  python tools/stress_small_deck.py [det_steps] [stoch_steps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rigidmultiblobswall_amd import structures as st
from rigidmultiblobswall_amd.rigid_integrator import RigidIntegrator
R, eta = 1.0155, 0.957e-3
shell = st.icosahedron_shell(0.792079207921 * R)
a = st.min_blob_separation(shell) / 2
n_det = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
n_sto = int(sys.argv[2]) if len(sys.argv) > 2 else 400
shell42 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "g9_rigid_det_euler_42blob_shells.npz"))["vertex_shell42"]
a42 = st.min_blob_separation(shell42) / 2
for nb, scheme, tol, steps, dt, body, a in ((64, "deterministic_adams_bashforth", 1e-8, n_det, 0.002, shell, a), (30, "deterministic_midpoint", 1e-9, n_det // 3, 0.004, shell, a),
                                            (64, "stochastic_Slip_Trapz", 1e-6, n_sto, 0.002, shell, a), (100, "stochastic_first_order_RFD", 1e-6, n_sto, 0.002, shell, a),
                                            (24, "stochastic_Slip_Trapz", 1e-6, n_sto // 2, 0.002, shell42, a42), (24, "deterministic_adams_bashforth", 1e-8, n_det // 6, 0.002, shell42, a42)):
  loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=nb)
  integ = RigidIntegrator([body] * nb, loc, quat, scheme, a, eta, tolerance=tol, device="cuda:0", seed=3)
  integ.kT, integ.g = 0.0041419464, 0.0024892 * 12
  integ.repulsion_strength_wall, integ.debye_length_wall = 0.0165677856, 0.0656
  integ.repulsion_strength, integ.debye_length = 0.0165677856, 0.0656
  t0 = time.perf_counter()
  for step in range(steps):
    integ.advance_time_step(dt, step=step)
    if step % 250 == 249:
      q = integ.orientation
      assert bool(torch.isfinite(integ.location).all()) and float((torch.linalg.norm(q, dim=1) - 1).abs().max()) < 1e-10
      print("  %s step %d: %.2f ms/step, z in [%.2f, %.2f], gmres its %d, rejected %d" % (
          scheme, step + 1, 1e3 * (time.perf_counter() - t0) / (step + 1), float(integ.location[:, 2].min()), float(integ.location[:, 2].max()),
          integ.det_iterations_count, integ.invalid_configuration_count), flush=True)
  torch.cuda.synchronize()
  ws = getattr(integ.susp, "_arnoldi_ws", None)
  print("%s, %d bodies x %d blobs: %d steps ok, %.2f ms/step; graphs captured %s, replays %s; library Lanczos loops %d" % (
      scheme, nb, body.shape[0], steps, 1e3 * (time.perf_counter() - t0) / steps, None if ws is None else ws.captures, None if ws is None else ws.replays,
      integ.susp.lanczos_native_loop_calls), flush=True)
  assert bool(torch.isfinite(integ.location).all())
  integ.close()
print("SOAK OK")
