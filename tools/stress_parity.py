"""Randomised differential test: symmetric path, one-sided sweep and the oracle on random sizes, kinds, wall modes,
periodic boxes and blob clouds (dilute / dense overlapping / partly below the wall).  Prints the worst relative error."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle
from rigidmultiblobswall_amd import MobilityContext
oracle.build()
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
names = {"tt": "trans_times_force", "tr": "trans_times_torque", "rt": "rot_times_force", "rr": "rot_times_torque"}
worst = 0.0
ctx = MobilityContext(0)
for case in range(n_cases):
  N = int(rng.choice([1, 2, 3, 63, 64, 65, 127, 128, 129, 200, 511, 777, 1500, 2500]))
  kind = str(rng.choice(["tt", "tr", "rt", "rr"]))
  wall = bool(rng.rand() < 0.7)
  a = float(0.1 + rng.rand())
  eta = float(0.5 + rng.rand())
  style = int(rng.randint(3))
  box = a * (N ** (1.0 / 3.0)) * [6.0, 2.2, 3.0][style]
  r = rng.rand(N, 3) * box
  if style == 2:
    r[:, 2] -= 0.15 * box                     # some blobs below the wall / below z = a
  elif wall:
    r[:, 2] += 1.05 * a
  periodic = rng.rand() < 0.35
  L = np.zeros(3)
  if periodic:
    L[0] = box * (1.0 + rng.rand())
    if rng.rand() < 0.5:
      L[1] = box * (1.0 + rng.rand())
    if not wall and rng.rand() < 0.3:
      L[2] = box * (1.0 + rng.rand())
  v = rng.randn(N, 3)
  pre = "single_wall" if wall else "no_wall"
  ref = getattr(oracle, "%s_mobility_%s_oracle" % (pre, names[kind]))(r, v, eta, a, periodic_length=L)
  ctx.set_positions(r, a, L, wall=wall)
  res = {}
  for det in (0, 1, 2):      # atomic symmetric path, one-sided sweep, ordered-reduction symmetric path
    ctx.set_option("deterministic", det)
    res[det] = ctx.matvec(kind, v, eta)
  ctx.set_option("deterministic", 0)
  nrm = np.linalg.norm(ref)
  if not np.isfinite(nrm):
    continue
  # a product that vanishes by symmetry (one blob and its +-L images in a coupling block) leaves rounding noise over
  # rounding noise: compare on the natural scale |v| / (8 pi eta a^2) instead
  scale = np.linalg.norm(v) / (8.0 * np.pi * eta * a * a)
  if nrm < 1e-9 * scale:
    continue
  errs = [np.linalg.norm(res[d] - ref) / max(nrm, 1e-300) for d in (0, 1, 2)]
  worst = max(worst, max(errs))
  if max(errs) > 1e-12:
    print("CASE %d N=%d kind=%s wall=%s style=%d L=%s errs=%s" % (case, N, kind, wall, style, L, errs), flush=True)
print("cases %d, worst relative error %.3e" % (n_cases, worst))
ctx.close()

# --- forces, fused tt+tr, source->target --------------------------------------------------------------------
from rigidmultiblobswall_amd import mobility as mob
from rigidmultiblobswall_amd.forces import calc_blob_blob_forces_hip
worst2 = 0.0
for case in range(n_cases // 3):
  N = int(rng.choice([2, 64, 129, 300, 1000, 2000]))
  a = float(0.1 + rng.rand()); eta = float(0.5 + rng.rand())
  box = a * (N ** (1.0 / 3.0)) * float(rng.choice([2.2, 5.0]))
  r = rng.rand(N, 3) * box; r[:, 2] += 0.5 * a
  L = np.zeros(3)
  if rng.rand() < 0.4:
    L[:2] = box * (1.0 + rng.rand(2))
  f, t = rng.randn(N, 3), rng.randn(N, 3)
  kw = dict(periodic_length=L, repulsion_strength=float(rng.rand() * 4), debye_length=float(a * (0.05 + rng.rand())), blob_radius=a)
  F, Fr = calc_blob_blob_forces_hip(r, **kw), oracle.calc_blob_blob_forces_oracle(r, **kw)
  e1 = np.linalg.norm(F - Fr) / max(np.linalg.norm(Fr), 1e-300)
  u = mob.single_wall_mobility_trans_times_force_torque_hip(r, f, t, eta, a, periodic_length=L)
  ur = oracle.single_wall_mobility_trans_times_force_torque_oracle(r, f, t, eta, a, periodic_length=L)
  e2 = np.linalg.norm(u - ur) / np.linalg.norm(ur)
  nt = int(rng.choice([1, 70, 400]))
  tgt = rng.rand(nt, 3) * box; tgt[:, 2] += 0.2 * a
  rs, rt = a * (0.3 + rng.rand(N)), a * rng.rand(nt) * (rng.rand() < 0.7)
  w = bool(rng.rand() < 0.6)
  fn = mob.single_wall_mobility_trans_times_force_source_target_hip if w else mob.no_wall_mobility_trans_times_force_source_target_hip
  fo = oracle.single_wall_mobility_trans_times_force_source_target_oracle if w else oracle.no_wall_mobility_trans_times_force_source_target_oracle
  s, sr = fn(r, tgt, f, rs, rt, eta), fo(r, tgt, f, rs, rt, eta)
  e3 = np.linalg.norm(s - sr) / np.linalg.norm(sr)
  if N >= 128:      # sources == targets: the symmetric radii path
    rad = a * (0.3 + rng.rand(N))
    s2, s2r = fn(r, r, f, rad, rad, eta, periodic_length=L), fo(r, r, f, rad, rad, eta, periodic_length=L)
    e3 = max(e3, np.linalg.norm(s2 - s2r) / np.linalg.norm(s2r))
  worst2 = max(worst2, e1, e2, e3)
  if max(e1, e2, e3) > 1e-10:
    print("CASE2 %d N=%d L=%s forces %.2e fused %.2e source_target %.2e" % (case, N, L, e1, e2, e3), flush=True)
print("forces / fused / source-target: %d cases, worst relative error %.3e" % (n_cases // 3, worst2))

# --- round 3: pair shards (what the ranks of a G-GPU run evaluate), atomic and bit-reproducible, forces, free surface ---
import torch
rng = np.random.RandomState(1234 + (int(sys.argv[1]) if len(sys.argv) > 1 else 0))
worst_shard = 0.0
ctx = MobilityContext(0)
for case in range(max(20, n_cases // 4)):
  N = int(rng.choice([5, 64, 65, 127, 128, 129, 300, 777, 1500, 4097, 9000]))
  G = int(rng.choice([2, 3, 5, 8]))
  kind = str(rng.choice(["tt", "tr", "rt", "rr"]))
  wall = bool(rng.rand() < 0.7)
  a, eta = float(0.1 + rng.rand()), float(0.5 + rng.rand())
  box = a * (N ** (1.0 / 3.0)) * float(rng.choice([6.0, 2.2]))
  r = rng.rand(N, 3) * box
  if wall:
    r[:, 2] += 1.05 * a
  L = np.zeros(3)
  if rng.rand() < 0.3:
    L[0] = box * (1.0 + rng.rand())
  v = rng.randn(N, 3)
  pre = "single_wall" if wall else "no_wall"
  ref = getattr(oracle, "%s_mobility_%s_oracle" % (pre, names[kind]))(r, v, eta, a, periodic_length=L)
  vd = torch.as_tensor(v.reshape(-1), device="cuda")
  ctx.set_positions(r, a, L, wall=wall)
  errs = []
  for det in (0, 2):
    ctx.set_option("deterministic", det)
    tot = sum(ctx.matvec_pairshard_device(kind, vd, eta, g, G).cpu().numpy() for g in range(G))
    errs.append(np.linalg.norm(tot - ref) / np.linalg.norm(ref))
    if det == 2:
      again = sum(ctx.matvec_pairshard_device(kind, vd, eta, g, G).cpu().numpy() for g in range(G))
      assert np.array_equal(again, tot), "deterministic shards differ between launches"
  ctx.set_option("deterministic", 0)
  eps, b = float(0.1 + rng.rand()), float(a * (0.1 + 0.4 * rng.rand()))
  ctx.set_positions(r, a, L, wall=False)
  F = sum(ctx.blob_blob_force_pairshard_device(eps, b, a, g, G).cpu().numpy() for g in range(G)).reshape(-1, 3)
  F_ref = oracle.calc_blob_blob_forces_oracle(r, periodic_length=L, repulsion_strength=eps, debye_length=b, blob_radius=a)
  errs.append(np.linalg.norm(F - F_ref) / max(np.linalg.norm(F_ref), 1e-300))
  if not np.any(L > 0):
    u = sum(ctx.matvec_pairshard_device("tt_free", vd, eta, g, G).cpu().numpy() for g in range(G))
    u_ref = oracle.free_surface_mobility_trans_times_force_oracle(r, v, eta, a)
    errs.append(np.linalg.norm(u - u_ref) / np.linalg.norm(u_ref))
  worst_shard = max(worst_shard, max(errs))
  if max(errs) > 1e-12:
    print("SHARD CASE %d N=%d G=%d kind=%s wall=%s L=%s errs=%s" % (case, N, G, kind, wall, L, errs), flush=True)
print("pair-shard cases %d, worst relative error %.3e" % (max(20, n_cases // 4), worst_shard))
ctx.close()

# --- round 4: workgroup-cooperative kernels, the multi-device engine (one GPU listed G times), sorted force culling ------
from rigidmultiblobswall_amd.multi import MultiContext
rng = np.random.RandomState(4321 + (int(sys.argv[1]) if len(sys.argv) > 1 else 0))
worst4 = 0.0
ctx = MobilityContext(0)
engines = {}
n4 = max(24, n_cases // 4)
for case in range(n4):
  N = int(rng.choice([3, 64, 65, 128, 129, 300, 777, 1500, 2049, 4097, 7000]))
  G = int(rng.choice([2, 3, 5, 8]))
  wall = bool(rng.rand() < 0.7)
  a, eta = float(0.1 + rng.rand()), float(0.5 + rng.rand())
  box = a * (N ** (1.0 / 3.0)) * float(rng.choice([6.0, 2.2]))
  r = rng.rand(N, 3) * box
  if wall:
    r[:, 2] += (1.05 if rng.rand() < 0.5 else 0.6) * a
  L = np.zeros(3)
  if rng.rand() < 0.3:
    L[int(rng.randint(2))] = box * (1.0 + rng.rand())
  v, w = rng.randn(N, 3), rng.randn(N, 3)
  pre = "single_wall" if wall else "no_wall"
  errs = []
  # (i) cooperative kernels forced, every kind + the fused row, against the oracle
  ctx.set_positions(r, a, L, wall=wall)
  ctx.set_option("sym_coop", 2)
  for kind in ("tt", "tr", "rt", "rr"):
    ref = getattr(oracle, "%s_mobility_%s_oracle" % (pre, names[kind]))(r, v, eta, a, periodic_length=L)
    nrm = np.linalg.norm(ref)
    if np.isfinite(nrm) and nrm > 1e-9 * np.linalg.norm(v) / (8.0 * np.pi * eta * a * a):
      errs.append(np.linalg.norm(ctx.matvec(kind, v, eta) - ref) / nrm)
  if wall:
    ref = oracle.single_wall_mobility_trans_times_force_torque_oracle(r, v, w, eta, a, periodic_length=L)
    errs.append(np.linalg.norm(ctx.matvec("tt_tr", v, eta, vec2=w) - ref) / np.linalg.norm(ref))
  ctx.set_option("sym_coop", 1)
  # (ii) the engine with the device listed G times: tt + rr + fused + forces against the oracle
  if G not in engines:
    engines[G] = MultiContext([0] * G)
  m = engines[G]
  m.set_option("deterministic", int(rng.choice([0, 2])))
  m.set_positions(r, a, L, wall)
  for kind in ("tt", "rr"):
    ref = getattr(oracle, "%s_mobility_%s_oracle" % (pre, names[kind]))(r, v, eta, a, periodic_length=L)
    errs.append(np.linalg.norm(m.matvec(kind, v, eta) - ref) / np.linalg.norm(ref))
  if wall:
    ref = oracle.single_wall_mobility_trans_times_force_torque_oracle(r, v, w, eta, a, periodic_length=L)
    errs.append(np.linalg.norm(m.matvec("tt_tr", v, eta, vec2=w) - ref) / np.linalg.norm(ref))
  # (iii) forces: sorted culling on a random permutation, single context and engine
  eps, b = float(0.1 + rng.rand()), float(a * (0.02 + 0.3 * rng.rand()))
  perm = rng.permutation(N)
  F_ref = oracle.calc_blob_blob_forces_oracle(r, periodic_length=L, repulsion_strength=eps, debye_length=b, blob_radius=a)
  ctx.set_positions(r[perm], a, L, wall=False)
  errs.append(np.linalg.norm(ctx.blob_blob_force(eps, b, a) - F_ref[perm]) / max(np.linalg.norm(F_ref), 1e-300))
  m.set_positions(r[perm], a, L, False)
  errs.append(np.linalg.norm(m.blob_blob_force(eps, b, a) - F_ref[perm]) / max(np.linalg.norm(F_ref), 1e-300))
  worst4 = max(worst4, max(errs))
  if max(errs) > 1e-11:
    print("R4 CASE %d N=%d G=%d wall=%s L=%s errs=%s" % (case, N, G, wall, L, ["%.1e" % e for e in errs]), flush=True)
print("round-4 cases (cooperative kernels, multi-device engine, sorted force culling) %d, worst relative error %.3e" % (n4, worst4))

# --- end of round 4: two target blobs per lane (sym2t_kernel) forced at random sizes, kinds, clouds, pair shards -------------
rng = np.random.RandomState(987 + (int(sys.argv[1]) if len(sys.argv) > 1 else 0))
worst2t, n2t = 0.0, max(40, n_cases // 3)
ctx2 = MobilityContext(0)
ctx2.set_option("sym_two_targets", 2)
for case in range(n2t):
  N = int(rng.choice([193, 256, 257, 320, 449, 777, 1025, 1500, 2049, 3000, 4097]))
  kind = str(rng.choice(["tt", "tr", "rt", "rr"]))
  wall = bool(rng.rand() < 0.7)
  a, eta = float(0.1 + rng.rand()), float(0.5 + rng.rand())
  style = int(rng.randint(3))
  box = a * (N ** (1.0 / 3.0)) * [6.0, 2.2, 3.0][style]
  r = rng.rand(N, 3) * box
  if style == 2:
    r[:, 2] -= 0.15 * box
  elif wall:
    r[:, 2] += 1.05 * a
  v = rng.randn(N, 3)
  ref = getattr(oracle, "%s_mobility_%s_oracle" % ("single_wall" if wall else "no_wall", names[kind]))(r, v, eta, a)
  nrm = np.linalg.norm(ref)
  if not (np.isfinite(nrm) and nrm > 0):
    continue
  ctx2.set_positions(r, a, None, wall=wall)
  got = ctx2.matvec(kind, v, eta)
  assert ctx2.get_option("last_path") == 4
  errs = [np.linalg.norm(got - ref) / nrm]
  import torch
  G = int(rng.choice([2, 3, 7]))
  vd = torch.as_tensor(v.reshape(-1), device="cuda")
  tot = sum(ctx2.matvec_pairshard_device(kind, vd, eta, g, G) for g in range(G)).cpu().numpy()
  errs.append(np.linalg.norm(tot - ref) / nrm)
  worst2t = max(worst2t, max(errs))
  if max(errs) > 1e-11:
    print("2T CASE %d N=%d kind=%s wall=%s style=%d G=%d errs=%s" % (case, N, kind, wall, style, G, ["%.1e" % e for e in errs]), flush=True)
ctx2.close()
print("two-targets-per-lane cases %d (whole products + pair shards against the oracle), worst relative error %.3e" % (n2t, worst2t))
for m in engines.values():
  m.close()
ctx.close()
