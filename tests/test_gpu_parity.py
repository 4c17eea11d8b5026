"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle, the committed
golden vectors from the reference, and size-independent properties at full size.

Tolerances (fp64, relative L2 `|u - u_ref| / |u_ref|`, the reference's own metric,
mobility/test_blobs.py:129-135), from BASELINE.md section 3 / SURVEY.md section 8(d):
  1e-12  well-separated wall clouds (D2) and everything at N <= 1e4 vs the oracle
  1e-10  dense overlapping clouds (D1, test_blobs distribution: long cancelling sums)
  1e-13  G-shard vs 1-shard (same kernel, different target ranges)
"""
import numpy as np
import pytest

from conftest import KERNEL_KEYS, golden_files, load_golden, rel_err

pytestmark = pytest.mark.gpu

TOL_D2 = 1e-12
TOL_D1 = 1e-10
TOL_SHARD = 1e-13


@pytest.fixture(scope="module")
def mob():
  from rigidmultiblobswall_amd import mobility
  return mobility


@pytest.fixture(scope="module")
def Ctx():
  from rigidmultiblobswall_amd import MobilityContext
  return MobilityContext


def d1_cloud(N, seed=0):
  """mobility/test_blobs.py:31-44 at constant number density."""
  rng = np.random.RandomState(seed)
  eta, a = 7.0, 0.13
  s = (N / 1000.0) ** (1.0 / 3.0)
  return s * 5 * a * rng.rand(N, 3), rng.randn(N, 3), eta, a


def d2_cloud(N, seed=0):
  """5% volume fraction above the wall, no blob below z = 1.1a."""
  rng = np.random.RandomState(seed)
  a, eta = 0.5, 1.0
  Lbox = (N * (4.0 / 3.0) * np.pi * a ** 3 / 0.05) ** (1.0 / 3.0)
  r = rng.rand(N, 3) * Lbox
  r[:, 2] += 1.1 * a
  return r, rng.randn(N, 3), eta, a


ALL_STEMS = sorted(KERNEL_KEYS.values())


# ---------------------------------------------------------------------------------------------
# 1. golden vectors from the reference itself
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", golden_files("g[123]_*.npz"), ids=lambda p: p.split("/")[-1][:-4])
def test_golden_vectors(mob, path):
  g = load_golden(path)
  r, v, eta, a, L = g["r_vectors"], g["vector"], float(g["eta"]), float(g["a"]), g["periodic_length"]
  for key, stem in KERNEL_KEYS.items():
    if key not in g:
      continue
    u = getattr(mob, stem + "_hip")(r, v, eta, a, periodic_length=L)
    assert u.shape == (3 * len(r),)
    # 1e-12 for the well-separated D2-style cloud, 1e-10 for the dense / contact test_blobs clouds
    tol = TOL_D2 if "wall_cloud" in path else TOL_D1
    assert rel_err(u, g[key]) < tol, (key, rel_err(u, g[key]))


@pytest.mark.parametrize("path", golden_files("g5_*.npz"), ids=lambda p: p.split("/")[-1][:-4])
def test_golden_forces(path):
  from rigidmultiblobswall_amd.forces import calc_blob_blob_forces_hip, calc_blob_blob_forces_radii_hip
  g = load_golden(path)
  kw = dict(periodic_length=g["periodic_length"], repulsion_strength=float(g["repulsion_strength"]),
            debye_length=float(g["debye_length"]), blob_radius=float(g["blob_radius"]))
  if "radius_blobs" in g:
    F = calc_blob_blob_forces_radii_hip(g["r_vectors"], g["radius_blobs"], **kw)
  else:
    F = calc_blob_blob_forces_hip(g["r_vectors"], **kw)
  assert F.shape == g["force"].shape
  assert rel_err(F, g["force"]) < TOL_D2
  if "force_tree" in g:
    # the reference's k-d tree variant drops pairs beyond 2 a + 30 b: e^-30 of a contact force each
    from rigidmultiblobswall_amd.forces import calc_blob_blob_forces_tree_hip
    from rigidmultiblobswall_amd import dispatch
    assert dispatch.set_blob_blob_forces("tree_hip") is calc_blob_blob_forces_tree_hip
    Ft = calc_blob_blob_forces_tree_hip(g["r_vectors"], **kw)
    assert np.array_equal(Ft, F) and rel_err(Ft, g["force_tree"]) < 1e-12


# ---------------------------------------------------------------------------------------------
# 2. HIP vs oracle on seeded clouds, every kernel of the surface
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("stem", ALL_STEMS)
@pytest.mark.parametrize("N", [1, 2, 63, 64, 65, 513, 2000])
def test_vs_oracle_d2(mob, oracle, stem, N):
  r, v, eta, a = d2_cloud(N, seed=N)
  u = getattr(mob, stem + "_hip")(r, v, eta, a)
  ref = getattr(oracle, stem + "_oracle")(r, v, eta, a)
  assert np.all(np.isfinite(u))
  assert rel_err(u, ref) < TOL_D2, rel_err(u, ref)


@pytest.mark.parametrize("stem", ALL_STEMS)
def test_vs_oracle_d1_dense_overlapping(mob, oracle, stem):
  r, v, eta, a = d1_cloud(3000, seed=3)   # ~1/7 of the blobs below z = a: B-damping path
  assert np.sum(r[:, 2] < a) > 100
  u = getattr(mob, stem + "_hip")(r, v, eta, a)
  ref = getattr(oracle, stem + "_oracle")(r, v, eta, a)
  assert rel_err(u, ref) < TOL_D1, rel_err(u, ref)


# ---------------------------------------------------------------------------------------------
# 2b. sym2t_kernel (two target blobs per lane) DIRECTLY against the oracle / the reference's fixtures, at the sizes where
#     it is the default: every kind x wall / no wall, whole products and pair shards.  Reference
#     arithmetic: mobility_numba.py:124-287 (tt), :548-686 (tr), :938-1073 (rt), :1189-1328 (rr) and their no-wall twins.
# ---------------------------------------------------------------------------------------------
def _edge_sample(n, k=96, seed=1):
  """Targets incl. the edges of the 64-blob tiles, both rows of the first / last row pair and the partial last tile."""
  tg = np.random.RandomState(seed).choice(n, k, replace=False)
  edges = [0, 63, 64, 127, 128, 191, n - 1, n - 64, n - 65, n - 128, (n // 2) // 64 * 64, (n // 2) // 64 * 64 + 63]
  tg[:len(edges)] = [e for e in edges]
  return np.unique(tg[(tg >= 0) & (tg < n)])


def _oracle_on_targets(oracle, kind, wall, r, v, eta, a, tg):
  """The reference wrapper's result (shift_heights + B on both sides, mobility/mobility.py:1132-1163) on `tg` only."""
  if not wall:
    return oracle.raw_matvec_targets(kind, 0, r, v, eta, a, tg)
  r_eff, bdiag, _ = oracle.wall_regularisation(r, a)
  ref = oracle.raw_matvec_targets(kind, 1, r_eff, np.asarray(v).reshape(-1, 3) * bdiag[:, None], eta, a, tg)
  return (ref.reshape(-1, 3) * bdiag[tg][:, None]).reshape(-1)


@pytest.mark.parametrize("N", [10000, 24576, 24577])
@pytest.mark.parametrize("wall", [True, False], ids=["wall", "no_wall"])
def test_two_targets_kernel_vs_oracle_where_it_is_the_default(Ctx, oracle, N, wall):
  """1e4 (configs[1]), 24 576 (configs[2]) and 24 577 blobs (odd tile count + a one-blob last tile); D2 cloud lowered so
  that ~2 % of the blobs sit below z = a (height clamp + B-damping path); default options -- last_path must say sym2t."""
  import torch
  r, v, eta, a = d2_cloud(N, seed=N % 1000)
  r = r.copy(); r[:, 2] -= 0.45 * a          # z in [0.65 a, ...): some blobs below z = a
  assert not wall or np.sum(r[:, 2] < a) > 20
  tg = _edge_sample(N)
  vd = torch.as_tensor(v.reshape(-1), device="cuda")
  ctx = Ctx(0)
  try:
    ctx.set_positions(torch.as_tensor(r.reshape(-1), device="cuda"), a, None, wall)
    for kind in ("tt", "tr", "rt", "rr"):
      ref = _oracle_on_targets(oracle, kind, wall, r, v, eta, a, tg)
      u = ctx.matvec_device(kind, vd, eta).cpu().numpy()
      assert ctx.get_option("last_path") == 4, (kind, ctx.get_option("last_path"))      # rmb::sym2t_kernel<kind, wall>
      assert np.all(np.isfinite(u))
      e = rel_err(u.reshape(-1, 3)[tg].reshape(-1), ref)
      assert e < TOL_D2, (N, wall, kind, e)
      for G in (2, 8):        # pair shards of an N-GPU run, summed as the all-reduce would
        tot = torch.zeros(3 * N, dtype=torch.float64, device="cuda")
        for g in range(G):
          tot += ctx.matvec_pairshard_device(kind, vd, eta, g, G)
        e = rel_err(tot.cpu().numpy().reshape(-1, 3)[tg].reshape(-1), ref)
        assert e < TOL_D2, (N, wall, kind, G, e)
  finally:
    ctx.close()


@pytest.mark.parametrize("name", ["g1_test_blobs_N1000", "g1_test_blobs_N300", "g3_wall_cloud_N200"])
def test_two_targets_kernel_forced_on_the_reference_fixtures(Ctx, name):
  """The reference's own outputs (tests/golden, generated from the imported reference) with sym_two_targets = 2 forced, so
  the fixtures that normally run on the cooperative kernel meet every sym2t instantiation too."""
  import os
  import torch
  from conftest import GOLDEN
  g = load_golden(os.path.join(GOLDEN, name + ".npz"))
  r, v, eta, a = g["r_vectors"], g["vector"], float(g["eta"]), float(g["a"])
  assert not np.any(g["periodic_length"])
  vd = torch.as_tensor(np.ascontiguousarray(v).reshape(-1), device="cuda")
  ctx = Ctx(0)
  try:
    ctx.set_option("sym_two_targets", 2)
    for wall in (True, False):
      ctx.set_positions(torch.as_tensor(np.ascontiguousarray(r).reshape(-1), device="cuda"), a, None, wall)
      for kind in ("tt", "tr", "rt", "rr"):
        key = ("wall_" if wall else "no_wall_") + kind
        if key not in g:            # the N = 1000 fixture holds tt only (the reference's pure-Python kernels take minutes there)
          continue
        u = ctx.matvec_device(kind, vd, eta).cpu().numpy()
        assert ctx.get_option("last_path") == 4, (key, ctx.get_option("last_path"))
        tol = TOL_D2 if "wall_cloud" in name else TOL_D1
        assert rel_err(u, g[key]) < tol, (name, key, rel_err(u, g[key]))
  finally:
    ctx.close()


def test_config2_size_1e4_wall_tt(mob, oracle):
  """BASELINE.json configs[1]: 1e4 random blobs above a wall, single_wall_mobility_trans_times_force."""
  r, f, eta, a = d2_cloud(10000, seed=0)
  u = mob.single_wall_mobility_trans_times_force_hip(r, f, eta, a)
  ref = oracle.single_wall_mobility_trans_times_force_oracle(r, f, eta, a)
  assert rel_err(u, ref) < TOL_D2, rel_err(u, ref)


@pytest.mark.parametrize("stem", ["single_wall_mobility_trans_times_force", "no_wall_mobility_rot_times_torque",
                                  "single_wall_mobility_trans_times_torque", "single_wall_mobility_rot_times_force"])
@pytest.mark.parametrize("L", [(9.0, 11.0, 0.0), (7.5, 0.0, 0.0), (0.0, 8.0, 0.0)])
def test_pseudo_periodic(mob, oracle, stem, L):
  r, v, eta, a = d2_cloud(300, seed=7)
  L = np.array(L)
  u = getattr(mob, stem + "_hip")(r, v, eta, a, periodic_length=L)
  ref = getattr(oracle, stem + "_oracle")(r, v, eta, a, periodic_length=L)
  assert rel_err(u, ref) < TOL_D2, rel_err(u, ref)


@pytest.mark.parametrize("kind", ["tt", "tr", "rt", "rr"])
def test_periodic_symmetric_vs_sweep(Ctx, oracle, kind):
  """Pseudo-periodic images through both device paths (N large enough for the symmetric kernel,
  incl. a partial last tile and self-images)."""
  r, f, eta, a = d2_cloud(1000, seed=90)
  L = np.array([14.0, 16.0, 0.0])
  ctx = Ctx(0)
  ctx.set_positions(r, a, L, wall=True)
  u_sym = ctx.matvec(kind, f, eta)
  assert ctx.last_launch()["chunks"] == 0
  ctx.set_option("deterministic", 1)
  u_det = ctx.matvec(kind, f, eta)
  stem = {"tt": "trans_times_force", "tr": "trans_times_torque", "rt": "rot_times_force", "rr": "rot_times_torque"}[kind]
  ref = getattr(oracle, "single_wall_mobility_%s_oracle" % stem)(r, f, eta, a, periodic_length=L)
  assert rel_err(u_det, ref) < TOL_D2 and rel_err(u_sym, ref) < TOL_D2
  ctx.close()


def test_fully_periodic_no_wall(mob, oracle):
  r, v, eta, a = d2_cloud(200, seed=8)
  L = np.array([9.0, 10.0, 11.0])
  for stem in ("no_wall_mobility_trans_times_force", "no_wall_mobility_trans_times_torque"):
    u = getattr(mob, stem + "_hip")(r, v, eta, a, periodic_length=L)
    ref = getattr(oracle, stem + "_oracle")(r, v, eta, a, periodic_length=L)
    assert rel_err(u, ref) < TOL_D2


@pytest.mark.parametrize("wall", [True, False])
def test_fused_force_torque(mob, oracle, wall):
  """K11/K12 (mobility_pycuda.py:1266, :1394): one sweep == M_tt f + M_tr tau."""
  r, f, eta, a = d1_cloud(1500, seed=9)
  t = np.random.RandomState(10).randn(*f.shape)
  pre = "single_wall" if wall else "no_wall"
  u = getattr(mob, pre + "_mobility_trans_times_force_torque_hip")(r, f, t, eta, a)
  ref = getattr(oracle, pre + "_mobility_trans_times_force_torque_oracle")(r, f, t, eta, a)
  assert rel_err(u, ref) < TOL_D1
  u2 = getattr(mob, pre + "_mobility_trans_times_force_hip")(r, f, eta, a) + \
      getattr(mob, pre + "_mobility_trans_times_torque_hip")(r, t, eta, a)
  assert rel_err(u, u2) < 1e-13


@pytest.mark.parametrize("wall", [True, False])
@pytest.mark.parametrize("L", [(0.0, 0.0, 0.0), (9.0, 8.0, 0.0)])
def test_fused_force_torque_symmetric_two_pass(Ctx, oracle, wall, L):
  """K11/K12 at N >= 128 run as two symmetric passes into one output (rmb_entry.hip); must equal the one-sided
  fused sweep (option deterministic) and the oracle."""
  import torch
  r, f, eta, a = d2_cloud(1500, seed=31)
  t = np.random.RandomState(32).randn(*f.shape)
  ctx = Ctx(0)
  ctx.set_positions(r, a, np.array(L), wall=wall)
  u_sym = ctx.matvec("tt_tr", f, eta, vec2=t)
  assert ctx.last_launch()["chunks"] == 0
  ctx.set_option("deterministic", 1)
  u_sweep = ctx.matvec("tt_tr", f, eta, vec2=t)
  assert ctx.last_launch()["chunks"] >= 1
  ctx.set_option("deterministic", 0)
  ctx.set_option("fused_symmetric", 0)
  u_off = ctx.matvec("tt_tr", f, eta, vec2=t)
  assert ctx.last_launch()["chunks"] >= 1
  pre = "single_wall" if wall else "no_wall"
  ref = getattr(oracle, pre + "_mobility_trans_times_force_torque_oracle")(r, f, t, eta, a, periodic_length=np.array(L))
  assert rel_err(u_sym, ref) < TOL_D2 and rel_err(u_sweep, ref) < TOL_D2 and rel_err(u_off, ref) < TOL_D2
  assert rel_err(u_sym, u_sweep) < 1e-13
  ctx.close()


# ---------------------------------------------------------------------------------------------
# 3. boundary behaviour the callers rely on
# ---------------------------------------------------------------------------------------------
def test_inputs_not_mutated_flat_and_strided_inputs(mob, oracle):
  r, f, eta, a = d1_cloud(700, seed=11)
  r0, f0 = r.copy(), f.copy()
  big = np.zeros(3 * 700 + 50)
  big[:2100] = f.reshape(-1)
  u_view = mob.single_wall_mobility_trans_times_force_hip(r, big[0:2100], eta, a)      # slice, as multi_bodies.py:445
  u_flat = mob.single_wall_mobility_trans_times_force_hip(r.reshape(-1), f.reshape(-1), eta, a)
  fs = np.asfortranarray(f)                                                                # non C-contiguous
  u_f = mob.single_wall_mobility_trans_times_force_hip(r, fs, eta, a)
  u = mob.single_wall_mobility_trans_times_force_hip(r, f, eta, a, step=3, update_PC=1)   # extra kwargs ignored
  assert np.array_equal(r, r0) and np.array_equal(f, f0)
  for other in (u_view, u_flat, u_f):
    assert rel_err(other, u) < 1e-14      # default tt path accumulates with atomics: equal to rounding
  assert u.dtype == np.float64 and u.flags["C_CONTIGUOUS"] and u.flags["OWNDATA"]


def test_empty_input(mob):
  u = mob.single_wall_mobility_trans_times_force_hip(np.zeros((0, 3)), np.zeros((0, 3)), 1.0, 0.1)
  assert u.shape == (0,)


def test_blob_at_and_below_wall_has_zero_mobility(mob, oracle):
  r, f, eta, a = d2_cloud(100, seed=12)
  r[0, 2] = 0.0
  r[1, 2] = -0.3 * a          # below the wall: B negative, as the reference computes it
  r[2, 2] = a                 # exactly z = a: clamped by `<=`, B = 1
  u = mob.single_wall_mobility_trans_times_force_hip(r, f, eta, a)
  ref = oracle.single_wall_mobility_trans_times_force_oracle(r, f, eta, a)
  assert np.all(u.reshape(-1, 3)[0] == 0.0)
  assert rel_err(u, ref) < TOL_D2


def test_coincident_blobs_follow_reference_nan_policy(mob, oracle):
  """Two distinct blobs at the same point: the reference divides by zero (mobility_numba.py:214)."""
  r, f, eta, a = d2_cloud(10, seed=13)
  r[4] = r[7]
  u = mob.no_wall_mobility_trans_times_force_hip(r, f, eta, a).reshape(-1, 3)
  assert not np.all(np.isfinite(u[4])) and not np.all(np.isfinite(u[7]))
  ok = [i for i in range(10) if i not in (4, 7)]
  assert np.all(np.isfinite(u[ok]))


def test_positions_cache_is_invalidated(mob, oracle):
  r, f, eta, a = d2_cloud(400, seed=14)
  u1 = mob.single_wall_mobility_trans_times_force_hip(r, f, eta, a)
  r2 = r.copy()
  r2[17, 0] += 1e-6 * a          # RFD-sized displacement (doc/README.md:512-523) must be seen
  u2 = mob.single_wall_mobility_trans_times_force_hip(r2, f, eta, a)
  ref2 = oracle.single_wall_mobility_trans_times_force_oracle(r2, f, eta, a)
  assert rel_err(u2, ref2) < TOL_D2
  assert not np.array_equal(u1, u2)
  u3 = mob.single_wall_mobility_trans_times_force_hip(r, f, eta, 1.01 * a)
  assert rel_err(u3, oracle.single_wall_mobility_trans_times_force_oracle(r, f, eta, 1.01 * a)) < TOL_D2


def test_rfd_difference_is_resolved(mob, oracle):
  """(M(r + d) - M(r)) f / delta with delta = 1e-6 a (quaternion_integrator_multi_bodies.py:1007)."""
  r, f, eta, a = d2_cloud(500, seed=15)
  W = np.random.RandomState(16).randn(*r.shape)
  delta = 1e-6 * a
  g_hip = (mob.single_wall_mobility_trans_times_force_hip(r + delta * W, f, eta, a) -
           mob.single_wall_mobility_trans_times_force_hip(r, f, eta, a)) / delta
  g_ref = (oracle.single_wall_mobility_trans_times_force_oracle(r + delta * W, f, eta, a) -
           oracle.single_wall_mobility_trans_times_force_oracle(r, f, eta, a)) / delta
  assert rel_err(g_hip, g_ref) < 1e-5


# ---------------------------------------------------------------------------------------------
# 4. persistent context, chunking, target shards
# ---------------------------------------------------------------------------------------------
def test_context_chunk_count_does_not_change_result(Ctx):
  r, f, eta, a = d2_cloud(5000, seed=17)
  ctx = Ctx(0)
  ctx.set_option("deterministic", 1)      # the chunked sweep kernel (the symmetric kernel has no chunks)
  ctx.set_positions(r, a, wall=True)
  outs = []
  for chunks in (1, 2, 7, 10):
    ctx.set_option("chunks", chunks)
    outs.append(ctx.matvec("tt", f, eta))
    assert ctx.last_launch()["chunks"] == chunks
  ctx.set_option("chunks", 0)
  outs.append(ctx.matvec("tt", f, eta))
  for o in outs[1:]:
    assert rel_err(o, outs[0]) < TOL_SHARD
  ctx.close()


@pytest.mark.parametrize("G", [2, 3, 8])
def test_target_shards_equal_single_range(Ctx, G):
  """What each rank of a G-GPU job computes (rmb_set_target_range) concatenates to the 1-GPU result."""
  from rigidmultiblobswall_amd.distributed import partition
  r, f, eta, a = d1_cloud(4001, seed=18)
  ctx = Ctx(0)
  ctx.set_positions(r, a, wall=True)
  for kind in ("tt", "rr"):
    full = ctx.matvec(kind, f, eta)
    parts = []
    for g in range(G):
      b, e, _ = partition(len(r), G, g)
      ctx.set_target_range(b, e)
      parts.append(ctx.matvec(kind, f, eta))
      assert parts[-1].shape == (3 * (e - b),)
    ctx.set_target_range(0, len(r))
    assert rel_err(np.concatenate(parts), full) < TOL_SHARD
  ctx.close()


@pytest.mark.parametrize("kind", ["tt", "tr", "rt", "rr"])
@pytest.mark.parametrize("wall", [True, False])
@pytest.mark.parametrize("N", [128, 1000, 4097, 20000])
def test_symmetric_kernel_matches_deterministic_sweep(Ctx, oracle, wall, N, kind):
  """tt / tr / rt / rr have two device paths: sym_kernel (each unordered pair once, atomics) and sweep_kernel
  (every ordered pair, atomic-free).  Both must agree to rounding and with the oracle."""
  r, f, eta, a = d1_cloud(N, seed=30 + N) if N <= 4097 else d2_cloud(N, seed=30)
  ctx = Ctx(0)
  ctx.set_positions(r, a, wall=wall)
  u_sym = ctx.matvec(kind, f, eta)
  assert ctx.last_launch()["chunks"] == 0
  u_sym2 = ctx.matvec(kind, f, eta)
  ctx.set_option("deterministic", 1)
  u_det = ctx.matvec(kind, f, eta)
  assert ctx.last_launch()["chunks"] >= 1
  u_det2 = ctx.matvec(kind, f, eta)
  assert np.array_equal(u_det, u_det2)              # sweep path is bit-reproducible
  assert rel_err(u_sym, u_det) < 1e-13 and rel_err(u_sym2, u_det) < 1e-13
  if N <= 4097:
    pre = "single_wall" if wall else "no_wall"
    stem = {"tt": "trans_times_force", "tr": "trans_times_torque", "rt": "rot_times_force", "rr": "rot_times_torque"}[kind]
    ref = getattr(oracle, "%s_mobility_%s_oracle" % (pre, stem))(r, f, eta, a)
    assert rel_err(u_sym, ref) < TOL_D1
  ctx.close()


@pytest.mark.parametrize("kind", ["tt", "tr", "rt", "rr"])
@pytest.mark.parametrize("G", [2, 3, 8])
@pytest.mark.parametrize("wall", [True, False])
def test_pair_shards_sum_to_full_product(Ctx, oracle, G, wall, kind):
  """What the G ranks of a pair-sharded job compute (rmb_matvec_pairshard_device), summed as the
  all-reduce would, equals the single-GPU product; each shard's output covers all targets."""
  import torch
  N = 3001
  r, f, eta, a = d1_cloud(N, seed=40)
  ctx = Ctx(0)
  ctx.set_positions(torch.as_tensor(r.reshape(-1), device="cuda"), a, wall=wall)
  fd = torch.as_tensor(f.reshape(-1), device="cuda")
  total = torch.zeros(3 * N, dtype=torch.float64, device="cuda")
  norms = []
  for g in range(G):
    part = ctx.matvec_pairshard_device(kind, fd, eta, g, G)
    norms.append(float(part.norm()))
    total += part
  full = ctx.matvec_device(kind, fd, eta)
  torch.cuda.synchronize()
  pre = "single_wall" if wall else "no_wall"
  stem = {"tt": "trans_times_force", "tr": "trans_times_torque", "rt": "rot_times_force", "rr": "rot_times_torque"}[kind]
  ref = getattr(oracle, "%s_mobility_%s_oracle" % (pre, stem))(r, f, eta, a)
  assert rel_err(total.cpu().numpy(), ref) < TOL_D1
  assert rel_err(total.cpu().numpy(), full.cpu().numpy()) < 1e-13
  assert all(x > 0 for x in norms)
  ctx.close()


def test_sharded_replicated_world1(oracle):
  import torch
  from rigidmultiblobswall_amd.distributed import HipBackend, ShardedMobility
  r, f, eta, a = d2_cloud(2000, seed=41)
  sm = ShardedMobility(HipBackend("cuda:0"), device="cuda:0")
  sm.set_positions(r, a, wall=True)
  for kind, stem in (("tt", "trans_times_force"), ("rr", "rot_times_torque")):
    u = sm.matvec_replicated(kind, f, eta)
    torch.cuda.synchronize()
    ref = getattr(oracle, "single_wall_mobility_%s_oracle" % stem)(r, f, eta, a)
    assert rel_err(u.cpu().numpy(), ref) < TOL_D2


def test_device_resident_path_and_timing(Ctx, oracle):
  import torch
  r, f, eta, a = d2_cloud(3000, seed=19)
  ctx = Ctx(0)
  ctx.set_option("timing", 1)
  rd = torch.as_tensor(r, device="cuda")
  fd = torch.as_tensor(f.reshape(-1), device="cuda")
  ctx.set_positions(rd, a, wall=True)
  out = torch.empty(3 * 3000, dtype=torch.float64, device="cuda")
  for _ in range(3):
    ctx.matvec_device("tt", fd, eta, out=out)
  torch.cuda.synchronize()
  ref = oracle.single_wall_mobility_trans_times_force_oracle(r, f, eta, a)
  assert rel_err(out.cpu().numpy(), ref) < TOL_D2
  ms = ctx.timing_collect()
  assert len(ms) == 3 and np.all(ms > 0)
  ctx.close()


def test_device_path_is_ordered_against_torch_kernels(Ctx, oracle):
  """Regression: torch kernels produce the input and consume the output of rmb_matvec_device with no
  host synchronisation in between (what a device-resident Krylov loop does)."""
  import torch
  r, f, eta, a = d2_cloud(2500, seed=50)
  ctx = Ctx(0)
  ctx.set_positions(torch.as_tensor(r.reshape(-1), device="cuda"), a, wall=True)
  base = torch.as_tensor(f.reshape(-1), device="cuda")
  acc = torch.zeros_like(base)
  for k in range(20):
    v = base * float(k + 1) + 0.0          # produced by a torch kernel right before the call
    u = ctx.matvec_device("tt", v, eta)
    acc += u / float(k + 1)                # consumed by a torch kernel right after
    del v, u
  torch.cuda.synchronize()
  ref = oracle.single_wall_mobility_trans_times_force_oracle(r, f, eta, a)
  assert rel_err(acc.cpu().numpy() / 20.0, ref) < TOL_D2
  ctx.close()


def test_sharded_mobility_single_process_world1(oracle):
  """ShardedMobility with the HIP backend, world size 1 (the multi-rank logic is covered on CPU/gloo)."""
  import torch
  from rigidmultiblobswall_amd.distributed import HipBackend, ShardedMobility
  r, f, eta, a = d2_cloud(1000, seed=20)
  sm = ShardedMobility(HipBackend("cuda:0"), device="cuda:0")
  sm.set_positions(r, a, wall=True)
  u = sm.matvec("tt", f, eta)
  torch.cuda.synchronize()
  ref = oracle.single_wall_mobility_trans_times_force_oracle(r, f, eta, a)
  assert rel_err(u.cpu().numpy(), ref) < TOL_D2


# ---------------------------------------------------------------------------------------------
# 5. full-size (BASELINE.json sizes) through size-independent properties + oracle spot checks
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N", [24576, 100000])
def test_large_linearity_symmetry_and_spot_check(Ctx, oracle, N):
  r, f, eta, a = d2_cloud(N, seed=21)
  g = np.random.RandomState(22).randn(*f.shape)
  ctx = Ctx(0)
  ctx.set_positions(r, a, wall=True)
  Mf = ctx.matvec("tt", f, eta)
  Mg = ctx.matvec("tt", g, eta)
  # linearity
  Mfg = ctx.matvec("tt", 2.0 * f - 0.5 * g, eta)
  assert rel_err(Mfg, 2.0 * Mf - 0.5 * Mg) < 1e-12
  # symmetry  g.M f = f.M g,  positivity f.M f > 0   (mobility_test.py:101-130)
  gMf, fMg = np.dot(g.reshape(-1), Mf), np.dot(f.reshape(-1), Mg)
  assert abs(gMf - fMg) < 1e-11 * (np.linalg.norm(g) * np.linalg.norm(Mf))
  assert np.dot(f.reshape(-1), Mf) > 0
  # transpose pair  g.M_tr f = f.M_rt g
  Mtr_f = ctx.matvec("tr", f, eta)
  Mrt_g = ctx.matvec("rt", g, eta)
  assert abs(np.dot(g.reshape(-1), Mtr_f) - np.dot(f.reshape(-1), Mrt_g)) < 1e-11 * np.linalg.norm(g) * np.linalg.norm(Mtr_f)
  # oracle on a random subset of targets (all N sources each)
  tg = np.random.RandomState(23).choice(N, 48, replace=False)
  r_eff, b, _ = oracle.wall_regularisation(r, a)
  ref = oracle.raw_matvec_targets("tt", 1, r_eff, f, eta, a, tg)
  got = Mf.reshape(-1, 3)[tg].reshape(-1)
  assert rel_err(got, ref) < TOL_D2
  ctx.close()


# 300007: N % 64 = 39, a partial last tile;  2200007: past BASELINE's largest size -- 34 376 tiles, 3.8e10 rotation steps
# (beyond 2^32), 2.4e12 pair evaluations, ~6.5 s per product
@pytest.mark.parametrize("N", [262144, 300007, 1000000, 2200007])
def test_baseline_full_sizes_spot_check_and_symmetry(Ctx, oracle, N):
  """BASELINE.json configs[4] / configs[3] sizes on one GPU (and one size beyond them): oracle on a sample of targets (all N
  sources each), reciprocity g.Mf = f.Mg, and the symmetric path against the one-sided sweep on the same sample."""
  import torch
  r, f, eta, a = d2_cloud(N, seed=31)
  g = np.random.RandomState(32).randn(*f.shape)
  ctx = Ctx(0)
  rd = torch.as_tensor(r.reshape(-1), device="cuda")
  fd, gd = torch.as_tensor(f.reshape(-1), device="cuda"), torch.as_tensor(g.reshape(-1), device="cuda")
  ctx.set_positions(rd, a, wall=True)
  Mf = ctx.matvec_device("tt", fd, eta)
  assert ctx.last_launch()["chunks"] == 0
  Mg = ctx.matvec_device("tt", gd, eta)
  gMf, fMg = float(torch.dot(gd, Mf)), float(torch.dot(fd, Mg))
  assert abs(gMf - fMg) < 1e-11 * float(torch.linalg.norm(gd) * torch.linalg.norm(Mf))
  # 24 random targets plus deterministic picks on the tile structure of the symmetric kernel: first / last blob of
  # the first tile, first blob of the second, the tile-row boundary in the middle, and the last (partial) tile
  last_tile = 64 * ((N - 1) // 64)
  picks = [0, 63, 64, 64 * (N // 128) - 1, 64 * (N // 128), last_tile - 1, last_tile, N - 1]
  tg = np.unique(np.concatenate([np.random.RandomState(33).choice(N, 24, replace=False), np.array(picks)]))
  r_eff, b, _ = oracle.wall_regularisation(r, a)
  ref = oracle.raw_matvec_targets("tt", 1, r_eff, f, eta, a, tg)
  got = Mf.cpu().numpy().reshape(-1, 3)[tg].reshape(-1)
  assert rel_err(got, ref) < TOL_D2
  ctx.close()


def test_forces_vs_oracle_and_newton_third_law(oracle):
  from rigidmultiblobswall_amd.forces import calc_blob_blob_forces_hip
  rng = np.random.RandomState(24)
  N, a, b, eps = 5000, 0.13, 0.01, 3.92
  r = rng.rand(N, 3) * (N ** (1.0 / 3.0)) * 2.2 * a
  for L in (np.zeros(3), np.array([3.0, 3.5, 0.0])):
    kw = dict(periodic_length=L, repulsion_strength=eps, debye_length=b, blob_radius=a)
    F = calc_blob_blob_forces_hip(r, **kw)
    ref = oracle.calc_blob_blob_forces_oracle(r, **kw)
    assert rel_err(F, ref) < TOL_D2
    assert np.abs(F.sum(axis=0)).max() < 1e-9 * np.abs(F).sum()     # pairwise antisymmetric


# ---------------------------------------------------------------------------------------------
# 6. dense builders (mobility/mobility.py:967-1013, :1018-1116) against the reference's dense products
# ---------------------------------------------------------------------------------------------
def test_dense_builders_match_reference_dense_products(mob):
  g = load_golden(golden_files("g3_wall_cloud_N200.npz")[0])
  r, f, eta, a = g["r_vectors"], g["vector"].reshape(-1), float(g["eta"]), float(g["a"])
  M = mob.single_wall_fluid_mobility_hip(r, eta, a)
  assert M.shape == (600, 600)
  assert rel_err(M @ f, g["dense_wall_tt"]) < 1e-13
  assert np.abs(M - M.T).max() < 1e-13 * np.abs(M).max() and np.linalg.eigvalsh(0.5 * (M + M.T)).min() > 0
  M0 = mob.rotne_prager_tensor_hip(r, eta, a)
  assert rel_err(M0 @ f, g["dense_no_wall_tt"]) < 1e-13


@pytest.mark.parametrize("L", [(0.0, 0.0, 0.0), (3.0, 3.5, 0.0)])
def test_symmetric_force_kernel_matches_sweep_and_oracle(Ctx, oracle, L):
  """K15 has two device paths too: sym_force_kernel (F_ji = -F_ij, each pair once) and force_sweep_kernel."""
  rng = np.random.RandomState(70)
  N, a, b, eps = 3000, 0.13, 0.01, 3.92
  r = rng.rand(N, 3) * (N ** (1.0 / 3.0)) * 2.2 * a
  L = np.array(L)
  ctx = Ctx(0)
  ctx.set_positions(r, a, L, wall=False)
  F_sym = ctx.blob_blob_force(eps, b, a)
  assert ctx.last_launch()["chunks"] == 0
  ctx.set_option("deterministic", 1)
  F_det = ctx.blob_blob_force(eps, b, a)
  assert ctx.last_launch()["chunks"] >= 1
  ref = oracle.calc_blob_blob_forces_oracle(r, periodic_length=L, repulsion_strength=eps, debye_length=b, blob_radius=a)
  assert rel_err(F_det, ref) < TOL_D2 and rel_err(F_sym, ref) < TOL_D2
  assert np.abs(F_sym.sum(axis=0)).max() < 1e-10 * np.abs(F_sym).sum()
  ctx.close()


def test_radii_forces_vs_oracle_and_equal_radii_limit(Ctx, oracle):
  """Per-blob radii (forces_numba.py:73-137): against the oracle on a cloud with chunked sources and a target
  sub-range, and equal to the single-radius kernel when all radii are a."""
  import torch
  rng = np.random.RandomState(81)
  N, a, b, eps = 3000, 0.13, 0.02, 3.92
  r = rng.rand(N, 3) * (N ** (1.0 / 3.0)) * 2.2 * a
  rad = a * (0.4 + 1.2 * rng.rand(N))
  ctx = Ctx(0)
  for L in (np.zeros(3), np.array([3.0, 3.5, 0.0])):
    ctx.set_positions(r, 1.0, L, wall=False)
    F = ctx.blob_blob_force_radii(rad, eps, b)
    ref = oracle.calc_blob_blob_forces_radii_oracle(r, rad, periodic_length=L, repulsion_strength=eps, debye_length=b)
    assert rel_err(F, ref) < TOL_D2
    Fd = ctx.blob_blob_force_radii_device(torch.as_tensor(rad, device="cuda"), eps, b).cpu().numpy().reshape(-1, 3)
    assert rel_err(Fd, ref) < TOL_D2
    ctx.set_target_range(1000, 1777)
    assert rel_err(ctx.blob_blob_force_radii(rad, eps, b), ref[1000:1777]) < TOL_D2
    ctx.set_target_range(0, N)
    same = ctx.blob_blob_force_radii(np.full(N, a), eps, b)
    ctx.set_option("deterministic", 1)
    assert rel_err(same, ctx.blob_blob_force(eps, b, a)) < 1e-14
    ctx.set_option("deterministic", 0)
  ctx.close()


def test_force_edge_separations_elementwise(Ctx, oracle):
  """exp_nonpositive (pair_ops.h) over its whole range: gaps from r = 2a exactly up to (r - 2a)/b = 900
  (exp underflows through the denormals to 0), a coincident pair (r = 0) and a pair closer than 1e-25
  (forces_numba.py:44-47 clamps r there).  Compared per component, not in norm."""
  a, b, eps = 0.13, 0.01, 3.92
  n_line = 200
  gaps = 2 * a + b * np.linspace(0.0, 900.0, n_line)
  r = np.zeros((n_line + 4, 3))
  r[:n_line, 0] = 50.0 + np.cumsum(gaps)
  r[n_line] = r[n_line + 1] = (-40.0, 3.0, 1.0)                   # coincident
  r[n_line + 2] = (0.0, 0.0, 0.0); r[n_line + 3] = (1e-30, 0.0, 0.0)  # r < 1e-25
  ref = oracle.calc_blob_blob_forces_oracle(r, periodic_length=np.zeros(3), repulsion_strength=eps, debye_length=b,
                                            blob_radius=a)
  ctx = Ctx(0)
  ctx.set_positions(r, a, np.zeros(3), wall=False)
  for det in (0, 1):
    ctx.set_option("deterministic", det)
    F = ctx.blob_blob_force(eps, b, a)
    assert np.all(np.isfinite(F))
    assert np.all(np.abs(F - ref) <= 1e-12 * np.abs(ref) + 1e-290), np.abs(F - ref).max()
  ctx.close()


@pytest.mark.parametrize("span,zmax", [(3000.0, 1.2), (30000.0, 1.02), (1e6, 1.0)])
def test_flat_layer_far_field_cancellation(mob, oracle, span, zmax):
  """A monolayer on the wall, thousands of radii wide: every far pair is the near-cancellation of the RPY tensor with
  its image, and the closed-form block works with 1 - rho^2/R^2 and 1 - r^2/R^2 there (pair_blocks.h)."""
  N = 2500
  rng = np.random.RandomState(int(span) % 9973)
  a, eta = 0.5, 1.0
  r = np.column_stack([rng.rand(N) * span * a, rng.rand(N) * span * a, a * (1.0 + (zmax - 1.0) * rng.rand(N))])
  f = rng.randn(N, 3)
  for nm in ("trans_times_force", "trans_times_torque", "rot_times_force", "rot_times_torque"):
    u = getattr(mob, "single_wall_mobility_" + nm + "_hip")(r, f, eta, a)
    ref = getattr(oracle, "single_wall_mobility_" + nm + "_oracle")(r, f, eta, a)
    assert rel_err(u, ref) < 1e-13, (nm, rel_err(u, ref))


# ---------------------------------------------------------------------------------------------
# 7. source -> target products with per-blob radii (K13)
# ---------------------------------------------------------------------------------------------
def test_source_target_golden(mob):
  g = load_golden(golden_files("g4_source_target.npz")[0])
  for name in ("small", "mixed", "periodic"):
    args = [g[name + "_" + k] for k in ("source", "target", "force", "radius_source", "radius_target")]
    for wall, fn in ((1, mob.single_wall_mobility_trans_times_force_source_target_hip),
                     (0, mob.no_wall_mobility_trans_times_force_source_target_hip),
                     (2, mob.free_surface_mobility_trans_times_force_source_target_hip)):
      u = fn(*args, float(g[name + "_eta"]), periodic_length=g[name + "_L"])
      assert u.shape == (3 * len(args[1]),)
      assert rel_err(u, g["%s_wall%d" % (name, wall)]) < TOL_D1, (name, wall, rel_err(u, g["%s_wall%d" % (name, wall)]))
  u = mob.mobility_radii_trans_times_force(g["mixed_source"], g["mixed_force"], float(g["mixed_eta"]), 0.3, g["mixed_radius_source"],
                                           mob.single_wall_mobility_trans_times_force_source_target_hip)
  assert rel_err(u, g["mixed_radii_self_wall1"]) < TOL_D1
  # the reference's pure-Python twins of the two products (mobility.py:830-960) under their own names; they take no
  # periodic_length (the reference: "pseudo-PBC are not implemented for this function"), so the open-boundary cases
  for name in ("small", "mixed"):
    args = [g[name + "_" + k] for k in ("source", "target", "force", "radius_source", "radius_target")]
    for wall, fn in ((1, mob.mobility_vector_product_source_target_one_wall_hip), (0, mob.mobility_vector_product_source_target_unbounded_hip)):
      assert rel_err(fn(*args, float(g[name + "_eta"])), g["%s_wall%d" % (name, wall)]) < TOL_D1


@pytest.mark.parametrize("ns,nt", [(1, 1), (3, 700), (5000, 64), (4000, 3000)])
def test_source_target_vs_oracle(mob, oracle, ns, nt):
  rng = np.random.RandomState(ns + nt)
  box = (max(ns, nt) ** (1.0 / 3.0)) * 1.2
  src = rng.rand(ns, 3) * box
  tgt = rng.rand(nt, 3) * box
  rs = 0.1 + 0.4 * rng.rand(ns)
  rt = 0.1 + 0.4 * rng.rand(nt)
  f = rng.randn(ns, 3)
  for pre in ("single_wall", "no_wall", "free_surface"):
    u = getattr(mob, pre + "_mobility_trans_times_force_source_target_hip")(src, tgt, f, rs, rt, 0.8)
    ref = getattr(oracle, pre + "_mobility_trans_times_force_source_target_oracle")(src, tgt, f, rs, rt, 0.8)
    assert np.all(np.isfinite(u))
    assert rel_err(u, ref) < TOL_D1, rel_err(u, ref)


def test_source_target_equal_radii_reduces_to_tt(mob):
  """With one radius and sources == targets the K13 kernel must reproduce the single-radius tt product."""
  r, f, eta, a = d2_cloud(1500, seed=80)
  ones = np.full(len(r), a)
  u13 = mob.mobility_radii_trans_times_force(r, f, eta, a, ones, mob.single_wall_mobility_trans_times_force_source_target_hip)
  u1 = mob.single_wall_mobility_trans_times_force_hip(r, f, eta, a)
  assert rel_err(u13, u1) < 1e-12


# ---------------------------------------------------------------------------------------------
# 8. two vectors in one pass (rmb_matvec2_device)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("wall", [True, False])
@pytest.mark.parametrize("L", [(0.0, 0.0, 0.0), (0.0, 7.5, 0.0), (9.0, 8.0, 0.0)])
@pytest.mark.parametrize("N", [127, 128, 1000, 4100])
def test_two_vector_product_equals_two_products(Ctx, oracle, wall, L, N):
  import torch
  r, f, eta, a = d1_cloud(N, seed=N) if N == 1000 else d2_cloud(N, seed=N)
  g = np.random.RandomState(N + 1).randn(*f.shape)
  ctx = Ctx(0)
  ctx.set_positions(r, a, np.array(L), wall=wall)
  fd, gd = torch.as_tensor(f.reshape(-1), device="cuda"), torch.as_tensor(g.reshape(-1), device="cuda")
  both = torch.stack(ctx.matvec2_device("tt", fd, gd, eta)).cpu().numpy()
  pre = "single_wall" if wall else "no_wall"
  fn = getattr(oracle, pre + "_mobility_trans_times_force_oracle")
  tol = TOL_D1 if N == 1000 else TOL_D2
  assert rel_err(both[0], fn(r, f, eta, a, periodic_length=np.array(L))) < tol
  assert rel_err(both[1], fn(r, g, eta, a, periodic_length=np.array(L))) < tol
  # pair shards of the two-vector product sum to it
  parts = sum(torch.stack(ctx.matvec2_device("tt", fd, gd, eta, shard=s, nshards=3)) for s in range(3)).cpu().numpy() \
      if N >= 128 else both
  assert rel_err(parts, both) < 1e-13
  # a later single-vector product still sees clean accumulators
  assert rel_err(ctx.matvec_device("tt", gd, eta).cpu().numpy(), both[1]) < 1e-13
  ctx.close()
