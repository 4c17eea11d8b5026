"""Single-process multi-device engine (rmb_multi_*, multi.MultiContext) on ONE GPU: the same device listed G = 2, 3, 8
times runs the G-device code path -- G shard contexts on G streams, pair shard g of G each, the fixed-order slice
reduction through the reads that are peer reads on a node -- so every product of the surface can be held against the
one-context result (<= 1e-13, SURVEY 8d "G-GPU vs 1-GPU") and against the CPU oracle (<= 1e-12).  The reference has no
counterpart (single device); its call shape (one process, module-level functions: multi_bodies/multi_bodies.py:233-287,
:445) is why the engine sits behind `mobility.set_devices`."""
import os

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL_VS_SINGLE = 1e-13
TOL_VS_ORACLE = 1e-12
ORACLE_STEM = {"tt": "trans_times_force", "tr": "trans_times_torque", "rt": "rot_times_force", "rr": "rot_times_torque"}


def _cloud(n, seed, a=0.4):
  """5 % volume-fraction cloud above the wall (D2), a few blobs pushed below z = a so the B-damping is exercised."""
  rng = np.random.RandomState(seed)
  box = (n * (4.0 / 3.0) * np.pi * a ** 3 / 0.05) ** (1.0 / 3.0)
  r = rng.rand(n, 3) * box
  r[:, 2] += 1.1 * a
  r[:: 17, 2] = 0.6 * a
  return r, rng.randn(n, 3), rng.randn(n, 3), 1.3, a, box


@pytest.fixture(scope="module")
def torch_mod():
  import torch
  return torch


def _engine(G, **options):
  from rigidmultiblobswall_amd.multi import MultiContext
  m = MultiContext([0] * G)
  for k, v in options.items():
    m.set_option(k, v)
  return m


@pytest.mark.parametrize("G", [2, 3, 8])
@pytest.mark.parametrize("n", [90, 1000, 4100])          # < 128 blobs; partial last tile; several tile rows
def test_host_products_equal_single_context_and_oracle(oracle, G, n):
  from rigidmultiblobswall_amd import MobilityContext
  r, f, t, eta, a, _ = _cloud(n, 10 + n)
  single = MobilityContext(0)
  multi = _engine(G)
  try:
    for wall in (True, False):
      single.set_positions(r, a, None, wall)
      multi.set_positions(r, a, None, wall)
      assert multi.n == n and multi.n_shards == G
      for kind in ("tt", "tr", "rt", "rr"):
        u1 = single.matvec(kind, f, eta)
        uG = multi.matvec(kind, f, eta)
        assert rel_err(uG, u1) < TOL_VS_SINGLE, (kind, wall, rel_err(uG, u1))
        ref = getattr(oracle, ("single_wall" if wall else "no_wall") + "_mobility_" + ORACLE_STEM[kind] + "_oracle")(r, f, eta, a)
        assert rel_err(uG, ref) < TOL_VS_ORACLE, (kind, wall, rel_err(uG, ref))
      # fused force + torque, in-plane rows / columns
      u1 = single.matvec("tt_tr", f, eta, vec2=t)
      uG = multi.matvec("tt_tr", f, eta, vec2=t)
      assert rel_err(uG, u1) < TOL_VS_SINGLE
      if wall:
        for kind in ("tt", "tr"):
          assert rel_err(multi.matvec(kind, f, eta, in_plane=True), single.matvec(kind, f, eta, in_plane=True)) < TOL_VS_SINGLE
        uo = oracle.single_wall_mobility_trans_times_force_torque_oracle(r, f, t, eta, a)
        assert rel_err(uG, uo) < TOL_VS_ORACLE
      else:
        u1 = single.matvec("tt_free", f, eta)
        assert rel_err(multi.matvec("tt_free", f, eta), u1) < TOL_VS_SINGLE
        # the engine used to drop in_plane for the free-surface block (the single context honours it)
        u1 = single.matvec("tt_free", f, eta, in_plane=True)
        uG = multi.matvec("tt_free", f, eta, in_plane=True)
        assert rel_err(uG, u1) < TOL_VS_SINGLE and np.all(uG.reshape(-1, 3)[:, 2] == 0.0)
        assert rel_err(u1, single.matvec("tt_free", f, eta)) > 1e-3          # the mask does something
  finally:
    single.close()
    multi.close()


@pytest.mark.parametrize("G", [2, 3, 8])
def test_pseudo_periodic_products_equal_single_context(oracle, G):
  from rigidmultiblobswall_amd import MobilityContext
  n = 700
  r, f, t, eta, a, box = _cloud(n, 5)
  L = np.array([box, box, 0.0])
  single, multi = MobilityContext(0), _engine(G)
  try:
    single.set_positions(r, a, L, True)
    multi.set_positions(r, a, L, True)
    for kind in ("tt", "rr"):
      u1, uG = single.matvec(kind, f, eta), multi.matvec(kind, f, eta)
      assert rel_err(uG, u1) < TOL_VS_SINGLE, (kind, rel_err(uG, u1))
    uo = oracle.single_wall_mobility_trans_times_force_oracle(r, f, eta, a, periodic_length=L)
    assert rel_err(multi.matvec("tt", f, eta), uo) < 1e-11
  finally:
    single.close()
    multi.close()


@pytest.mark.parametrize("G", [2, 3, 8])
def test_device_entry_points_equal_single_context(torch_mod, G):
  torch = torch_mod
  from rigidmultiblobswall_amd import MobilityContext
  n = 2500
  r, f, t, eta, a, _ = _cloud(n, 21)
  dev = torch.device("cuda:0")
  rd, fd, td = (torch.as_tensor(x.reshape(-1), device=dev) for x in (r, f, t))
  single, multi = MobilityContext(0), _engine(G)
  try:
    single.set_positions(rd, a, None, True)
    multi.set_positions(rd, a, None, True)
    for kind, v2 in (("tt", None), ("rr", None), ("tt_tr", td)):
      u1 = single.matvec_device(kind, fd, eta, vec2=v2)
      uG = multi.matvec_device(kind, fd, eta, vec2=v2)
      assert rel_err(uG.cpu().numpy(), u1.cpu().numpy()) < TOL_VS_SINGLE, kind
    # multi-block / multi-vector operations: one pass per shard for all blocks, one reduction for all outputs
    for op, vecs in (("grand", (fd, td)), ("force_column", (fd,)), ("tt_multi", (fd, td, fd + td)),
                     ("velocity_from_force_torque", (fd, td))):
      o1 = single.matvec_op_device(op, vecs, eta)
      oG = multi.matvec_op_device(op, vecs, eta)
      assert len(o1) == len(oG)
      for x, y in zip(oG, o1):
        assert rel_err(x.cpu().numpy(), y.cpu().numpy()) < TOL_VS_SINGLE, op
    a2, b2 = multi.matvec2_device("tt", fd, td, eta)
    assert rel_err(a2.cpu().numpy(), single.matvec_device("tt", fd, eta).cpu().numpy()) < TOL_VS_SINGLE
    assert rel_err(b2.cpu().numpy(), single.matvec_device("tt", td, eta).cpu().numpy()) < TOL_VS_SINGLE
    # output into a caller's tensor, inputs produced by torch kernels on a side stream just before the call
    side = torch.cuda.Stream(device=dev)
    out = torch.empty(3 * n, dtype=torch.float64, device=dev)
    with torch.cuda.stream(side):
      g = fd * 2.0 - td
      multi.matvec_device("tt", g, eta, out=out)
      h = out * 1.0
    side.synchronize()
    ref = single.matvec_device("tt", g, eta)
    torch.cuda.synchronize()
    assert rel_err(h.cpu().numpy(), ref.cpu().numpy()) < TOL_VS_SINGLE
    del side                                   # the engine must not touch the destroyed handle on the next call
    u = multi.matvec_device("tt", fd, eta)
    assert rel_err(u.cpu().numpy(), single.matvec_device("tt", fd, eta).cpu().numpy()) < TOL_VS_SINGLE
  finally:
    single.close()
    multi.close()


@pytest.mark.parametrize("G", [2, 3, 8])
def test_forces_equal_single_context_and_oracle(oracle, torch_mod, G):
  from rigidmultiblobswall_amd import MobilityContext
  n = 1500
  rng = np.random.RandomState(3)
  a = 0.3
  r = rng.rand(n, 3) * 6.0 * a * (n / 100.0) ** (1.0 / 3.0)
  kw = dict(repulsion_strength=0.7, debye_length=0.4 * a, blob_radius=a)
  single, multi = MobilityContext(0), _engine(G)
  try:
    for L in (None, np.array([0.0, 9.0, 0.0])):
      single.set_positions(r, a, L, False)
      multi.set_positions(r, a, L, False)
      f1 = single.blob_blob_force(kw["repulsion_strength"], kw["debye_length"], a)
      fG = multi.blob_blob_force(kw["repulsion_strength"], kw["debye_length"], a)
      assert fG.shape == (n, 3)
      assert rel_err(fG, f1) < TOL_VS_SINGLE
      fo = oracle.calc_blob_blob_forces_oracle(r, periodic_length=np.zeros(3) if L is None else L, **kw)
      assert rel_err(fG, fo) < TOL_VS_ORACLE
      fd = multi.blob_blob_force_device(kw["repulsion_strength"], kw["debye_length"], a)
      assert rel_err(fd.cpu().numpy(), f1) < TOL_VS_SINGLE
  finally:
    single.close()
    multi.close()


@pytest.mark.parametrize("G", [2, 3, 8])
def test_deterministic_mode_is_bit_reproducible_across_calls_and_engines(G):
  n = 1300
  r, f, t, eta, a, _ = _cloud(n, 33)
  results = []
  for _ in range(2):
    multi = _engine(G, deterministic=2)
    try:
      multi.set_positions(r, a, None, True)
      results.append([multi.matvec("tt", f, eta), multi.matvec("tt", f, eta), multi.matvec("rr", t, eta),
                      multi.matvec("tt_tr", f, eta, vec2=t)])
    finally:
      multi.close()
  assert np.array_equal(results[0][0], results[0][1])
  for x, y in zip(results[0], results[1]):
    assert np.array_equal(x, y)
  # and it is the right product
  from rigidmultiblobswall_amd import MobilityContext
  single = MobilityContext(0)
  try:
    single.set_positions(r, a, None, True)
    assert rel_err(results[0][0], single.matvec("tt", f, eta)) < TOL_VS_SINGLE
  finally:
    single.close()


def test_staged_path_without_peer_access(monkeypatch, torch_mod):
  """RMB_MULTI_NO_PEER=1: slices and inputs travel by hipMemcpyPeerAsync instead of peer-mapped loads / stores (what a
  node without peer access between two listed devices runs)."""
  torch = torch_mod
  from rigidmultiblobswall_amd import MobilityContext
  monkeypatch.setenv("RMB_MULTI_NO_PEER", "1")
  n = 900
  r, f, t, eta, a, _ = _cloud(n, 8)
  single, multi = MobilityContext(0), _engine(3)
  try:
    assert multi.get_option("peer") == 0
    single.set_positions(r, a, None, True)
    multi.set_positions(r, a, None, True)
    assert rel_err(multi.matvec("tt", f, eta), single.matvec("tt", f, eta)) < TOL_VS_SINGLE
    fd, td = (torch.as_tensor(x.reshape(-1), device="cuda:0") for x in (f, t))
    o1 = single.matvec_op_device("grand", (fd, td), eta)
    oG = multi.matvec_op_device("grand", (fd, td), eta)
    for x, y in zip(oG, o1):
      assert rel_err(x.cpu().numpy(), y.cpu().numpy()) < TOL_VS_SINGLE
  finally:
    single.close()
    multi.close()


@pytest.mark.parametrize("no_peer", ["0", "1"])
def test_remote_shard_paths_on_one_gpu(monkeypatch, torch_mod, no_peer):
  """RMB_MULTI_FORCE_REMOTE=1 makes every shard but the first behave as if it sat on another device than devices[0]:
  inputs are pulled with hipMemcpyPeerAsync, slices are reduced into the shard's own buffer and handed to devices[0]
  with a peer copy -- exactly the calls a node issues (caller-owned memory is never touched by peer-mapped loads)."""
  torch = torch_mod
  from rigidmultiblobswall_amd import MobilityContext
  monkeypatch.setenv("RMB_MULTI_FORCE_REMOTE", "1")
  monkeypatch.setenv("RMB_MULTI_NO_PEER", no_peer)
  n = 1700
  r, f, t, eta, a, _ = _cloud(n, 14)
  rd, fd, td = (torch.as_tensor(x.reshape(-1), device="cuda:0") for x in (r, f, t))
  single, multi = MobilityContext(0), _engine(4)
  try:
    single.set_positions(rd, a, None, True)
    multi.set_positions(rd, a, None, True)
    for kind, v2 in (("tt", None), ("tt_tr", td)):
      assert rel_err(multi.matvec_device(kind, fd, eta, vec2=v2).cpu().numpy(),
                     single.matvec_device(kind, fd, eta, vec2=v2).cpu().numpy()) < TOL_VS_SINGLE
    o1 = single.matvec_op_device("grand", (fd, td), eta)
    oG = multi.matvec_op_device("grand", (fd, td), eta)
    for x, y in zip(oG, o1):
      assert rel_err(x.cpu().numpy(), y.cpu().numpy()) < TOL_VS_SINGLE
    assert rel_err(multi.matvec("rr", t, eta), single.matvec("rr", t, eta)) < TOL_VS_SINGLE
    single.set_positions(rd, a, None, False)
    multi.set_positions(rd, a, None, False)
    F1 = single.blob_blob_force_device(0.6, 0.2 * a, a).cpu().numpy()
    assert rel_err(multi.blob_blob_force_device(0.6, 0.2 * a, a).cpu().numpy(), F1) < TOL_VS_SINGLE
  finally:
    single.close()
    multi.close()


def test_rccl_reduction_with_one_rank_and_its_argument_checks(torch_mod):
  """"reduce" = 1: librccl is resolved at run time and ncclCommInitAll / grouped ncclAllReduce run -- with the one
  device of this box that is a one-rank communicator; duplicates are refused (one rank per device)."""
  from rigidmultiblobswall_amd import MobilityContext, _lib
  from rigidmultiblobswall_amd.multi import MultiContext
  n = 600
  r, f, t, eta, a, _ = _cloud(n, 4)
  single, multi = MobilityContext(0), MultiContext([0])
  try:
    multi.set_option("reduce", 1)
    single.set_positions(r, a, None, True)
    multi.set_positions(r, a, None, True)
    assert rel_err(multi.matvec("tt", f, eta), single.matvec("tt", f, eta)) < TOL_VS_SINGLE
    fd = torch_mod.as_tensor(f.reshape(-1), device="cuda:0")
    assert rel_err(multi.matvec_device("rr", fd, eta).cpu().numpy(), single.matvec("rr", f, eta)) < TOL_VS_SINGLE
  finally:
    single.close()
    multi.close()
  dup = MultiContext([0, 0])
  try:
    with pytest.raises(_lib.RmbError):
      dup.set_option("reduce", 1)
  finally:
    dup.close()
  with pytest.raises(_lib.RmbError):
    MultiContext([0, 99])


def test_plugin_surface_uses_the_configured_devices(oracle, monkeypatch):
  """mobility.set_devices([...]): the reference-shaped functions run on the engine above `multi_min_blobs` and on
  devices()[0] below; same results either way, equal to the oracle."""
  from rigidmultiblobswall_amd import forces, mobility
  n = 1100
  r, f, t, eta, a, _ = _cloud(n, 12)
  kw = dict(periodic_length=np.zeros(3), repulsion_strength=0.5, debye_length=0.1, blob_radius=a)
  try:
    mobility.set_devices([0])
    u1 = mobility.single_wall_mobility_trans_times_force_hip(r, f, eta, a)
    w1 = mobility.single_wall_mobility_rot_times_torque_hip(r, t, eta, a)
    f1 = forces.calc_blob_blob_forces_hip(r, **kw)
    monkeypatch.setattr(mobility, "multi_min_blobs", 0)
    mobility.set_devices([0, 0, 0])
    assert mobility.active_devices(n) == [0, 0, 0]
    uG = mobility.single_wall_mobility_trans_times_force_hip(r, f, eta, a)
    uG2 = mobility.single_wall_mobility_trans_times_force_hip(r, f, eta, a)        # positions cached on the engine
    wG = mobility.single_wall_mobility_rot_times_torque_hip(r, t, eta, a, step=3)   # extra kwargs are ignored
    fG = forces.calc_blob_blob_forces_hip(r, **kw)
    assert type(mobility._context(n)).__name__ == "MultiContext"
    assert rel_err(uG, u1) < TOL_VS_SINGLE and rel_err(uG2, u1) < TOL_VS_SINGLE and rel_err(wG, w1) < TOL_VS_SINGLE
    assert rel_err(fG, f1) < TOL_VS_SINGLE
    assert rel_err(uG, oracle.single_wall_mobility_trans_times_force_oracle(r, f, eta, a)) < TOL_VS_ORACLE
    # below the threshold the one-device context serves the call
    monkeypatch.setattr(mobility, "multi_min_blobs", 10 * n)
    assert mobility.active_devices(n) == [0]
    assert rel_err(mobility.single_wall_mobility_trans_times_force_hip(r, f, eta, a), u1) < TOL_VS_SINGLE
    # environment form
    mobility.set_devices(None)
    monkeypatch.setenv("RMB_DEVICES", "0,0")
    assert mobility.devices() == [0, 0]
    monkeypatch.delenv("RMB_DEVICES")
    monkeypatch.setenv("RMB_DEVICE", "0")
    assert mobility.devices() == [0]
  finally:
    mobility.set_devices(None)
    forces.reset()


def test_device_selection_is_honoured_and_fails_loudly(monkeypatch):
  """RMB_DEVICE / set_device pick the device of the module-level context and of the library's default context; an index
  that is not there is an error, never a silent fall back to device 0."""
  import ctypes
  from rigidmultiblobswall_amd import _lib, mobility
  lib = _lib.load()
  h = ctypes.c_void_p()
  monkeypatch.setenv("RMB_DEVICE", "0")
  _lib.check(lib.rmb_ctx_create(-1, ctypes.byref(h)))
  lib.rmb_ctx_destroy(h)
  monkeypatch.setenv("RMB_DEVICE", "7")
  assert lib.rmb_ctx_create(-1, ctypes.byref(h)) == -1 and b"out of range" in lib.rmb_last_error()
  monkeypatch.setenv("RMB_DEVICE", "gpu0")
  assert lib.rmb_ctx_create(-1, ctypes.byref(h)) == -1 and b"RMB_DEVICE" in lib.rmb_last_error()
  monkeypatch.delenv("RMB_DEVICE")
  assert lib.rmb_default_ctx_set_device(5) == -1
  _lib.check(lib.rmb_default_ctx_set_device(0))
  r = np.random.RandomState(0).rand(40, 3) + 1.0
  try:
    mobility.set_device(0)
    u = mobility.no_wall_mobility_trans_times_force_hip(r, r, 1.0, 0.1)
    assert np.all(np.isfinite(u))
    with pytest.raises(_lib.RmbError):
      mobility.set_device(3)
  finally:
    mobility.set_devices(None)


def test_rigid_solver_runs_on_the_engine(torch_mod):
  """A device-resident caller written against MobilityContext (RigidSuspension: GMRES + block-diagonal preconditioner)
  takes the engine unchanged: same iteration count (to the one the atomics may move) and the same velocities."""
  from rigidmultiblobswall_amd import structures as st
  from rigidmultiblobswall_amd.rigid import RigidSuspension
  R = 1.0
  shell = st.icosahedron_shell(0.79 * R)
  a = st.min_blob_separation(shell) / 2
  nb = 40
  loc, quat, _ = st.roller_monolayer(nb, radius=R, seed=2)
  FT = np.zeros((nb, 6)); FT[:, 2] = -0.1; FT[:, 4] = 1.0
  multi = _engine(3)
  try:
    ref = RigidSuspension([shell] * nb, loc, quat, a, 1.0)
    U1, _, info1 = ref.solve_mobility_problem(force_torque=FT, tol=1e-9)
    ref.close()
    sus = RigidSuspension([shell] * nb, loc, quat, a, 1.0, ctx=multi)
    assert sus._native_blocks() is multi.helper_context        # the O(N) helper kernels run beside the engine's products
    UG, _, infoG = sus.solve_mobility_problem(force_torque=FT, tol=1e-9)
    assert abs(infoG["iterations"] - info1["iterations"]) <= 1
    assert rel_err(UG, U1) < 1e-8
    sus.native_helpers = False                                 # and the torch operations give the same
    UT, _, infoT = sus.solve_mobility_problem(force_torque=FT, tol=1e-9)
    assert abs(infoT["iterations"] - info1["iterations"]) <= 1 and rel_err(UT, U1) < 1e-8
  finally:
    multi.close()


def test_bench_probe_of_the_multi_device_surface_runs(tmp_path):
  """tools/multi_surface_probe.py is what bench.py's `multi_device_surface` extra runs in a child process when a one-rank
  run sees several devices; here with this box's GPU listed three times."""
  import json
  import subprocess
  import sys
  from conftest import ROOT
  res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "multi_surface_probe.py"), "0,0,0", "3000"],
                       capture_output=True, text=True, timeout=240)
  assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
  row = json.loads([l for l in res.stdout.split("\n") if l.startswith("{")][-1])["sizes"][0]
  assert row["n_blobs"] == 3000 and row["all_devices"]["devices"] == [0, 0, 0] and row["one_device"]["devices"] == [0]
  assert row["rel_diff_all_vs_one"] < 1e-13 and row["peer_access"] is True


def test_host_timing_of_the_synchronous_entry_point():
  """rmb_last_host_timing: host wall clock of the stages of the last rmb_matvec (bench.py's host_surface.breakdown_us)."""
  from rigidmultiblobswall_amd import MobilityContext
  r, f, t, eta, a, _ = _cloud(2000, 1)
  ctx = MobilityContext(0)
  try:
    ctx.set_positions(r, a, None, True)
    ctx.matvec("tt", f, eta)
    ht = ctx.last_host_timing()
    assert ht["c_call_us"] > 0 and ht["upload_us"] > 0 and ht["launch_us"] > 0 and ht["wait_and_download_us"] > 0
    assert abs(ht["upload_us"] + ht["launch_us"] + ht["wait_and_download_us"] - ht["c_call_us"]) < 1e-6 * ht["c_call_us"] + 1e-3
  finally:
    ctx.close()


def test_engine_restores_the_callers_device_and_never_moves_its_partials(torch_mod):
  """(i) every rmb_multi_* entry point leaves the calling thread's current HIP device as it found it (on a node
  the engine walks over devices[g]; here: still cuda:0 and torch's current stream untouched); (ii) the partial buffers
  peers read are sized for the largest product (4 outputs) when the positions are set and never reallocated by a
  product: a one-output product followed by GRAND (two outputs) and a four-vector pass, all asynchronous, equal the
  single context."""
  torch = torch_mod
  from rigidmultiblobswall_amd import MobilityContext
  n = 3000
  r, f, t, eta, a, _ = _cloud(n, 33)
  dev = torch.device("cuda:0")
  rd, fd, td = (torch.as_tensor(x.reshape(-1), device=dev) for x in (r, f, t))
  single, multi = MobilityContext(0), _engine(3)
  try:
    single.set_positions(rd, a, None, True)
    multi.set_positions(rd, a, None, True)
    assert torch.cuda.current_device() == 0
    outs = [multi.matvec_device("tt", fd, eta)]                          # n_out = 1 ...
    outs += list(multi.matvec_op_device("grand", (fd, td), eta))         # ... then 2, enqueued behind it
    outs += list(multi.matvec_op_device("rr_multi", (fd, td, fd - td, fd + td), eta))     # ... then 4
    outs += [multi.matvec_device("tt", td, eta)]
    assert torch.cuda.current_device() == 0
    torch.cuda.synchronize()
    refs = [single.matvec_device("tt", fd, eta)] + list(single.matvec_op_device("grand", (fd, td), eta)) + \
           list(single.matvec_op_device("rr_multi", (fd, td, fd - td, fd + td), eta)) + [single.matvec_device("tt", td, eta)]
    for k, (x, y) in enumerate(zip(outs, refs)):
      assert rel_err(x.cpu().numpy(), y.cpu().numpy()) < TOL_VS_SINGLE, k
    # a larger configuration on the same engine: the partials grow at set_positions (after a drain), not inside a product
    n2 = 5000
    r2, f2, t2, _, _, _ = _cloud(n2, 34)
    multi.set_positions(r2, a, None, True)
    single.set_positions(r2, a, None, True)
    assert rel_err(multi.matvec("tt", f2, eta), single.matvec("tt", f2, eta)) < TOL_VS_SINGLE
    o1, o2 = multi.matvec_op_device("grand", tuple(torch.as_tensor(x.reshape(-1), device=dev) for x in (f2, t2)), eta)
    s1, s2 = single.matvec_op_device("grand", tuple(torch.as_tensor(x.reshape(-1), device=dev) for x in (f2, t2)), eta)
    assert rel_err(o1.cpu().numpy(), s1.cpu().numpy()) < TOL_VS_SINGLE and rel_err(o2.cpu().numpy(), s2.cpu().numpy()) < TOL_VS_SINGLE
  finally:
    single.close()
    multi.close()
