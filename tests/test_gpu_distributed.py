"""Multi-rank path with the real HIP kernels: 2 and 3 ranks share cuda:0 over gloo and execute exactly the per-rank
launches of an N-GPU run (rmb_matvec_pairshard_device / rmb_matvec_op_pairshard_device + all-reduce; target range +
all-gather), checked against the single-context products.  RCCL itself needs one device per rank and is exercised by
`bench.py --gpus N` on a multi-GPU node; everything else of the N > 1 path runs here."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_one_gpu_match_single_context(world):
  procs = []
  port = _free_port()
  for rank in range(world):
    env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gpu_dist_worker.py")], env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
  outs = []
  try:
    for p in procs:
      outs.append(p.communicate(timeout=240)[0])
  finally:
    for p in procs:
      if p.poll() is None:
        p.kill()
  for rank, (p, o) in enumerate(zip(procs, outs)):
    assert p.returncode == 0, "rank %d failed:\n%s" % (rank, o[-3000:])
  assert "checks on %d ranks ok" % world in outs[0]


def test_one_rank_group_runs_every_collective_through_rccl():
  """The box has one GPU and RCCL wants one device per rank, so the N > 1 collectives cannot meet a second rank here --
  but they can RUN: a one-rank "nccl" group with always_exchange issues the fp64 all-reduce / all-gather / broadcast of
  every product through RCCL (communicator set-up, the HIP stream hand-off between the library's stream and RCCL's,
  fp64 SUM) and the results must still equal the single context."""
  env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
             RMB_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
  res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_gpu_dist_worker.py")], env=env, capture_output=True,
                       text=True, timeout=300)
  assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
  assert "checks on 1 ranks ok (nccl)" in res.stdout


def test_bench_spawns_its_own_ranks():
  """`python bench.py --gpus 2` launched plainly must start 2 ranks itself and print n_gpus = 2 (gloo rehearsal on the
  one GPU of this box; on a multi-GPU node the same command uses RCCL)."""
  import json
  env = dict(os.environ, RMB_BENCH_BACKEND="gloo")
  env.pop("WORLD_SIZE", None); env.pop("RANK", None)
  res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--prewarm-ms", "20", "--no-sweep", "--no-cpu"], env=env, capture_output=True, text=True, timeout=300)
  assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
  line = json.loads([l for l in res.stdout.split("\n") if l.startswith("{")][-1])
  assert line["n_gpus"] == 2 and line["world_size"] == 2 and line["steps"] == 5 and line["collective_backend"] == "gloo"
  assert line["value"] > 0 and line["roofline"]["kernel_ms_avg"] > 0
  # a rank that fails must fail the launcher
  res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--no-sweep", "--no-cpu"], env=dict(env, RMB_BENCH_BACKEND="nccl"), capture_output=True, text=True,
                       timeout=300)
  import torch
  if torch.cuda.device_count() < 2:
    assert res.returncode != 0        # 2 RCCL ranks on a 1-GPU box: refused before any collective
  else:
    assert res.returncode == 0


def _bench(args, env_extra, timeout):
  import json
  import time
  env = dict(os.environ, RMB_BENCH_BACKEND="gloo", **env_extra)
  env.pop("WORLD_SIZE", None); env.pop("RANK", None)
  t0 = time.time()
  res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
  rows = [l for l in res.stdout.split("\n") if l.startswith("{")]
  return res, (json.loads(rows[-1]) if rows else None), time.time() - t0


def test_bench_rank_failure_in_an_extra_ends_the_run_within_seconds_and_keeps_the_headline():
  """One rank raises inside a multi-rank extra (whose other ranks are inside collectives): the run must end within
  seconds, with the headline on the line and the failure recorded -- never a hang."""
  res, line, took = _bench(["--gpus", "2", "--steps", "5", "--warmup", "2", "--prewarm-ms", "20", "--no-cpu", "--no-host-surface"],
                           {"RMB_BENCH_INJECT": "fail:1:decompositions"}, 240)
  assert took < 120, took
  assert line is not None, res.stdout[-2000:] + res.stderr[-3000:]
  assert line["value"] > 0 and line["n_gpus"] == 2
  assert "extras_aborted" in line and "rank 1 failed in stage 'decompositions'" in line["extras_aborted"]["reason"]
  assert "[bench headline]" in res.stderr            # the headline was out before any extra started


def test_bench_hang_in_an_extra_is_cut_at_the_wall_clock_budget():
  """One rank never comes back from a stage: at the budget rank 0 prints the line with what exists and all ranks leave."""
  res, line, took = _bench(["--gpus", "2", "--steps", "5", "--warmup", "2", "--prewarm-ms", "20", "--no-cpu", "--no-host-surface",
                            "--budget-s", "45"], {"RMB_BENCH_INJECT": "hang:1:decompositions"}, 240)
  assert took < 100, took
  assert line is not None and line["value"] > 0
  assert line["extras_aborted"]["reason"] == "wall-clock budget exhausted" and line["extras_aborted"]["stage"] == "decompositions"


def test_bench_single_rank_extra_failure_is_recorded_and_the_run_goes_on():
  res, line, took = _bench(["--steps", "5", "--warmup", "2", "--prewarm-ms", "20", "--no-cpu", "--no-sweep"],
                           {"RMB_BENCH_INJECT": "fail:0:host_surface"}, 240)
  assert res.returncode == 0 and line is not None
  assert line["host_surface"] == {"error": "RuntimeError: injected failure"} and "extras_aborted" not in line
  assert line["build"]["mode"] in ("compiled", "reused")
