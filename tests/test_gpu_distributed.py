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


@pytest.mark.bench_harness
def test_bench_spawns_its_own_ranks():
  """`python bench.py --gpus 2` launched plainly must start 2 ranks itself and print n_gpus = 2 (gloo rehearsal on the
  one GPU of this box; on a multi-GPU node the same command uses RCCL)."""
  import json
  env = dict(os.environ, RMB_BENCH_BACKEND="gloo")
  env.pop("WORLD_SIZE", None); env.pop("RANK", None)
  res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--prewarm-ms", "20", "--no-sweep", "--no-cpu"], env=env, capture_output=True, text=True, timeout=300)
  say = "rc %s\n--- stdout tail ---\n%s\n--- stderr tail ---\n%s" % (res.returncode, res.stdout[-2000:], res.stderr[-3000:])
  assert res.returncode == 0, say
  line = json.loads([l for l in res.stdout.split("\n") if l.startswith("{")][-1])
  assert line["n_gpus"] == 2 and line["world_size"] == 2 and line["steps"] == 5 and line["collective_backend"] == "gloo", say
  assert line["value"] > 0 and line["roofline"]["kernel_ms_avg"] > 0, say
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
  t0 = time.monotonic()
  res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
  rows = [l for l in res.stdout.split("\n") if l.startswith("{")]
  return res, (json.loads(rows[-1]) if rows else None), time.monotonic() - t0


def _say(res, line, took):
  """Everything a failed harness assertion needs to be diagnosed from the record alone."""
  return "took %.1f s, rc %s\nline: %s\n--- stdout tail ---\n%s\n--- stderr tail ---\n%s" % (
      took, res.returncode, line, res.stdout[-1500:], res.stderr[-3000:])


@pytest.mark.bench_harness
def test_bench_rank_failure_in_an_extra_ends_the_run_within_seconds_and_keeps_the_headline():
  """One rank raises inside a multi-rank extra (whose other ranks are inside collectives): the run must end within
  seconds, with the headline on the line and the failure recorded -- never a hang."""
  res, line, took = _bench(["--gpus", "2", "--steps", "5", "--warmup", "2", "--prewarm-ms", "20", "--no-cpu", "--no-host-surface"],
                           {"RMB_BENCH_INJECT": "fail:1:decompositions"}, 400)
  say = _say(res, line, took)
  assert line is not None, say
  assert line.get("value", 0) > 0 and line.get("n_gpus") == 2, say
  assert "rank 1 failed in stage 'decompositions'" in line.get("extras_aborted", {}).get("reason", ""), say
  assert "[bench headline]" in res.stderr, say            # the headline was out before any extra started


@pytest.mark.bench_harness
def test_bench_hang_in_an_extra_is_cut_at_its_stage_limit():
  """One rank never comes back from a stage: when the stage's own time limit is up -- counted from the stage's begin(),
  so independent of how long interpreter start, rendezvous and the headline took on this box (the round-4 driver run
  lost the headline to a 45 s whole-run budget) -- rank 0 prints the line with what exists and all ranks leave."""
  res, line, took = _bench(["--gpus", "2", "--steps", "5", "--warmup", "2", "--prewarm-ms", "20", "--no-cpu", "--no-host-surface",
                            "--stage-limit-s", "8"], {"RMB_BENCH_INJECT": "hang:1:decompositions"}, 400)
  say = _say(res, line, took)
  assert line is not None, say
  assert line.get("value", 0) > 0, say
  ab = line.get("extras_aborted", {})
  assert ab.get("reason", "").startswith("stage exceeded its time limit") and ab.get("stage") == "decompositions", say


@pytest.mark.bench_harness
def test_bench_hang_before_the_headline_is_cut_at_the_wall_clock_budget():
  """The whole-run budget still ends a run that hangs BEFORE the headline exists: an error row without `value`, rc != 0."""
  res, line, took = _bench(["--gpus", "2", "--steps", "5", "--warmup", "2", "--prewarm-ms", "20", "--no-cpu", "--no-host-surface",
                            "--no-sweep", "--budget-s", "3"], {"RMB_BENCH_INJECT": "hang:0:headline"}, 400)
  say = _say(res, line, took)
  assert res.returncode != 0, say
  assert line is not None and "value" not in line and line["extras_aborted"]["reason"] == "wall-clock budget exhausted", say


@pytest.mark.bench_harness
def test_bench_single_rank_extra_failure_is_recorded_and_the_run_goes_on():
  res, line, took = _bench(["--steps", "5", "--warmup", "2", "--prewarm-ms", "20", "--no-cpu", "--no-sweep"],
                           {"RMB_BENCH_INJECT": "fail:0:host_surface"}, 400)
  say = _say(res, line, took)
  assert res.returncode == 0 and line is not None, say
  assert line.get("host_surface") == {"error": "RuntimeError: injected failure"} and "extras_aborted" not in line, say
  assert line["build"]["mode"] in ("compiled", "reused"), say
