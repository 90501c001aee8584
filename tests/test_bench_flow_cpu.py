"""bench.py's multi-rank control flow at world 2 on the CPU (VZ_BENCH_DRY=1: no GPU call): rendezvous over gloo at 127.0.0.1, the
timed region with its barriers and the max over ranks, the tensor-parallel CHILD processes on their own port (without the
torchrun agent's store), and the JSON line - the north-star partition's numbers as `value`, the replicas beside them."""
import json
import os
import subprocess
import sys

from util import REPO


def _run(extra_env, port):
    env = dict(os.environ, VZ_BENCH_DRY="1", MASTER_ADDR="127.0.0.1", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--new-tokens", "4"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE JSON line, from rank 0 of the parent group only
    return json.loads(lines[0])


def test_two_rank_flow_reports_the_tensor_parallel_partition():
    line = _run({}, 29611)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["higher_is_better"] is True
    assert line["scaling"] == "strong" and line["config"]["parallelism"].startswith("tp2 + tile-dp2")
    assert set(line["config"]) <= {"workload", "parallelism", "tune"}                  # the workload, no model keys
    rep = line["replicas"]
    assert rep["scaling"] == "weak" and rep["parallelism"].startswith("dp2 replicas") and rep["value"] > 0 and line["value"] > 0
    # replicas count both ranks' tokens, the tensor-parallel request counts one: 2 x in the dry run
    assert 1.5 < rep["value"] / line["value"] < 2.6


def test_failed_tensor_parallel_child_falls_back_to_replicas():
    line = _run({"VZ_BENCH_TP_TIMEOUT": "0"}, 29641)
    assert line["scaling"] == "weak" and "replicas" not in line and line["tensor_parallel"]["value"] is None
    assert "did not produce a number" in line["config"]["parallelism"] and line["value"] > 0
