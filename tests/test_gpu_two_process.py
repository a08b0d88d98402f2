"""The N > 1 path end to end with REAL separate processes: bench.py under torch.distributed.run with two ranks. On the
one-GPU test box both ranks share cuda:0 and the all-reduce is staged through the host (--backend gloo; RCCL refuses two
ranks on one device) - everything else (sharding by keyframe window, per-rank interior elimination, separator exchange,
replicated separator solve, update exchange, lock-step lambda search) is the production path.  The result must agree with
one process solving the same doubled trajectory."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    return json.loads(lines[0])


def test_two_ranks_agree_with_one_process_on_the_same_graph():
    common = ["--steps", "6", "--warmup", "1", "--config", "1", "--no-cpu-baseline", "--no-frontend"]
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29517", "bench.py", "--gpus", "2", "--backend", "gloo"] + common)
    one = _run([sys.executable, "bench.py", "--gpus", "1", "--scale", "2"] + common)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1 and two["scaling"] == "weak"
    assert two["config"]["factors"] == one["config"]["factors"]
    assert two["steps"] == one["steps"] and two["config"]["inner_iterations"] == one["config"]["inner_iterations"]
    assert abs(two["config"]["error_before"] - one["config"]["error_before"]) <= 1e-9 * one["config"]["error_before"]
    assert abs(two["config"]["error_after"] - one["config"]["error_after"]) <= 1e-6 * one["config"]["error_after"]
    assert any(k["name"] == "allreduce" and k["launches"] > 0 for k in two["kernels"])
