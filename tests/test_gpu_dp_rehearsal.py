"""GPU: the N>1 code path of the row-sparse step (graph A -> eager exchange -> graph B) with two
ranks on cuda:0 over gloo.  Replicas that start identical and see different shards must stay
bit-identical (deterministic ownership merge), with and without HIP graphs."""
import json
import math
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("kind", ["fused", "autograd"])
@pytest.mark.parametrize("mode", ["graph", "eager"])
def test_two_ranks_stay_bit_identical(mode, kind):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dp_rehearsal_worker.py"), mode, "4", kind]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, p.stdout[-2000:]
    ranks = json.loads(line[0][len("RESULT "):])
    assert len(ranks) == 2
    assert ranks[0]["flat"] == ranks[1]["flat"], "dense parameters diverged between replicas"
    assert ranks[0]["tables"] == ranks[1]["tables"], "embedding tables diverged between replicas"
    assert all(math.isfinite(r["loss"]) for r in ranks)
    assert ranks[0]["moved"] > 0.5        # the tables really were updated


def _single_rank_rccl(extra_env):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dp_rehearsal_worker.py"), "graph", "4", "fused", "nccl"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DFM_FORCE_DP_PATH="1", **extra_env)
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, p.stdout[-2000:]
    return json.loads(line[0][len("RESULT "):])[0]


def test_exchange_captured_in_the_graph_matches_the_split_path():
    """DFM_DP_GRAPH_COLLECTIVE=1 captures the RCCL all-gather inside the step's graph (one launch per
    step).  With a single-rank RCCL communicator on this box it must reproduce the split path
    (graph A -> eager exchange -> graph B) bit for bit."""
    split = _single_rank_rccl({})
    fused = _single_rank_rccl({"DFM_DP_GRAPH_COLLECTIVE": "1"})
    assert split["flat"] == fused["flat"] and split["tables"] == fused["tables"]
    assert split["loss"] == fused["loss"] and split["moved"] > 0.5
