"""GPU cross-check of the multi-attractor kernel fed by a SECOND, independent exchange: the
runner of tests/torch_exchange.py (test scaffolding: all-gather through torch.distributed's
RCCL backend on its own stream, double-buffered).  The product's exchange is psim_exchange_*
(tests/test_gpu_exchange.py); both must leave the particles the oracle leaves.  One rank here
(a single-GPU box); the world_size > 1 host logic is covered on CPU over gloo."""
import os
import socket
import subprocess
import sys

import pytest

import psim_testlib as T

pytestmark = pytest.mark.gpu
WORKER = os.path.join(T.ROOT, "tests", "_multi_gpu_worker.py")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("lookahead", [1, 5])
def test_multi_attractor_runner_single_rank(tmp_path, lookahead):
    """lookahead 1: one all-gather per step; 5: one per 5 steps (12 steps = 2 full cycles
    and a partial one).  Both must equal the oracle with all attractors applied in order."""
    out = tmp_path / "result.txt"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-u", WORKER, "300007", "12", str(out), str(lookahead)],
                       env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert out.read_text().startswith("OK world=1")
