"""CPU tests of the multi-GPU host logic: shard ranges, attractor paths and the per-step
all-gather, run as a world_size-2 (and 3) gloo job."""
import importlib
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import psim_testlib as T

par = importlib.import_module("particle-simulation-3d_amd.parallel")
WORKER = os.path.join(T.ROOT, "tests", "_gloo_worker.py")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("total,world", [(10, 1), (10, 3), (100_000_000, 8), (1_000_000_007, 8),
                                         (7, 8)])
def test_shard_ranges_tile_the_array(total, world):
    ranges = [par.shard_range(total, r, world) for r in range(world)]
    assert ranges[0][0] == 0
    assert sum(c for _, c in ranges) == total
    for (f0, c0), (f1, _) in zip(ranges[:-1], ranges[1:]):
        assert f0 + c0 == f1
    counts = [c for _, c in ranges]
    assert max(counts) - min(counts) <= 1
    with pytest.raises(ValueError):
        par.shard_range(total, world, world)


def test_attractor_paths_are_reproducible_and_bounded():
    a = par.attractors_at(123, 8)
    b = par.attractors_at(123, 8)
    assert a.tobytes() == b.tobytes() and a.dtype.itemsize == 32
    assert (np.abs(a["position"]) <= 40.0).all() and (a["active"] == 1).all()
    assert par.attractors_at(124, 8).tobytes() != a.tobytes()           # they move
    assert par.attractor_at(5, 2).tobytes() == par.attractors_at(5, 8)[2:3].tobytes()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_multi_attractor_job_over_gloo(tmp_path, world):
    out = tmp_path / "result.txt"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], WORKER, "20011", "6", str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert out.read_text().startswith(f"OK world={world}")
