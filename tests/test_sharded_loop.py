"""The headless C++ rank (tests/cpp/sharded_loop.cpp): one shard of a larger set driven through
the C++ mirror of include/psim.hpp -- the attractor exchange in all three pacings, the list step that
allocates nothing, the sharded way of writing a checkpoint, the occupancy and device-count
queries.  Its recorded steps are replayed through the CPU oracle; the particles must agree bit
for bit.  The program maps RCCL and runs as a process of its own (no torch in it)."""
import os
import subprocess

import numpy as np
import pytest

import psim_testlib as T
from psim_testlib import psim

CPP_DIR = os.path.join(T.ROOT, "tests", "cpp")
BINARY = os.path.join(CPP_DIR, "sharded_loop")


def test_cpp_rank_builds_with_a_plain_host_compiler():
    subprocess.check_call(["make", "-C", CPP_DIR, "sharded_loop"], stdout=subprocess.DEVNULL)
    assert os.access(BINARY, os.X_OK)


def read_log(path):
    """[(params, attractors)] from the program's log: 48 B, a u32 count, count x 32 B."""
    raw = np.fromfile(path, dtype=np.uint8)
    steps, at = [], 0
    while at < len(raw):
        params = T.params_from_bytes(raw[at:at + 48])
        n = int(raw[at + 48:at + 52].view(np.uint32)[0])
        at += 52
        attractors = raw[at:at + 32 * n].view(psim.ATTRACTOR_DTYPE).copy()
        at += 32 * n
        steps.append((params, attractors))
    return steps


@pytest.mark.gpu
@pytest.mark.parametrize("count,first,total", [(120_003, 0, 120_003),          # a whole set
                                               (90_001, 40_000, 200_000)])    # a shard in the middle
def test_sharded_loop_matches_oracle_replay(tmp_path, count, first, total):
    if not os.access(BINARY, os.X_OK):
        subprocess.check_call(["make", "-C", CPP_DIR, "sharded_loop"], stdout=subprocess.DEVNULL)
    steps = 45
    log, out = str(tmp_path / "log.bin"), str(tmp_path / "particles.bin")
    r = subprocess.run([BINARY, str(count), str(first), str(total), str(steps), log, out],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert f"steps={2 * steps} particles={count}" in r.stdout
    taken = read_log(log)
    assert len(taken) == 2 * steps
    assert [len(a) for _, a in taken] == [1] * steps + [3] * steps
    state = T.oracle_gen_uniform_aos(first, count)
    for params, attractors in taken:
        state = T.oracle_update_aos(state, params, 1, attractors)
    got = np.fromfile(out, dtype=psim.PARTICLE_DTYPE)
    T.assert_particles_equal(got, state, "sharded loop replay")
    assert not os.path.exists(out + ".ckp")
