"""The headless C++ caller (tests/cpp/frame_loop.cpp): a frame loop shaped like the
reference application's (src/app.rs:202-316) driving the trait-shaped C++ mirror over
the C ABI.  On the GPU its recorded SimParams stream is replayed through the CPU oracle
and the final particles must agree bit for bit."""
import os
import subprocess

import numpy as np
import pytest

import psim_testlib as T
from psim_testlib import psim

CPP_DIR = os.path.join(T.ROOT, "tests", "cpp")
BINARY = os.path.join(CPP_DIR, "frame_loop")


def build():
    subprocess.check_call(["make", "-C", CPP_DIR], stdout=subprocess.DEVNULL)


def test_cpp_caller_builds_with_a_plain_host_compiler():
    build()
    assert os.access(BINARY, os.X_OK)


@pytest.mark.skipif(T.gpu_available(), reason="a GPU is present")
def test_cpp_caller_fails_loudly_without_a_gpu(tmp_path):
    build()
    r = subprocess.run([BINARY, "100", "0", "4", str(tmp_path / "p"), str(tmp_path / "q")],
                       capture_output=True, text=True)
    assert r.returncode == 1
    assert "no CPU path" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("count,generation", [(100_000, psim.GEN_HOLLOW),
                                              (250_001, psim.GEN_UNIFORM)])
def test_frame_loop_matches_oracle_replay(tmp_path, count, generation):
    if not os.access(BINARY, os.X_OK):
        build()
    frames = 72
    pfile, qfile = str(tmp_path / "params.bin"), str(tmp_path / "particles.bin")
    r = subprocess.run([BINARY, str(count), str(generation), str(frames), pfile, qfile],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(pfile, dtype=np.uint8).reshape(-1, 48)
    assert len(raw) == frames - 4                      # frames 44..47 were paused
    got = np.fromfile(qfile, dtype=psim.PARTICLE_DTYPE)
    assert len(got) == count
    state = (T.oracle_gen_hollow_aos(count) if generation == psim.GEN_HOLLOW
             else T.oracle_gen_uniform_aos(0, count))
    for row in raw:
        state = T.oracle_update_aos(state, T.params_from_bytes(row), 1)
    T.assert_particles_equal(got, state, "frame loop replay")
    # the script visited colour modes 1 and 2 and went back to 0: colours keep the last
    # computed value (CPU semantics, cpu.rs:106), they are not reset to initial_color
    assert got["color"].tobytes() != got["initial_color"].tobytes()
