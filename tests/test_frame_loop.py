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
    # the statistics panel's numbers (app.rs:325-330): 72 frames of ~20 ms pass one second
    # once, and the update call of an asynchronous backend returns in well under a frame
    shown = dict(kv.split("=") for kv in r.stdout.splitlines()[-1].split())
    assert 40.0 < float(shown["fps"]) < 70.0 and 0.0 < float(shown["update_time_ms"]) < 16.0
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


STAGED = os.path.join(CPP_DIR, "staged_frames")


@pytest.mark.gpu
@pytest.mark.parametrize("count,frames", [(0, 3), (1, 5), (70_001, 25), (1_000_000, 12)])
def test_staged_frames_one_frame_ahead_deliver_the_blocking_loops_frames(count, frames):
    """tests/cpp/staged_frames.cpp makes the calls of rust/hip_backend.rs's `update`: the
    pipelined loop (psim_update + psim_download_aos_async of frame k, then
    psim_download_aos_wait for frame k-1) must hand the renderer exactly the frames the
    blocking loop hands it, one frame later (the program compares the hashes itself)."""
    if not os.access(STAGED, os.X_OK):
        build()
    r = subprocess.run([STAGED, str(count), str(frames), "check"], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    blocking_fps, pipelined_fps, blocking_wait, pipelined_wait = map(float, r.stdout.split())
    assert blocking_fps > 0 and pipelined_fps > 0
    print(f"{count} particles: {blocking_fps:.0f} -> {pipelined_fps:.0f} frames/s (with the "
          f"frame hashed), host waits {blocking_wait:.3f} -> {pipelined_wait:.3f} ms per frame")
