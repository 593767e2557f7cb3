"""GPU test of psim_update_multi_host_noalloc, the entry point that allocates nothing: its
attractor list goes through a static device array of the library instead of page-locked
staging and the handle's device scratch.  Checked against psim_update_multi_host -- the
existing entry point it is a variant of -- bit for bit, on two calls in a row of which the
second has the smaller input (fewer particles, a shorter list), and against the CPU oracle.
Collected last (the file name sorts behind the other tests)."""
import numpy as np
import pytest

import psim_testlib as T
from psim_testlib import psim

pytestmark = pytest.mark.gpu

Sim = psim.HipParticleSimulation


def attractor_list(n, seed, radius):
    rng = np.random.default_rng(seed)
    a = np.zeros(n, dtype=psim.ATTRACTOR_DTYPE)
    a["position"] = rng.uniform(-45, 45, (n, 3)).astype(np.float32)
    a["force"] = rng.uniform(1, 60, n).astype(np.float32)
    a["radius"] = np.float32(radius)
    a["active"] = 1
    a["active"][n // 3] = 0                              # one switched off in the middle
    return a


def run_pair(n, attractors, params, steps):
    """The same state stepped through the existing entry point and the new one."""
    with Sim.new(n, psim.GEN_UNIFORM) as old, Sim.new(n, psim.GEN_UNIFORM) as new:
        start = old.download_aos()
        for _ in range(steps):
            old.update_multi(params, attractors)
        want = old.download_aos()                        # finished before the other handle runs
        for _ in range(steps):
            new.update_multi_noalloc(params, attractors)
        got = new.download_aos()
    return start, want, got


def test_two_calls_in_a_row_the_second_smaller():
    params = T.bench_params(gravity=0.7, color_mode=1)
    long_list = attractor_list(40, 1, 12.0)              # 5 chunks of the kernel arguments
    short_list = attractor_list(11, 2, 30.0)             # records 11..39 of the array are stale
    for n, attractors in ((300_003, long_list), (100_001, short_list)):
        start, want, got = run_pair(n, attractors, params, 3)
        assert got.tobytes() == want.tobytes(), f"{n} particles, {len(attractors)} attractors"
        T.assert_particles_equal(got, T.oracle_update_aos(start, params, 3, attractors),
                                 f"oracle, {len(attractors)} attractors")


def test_list_sizes_and_limits():
    params = T.bench_params()
    for k in (0, 1, 8, 9, 64):
        attractors = attractor_list(max(k, 1), 10 + k, 15.0)[:k]
        start, want, got = run_pair(20_002, attractors, params, 2)
        assert got.tobytes() == want.tobytes(), f"{k} attractors"
    with Sim.new(1000, psim.GEN_UNIFORM) as sim:
        before = sim.download_aos()
        with pytest.raises(psim.PsimError, match="static device list") as e:
            sim.update_multi_noalloc(params, np.zeros(65, dtype=psim.ATTRACTOR_DTYPE))
        assert e.value.code == 1
        assert sim.download_aos().tobytes() == before.tobytes()


def test_capture_into_a_graph_replays_the_list():
    """What the entry point is for: with more than 8 attractors the existing entry point
    stages through page-locked memory and cannot be captured; this one only enqueues kernels
    whose arguments travel by value."""
    torch = pytest.importorskip("torch")
    params = T.bench_params(gravity=0.3)
    attractors = attractor_list(20, 5, 14.0)
    n = 50_000
    stream = torch.cuda.Stream()
    with Sim.new(n, psim.GEN_UNIFORM) as sim:
        start = sim.download_aos()
        sim.set_stream(stream.cuda_stream)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(stream):
            graph.capture_begin()
            sim.update_multi_noalloc(params, attractors)         # recorded, not run
            sim.update_multi_noalloc(params, attractors[:12])
            graph.capture_end()
        torch.cuda.synchronize()
        assert sim.download_aos().tobytes() == start.tobytes(), "capture must not step the state"
        graph.replay()
        graph.replay()
        torch.cuda.synchronize()
        got = sim.download_aos()
        sim.set_stream(0)
    want = start
    for _ in range(2):
        want = T.oracle_update_aos(want, params, 1, attractors)
        want = T.oracle_update_aos(want, params, 1, attractors[:12])
    T.assert_particles_equal(got, want, "graph replay")


def test_capture_under_the_compute_shaders_colour_rule():
    """Under PSIM_COLOR_RESET a step in colour mode 0 that follows a colouring step also
    enqueues the device-to-device copy that resets the colours, and whether it does is decided
    when the call is made -- at capture.  A graph captured in the colour state it is replayed in
    (here: a colouring step, then a mode-0 step) must replay exactly what the same calls do
    eagerly, and leave colour == initial colour."""
    torch = pytest.importorskip("torch")
    colouring = T.bench_params(gravity=0.3, color_mode=1)
    plain = T.bench_params(gravity=0.3, color_mode=0)
    attractors = attractor_list(12, 9, 20.0)
    n = 40_000
    stream = torch.cuda.Stream()
    with Sim.new(n, psim.GEN_UNIFORM) as sim, Sim.new(n, psim.GEN_UNIFORM) as eager:
        sim.set_color_semantics(psim.COLOR_RESET)
        eager.set_color_semantics(psim.COLOR_RESET)
        sim.set_stream(stream.cuda_stream)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(stream):
            graph.capture_begin()
            sim.update_multi_noalloc(colouring, attractors)
            sim.update_multi_noalloc(plain, attractors)          # carries the colour reset
            graph.capture_end()
        graph.replay()
        graph.replay()
        torch.cuda.synchronize()
        got = sim.download_aos()
        sim.set_stream(0)
        for _ in range(2):
            eager.update_multi_noalloc(colouring, attractors)
            eager.update_multi_noalloc(plain, attractors)
        want = eager.download_aos()
    assert got.tobytes() == want.tobytes()
    assert got["color"].tobytes() == got["initial_color"].tobytes()
    assert (got["velocity"] != 0).any()
