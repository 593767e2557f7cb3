"""psim_update under HIP stream capture: the update entry point only enqueues kernels on the
handle's stream (parameters travel by value), so a caller can capture a run of steps into a
graph and replay it.  The replayed steps must equal the same steps launched one by one."""
import time

import pytest

import psim_testlib as T
from psim_testlib import psim

pytestmark = pytest.mark.gpu


def test_captured_updates_replay_bit_identically():
    import torch
    n, steps = 200_000, 32
    params = [T.bench_params(mouse_radius=30.0, delta_time=0.004 * (1 + k % 3),
                             color_mode=(1 if k == 20 else 0)) for k in range(steps)]
    with psim.HipParticleSimulation.new(n, psim.GEN_UNIFORM) as ref:
        start = ref.download_aos()
        for p in params:
            ref.update(p)
        want_once = ref.download_aos()
        for p in params:
            ref.update(p)
        want_twice = ref.download_aos()
    oracle = start
    for p in params:
        oracle = T.oracle_update_aos(oracle, p)
    T.assert_particles_equal(want_once, oracle, "separate launches")

    stream = torch.cuda.Stream()
    with psim.HipParticleSimulation.new(n, psim.GEN_UNIFORM) as sim:
        sim.set_stream(stream.cuda_stream)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(stream):
            graph.capture_begin()
            for p in params:
                sim.update(p)                    # recorded, not run
            graph.capture_end()
        torch.cuda.synchronize()
        assert sim.download_aos().tobytes() == start.tobytes(), "capture must not step the state"
        graph.replay()
        torch.cuda.synchronize()
        assert sim.download_aos().tobytes() == want_once.tobytes()
        graph.replay()
        torch.cuda.synchronize()
        assert sim.download_aos().tobytes() == want_twice.tobytes()
        # informational: replay cost per step against separate launches
        t0 = time.perf_counter()
        for _ in range(20):
            graph.replay()
        torch.cuda.synchronize()
        per_step = (time.perf_counter() - t0) / (20 * steps) * 1e6
        print(f"graph replay: {per_step:.2f} us per step at {n} particles")
        sim.set_stream(0)
