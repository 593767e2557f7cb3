"""GPU tests of psim_update_batch: several consecutive steps in one pass over the state
must equal, bit for bit, the same steps taken one launch at a time and the CPU oracle."""
import numpy as np
import pytest

import psim_testlib as T
from psim_testlib import psim

pytestmark = pytest.mark.gpu

Sim = psim.HipParticleSimulation


def param_stream(rng, k, colour=True):
    """A frame-by-frame parameter stream like the application's: variable delta_time,
    the attractor moving and being released, sliders and colour mode changing."""
    out = []
    att = np.array([0.0, 0.0, 48.0], dtype=np.float32)
    for i in range(k):
        att = att + rng.uniform(-1.5, 1.5, size=3).astype(np.float32)
        out.append(T.make_params(
            delta_time=float(rng.choice([0.016, 0.0175, 0.01525, 0.033])),
            gravity=float(rng.choice([0.0, 0.0, 2.5])),
            color_mode=int(rng.choice([0, 0, 1, 2])) if colour else 0,
            mouse_force=float(rng.uniform(1, 100)),
            mouse_radius=float(rng.uniform(5, 50)),
            is_mouse_dragging=int(rng.random() < 0.7),
            max_dist_for_color=float(rng.uniform(20, 80)),
            mouse_position=tuple(float(v) for v in att)))
    return out


@pytest.mark.parametrize("k", [1, 2, 3, 5, 63, 64, 65, 200])
@pytest.mark.parametrize("n", [7, 100_003])
def test_batch_equals_single_steps_and_oracle(n, k):
    rng = np.random.default_rng(1000 * k + n % 97)
    stream = param_stream(rng, k)
    with Sim.new(n, psim.GEN_UNIFORM) as fused, Sim.new(n, psim.GEN_UNIFORM) as single:
        start = fused.download_aos()
        fused.update_batch(stream)
        for p in stream:
            single.update(p)
        a, b = fused.download_aos(), single.download_aos()
    assert a.tobytes() == b.tobytes()
    want = start
    for p in stream:
        want = T.oracle_update_aos(want, p, 1)
    T.assert_particles_equal(a, want, f"batch of {k}")


def test_batch_colour_rules():
    """Colour after a batch = the colour computed at the LAST step whose mode is 1 or 2;
    mode-0 steps after it keep it, a batch without such a step leaves colours alone."""
    n = 5000
    rng = np.random.default_rng(5)
    base = param_stream(rng, 10, colour=False)
    with Sim.new(n, psim.GEN_UNIFORM) as sim:
        before = sim.download_aos()
        sim.update_batch(base)
        after = sim.download_aos()
        assert after["color"].tobytes() == before["color"].tobytes()
        mixed = param_stream(rng, 10, colour=False)
        mixed[2].color_mode = 2
        mixed[6].color_mode = 1
        state = after
        sim.update_batch(mixed)
        got = sim.download_aos()
    for p in mixed:
        state = T.oracle_update_aos(state, p, 1)
    T.assert_particles_equal(got, state, "mixed colour modes")
    assert got["color"].tobytes() != after["color"].tobytes()


def test_batch_on_the_streaming_size():
    """20 M particles (past the Infinity Cache: non-temporal path), 96 steps in two passes."""
    n, k = 20_000_000, 96
    stream = param_stream(np.random.default_rng(9), k, colour=False)
    with Sim.new(n, psim.GEN_UNIFORM) as sim:
        sim.update_batch(stream)
        got = sim.download_soa(n - 50_001, 50_001)
        gpu_sum = sim.checksum()
    want = T.oracle_gen_uniform_soa(n - 50_001, 50_001)
    for p in stream:
        want, _ = T.oracle_update_soa(want, p, 1)
    T.assert_same_f32(got, want, "tail slice")
    with Sim.new(n, psim.GEN_UNIFORM) as ref:
        for p in stream:
            ref.update(p)
        assert ref.checksum() == gpu_sum


def test_batch_edge_arguments():
    with Sim.new(100, psim.GEN_UNIFORM) as sim:
        before = sim.download_aos()
        sim.update_batch([])                                   # zero steps: nothing happens
        assert sim.download_aos().tobytes() == before.tobytes()
        lib = psim.load_library()
        assert lib.psim_update_batch(sim.handle, None, 3) == 1
    with Sim.new(0, psim.GEN_UNIFORM) as empty:
        empty.update_batch(param_stream(np.random.default_rng(1), 4))
        empty.sync()
