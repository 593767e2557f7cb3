"""GPU tests of every access mode and launch shape of the update kernels at a small, ragged
size: the library picks them by state size (plain / cache-resident sc1 stores / streaming
nt sc1 stores through per-wave buffer descriptors; one pass or a capped grid-stride grid;
reserved LDS), so the overrides (read when a handle is created) force each combination here
and the results must still equal the CPU oracle bit for bit."""
import os

import numpy as np
import pytest

import psim_testlib as T
from psim_testlib import psim
from test_gpu_batch import param_stream

pytestmark = pytest.mark.gpu

Sim = psim.HipParticleSimulation
KNOBS = ("PSIM_NT", "PSIM_RESERVE_LDS", "PSIM_MAX_BLOCKS")


@pytest.fixture
def knobs():
    saved = {k: os.environ.get(k) for k in KNOBS}
    yield os.environ
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def attractors(k):
    rng = np.random.default_rng(40 + k)
    a = np.zeros(k, dtype=psim.ATTRACTOR_DTYPE)
    a["position"] = rng.uniform(-40, 40, size=(k, 3)).astype(np.float32)
    a["force"] = rng.uniform(1, 50, size=k).astype(np.float32)
    a["radius"] = rng.uniform(5, 40, size=k).astype(np.float32)
    a["active"] = 1
    return a


@pytest.mark.parametrize("blocks", ["0", "7"])
@pytest.mark.parametrize("reserve", ["0", "26624"])
@pytest.mark.parametrize("access", ["0", "1", "2"])
def test_update_under_every_access_mode_and_grid(knobs, access, reserve, blocks):
    knobs["PSIM_NT"], knobs["PSIM_RESERVE_LDS"], knobs["PSIM_MAX_BLOCKS"] = access, reserve, blocks
    n = 100_003                                    # 25 000 quads: 390 full waves, one partial, 3 left over
    rng = np.random.default_rng(int(access) * 7 + int(blocks))
    stream = param_stream(rng, 6)                  # colour modes 0, 1, 2 mixed
    with Sim.new(n, psim.GEN_UNIFORM) as sim:
        want = sim.download_aos()
        for p in stream:                           # one launch per step
            sim.update(p)
            want = T.oracle_update_aos(want, p, 1)
        T.assert_particles_equal(sim.download_aos(), want, "single steps")
        sim.update_batch(stream)                   # the same steps fused
        for p in stream:
            want = T.oracle_update_aos(want, p, 1)
        T.assert_particles_equal(sim.download_aos(), want, "fused steps")
        lst = attractors(3)                        # list in kernel arguments
        sim.update_multi(stream[1], lst)
        want = T.oracle_update_aos(want, stream[1], 1, attractors=lst)
        T.assert_particles_equal(sim.download_aos(), want, "attractor list")


@pytest.mark.parametrize("n", [1, 3, 4, 63 * 4, 64 * 4, 64 * 4 + 1, 65 * 4 + 3])
def test_streaming_stores_at_wave_boundaries(knobs, n):
    """The streaming stores rely on the bounds check of a per-wave buffer descriptor for the
    last, partial wave: sizes around one wave's 256 particles."""
    knobs["PSIM_NT"] = "1"
    p = T.bench_params(mouse_radius=50.0, color_mode=1)
    with Sim.new(n + 64, psim.GEN_UNIFORM) as sim:
        sim.resize_buffer(n, psim.GEN_UNIFORM)     # capacity stays larger than the count
        start = sim.download_aos()
        view = sim.get_particle_buffer()
        assert view.count == n and view.capacity >= n + 64
        sim.update(p)
        T.assert_particles_equal(sim.download_aos(), T.oracle_update_aos(start, p, 1),
                                 f"streaming stores, n = {n}")


@pytest.mark.parametrize("n", [1_000_000, 4_200_000, 12_500_000, 16_000_000])
def test_achieved_residency_equals_the_intended_one(n):
    """The residency cap (untouched LDS reserved per single-wave workgroup) is worth a fifth
    of the update's bandwidth at streaming sizes.  psim_launch_shape reports what the library
    intends; here the runtime is asked what a CU really holds for the very kernel instantiation
    and LDS reservation a launch would use -- so a runtime update or a part with another LDS
    size cannot undo the cap unnoticed.  Sizes: fits the L2s / the Infinity Cache / the band
    just above it (one GPU's share of 100 M over 8) / streaming."""
    with Sim.new(n, psim.GEN_UNIFORM) as sim:
        for na in (0, 1, 2, 4, 8, 16):
            shape = psim.launch_shape(n, max(na, 1))
            for colour in (0, 1, 2):
                got, lds = sim.launch_occupancy(na, colour)
                reserve = shape["reserve_lds_colour"] if colour else shape["reserve_lds"]
                what = (f"{n} particles, {na} attractors, colour mode {colour}: reserve {reserve}, "
                        f"LDS per workgroup {lds}, resident {got}")
                if reserve:
                    # the reservation is what a workgroup takes, unless the kernel's own
                    # arrays are larger (colour tile + list + test records + pair queue = 11520 B)
                    assert lds == max(reserve, 11520 if colour and na >= 2 else 0), what
                    assert got == 163840 // lds, what
                    if colour == 0:
                        assert got == shape["resident_per_cu"], what
                else:
                    assert shape["resident_per_cu"] == 0 and got >= 8, what
