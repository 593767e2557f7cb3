"""GPU tests of the state-management entry points around the update: construction,
generators, resize / reset semantics, pause flag, AoS import / export, shards."""
import ctypes

import numpy as np
import pytest

import psim_testlib as T
from psim_testlib import psim

pytestmark = pytest.mark.gpu

Sim = psim.HipParticleSimulation


def random_particles(n, seed=0):
    rng = np.random.default_rng(seed)
    p = np.zeros(n, dtype=psim.PARTICLE_DTYPE)
    raw = rng.uniform(-100, 100, size=(n, 16)).astype(np.float32)
    p.view(np.float32).reshape(n, 16)[:] = raw
    return p


# ---- trait surface -----------------------------------------------------------------------

def test_new_count_method_pause():
    with Sim.new(12_345, psim.GEN_HOLLOW) as sim:
        assert sim.get_particle_count() == 12_345
        assert sim.get_method() == psim.METHOD_HIP
        assert sim.is_paused() is False                       # cpu.rs:34 / compute.rs:118
        sim.set_paused(True)
        assert sim.is_paused() is True
        # like the reference, update does not consult the flag (cpu.rs:46-48 is commented
        # out; the caller gates on is_paused, app.rs:283)
        before = sim.download_aos()
        sim.update(T.bench_params(gravity=1.0))
        after = sim.download_aos()
        assert before.tobytes() != after.tobytes()
        sim.set_paused(False)
        assert sim.is_paused() is False


def test_zero_and_one_particle():
    with Sim.new(0, psim.GEN_UNIFORM) as sim:
        assert sim.get_particle_count() == 0
        sim.update(T.bench_params())
        sim.sync()
        assert len(sim.download_aos()) == 0
        assert sim.checksum() == [0] * 6
    with Sim.new(1, psim.GEN_UNIFORM) as sim:
        start = sim.download_aos()
        sim.update(T.bench_params(mouse_radius=50.0))
        T.assert_particles_equal(sim.download_aos(),
                                 T.oracle_update_aos(start, T.bench_params(mouse_radius=50.0)),
                                 "single particle")


# ---- generators ----------------------------------------------------------------------------

@pytest.mark.parametrize("n", [1, 2, 1000, 100_000])
def test_hollow_generator_matches_oracle(n):
    with Sim.new(n, psim.GEN_HOLLOW) as sim:
        T.assert_particles_equal(sim.download_aos(), T.oracle_gen_hollow_aos(n), f"hollow {n}")


@pytest.mark.parametrize("n", [1, 1000, 100_000, 2_500_001])
def test_filled_generator_matches_oracle(n):
    # from 65 536 particles on, host threads each jump the random stream to their slice
    with Sim.new(n, psim.GEN_FILLED) as sim:
        T.assert_particles_equal(sim.download_aos(), T.oracle_gen_filled_aos(0, n), f"filled {n}")


def test_filled_shard_far_into_the_stream():
    """A shard starts 6e8 draws into the one sequential stream: the library jumps there
    (the state transition is linear over GF(2)), the oracle draws its way."""
    first = 200_000_000
    with Sim.new_shard(70_000, first, 250_000_000, psim.GEN_FILLED) as shard:
        T.assert_particles_equal(shard.download_aos(), T.oracle_gen_filled_aos(first, 70_000),
                                 "filled shard at 2e8")


def test_default_configuration_of_the_app():
    """The app's compute-backend start: 1 M particles, hollow sphere (app.rs:103,117),
    then frames with the default UI values and the attractor held (app.rs:143-148)."""
    n = 1_000_000
    params = T.bench_params()
    with Sim.new(n, psim.GEN_HOLLOW) as sim:
        start = sim.download_aos()
        for _ in range(20):
            sim.update(params)
        got = sim.download_aos()
    T.assert_particles_equal(got, T.oracle_update_aos(start, params, 20), "1M hollow x 20")


# ---- resize / reset (compute-backend semantics, compute.rs:146-216) ---------------------------

def test_resize_regenerates_everything():
    with Sim.new(5000, psim.GEN_HOLLOW) as sim:
        for _ in range(3):
            sim.update(T.bench_params(gravity=2.0))
        sim.resize_buffer(5000, psim.GEN_HOLLOW)              # equal count: no-op
        assert sim.download_aos().tobytes() != T.oracle_gen_hollow_aos(5000).tobytes()
        sim.resize_buffer(20_000, psim.GEN_HOLLOW)            # grow: fresh state
        assert sim.get_particle_count() == 20_000
        T.assert_particles_equal(sim.download_aos(), T.oracle_gen_hollow_aos(20_000), "grown")
        sim.update(T.bench_params(gravity=2.0))
        sim.resize_buffer(300, psim.GEN_FILLED)               # shrink: fresh state, new mode
        assert sim.get_particle_count() == 300
        T.assert_particles_equal(sim.download_aos(), T.oracle_gen_filled_aos(0, 300), "shrunk")
        view = sim.get_particle_buffer()
        assert view.count == 300 and view.capacity >= 20_000  # storage kept on shrink


def test_reset_regenerates_current_count():
    with Sim.new(7777, psim.GEN_UNIFORM) as sim:
        for _ in range(2):
            sim.update(T.bench_params())
        sim.reset(psim.GEN_HOLLOW)
        assert sim.get_particle_count() == 7777
        T.assert_particles_equal(sim.download_aos(), T.oracle_gen_hollow_aos(7777), "reset")
        sim.reset(psim.GEN_UNIFORM)
        T.assert_particles_equal(sim.download_aos(), T.oracle_gen_uniform_aos(0, 7777), "reset u")


# ---- AoS import / export ----------------------------------------------------------------------

@pytest.mark.parametrize("n", [1, 255, 256, 257, 10_000])
def test_aos_round_trip_keeps_all_64_bytes(n):
    p = random_particles(n, n)
    with Sim.new(n, psim.GEN_UNIFORM) as sim:
        sim.upload_aos(p)
        assert sim.download_aos().tobytes() == p.tobytes()
        # partial ranges
        if n >= 257:
            part = random_particles(100, 5)
            sim.upload_aos(part, first=130)
            p[130:230] = part
            assert sim.download_aos().tobytes() == p.tobytes()
            assert sim.download_aos(200, 57).tobytes() == p[200:257].tobytes()
        soa = sim.download_soa()
        T.assert_same_f32(soa, T.aos_to_soa(p), "soa view of uploaded data")


@pytest.mark.parametrize("stream", ["0", "1"])
@pytest.mark.parametrize("ppl", ["1", "2", "4"])
def test_transposes_with_several_particles_per_lane(ppl, stream):
    """The AoS transposes exist with 1, 2 and 4 particles per lane, plain and streaming; the
    library picks by size (psim_debug_set_tuning forces a form).  Every form must move all
    64 bytes of every record, whole tiles, ragged tails and ranges that start inside a tile or
    off a 4-particle boundary alike."""
    psim.debug_set_tuning("aos_ppl", int(ppl))
    psim.debug_set_tuning("aos_stream", int(stream))
    try:
        n = 70_001                                       # 273 tiles of 256 and a tail of 113
        p = random_particles(n, 3)
        with Sim.new(n, psim.GEN_UNIFORM) as sim:
            view = sim.get_particle_buffer()
            sim.upload_aos(p)
            assert sim.download_aos().tobytes() == p.tobytes()
            for first, m in ((0, 256), (4, 1000), (128, 512), (7, 300), (32, n - 32), (n - 5, 5),
                             (1024, 64), (64, 63)):
                part = random_particles(m, first + m)
                sim.upload_aos(part, first=first)
                p[first:first + m] = part
                assert sim.download_aos(first, m).tobytes() == part.tobytes(), (first, m)
            assert sim.download_aos().tobytes() == p.tobytes()
            T.assert_same_f32(sim.download_soa(), T.aos_to_soa(p), "SoA side")
            # device to device, into memory of another handle at an unaligned record offset
            with Sim.new(2 * n, psim.GEN_UNIFORM) as other:
                buf = other.get_particle_buffer().color + 64 * 3
                sim.export_aos_device(buf, first=8, n=n - 8)
                sim.sync()
                sim.reset(psim.GEN_UNIFORM)
                sim.import_aos_device(buf, first=8, n=n - 8)
                assert sim.download_aos(8, n - 8).tobytes() == p[8:].tobytes()
            assert view.count == n
    finally:
        psim.debug_set_tuning("aos_ppl", -1)
        psim.debug_set_tuning("aos_stream", -1)


def test_soa_upload_download():
    n = 5000
    soa = np.random.default_rng(3).uniform(-9, 9, size=(6, n)).astype(np.float32)
    with Sim.new(n, psim.GEN_UNIFORM) as sim:
        sim.upload_soa(soa)
        assert sim.download_soa().tobytes() == soa.tobytes()
        sim.upload_soa(soa[:, :100] * 2, first=50)
        soa[:, 50:150] = soa[:, :100] * 2
        assert sim.download_soa().tobytes() == soa.tobytes()
        assert sim.download_soa(40, 20).tobytes() == np.ascontiguousarray(soa[:, 40:60]).tobytes()


def test_device_side_export_import():
    """Export into a device AoS buffer (as a renderer's vertex buffer would be) and
    import it into a second simulation."""
    n = 3000
    p = random_particles(n, 9)
    with Sim.new(n, psim.GEN_UNIFORM) as a, Sim.new(n, psim.GEN_UNIFORM) as b, \
            Sim.new(16 * n, psim.GEN_UNIFORM) as scratch:
        a.upload_aos(p)
        buf = scratch.get_particle_buffer().color              # n * 64 B of device memory
        a.export_aos_device(buf)
        a.sync()
        b.import_aos_device(buf)
        assert b.download_aos().tobytes() == p.tobytes()


def test_paddings_and_initial_colour_survive_updates():
    n = 2048
    p = random_particles(n, 11)
    with Sim.new(n, psim.GEN_UNIFORM) as sim:
        sim.upload_aos(p)
        for mode in (0, 1, 2):
            sim.update(T.bench_params(color_mode=mode))
        out = sim.download_aos()
    for f in ("padding1", "padding2", "initial_color"):
        assert out[f].tobytes() == p[f].tobytes()


# ---- shards --------------------------------------------------------------------------------------

def test_shards_tile_the_global_array():
    total = 100_003
    params = T.bench_params(mouse_radius=50.0)
    cuts = [0, 25_000, 25_001, 70_000, total]
    with Sim.new(total, psim.GEN_UNIFORM) as whole:
        for _ in range(3):
            whole.update(params)
        ref = whole.download_aos()
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        with Sim.new_shard(hi - lo, lo, total, psim.GEN_UNIFORM) as shard:
            for _ in range(3):
                shard.update(params)
            assert shard.download_aos().tobytes() == ref[lo:hi].tobytes()


@pytest.mark.parametrize("mode", [psim.GEN_HOLLOW, psim.GEN_FILLED])
def test_reference_generators_shard_consistently(mode):
    total = 10_000
    with Sim.new(total, mode) as whole:
        ref = whole.download_aos()
    with Sim.new_shard(3000, 6000, total, mode) as shard:
        assert shard.download_aos().tobytes() == ref[6000:9000].tobytes()


def test_page_locked_host_buffers():
    n = 300_001
    with Sim.new(n, psim.GEN_UNIFORM) as sim:
        sim.update(T.bench_params(mouse_radius=50.0, color_mode=1))
        want = sim.download_aos()
        pinned = psim.pinned_empty(n)
        got = sim.download_aos(out=pinned)
        assert got.tobytes() == want.tobytes()
        part = sim.download_aos(1000, 5000, out=pinned)          # a slice into a larger buffer
        assert len(part) == 5000 and part.tobytes() == want[1000:6000].tobytes()
        pinned[:] = want[::-1]                                   # upload from page-locked memory
        sim.upload_aos(pinned)
        assert sim.download_aos().tobytes() == want[::-1].tobytes()
        with pytest.raises(ValueError):
            sim.download_aos(out=np.empty(10, dtype=psim.PARTICLE_DTYPE))
    empty = psim.pinned_empty(0)
    assert len(empty) == 0
    lib = psim.load_library()
    lib.psim_host_free(None)                                     # NULL is allowed
    assert lib.psim_host_alloc(None, 64) == 1


def test_compute_shader_colour_semantics():
    """PSIM_COLOR_RESET: a step in a mode other than 1/2 leaves colour == initial colour
    (compute.wgsl:75-77, 90-92); the default keeps the current colour (cpu.rs:106).  The
    expected state is the CPU oracle's with that one rule applied after every such step."""
    n = 70_001
    modes = [0, 1, 1, 0, 0, 2, 7, 1, 2, 0]
    for fused in (False, True):
        with Sim.new(n, psim.GEN_UNIFORM) as sim:
            assert sim.get_color_semantics() == psim.COLOR_KEEP
            sim.set_color_semantics(psim.COLOR_RESET)
            assert sim.get_color_semantics() == psim.COLOR_RESET
            state = sim.download_aos()
            stream = [T.bench_params(mouse_radius=50.0, color_mode=m) for m in modes]
            for i, p in enumerate(stream):
                state = T.oracle_update_aos(state, p, 1)
                if p.color_mode not in (1, 2):
                    state["color"] = state["initial_color"]
                if not fused:
                    sim.update(p)
                    T.assert_particles_equal(sim.download_aos(), state, f"reset semantics, step {i}")
            if fused:
                sim.update_batch(stream)
                T.assert_particles_equal(sim.download_aos(), state, "reset semantics, fused")
            # uploaded colours count as changed: the next mode-0 step resets them
            custom = state.copy()
            custom["color"] = 0.25
            sim.upload_aos(custom)
            sim.update(stream[0])
            want = T.oracle_update_aos(custom, stream[0], 1)
            want["color"] = want["initial_color"]
            T.assert_particles_equal(sim.download_aos(), want, "reset after upload")
            with pytest.raises(psim.PsimError):
                sim.set_color_semantics(5)


def test_async_download_is_a_snapshot_in_stream_order():
    """Frame k is enqueued for download, then further updates are enqueued behind it: after
    sync the buffer holds frame k, the device holds frame k + 3."""
    n = 5_000_000                                  # two staging chunks
    params = T.bench_params(mouse_radius=50.0)
    with Sim.new(n, psim.GEN_UNIFORM) as sim:
        sim.update(params)
        want_k = sim.download_aos()
        frame = psim.pinned_empty(n)
        got = sim.download_aos_async(frame)
        for _ in range(3):
            sim.update(params)
        sim.sync()
        assert got.tobytes() == want_k.tobytes()
        T.assert_particles_equal(sim.download_aos()[:200_000],
                                 T.oracle_update_aos(want_k[:200_000].copy(), params, 3),
                                 "updates behind the async download")


def test_uniform_shard_beyond_u32_indices():
    first = 5_000_000_000
    with Sim.new_shard(1000, first, 8_000_000_000, psim.GEN_UNIFORM) as shard:
        T.assert_particles_equal(shard.download_aos(), T.oracle_gen_uniform_aos(first, 1000),
                                 "shard past 2^32")


def test_reset_of_a_set_beyond_u32_keeps_to_the_uniform_state():
    """The reference's generators index with u32 (mod.rs:150-188): a shard of a set of more
    than 2^32 particles refuses them on reset as it does on creation, and stays as it was."""
    first, total = 5_000_000_000, 8_000_000_000
    lib = psim.load_library()
    h = ctypes.c_void_p()
    assert lib.psim_create_shard(ctypes.byref(h), 1000, first, total, psim.GEN_HOLLOW, 0) == 1
    with Sim.new_shard(1000, first, total, psim.GEN_UNIFORM) as shard:
        shard.update(T.bench_params(gravity=1.0))
        before = shard.download_aos()
        for mode in (psim.GEN_HOLLOW, psim.GEN_FILLED):
            with pytest.raises(psim.PsimError, match="u32") as e:
                shard.reset(mode)
            assert e.value.code == 1
        assert shard.download_aos().tobytes() == before.tobytes()
        shard.reset(psim.GEN_UNIFORM)
        T.assert_particles_equal(shard.download_aos(), T.oracle_gen_uniform_aos(first, 1000),
                                 "reset shard past 2^32")


# ---- errors -----------------------------------------------------------------------------------------

def test_error_codes():
    lib = psim.load_library()
    h = ctypes.c_void_p()
    assert lib.psim_create(ctypes.byref(h), 10, psim.GEN_UNIFORM, 9999) == 1
    assert b"out of range" in lib.psim_last_error()
    with Sim.new(100, psim.GEN_UNIFORM) as sim:
        with pytest.raises(psim.PsimError) as e:
            sim.download_aos(50, 51)
        assert e.value.code == 1
        with pytest.raises(psim.PsimError):
            sim.resize_buffer(10, 42)
        with pytest.raises(psim.PsimError):
            sim.update_multi(T.bench_params(), np.zeros(65, dtype=psim.ATTRACTOR_DTYPE))
        with pytest.raises(psim.PsimError):
            sim.update_multi_device(T.bench_params(), 0, 3)
        with pytest.raises(psim.PsimError):
            sim.update_multi_device(T.bench_params(), sim.get_particle_buffer().color, 65)
    with Sim.new_shard(10, 0, 100, psim.GEN_UNIFORM) as shard:
        with pytest.raises(psim.PsimError, match="shard"):
            shard.resize_buffer(20, psim.GEN_UNIFORM)


def test_timer_reports_device_time():
    with Sim.new(1_000_000, psim.GEN_UNIFORM) as sim:
        sim.timer_start()
        for _ in range(10):
            sim.update(T.bench_params())
        ms = sim.timer_stop()
        assert 0.0 < ms < 1000.0


# ---- the CPU backend's resize semantics -------------------------------------------------------

def _generated(mode, n):
    return T.oracle_gen_hollow_aos(n) if mode == psim.GEN_HOLLOW else T.oracle_gen_filled_aos(0, n)


@pytest.mark.parametrize("mode", [psim.GEN_HOLLOW, psim.GEN_FILLED])
def test_cpu_backend_resize_semantics(mode):
    """PSIM_RESIZE_APPEND follows src/simulation/cpu.rs:123-186: growing appends a fresh set
    of the additional particles and keeps the stepped ones (also across a reallocation),
    shrinking only hides the tail, growing back reveals it unchanged, reset forgets it."""
    params = T.bench_params(mouse_radius=40.0, gravity=1.0)
    with Sim.new(1000, mode) as sim:
        sim.set_resize_semantics(psim.RESIZE_APPEND)
        assert sim.get_resize_semantics() == psim.RESIZE_APPEND
        for _ in range(3):
            sim.update(params)
        model = T.oracle_update_aos(_generated(mode, 1000), params, 3)
        # grow within the allocation's slack (capacity is a multiple of 1024)
        sim.resize_buffer(1020, mode)
        model = np.concatenate([model, _generated(mode, 20)])
        T.assert_particles_equal(sim.download_aos(), model, "grown by 20")
        # grow past the allocation: a larger buffer, every particle kept
        sim.update(params)
        model = T.oracle_update_aos(model, params)
        other = psim.GEN_FILLED if mode == psim.GEN_HOLLOW else psim.GEN_HOLLOW
        sim.resize_buffer(5000, other)                   # the mode of the call generates
        model = np.concatenate([model, _generated(other, 3980)])
        T.assert_particles_equal(sim.download_aos(), model, "grown past the capacity")
        # shrink: the tail is hidden and frozen while the head steps on
        sim.resize_buffer(700, mode)
        assert sim.get_particle_count() == 700
        sim.update(params)
        model[:700] = T.oracle_update_aos(model[:700], params)
        T.assert_particles_equal(sim.download_aos(), model[:700], "shrunk")
        # grow back within what was generated: nothing new, the frozen tail returns
        sim.resize_buffer(4000, mode)
        T.assert_particles_equal(sim.download_aos(), model[:4000], "grown back")
        # same count: no-op
        sim.resize_buffer(4000, other)
        T.assert_particles_equal(sim.download_aos(), model[:4000], "no-op")
        # reset regenerates the active count and forgets the rest (cpu.rs:172-186)
        sim.reset(mode)
        T.assert_particles_equal(sim.download_aos(), _generated(mode, 4000), "reset")
        sim.resize_buffer(4500, mode)
        T.assert_particles_equal(sim.download_aos(),
                                 np.concatenate([_generated(mode, 4000), _generated(mode, 500)]),
                                 "grown after reset: 4000 is the new high-water mark")
        # back to the compute backend's rule: everything regenerated
        sim.set_resize_semantics(psim.RESIZE_REGENERATE)
        sim.resize_buffer(300, mode)
        T.assert_particles_equal(sim.download_aos(), _generated(mode, 300), "regenerated")


def test_append_is_refused_on_shards():
    with Sim.new_shard(100, 50, 1000, psim.GEN_UNIFORM) as shard:
        with pytest.raises(psim.PsimError):
            shard.set_resize_semantics(psim.RESIZE_APPEND)
