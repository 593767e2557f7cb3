"""GPU tests of the single-process multi-shard group (psim_group_*).  One GPU is visible
here, so the shards are placed on the same device: what is checked is the partition -- the
group must behave exactly like one handle holding the whole set."""
import numpy as np
import pytest

import psim_testlib as T
from psim_testlib import psim

pytestmark = pytest.mark.gpu

Sim = psim.HipParticleSimulation
Group = psim.HipParticleGroup


def params(**kw):
    return T.bench_params(mouse_radius=30.0, gravity=0.5, **kw)


@pytest.mark.parametrize("total,shards", [(100_003, 3), (5, 8), (1, 1), (0, 2), (1_000_000, 4)])
@pytest.mark.parametrize("mode", [psim.GEN_UNIFORM, psim.GEN_HOLLOW, psim.GEN_FILLED])
def test_group_equals_one_handle(total, shards, mode):
    if mode == psim.GEN_HOLLOW and total == 1:
        pytest.skip("a one-particle hollow sphere is NaN by construction (mod.rs:153)")
    with Group(total, [0] * shards, mode) as g, Sim.new(total, mode) as one:
        assert g.get_particle_count() == total and g.shard_count() == shards
        sizes = [g.shard(k).get_particle_count() for k in range(shards)]
        assert sum(sizes) == total and max(sizes) - min(sizes) <= 1
        assert g.download_aos().tobytes() == one.download_aos().tobytes()
        for step in range(4):
            p = params(color_mode=step % 3)
            g.update(p)
            one.update(p)
        g.update_batch([params(), params(delta_time=0.01)])
        one.update_batch([params(), params(delta_time=0.01)])
        g.sync()
        assert g.download_aos().tobytes() == one.download_aos().tobytes()
        assert g.checksum() == one.checksum()
        g.reset(mode)
        one.reset(mode)
        assert g.download_aos().tobytes() == one.download_aos().tobytes()


def test_ranges_across_shard_boundaries():
    total = 10_007
    rng = np.random.default_rng(11)
    with Group(total, [0, 0, 0], psim.GEN_UNIFORM) as g:
        whole = g.download_aos()
        cuts = [0, total // 3, 2 * total // 3, total]
        for first, n in [(0, 1), (cuts[1] - 2, 5), (cuts[1], cuts[2] - cuts[1]), (cuts[2] - 1, 2),
                         (17, total - 17), (total, 0), (cuts[1] - 1, cuts[2] - cuts[1] + 2)]:
            assert g.download_aos(first, n).tobytes() == whole[first:first + n].tobytes()
        patch = whole[cuts[1] - 3:cuts[2] + 3].copy()
        patch["position"] = rng.uniform(-5, 5, patch["position"].shape).astype(np.float32)
        g.upload_aos(patch, cuts[1] - 3)
        whole[cuts[1] - 3:cuts[2] + 3] = patch
        assert g.download_aos().tobytes() == whole.tobytes()
        g.update(params())
        T.assert_particles_equal(g.download_aos(), T.oracle_update_aos(whole, params()), "after upload")
        with pytest.raises(psim.PsimError):
            g.download_aos(total - 1, 2)


def test_group_checkpoint_restores_into_a_handle_and_back(tmp_path):
    path = tmp_path / "group.ckp"
    total = 300_001
    with Group(total, [0, 0, 0, 0], psim.GEN_UNIFORM) as g:
        g.update(params(color_mode=1))
        g.save_checkpoint(path, step=1)
        state = g.download_aos()
        g.update(params())
        stepped = g.download_aos()
    assert psim.checkpoint_info(path)["particle_count"] == total
    assert psim.read_checkpoint_records(path).tobytes() == state.tobytes()
    with Sim.new(total, psim.GEN_HOLLOW) as one:
        assert one.load_checkpoint(path) == 1
        one.update(params())
        assert one.download_aos().tobytes() == stepped.tobytes()
        one.save_checkpoint(path, step=2)
    with Group(total, [0, 0], psim.GEN_HOLLOW) as g:
        assert g.load_checkpoint(path) == 2
        assert g.download_aos().tobytes() == stepped.tobytes()


def test_bad_device_is_reported():
    with pytest.raises(psim.PsimError) as e:
        Group(1000, [0, 99])
    assert e.value.code == 1 and "99" in str(e.value)


@pytest.mark.parametrize("k", [1, 8, 20])
def test_group_multi_attractor_step(k):
    """One host list for every shard: the several-GPU multi-attractor configuration of a
    single process needs no exchange between devices."""
    rng = np.random.default_rng(k)
    attrs = np.zeros(k, dtype=psim.ATTRACTOR_DTYPE)
    attrs["position"] = rng.uniform(-40, 40, size=(k, 3)).astype(np.float32)
    attrs["force"], attrs["radius"], attrs["active"] = 20.0, 12.0, 1
    total = 50_003
    with Group(total, [0, 0, 0], psim.GEN_UNIFORM) as g:
        start = g.download_aos()
        for _ in range(3):
            g.update_multi(params(), attrs)
        got = g.download_aos()
    T.assert_particles_equal(got, T.oracle_update_aos(start, params(), 3, attrs), f"{k} attractors")


def test_group_over_every_visible_device():
    """One shard per visible GPU (and a second round of them, so that two shards share each
    device): with one GPU this is the partition test again; on a larger machine the shards
    really live on distinct devices, are created, reset, summed and transferred on all of
    them at once, and must still equal one handle holding the whole set."""
    ndev = psim.device_count()
    assert ndev >= 1
    devices = list(range(ndev)) * 2
    total = 1_000_003
    with Group(total, devices, psim.GEN_FILLED) as g, Sim.new(total, psim.GEN_FILLED) as one:
        assert [g.shard(k).device() for k in range(len(devices))] == devices
        assert g.download_aos().tobytes() == one.download_aos().tobytes()
        lst = np.zeros(3, dtype=psim.ATTRACTOR_DTYPE)
        lst["position"] = [(0, 0, 30), (20, -10, 0), (-30, 5, 5)]
        lst["force"], lst["radius"], lst["active"] = 30.0, 25.0, 1
        for step in range(3):
            p = params(color_mode=step % 3)
            g.update(p)
            one.update(p)
            g.update_multi(p, lst)
            one.update_multi(p, lst)
        g.sync()
        assert g.checksum() == one.checksum()
        assert g.download_aos().tobytes() == one.download_aos().tobytes()
        g.reset(psim.GEN_UNIFORM)
        one.reset(psim.GEN_UNIFORM)
        assert g.checksum() == one.checksum()


def test_group_reports_the_failing_shard():
    """Errors of the concurrently created shards reach the caller with their text."""
    with pytest.raises(psim.PsimError) as e:
        Group(1000, [0, 9999], psim.GEN_UNIFORM)
    assert e.value.code == 1 and "out of range" in str(e.value)
