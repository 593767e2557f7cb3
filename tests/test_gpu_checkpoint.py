"""GPU tests of the checkpoint entry points: the file holds exactly the 64-byte records the
handle would download, restores into any sharding of the same set, and stepping on from a
restored state equals stepping on without the detour (and the oracle)."""
import numpy as np
import pytest

import psim_testlib as T
from psim_testlib import psim

pytestmark = pytest.mark.gpu

Sim = psim.HipParticleSimulation


def colour_params(mode):
    return T.bench_params(mouse_radius=30.0, gravity=0.5, color_mode=mode)


@pytest.mark.parametrize("n", [1, 1000, 300_001, 1_300_000])
def test_round_trip_and_continuation(tmp_path, n):
    # 1.3 M particles cross three 512 Ki-particle slots of the transfer pipeline
    path = tmp_path / "state.ckp"
    with Sim.new(n, psim.GEN_UNIFORM) as sim:
        start = sim.download_aos()
        for _ in range(3):
            sim.update(colour_params(1))
        sim.save_checkpoint(path, step=3)
        saved = sim.download_aos()
        for _ in range(2):
            sim.update(colour_params(0))
        straight = sim.download_aos()
    info = psim.checkpoint_info(path)
    assert info["particle_count"] == n and info["step"] == 3
    assert info["generation_mode"] == psim.GEN_UNIFORM and info["color_semantics"] == psim.COLOR_KEEP
    assert path.stat().st_size == 64 + 64 * n
    assert psim.read_checkpoint_records(path).tobytes() == saved.tobytes()
    want = T.oracle_update_aos(start, colour_params(1), 3)
    T.assert_particles_equal(saved, want, "checkpointed state")
    with Sim.new(n, psim.GEN_HOLLOW) as other:          # a different state to overwrite
        assert other.load_checkpoint(path) == 3
        assert other.download_aos().tobytes() == saved.tobytes()
        for _ in range(2):
            other.update(colour_params(0))
        assert other.download_aos().tobytes() == straight.tobytes()
    T.assert_particles_equal(straight, T.oracle_update_aos(want, colour_params(0), 2), "continued")


def test_empty_set(tmp_path):
    path = tmp_path / "empty.ckp"
    with Sim.new(0, psim.GEN_UNIFORM) as sim:
        sim.save_checkpoint(path, step=9)
        assert path.stat().st_size == 64
        assert sim.load_checkpoint(path) == 9


def test_overwriting_a_longer_file_truncates_it(tmp_path):
    path = tmp_path / "state.ckp"
    with Sim.new(5000, psim.GEN_UNIFORM) as big, Sim.new(1200, psim.GEN_FILLED) as small:
        big.save_checkpoint(path)
        small.save_checkpoint(path, step=1)
        assert path.stat().st_size == 64 + 64 * 1200
        assert psim.read_checkpoint_records(path).tobytes() == small.download_aos().tobytes()


def test_count_mismatch_and_bad_file(tmp_path):
    path = tmp_path / "state.ckp"
    with Sim.new(1000, psim.GEN_UNIFORM) as sim:
        sim.save_checkpoint(path)
    with Sim.new(999, psim.GEN_UNIFORM) as sim:
        before = sim.download_aos()
        with pytest.raises(psim.PsimError) as e:
            sim.load_checkpoint(path)
        assert e.value.code == 1 and "1000" in str(e.value)
        with pytest.raises(psim.PsimError) as e:
            sim.load_checkpoint(tmp_path / "missing.ckp")
        assert e.value.code == 5
        with pytest.raises(psim.PsimError) as e:
            sim.save_checkpoint(tmp_path / "no_such_dir" / "x.ckp")
        assert e.value.code == 5
        assert sim.download_aos().tobytes() == before.tobytes()   # nothing was touched


def test_shards_write_one_file_and_any_sharding_restores_it(tmp_path):
    path = tmp_path / "whole.ckp"
    total = 700_003
    cuts = [0, 250_001, 250_001, 600_000, total]         # the second shard is empty
    shards = [Sim.new_shard(cuts[k + 1] - cuts[k], cuts[k], total, psim.GEN_UNIFORM)
              for k in range(4)]
    try:
        for s in shards:
            s.update(colour_params(2))
        psim.checkpoint_begin(path)
        for k, s in enumerate(reversed(shards)):         # any order
            s.save_checkpoint(path)
            with pytest.raises(psim.PsimError):          # no shard wrote a header
                psim.checkpoint_info(path)
            if k == 1:                                   # the last particle's owner is in, not all
                assert path.stat().st_size == 64 + 64 * total
        shards[2].commit_checkpoint(path, step=1)        # any shard may commit
        pieces = [s.download_aos() for s in shards]
    finally:
        for s in shards:
            s.close()
    whole = np.concatenate(pieces)
    assert path.stat().st_size == 64 + 64 * total
    assert psim.checkpoint_info(path)["particle_count"] == total
    assert psim.read_checkpoint_records(path).tobytes() == whole.tobytes()
    with Sim.new(total, psim.GEN_HOLLOW) as sim:         # one handle restores the whole set
        assert sim.load_checkpoint(path) == 1
        assert sim.download_aos().tobytes() == whole.tobytes()
        sim.update(colour_params(0))
        stepped = sim.download_aos()
    # ... and so does a different sharding; stepping on gives the same particles
    first, count = 123_457, 400_000
    with Sim.new_shard(count, first, total, psim.GEN_HOLLOW) as part:
        part.load_checkpoint(path)
        assert part.download_aos().tobytes() == whole[first:first + count].tobytes()
        part.update(colour_params(0))
        assert part.download_aos().tobytes() == stepped[first:first + count].tobytes()


def test_restored_colours_follow_the_reset_rule(tmp_path):
    """Under the compute shader's colour rule a step in mode 0 resets the colours to the
    initial ones; a restore brings colours in that the handle must treat as changed."""
    path = tmp_path / "state.ckp"
    with Sim.new(5000, psim.GEN_UNIFORM) as sim:
        sim.update(colour_params(1))
        sim.save_checkpoint(path)
    with Sim.new(5000, psim.GEN_UNIFORM) as sim:
        sim.set_color_semantics(psim.COLOR_RESET)
        sim.load_checkpoint(path)
        loaded = sim.download_aos()
        assert loaded["color"].tobytes() != loaded["initial_color"].tobytes()
        sim.update(colour_params(0))
        after = sim.download_aos()
        assert after["color"].tobytes() == after["initial_color"].tobytes()


def test_an_interrupted_sharded_write_leaves_no_valid_file(tmp_path):
    """Overwriting a valid checkpoint of the same size shard by shard: between begin and
    commit the file is rejected; a commit before the last particle's owner has written fails."""
    path = tmp_path / "whole.ckp"
    total = 10_000
    with Sim.new(total, psim.GEN_FILLED) as whole:
        whole.save_checkpoint(path, step=5)
    assert psim.checkpoint_info(path)["step"] == 5
    with Sim.new_shard(6000, 0, total, psim.GEN_UNIFORM) as a, \
            Sim.new_shard(4000, 6000, total, psim.GEN_UNIFORM) as b:
        psim.checkpoint_begin(path)
        a.save_checkpoint(path)                          # the run "dies" here: b never writes
        with pytest.raises(psim.PsimError) as e:
            psim.checkpoint_info(path)
        assert e.value.code == 5
        with Sim.new(total, psim.GEN_HOLLOW) as reader:
            with pytest.raises(psim.PsimError):
                reader.load_checkpoint(path)
        # a fresh file, where the size tells that a shard is missing
        fresh = tmp_path / "fresh.ckp"
        psim.checkpoint_begin(fresh)
        a.save_checkpoint(fresh)
        with pytest.raises(psim.PsimError) as e:
            a.commit_checkpoint(fresh, step=1)
        assert e.value.code == 5 and "has not written" in str(e.value)
        b.save_checkpoint(fresh)
        b.commit_checkpoint(fresh, step=2)
        assert psim.checkpoint_info(fresh)["step"] == 2
        want = np.concatenate([a.download_aos(), b.download_aos()])
    assert psim.read_checkpoint_records(fresh).tobytes() == want.tobytes()


def test_a_shard_that_writes_without_begin_still_invalidates_the_old_header(tmp_path):
    """A caller that forgets psim_checkpoint_begin: the first shard that writes over an existing
    valid checkpoint of the same size must itself leave no valid header -- or half-rewritten
    records would pass for the old, complete file."""
    path = tmp_path / "whole.ckp"
    total = 10_000
    with Sim.new(total, psim.GEN_FILLED) as whole:
        whole.save_checkpoint(path, step=5)
    assert psim.checkpoint_info(path)["step"] == 5
    with Sim.new_shard(6000, 0, total, psim.GEN_UNIFORM) as a, \
            Sim.new_shard(4000, 6000, total, psim.GEN_UNIFORM) as b:
        a.save_checkpoint(path)                          # no begin; b never writes
        with pytest.raises(psim.PsimError) as e:
            psim.checkpoint_info(path)
        assert e.value.code == 5
        b.save_checkpoint(path)
        with pytest.raises(psim.PsimError):
            psim.checkpoint_info(path)                   # still none: only the commit makes one
        b.commit_checkpoint(path, step=9)
        assert psim.checkpoint_info(path)["step"] == 9
        want = np.concatenate([a.download_aos(), b.download_aos()])
    assert psim.read_checkpoint_records(path).tobytes() == want.tobytes()
