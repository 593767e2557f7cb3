"""GPU tests of the attractor exchange behind the C ABI (psim_exchange_*): RCCL all-gather
of one attractor per rank on a communication stream into a ring of 16 receive lists, feeding
the multi-attractor kernel.  A one-GPU machine can hold one rank only (RCCL refuses two ranks on
one device), so these run a world of one: the collective, the streams, the events and the
buffer rotation are the ones every world size uses.  No torch anywhere in this file."""
import importlib
import os
import subprocess

import numpy as np
import pytest

import psim_testlib as T
from psim_testlib import psim

pytestmark = pytest.mark.gpu

par = importlib.import_module("particle-simulation-3d_amd.parallel")


@pytest.mark.parametrize("steps", [1, 2, 7, 40])          # 40: the ring of 16 lists goes round
def test_one_rank_steps_equal_the_oracle(steps):
    n = 300_007
    params = T.bench_params(gravity=0.5, is_mouse_dragging=0, color_mode=2)
    runner = par.NativeMultiAttractorRunner(n, 0, 1, 0, params, psim.exchange_unique_id(),
                                            force=60.0, radius=30.0)
    try:
        for _ in range(steps):
            runner.step()
        runner.finish()
        got = runner.sim.download_aos()
        last = runner.exchange.last_list()
    finally:
        runner.close()
    want = T.oracle_gen_uniform_aos(0, n)
    for t in range(steps):
        want = T.oracle_update_aos(want, params, 1, par.attractors_at(t, 1, force=60.0, radius=30.0))
    T.assert_particles_equal(got, want, f"{steps} exchanged steps")
    assert last.tobytes() == par.attractors_at(steps - 1, 1, force=60.0, radius=30.0).tobytes()


def test_one_rank_equals_the_single_attractor_update():
    """With one rank the exchanged step is the reference's own step with that attractor."""
    n = 100_003
    mine = par.attractor_at(5, 0, force=40.0, radius=25.0)
    params = T.bench_params(mouse_force=40.0, mouse_radius=25.0,
                            mouse_position=tuple(float(v) for v in mine["position"][0]))
    with psim.HipParticleSimulation.new(n, psim.GEN_UNIFORM) as plain, \
            psim.HipParticleSimulation.new_shard(n, 0, n, psim.GEN_UNIFORM) as sharded:
        with psim.HipAttractorExchange(sharded, psim.exchange_unique_id(), 0, 1) as ex:
            for _ in range(3):
                plain.update(params)
                ex.step(params, mine)
            ex.sync()
            assert sharded.download_aos().tobytes() == plain.download_aos().tobytes()


def test_file_rendezvous_and_a_second_communicator(tmp_path):
    """Rank 0 publishes the id as a file; a later reader gets the same bytes; two exchanges
    (two communicators) can live in one process."""
    path = tmp_path / "id"
    ident = psim.exchange_rendezvous_file(path, 0)
    assert len(ident) == 128 and path.read_bytes() == ident
    assert psim.exchange_rendezvous_file(path, 1, timeout_ms=100) == ident
    params = T.bench_params()
    with psim.HipParticleSimulation.new_shard(10_000, 0, 10_000, psim.GEN_UNIFORM) as a, \
            psim.HipParticleSimulation.new_shard(10_000, 0, 10_000, psim.GEN_UNIFORM) as b:
        with psim.HipAttractorExchange(a, ident, 0, 1) as xa, \
                psim.HipAttractorExchange(b, psim.exchange_unique_id(), 0, 1) as xb:
            for t in range(4):
                xa.step(params, par.attractor_at(t, 0))
                xb.step(params, par.attractor_at(t, 0))
            xa.sync()
            xb.sync()
            assert a.download_aos().tobytes() == b.download_aos().tobytes()


def test_errors():
    with psim.HipParticleSimulation.new_shard(1000, 0, 1000, psim.GEN_UNIFORM) as sim:
        with psim.HipAttractorExchange(sim, psim.exchange_unique_id(), 0, 1) as ex:
            with pytest.raises(psim.PsimError) as e:
                ex.last_list()                           # no step yet
            assert e.value.code == 1
        with pytest.raises(ValueError):
            psim.HipAttractorExchange(sim, b"short", 0, 1)


def test_host_time_of_a_step_hides_behind_the_kernel():
    """6.25 M particles (one GPU's share of the 50 M configuration): the exchanged step must
    cost about what its kernel costs -- nothing in it waits on the host.  Informational
    figures are printed; the bound asserted is loose (shared test machines)."""
    n, steps = 6_250_000, 300
    params = T.bench_params()
    runner = par.NativeMultiAttractorRunner(n, 0, 1, 0, params, psim.exchange_unique_id())
    try:
        for _ in range(30):
            runner.step()
        runner.finish()
        runner.sim.timer_start()
        for _ in range(steps):
            runner.step()
        exchanged = runner.sim.timer_stop() / steps * 1e3
        runner.finish()
        table = par.attractors_at(0, 1)
        for _ in range(30):
            runner.sim.update_multi(params, table)
        runner.sim.timer_start()
        for _ in range(steps):
            runner.sim.update_multi(params, table)
        kernel = runner.sim.timer_stop() / steps * 1e3
    finally:
        runner.close()
    print(f"exchanged step {exchanged:.1f} us, the same kernel from a host list {kernel:.1f} us")
    assert exchanged < 1.3 * kernel + 8.0


def test_plain_c_host_of_the_exchange(tmp_path):
    """tests/cpp/exchange_loop.c: a C99 program with nothing but include/psim.h -- file
    rendezvous, one exchanged step per frame with an attractor on an exactly representable
    path -- must leave the particles the oracle leaves (compared by checksum)."""
    cpp_dir = os.path.join(T.ROOT, "tests", "cpp")
    subprocess.check_call(["make", "-C", cpp_dir, "exchange_loop"], stdout=subprocess.DEVNULL)
    n, steps = 200_003, 21
    r = subprocess.run([os.path.join(cpp_dir, "exchange_loop"), "0", "1", "0", str(n), str(steps),
                        str(tmp_path / "id")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    fields = r.stdout.strip().splitlines()[-1].split()
    host_us, device_us = float(fields[0]), float(fields[1])
    got = [int(v) for v in fields[2:8]]
    params = T.make_params(delta_time=0.016, gravity=0.25, color_mode=0, mouse_force=5.0,
                           mouse_radius=10.0, is_mouse_dragging=0, damping=0.99,
                           max_dist_for_color=50.0)
    want = T.oracle_gen_uniform_aos(0, n)
    for t in range(2 * steps):                           # the program runs its steps twice
        a = np.zeros(1, dtype=psim.ATTRACTOR_DTYPE)
        a["position"][0] = (-40.0 + 0.25 * (t % 64), 30.0 - 0.5 * (t % 32), -20.0)
        a["force"], a["radius"], a["active"] = 40.0, 25.0, 1
        want = T.oracle_update_aos(want, params, 1, a)
    assert got == T.oracle_checksum_soa(T.aos_to_soa(want))
    print(f"C host: {host_us:.1f} us of host time, {device_us:.1f} us of device time per step")
