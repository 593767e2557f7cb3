"""GPU tests of the attractor exchange behind the C ABI (psim_exchange_*): RCCL all-gather
of one attractor per rank on a communication stream into a ring of 16 receive lists, feeding
the multi-attractor kernel.  A one-GPU machine can hold one rank only (RCCL refuses two ranks on
one device), so these run a world of one: the collective, the streams, the events and the
buffer rotation are the ones every world size uses.

The cases live in tests/_exchange_worker.py and run there, all in ONE child process: the
exchange maps RCCL, other tests of this suite import torch with its own ROCm libraries, and a
process holding both aborts at exit."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import psim_testlib as T
from psim_testlib import psim

pytestmark = pytest.mark.gpu

WORKER = os.path.join(T.ROOT, "tests", "_exchange_worker.py")
CASES = [
    "case_errors",
    "case_file_rendezvous_and_a_second_communicator",
    "case_host_paced_steps_equal_the_oracle_40",
    "case_host_time_of_a_step_hides_behind_the_kernel",
    "case_one_rank_equals_the_single_attractor_update",
    "case_one_rank_steps_equal_the_oracle_1",
    "case_one_rank_steps_equal_the_oracle_2",
    "case_one_rank_steps_equal_the_oracle_40",
    "case_one_rank_steps_equal_the_oracle_7",
    "case_pacing_switched_every_5_steps_equals_the_oracle",
    "case_pipelined_steps_equal_the_oracle_1",
    "case_pipelined_steps_equal_the_oracle_40",
    "case_pipelined_the_handle_is_one_step_behind_until_sync",
    "case_three_thousand_small_steps_reuse_the_ring_correctly",
    "case_unknown_pacing_is_rejected",
    "case_with_a_timeout_set_device_paced_steps_equal_the_oracle_40",
    "case_with_a_timeout_set_host_paced_steps_equal_the_oracle_40",
    "case_with_a_timeout_set_pipelined_steps_equal_the_oracle_40",
    "case_with_a_timeout_set_the_pacing_may_still_change",
]


@pytest.fixture(scope="module")
def worker_results(tmp_path_factory):
    d = tmp_path_factory.mktemp("exchange")
    out = d / "results.json"
    r = subprocess.run([sys.executable, "-u", WORKER, str(out), str(d)], capture_output=True,
                       text=True, timeout=270)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    print(r.stdout[-2000:])
    return json.loads(out.read_text())


def test_worker_ran_every_case(worker_results):
    assert sorted(worker_results) == CASES


@pytest.mark.parametrize("case", CASES)
def test_exchange(worker_results, case):
    assert worker_results[case] == "ok", worker_results[case]


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


def build_c_hosts():
    cpp_dir = os.path.join(T.ROOT, "tests", "cpp")
    subprocess.check_call(["make", "-C", cpp_dir, "exchange_loop", "libfake_rccl.so"],
                          stdout=subprocess.DEVNULL)
    return cpp_dir


def start_ranks(tmp_path, world, n, steps, pacing, extra=lambda rank: (), env_extra=None):
    """`world` C host processes (tests/cpp/exchange_loop) sharing GPU 0, the exchange pointed at
    the test double; rank 0 is started last, so the others really wait for its rendezvous file.
    Returns {rank: process}."""
    cpp_dir = build_c_hosts()
    env = dict(os.environ, PSIM_RCCL_LIBRARY=os.path.join(cpp_dir, "libfake_rccl.so"))
    env.update(env_extra or {})
    return {r: subprocess.Popen([os.path.join(cpp_dir, "exchange_loop"), str(r), str(world), "0",
                                 str(n), str(steps), str(tmp_path / "id"), pacing, *extra(r)],
                                env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            for r in reversed(range(world))}


def checksums_of(procs, timeout=150):
    outs = {}
    for r, p in procs.items():
        out, err = p.communicate(timeout=timeout)
        assert p.returncode == 0, f"rank {r}: {err[-2000:]}"
        outs[r] = [int(v) for v in out.strip().splitlines()[-1].split()[2:8]]
    return outs


def c_host_attractors(world, t, force=40.0, radius=25.0):
    """attractor_at() of tests/cpp/exchange_loop.c for all ranks."""
    a = np.zeros(world, dtype=psim.ATTRACTOR_DTYPE)
    for k in range(world):
        a["position"][k] = (-40.0 + 10.0 * k + 0.25 * (t % 64), 30.0 - 0.5 * (t % 32),
                            -20.0 + 5.0 * k)
    a["force"], a["radius"], a["active"] = force, radius, 1
    return a


C_HOST_PARAMS = dict(delta_time=0.016, gravity=0.25, color_mode=0, mouse_force=5.0,
                     mouse_radius=10.0, is_mouse_dragging=0, damping=0.99, max_dist_for_color=50.0)


@pytest.mark.parametrize("world,pacing,timeout_ms", [(2, "device", 0), (3, "device", 0),
                                                     (3, "host", 0), (3, "device", 20000),
                                                     (3, "host", 20000), (2, "pipelined", 0),
                                                     (3, "pipelined", 0), (3, "pipelined", 20000)])
def test_several_ranks_on_one_gpu_through_a_test_double(tmp_path, world, pacing, timeout_ms):
    """The exchange's logic for a world larger than one -- every rank's place in the receive
    lists, the ring, the list length the kernel gets, shards of one global array, ranks that
    really wait for the rendezvous file -- with `world` C host processes sharing GPU 0.  RCCL
    itself refuses that, so PSIM_RCCL_LIBRARY points the exchange at tests/cpp/libfake_rccl.so,
    a test double whose all-gather is asynchronous and stream-ordered like the real one but goes
    through shared memory (see its header); the real collective runs in the cases above (one
    rank) and in bench.py on a multi-GPU node.  With a timeout set (and nobody gone) the polling
    waits must change nothing."""
    n, steps = 50_003, 20                                # 40 steps in all: the ring goes round
    extra = (lambda r: (f"timeout_ms={timeout_ms}",)) if timeout_ms else (lambda r: ())
    outs = checksums_of(start_ranks(tmp_path, world, n, steps, pacing, extra))
    params = T.make_params(**C_HOST_PARAMS)
    for r in range(world):
        want = T.oracle_gen_uniform_aos(r * n, n)
        for t in range(2 * steps):
            want = T.oracle_update_aos(want, params, 1, c_host_attractors(world, t))
        assert outs[r] == T.oracle_checksum_soa(T.aos_to_soa(want)), f"rank {r} of {world}"


@pytest.mark.parametrize("pacing", ["host", "device", "pipelined"])
def test_an_all_gather_kernel_as_large_as_rccls_changes_no_result(tmp_path, pacing):
    """The test double with its all-gather kernel given the resource footprint of RCCL's own
    device kernel (37 664 B of LDS, 248 vector registers, 512 threads: a workgroup that waits
    for most of a CU, which the update kernels only yield when their grid runs dry): two ranks
    sharing the GPU still deliver the oracle's particles, whoever waits for whom."""
    world, n, steps = 2, 300_007, 12
    outs = checksums_of(start_ranks(tmp_path, world, n, steps, pacing,
                                    env_extra={"PSIM_FAKE_RCCL_FAT": "1"}))
    params = T.make_params(**C_HOST_PARAMS)
    for r in range(world):
        want = T.oracle_gen_uniform_aos(r * n, n)
        for t in range(2 * steps):
            want = T.oracle_update_aos(want, params, 1, c_host_attractors(world, t))
        assert outs[r] == T.oracle_checksum_soa(T.aos_to_soa(want)), f"rank {r} of {world}"


def test_ranks_may_pace_themselves_differently(tmp_path):
    """The pacing is each rank's own business: rank 0 host-paced, rank 1 pipelined, rank 2
    device-paced with a host that has other work per step (so it never runs ahead)."""
    world, n, steps = 3, 50_003, 20
    cpp_dir = build_c_hosts()
    env = dict(os.environ, PSIM_RCCL_LIBRARY=os.path.join(cpp_dir, "libfake_rccl.so"))
    how = {0: ("host",), 1: ("pipelined",), 2: ("device", "nap_us=200")}
    procs = {r: subprocess.Popen([os.path.join(cpp_dir, "exchange_loop"), str(r), str(world), "0",
                                  str(n), str(steps), str(tmp_path / "id"), *how[r]], env=env,
                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in reversed(range(world))}
    outs = checksums_of(procs)
    params = T.make_params(**C_HOST_PARAMS)
    for r in range(world):
        want = T.oracle_gen_uniform_aos(r * n, n)
        for t in range(2 * steps):
            want = T.oracle_update_aos(want, params, 1, c_host_attractors(world, t))
        assert outs[r] == T.oracle_checksum_soa(T.aos_to_soa(want)), f"rank {r}"


@pytest.mark.parametrize("pacing", ["host", "device", "pipelined"])
def test_fifty_million_particles_over_eight_ranks_with_eight_attractors(tmp_path, pacing):
    """The 50 M-particle configuration in its own shape -- 8 ranks of 6.25 M particles, one
    moving attractor each with r = 10 and f = 5, all-gathered every step -- with the 8 rank
    processes sharing this machine's GPU over the test double.  Every rank's shard of the one
    50 M array after 24 exchanged steps must carry the oracle's checksum."""
    world, n, steps = 8, 6_250_000, 12                   # 24 steps in all: the ring goes round
    procs = start_ranks(tmp_path, world, n, steps, pacing,
                        lambda r: ("force=5", "radius=10", "timeout_ms=60000"))
    outs = checksums_of(procs, timeout=240)
    params = T.make_params(**C_HOST_PARAMS)
    for r in range(world):
        want = T.oracle_gen_uniform_soa(r * n, n)
        for t in range(2 * steps):
            want, _ = T.oracle_update_soa(want, params, 1,
                                          attractors=c_host_attractors(world, t, 5.0, 10.0))
        assert outs[r] == T.oracle_checksum_soa(want), f"rank {r} of {world}, {pacing}-paced"


@pytest.mark.parametrize("pacing,steps,where", [
    ("host", 30, "psim_exchange_step"),      # the gather of the step after the rank left
    ("device", 30, "psim_exchange_step"),    # 16 steps later: the gather that used the list before
    ("device", 12, "psim_exchange_sync"),    # the run ends first: the wait at its end
    ("pipelined", 30, "psim_exchange_step"), # one call later than host-paced: the gather before
])
def test_a_rank_that_has_gone_is_an_error_not_a_hang(tmp_path, pacing, steps, where):
    """Three ranks; rank 2 leaves after 5 steps, as a crashed process would.  With
    psim_exchange_set_timeout the other two must come back from the exchange with PSIM_ERR_COMM
    within the timeout -- not hang in a collective that will never complete -- and must be able
    to tear down (the C host exits with status 3 from main, after its error message)."""
    world, n, timeout_ms = 3, 50_003, 3000
    t0 = time.monotonic()
    procs = start_ranks(tmp_path, world, n, steps, pacing,
                        lambda r: ("die_after=5",) if r == 2 else (f"timeout_ms={timeout_ms}",))
    results = {r: (p.communicate(timeout=120), p.returncode) for r, p in procs.items()}
    elapsed = time.monotonic() - t0
    (_, err2), rc2 = results[2]
    assert rc2 == 0 and "leaves after 5 steps" in err2
    for r in (0, 1):
        (out, err), rc = results[r]
        assert rc == 3, f"rank {r}: status {rc}\n{out[-500:]}\n{err[-2000:]}"
        assert where in err and "status 6" in err, err[-2000:]
        assert "did not finish within 3000 ms" in err or "communicator reports" in err, err[-2000:]
        took = float(err.rsplit("after ", 1)[1].split(" s]")[0])
        assert took < timeout_ms * 1e-3 + 2.0, err[-500:]
        assert err.rstrip().endswith("torn down"), err[-500:]
    assert elapsed < 60, f"the run took {elapsed:.0f} s: somebody hung"
