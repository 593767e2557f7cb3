"""CPU test of the host logic that picks the update kernel's launch shape by state size
(psim_launch_shape; no device needed).  The table is the outcome of the measurements kept
under profiles/tuning/: it is pinned here so that a change of it is a deliberate one."""
import pytest

from psim_testlib import psim

PLAIN, RESIDENT, STREAM = psim.ACCESS_PLAIN, psim.ACCESS_RESIDENT, psim.ACCESS_STREAM


@pytest.mark.parametrize("count,access,reserve", [
    (100_000, PLAIN, 0),            # fits the L2s: plain accesses, full occupancy
    (1_000_000, PLAIN, 0),
    (3_999_999, PLAIN, 0),
    (4_000_000, RESIDENT, 26624),   # 96 MB .. 268 MB: the Infinity Cache, sc1 stores, 6 per CU
    (6_250_000, RESIDENT, 26624),   # the 50 M / 8 GPU configuration
    (10_000_000, RESIDENT, 26624),  # the 10 M configuration
    (11_000_000, RESIDENT, 26624),
    (12_500_000, PLAIN, 23040),     # just past the cache: plain accesses, 7 per CU
    (14_000_000, PLAIN, 23040),
    (16_000_000, STREAM, 26624),    # streamed from HBM: nt loads, nt sc1 stores, 6 per CU
    (100_000_000, STREAM, 26624),   # the headline configuration
    (125_000_000, STREAM, 26624),   # 1 G particles over 8 GPUs
    (1_000_000_000, STREAM, 26624),
])
def test_single_attractor_shape_by_size(count, access, reserve):
    s = psim.launch_shape(count)
    assert (s["access"], s["reserve_lds"], s["reserve_lds_colour"]) == (access, reserve, reserve)
    assert s["workgroup_size"] == 64                            # single-wave workgroups
    assert s["workgroups"] == max(1, -(-(count // 4) // 64))    # one pass over the quads
    assert s["resident_per_cu"] == (163840 // reserve if reserve else 0)


def test_attractor_lists():
    for n in (6_250_000, 100_000_000):
        assert psim.launch_shape(n, 1)["reserve_lds"] == 26624
    for a in (2, 3, 4):                                        # streaming: 13 workgroups per CU
        s = psim.launch_shape(100_000_000, a)
        assert s["reserve_lds"] == 11776 and s["resident_per_cu"] == 13
        assert psim.launch_shape(6_250_000, a)["reserve_lds"] == 0     # cache-resident: no cap
    for a in (5, 6, 8):                                        # streaming: 16 workgroups per CU
        s = psim.launch_shape(100_000_000, a)
        assert s["reserve_lds"] == 10240 and s["resident_per_cu"] == 16
        assert psim.launch_shape(6_250_000, a)["reserve_lds"] == 0
    for a in (9, 16, 64):                                      # longer lists: no cap anywhere
        assert psim.launch_shape(100_000_000, a)["reserve_lds"] == 0
        assert psim.launch_shape(6_250_000, a)["reserve_lds"] == 0
    assert psim.launch_shape(100_000, 8)["reserve_lds"] == 0


def test_edges():
    s = psim.launch_shape(0)
    assert s["workgroups"] == 1 and s["access"] == PLAIN
    assert psim.launch_shape(3)["workgroups"] == 1              # the tail alone
    assert psim.launch_shape(4_294_967_295)["workgroups"] == 16_777_216
    with pytest.raises(psim.PsimError):
        psim.launch_shape(1000, 65)
    assert psim.load_library().psim_launch_shape(10, 1, None) == 1
