"""GPU tests of the device-resident hand-off of the particle records (psim_shared_aos): the
replacement of the reference's `get_particle_buffer` (src/simulation/mod.rs:44) for a renderer
that draws from device memory, as the reference's compute backend lets its renderer do
(src/simulation/compute.rs:28-34, src/custom_renderer.rs:36-38).  No renderer here: a second
process imports the dma-buf descriptor as external memory and reads the frames."""
import os
import subprocess

import numpy as np
import pytest

import psim_testlib as T
from psim_testlib import psim

pytestmark = pytest.mark.gpu

CPP_DIR = os.path.join(T.ROOT, "tests", "cpp")
Sim = psim.HipParticleSimulation


@pytest.mark.parametrize("count,frames", [(0, 2), (1, 3), (70_001, 6), (2_000_000, 4)])
def test_another_process_reads_every_frame_through_one_mapping(count, frames):
    """tests/cpp/shared_records.cpp: producer and consumer processes (forked before either
    touches the GPU); the consumer maps the descriptor once and must find every exported frame
    there, byte for byte (hash of an ordinary download made by the producer)."""
    binary = os.path.join(CPP_DIR, "shared_records")
    if not os.access(binary, os.X_OK):
        subprocess.check_call(["make", "-C", CPP_DIR, "shared_records"], stdout=subprocess.DEVNULL)
    r = subprocess.run([binary, str(count), str(frames)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-500:] + r.stderr[-2000:]
    assert r.stdout.split()[:4] == ["producer", "0", "consumer", "0"]


def test_the_shared_buffer_holds_the_records_and_follows_a_resize():
    n = 50_001
    with Sim.new(n, psim.GEN_FILLED) as sim, Sim.new(n, psim.GEN_FILLED) as reader:
        ptr, fd, size = sim.shared_aos()
        try:
            assert ptr and fd >= 0 and size >= n * 64 and size % 4096 == 0
            assert os.fstat(fd).st_size in (0, size)     # a dma-buf: the kernel knows its size, or says 0
            ptr2, fd2, size2 = sim.shared_aos()          # the same memory, another descriptor
            assert (ptr2, size2) == (ptr, size) and fd2 != fd
            os.close(fd2)
            sim.update(T.bench_params(mouse_radius=40.0, color_mode=2))
            sim.export_aos_device(ptr)
            sim.sync()
            want = sim.download_aos()
            reader.import_aos_device(ptr)                # read back through another handle
            reader.sync()
            assert reader.download_aos().tobytes() == want.tobytes()
            # growing the set beyond the storage retires the buffer: a new one, large enough
            sim.resize_buffer(64 * n, psim.GEN_HOLLOW)
            with pytest.raises(psim.PsimError, match="do not fit the shared record buffer"):
                sim.export_aos_device(ptr)               # the old buffer: refused, not overrun
            ptr3, fd3, size3 = sim.shared_aos()
            os.close(fd3)
            assert size3 >= 64 * n * 64
            sim.export_aos_device(ptr3)
            sim.sync()
        finally:
            os.close(fd)


def test_arguments_are_checked():
    import ctypes
    lib = psim.load_library()
    with Sim.new(100, psim.GEN_UNIFORM) as sim:
        assert lib.psim_shared_aos(sim.handle, None, None, None) == 1
    ptr, fd, size = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_uint64()
    assert lib.psim_shared_aos(None, ctypes.byref(ptr), ctypes.byref(fd), ctypes.byref(size)) == 1
