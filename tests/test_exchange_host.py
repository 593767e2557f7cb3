"""The host side of the attractor exchange (psim_exchange_*), checked without a GPU: argument
validation, the file rendezvous of the non-zero ranks, and that nothing but the exchange
maps RCCL."""
import ctypes
import importlib
import os
import subprocess
import sys
import threading
import time

import pytest

import psim_testlib as T

psim = importlib.import_module("particle-simulation-3d_amd")


def test_waiting_ranks_read_the_published_id(tmp_path):
    path = tmp_path / "id"
    ident = bytes(range(128))
    path.write_bytes(ident)
    assert psim.exchange_rendezvous_file(path, 1, timeout_ms=100) == ident
    assert psim.exchange_rendezvous_file(path, 7, timeout_ms=100) == ident


def test_waiting_rank_sees_an_id_that_appears_later_and_never_half_of_it(tmp_path):
    path = tmp_path / "id"
    ident = bytes(255 - k for k in range(128))

    def publish():
        time.sleep(0.2)
        (tmp_path / "id.tmp").write_bytes(ident)
        os.rename(tmp_path / "id.tmp", path)        # as rank 0 does: complete or absent

    t = threading.Thread(target=publish)
    t.start()
    try:
        assert psim.exchange_rendezvous_file(path, 3, timeout_ms=5000) == ident
    finally:
        t.join()


def test_a_missing_id_times_out_with_a_comm_error(tmp_path):
    t0 = time.perf_counter()
    with pytest.raises(psim.PsimError) as e:
        psim.exchange_rendezvous_file(tmp_path / "never", 2, timeout_ms=150)
    assert e.value.code == 6 and "rank 2" in str(e.value)
    assert 0.1 < time.perf_counter() - t0 < 5.0
    # a short file is not an id either
    (tmp_path / "short").write_bytes(b"x" * 100)
    with pytest.raises(psim.PsimError):
        psim.exchange_rendezvous_file(tmp_path / "short", 1, timeout_ms=50)


def test_argument_validation():
    lib = psim.load_library()
    buf = (ctypes.c_uint8 * 128)()
    x = ctypes.c_void_p()
    fake = ctypes.c_void_p(8)                            # never dereferenced by these calls
    assert lib.psim_exchange_unique_id(None) == 1
    assert lib.psim_exchange_rendezvous_file(None, 0, buf, 10) == 1
    assert lib.psim_exchange_rendezvous_file(b"/tmp/x", -1, buf, 10) == 1
    assert lib.psim_exchange_create(None, fake, buf, 0, 1) == 1
    assert lib.psim_exchange_create(ctypes.byref(x), None, buf, 0, 1) == 1
    assert lib.psim_exchange_create(ctypes.byref(x), fake, None, 0, 1) == 1
    assert lib.psim_exchange_create(ctypes.byref(x), fake, buf, 0, 0) == 1
    assert lib.psim_exchange_create(ctypes.byref(x), fake, buf, 0, 65) == 1
    assert b"at most 64" in lib.psim_last_error()
    assert lib.psim_exchange_create(ctypes.byref(x), fake, buf, 2, 2) == 1
    assert lib.psim_exchange_create(ctypes.byref(x), fake, buf, -1, 2) == 1
    assert x.value is None
    assert lib.psim_exchange_step(None, None, None) == 1
    assert lib.psim_exchange_sync(None) == 1
    assert lib.psim_exchange_set_timeout(None, 1000) == 1
    assert lib.psim_exchange_set_pacing(None, 0) == 1
    assert lib.psim_exchange_last_list(None, None) == 1
    assert lib.psim_exchange_world(None) == 0 and lib.psim_exchange_rank(None) == -1
    lib.psim_exchange_destroy(None)


@pytest.mark.skipif(T.gpu_available(), reason="the machine has a GPU")
def test_without_a_gpu_the_id_is_a_comm_error():
    with pytest.raises(psim.PsimError) as e:
        psim.exchange_unique_id()
    assert e.value.code == 6


def test_rccl_is_mapped_by_the_exchange_only():
    """librccl.so.1 is half a gigabyte: loading the library and calling entry points outside
    the exchange section must not map it; the first exchange call does."""
    code = (
        "import importlib, sys\n"
        f"sys.path.insert(0, {T.ROOT!r})\n"
        "psim = importlib.import_module('particle-simulation-3d_amd')\n"
        "psim.load_library(); psim.launch_shape(1000)\n"
        "maps = open('/proc/self/maps').read()\n"
        "assert 'librccl' not in maps, 'RCCL mapped by a plain load'\n"
        "try:\n    psim.exchange_unique_id()\nexcept psim.PsimError:\n    pass\n"
        "assert 'librccl' in open('/proc/self/maps').read(), 'the exchange did not load RCCL'\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
