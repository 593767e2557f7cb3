"""CPU tests of the boundary: the C-ABI library loads and exports what include/psim.h
declares.  No compute calls are made (there is no GPU here, and no CPU path)."""
import ctypes
import os
import re

import pytest

import psim_testlib as T
from psim_testlib import psim

HEADER = os.path.join(T.ROOT, "include", "psim.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)          # drop comments
    return sorted(set(re.findall(r"\b(psim_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_trait_surface():
    names = declared_symbols()
    # one entry point per method of `trait ParticleSimulation` (mod.rs:21-50)
    for required in ("psim_create", "psim_update", "psim_resize", "psim_device_soa",
                     "psim_method", "psim_particle_count", "psim_reset", "psim_is_paused",
                     "psim_set_paused"):
        assert required in names


def test_library_exports_every_declared_symbol():
    lib = psim.load_library()
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} is declared in psim.h but not exported"


def test_python_binding_covers_every_declared_symbol():
    assert sorted(psim.ABI) == declared_symbols()


def test_header_is_plain_c_and_links(tmp_path):
    """include/psim.h compiles as C99 (and as C++ through psim.hpp; the replay helper tests/cpp/psim_caller.hpp on top of it), the
    record sizes are the reference's (mod.rs:52-67, 86-98), and a C program links against the
    library with nothing but the header."""
    import subprocess
    src = tmp_path / "abi_check.c"
    src.write_text('''
#include <stddef.h>
#include <stdio.h>
#include "psim.h"
_Static_assert(sizeof(psim_sim_params) == 48, "SimParams");
_Static_assert(offsetof(psim_sim_params, mouse_position) == 32, "SimParams");
_Static_assert(sizeof(psim_particle) == 64, "Particle");
_Static_assert(offsetof(psim_particle, initial_color) == 48, "Particle");
_Static_assert(sizeof(psim_attractor) == 32, "attractor");
int main(void) {
    psim_handle *h = NULL;
    int rc = psim_create(&h, 16, PSIM_GEN_UNIFORM, 0);
    printf("%s|%d|%s\\n", psim_version(), rc, rc ? psim_last_error() : "");
    if (h) psim_destroy(h);
    return 0;
}
''')
    libdir = os.path.dirname(psim.LIBRARY_PATH)
    exe = tmp_path / "abi_check"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror",
                           "-I", os.path.join(T.ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lpsim_hip", f"-Wl,-rpath,{libdir}",
                           "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe)], text=True)
    version, rc, message = out.strip().split("|")
    assert "gfx950" in version
    if not T.gpu_available():
        assert int(rc) == 4 and "no CPU path" in message
    cpp = tmp_path / "hpp_check.cpp"
    cpp.write_text('#include "psim.hpp"\n#include "psim_caller.hpp"\n'
                   'int main() { psim::CallerState s; psim::FrameInput in{}; in.screen_width = 1;'
                   ' in.screen_height = 1; return s.frame(in).damping == 0.99f ? 0 : 1; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                           "-I", os.path.join(T.ROOT, "include"),
                           "-I", os.path.join(T.ROOT, "tests", "cpp"), str(cpp)])


def test_library_exports_nothing_else_public():
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", psim.LIBRARY_PATH], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    extra = {s for s in exported if not s.startswith("psim_") and not s.startswith("_")}
    assert not extra, f"unexpected exported symbols: {sorted(extra)}"
    assert {s for s in exported if s.startswith("psim_")} == set(declared_symbols())


def test_library_does_not_link_the_oracle():
    import subprocess
    out = subprocess.check_output(["readelf", "-d", psim.LIBRARY_PATH], text=True)
    assert "oracle" not in out
    sym = subprocess.check_output(["nm", "-D", psim.LIBRARY_PATH], text=True)
    assert "psim_oracle" not in sym


def test_version_and_error_strings():
    lib = psim.load_library()
    assert b"gfx950" in lib.psim_version()
    assert isinstance(lib.psim_last_error(), bytes)


def test_null_arguments_are_rejected_without_a_device():
    lib = psim.load_library()
    assert lib.psim_create(None, 10, 0, 0) == 1                 # PSIM_ERR_INVALID
    assert b"null" in lib.psim_last_error()
    h = ctypes.c_void_p()
    assert lib.psim_create(ctypes.byref(h), 10, 99, 0) == 1     # unknown generation mode
    assert lib.psim_create(ctypes.byref(h), 0xFFFFFFFF, 2, 0) == 1  # beyond PSIM_MAX_PARTICLES
    assert b"exceed" in lib.psim_last_error()
    assert lib.psim_update(None, ctypes.byref(psim.SimParams())) == 1
    assert lib.psim_particle_count(None) == 0
    assert lib.psim_is_paused(None) == 0
    lib.psim_destroy(None)                                      # no-op
    assert lib.psim_host_alloc(None, 64) == 1
    lib.psim_host_free(None)                                    # no-op
    assert lib.psim_set_color_semantics(None, 1) == 1
    assert lib.psim_color_semantics(None) == 0
    assert lib.psim_set_resize_semantics(None, 1) == 1
    assert lib.psim_resize_semantics(None) == 0
    assert lib.psim_download_aos_async(None, None, 0, 0) == 1
    assert lib.psim_download_aos_wait(None, None) == 1
    assert lib.psim_shared_aos(None, None, None, None) == 1
    g = ctypes.c_void_p()
    ids = (ctypes.c_int * 2)(0, 0)
    assert lib.psim_group_create(None, 10, ids, 2, 2) == 1
    assert lib.psim_group_create(ctypes.byref(g), 10, None, 2, 2) == 1
    assert lib.psim_group_create(ctypes.byref(g), 10, ids, 0, 2) == 1
    assert lib.psim_group_create(ctypes.byref(g), 2**40, ids, 2, 2) == 1     # too many per shard
    assert b"per shard" in lib.psim_last_error()
    assert lib.psim_group_update(None, ctypes.byref(psim.SimParams())) == 1
    assert lib.psim_group_particle_count(None) == 0 and lib.psim_group_shards(None) == 0
    assert lib.psim_group_shard(None, 0) is None
    lib.psim_group_destroy(None)                                # no-op


def test_tuning_overrides_are_named_and_checked():
    """psim_debug_set_tuning is the one way to change a launch shape; it needs no device, knows
    its names and refuses others.  (That the environment alone changes nothing is a GPU test:
    tests/test_gpu_launch_shapes.py.)"""
    lib = psim.load_library()
    for name in ("nt", "reserve_lds", "max_blocks", "skew_bytes", "queue_from", "aos_stream",
                 "aos_reserve", "aos_block", "aos_ppl", "checksum_blocks", "exchange_event_fences",
                 "exchange_comm_priority"):
        assert lib.psim_debug_set_tuning(name.encode(), 1) == 0, name
        assert lib.psim_debug_set_tuning(name.encode(), -1) == 0, name
    assert lib.psim_debug_set_tuning(b"PSIM_NT", 1) == 1
    assert b"unknown tuning override" in lib.psim_last_error()
    assert lib.psim_debug_set_tuning(None, 1) == 1
    with pytest.raises(psim.PsimError):
        psim.debug_set_tuning("no_such_override", 0)
    # the automatic launch shape does not move with an override: psim_launch_shape reports the
    # library's own choice
    before = psim.launch_shape(100_000_000, 1)
    psim.debug_set_tuning("reserve_lds", 4096)
    try:
        assert psim.launch_shape(100_000_000, 1) == before
    finally:
        psim.debug_set_tuning("reserve_lds", -1)


@pytest.mark.skipif(T.gpu_available(), reason="a GPU is present")
def test_page_locked_memory_needs_a_device():
    with pytest.raises(psim.PsimError) as e:
        psim.pinned_empty(16)
    assert e.value.code == 4


@pytest.mark.skipif(T.gpu_available(), reason="a GPU is present")
def test_create_fails_loudly_without_a_gpu():
    """No HIP device -> PSIM_ERR_NO_DEVICE; nothing falls back to the CPU."""
    with pytest.raises(psim.PsimError) as e:
        psim.HipParticleSimulation.new(1000, psim.GEN_UNIFORM)
    assert e.value.code == 4
    assert "no CPU path" in str(e.value)


def test_missing_library_is_an_error(monkeypatch):
    monkeypatch.setattr(psim, "_lib", None)
    monkeypatch.setattr(psim, "LIBRARY_PATH", "/nonexistent/libpsim_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        psim.load_library()
