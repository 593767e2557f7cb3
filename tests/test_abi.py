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
    assert lib.psim_update(None, ctypes.byref(psim.SimParams())) == 1
    assert lib.psim_particle_count(None) == 0
    assert lib.psim_is_paused(None) == 0
    lib.psim_destroy(None)                                      # no-op


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
