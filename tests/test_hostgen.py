"""CPU test of the library's host-side generators (csrc/psim_hostgen.h, the code behind
psim_create / psim_reset for the reference's two initial states, src/simulation/mod.rs:145-192):
tests/cpp/hostgen_check.cpp is built from that header with a plain host compiler under
AddressSanitizer + UBSan, checks the GF(2) jump-ahead of the random stream against drawing
one by one, and writes the states it generated; here they are compared with the oracle."""
import os
import subprocess

import numpy as np

import psim_testlib as T
from psim_testlib import psim

CPP_DIR = os.path.join(T.ROOT, "tests", "cpp")
BINARY = os.path.join(CPP_DIR, "hostgen_check")


def test_host_generators_and_jump_ahead(tmp_path):
    subprocess.check_call(["make", "-C", CPP_DIR, "hostgen_check"], stdout=subprocess.DEVNULL)
    r = subprocess.run([BINARY, str(tmp_path)], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "hostgen ok" in r.stdout
    n = 300001
    hollow = np.fromfile(tmp_path / "hollow.bin", dtype=psim.PARTICLE_DTYPE)
    filled = np.fromfile(tmp_path / "filled.bin", dtype=psim.PARTICLE_DTYPE)
    far = np.fromfile(tmp_path / "filled_far.bin", dtype=psim.PARTICLE_DTYPE)
    T.assert_particles_equal(hollow, T.oracle_gen_hollow_aos(n), "library hollow state")
    T.assert_particles_equal(filled, T.oracle_gen_filled_aos(0, n), "library filled state")
    T.assert_particles_equal(far, T.oracle_gen_filled_aos(200_000_000, 1000),
                             "library filled state, 6e8 draws into the stream")
