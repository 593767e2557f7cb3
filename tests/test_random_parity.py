"""Seeded random sweeps (tests/random_cases.py).

CPU: the C oracle against the independent numpy-float32 restatement of the reference's
arithmetic (tests/gen_golden.py) -- two codings of src/simulation/cpu.rs:65-113 must agree
bit for bit on hundreds of random states and parameter sets, including subnormals,
infinities, NaNs and out-of-slider parameters.
GPU: the HIP path through the C ABI against the oracle on the same cases.
"""
import numpy as np
import pytest

import psim_testlib as T
import random_cases
from gen_golden import numpy_run
from psim_testlib import psim

CPU_CASES = 300
GPU_CASES = 200


def test_oracle_equals_numpy_restatement_on_random_cases():
    checked = 0
    with np.errstate(all="ignore"):
        for i, particles, params, attractors, steps in random_cases.cases(CPU_CASES):
            want = numpy_run(particles, params, steps, attractors)
            got = T.oracle_update_aos(particles, params, steps, attractors)
            T.assert_particles_equal(got, want, f"random case {i}")
            checked += 1
    assert checked == CPU_CASES


@pytest.mark.gpu
def test_gpu_equals_oracle_on_random_cases():
    Sim = psim.HipParticleSimulation
    with Sim.new(1000, psim.GEN_UNIFORM) as sim:
        for i, particles, params, attractors, steps in random_cases.cases(GPU_CASES, seed=7):
            sim.resize_buffer(len(particles), psim.GEN_UNIFORM)
            sim.upload_aos(particles)
            for _ in range(steps):
                if attractors is None:
                    sim.update(params)
                else:
                    sim.update_multi(params, attractors)
            got = sim.download_aos()
            want = T.oracle_update_aos(particles, params, steps, attractors)
            T.assert_particles_equal(got, want, f"random case {i}")
