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
@pytest.mark.parametrize("dragging", [0, 1])
def test_long_run_stays_bit_identical(dragging):
    """12 000 steps: with damping 0.99 the velocities decay through the subnormal range
    (0.99^12000 ~ 4e-53) until rounding pins them at a few dozen units of 2^-149, so every
    step of the decay must round like the oracle; with the attractor held, particles near
    it keep moving instead."""
    n, steps = 100_000, 12_000
    params = T.bench_params(is_mouse_dragging=dragging, mouse_radius=30.0, mouse_force=20.0)
    with psim.HipParticleSimulation.new(n, psim.GEN_UNIFORM) as sim:
        for _ in range(steps):
            sim.update(params)
        got = sim.download_soa()
    want, _ = T.oracle_update_soa(T.oracle_gen_uniform_soa(0, n), params, steps)
    T.assert_same_f32(got, want, f"{steps} steps, dragging={dragging}")
    if not dragging:
        assert np.abs(want[3:]).max() < 1e-42        # the decay ended deep in the subnormals


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
