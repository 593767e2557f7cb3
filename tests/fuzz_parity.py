#!/usr/bin/env python3
"""Longer seeded fuzz of the GPU path against the CPU oracle than the test-suite runs:
tests/random_cases.py cases (small states, wild values, attractor lists of 1-64) for several
seeds, each through two entry points (update / update_batch, update_multi /
update_multi_noalloc).  usage: fuzz_parity.py [cases_per_seed] [seed ...]"""
import copy
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
psim = importlib.import_module("particle-simulation-3d_amd")
import psim_testlib as T  # noqa: E402
import random_cases  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seeds = [int(s) for s in sys.argv[2:]] or [1, 2, 3]
sim = psim.HipParticleSimulation.new(1000, psim.GEN_UNIFORM)
for seed in seeds:
    lists = 0
    for i, particles, params, attractors, steps in random_cases.cases(count, seed=seed):
        sim.resize_buffer(len(particles), psim.GEN_UNIFORM)
        sim.upload_aos(particles)
        for _ in range(steps):
            if attractors is None:
                sim.update(params)
            else:
                sim.update_multi(params, attractors)
        want = T.oracle_update_aos(particles, params, steps, attractors)
        T.assert_particles_equal(sim.download_aos(), want, f"seed {seed} case {i}")
        # the same steps through the other entry points: one fused batch (single attractor),
        # the list step that allocates nothing (lists)
        sim.upload_aos(particles)
        if attractors is None:
            sim.update_batch([params] * steps)
            T.assert_particles_equal(sim.download_aos(), want, f"seed {seed} case {i}, fused batch")
            # ... and a batch whose steps differ: time step, colour mode, attractor on and off
            varied = []
            for j in range(steps + 3):
                q = copy.copy(params)
                q.delta_time = float(np.float32(params.delta_time) * np.float32(1.0 + 0.125 * j))
                q.color_mode = (params.color_mode + j) % 3
                q.is_mouse_dragging = (params.is_mouse_dragging + j) % 2
                varied.append(q)
            want = particles
            for q in varied:
                want = T.oracle_update_aos(want, q, 1)
            sim.upload_aos(particles)
            sim.update_batch(varied)
        else:
            for _ in range(steps):
                sim.update_multi_noalloc(params, attractors)
        T.assert_particles_equal(sim.download_aos(), want, f"seed {seed} case {i}, second entry point")
        lists += attractors is not None
    print(f"seed {seed}: {count} cases ({lists} with attractor lists) bit-identical", flush=True)
