#!/usr/bin/env python3
"""What creating a simulation costs (`ParticleSimulation::new`, reference
src/simulation/mod.rs:22-29 -> generate_initial_particles, mod.rs:145-192): the first handle of
a process (runtime start-up and code-object load included), then, per generator and particle
count, a fresh handle and a reset.  The reference's README lists start-up time as an open
item; here the two sphere generators run on the host (their sin / cos / cbrt / acos must be
the host library's for identical particles) and travel to the device, the synthetic uniform
state is generated on the device."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
psim = importlib.import_module("particle-simulation-3d_amd")
psim.load_library()
t1 = time.perf_counter()
sim = psim.HipParticleSimulation.new(1_000_000, psim.GEN_HOLLOW)
sim.sync()
t2 = time.perf_counter()
params = psim.SimParams(is_mouse_dragging=1, mouse_position=(0.0, 0.0, 48.0))
sim.update(params)
sim.sync()
t3 = time.perf_counter()
sim.close()
print(f"library load {1e3 * (t1 - t0):.1f} ms | first handle of the process (1 M particles, hollow "
      f"sphere; runtime start-up included) {1e3 * (t2 - t1):.1f} ms | first update {1e3 * (t3 - t2):.2f} ms")
print("particles   generator   new (ms)   reset (ms)")
for n in (100_000, 1_000_000, 10_000_000, 100_000_000):
    for name, mode in (("hollow", psim.GEN_HOLLOW), ("filled", psim.GEN_FILLED),
                       ("uniform", psim.GEN_UNIFORM)):
        best_new, best_reset = 1e9, 1e9
        for _ in range(3):
            a = time.perf_counter()
            s = psim.HipParticleSimulation.new(n, mode)
            s.sync()
            b = time.perf_counter()
            s.reset(mode)
            s.sync()
            c = time.perf_counter()
            s.close()
            best_new, best_reset = min(best_new, b - a), min(best_reset, c - b)
        print(f"{n:>11} {name:>9} {1e3 * best_new:10.2f} {1e3 * best_reset:10.2f}", flush=True)
