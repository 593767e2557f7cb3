#!/usr/bin/env python3
"""Host-side cost of one psim_update call at launch-bound sizes: wall time per step of a
tight loop of updates (ctypes caller), against psim_update_batch for the same steps."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
psim = importlib.import_module("particle-simulation-3d_amd")

params = psim.SimParams(delta_time=0.016, gravity=0.0, color_mode=0, mouse_force=5.0,
                        mouse_radius=10.0, is_mouse_dragging=1, damping=0.99,
                        max_dist_for_color=50.0, mouse_position=(0.0, 0.0, 48.0))
steps = 2048
print(f"{'particles':>10} | {'update us/step':>14} | {'device us/step':>14} | {'batch us/step':>13}")
for n in (1_000, 10_000, 100_000, 1_000_000, 4_000_000):
    sim = psim.HipParticleSimulation.new(n, psim.GEN_UNIFORM)
    for _ in range(64):
        sim.update(params)
    sim.sync()
    t0 = time.perf_counter()
    sim.timer_start()
    for _ in range(steps):
        sim.update(params)
    dev = sim.timer_stop()
    wall = time.perf_counter() - t0
    batch = [params] * steps
    sim.update_batch(batch)
    sim.sync()
    t0 = time.perf_counter()
    sim.update_batch(batch)
    sim.sync()
    wb = time.perf_counter() - t0
    print(f"{n:>10} | {wall / steps * 1e6:14.2f} | {dev / steps * 1e3:14.2f} | {wb / steps * 1e6:13.2f}",
          flush=True)
    sim.close()
