#!/usr/bin/env python3
"""Frame rate of the staged coupling to a renderer (update, then the 64-byte records to host
memory every frame, as the reference's CPU backend hands them to wgpu,
src/simulation/cpu.rs:116-120) at the application's sizes: pageable destination,
page-locked destination, and page-locked with the download of frame k enqueued before the
host waits (psim_download_aos_async + psim_sync)."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
psim = importlib.import_module("particle-simulation-3d_amd")


def main():
    params = psim.SimParams(is_mouse_dragging=1, mouse_position=(0.0, 0.0, 48.0))
    frames = 60
    print(f"{'particles':>10} | {'variant':<34} | {'ms/frame':>9} | {'frames/s':>9} | GB/s to host")
    for n in (100_000, 1_000_000, 10_000_000):
        sim = psim.HipParticleSimulation.new(n, psim.GEN_HOLLOW)
        pinned = psim.pinned_empty(n)
        pageable = None

        def run(body):
            body()
            sim.sync()
            t0 = time.perf_counter()
            for _ in range(frames):
                body()
            sim.sync()
            return (time.perf_counter() - t0) / frames

        def pageable_frame():
            nonlocal pageable
            sim.update(params)
            pageable = sim.download_aos()

        def pinned_frame():
            sim.update(params)
            sim.download_aos(out=pinned)

        def async_frame():
            sim.update(params)
            sim.download_aos_async(pinned)
            sim.sync()

        for name, body in (("pageable destination", pageable_frame),
                           ("page-locked destination", pinned_frame),
                           ("page-locked, enqueue then one sync", async_frame),
                           ("update only (state stays in HBM)", lambda: sim.update(params))):
            dt = run(body)
            moved = 0.0 if name.startswith("update only") else 64 * n / dt / 1e9
            print(f"{n:>10} | {name:<34} | {dt * 1e3:9.4f} | {1 / dt:9.0f} | {moved:6.1f}", flush=True)
        sim.close()


if __name__ == "__main__":
    main()
