#!/usr/bin/env python3
"""Sweeps the LDS reservation (= resident workgroups per CU) of the AoS export / import
kernels at 100 M particles; one child process per value because the override is read once
per process.  Prints GB/s of the 128 B per particle moved."""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(n):
    sys.path.insert(0, ROOT)
    psim = importlib.import_module("particle-simulation-3d_amd")
    sim = psim.HipParticleSimulation.new(n, psim.GEN_UNIFORM)
    scratch = psim.HipParticleSimulation.new(4 * n, psim.GEN_UNIFORM)
    buf = scratch.get_particle_buffer().color
    scratch.set_stream(sim.get_stream())
    out = []
    for fn in (lambda: sim.export_aos_device(buf), lambda: sim.import_aos_device(buf)):
        best = 1e9
        for _ in range(3):
            fn()
            sim.sync()
            sim.timer_start()
            for _ in range(10):
                fn()
            best = min(best, sim.timer_stop() / 10)
        out.append(128 * n / best / 1e6)
    print(f"block {os.environ.get('PSIM_AOS_BLOCK', 'auto'):>4} reserve {os.environ.get('PSIM_AOS_RESERVE', 'auto'):>6}: export {out[0]:7.1f} GB/s  "
          f"import {out[1]:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(int(sys.argv[2]))
    else:
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
        for r in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("", "0", "6144", "10240", "15872", "24064", "36864", "65024")):
            env = dict(os.environ, PSIM_EXPERIMENT="1")   # the switch that lets PSIM_* reach a launch
            env.pop("PSIM_AOS_RESERVE", None)
            if r:
                env["PSIM_AOS_RESERVE"] = r
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(n)], env=env,
                           check=True, timeout=120)
