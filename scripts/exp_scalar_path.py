#!/usr/bin/env python3
"""Rate of the stateless entry point psim_update_soa on caller-owned arrays: 16-byte
aligned (the vector kernel) against arrays that start 4 bytes off (the 4-byte fallback
kernel).  Prints ms per launch and GB/s of the 48 B per particle."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
psim = importlib.import_module("particle-simulation-3d_amd")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    owner = psim.HipParticleSimulation.new(n + 4, psim.GEN_UNIFORM)
    v = owner.get_particle_buffer()
    params = psim.SimParams(is_mouse_dragging=1, mouse_position=(0.0, 0.0, 48.0))
    stream = owner.get_stream()
    for name, off in (("aligned (vector kernel)", 0), ("16 bytes off (vector kernel)", 16),
                      ("64 bytes off (vector kernel)", 64), ("4 bytes off (scalar kernel)", 4)):
        arrays = [p + off for p in (v.px, v.py, v.pz, v.vx, v.vy, v.vz)]
        best = 1e9
        for _ in range(3):
            psim.update_soa(*arrays, 0, n, params, stream)
            owner.sync()
            owner.timer_start()
            for _ in range(10):
                psim.update_soa(*arrays, 0, n, params, stream)
            best = min(best, owner.timer_stop() / 10)
        print(f"{name:<30}: {best:.4f} ms  {48 * n / best / 1e6:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
