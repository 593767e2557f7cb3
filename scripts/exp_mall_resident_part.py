#!/usr/bin/env python3
"""Can a part of a streaming state live in the 256 MiB memory-side cache from step to step?

100 M particles stream through HBM every step (traffic = 1.0000 x the algorithmic bytes).  If the
first K particles were stepped with cache-resident accesses (plain loads, sc1 stores) and the
rest with the streaming ones (nt loads, nt sc1 stores), and the streaming part did not evict
them, their reads -- K x 24 B per step -- would come from the cache.  Two handles stand in for
the two parts of one state: A with K particles (the library picks the resident mode up to
11 M), B with the rest; a step updates both on one stream.  Printed: time per step against
the single 100 M handle's.
"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
psim = importlib.import_module("particle-simulation-3d_amd")


def timed(steps, fn, sim):
    for _ in range(10):
        fn()
    sim.sync()
    best = 1e9
    for _ in range(3):
        sim.timer_start()
        for _ in range(steps):
            fn()
        best = min(best, sim.timer_stop() / steps)
    return best


def main():
    total = 100_000_000
    params = psim.SimParams(is_mouse_dragging=1, mouse_position=(0.0, 0.0, 48.0))
    with psim.HipParticleSimulation.new(total, psim.GEN_UNIFORM) as whole:
        base = timed(50, lambda: whole.update(params), whole)
    print(f"one handle, {total} particles: {base * 1e3:.1f} us per step")
    for k in (4_000_000, 6_000_000, 8_000_000, 10_000_000):
        with psim.HipParticleSimulation.new(k, psim.GEN_UNIFORM) as a, \
                psim.HipParticleSimulation.new(total - k, psim.GEN_UNIFORM) as b:
            b.set_stream(a.get_stream())
            shape = psim.launch_shape(k, 1)

            def both():
                a.update(params)
                b.update(params)
            t = timed(50, both, a)
            ta = timed(50, lambda: a.update(params), a)
            tb = timed(50, lambda: b.update(params), a)
            b.set_stream(0)
        print(f"{k} resident (access mode {shape['access']}) + {total - k} streaming: "
              f"{t * 1e3:.1f} us per step ({(t / base - 1) * 100:+.1f} %); alone: {ta * 1e3:.1f} + {tb * 1e3:.1f} us",
              flush=True)


if __name__ == "__main__":
    main()
