#!/usr/bin/env python3
"""AoS export / import with 1, 2 or 4 particles per lane (override aos_ppl) by extra LDS
reservation (override aos_reserve, bytes of dynamic LDS on top of what the kernel declares: fewer
resident waves per CU).  Prints GB/s of the 128 B per particle moved; every variant's result
is first compared with the one-particle-per-lane kernels' (all 64 bytes of every record)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
psim = importlib.import_module("particle-simulation-3d_amd")


def rate(sim, fn, n):
    best = 1e9
    for _ in range(3):
        fn()
        sim.sync()
        sim.timer_start()
        for _ in range(10):
            fn()
        best = min(best, sim.timer_stop() / 10)
    return 128 * n / best / 1e6


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    sim = psim.HipParticleSimulation.new(n, psim.GEN_UNIFORM)
    scratch = psim.HipParticleSimulation.new(4 * n, psim.GEN_UNIFORM)
    buf = scratch.get_particle_buffer().color
    scratch.set_stream(sim.get_stream())
    sim.update(psim.SimParams(color_mode=1, is_mouse_dragging=1, mouse_radius=40.0))
    psim.debug_set_tuning("aos_ppl", 1)
    psim.debug_set_tuning("aos_reserve", -1)
    sim.export_aos_device(buf)
    sim.sync()
    want = sim.checksum()
    probe = slice(0, min(n, 300_000))
    want_head = sim.download_aos(0, probe.stop)
    print(f"particles={n}")
    sweep = ((1, ("",)), (2, ("0", "1792", "3840", "7936", "12032", "18176", "24320", "32512")),
             (4, ("0", "1920", "6016", "10112", "16256", "24448", "30592", "36736", "48000", "64000")))
    if len(sys.argv) > 2:                                 # e.g. "4:24448,30592 2:12032"
        sweep = tuple((int(a.split(":")[0]), tuple(a.split(":")[1].split(","))) for a in sys.argv[2:])
    for ppl, reserves in sweep:
        for r in reserves:
            psim.debug_set_tuning("aos_ppl", ppl)
            psim.debug_set_tuning("aos_reserve", int(r) if r else -1)
            # correctness: records out with this variant, state wiped, records back in
            sim.export_aos_device(buf)
            sim.reset(psim.GEN_HOLLOW if n <= 20_000_000 else psim.GEN_UNIFORM)
            sim.import_aos_device(buf)
            sim.sync()
            ok = sim.checksum() == want
            psim.debug_set_tuning("aos_ppl", 1)
            ok = ok and sim.download_aos(0, probe.stop).tobytes() == want_head.tobytes()
            psim.debug_set_tuning("aos_ppl", ppl)
            ex = rate(sim, lambda: sim.export_aos_device(buf), n)
            im = rate(sim, lambda: sim.import_aos_device(buf), n)
            print(f"particles per lane {ppl}  extra LDS {r or 'auto':>6}: export {ex:7.1f} GB/s  "
                  f"import {im:7.1f} GB/s  round trip {'identical' if ok else 'DIFFERS'}", flush=True)


if __name__ == "__main__":
    main()
