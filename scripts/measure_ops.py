#!/usr/bin/env python3
"""Times every kernel of the path at one particle count with the automatic launch shape:
update (single attractor, colour modes, all-in-range), multi-attractor lists of several
lengths (host list in kernel arguments, device list staged in LDS), AoS export/import,
the generator and the checksum.  Prints ms per launch and GB/s of algorithmic bytes."""
import argparse
import importlib
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
psim = importlib.import_module("particle-simulation-3d_amd")


def attractors(k, radius):
    rng = np.random.default_rng(k)
    a = np.zeros(k, dtype=psim.ATTRACTOR_DTYPE)
    a["position"] = rng.uniform(-40, 40, size=(k, 3)).astype(np.float32)
    a["force"] = 5.0
    a["radius"] = radius
    a["active"] = 1
    return a


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=100_000_000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    n = args.particles
    sim = psim.HipParticleSimulation.new(n, psim.GEN_UNIFORM)
    scratch = psim.HipParticleSimulation.new(max(4 * n, 4096), psim.GEN_UNIFORM)
    aos_buf = scratch.get_particle_buffer().color            # >= 64 B x n of device memory
    attr_buf = scratch.get_particle_buffer().padding
    scratch.set_stream(sim.get_stream())

    def P(**kw):
        base = dict(is_mouse_dragging=1, mouse_position=(0.0, 0.0, 48.0))
        base.update(kw)
        return psim.SimParams(**base)

    ops = []                                                   # (name, bytes/particle, fn)
    ops.append(("update single r=10 colour 0", 48, lambda: sim.update(P())))
    ops.append(("update single r=50 colour 0 (all in range)", 48,
                lambda: sim.update(P(mouse_radius=50.0))))
    ops.append(("update no attractor", 48, lambda: sim.update(P(is_mouse_dragging=0))))
    ops.append(("update single r=50 colour 1", 64,
                lambda: sim.update(P(mouse_radius=50.0, color_mode=1))))
    ops.append(("update single r=50 colour 2", 64,
                lambda: sim.update(P(mouse_radius=50.0, color_mode=2))))
    for k in (1, 2, 4, 8):
        a = attractors(k, 10.0)
        ops.append((f"update multi host list A={k} r=10", 48,
                    lambda a=a: sim.update_multi(P(), a)))
    a64 = attractors(64, 10.0)
    for i in range(64):
        sim.store_attractor(attr_buf + 32 * i, a64[i:i + 1])
    for k in (8, 64):
        ops.append((f"update multi device list (LDS) A={k} r=10", 48,
                    lambda k=k: sim.update_multi_device(P(), attr_buf, k)))
    ops.append(("export AoS (device)", 128, lambda: sim.export_aos_device(aos_buf)))
    ops.append(("import AoS (device)", 128, lambda: sim.import_aos_device(aos_buf)))
    ops.append(("generate uniform", 64, lambda: sim.reset(psim.GEN_UNIFORM)))
    ops.append(("checksum", 24, lambda: sim.checksum()))
    sim.sync()

    print(f"particles={n} steps={args.steps} rounds={args.rounds}")
    print(f"{'operation':<48} | {'ms':>9} | {'GB/s':>9} | {'Gparticles/s':>12}")
    for name, bpp, fn in ops:
        sync_each = name in ("generate uniform", "checksum")
        times = []
        for _ in range(args.rounds):
            fn()
            sim.sync()
            sim.timer_start()
            for _ in range(args.steps):
                fn()
            times.append(sim.timer_stop() / args.steps)
        med = statistics.median(times)
        print(f"{name:<48} | {med:9.4f} | {bpp * n / med / 1e6:9.1f} | {n / med / 1e6:12.2f}"
              + ("   (host-synchronous call)" if sync_each else ""), flush=True)


    # fused steps: k parameter sets per pass over the state
    for k in (2, 4, 8, 16, 64, 256):
        stream = [P(delta_time=0.016 + 1e-4 * (i % 7)) for i in range(k)]
        times = []
        for _ in range(args.rounds):
            sim.update_batch(stream)
            sim.sync()
            sim.timer_start()
            for _ in range(max(1, args.steps // 4)):
                sim.update_batch(stream)
            times.append(sim.timer_stop() / max(1, args.steps // 4))
        med = statistics.median(times)
        print(f"{'update batch of %d steps (fused passes)' % k:<48} | {med:9.4f} | "
              f"{'':>9} | {n * k / med / 1e6:12.2f}   ({med / k * 1e3:.2f} us per step)", flush=True)

    # host transfers of the 64-byte records (device transposition + PCIe copy, synchronous)
    import time
    m = min(n, 10_000_000)
    for name, fn in (("download AoS to host", lambda: sim.download_aos(0, m)),):
        fn()
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        print(f"{name:<48} | {dt * 1e3:9.3f} | {64 * m / dt / 1e9:9.2f} | {m / dt / 1e9:12.3f}"
              f"   ({m} particles, pageable host memory)")
    t0 = time.perf_counter()
    sim.upload_aos(out, 0)
    dt = time.perf_counter() - t0
    print(f"{'upload AoS from host':<48} | {dt * 1e3:9.3f} | {64 * m / dt / 1e9:9.2f} | "
          f"{m / dt / 1e9:12.3f}   ({m} particles, pageable host memory)")

    pinned = psim.pinned_empty(m)
    for name, fn in (("download AoS to host", lambda: sim.download_aos(0, m, out=pinned)),
                     ("upload AoS from host", lambda: sim.upload_aos(pinned, 0))):
        fn()
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        print(f"{name:<48} | {dt * 1e3:9.3f} | {64 * m / dt / 1e9:9.2f} | {m / dt / 1e9:12.3f}"
              f"   ({m} particles, page-locked host memory)")
    del pinned

    # the reference's two initial states (reset regenerates every particle)
    for name, mode in (("reset hollow sphere", psim.GEN_HOLLOW), ("reset filled sphere", psim.GEN_FILLED)):
        sim.reset(mode)
        sim.sync()
        t0 = time.perf_counter()
        sim.reset(mode)
        sim.sync()
        dt = time.perf_counter() - t0
        print(f"{name:<48} | {dt * 1e3:9.3f} | {64 * n / dt / 1e9:9.2f} | {n / dt / 1e9:12.3f}"
              f"   ({n} particles)")


if __name__ == "__main__":
    main()
