#!/usr/bin/env python3
"""Workload for profiling every kernel of the path once each: one configuration per kernel
instantiation, a few launches of each, so that rocprofv3's per-kernel averages and counters
belong to exactly one configuration.  Prints the configuration table as JSON on stdout
(kernel-name fragment -> algorithmic bytes per launch), which scripts/summarize_ops_profile.py
joins with the rocprofv3 output."""
import argparse
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
psim = importlib.import_module("particle-simulation-3d_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=100_000_000)
    ap.add_argument("--launches", type=int, default=5)
    args = ap.parse_args()
    n = args.particles
    sim = psim.HipParticleSimulation.new(n, psim.GEN_UNIFORM)
    scratch = psim.HipParticleSimulation.new(max(4 * n, 4096), psim.GEN_UNIFORM)
    aos_buf = scratch.get_particle_buffer().color
    attr_buf = scratch.get_particle_buffer().padding
    scratch.sync()

    def P(**kw):
        base = dict(is_mouse_dragging=1, mouse_position=(0.0, 0.0, 48.0))
        base.update(kw)
        return psim.SimParams(**base)

    rng = np.random.default_rng(8)
    a8 = np.zeros(8, dtype=psim.ATTRACTOR_DTYPE)
    a8["position"] = rng.uniform(-40, 40, size=(8, 3)).astype(np.float32)
    a8["force"], a8["radius"], a8["active"] = 5.0, 10.0, 1
    for i in range(8):
        sim.store_attractor(attr_buf + 32 * i, a8[i:i + 1])
    batch = [P(delta_time=0.016 + 1e-4 * (i % 7)) for i in range(16)]

    # (kernel-name fragment as rocprofv3 prints it, what it is, algorithmic bytes, call)
    ops = [
        ("update_kernel<0, 1, 2>", "update, single attractor r=10, colour 0", 48 * n,
         lambda: sim.update(P())),
        ("update_kernel<0, 0, 2>", "update, no attractor", 48 * n,
         lambda: sim.update(P(is_mouse_dragging=0))),
        ("update_kernel<1, 1, 2>", "update, r=50, colour mode 1", 64 * n,
         lambda: sim.update(P(mouse_radius=50.0, color_mode=1))),
        ("update_kernel<2, 1, 2>", "update, r=50, colour mode 2", 64 * n,
         lambda: sim.update(P(mouse_radius=50.0, color_mode=2))),
        ("update_kernel<0, 4, 2>", "update, 8 attractors r=10 through the pair queue", 48 * n,
         lambda: sim.update_multi_device(P(), attr_buf, 8)),
        ("update_batch_kernel<2>", "16 fused steps [bound by instruction issue]", 48 * n,
         lambda: sim.update_batch(batch)),
        ("export_aos_kernel<true, 256>", "SoA -> 64-byte records", 128 * n,
         lambda: sim.export_aos_device(aos_buf)),
        ("import_aos_wide_kernel<true, 2>", "64-byte records -> SoA (two particles per lane)", 128 * n,
         lambda: sim.import_aos_device(aos_buf)),
        ("gen_uniform_kernel<true>", "synthetic state generator", 64 * n,
         lambda: sim.reset(psim.GEN_UNIFORM)),
        ("checksum_kernel", "order-independent checksum", 24 * n,
         lambda: sim.checksum()),
    ]
    for _, _, _, fn in ops:
        for _ in range(args.launches):
            fn()
        sim.sync()
    json.dump({"particles": n, "launches": args.launches,
               "kernels": [{"name": k, "what": w, "algorithmic_bytes": b,
                            "bound": "valu" if "instruction issue" in w else "hbm"}
                           for k, w, b, _ in ops]},
              sys.stdout)
    print()


if __name__ == "__main__":
    main()
