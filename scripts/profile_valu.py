#!/usr/bin/env python3
"""One VALU-bound operation of the path, launched a few times, for a rocprofv3 counter pass
(scripts/profile_valu.sh): fused batches of 16 / 64 steps, the 64-attractor list, and the
single step for comparison.  Prints what ran as JSON."""
import argparse
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
psim = importlib.import_module("particle-simulation-3d_amd")

OPS = ("single", "batch16", "batch64", "list64", "list8")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--op", choices=OPS, required=True)
    ap.add_argument("--particles", type=int, default=100_000_000)
    ap.add_argument("--launches", type=int, default=3)
    ap.add_argument("--radius", type=float, default=10.0)
    args = ap.parse_args()
    n = args.particles
    sim = psim.HipParticleSimulation.new(n, psim.GEN_UNIFORM)

    def P(**kw):
        base = dict(is_mouse_dragging=1, mouse_position=(0.0, 0.0, 48.0), mouse_radius=args.radius)
        base.update(kw)
        return psim.SimParams(**base)

    steps = 1
    if args.op == "single":
        fn, kernel = (lambda: sim.update(P())), "update_kernel<0, 1, 2>"
    elif args.op.startswith("batch"):
        steps = int(args.op[5:])
        batch = [P(delta_time=0.016 + 1e-4 * (i % 7)) for i in range(steps)]
        fn, kernel = (lambda: sim.update_batch(batch)), "update_batch_kernel<2>"
    else:
        k = int(args.op[4:])
        rng = np.random.default_rng(8)
        a = np.zeros(k, dtype=psim.ATTRACTOR_DTYPE)
        a["position"] = rng.uniform(-40, 40, size=(k, 3)).astype(np.float32)
        a["force"], a["radius"], a["active"] = 5.0, args.radius, 1
        fn, kernel = (lambda: sim.update_multi(P(), a)), "update_kernel<0, 4, 2>"
    for _ in range(args.launches):
        fn()
    sim.sync()
    sim.timer_start()
    for _ in range(args.launches):
        fn()
    ms = sim.timer_stop() / args.launches
    json.dump({"op": args.op, "kernel": kernel, "particles": n, "steps_per_launch": steps,
               "attractors": int(args.op[4:]) if args.op.startswith("list") else 1,
               "launches": 2 * args.launches, "hip_event_ms_per_launch": ms}, sys.stdout)
    print()


if __name__ == "__main__":
    main()
