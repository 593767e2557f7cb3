#!/usr/bin/env python3
"""Times the update kernel under different launch shapes (psim_debug_set_tuning overrides are
read when a handle is created).  Variants run interleaved, several rounds, in one
process; prints median / best achieved GB/s (48 B per particle-update)."""
import argparse
import importlib
import itertools
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
psim = importlib.import_module("particle-simulation-3d_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=100_000_000)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--nt", default="-1,0,1")
    ap.add_argument("--unroll", default="1")
    ap.add_argument("--blocks", default="-1")
    ap.add_argument("--skew", default="0")
    ap.add_argument("--lds", default="-1")
    ap.add_argument("--radius", type=float, default=10.0)
    ap.add_argument("--color", type=int, default=0)
    args = ap.parse_args()
    n = args.particles
    params = psim.SimParams(is_mouse_dragging=1, mouse_position=(0.0, 0.0, 48.0),
                            mouse_radius=args.radius, color_mode=args.color)
    variants = list(itertools.product(args.nt.split(","), args.unroll.split(","),
                                      args.blocks.split(","), args.skew.split(","),
                                      args.lds.split(",")))
    sims = {}
    for v in variants:
        for name, value in (("nt", v[0]), ("max_blocks", v[2]), ("skew_bytes", v[3]),
                            ("reserve_lds", v[4])):
            psim.debug_set_tuning(name, int(value))      # copied when the handle is created
        sims[v] = psim.HipParticleSimulation.new(n, psim.GEN_UNIFORM)
        for _ in range(3):
            sims[v].update(params)
        sims[v].sync()
    times = {v: [] for v in variants}
    for r in range(args.rounds):
        for v in variants:
            s = sims[v]
            s.timer_start()
            for _ in range(args.steps):
                s.update(params)
            times[v].append(s.timer_stop() / args.steps)
        print(f"round {r} done", flush=True)
    bytes_per = (48 if args.color == 0 else 64) * n
    print(f"particles={n} steps={args.steps} rounds={args.rounds} radius={args.radius} "
          f"color={args.color}")
    print("nt unroll blocks skew   lds | median ms | median GB/s | best GB/s")
    for v in sorted(variants, key=lambda v: statistics.median(times[v])):
        med = statistics.median(times[v])
        best = min(times[v])
        print(f"{v[0]:>2} {v[1]:>6} {v[2]:>6} {v[3]:>5} {v[4]:>5} | {med:9.4f} | {bytes_per / med / 1e6:11.1f} | "
              f"{bytes_per / best / 1e6:9.1f}")


if __name__ == "__main__":
    main()
