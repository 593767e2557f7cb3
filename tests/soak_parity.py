#!/usr/bin/env python3
"""Full-size parity soak: the benchmark state (default 100 M particles) is stepped through a
scripted parameter stream on the GPU (single updates and fused batches, through the C ABI)
and by the CPU oracle; device checksums are compared with the oracle's every `--every` steps
and the whole state value by value at the end.  The bar is 1e-5 relative; the comparison is
for bit equality and reports the largest relative error if that fails.

The oracle is the checker here, never the thing measured."""
import argparse
import ctypes
import importlib
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
psim = importlib.import_module("particle-simulation-3d_amd")
import psim_testlib as T  # noqa: E402


def stream(steps):
    """A drag across the cloud with changing frame times, gravity and attractor settings."""
    out = []
    for i in range(steps):
        ang = 2 * math.pi * i / 240.0
        out.append(psim.SimParams(
            delta_time=(0.016, 0.0175, 0.01525, 0.033)[i % 4],
            gravity=0.0 if i < steps // 3 else 2.5,
            color_mode=0,
            mouse_force=5.0 if (i // 64) % 2 == 0 else 60.0,
            mouse_radius=10.0 if (i // 96) % 2 == 0 else 35.0,
            is_mouse_dragging=0 if (i // 40) % 5 == 4 else 1,
            damping=0.99,
            max_dist_for_color=50.0,
            mouse_position=(30.0 * math.cos(ang), 30.0 * math.sin(ang), 48.0 - 0.2 * i)))
    return out


def mixed_soak(n, steps, every, n_list=3):
    """Every feature of the path in one run, on the reference's 64-byte records: colour modes
    cycling, an attractor list (host list, 3 entries unless --attractors says otherwise) every seventh step, fused batches on alternate
    chunks, compute-shader colour semantics switched on for the second half."""
    lib = T.load_oracle()
    threads = int(lib.psim_oracle_max_threads())
    params = stream(steps)
    for i, p in enumerate(params):
        p.color_mode = (0, 1, 2, 0, 0, 1, 7)[(i // 5) % 7]
    rng = np.random.default_rng(3)
    lst = np.zeros(n_list, dtype=psim.ATTRACTOR_DTYPE)
    lst["position"] = rng.uniform(-40, 40, size=(n_list, 3)).astype(np.float32)
    lst["force"], lst["radius"], lst["active"] = 20.0, 15.0, 1
    if n_list > 3:                                   # a spread of radii for longer lists
        lst["radius"] = rng.choice(np.array([4.0, 10.0, 15.0], np.float32), n_list)

    sim = psim.HipParticleSimulation.new(n, psim.GEN_UNIFORM)
    ref = T.oracle_gen_uniform_aos(0, n)
    ref_ptr = T._ptr(ref)
    lst_ptr = T._ptr(lst)
    print(f"mixed soak: particles={n} steps={steps} oracle threads={threads}", flush=True)
    reset = False
    dirty = False                                   # colours differ from the initial colours
    done = 0
    while done < steps:
        m = min(every, steps - done)
        if done >= steps // 2 and not reset:
            sim.set_color_semantics(psim.COLOR_RESET)
            reset = True
        chunk = list(range(done, done + m))
        fused = (done // every) % 2 == 1
        pending = []
        for i in chunk:
            multi = i % 7 == 6
            if fused and not multi:
                pending.append(params[i])
            else:
                if pending:
                    sim.update_batch(pending)
                    pending = []
                if multi:
                    sim.update_multi(params[i], lst)
                else:
                    sim.update(params[i])
            lib.psim_oracle_update_aos(ref_ptr, n, ctypes.byref(params[i]),
                                       lst_ptr if multi else None, n_list if multi else 0, threads)
            if params[i].color_mode in (1, 2):
                dirty = True
            elif reset and dirty:                   # the shader's rule, applied to the oracle's state
                ref["color"] = ref["initial_color"]
                dirty = False
        if pending:
            sim.update_batch(pending)
        done += m
        want = T.oracle_checksum_soa(T.aos_to_soa(ref))
        ok = sim.checksum() == want
        print(f"step {done:5d}: checksums {'equal' if ok else 'DIFFER'}"
              f"{' (colour reset semantics)' if reset else ''}", flush=True)
        if not ok:
            sys.exit(1)
    got = sim.download_aos()
    T.assert_particles_equal(got, ref, "mixed soak, final records")
    print(f"final records: bit-identical over {n} particles x 64 bytes "
          f"(position, velocity, colour, initial colour, paddings)", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=100_000_000)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--every", type=int, default=50)
    ap.add_argument("--attractors", type=int, default=3,
                    help="length of the attractor list of --mix (at most 8: a host list)")
    ap.add_argument("--mix", action="store_true",
                    help="colour modes, attractor lists, fused batches and colour semantics mixed")
    args = ap.parse_args()
    n = args.particles
    if args.mix:
        return mixed_soak(n, args.steps, args.every, args.attractors)
    lib = T.load_oracle()
    threads = int(lib.psim_oracle_max_threads())
    params = stream(args.steps)

    sim = psim.HipParticleSimulation.new(n, psim.GEN_UNIFORM)
    ref = T.oracle_gen_uniform_soa(0, n)
    ptr = [T._ptr(ref[k]) for k in range(6)]
    print(f"particles={n} steps={args.steps} oracle threads={threads}", flush=True)
    assert sim.checksum() == T.oracle_checksum_soa(ref), "initial state differs"

    t_gpu = t_cpu = 0.0
    done = 0
    while done < args.steps:
        m = min(args.every, args.steps - done)
        chunk = params[done:done + m]
        t0 = time.perf_counter()
        if (done // args.every) % 2 == 0:
            for p in chunk:
                sim.update(p)
        else:
            sim.update_batch(chunk)                     # the same steps, fused
        got = sim.checksum()
        t_gpu += time.perf_counter() - t0
        t0 = time.perf_counter()
        for p in chunk:
            lib.psim_oracle_update_soa(*ptr, None, n, ctypes.byref(p), None, 0, threads)
        t_cpu += time.perf_counter() - t0
        want = T.oracle_checksum_soa(ref)
        done += m
        ok = got == want
        print(f"step {done:5d}: checksums {'equal' if ok else 'DIFFER'}  "
              f"(gpu {t_gpu:6.2f} s, oracle {t_cpu:6.2f} s)", flush=True)
        if not ok:
            break

    final = sim.download_soa()
    same = bool(np.array_equal(final.view(np.uint32), ref.view(np.uint32)))
    nan = int(np.isnan(ref).sum())
    if not same and nan:                                # NaN payloads are not compared
        both = np.isnan(final) & np.isnan(ref)
        same = bool(np.array_equal(np.where(both, 0, final.view(np.uint32)),
                                   np.where(both, 0, ref.view(np.uint32))))
    print(f"final state: {'bit-identical' if same else 'DIFFERENT'} over {6 * n} values "
          f"({nan} NaN)", flush=True)
    if not same:
        print(f"max relative error {T.max_rel_err(final, ref):.3e} (bar 1e-5)")
        sys.exit(1)


if __name__ == "__main__":
    main()
