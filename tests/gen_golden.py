#!/usr/bin/env python3
"""Generates tests/golden/*.npz: known-answer vectors for the particle update.

The reference (Rust) cannot be built in this environment and ships no tests, so
nothing of the reference's own pins the numbers.  These vectors come from an
INDEPENDENT restatement of the reference's arithmetic written here, op by op, on
numpy float32 arrays (each numpy ufunc call on float32 is one correctly rounded
IEEE operation: no fused multiply-adds).  The C oracle (oracle/psim_oracle.c) must
reproduce every vector bit for bit (tests/test_oracle_golden.py), so a misreading
in one of the two codings shows up as a mismatch.

What is restated: src/simulation/cpu.rs:65-113 (update body), the glam Vec3 scalar
semantics it calls (length = sqrt((x*x + y*y) + z*z); normalize = v * (1/length)),
Rust's f32::min / max (NaN-ignoring) and f32::clamp (NaN passes through).

Run:  python tests/gen_golden.py      (rewrites tests/golden/)
"""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from psim_testlib import GOLDEN_DIR, make_params, params_to_bytes, psim  # noqa: E402

F = np.float32


# ---- the update, numpy float32 ---------------------------------------------------

def _length(x, y, z):
    return np.sqrt((x * x + y * y) + z * z)


def _attract(vel, pos, apos, aforce, aradius, dt):
    """cpu.rs:76-83 for one attractor, masked where dist < 2r."""
    dx = F(apos[0]) - pos[0]
    dy = F(apos[1]) - pos[1]
    dz = F(apos[2]) - pos[2]
    dist = _length(dx, dy, dz)
    two_r = F(aradius) * F(2.0)
    with np.errstate(all="ignore"):
        inside = dist < two_r                      # False for NaN
        t = F(1.0) - dist / two_r
        ff = (t * t) * F(2.0)
        inv = F(1.0) / dist
        fx = ((dx * inv) * F(aforce)) * ff
        fy = ((dy * inv) * F(aforce)) * ff
        fz = ((dz * inv) * F(aforce)) * ff
        nvx = vel[0] + fx * dt
        nvy = vel[1] + fy * dt
        nvz = vel[2] + fz * dt
    return [np.where(inside, nvx, vel[0]), np.where(inside, nvy, vel[1]),
            np.where(inside, nvz, vel[2])]


def numpy_step(particles, params, attractors=None):
    """One update of an AoS particle array (returns a new array)."""
    out = particles.copy()
    dt = F(params.delta_time)
    pos = [out["position"][:, k].copy() for k in range(3)]
    vel = [out["velocity"][:, k].copy() for k in range(3)]

    vel[1] = vel[1] - F(params.gravity) * dt                         # cpu.rs:72

    if attractors is not None:
        for a in attractors:
            if a["active"] > 0:
                vel = _attract(vel, pos, a["position"], a["force"], a["radius"], dt)
    elif params.is_mouse_dragging > 0:                                # cpu.rs:75
        vel = _attract(vel, pos, list(params.mouse_position), params.mouse_force,
                       params.mouse_radius, dt)

    with np.errstate(all="ignore"):
        pos = [pos[k] + vel[k] * dt for k in range(3)]                # cpu.rs:87
        vel = [vel[k] * F(params.damping) for k in range(3)]          # cpu.rs:90

        if params.color_mode == 1:                                    # cpu.rs:94-99
            speed = _length(*vel)
            ns = np.fmin(speed / F(5.0), F(1.0))                      # NaN-ignoring min
            out["color"][:, 0] = ns
            out["color"][:, 1] = F(0.5) - ns * F(0.5)
            out["color"][:, 2] = F(1.0) - ns
            out["color"][:, 3] = F(1.0)
        elif params.color_mode == 2:                                  # cpu.rs:100-105
            d = _length(*pos)
            nd = d / np.fmax(F(params.max_dist_for_color), F(0.01))
            nd = np.where(nd < F(0.0), F(0.0), nd)                    # clamp keeps NaN
            nd = np.where(nd > F(1.0), F(1.0), nd)
            out["color"][:, 0] = nd
            out["color"][:, 1] = F(0.0)
            out["color"][:, 2] = F(1.0) - nd
            out["color"][:, 3] = F(1.0)
        # else: colour kept (cpu.rs:106)

    for k in range(3):
        out["position"][:, k] = pos[k]
        out["velocity"][:, k] = vel[k]
    return out


def numpy_run(particles, params, steps, attractors=None):
    p = particles
    for _ in range(steps):
        p = numpy_step(p, params, attractors)
    return p


# ---- the synthetic uniform state, numpy uint64 -------------------------------------

def splitmix64_at(seed, counter):
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (np.asarray(counter, dtype=np.uint64) + np.uint64(1)) * np.uint64(
            0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def numpy_gen_uniform(first, n, seed=69):
    g = np.arange(first, first + n, dtype=np.uint64)
    out = np.zeros(n, dtype=psim.PARTICLE_DTYPE)
    for c in range(3):
        up = (splitmix64_at(seed, np.uint64(6) * g + np.uint64(c)) >> np.uint64(40)).astype(
            np.uint32).astype(F) * F(2.0 ** -24)
        uv = (splitmix64_at(seed, np.uint64(6) * g + np.uint64(3 + c)) >> np.uint64(40)).astype(
            np.uint32).astype(F) * F(2.0 ** -24)
        out["position"][:, c] = up * F(100.0) - F(50.0)
        out["velocity"][:, c] = uv * F(2.0) - F(1.0)
        col = (out["position"][:, c] / F(50.0) + F(1.0)) * F(0.5)
        out["color"][:, c] = col
        out["initial_color"][:, c] = col
    out["color"][:, 3] = 1.0
    out["initial_color"][:, 3] = 1.0
    return out


# ---- cases ---------------------------------------------------------------------------

def particles_from(pos, vel, color=None):
    pos = np.asarray(pos, dtype=F).reshape(-1, 3)
    vel = np.asarray(vel, dtype=F).reshape(-1, 3)
    p = np.zeros(len(pos), dtype=psim.PARTICLE_DTYPE)
    p["position"] = pos
    p["velocity"] = vel
    p["padding1"] = F(7.25)            # paddings must survive an update untouched
    p["padding2"] = F(-3.5)
    p["initial_color"] = (0.25, 0.5, 0.75, 1.0)
    p["color"] = (0.1, 0.2, 0.3, 0.4) if color is None else color
    return p


def next_after(x, direction):
    return np.nextafter(F(x), F(direction), dtype=F)


def build_cases():
    """name -> (particles, params, attractors or None)."""
    cases = {}
    rng = np.random.default_rng(20260128)
    att = (0.0, 0.0, 48.0)                    # the app's initial attractor, app.rs:147

    def rnd(n):
        pos = rng.uniform(-50, 50, size=(n, 3)).astype(F)
        vel = rng.uniform(-1, 1, size=(n, 3)).astype(F)
        return particles_from(pos, vel)

    # (1) not dragging, no gravity: drift + damping only
    cases["drift"] = (rnd(64), make_params(is_mouse_dragging=0, mouse_position=att), None)
    # (2) gravity at the UI maximum (app.rs:437)
    cases["gravity5"] = (rnd(64), make_params(gravity=5.0, is_mouse_dragging=0), None)
    # (3) dragging, every particle out of range (dist >= 2r)
    far = particles_from(rng.uniform(-50, -10, size=(64, 3)), rng.uniform(-1, 1, size=(64, 3)))
    cases["drag_out_of_range"] = (far, make_params(is_mouse_dragging=1, mouse_position=att), None)
    # (4) dragging, in range, mid distance
    near = particles_from(np.asarray(att, dtype=F) + rng.uniform(-9, 9, size=(64, 3)).astype(F),
                          rng.uniform(-1, 1, size=(64, 3)))
    cases["drag_in_range"] = (near, make_params(is_mouse_dragging=1, mouse_position=att), None)
    # (5) distance within an ulp of 2r, on both sides, along each axis
    two_r = F(20.0)
    b = []
    for axis in range(3):
        for d in (next_after(two_r, 0), two_r, next_after(two_r, 100)):
            p = np.array(att, dtype=F)
            p[axis] = p[axis] - d
            b.append(p)
    cases["boundary_2r"] = (particles_from(b, np.zeros((len(b), 3))),
                            make_params(is_mouse_dragging=1, mouse_position=att), None)
    # (6) a particle exactly at the attractor: 0 * (1/0) = NaN propagates (cpu.rs:76-82)
    at = particles_from([att, (1.0, 2.0, 3.0)], [(0.5, 0.5, 0.5), (0.0, 0.0, 0.0)])
    cases["at_attractor"] = (at, make_params(is_mouse_dragging=1, mouse_position=att), None)
    cases["at_attractor_color1"] = (at.copy(), make_params(is_mouse_dragging=1, color_mode=1,
                                                           mouse_position=att), None)
    cases["at_attractor_color2"] = (at.copy(), make_params(is_mouse_dragging=1, color_mode=2,
                                                           mouse_position=att), None)
    # (7) colour modes
    fast = particles_from(rng.uniform(-50, 50, size=(64, 3)), rng.uniform(-12, 12, size=(64, 3)))
    cases["color1_speed"] = (fast, make_params(color_mode=1, is_mouse_dragging=1,
                                               mouse_position=att), None)
    cases["color2_distance"] = (rnd(64), make_params(color_mode=2, is_mouse_dragging=1,
                                                     mouse_position=att), None)
    cases["color2_small_maxdist"] = (rnd(64), make_params(color_mode=2, max_dist_for_color=0.001),
                                     None)
    cases["color2_far"] = (particles_from(rng.uniform(-500, 500, size=(64, 3)),
                                          np.zeros((64, 3))),
                           make_params(color_mode=2, max_dist_for_color=50.0), None)
    cases["color0_keeps_current"] = (rnd(64), make_params(color_mode=0, is_mouse_dragging=1,
                                                          mouse_position=att), None)
    cases["color_mode_other_keeps"] = (rnd(16), make_params(color_mode=7), None)
    # (8) radius / force at the UI extremes (app.rs:407-409)
    for r in (1.0, 50.0):
        for f in (0.0, 100.0):
            around = particles_from(
                np.asarray(att, dtype=F) + rng.uniform(-2 * r, 2 * r, size=(64, 3)).astype(F),
                rng.uniform(-1, 1, size=(64, 3)))
            cases[f"ui_r{int(r)}_f{int(f)}"] = (
                around, make_params(is_mouse_dragging=1, mouse_radius=r, mouse_force=f,
                                    mouse_position=att), None)
    # (9) time steps
    for dt in (0.001, 0.1):
        cases[f"dt_{dt}"] = (near.copy(), make_params(delta_time=dt, gravity=1.0,
                                                      is_mouse_dragging=1, mouse_position=att),
                             None)
    # (10) uniform-random batch (the benchmark's distribution) x parameter sweep
    batch = numpy_gen_uniform(0, 1024)
    cases["uniform_bench"] = (batch, make_params(is_mouse_dragging=1, mouse_position=att), None)
    cases["uniform_stress"] = (batch.copy(), make_params(is_mouse_dragging=1, mouse_radius=50.0,
                                                         mouse_force=100.0, gravity=2.5,
                                                         mouse_position=att), None)
    cases["uniform_color1"] = (batch.copy(), make_params(is_mouse_dragging=1, color_mode=1,
                                                         mouse_radius=50.0, mouse_force=100.0,
                                                         mouse_position=att), None)
    cases["uniform_color2"] = (batch.copy(), make_params(is_mouse_dragging=1, color_mode=2,
                                                         mouse_radius=25.0, mouse_force=50.0,
                                                         mouse_position=att), None)
    # odd sizes: the kernel steps particles in groups of four plus a tail
    for n in (1, 2, 3, 5, 7, 1021):
        cases[f"ragged_{n}"] = (numpy_gen_uniform(100, n),
                                make_params(is_mouse_dragging=1, mouse_radius=50.0,
                                            mouse_force=20.0, mouse_position=att), None)
    # multi-attractor extension: 8 attractors, one inactive; and its reduction to the
    # reference with a single attractor equal to the mouse fields
    attrs = np.zeros(8, dtype=psim.ATTRACTOR_DTYPE)
    attrs["position"] = rng.uniform(-40, 40, size=(8, 3)).astype(F)
    attrs["force"] = rng.uniform(1, 100, size=8).astype(F)
    attrs["radius"] = rng.uniform(5, 50, size=8).astype(F)
    attrs["active"] = 1
    attrs["active"][3] = 0
    cases["multi8"] = (batch.copy(), make_params(is_mouse_dragging=0, gravity=1.0), attrs)
    one = np.zeros(1, dtype=psim.ATTRACTOR_DTYPE)
    one["position"] = att
    one["force"] = 5.0
    one["radius"] = 10.0
    one["active"] = 1
    cases["multi1_equals_single"] = (batch.copy(),
                                     make_params(is_mouse_dragging=1, mouse_position=att), one)
    # numeric extremes: subnormal state (no flush to zero anywhere: sqrt, divide and the
    # compare must all work on subnormals), and magnitudes whose squares overflow
    tiny = particles_from(rng.uniform(-1, 1, size=(64, 3)) * 1e-39,
                          rng.uniform(-1, 1, size=(64, 3)) * 1e-40)
    cases["subnormal_state"] = (tiny, make_params(is_mouse_dragging=1, mouse_radius=1e-38,
                                                  mouse_force=5.0, gravity=1e-38,
                                                  mouse_position=(0.0, 0.0, 0.0)), None)
    cases["subnormal_color2"] = (tiny.copy(), make_params(color_mode=2, max_dist_for_color=50.0,
                                                          is_mouse_dragging=0), None)
    huge = particles_from(rng.uniform(-1, 1, size=(64, 3)) * 1e30,
                          rng.uniform(-1, 1, size=(64, 3)) * 1e25)
    huge["position"][:8] = rng.uniform(-1, 1, size=(8, 3)) * 1e18       # squares still finite
    cases["huge_magnitudes"] = (huge, make_params(is_mouse_dragging=1, mouse_radius=1e20,
                                                  mouse_force=100.0, color_mode=1,
                                                  mouse_position=att), None)
    cases["huge_color2"] = (huge.copy(), make_params(color_mode=2, is_mouse_dragging=0), None)
    return cases


STEPS_LONG = 100


def main():
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    arrays = {}
    names = []
    for name, (particles, params, attractors) in build_cases().items():
        names.append(name)
        arrays[f"{name}.params"] = params_to_bytes(params)
        arrays[f"{name}.in"] = particles.view(np.uint8)
        arrays[f"{name}.out1"] = numpy_run(particles, params, 1, attractors).view(np.uint8)
        arrays[f"{name}.out{STEPS_LONG}"] = numpy_run(particles, params, STEPS_LONG,
                                                      attractors).view(np.uint8)
        if attractors is not None:
            arrays[f"{name}.attractors"] = attractors.view(np.uint8)
    arrays["names"] = np.array(names)
    np.savez_compressed(os.path.join(GOLDEN_DIR, "update_cases.npz"), **arrays)

    # generator vectors: SplitMix64 outputs and the first particles of the uniform state
    gen = {
        "splitmix64_seed0": splitmix64_at(0, np.arange(8, dtype=np.uint64)),
        "splitmix64_seed69": splitmix64_at(69, np.arange(8, dtype=np.uint64)),
        "uniform_first64": numpy_gen_uniform(0, 64).view(np.uint8),
        "uniform_at_1e9": numpy_gen_uniform(999_999_990, 10).view(np.uint8),
    }
    np.savez_compressed(os.path.join(GOLDEN_DIR, "generator_vectors.npz"), **gen)
    total = sum(os.path.getsize(os.path.join(GOLDEN_DIR, f)) for f in os.listdir(GOLDEN_DIR))
    print(f"wrote {len(names)} update cases; {total / 1024:.0f} KiB in {GOLDEN_DIR}")


if __name__ == "__main__":
    main()
