"""Seeded random update cases shared by the CPU and GPU parity tests: ordinary values
mixed with the awkward ones (zeros of both signs, subnormals, huge magnitudes, infinities,
NaNs, negative radii and time steps)."""
import numpy as np

import psim_testlib as T
from psim_testlib import psim

F = np.float32
SPECIALS = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-39, 1.17549435e-38, 1e-20, 1.0, -1.0, 48.0,
                     1e19, -1e19, 3e38, -3e38, np.inf, -np.inf, np.nan], dtype=F)


def random_particles(rng, n, special_fraction):
    p = np.zeros(n, dtype=psim.PARTICLE_DTYPE)
    p["position"] = rng.uniform(-60, 60, size=(n, 3)).astype(F)
    p["velocity"] = (rng.standard_normal(size=(n, 3)) * 3).astype(F)
    p["color"] = rng.uniform(0, 1, size=(n, 4)).astype(F)
    p["initial_color"] = rng.uniform(0, 1, size=(n, 4)).astype(F)
    p["padding1"] = rng.uniform(-9, 9, size=n).astype(F)
    p["padding2"] = rng.uniform(-9, 9, size=n).astype(F)
    for field in ("position", "velocity"):
        mask = rng.random(size=(n, 3)) < special_fraction
        p[field][mask] = rng.choice(SPECIALS, size=int(mask.sum()))
    return p


def random_params(rng, wild):
    """`wild` widens the ranges past what the application's sliders allow."""
    def pick(lo, hi, odd):
        if wild and rng.random() < 0.25:
            return float(rng.choice(np.asarray(odd, dtype=F)))
        return float(F(rng.uniform(lo, hi)))

    att = [pick(-55, 55, [0.0, -0.0, 1e-39, 1e19, np.inf, np.nan]) for _ in range(3)]
    return T.make_params(
        delta_time=pick(0.0005, 0.1, [0.0, -0.016, 1e-39, 10.0, np.inf, np.nan]),
        gravity=pick(0.0, 5.0, [-5.0, 1e-40, 1e30, np.nan]),
        color_mode=int(rng.integers(0, 4)),
        mouse_force=pick(0.0, 100.0, [-50.0, 1e-42, 1e30, np.inf]),
        mouse_radius=pick(1.0, 50.0, [0.0, -3.0, 1e-39, 1e30, np.inf, np.nan]),
        is_mouse_dragging=int(rng.integers(0, 3)),
        damping=pick(0.9, 1.0, [0.0, 1.5, -1.0, np.nan]),
        max_dist_for_color=pick(1.0, 100.0, [0.0, -1.0, 0.001, 1e-40, np.inf, np.nan]),
        mouse_position=att)


def random_attractors(rng, k, wild):
    a = np.zeros(k, dtype=psim.ATTRACTOR_DTYPE)
    a["position"] = rng.uniform(-55, 55, size=(k, 3)).astype(F)
    a["force"] = rng.uniform(-20, 100, size=k).astype(F)
    a["radius"] = rng.uniform(0.5, 60, size=k).astype(F)
    a["active"] = rng.integers(0, 3, size=k)
    if wild and k:
        a["radius"][rng.integers(0, k)] = rng.choice(np.asarray([0.0, -1.0, np.inf, np.nan], dtype=F))
    return a


def cases(count, seed=20260128):
    """Yields (index, particles, params, attractors or None, steps)."""
    rng = np.random.default_rng(seed)
    for i in range(count):
        wild = i % 3 == 2
        n = int(rng.choice([1, 3, 4, 5, 63, 64, 65, 255, 257, 1000]))
        particles = random_particles(rng, n, 0.02 if wild else 0.0)
        params = random_params(rng, wild)
        # every fourth case a list: mostly 1-8 attractors, every sixteenth case 9-64
        k = int(rng.integers(9, 65)) if i % 16 == 15 else int(rng.integers(1, 9))
        attractors = random_attractors(rng, k, wild) if i % 4 == 3 else None
        steps = int(rng.choice([1, 2, 7]))
        yield i, particles, params, attractors, steps
