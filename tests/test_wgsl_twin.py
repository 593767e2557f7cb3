"""The reference's OTHER implementation of the update as a second witness.

The reference holds the per-frame update twice: the CPU path the oracle restates
(src/simulation/cpu.rs:65-113) and the compute shader (src/shaders/compute.wgsl:31-99), written
by the same authors in another language.  `wgsl_main` below restates the SHADER, statement by
statement, from its own text -- `normalize(dir)` is WGSL's `dir / length(dir)`, the distance
ratio is clamped, colours go through `clamp`, mode 0 resets the colour to the initial one -- and
the oracle, derived from the Rust text, must agree with it wherever the two sources say the
same thing.  A misreading of the force law, of the order of gravity / force / integration /
damping or of a colour formula in either restatement shows here as a difference far above
rounding.  Where the two sources differ on purpose (what mode 0 does to the colour), the test
checks exactly that difference, and on the GPU the shader's rule (PSIM_COLOR_RESET).

WGSL leaves the precision of length / normalize to the implementation, so this witness pins the
ALGORITHM to the bar the path is held to (1e-5 relative), not the bits; parity with the
reference's arithmetic stays unpinned (the reference ships no vectors).  The yardstick for a
component is the magnitude of the vector it belongs to, with a floor of the world's scale
(50 for positions, 1 for velocities and colours): a component that passes through zero has no
relative error of its own.
"""
import numpy as np
import pytest

import psim_testlib as T
from psim_testlib import psim

F = np.float32
REL = 1e-5


def wgsl_main(particles, params):
    """One dispatch of compute.wgsl `main` over every particle (compute.wgsl:31-99), in fp32."""
    out = particles.copy()
    delta_time, gravity = F(params.delta_time), F(params.gravity)          # wgsl:40-43
    damping, max_dist = F(params.damping), F(params.max_dist_for_color)
    position = out["position"].astype(F).copy()                            # wgsl:46-49
    velocity = out["velocity"].astype(F).copy()
    initial_color = out["initial_color"].copy()
    velocity[:, 1] -= gravity * delta_time                                 # wgsl:52
    if params.is_mouse_dragging > 0:                                       # wgsl:55
        mouse = np.array([F(v) for v in params.mouse_position], dtype=F)
        direction = mouse - position                                       # wgsl:56
        dist = np.sqrt((direction * direction).sum(axis=1, dtype=F))       # wgsl:57 length()
        reach = F(params.mouse_radius) * F(2.0)
        inside = dist < reach                                              # wgsl:59
        with np.errstate(all="ignore"):
            normalized_dist = np.clip(dist / reach, F(0.0), F(1.0))        # wgsl:61
            force_factor = (F(1.0) - normalized_dist) * (F(1.0) - normalized_dist) * F(2.0)
            unit = direction / dist[:, None]                               # normalize(dir)
            push = unit * F(params.mouse_force) * force_factor[:, None] * delta_time   # wgsl:63
        velocity = np.where(inside[:, None], velocity + push, velocity)
    position = position + velocity * delta_time                            # wgsl:68
    velocity = velocity * damping                                          # wgsl:71
    if params.color_mode == 1:                                             # wgsl:78-82
        speed = np.sqrt((velocity * velocity).sum(axis=1, dtype=F))
        s = np.clip(speed / F(5.0), F(0.0), F(1.0))
        color = np.stack([s, F(0.5) - s * F(0.5), F(1.0) - s, np.ones_like(s)], axis=1)
    elif params.color_mode == 2:                                           # wgsl:83-89
        far = np.sqrt((position * position).sum(axis=1, dtype=F))
        d = np.clip(far / max(max_dist, F(0.01)), F(0.0), F(1.0))
        color = np.stack([d, np.zeros_like(d), F(1.0) - d, np.ones_like(d)], axis=1)
    else:                                                                  # wgsl:75-77, 90-92
        color = initial_color
    out["position"], out["velocity"], out["color"] = position, velocity, color   # wgsl:96-98
    return out


def assert_within_bar(got, want, what):
    for field, floor in (("position", 50.0), ("velocity", 1.0), ("color", 1.0)):
        g, w = got[field].astype(np.float64), want[field].astype(np.float64)
        scale = np.maximum(np.sqrt((w * w).sum(axis=1)), floor)[:, None]
        worst = float((np.abs(g - w) / scale).max())
        assert worst <= REL, f"{what}: {field} differs by {worst:.3e} of its vector's magnitude"


CASES = {
    "benchmark": dict(),
    "sliders at their maxima": dict(mouse_radius=50.0, mouse_force=100.0, gravity=5.0),
    "not dragging, gravity": dict(is_mouse_dragging=0, gravity=2.0),
    "small radius, long frame": dict(mouse_radius=3.0, mouse_force=40.0, delta_time=0.033),
    "colour by speed": dict(color_mode=1, mouse_radius=35.0, mouse_force=60.0),
    "colour by distance": dict(color_mode=2, mouse_radius=25.0, max_dist_for_color=20.0),
    "colour by distance, tiny range": dict(color_mode=2, max_dist_for_color=0.001),
}


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("steps", [1, 12])
def test_oracle_agrees_with_the_shader_where_the_sources_agree(name, steps):
    params = T.bench_params(**CASES[name])
    start = T.oracle_gen_uniform_aos(0, 30_000)
    cpu = T.oracle_update_aos(start, params, steps)
    shader = start
    for _ in range(steps):
        shader = wgsl_main(shader, params)
    if params.color_mode not in (1, 2):
        # the one place the sources disagree: the CPU path keeps the colour, the shader
        # rewrites it with the initial colour (the same values here, never having changed)
        assert cpu["color"].tobytes() == start["color"].tobytes()
        assert shader["color"].tobytes() == start["initial_color"].tobytes()
    assert_within_bar(cpu, shader, f"{name}, {steps} steps")
    # the case moved the particles at all
    assert cpu["position"].tobytes() != start["position"].tobytes()
    if params.is_mouse_dragging and CASES[name].get("mouse_radius", 10.0) >= 10.0:
        mouse = np.array(params.mouse_position, dtype=np.float32)
        near = np.sqrt(((start["position"] - mouse) ** 2).sum(axis=1)) < params.mouse_radius * 2
        assert near.any(), "no particle inside the influence sphere: the force law went untested"


def test_the_sources_differ_only_in_what_mode_zero_does_to_the_colour():
    """After a colouring step, a mode-0 step: the CPU path keeps the computed colours
    (cpu.rs:106), the shader puts the initial ones back (compute.wgsl:75-77)."""
    colouring, plain = T.bench_params(color_mode=1, mouse_radius=40.0), T.bench_params()
    start = T.oracle_gen_uniform_aos(0, 5_000)
    cpu = T.oracle_update_aos(T.oracle_update_aos(start, colouring), plain)
    shader = wgsl_main(wgsl_main(start, colouring), plain)
    assert shader["color"].tobytes() == start["initial_color"].tobytes()
    assert cpu["color"].tobytes() != start["initial_color"].tobytes()
    cpu["color"] = shader["color"]
    assert_within_bar(cpu, shader, "positions and velocities")


@pytest.mark.gpu
def test_gpu_under_the_shaders_colour_rule_matches_the_shader():
    """The HIP path with PSIM_COLOR_RESET against the shader's restatement over a frame
    sequence that visits every colour mode and returns to 0: all 64 bytes of every record,
    positions / velocities / computed colours within the bar, reset colours exactly."""
    n = 200_003
    frames = [T.bench_params(color_mode=m, mouse_radius=r, mouse_force=f, gravity=g)
              for m, r, f, g in ((0, 10.0, 5.0, 0.0), (1, 40.0, 60.0, 1.0), (0, 40.0, 60.0, 1.0),
                                 (2, 25.0, 20.0, 0.0), (2, 25.0, 20.0, 0.0), (0, 10.0, 5.0, 2.0))]
    with psim.HipParticleSimulation.new(n, psim.GEN_UNIFORM) as sim:
        sim.set_color_semantics(psim.COLOR_RESET)
        start = sim.download_aos()
        shader = start
        for k, params in enumerate(frames):
            sim.update(params)
            shader = wgsl_main(shader, params)
            got = sim.download_aos()
            assert_within_bar(got, shader, f"frame {k}")
            if params.color_mode == 0:
                assert got["color"].tobytes() == start["initial_color"].tobytes(), f"frame {k}"
        for field in ("padding1", "padding2", "initial_color"):
            assert got[field].tobytes() == start[field].tobytes(), field
