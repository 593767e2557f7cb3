"""CPU tests: the C oracle against the committed golden vectors and hand-derived answers.

The golden vectors come from tests/gen_golden.py (an independent numpy-float32
restatement of src/simulation/cpu.rs:65-113 of the reference); the reference itself
holds no tests or fixtures for this path, so these are what pin the oracle.
"""
import ctypes

import numpy as np
import pytest

import psim_testlib as T
from psim_testlib import psim

STEPS_LONG = 100


def _case(golden, name):
    particles = golden[f"{name}.in"].view(psim.PARTICLE_DTYPE)
    params = T.params_from_bytes(golden[f"{name}.params"])
    key = f"{name}.attractors"
    attractors = golden[key].view(psim.ATTRACTOR_DTYPE) if key in golden.files else None
    return particles, params, attractors


def case_names():
    import os
    g = np.load(os.path.join(T.GOLDEN_DIR, "update_cases.npz"))
    return [str(n) for n in g["names"]]


def test_layouts_match_reference():
    # SimParams 48 B / Particle 64 B with the reference's offsets (mod.rs:52-98)
    assert T.load_oracle().psim_oracle_layout_ok() == 1
    assert ctypes.sizeof(psim.SimParams) == 48
    assert psim.SimParams.color_mode.offset == 8
    assert psim.SimParams.is_mouse_dragging.offset == 20
    assert psim.SimParams.mouse_position.offset == 32
    assert psim.PARTICLE_DTYPE.fields["velocity"][1] == 16
    assert psim.PARTICLE_DTYPE.fields["color"][1] == 32
    assert psim.PARTICLE_DTYPE.fields["initial_color"][1] == 48


def test_sim_params_defaults_match_reference():
    # impl Default for SimParams, mod.rs:69-84
    p = psim.SimParams()
    assert np.float32(p.delta_time) == np.float32(0.016)
    assert p.gravity == 0.0 and p.color_mode == 0 and p.mouse_force == 5.0
    assert p.mouse_radius == 10.0 and p.is_mouse_dragging == 0
    assert np.float32(p.damping) == np.float32(0.99) and p.max_dist_for_color == 50.0
    assert list(p.mouse_position) == [0.0, 0.0, 0.0]


@pytest.mark.parametrize("name", case_names())
def test_oracle_reproduces_golden(golden_update, name):
    particles, params, attractors = _case(golden_update, name)
    for steps in (1, STEPS_LONG):
        want = golden_update[f"{name}.out{steps}"].view(psim.PARTICLE_DTYPE)
        got = T.oracle_update_aos(particles, params, steps, attractors)
        T.assert_particles_equal(got, want, f"{name}/{steps} steps")


@pytest.mark.parametrize("name", ["uniform_stress", "uniform_color1", "uniform_color2",
                                  "at_attractor", "multi8", "ragged_7"])
def test_oracle_soa_equals_aos(golden_update, name):
    particles, params, attractors = _case(golden_update, name)
    want = T.oracle_update_aos(particles, params, 3, attractors)
    soa, col = T.oracle_update_soa(T.aos_to_soa(particles), params, 3,
                                   color=particles["color"].copy(), attractors=attractors)
    T.assert_same_f32(soa, T.aos_to_soa(want), f"{name} soa")
    T.assert_same_f32(col, want["color"], f"{name} colour")


def test_oracle_independent_of_thread_count(golden_update):
    particles, params, _ = _case(golden_update, "uniform_stress")
    a = T.oracle_update_aos(particles, params, 5, threads=1)
    b = T.oracle_update_aos(particles, params, 5, threads=3)
    assert a.tobytes() == b.tobytes()


def test_hand_derived_in_range_particle():
    """One particle worked by hand, every intermediate exact in fp32.

    attractor (0,0,48), r = 10 (2r = 20), force 5, dt = 0.5, damping = 0.5, g = 0.
    particle p = (0,0,38), v = 0:  dir = (0,0,10), dist = 10 < 20;
    t = 1 - 10/20 = 0.5; ff = 0.5*0.5*2 = 0.5; 1/dist = fl(0.1);
    n.z = 10*fl(0.1) = 1.00000001490116 -> rounds to 1.0;
    force.z = 1*5*0.5 = 2.5; v.z = 0 + 2.5*0.5 = 1.25;
    p.z = 38 + 1.25*0.5 = 38.625; v.z *= 0.5 -> 0.625.
    """
    p = np.zeros(1, dtype=psim.PARTICLE_DTYPE)
    p["position"] = (0.0, 0.0, 38.0)
    p["color"] = (0.1, 0.2, 0.3, 0.4)
    params = T.make_params(delta_time=0.5, damping=0.5, is_mouse_dragging=1,
                           mouse_position=(0.0, 0.0, 48.0), mouse_radius=10.0, mouse_force=5.0)
    out = T.oracle_update_aos(p, params)
    assert out["position"][0].tolist() == [0.0, 0.0, 38.625]
    assert out["velocity"][0].tolist() == [0.0, 0.0, 0.625]
    assert out["color"][0].tolist() == p["color"][0].tolist()      # mode 0 keeps colour


def test_hand_derived_gravity_and_drift():
    """g = 4, dt = 0.25, damping = 0.5, no attractor: v.y = 1 - 4*0.25 = 0;
    v = (2, 0, -4); p = (1,2,3) + v*0.25 = (1.5, 2, 2); v *= 0.5 -> (1, 0, -2)."""
    p = np.zeros(1, dtype=psim.PARTICLE_DTYPE)
    p["position"] = (1.0, 2.0, 3.0)
    p["velocity"] = (2.0, 1.0, -4.0)
    params = T.make_params(delta_time=0.25, gravity=4.0, damping=0.5, is_mouse_dragging=0)
    out = T.oracle_update_aos(p, params)
    assert out["position"][0].tolist() == [1.5, 2.0, 2.0]
    assert out["velocity"][0].tolist() == [1.0, 0.0, -2.0]


def test_hand_derived_colour_modes():
    """Mode 1 uses the DAMPED velocity: v = (3,4,0)*1.0 -> speed 5 -> ns = 1 ->
    [1, 0, 0, 1]; with damping 0.5: speed 2.5 -> ns = 0.5 -> [0.5, 0.25, 0.5, 1].
    Mode 2 uses the NEW position: p = (30,40,0), v = 0, max_dist 100 -> 0.5 ->
    [0.5, 0, 0.5, 1]."""
    p = np.zeros(1, dtype=psim.PARTICLE_DTYPE)
    p["velocity"] = (3.0, 4.0, 0.0)
    out = T.oracle_update_aos(p, T.make_params(delta_time=0.0, damping=1.0, color_mode=1))
    assert out["color"][0].tolist() == [1.0, 0.0, 0.0, 1.0]
    out = T.oracle_update_aos(p, T.make_params(delta_time=0.0, damping=0.5, color_mode=1))
    assert out["color"][0].tolist() == [0.5, 0.25, 0.5, 1.0]
    q = np.zeros(1, dtype=psim.PARTICLE_DTYPE)
    q["position"] = (30.0, 40.0, 0.0)
    out = T.oracle_update_aos(q, T.make_params(color_mode=2, max_dist_for_color=100.0))
    assert out["color"][0].tolist() == [0.5, 0.0, 0.5, 1.0]
    # max_dist below 0.01 is clamped up to 0.01 (cpu.rs:103): 50/0.01 -> clamp -> 1
    out = T.oracle_update_aos(q, T.make_params(color_mode=2, max_dist_for_color=0.0))
    assert out["color"][0].tolist() == [1.0, 0.0, 0.0, 1.0]


def test_particle_at_attractor_goes_nan():
    # cpu.rs:76-82: dir = 0, dist = 0 < 2r, normalize = 0 * (1/0) = NaN
    p = np.zeros(2, dtype=psim.PARTICLE_DTYPE)
    p["position"][0] = (0.0, 0.0, 48.0)
    p["position"][1] = (5.0, 5.0, 5.0)
    out = T.oracle_update_aos(p, T.bench_params())
    assert np.isnan(out["position"][0]).all() and np.isnan(out["velocity"][0]).all()
    assert np.isfinite(out["position"][1]).all()
    # mode 1 on a NaN velocity: f32::min ignores NaN -> ns = 1
    out = T.oracle_update_aos(p, T.bench_params(color_mode=1))
    assert out["color"][0].tolist() == [1.0, 0.0, 0.0, 1.0]
    # mode 2 on a NaN position: clamp lets NaN through
    out = T.oracle_update_aos(p, T.bench_params(color_mode=2))
    assert np.isnan(out["color"][0][0]) and out["color"][0][1] == 0.0


def test_paddings_and_initial_colour_untouched(golden_update):
    particles, params, _ = _case(golden_update, "uniform_color1")
    particles = particles.copy()
    particles["padding1"] = 11.0
    particles["padding2"] = -13.0
    out = T.oracle_update_aos(particles, params, 4)
    assert (out["padding1"] == 11.0).all() and (out["padding2"] == -13.0).all()
    assert out["initial_color"].tobytes() == particles["initial_color"].tobytes()


def test_single_attractor_list_equals_mouse_fields(golden_update):
    particles, params, attractors = _case(golden_update, "multi1_equals_single")
    a = T.oracle_update_aos(particles, params, 10, attractors)
    b = T.oracle_update_aos(particles, params, 10, None)
    assert a.tobytes() == b.tobytes()


# ---- generators ---------------------------------------------------------------------

def test_splitmix64_known_answers(golden_generator):
    # First outputs of the published SplitMix64 stream for seed 0.
    g = golden_generator["splitmix64_seed0"]
    assert int(g[0]) == 0xE220A8397B1DCDAF
    assert int(g[1]) == 0x6E789E6AA1B965F4
    assert int(g[2]) == 0x06C45D188009454F


def test_oracle_uniform_matches_golden(golden_generator):
    want = golden_generator["uniform_first64"].view(psim.PARTICLE_DTYPE)
    got = T.oracle_gen_uniform_aos(0, 64)
    T.assert_particles_equal(got, want, "uniform[0:64]")
    want = golden_generator["uniform_at_1e9"].view(psim.PARTICLE_DTYPE)
    got = T.oracle_gen_uniform_aos(999_999_990, 10)
    T.assert_particles_equal(got, want, "uniform[1e9-10:1e9]")
    soa = T.oracle_gen_uniform_soa(0, 64)
    T.assert_same_f32(soa, T.aos_to_soa(golden_generator["uniform_first64"].view(
        psim.PARTICLE_DTYPE)), "uniform soa")


def test_uniform_is_index_addressed_and_in_range():
    whole = T.oracle_gen_uniform_aos(0, 4096)
    part = T.oracle_gen_uniform_aos(1000, 500)
    assert whole[1000:1500].tobytes() == part.tobytes()
    assert (whole["position"] >= -50).all() and (whole["position"] < 50).all()
    assert (whole["velocity"] >= -1).all() and (whole["velocity"] < 1).all()
    assert abs(float(whole["position"].mean())) < 2.0
    np.testing.assert_array_equal(whole["color"], whole["initial_color"])
    np.testing.assert_array_equal(
        whole["color"][:, :3],
        (whole["position"] / np.float32(50) + np.float32(1)) * np.float32(0.5))


def test_xoshiro256pp_known_answers():
    # Published test vector of Xoshiro256++ for the state {1, 2, 3, 4}.
    lib = T.load_oracle()
    s = (ctypes.c_uint64 * 4)(1, 2, 3, 4)
    out = (ctypes.c_uint64 * 10)()
    lib.psim_oracle_xoshiro256pp(s, out, ctypes.c_uint64(10))
    assert list(out) == [41943041, 58720359, 3588806011781223, 3591011842654386,
                         9228616714210784205, 9973669472204895162, 14011001112246962877,
                         12406186145184390807, 15849039046786891736, 10450023813501588000]


def test_hollow_sphere_properties():
    # mod.rs:150-167: on the r = 50 sphere, y from +50 to -50, zero velocity
    p = T.oracle_gen_hollow_aos(1000)
    r = np.sqrt((p["position"].astype(np.float64) ** 2).sum(axis=1))
    np.testing.assert_allclose(r, 50.0, rtol=1e-5)
    assert p["position"][0][1] == 50.0 and p["position"][-1][1] == -50.0
    assert (p["velocity"] == 0).all() and (p["color"][:, 3] == 1).all()
    np.testing.assert_array_equal(p["color"], p["initial_color"])
    # a sub-range equals the slice of the whole set
    assert T.oracle_gen_hollow_aos(1000, 200, 300).tobytes() == p[200:500].tobytes()
    # count == 1: 0/0 -> NaN y, as the reference computes (mod.rs:153)
    one = T.oracle_gen_hollow_aos(1)
    assert np.isnan(one["position"][0][1])


def test_filled_sphere_properties():
    # mod.rs:168-188: inside the r = 50 ball, seeded (same output every call)
    p = T.oracle_gen_filled_aos(0, 2000)
    r = np.sqrt((p["position"].astype(np.float64) ** 2).sum(axis=1))
    assert (r <= 50.0 * (1 + 1e-5)).all()
    assert 30.0 < r.mean() < 45.0            # E[r] = 37.5 for a uniform ball
    assert T.oracle_gen_filled_aos(500, 100).tobytes() == p[500:600].tobytes()


def test_checksum_oracle_is_sum_of_bit_patterns():
    soa = T.oracle_gen_uniform_soa(0, 1000)
    want = [int(soa[k].view(np.uint32).astype(np.uint64).sum()) for k in range(6)]
    assert T.oracle_checksum_soa(soa) == want


def test_oracle_runs_clean_under_sanitizers():
    """AddressSanitizer + UBSan over every oracle entry point (host code only)."""
    import os
    import subprocess
    subprocess.check_call(["make", "-C", T.ORACLE_DIR, "selftest"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, OMP_NUM_THREADS="4")
    r = subprocess.run([os.path.join(T.ORACLE_DIR, "selftest")], capture_output=True, text=True,
                       env=env, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "oracle selftest OK" in r.stdout
