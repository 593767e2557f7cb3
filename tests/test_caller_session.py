"""The caller side of the path: one frame of input -> one SimParams
(tests/cpp/psim_caller.hpp, mirroring src/app.rs:202-301, 528-576 and src/camera.rs).
A seeded synthetic session (cursor drags, camera rotation and flight, scroll-wheel depth,
window resizes, slider changes, pauses) goes through the C++ mirror and through the numpy
restatement in oracle/caller_oracle.py; the SimParams streams must agree bit for bit.  On
the GPU the same session then drives the simulation, one update per frame and as one fused
batch, against the CPU oracle's replay."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

import psim_testlib as T
from psim_testlib import psim

sys.path.insert(0, os.path.join(T.ROOT, "oracle"))
import caller_oracle as C  # noqa: E402

CPP_DIR = os.path.join(T.ROOT, "tests", "cpp")
CALLER = os.path.join(CPP_DIR, "caller_params")
REPLAY = os.path.join(CPP_DIR, "session_replay")


def build():
    subprocess.check_call(["make", "-C", CPP_DIR], stdout=subprocess.DEVNULL)


def synthetic_session(frames, seed):
    rng = np.random.default_rng(seed)
    inputs = np.zeros(frames, C.INPUT_DTYPE)
    ui = np.zeros(frames, C.UI_DTYPE)
    screen = (1280.0, 720.0)
    cursor = np.array([640.0, 360.0])
    for i in range(frames):
        if i == frames // 3:
            screen = (1600.0, 900.0)
        if i == 2 * frames // 3:
            screen = (1000.0, 1000.0)
        step = rng.normal(0.0, 12.0, 2)
        cursor = np.clip(cursor + step, 0.0, np.array(screen) - 1.0)
        phase = (i // 12) % 6
        inputs[i]["delta_time"] = rng.choice([0.016, 0.0166667, 0.0175, 0.033, 0.008])
        inputs[i]["cursor"] = cursor
        inputs[i]["screen"] = screen
        inputs[i]["pointer_delta"] = step if phase in (2, 5) else (0.0, 0.0)
        inputs[i]["scroll_y"] = rng.choice([0.0, 0.0, 0.0, 30.0, -30.0, 7.5])
        inputs[i]["keys_down"] = rng.integers(0, 32) if phase in (1, 4) else 0
        inputs[i]["shift_down"] = int(phase == 4 and i % 3 == 0)
        inputs[i]["primary_down"] = int(phase in (0, 1, 3, 5))
        inputs[i]["secondary_down"] = int(phase in (2, 5))
        ui[i]["gravity"] = 0.0 if i < 40 else 2.5
        ui[i]["mouse_force"] = 5.0 if i < 60 else 60.0
        ui[i]["mouse_radius"] = 10.0 if i < 60 else 35.0
        ui[i]["max_dist_for_color"] = 50.0 if i < 90 else 20.0
        ui[i]["color_mode"] = (i // 50) % 3
        ui[i]["paused"] = int(100 <= i < 108)
    return inputs, ui


def run_caller(tmp_path, inputs, ui):
    build()
    fi, fu, fo = (str(tmp_path / n) for n in ("inputs.bin", "ui.bin", "params.bin"))
    inputs.tofile(fi)
    ui.tofile(fu)
    r = subprocess.run([CALLER, fi, fu, fo], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    return np.fromfile(fo, dtype=C.PARAMS_DTYPE)


def test_record_layouts_match_the_headers():
    assert C.PARAMS_DTYPE.itemsize == 48 and C.PARAMS_DTYPE.fields["mouse_position"][1] == 32
    assert C.INPUT_DTYPE.itemsize == 48


def test_cpp_mirror_matches_oracle_bit_for_bit(tmp_path):
    inputs, ui = synthetic_session(360, seed=7)
    got = run_caller(tmp_path, inputs, ui)
    want = C.replay(inputs, ui)
    assert len(got) == len(want) == 360 - 8
    assert got.tobytes() == want.tobytes()
    # the session really exercised the projection, the scroll depth and the camera
    pos = want["mouse_position"]
    assert np.ptp(pos[:, 0]) > 1 and np.ptp(pos[:, 1]) > 1 and np.ptp(pos[:, 2]) > 1
    assert set(np.unique(want["is_mouse_dragging"])) == {0, 1}
    assert set(np.unique(want["color_mode"])) == {0, 1, 2}


def one_frame(cursor, screen=(1280.0, 720.0), **kw):
    inputs = np.zeros(1, C.INPUT_DTYPE)
    ui = np.zeros(1, C.UI_DTYPE)
    inputs[0]["delta_time"] = 0.016
    inputs[0]["cursor"] = cursor
    inputs[0]["screen"] = screen
    inputs[0]["primary_down"] = 1
    for k, v in kw.items():
        inputs[0][k] = v
    ui[0] = (0.0, 5.0, 10.0, 50.0, 0, 0)
    return inputs, ui


def test_projection_by_hand(tmp_path):
    """Start-up camera at (0,0,100) looking down -z, attractor at (0,0,48): the drag plane
    is 52 units ahead, its visible height 2*52*tan(30 deg), width = height * aspect."""
    # cursor at the screen centre: the attractor stays where it is
    for inputs, ui in (one_frame((640.0, 360.0)),):
        p = run_caller(tmp_path, inputs, ui)[0]
        assert np.allclose(p["mouse_position"], (0.0, 0.0, 48.0), atol=1e-4)
        assert p["is_mouse_dragging"] == 1 and p["damping"] == np.float32(0.99)
    # right edge, mid height: +x by half the plane width; top edge: +y by half its height
    half_h = 52.0 * math.tan(math.pi / 6)
    p = run_caller(tmp_path, *one_frame((1280.0, 360.0)))[0]
    assert np.allclose(p["mouse_position"], (half_h * 1280 / 720, 0.0, 48.0), rtol=1e-5, atol=1e-4)
    p = run_caller(tmp_path, *one_frame((640.0, 0.0)))[0]
    assert np.allclose(p["mouse_position"], (0.0, half_h, 48.0), rtol=1e-5, atol=1e-4)
    # scroll wheel: 0.2 units per scroll unit along the view direction (-z), no drag
    p = run_caller(tmp_path, *one_frame((640.0, 360.0), primary_down=0, scroll_y=10.0))[0]
    assert np.allclose(p["mouse_position"], (0.0, 0.0, 46.0), atol=1e-4)
    assert p["is_mouse_dragging"] == 0
    # W for one 16 ms frame flies 50 units/s forward; the drag plane keeps its depth
    p = run_caller(tmp_path, *one_frame((640.0, 360.0), keys_down=1))[0]
    assert np.allclose(p["mouse_position"], (0.0, 0.0, 48.0), atol=1e-4)


def test_statistics_panel_numbers(tmp_path):
    """Frames per second (over at least one second, counted on every frame) and the smoothed
    host time of the update call (only frames that updated): src/app.rs:209-216, 303-315,
    shown at app.rs:325-330.  The C++ mirror against the numpy restatement, bit for bit, and
    the first values by hand."""
    build()
    inputs, ui = synthetic_session(360, seed=5)
    rng = np.random.default_rng(5)
    update_ms = rng.uniform(0.004, 0.4, len(inputs)).astype(np.float32)
    fi, fu, fo, fm, fs = (str(tmp_path / n) for n in ("i.bin", "u.bin", "p.bin", "ms.bin", "s.bin"))
    inputs.tofile(fi)
    ui.tofile(fu)
    update_ms.tofile(fm)
    r = subprocess.run([CALLER, fi, fu, fo, fm, fs], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(fs, dtype=np.float32).reshape(-1, 2)
    want = C.frame_stats(inputs["delta_time"], ui["paused"], update_ms)
    assert got.shape == want.shape == (360, 2) and got.tobytes() == want.tobytes()
    # by hand: no rate before a second has passed, then frames / seconds of that window
    elapsed = np.cumsum(inputs["delta_time"].astype(np.float64))
    first = int(np.argmax(elapsed >= 1.0 - 1e-6))
    assert (got[:first - 1, 0] == 0).all() and got[first + 1, 0] > 0
    assert abs(got[first + 1, 0] - (first + 1) / elapsed[first]) < 1.0
    # the smoothed time starts from 0 with alpha = 0.1 and stands still while paused
    assert got[0, 1] == np.float32(0.1) * update_ms[0]
    assert (got[100:108, 1] == got[99, 1]).all() and got[108, 1] != got[107, 1]


def test_empty_session(tmp_path):
    got = run_caller(tmp_path, np.zeros(0, C.INPUT_DTYPE), np.zeros(0, C.UI_DTYPE))
    assert len(got) == 0 and len(C.replay(np.zeros(0, C.INPUT_DTYPE), np.zeros(0, C.UI_DTYPE))) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["frames", "batch"])
def test_session_drives_the_simulation(tmp_path, mode):
    build()
    inputs, ui = synthetic_session(150, seed=11)
    count = 200_003
    fi, fu, fo, fq = (str(tmp_path / n) for n in ("inputs.bin", "ui.bin", "params.bin", "q.bin"))
    inputs.tofile(fi)
    ui.tofile(fu)
    cmd = [REPLAY, fi, fu, fo, str(count), str(psim.GEN_UNIFORM), fq]
    r = subprocess.run(cmd + (["batch"] if mode == "batch" else []), capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    want_params = C.replay(inputs, ui)
    raw = np.fromfile(fo, dtype=np.uint8).reshape(-1, 48)
    assert raw.tobytes() == want_params.tobytes()
    state = T.oracle_gen_uniform_aos(0, count)
    for row in raw:
        state = T.oracle_update_aos(state, T.params_from_bytes(row), 1)
    got = np.fromfile(fq, dtype=psim.PARTICLE_DTYPE)
    T.assert_particles_equal(got, state, f"session replay ({mode})")
