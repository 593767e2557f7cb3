"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU restatement, in numpy float32 scalars, of how the reference application turns one
frame of input into one SimParams (reference: src/app.rs:202-301 and :528-576, camera
basis and motion src/camera.rs:72-82, 106-174).  It checks tests/cpp/psim_caller.hpp.
cos/sin/tan go through the C library's cosf/sinf/tanf, which is what Rust's f32::cos etc.
and C++'s std::cos(float) call on this platform.

PARITY UNPINNED BY THE REFERENCE: the reference holds no tests or recorded sessions for
this code, and its glam Vec3 arithmetic (cross, dot, normalize) is restated from the
published behaviour of glam 0.30.10.
"""
import ctypes
import ctypes.util

import numpy as np

F = np.float32
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _name in ("cosf", "sinf", "tanf"):
    getattr(_libm, _name).restype = ctypes.c_float
    getattr(_libm, _name).argtypes = [ctypes.c_float]


def cosf(x): return F(_libm.cosf(float(x)))
def sinf(x): return F(_libm.sinf(float(x)))
def tanf(x): return F(_libm.tanf(float(x)))


PI = F(np.pi)
KEYS = (1, 2, 4, 8, 16)          # W, S, A, D, Space: app.rs:219-225

INPUT_DTYPE = np.dtype([("delta_time", "<f4"), ("cursor", "<f4", 2), ("screen", "<f4", 2),
                        ("pointer_delta", "<f4", 2), ("scroll_y", "<f4"), ("keys_down", "<u4"),
                        ("shift_down", "<u4"), ("primary_down", "<u4"), ("secondary_down", "<u4")])
UI_DTYPE = np.dtype([("gravity", "<f4"), ("mouse_force", "<f4"), ("mouse_radius", "<f4"),
                     ("max_dist_for_color", "<f4"), ("color_mode", "<u4"), ("paused", "<u4")])
PARAMS_DTYPE = np.dtype([("delta_time", "<f4"), ("gravity", "<f4"), ("color_mode", "<u4"),
                         ("mouse_force", "<f4"), ("mouse_radius", "<f4"),
                         ("is_mouse_dragging", "<u4"), ("damping", "<f4"),
                         ("max_dist_for_color", "<f4"), ("mouse_position", "<f4", 3),
                         ("_padding2", "<u4")])
assert INPUT_DTYPE.itemsize == 48 and UI_DTYPE.itemsize == 24 and PARAMS_DTYPE.itemsize == 48


def v_add(a, b): return (a[0] + b[0], a[1] + b[1], a[2] + b[2])
def v_sub(a, b): return (a[0] - b[0], a[1] - b[1], a[2] - b[2])
def v_scale(a, s): return (a[0] * s, a[1] * s, a[2] * s)
def v_dot(a, b): return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def v_cross(a, b):
    return (a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1])


def v_normalize(a):
    with np.errstate(all="ignore"):
        return v_scale(a, F(1.0) / np.sqrt(v_dot(a, a)))


Y = (F(0), F(1), F(0))


class Camera:
    def __init__(self, aspect):                                      # camera.rs:72-82
        self.position = (F(0), F(0), F(100))
        self.yaw = -PI / F(2)
        self.pitch = F(0)
        self.fov = PI / F(3)
        self.aspect = F(aspect)
        self.movement_speed = F(50)
        self.rotation_speed = F(0.003)

    def forward(self):                                               # camera.rs:106-113
        cp = cosf(self.pitch)
        return v_normalize((cosf(self.yaw) * cp, sinf(self.pitch), sinf(self.yaw) * cp))

    def right(self):                                                 # camera.rs:115-117
        return v_normalize(v_cross(self.forward(), Y))

    def up(self):                                                    # camera.rs:119-121
        return v_cross(self.right(), self.forward())

    def keyboard(self, key, shift, dt):                              # camera.rs:123-166
        fwd, right = self.forward(), self.right()
        speed = self.movement_speed * dt
        if key == 1:
            self.position = v_add(self.position, v_scale(fwd, speed))
        elif key == 2:
            self.position = v_sub(self.position, v_scale(fwd, speed))
        elif key == 4:
            self.position = v_sub(self.position, v_scale(right, speed))
        elif key == 8:
            self.position = v_add(self.position, v_scale(right, speed))
        elif key == 16:
            self.position = v_add(self.position, v_scale(Y, speed))
        elif shift:
            self.position = v_sub(self.position, v_scale(Y, speed))

    def mouse(self, dx, dy):                                         # camera.rs:168-174
        self.yaw = self.yaw + dx * self.rotation_speed
        lo, hi = -PI / F(2) + F(0.01), PI / F(2) - F(0.01)
        p = self.pitch - dy * self.rotation_speed
        if p < lo:
            p = lo
        if p > hi:
            p = hi
        self.pitch = p


def replay(inputs, ui):
    """inputs: INPUT_DTYPE array, ui: UI_DTYPE array (one record per frame) -> PARAMS_DTYPE
    array holding the SimParams of the frames that were not paused."""
    assert len(inputs) == len(ui)
    out = []
    if len(inputs) == 0:
        return np.zeros(0, PARAMS_DTYPE)
    with np.errstate(all="ignore"):
        cam = Camera(inputs[0]["screen"][0] / inputs[0]["screen"][1])
        attractor = (F(0), F(0), F(48))                              # app.rs:147
        for fin, u in zip(inputs, ui):
            dt = fin["delta_time"]
            dragging = bool(fin["primary_down"])                     # app.rs:547
            dx, dy = fin["pointer_delta"]
            if fin["secondary_down"] and (dx != 0 or dy != 0):       # app.rs:548-558
                cam.mouse(dx, dy)
            if fin["scroll_y"] != 0:                                 # app.rs:561-575
                attractor = v_add(attractor, v_scale(cam.forward(), fin["scroll_y"] * F(0.2)))
            for key in KEYS:                                         # app.rs:219-230
                if fin["keys_down"] & key:
                    cam.keyboard(key, bool(fin["shift_down"]), dt)
            if fin["shift_down"]:                                    # app.rs:232-234
                cam.keyboard(0, True, dt)
            if dragging:                                             # app.rs:245-280
                w, h = fin["screen"]
                x, y = fin["cursor"]
                ndc_x = (F(2) * x / w) - F(1)
                ndc_y = F(1) - (F(2) * y / h)
                fwd, right, up = cam.forward(), cam.right(), cam.up()
                distance = v_dot(v_sub(attractor, cam.position), fwd)
                centre = v_add(cam.position, v_scale(fwd, distance))
                height = F(2) * distance * tanf(cam.fov / F(2))
                width = height * cam.aspect
                attractor = v_add(v_add(centre, v_scale(right, ndc_x * width / F(2))),
                                  v_scale(up, ndc_y * height / F(2)))
            if not u["paused"]:                                      # app.rs:283-301
                p = np.zeros((), PARAMS_DTYPE)
                p["delta_time"] = dt
                p["gravity"] = u["gravity"]
                p["color_mode"] = u["color_mode"]
                p["mouse_force"] = u["mouse_force"]
                p["mouse_radius"] = u["mouse_radius"]
                p["is_mouse_dragging"] = 1 if dragging else 0
                p["damping"] = F(0.99)
                p["max_dist_for_color"] = u["max_dist_for_color"]
                p["mouse_position"] = attractor
                out.append(p)
            aspect = fin["screen"][0] / fin["screen"][1]             # app.rs:587-592
            if abs(aspect - cam.aspect) > F(0.001):
                cam.aspect = aspect
    return np.array(out, dtype=PARAMS_DTYPE) if out else np.zeros(0, PARAMS_DTYPE)


def frame_stats(delta_times, paused, update_ms):
    """The two numbers of the application's statistics panel after every frame (reference:
    src/app.rs:209-216 frames per second over at least one second, counted on every frame;
    src/app.rs:309-315 the host time of the update call smoothed with alpha = 0.1, only on
    frames that updated; initial values src/app.rs:151-155) -> float32 array [frames, 2]."""
    fps, counter, timer, smoothed = F(0), 0, F(0), F(0)
    alpha = F(0.1)
    out = np.zeros((len(delta_times), 2), dtype=np.float32)
    for k, (dt, p, ms) in enumerate(zip(delta_times, paused, update_ms)):
        counter += 1
        timer = timer + F(dt)
        if timer >= F(1.0):
            fps = F(counter) / timer
            counter, timer = 0, F(0)
        if not p:
            smoothed = (F(1.0) - alpha) * smoothed + alpha * F(ms)
        out[k] = (fps, smoothed)
    return out
