// psim_caller.hpp -- replay / test infrastructure, not part of the library boundary: host-side mirror of the code that feeds the update path in the
// reference application: how one frame's input becomes one SimParams.
//
//   * the camera basis and its keyboard / pointer motion  (reference: src/camera.rs:106-174,
//     initial values src/camera.rs:72-82)
//   * per-frame input handling: drag flag, right-button camera rotation, scroll wheel =
//     attractor depth along the view direction            (reference: src/app.rs:528-576)
//   * keyboard camera motion, projection of the cursor onto the plane through the
//     attractor that faces the camera, SimParams assembly (reference: src/app.rs:202-301)
//   * the two figures the application shows about itself: frames per second and the smoothed
//     host time of the update call                        (reference: src/app.rs:209-216, 303-315)
//
// All arithmetic is fp32 in the reference's operation order (glam Vec3: dot =
// (x*x' + y*y') + z*z', normalize = v * (1 / length), cross as written below); compile
// host code that includes this header without floating-point contraction
// (-ffp-contract=off) for results identical to a build without fused multiply-adds.
// Header-only, no GPU calls: a headless driver replays a recorded FrameInput stream
// through CallerState::frame() and hands each SimParams to psim_update().
#ifndef PSIM_CALLER_HPP
#define PSIM_CALLER_HPP

#include <cmath>
#include <cstdint>

#include "psim.h"

namespace psim {

struct Vec3 {
    float x, y, z;
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b)
{
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
inline Vec3 normalize(Vec3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }

constexpr float kPi = 3.14159274101257324f;   // std::f32::consts::PI

// Key bits of FrameInput::keys_down, in the order the reference polls them (app.rs:219-225).
enum : uint32_t { KEY_W = 1u, KEY_S = 2u, KEY_A = 4u, KEY_D = 8u, KEY_SPACE = 16u };

// `struct Camera` without its GPU buffers (camera.rs:22-37, defaults camera.rs:72-82).
struct Camera {
    Vec3 position{0.0f, 0.0f, 100.0f};
    float yaw = -kPi / 2.0f;
    float pitch = 0.0f;
    float fov = kPi / 3.0f;
    float aspect = 1.0f;
    float movement_speed = 50.0f;
    float rotation_speed = 0.003f;

    // camera.rs:106-113
    Vec3 get_forward() const
    {
        return normalize(Vec3{std::cos(yaw) * std::cos(pitch), std::sin(pitch),
                              std::sin(yaw) * std::cos(pitch)});
    }
    // camera.rs:115-117
    Vec3 get_right() const { return normalize(cross(get_forward(), Vec3{0.0f, 1.0f, 0.0f})); }
    // camera.rs:119-121
    Vec3 get_up() const { return cross(get_right(), get_forward()); }

    // camera.rs:123-166; `key` is one KEY_* bit or 0 for "no key".
    void process_keyboard(uint32_t key, bool shift_down, float dt)
    {
        const Vec3 forward = get_forward();
        const Vec3 right = get_right();
        const Vec3 up{0.0f, 1.0f, 0.0f};
        const float speed = movement_speed * dt;
        switch (key) {
        case KEY_W: position = position + forward * speed; break;
        case KEY_S: position = position - forward * speed; break;
        case KEY_A: position = position - right * speed; break;
        case KEY_D: position = position + right * speed; break;
        case KEY_SPACE: position = position + up * speed; break;
        default:
            if (shift_down) position = position - up * speed;
        }
    }

    // camera.rs:168-174
    void process_mouse_movement(float dx, float dy)
    {
        yaw += dx * rotation_speed;
        const float lo = -kPi / 2.0f + 0.01f, hi = kPi / 2.0f - 0.01f;
        float p = pitch - dy * rotation_speed;
        if (p < lo) p = lo;            // f32::clamp: NaN passes through
        if (p > hi) p = hi;
        pitch = p;
    }
};

// What the windowing layer reports for one frame (egui's input state in the reference).
// 48 bytes, no padding: a recorded session is a flat array of these.
struct FrameInput {
    float delta_time;        // wall-clock time since the previous frame (app.rs:204-205)
    float cursor_x, cursor_y;        // pointer hover position in points (app.rs:542-544)
    float screen_width, screen_height;   // content rectangle (app.rs:246)
    float pointer_dx, pointer_dy;    // pointer delta this frame (app.rs:552)
    float scroll_y;                  // raw scroll delta (app.rs:561-562)
    uint32_t keys_down;              // KEY_* bits
    uint32_t shift_down;
    uint32_t primary_down;           // left button: drags the attractor (app.rs:547)
    uint32_t secondary_down;         // right button: rotates the camera (app.rs:548)
};
static_assert(sizeof(FrameInput) == 48, "FrameInput is a flat 48-byte record");

// The application state that SimParams is assembled from (app.rs:143-148 for the initial
// values; the sliders of app.rs:407-437 and the colour combo of app.rs:496-507 write the
// public members directly).
struct CallerState {
    Camera camera;
    float mouse_position[3] = {0.0f, 0.0f, 48.0f};
    float gravity = 0.0f;
    float mouse_force = 5.0f;
    float mouse_radius = 10.0f;
    float max_dist_for_color = 50.0f;
    uint32_t color_mode = 0;
    bool mouse_dragging = false;

    // One frame: input handling (app.rs:528-576), then the part of update_simulation
    // that precedes the update call (app.rs:218-301).  Returns the SimParams the
    // reference would pass to `update` for this frame.
    psim_sim_params frame(const FrameInput &in)
    {
        mouse_dragging = in.primary_down != 0;
        if (in.secondary_down && (in.pointer_dx != 0.0f || in.pointer_dy != 0.0f))
            camera.process_mouse_movement(in.pointer_dx, in.pointer_dy);
        if (in.scroll_y != 0.0f) {                                   // app.rs:561-575
            const Vec3 forward = camera.get_forward();
            const Vec3 now{mouse_position[0], mouse_position[1], mouse_position[2]};
            const Vec3 moved = now + forward * (in.scroll_y * 0.2f);
            mouse_position[0] = moved.x, mouse_position[1] = moved.y, mouse_position[2] = moved.z;
        }

        for (uint32_t key : {KEY_W, KEY_S, KEY_A, KEY_D, KEY_SPACE})         // app.rs:219-230
            if (in.keys_down & key) camera.process_keyboard(key, in.shift_down != 0, in.delta_time);
        if (in.shift_down) camera.process_keyboard(0, true, in.delta_time);  // app.rs:232-234

        if (mouse_dragging) {                                        // app.rs:245-280
            const float ndc_x = (2.0f * in.cursor_x / in.screen_width) - 1.0f;
            const float ndc_y = 1.0f - (2.0f * in.cursor_y / in.screen_height);
            const Vec3 forward = camera.get_forward();
            const Vec3 right = camera.get_right();
            const Vec3 up = camera.get_up();
            const Vec3 current{mouse_position[0], mouse_position[1], mouse_position[2]};
            const float distance = dot(current - camera.position, forward);
            const Vec3 plane_center = camera.position + forward * distance;
            const float height = 2.0f * distance * std::tan(camera.fov / 2.0f);
            const float width = height * camera.aspect;
            const Vec3 world = plane_center + right * (ndc_x * width / 2.0f)
                               + up * (ndc_y * height / 2.0f);
            mouse_position[0] = world.x, mouse_position[1] = world.y, mouse_position[2] = world.z;
        }

        psim_sim_params p{};                                         // app.rs:290-301
        p.delta_time = in.delta_time;
        p.gravity = gravity;
        p.color_mode = color_mode;
        p.mouse_force = mouse_force;
        p.mouse_radius = mouse_radius;
        p.is_mouse_dragging = mouse_dragging ? 1u : 0u;
        p.damping = 0.99f;
        p.max_dist_for_color = max_dist_for_color;
        p.mouse_position[0] = mouse_position[0];
        p.mouse_position[1] = mouse_position[1];
        p.mouse_position[2] = mouse_position[2];
        p._padding2 = 0;

        // The aspect ratio follows the drawing rectangle only after the update of the
        // same frame (app.rs:587-592), so a window resize reaches the projection one
        // frame late -- mirrored here.
        const float aspect_ratio = in.screen_width / in.screen_height;
        if (std::fabs(aspect_ratio - camera.aspect) > 0.001f) camera.aspect = aspect_ratio;
        return p;
    }
};

// The statistics panel's two numbers (app.rs:325-330; initial values app.rs:151-155).
struct FrameStats {
    float fps = 0.0f;
    uint32_t fps_counter = 0;
    float fps_timer = 0.0f;
    float simulation_update_time = 0.0f;   // milliseconds, exponentially smoothed

    // Every frame, paused or not (app.rs:209-216): frames over at least one second.
    void frame(float delta_time)
    {
        fps_counter += 1;
        fps_timer += delta_time;
        if (fps_timer >= 1.0f) {
            fps = static_cast<float>(fps_counter) / fps_timer;
            fps_counter = 0;
            fps_timer = 0.0f;
        }
    }

    // After an update call that took `update_time_ms` of host time (app.rs:309-315).  For an
    // asynchronous backend that is the time to enqueue the step, as for the reference's
    // compute backend; psim_timer_start / psim_timer_stop give the device's time instead.
    void update_took(float update_time_ms)
    {
        const float ALPHA = 0.1f;
        simulation_update_time = (1.0f - ALPHA) * simulation_update_time + ALPHA * update_time_ms;
    }
};

}  // namespace psim

#endif  // PSIM_CALLER_HPP
