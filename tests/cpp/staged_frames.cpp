// staged_frames.cpp -- the frame loop of a caller that stages every frame through host memory
// for a renderer, as the reference's CPU backend does (src/simulation/cpu.rs:116-120:
// queue.write_buffer of the whole particle array per frame) and as rust/hip_backend.rs does
// for the wgpu vertex buffer.  Two ways of doing it, with the same calls the Rust binding
// makes:
//
//   blocking   per frame: psim_update, psim_download_aos (waits), hand the records over
//   pipelined  per frame: psim_update and psim_download_aos_async of frame k into one of two
//              page-locked buffers, THEN psim_download_aos_wait for frame k-1 in the other and
//              hand that over while the device works on frame k (one frame of latency)
//
// "Handing over" is what wgpu's queue.write_buffer does with the slice: a host copy into
// another buffer (here a plain allocation standing in for wgpu's staging memory).
//
// The program checks that the pipelined loop delivers exactly the frames of the blocking loop,
// each one frame later, and prints the frame rates of both.
//
// usage: staged_frames <particles> <frames> [check|time|nocopy]
//   check   hash every delivered frame and compare the two loops (default)
//   time    no hashing: the frame rates with the renderer-side copy
//   nocopy  no hashing and no renderer-side copy: what the library and the host link alone allow
// prints: blocking_fps pipelined_fps blocking_wait_ms pipelined_wait_ms
//         (the waits: host time per frame spent inside the download / the wait call)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "psim.hpp"

namespace {

double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// 64-bit FNV-1a over the frame as 8-byte words.
uint64_t frame_hash(const psim::Particle *p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    const uint64_t *w = reinterpret_cast<const uint64_t *>(p);
    for (size_t i = 0; i < n * sizeof(psim::Particle) / 8; ++i) h = (h ^ w[i]) * 1099511628211ull;
    return h;
}

psim::SimParams frame_params(int frame)
{
    psim::SimParams p = psim::default_sim_params();
    const float frame_ms[4] = {16.0f, 17.5f, 15.25f, 33.0f};     // app.rs:204-207: variable
    p.delta_time = frame_ms[frame % 4] / 1000.0f;
    p.is_mouse_dragging = 1u;
    p.mouse_position[0] = -20.0f + 0.5f * static_cast<float>(frame % 80);
    p.mouse_position[2] = 30.0f;
    p.mouse_radius = 25.0f;
    p.mouse_force = 40.0f;
    p.color_mode = (frame / 10) % 3;                             // the colours travel too
    return p;
}

}  // namespace

int main(int argc, char **argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s particles frames [check|time|nocopy]\n", argv[0]);
        return 2;
    }
    const uint32_t n = static_cast<uint32_t>(std::strtoul(argv[1], nullptr, 10));
    const int frames = std::atoi(argv[2]);
    const bool check = argc < 4 || std::strcmp(argv[3], "check") == 0;
    const bool copy = argc < 4 || std::strcmp(argv[3], "nocopy") != 0;
    try {
        psim::HipParticleSimulation sim(n, psim::SphereGeneration::Hollow);
        psim::PinnedParticles slot0(n), slot1(n);
        psim::PinnedParticles *slots[2] = {&slot0, &slot1};
        std::vector<psim::Particle> vertex_buffer(n);            // stands in for wgpu's copy
        const size_t bytes = static_cast<size_t>(n) * sizeof(psim::Particle);
        std::vector<uint64_t> blocking, pipelined;

        // ---- blocking: what the frame loop did before ----------------------------------------
        double waited = 0.0;
        double t0 = now_ms();
        for (int k = 0; k < frames; ++k) {
            sim.update(frame_params(k));
            const double w0 = now_ms();
            sim.download(slot0);
            waited += now_ms() - w0;
            if (copy) std::memcpy(vertex_buffer.data(), slot0.data(), bytes);
            if (check) blocking.push_back(frame_hash(vertex_buffer.data(), n));
        }
        const double blocking_ms = (now_ms() - t0) / frames, blocking_wait = waited / frames;

        // ---- pipelined: one frame ahead -------------------------------------------------------
        sim.reset(psim::SphereGeneration::Hollow);
        waited = 0.0;
        int in_flight = -1;
        t0 = now_ms();
        for (int k = 0; k < frames; ++k) {
            sim.update(frame_params(k));
            sim.download_async(*slots[k & 1]);                   // frame k is on its way
            if (in_flight >= 0) {                                // frame k-1 has had a frame's time
                const double w0 = now_ms();
                sim.download_wait(*slots[in_flight]);
                waited += now_ms() - w0;
                if (copy) std::memcpy(vertex_buffer.data(), slots[in_flight]->data(), bytes);
                if (check) pipelined.push_back(frame_hash(vertex_buffer.data(), n));
            }
            in_flight = k & 1;
        }
        const double pipelined_ms = (now_ms() - t0) / frames, pipelined_wait = waited / frames;
        if (in_flight >= 0) {                                    // the last frame, flushed
            sim.download_wait(*slots[in_flight]);
            std::memcpy(vertex_buffer.data(), slots[in_flight]->data(), bytes);
            if (check) pipelined.push_back(frame_hash(vertex_buffer.data(), n));
        }
        // a wait without a download in flight is refused, not a hang
        if (psim_download_aos_wait(sim.handle(), slot0.data()) != PSIM_ERR_INVALID) return 4;

        if (check) {
            if (pipelined.size() != blocking.size()) return 3;
            for (size_t k = 0; k < blocking.size(); ++k)
                if (pipelined[k] != blocking[k]) {
                    std::fprintf(stderr, "frame %zu differs between the two loops\n", k);
                    return 3;
                }
        }
        std::printf("%.1f %.1f %.4f %.4f\n", 1000.0 / blocking_ms, 1000.0 / pipelined_ms,
                    blocking_wait, pipelined_wait);
    } catch (const psim::Error &e) {
        std::fprintf(stderr, "psim error %d: %s\n", e.code(), e.what());
        return 1;
    }
    return 0;
}
