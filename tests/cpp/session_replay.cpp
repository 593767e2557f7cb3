// session_replay.cpp -- replays a recorded input session through the caller-side mirror
// (tests/cpp/psim_caller.hpp; reference: src/app.rs:202-301, 528-576) and, when built with
// PSIM_WITH_DEVICE, feeds the resulting SimParams stream to the simulation.
//
//   caller_params  <inputs> <ui> <params_out> [<update_ms> <stats_out>]  (no GPU, links nothing)
//   session_replay <inputs> <ui> <params_out> <count> <generation> <particles_out> [batch]
//
// <inputs>: psim::FrameInput records (48 B each).  <ui>: one UiFrame per frame -- the
// slider / combo / pause-button state of that frame (app.rs:346-348, 407-437, 496-507).
// <params_out>: the SimParams of every frame that was not paused, in order.
// With "batch" the whole stream goes through update_batch() instead of one update() per
// frame; the particles must come out the same.
// <update_ms>: one float per frame, the host time the update call of that frame took (a
// recorded stand-in for the wall clock of app.rs:303-309); <stats_out> then receives, per
// frame, the two numbers of the statistics panel: {fps, smoothed update time}.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "psim_caller.hpp"
#ifdef PSIM_WITH_DEVICE
#include "psim.hpp"
#endif

struct UiFrame {
    float gravity, mouse_force, mouse_radius, max_dist_for_color;
    uint32_t color_mode, paused;
};
static_assert(sizeof(UiFrame) == 24, "UiFrame is a flat 24-byte record");

template <class T> static bool read_records(const char *path, std::vector<T> &out)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize(bytes > 0 ? static_cast<size_t>(bytes) / sizeof(T) : 0);
    const size_t got = out.empty() ? 0 : std::fread(out.data(), sizeof(T), out.size(), f);
    std::fclose(f);
    return got == out.size() && static_cast<size_t>(bytes) == out.size() * sizeof(T);
}

template <class T> static bool write_records(const char *path, const std::vector<T> &v)
{
    FILE *f = std::fopen(path, "wb");
    if (!f) return false;
    const size_t put = v.empty() ? 0 : std::fwrite(v.data(), sizeof(T), v.size(), f);
    return std::fclose(f) == 0 && put == v.size();
}

int main(int argc, char **argv)
{
#ifdef PSIM_WITH_DEVICE
    if (argc != 7 && argc != 8) {
        std::fprintf(stderr, "usage: %s inputs ui params_out count generation particles_out [batch]\n", argv[0]);
        return 2;
    }
#else
    if (argc != 4 && argc != 6) {
        std::fprintf(stderr, "usage: %s inputs ui params_out [update_ms stats_out]\n", argv[0]);
        return 2;
    }
#endif
    std::vector<psim::FrameInput> inputs;
    std::vector<UiFrame> ui;
    if (!read_records(argv[1], inputs) || !read_records(argv[2], ui) || inputs.size() != ui.size()) {
        std::fprintf(stderr, "cannot read the session (inputs and ui must hold one record per frame)\n");
        return 4;
    }

    psim::CallerState app;
    if (!inputs.empty())                       // the window's aspect at start-up (camera.rs:40, app.rs:100)
        app.camera.aspect = inputs[0].screen_width / inputs[0].screen_height;
    std::vector<psim_sim_params> used;
    std::vector<float> update_ms, shown;
#ifndef PSIM_WITH_DEVICE
    if (argc == 6 && (!read_records(argv[4], update_ms) || update_ms.size() != inputs.size())) {
        std::fprintf(stderr, "update_ms must hold one float per frame\n");
        return 4;
    }
#endif
    psim::FrameStats stats;
    for (size_t i = 0; i < inputs.size(); ++i) {
        app.gravity = ui[i].gravity;
        app.mouse_force = ui[i].mouse_force;
        app.mouse_radius = ui[i].mouse_radius;
        app.max_dist_for_color = ui[i].max_dist_for_color;
        app.color_mode = ui[i].color_mode;
        const psim_sim_params p = app.frame(inputs[i]);
        if (!ui[i].paused) used.push_back(p);  // app.rs:283: a paused frame still moves the camera and the attractor
        if (!update_ms.empty()) {
            stats.frame(inputs[i].delta_time);                       // app.rs:209-216: every frame
            if (!ui[i].paused) stats.update_took(update_ms[i]);      // app.rs:309-315: frames that updated
            shown.push_back(stats.fps);
            shown.push_back(stats.simulation_update_time);
        }
    }
    if (!write_records(argv[3], used)) return 4;
#ifndef PSIM_WITH_DEVICE
    if (argc == 6 && !write_records(argv[5], shown)) return 4;
#endif

#ifdef PSIM_WITH_DEVICE
    const uint32_t count = static_cast<uint32_t>(std::strtoul(argv[4], nullptr, 10));
    const int generation = std::atoi(argv[5]);
    const bool batch = argc == 8 && std::strcmp(argv[7], "batch") == 0;
    try {
        psim::HipParticleSimulation sim(count, static_cast<psim::SphereGeneration>(generation));
        if (batch) {
            sim.update_batch(used);
        } else {
            for (const psim_sim_params &p : used) sim.update(p);
        }
        sim.sync();
        if (!write_records(argv[6], sim.download())) return 4;
    } catch (const psim::Error &e) {
        std::fprintf(stderr, "psim error %d: %s\n", e.code(), e.what());
        return 1;
    }
#endif
    std::printf("frames=%zu updates=%zu\n", inputs.size(), used.size());
    return 0;
}
