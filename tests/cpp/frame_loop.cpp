// frame_loop.cpp -- a headless caller shaped like the reference application's frame
// loop (reference: src/app.rs:202-316): per frame it assembles SimParams (variable
// delta_time, UI values, the dragged attractor position), skips the update while
// paused (app.rs:283) and otherwise calls update through the trait-shaped C++ mirror.
//
// Built with a plain host compiler against include/psim.hpp + libpsim_hip.so.
// It records the SimParams it used, frame by frame, and the final particles, so the
// test-suite can replay the same stream through the CPU oracle.  Half way through, the
// state takes a detour through a checkpoint file.
//
// usage: frame_loop <count> <generation 0|1|2> <frames> <params_out> <particles_out>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "psim.hpp"
#include "psim_caller.hpp"

int main(int argc, char **argv)
{
    if (argc != 6) {
        std::fprintf(stderr, "usage: %s count generation frames params_out particles_out\n", argv[0]);
        return 2;
    }
    const uint32_t count = static_cast<uint32_t>(std::strtoul(argv[1], nullptr, 10));
    const int generation = std::atoi(argv[2]);
    const int frames = std::atoi(argv[3]);
    try {
        psim::HipParticleSimulation sim(count, static_cast<psim::SphereGeneration>(generation));
        if (sim.get_method() != psim::SimulationMethod::Hip) return 3;

        // UI state as the app initialises it (app.rs:143-148, 298).
        float gravity = 0.0f, mouse_force = 5.0f, mouse_radius = 10.0f, max_dist = 50.0f;
        float attractor[3] = {0.0f, 0.0f, 48.0f};
        uint32_t color_mode = 0;
        std::vector<psim::SimParams> used;
        psim::FrameStats stats;                                // app.rs:209-216, 303-315

        for (int frame = 0; frame < frames; ++frame) {
            // a frame-time pattern standing in for the wall clock (app.rs:204-207)
            const float frame_ms[4] = {16.0f, 17.5f, 15.25f, 33.0f};
            const float delta_time = frame_ms[frame % 4] / 1000.0f;
            stats.frame(delta_time);

            // scripted "user": drags the attractor on a square path, toggles things
            const bool dragging = (frame / 8) % 4 != 3;
            if (dragging) {
                const int leg = (frame / 16) % 4;
                const float step = 0.75f;
                if (leg == 0) attractor[0] += step;
                if (leg == 1) attractor[1] += step;
                if (leg == 2) attractor[0] -= step;
                if (leg == 3) attractor[1] -= step;
                attractor[2] -= 0.5f;                          // scroll: depth (app.rs:560-575)
            }
            if (frame == 20) gravity = 2.5f;                   // sliders (app.rs:407-437)
            if (frame == 30) { mouse_radius = 35.0f; mouse_force = 60.0f; }
            if (frame == 40) color_mode = 1;                   // colour combo (app.rs:496-507)
            if (frame == 50) color_mode = 2;
            if (frame == 60) color_mode = 0;
            if (frame == 44) sim.set_paused(true);             // pause button (app.rs:346-348)
            if (frame == 48) sim.set_paused(false);

            if (frame == frames / 2) {
                // stop and resume: the state goes to a file, the simulation is reset, the
                // file comes back -- the run must end exactly where it would have anyway
                const std::string ckp = std::string(argv[5]) + ".ckp";
                sim.save_checkpoint(ckp, static_cast<uint64_t>(frame));
                sim.reset(psim::SphereGeneration::Uniform);
                if (sim.load_checkpoint(ckp) != static_cast<uint64_t>(frame)) return 6;
                std::remove(ckp.c_str());
            }

            if (!sim.is_paused()) {                            // app.rs:283
                psim::SimParams p = psim::default_sim_params();
                p.delta_time = delta_time;                     // app.rs:290-301
                p.gravity = gravity;
                p.color_mode = color_mode;
                p.mouse_force = mouse_force;
                p.mouse_radius = mouse_radius;
                p.mouse_position[0] = attractor[0];
                p.mouse_position[1] = attractor[1];
                p.mouse_position[2] = attractor[2];
                p.is_mouse_dragging = dragging ? 1u : 0u;
                p.damping = 0.99f;
                p.max_dist_for_color = max_dist;
                const auto update_start = std::chrono::steady_clock::now();   // app.rs:303
                sim.update(p);                                 // app.rs:306-307
                stats.update_took(std::chrono::duration<float, std::milli>(
                                      std::chrono::steady_clock::now() - update_start).count());
                used.push_back(p);
            }
        }
        sim.sync();
        const std::vector<psim::Particle> out = sim.download();
        // the same frame through a page-locked array, enqueued and awaited
        psim::PinnedParticles frame(out.size());
        sim.download_async(frame);
        sim.sync();
        if (!out.empty() && std::memcmp(frame.data(), out.data(), out.size() * sizeof(psim::Particle)) != 0)
            return 5;

        FILE *f = std::fopen(argv[4], "wb");
        if (!f) return 4;
        std::fwrite(used.data(), sizeof(psim::SimParams), used.size(), f);
        std::fclose(f);
        f = std::fopen(argv[5], "wb");
        if (!f) return 4;
        std::fwrite(out.data(), sizeof(psim::Particle), out.size(), f);
        std::fclose(f);
        std::printf("frames=%d updates=%zu particles=%zu\n", frames, used.size(), out.size());
        // what the reference's statistics panel would show (app.rs:325-330)
        std::printf("fps=%.1f update_time_ms=%.4f\n", stats.fps, stats.simulation_update_time);
    } catch (const psim::Error &e) {
        std::fprintf(stderr, "psim error %d: %s\n", e.code(), e.what());
        return 1;
    }
    return 0;
}
