// sharded_loop.cpp -- one rank of a headless, sharded run through the C++ mirror
// (include/psim.hpp): a shard of a larger set stepped through the attractor exchange
// (psim::HipAttractorExchange, all three pacings), through the list step that allocates nothing,
// a detour through a checkpoint written in the sharded way (begin / save / commit), and the
// occupancy and device-count queries.  It records every step it took -- the parameters and the
// attractor list that applied -- so the test-suite can replay the run through the CPU oracle.
//
// Built with a plain host compiler against include/psim.hpp + libpsim_hip.so.
// usage: sharded_loop <count> <first> <total> <steps> <log_out> <particles_out>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "psim.hpp"

namespace {

psim::Attractor attractor_at(int step, int k)
{
    psim::Attractor a{};
    a.position[0] = -30.0f + 0.5f * static_cast<float>(step % 97) + 7.0f * static_cast<float>(k);
    a.position[1] = 20.0f - 0.25f * static_cast<float>(step % 61);
    a.position[2] = -10.0f + 3.0f * static_cast<float>(k);
    a.force = 40.0f + static_cast<float>(k);
    a.radius = 18.0f;
    a.active = 1;
    return a;
}

void record(FILE *log, const psim::SimParams &p, const std::vector<psim::Attractor> &list)
{
    const uint32_t n = static_cast<uint32_t>(list.size());
    std::fwrite(&p, sizeof(p), 1, log);
    std::fwrite(&n, sizeof(n), 1, log);
    std::fwrite(list.data(), sizeof(psim::Attractor), list.size(), log);
}

}  // namespace

int main(int argc, char **argv)
{
    if (argc != 7) {
        std::fprintf(stderr, "usage: %s count first total steps log_out particles_out\n", argv[0]);
        return 2;
    }
    const uint32_t count = static_cast<uint32_t>(std::strtoul(argv[1], nullptr, 10));
    const uint64_t first = std::strtoull(argv[2], nullptr, 10);
    const uint64_t total = std::strtoull(argv[3], nullptr, 10);
    const int steps = std::atoi(argv[4]);
    try {
        if (psim::device_count() < 1) return 3;
        psim::HipParticleSimulation sim(count, first, total, psim::SphereGeneration::Uniform, 0);
        const uint32_t resident = sim.launch_occupancy(1, 0);
        if (resident == 0) return 4;
        FILE *log = std::fopen(argv[5], "wb");
        if (!log) return 5;

        psim::SimParams p = psim::default_sim_params();
        p.gravity = 0.75f;
        p.is_mouse_dragging = 0;                       // the lists act, not the SimParams attractor
        {
            psim::HipAttractorExchange exchange(sim, psim::HipAttractorExchange::unique_id(), 0, 1);
            if (exchange.world() != 1 || exchange.rank() != 0) return 6;
            for (int t = 0; t < steps; ++t) {
                if (t == steps / 4) exchange.set_pacing(PSIM_EXCHANGE_HOST_PACED);
                if (t == steps / 2) exchange.set_pacing(PSIM_EXCHANGE_PIPELINED);   // one call late
                if (t == 3 * steps / 4) exchange.set_pacing(PSIM_EXCHANGE_DEVICE_PACED);
                p.delta_time = 0.016f + 0.001f * static_cast<float>(t % 5);
                p.color_mode = static_cast<uint32_t>(t % 3);
                const psim::Attractor mine = attractor_at(t, 0);
                exchange.step(p, mine);
                record(log, p, {mine});
            }
            exchange.sync();
            const std::vector<psim::Attractor> last = exchange.last_list();
            if (last.size() != 1 || last[0].position[0] != attractor_at(steps - 1, 0).position[0])
                return 7;
        }

        // the state goes to a file the way several shards write one: no header until commit
        const std::string ckp = std::string(argv[6]) + ".ckp";
        psim::HipParticleSimulation::checkpoint_begin(ckp);
        sim.save_checkpoint(ckp);
        if (first == 0 && count == total) {            // a whole set: this shard completes the file
            sim.commit_checkpoint(ckp, static_cast<uint64_t>(steps));
            sim.reset(psim::SphereGeneration::Uniform);
            if (sim.load_checkpoint(ckp) != static_cast<uint64_t>(steps)) return 8;
        }
        std::remove(ckp.c_str());

        // lists of three attractors through the static device array
        for (int t = 0; t < steps; ++t) {
            p.delta_time = 0.016f;
            p.color_mode = 0;
            const std::vector<psim::Attractor> list = {attractor_at(t, 0), attractor_at(t, 1),
                                                       attractor_at(t, 2)};
            sim.update_multi_noalloc(p, list);
            record(log, p, list);
        }
        sim.sync();
        std::fclose(log);

        const std::vector<psim::Particle> out = sim.download();
        FILE *f = std::fopen(argv[6], "wb");
        if (!f) return 5;
        std::fwrite(out.data(), sizeof(psim::Particle), out.size(), f);
        std::fclose(f);
        std::printf("steps=%d particles=%zu resident=%u devices=%d\n", 2 * steps, out.size(),
                    resident, psim::device_count());
    } catch (const psim::Error &e) {
        std::fprintf(stderr, "psim error %d: %s\n", e.code(), e.what());
        return 1;
    }
    return 0;
}
