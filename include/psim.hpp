// psim.hpp -- C++ host-side mirror of the reference's `trait ParticleSimulation`
// (reference: src/simulation/mod.rs:21-50) over the C ABI of psim.h.
//
// Method names and argument meaning follow the trait; the wgpu device / queue /
// encoder / surface-format arguments are dropped.  Like the reference's methods
// (which return nothing and abort on a wgpu validation failure), the methods here do
// not return status codes: a failing call throws psim::Error carrying the C ABI's
// status and message.  Header-only; link against libpsim_hip.so.
#ifndef PSIM_HPP
#define PSIM_HPP

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "psim.h"

namespace psim {

// `enum SimulationMethod` (mod.rs:9-13) plus this backend.
enum class SimulationMethod { Cpu = PSIM_METHOD_CPU, ComputeShader = PSIM_METHOD_COMPUTE_SHADER,
                              Hip = PSIM_METHOD_HIP };

// `enum SphereGeneration` (mod.rs:15-19) plus the synthetic benchmark state.
enum class SphereGeneration { Hollow = PSIM_GEN_HOLLOW, Filled = PSIM_GEN_FILLED,
                              Uniform = PSIM_GEN_UNIFORM };

using SimParams = psim_sim_params;   // mod.rs:52-67
using Particle = psim_particle;      // mod.rs:86-98
using Attractor = psim_attractor;

// `impl Default for SimParams` (mod.rs:69-84).
inline SimParams default_sim_params()
{
    SimParams p{};
    p.delta_time = 0.016f;
    p.gravity = 0.0f;
    p.color_mode = 0;
    p.mouse_force = 5.0f;
    p.mouse_radius = 10.0f;
    p.is_mouse_dragging = 0;
    p.damping = 0.99f;
    p.max_dist_for_color = 50.0f;
    p.mouse_position[0] = p.mouse_position[1] = p.mouse_position[2] = 0.0f;
    p._padding2 = 0;
    return p;
}

class Error : public std::runtime_error {
public:
    Error(int code, const std::string &what) : std::runtime_error(what), code_(code) {}
    int code() const { return code_; }

private:
    int code_;
};

// Page-locked host array of 64-byte records: the host-side `Vec<Particle>` of a caller
// that stages frames through host memory (cpu.rs:8, :116-120), at the host link's rate.
class PinnedParticles {
public:
    explicit PinnedParticles(size_t n) : n_(n)
    {
        void *p = nullptr;
        const int rc = psim_host_alloc(&p, static_cast<uint64_t>(n) * sizeof(Particle));
        if (rc != PSIM_OK) throw Error(rc, psim_last_error());
        p_ = static_cast<Particle *>(p);
    }
    ~PinnedParticles() { psim_host_free(p_); }
    PinnedParticles(const PinnedParticles &) = delete;
    PinnedParticles &operator=(const PinnedParticles &) = delete;
    Particle *data() { return p_; }
    const Particle *data() const { return p_; }
    size_t size() const { return n_; }
    Particle &operator[](size_t i) { return p_[i]; }
    const Particle &operator[](size_t i) const { return p_[i]; }

private:
    Particle *p_ = nullptr;
    size_t n_ = 0;
};

class HipParticleSimulation {
public:
    // `fn new(device, initial_particle_count, surface_format, generation_mode) -> Self`
    HipParticleSimulation(uint32_t initial_particle_count, SphereGeneration generation_mode,
                          int device_id = 0)
    {
        check(psim_create(&h_, initial_particle_count, static_cast<int>(generation_mode),
                          device_id));
    }

    // One shard [global_first, global_first + count) of a `global_count` set.
    HipParticleSimulation(uint32_t count, uint64_t global_first, uint64_t global_count,
                          SphereGeneration generation_mode, int device_id)
    {
        check(psim_create_shard(&h_, count, global_first, global_count,
                                static_cast<int>(generation_mode), device_id));
    }

    ~HipParticleSimulation() { psim_destroy(h_); }
    HipParticleSimulation(const HipParticleSimulation &) = delete;
    HipParticleSimulation &operator=(const HipParticleSimulation &) = delete;
    HipParticleSimulation(HipParticleSimulation &&o) noexcept : h_(std::exchange(o.h_, nullptr)) {}

    // `fn update(&mut self, device, queue, encoder, params: &SimParams)`: asynchronous.
    void update(const SimParams &params) { check(psim_update(h_, &params)); }

    // Several consecutive steps in one pass over the state (same result as update() each).
    void update_batch(const std::vector<SimParams> &params)
    {
        check(psim_update_batch(h_, params.data(), static_cast<uint32_t>(params.size())));
    }

    // Multi-attractor extension, list in host memory or device memory (<= 64 attractors).
    void update_multi(const SimParams &params, const std::vector<Attractor> &attractors)
    {
        check(psim_update_multi_host(h_, &params, attractors.data(),
                                     static_cast<uint32_t>(attractors.size())));
    }
    void update_multi_device(const SimParams &params, const Attractor *attractors_device,
                             uint32_t n)
    {
        check(psim_update_multi(h_, &params, attractors_device, n));
    }

    // update_multi for callers that cannot allocate (HIP graph capture, serving loops): the
    // list goes through the library's static 2 KiB device array; <= 64 attractors.
    void update_multi_noalloc(const SimParams &params, const std::vector<Attractor> &attractors)
    {
        check(psim_update_multi_host_noalloc(h_, &params, attractors.data(),
                                             static_cast<uint32_t>(attractors.size())));
    }

    void sync() { check(psim_sync(h_)); }

    // Workgroups of this simulation's update kernel that one CU holds at once, as the runtime
    // reports it (the residency cap is worth a fifth of the update's bandwidth).
    uint32_t launch_occupancy(uint32_t n_attractors = 1, uint32_t color_mode = 0)
    {
        uint32_t workgroups = 0;
        check(psim_launch_occupancy(h_, n_attractors, color_mode, &workgroups, nullptr));
        return workgroups;
    }

    // `fn resize_buffer(&mut self, device, queue, new_count, generation_mode)`
    void resize_buffer(uint32_t new_count, SphereGeneration generation_mode)
    {
        check(psim_resize(h_, new_count, static_cast<int>(generation_mode)));
    }

    // `fn get_particle_buffer(&self) -> &wgpu::Buffer`: the device arrays.
    psim_soa_view get_particle_buffer()
    {
        psim_soa_view v{};
        check(psim_device_soa(h_, &v));
        return v;
    }

    // `fn get_method(&self) -> SimulationMethod`
    SimulationMethod get_method() const { return static_cast<SimulationMethod>(psim_method(h_)); }

    // `fn get_particle_count(&self) -> u32`
    uint32_t get_particle_count() const { return psim_particle_count(h_); }

    // `fn reset(&mut self, device, queue, generation_mode)`
    void reset(SphereGeneration generation_mode)
    {
        check(psim_reset(h_, static_cast<int>(generation_mode)));
    }

    // Colour of a step in a mode other than 1/2: PSIM_COLOR_KEEP (cpu.rs:106, default) or
    // PSIM_COLOR_RESET (compute.wgsl:75-77, 90-92).
    void set_color_semantics(int semantics) { check(psim_set_color_semantics(h_, semantics)); }
    int get_color_semantics() const { return psim_color_semantics(h_); }

    // What resize_buffer does: PSIM_RESIZE_REGENERATE (compute.rs:146-193, default) or
    // PSIM_RESIZE_APPEND (cpu.rs:123-158).
    void set_resize_semantics(int semantics) { check(psim_set_resize_semantics(h_, semantics)); }
    int get_resize_semantics() const { return psim_resize_semantics(h_); }

    // `fn is_paused(&self) -> bool` / `fn set_paused(&mut self, paused: bool)`
    bool is_paused() const { return psim_is_paused(h_) != 0; }
    void set_paused(bool paused) { psim_set_paused(h_, paused ? 1 : 0); }

    // The reference's 64-byte records, to and from host memory.
    std::vector<Particle> download()
    {
        std::vector<Particle> out(get_particle_count());
        check(psim_download_aos(h_, out.data(), 0, out.size()));
        return out;
    }
    void upload(const std::vector<Particle> &particles, uint64_t first = 0)
    {
        check(psim_upload_aos(h_, particles.data(), first, particles.size()));
    }
    // The same through a page-locked array (no allocation per frame).
    void download(PinnedParticles &out, uint64_t first = 0)
    {
        check(psim_download_aos(h_, out.data(), first, out.size()));
    }
    // Enqueue only; the records are valid after sync().
    void download_async(PinnedParticles &out, uint64_t first = 0)
    {
        check(psim_download_aos_async(h_, out.data(), first, out.size()));
    }
    // The device buffer of 64-byte records a renderer in another API or process maps
    // (psim_shared_aos): its device pointer, a new dma-buf descriptor (the caller closes it)
    // and its size.  export_to(ptr) fills it on the handle's stream.
    struct SharedRecords {
        Particle *device = nullptr;
        int dmabuf_fd = -1;
        uint64_t bytes = 0;
    };
    SharedRecords shared_records()
    {
        SharedRecords r;
        check(psim_shared_aos(h_, &r.device, &r.dmabuf_fd, &r.bytes));
        return r;
    }
    void export_to(Particle *device_buffer)
    {
        check(psim_export_aos_device(h_, device_buffer, 0, psim_particle_count(h_)));
    }
    // Waits for the most recent download_async into `out` and nothing enqueued after it.
    void download_wait(const PinnedParticles &out) { check(psim_download_aos_wait(h_, out.data())); }
    void upload(const PinnedParticles &particles, uint64_t first = 0)
    {
        check(psim_upload_aos(h_, particles.data(), first, particles.size()));
    }

    // Checkpoints (no counterpart in the reference): header + the 64-byte records.
    void save_checkpoint(const std::string &path, uint64_t step = 0)
    {
        check(psim_checkpoint_write(h_, path.c_str(), step));
    }
    uint64_t load_checkpoint(const std::string &path)
    {
        uint64_t step = 0;
        check(psim_checkpoint_read(h_, path.c_str(), &step));
        return step;
    }

    // One file written by several shards: checkpoint_begin(path) once, save_checkpoint on
    // every shard, then commit_checkpoint on any one of them once all have returned.
    static void checkpoint_begin(const std::string &path) { check(psim_checkpoint_begin(path.c_str())); }
    void commit_checkpoint(const std::string &path, uint64_t step = 0)
    {
        check(psim_checkpoint_commit(h_, path.c_str(), step));
    }

    psim_handle *handle() { return h_; }

private:
    static void check(int rc)
    {
        if (rc != PSIM_OK) throw Error(rc, psim_last_error());
    }
    psim_handle *h_ = nullptr;
};

// HIP devices this process sees.
inline int device_count() { return psim_device_count(); }

// The multi-attractor step of a set sharded over one process per GPU (psim_exchange_*): every
// rank owns one attractor; step() all-gathers the ranks' records with RCCL on a stream of its
// own and applies all of them to the particles of `sim`, which must outlive the exchange.
class HipAttractorExchange {
public:
    using Id = std::vector<uint8_t>;                 // PSIM_EXCHANGE_ID_BYTES bytes

    // ONE rank makes the id; the caller hands it to every rank ...
    static Id unique_id()
    {
        Id id(PSIM_EXCHANGE_ID_BYTES);
        check(psim_exchange_unique_id(id.data()));
        return id;
    }
    // ... or all ranks meet at a file that rank 0 publishes (fresh path per run).
    static Id rendezvous_file(const std::string &path, int rank, uint32_t timeout_ms = 60000)
    {
        Id id(PSIM_EXCHANGE_ID_BYTES);
        check(psim_exchange_rendezvous_file(path.c_str(), rank, id.data(), timeout_ms));
        return id;
    }

    HipAttractorExchange(HipParticleSimulation &sim, const Id &id, int rank, int world)
    {
        if (id.size() != PSIM_EXCHANGE_ID_BYTES) throw Error(PSIM_ERR_INVALID, "the id is 128 bytes");
        check(psim_exchange_create(&x_, sim.handle(), id.data(), rank, world));
    }
    ~HipAttractorExchange() { psim_exchange_destroy(x_); }
    HipAttractorExchange(const HipAttractorExchange &) = delete;
    HipAttractorExchange &operator=(const HipAttractorExchange &) = delete;

    // Collective and asynchronous: this rank's attractor joins the others', all are applied.
    void step(const SimParams &params, const Attractor &mine)
    {
        check(psim_exchange_step(x_, &params, &mine));
    }
    // PSIM_EXCHANGE_DEVICE_PACED (default), PSIM_EXCHANGE_HOST_PACED or PSIM_EXCHANGE_PIPELINED
    // (the simulation is then one step behind until sync())
    void set_pacing(int pacing) { check(psim_exchange_set_pacing(x_, pacing)); }
    // 0 (default): wait for ever; otherwise host-side waits give up after that long
    void set_timeout(uint32_t timeout_ms) { check(psim_exchange_set_timeout(x_, timeout_ms)); }
    void sync() { check(psim_exchange_sync(x_)); }
    std::vector<Attractor> last_list()
    {
        std::vector<Attractor> out(static_cast<size_t>(psim_exchange_world(x_)));
        check(psim_exchange_last_list(x_, out.data()));
        return out;
    }
    int rank() const { return psim_exchange_rank(x_); }
    int world() const { return psim_exchange_world(x_); }

private:
    static void check(int rc)
    {
        if (rc != PSIM_OK) throw Error(rc, psim_last_error());
    }
    psim_exchange *x_ = nullptr;
};

// One particle set over several GPUs, driven from one process (psim_group_*): shard k lives
// on device_ids[k].  The trait's methods apply to the whole set.
class HipParticleGroup {
public:
    HipParticleGroup(uint64_t total_count, const std::vector<int> &device_ids,
                     SphereGeneration generation_mode)
    {
        check(psim_group_create(&g_, total_count, device_ids.data(),
                                static_cast<uint32_t>(device_ids.size()),
                                static_cast<int>(generation_mode)));
    }
    ~HipParticleGroup() { psim_group_destroy(g_); }
    HipParticleGroup(const HipParticleGroup &) = delete;
    HipParticleGroup &operator=(const HipParticleGroup &) = delete;

    void update(const SimParams &params) { check(psim_group_update(g_, &params)); }
    void update_batch(const std::vector<SimParams> &params)
    {
        check(psim_group_update_batch(g_, params.data(), static_cast<uint32_t>(params.size())));
    }
    void update_multi(const SimParams &params, const std::vector<Attractor> &attractors)
    {
        check(psim_group_update_multi_host(g_, &params, attractors.data(),
                                           static_cast<uint32_t>(attractors.size())));
    }
    void sync() { check(psim_group_sync(g_)); }
    void reset(SphereGeneration generation_mode)
    {
        check(psim_group_reset(g_, static_cast<int>(generation_mode)));
    }
    uint64_t get_particle_count() const { return psim_group_particle_count(g_); }
    SimulationMethod get_method() const { return SimulationMethod::Hip; }
    uint32_t shard_count() const { return psim_group_shards(g_); }
    psim_handle *shard(uint32_t index) { return psim_group_shard(g_, index); }

    std::vector<Particle> download()
    {
        std::vector<Particle> out(get_particle_count());
        check(psim_group_download_aos(g_, out.data(), 0, out.size()));
        return out;
    }
    void upload(const std::vector<Particle> &particles, uint64_t first = 0)
    {
        check(psim_group_upload_aos(g_, particles.data(), first, particles.size()));
    }
    void save_checkpoint(const std::string &path, uint64_t step = 0)
    {
        check(psim_group_checkpoint_write(g_, path.c_str(), step));
    }
    uint64_t load_checkpoint(const std::string &path)
    {
        uint64_t step = 0;
        check(psim_group_checkpoint_read(g_, path.c_str(), &step));
        return step;
    }

private:
    static void check(int rc)
    {
        if (rc != PSIM_OK) throw Error(rc, psim_last_error());
    }
    psim_group *g_ = nullptr;
};

}  // namespace psim

#endif  // PSIM_HPP
