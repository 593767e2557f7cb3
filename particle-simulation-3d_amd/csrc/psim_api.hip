// psim_api.hip -- the extern "C" boundary declared in include/psim.h.
//
// A handle owns one device allocation carved into the SoA arrays, one stream and
// a small staging buffer.  All entry points return status codes; nothing throws
// across the boundary.  There is no CPU fallback: without a gfx950 device
// psim_create fails with PSIM_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <new>
#include <thread>
#include <vector>

#include "psim_checkpoint_file.h"
#include "psim_hostgen.h"
#include "psim_kernels.h"
#include "psim_tuning.h"

static_assert(sizeof(psim_sim_params) == 48, "SimParams is 48 bytes (mod.rs:52-67)");
static_assert(offsetof(psim_sim_params, color_mode) == 8, "layout");
static_assert(offsetof(psim_sim_params, is_mouse_dragging) == 20, "layout");
static_assert(offsetof(psim_sim_params, mouse_position) == 32, "layout");
static_assert(sizeof(psim_particle) == 64, "Particle is 64 bytes (mod.rs:86-98)");
static_assert(offsetof(psim_particle, velocity) == 16, "layout");
static_assert(offsetof(psim_particle, color) == 32, "layout");
static_assert(offsetof(psim_particle, initial_color) == 48, "layout");
static_assert(sizeof(psim_attractor) == 32, "attractor record is 32 bytes");
static_assert(sizeof(psim_checkpoint_header) == 64, "checkpoint header is 64 bytes");

#define PSIM_API extern "C" __attribute__((visibility("default")))

namespace {

thread_local char g_error[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace

// The error sink of psim_checkpoint_file.h.
int psim_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return code;
}

namespace {

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return fail(e_ == hipErrorOutOfMemory ? PSIM_ERR_NOMEM : PSIM_ERR_HIP,      \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                      \
    } while (0)

constexpr uint64_t kUniformSeed = 69;          // the reference's own seed, mod.rs:170
constexpr uint32_t kStageParticles = 1u << 22; // AoS staging chunk: 4 Mi particles = 256 MiB
constexpr uint32_t kBounceParticles = 1u << 19; // page-locked bounce slot: 512 Ki particles = 32 MiB
constexpr uint32_t kCapacityQuantum = 1024;    // particles; keeps every array 4 KiB-granular
constexpr uint32_t kListSlots = 4;             // host attractor lists in flight per handle
constexpr uint64_t kScratchBytes = 256 + kListSlots * PSIM_MAX_ATTRACTORS * sizeof(psim_attractor) + 3840;

// Overrides of the measured launch shape (psim_tuning.h: experiments and shape-forcing
// tests only), copied when a handle is created; -1 = automatic.
struct TuningOverrides {
    int nontemporal = -1;   // 0 plain | 1 streaming | 2 cache-resident accesses
    int reserve_lds = -1;   // bytes of LDS reserved per workgroup
    int max_blocks = -1;    // grid cap (0 = one pass)
};

TuningOverrides read_overrides()
{
    TuningOverrides o;
    o.nontemporal = psim::knob(psim::KNOB_NT);
    o.reserve_lds = psim::knob(psim::KNOB_RESERVE_LDS);
    o.max_blocks = psim::knob(psim::KNOB_MAX_BLOCKS);
    return o;
}

psim::LaunchTuning tuning_for(uint32_t n, uint32_t n_attractors, const TuningOverrides &o)
{
    psim::LaunchTuning t = psim::choose_tuning(n, n_attractors);
    if (o.nontemporal >= 0)
        t.access = o.nontemporal == 1 ? psim::PSIM_ACCESS_STREAM
                 : o.nontemporal == 2 ? psim::PSIM_ACCESS_RESIDENT : psim::PSIM_ACCESS_PLAIN;
    if (o.reserve_lds >= 0) t.reserve_lds = t.reserve_lds_colour = (uint32_t)o.reserve_lds;
    if (o.max_blocks >= 0) t.max_blocks = (uint32_t)o.max_blocks;
    return t;
}

// Fused passes: 2-4 steps are still HBM-bound and run 13 workgroups per CU on states of
// 96 MB and more; from 5 steps on the pass is VALU-bound and runs uncapped (100 M particles,
// ms per pass at 13 resident / uncapped: 2 steps 0.778 / 0.841, 4: 0.766 / 0.842, 8: 1.196 /
// 0.962, 16: 2.26 / 1.88).
psim::LaunchTuning batch_tuning_for(uint32_t n, uint32_t steps, const TuningOverrides &o)
{
    if (steps <= 1) return tuning_for(n, 1, o);
    TuningOverrides access_only = o;
    access_only.reserve_lds = -1;
    psim::LaunchTuning t = tuning_for(n, 1, access_only);
    t.reserve_lds = t.reserve_lds_colour = (24ull * n >= 96000000ull && steps <= 4) ? 11776u : 0u;
    if (o.reserve_lds >= 0) t.reserve_lds = t.reserve_lds_colour = (uint32_t)o.reserve_lds;
    return t;
}

}  // namespace

struct psim_handle;
namespace { void release_shared_aos(psim_handle *h); }

struct psim_handle {
    int device = 0;
    int num_cus = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    uint32_t count = 0;
    uint32_t capacity = 0;
    uint64_t global_first = 0;
    uint64_t global_count = 0;   // 0 = not a shard: the set is [0, count)
    int generation_mode = PSIM_GEN_HOLLOW;
    bool paused = false;
    int colour_semantics = PSIM_COLOR_KEEP;
    int resize_semantics = PSIM_RESIZE_REGENERATE;
    uint32_t watermark = 0;      // particles generated so far: the CPU backend's Vec length
    bool colours_changed = false;   // colour may differ from initial_colour
    void *slab = nullptr;       // one allocation, carved into `soa`
    psim::SoaPtrs soa = {};
    uint64_t *checksum_dev = nullptr;          // 8 x u64 inside the slab
    psim_attractor *list_dev = nullptr;        // kListSlots attractor lists inside the slab
    psim_attractor *list_host = nullptr;       // their page-locked staging
    psim_attractor *static_list = nullptr;     // this device's copy of the library's static list
    hipEvent_t list_copied[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t list_next = 0;
    psim_particle *aos_stage = nullptr;        // lazily allocated AoS staging chunk
    uint32_t stage_particles = 0;              // its size
    psim_particle *bounce = nullptr;           // two page-locked slots for pageable downloads
    hipEvent_t bounce_done[2] = {nullptr, nullptr};
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    // the most recent asynchronous downloads, by destination: what psim_download_aos_wait waits for
    struct PendingDownload {
        const void *dst = nullptr;
        hipEvent_t done = nullptr;
        uint64_t age = 0;
    } downloads[4];
    uint64_t download_clock = 0;
    // The records as a device buffer other APIs can map: a physical allocation of its own
    // (hipMemCreate), so that the exported descriptor refers to exactly this buffer.
    psim_particle *shared_aos = nullptr;
    uint64_t shared_bytes = 0;                 // mapped size (a multiple of the granularity)
    hipMemGenericAllocationHandle_t shared_memory = nullptr;
    TuningOverrides overrides;
};

namespace {

constexpr uint32_t kMaxParticles = PSIM_MAX_PARTICLES;   // largest multiple of the quantum in 32 bits

// Callers have checked n <= kMaxParticles.
uint32_t round_capacity(uint32_t n)
{
    uint64_t c = ((uint64_t)n + kCapacityQuantum - 1) / kCapacityQuantum * kCapacityQuantum;
    if (c == 0) c = kCapacityQuantum;
    return (uint32_t)c;
}

// Carves the slab: six fp32 arrays, colour, initial colour, padding, scratch.
// `skew` extra bytes separate consecutive fp32 arrays so that the six streams do
// not sit at the same offset modulo the memory-channel interleave.
int allocate(psim_handle *h, uint32_t capacity)
{
    const uint64_t cap = capacity;
    const int skew_knob = psim::knob(psim::KNOB_SKEW_BYTES);
    const uint64_t skew = (uint64_t)(skew_knob < 0 ? 0 : skew_knob) & ~255ull;
    const uint64_t f32_bytes = cap * 4 + skew;
    const uint64_t total = 6 * f32_bytes + cap * 16 * 2 + cap * 8 + kScratchBytes;
    void *slab = nullptr;
    HIP_TRY(hipMalloc(&slab, total));
    char *b = static_cast<char *>(slab);
    h->slab = slab;
    h->soa.px = reinterpret_cast<float *>(b + 0 * f32_bytes);
    h->soa.py = reinterpret_cast<float *>(b + 1 * f32_bytes);
    h->soa.pz = reinterpret_cast<float *>(b + 2 * f32_bytes);
    h->soa.vx = reinterpret_cast<float *>(b + 3 * f32_bytes);
    h->soa.vy = reinterpret_cast<float *>(b + 4 * f32_bytes);
    h->soa.vz = reinterpret_cast<float *>(b + 5 * f32_bytes);
    b += 6 * f32_bytes;
    h->soa.color = reinterpret_cast<float *>(b);
    b += cap * 16;
    h->soa.initial_color = reinterpret_cast<float *>(b);
    b += cap * 16;
    h->soa.padding = reinterpret_cast<float *>(b);
    b += cap * 8;
    h->checksum_dev = reinterpret_cast<uint64_t *>(b);
    h->list_dev = reinterpret_cast<psim_attractor *>(b + 256);
    h->capacity = capacity;
    return PSIM_OK;
}

// What allocate() sets, so that a resize that fails later can put the old storage back.
struct Storage {
    void *slab;
    psim::SoaPtrs soa;
    uint64_t *checksum_dev;
    psim_attractor *list_dev;
    uint32_t capacity;
};

Storage storage_of(const psim_handle *h)
{
    return Storage{h->slab, h->soa, h->checksum_dev, h->list_dev, h->capacity};
}

void restore_storage(psim_handle *h, const Storage &st)
{
    h->slab = st.slab;
    h->soa = st.soa;
    h->checksum_dev = st.checksum_dev;
    h->list_dev = st.list_dev;
    h->capacity = st.capacity;
}

// The device-side AoS staging buffer: the whole set when it is small (<= 1 Mi particles),
// otherwise the fixed 4 Mi-particle chunk.  Grows when a resize made the set larger.
int ensure_stage(psim_handle *h)
{
    const uint32_t want = h->capacity > (1u << 20) ? kStageParticles : h->capacity;
    if (h->aos_stage && h->stage_particles >= want) return PSIM_OK;
    if (h->aos_stage) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        (void)hipFree(h->aos_stage);
        h->aos_stage = nullptr;
        h->stage_particles = 0;
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->aos_stage),
                      (size_t)want * sizeof(psim_particle)));
    h->stage_particles = want;
    return PSIM_OK;
}

using psim_hostgen::gen_hollow;
using psim_hostgen::gen_filled;

// What to generate and where: particles [gen_first, gen_first + n) of a set of gen_total
// particles, into the handle's particles [dst_first, dst_first + n).
struct GenRange {
    uint32_t dst_first;
    uint32_t n;
    uint64_t gen_first;
    uint64_t gen_total;
};

int generate_impl(psim_handle *h, int mode, const GenRange &r);

// The host generators use standard containers and threads: nothing they throw may cross
// the C boundary.
int generate_range(psim_handle *h, int mode, const GenRange &r)
{
    try {
        return generate_impl(h, mode, r);
    } catch (const std::bad_alloc &) {
        return fail(PSIM_ERR_NOMEM, "host allocation failed while generating particles");
    } catch (...) {
        return fail(PSIM_ERR_HIP, "unexpected failure while generating particles");
    }
}

// Fills particles [0, count) of the handle with `mode`: the handle's own index range of its
// (possibly larger, sharded) set.
int generate(psim_handle *h, int mode)
{
    // The reference's two generators index particles with u32 (mod.rs:150-188): a set of more
    // than 2^32 particles (shards of a larger array) only has the synthetic uniform state.
    if (h->global_count > 0xFFFFFFFFull && mode != PSIM_GEN_UNIFORM)
        return fail(PSIM_ERR_INVALID, "the reference generators index particles with u32");
    const int rc = generate_range(h, mode, GenRange{0u, h->count, h->global_first,
                                                    h->global_count ? h->global_count
                                                                    : (uint64_t)h->count});
    if (rc != PSIM_OK) return rc;
    h->colours_changed = false;                  // every generator writes colour = initial colour
    h->watermark = h->count;
    return PSIM_OK;
}

int generate_impl(psim_handle *h, int mode, const GenRange &r)
{
    const uint64_t gfirst = r.gen_first;
    const uint32_t gcount32 = (uint32_t)r.gen_total;
    const uint64_t count = r.n;
    if (mode == PSIM_GEN_UNIFORM) {
        psim::SoaPtrs at = h->soa;               // the arrays from particle dst_first on
        at.px += r.dst_first; at.py += r.dst_first; at.pz += r.dst_first;
        at.vx += r.dst_first; at.vy += r.dst_first; at.vz += r.dst_first;
        at.color += 4ull * r.dst_first; at.initial_color += 4ull * r.dst_first;
        at.padding += 2ull * r.dst_first;
        HIP_TRY(psim::launch_gen_uniform(at, r.n, gfirst, kUniformSeed, h->stream));
        return PSIM_OK;
    }
    if (mode != PSIM_GEN_HOLLOW && mode != PSIM_GEN_FILLED)
        return fail(PSIM_ERR_INVALID, "unknown generation mode %d", mode);
    if (count == 0) return PSIM_OK;
    // Two page-locked chunks: while chunk k travels to the device and is transposed
    // there, the host threads already fill chunk k + 1.
    const uint64_t chunk = 1u << 20;
    static_assert(2 * (1u << 20) <= kStageParticles, "two chunks fit the device staging buffer");
    const uint64_t slot_particles = count < chunk ? count : chunk;
    const int slots = count > chunk ? 2 : 1;
    int rc = ensure_stage(h);
    if (rc != PSIM_OK) return rc;
    struct Staging {
        hipStream_t stream = nullptr;
        psim_particle *host = nullptr;
        hipEvent_t done[2] = {nullptr, nullptr};
        ~Staging()
        {
            (void)hipStreamSynchronize(stream);      // an error path may leave copies in flight
            if (host) (void)hipHostFree(host);
            for (hipEvent_t e : done)
                if (e) (void)hipEventDestroy(e);
        }
    } st;
    st.stream = h->stream;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&st.host),
                          (size_t)(slot_particles * slots) * sizeof(psim_particle),
                          hipHostMallocDefault));
    for (int b = 0; b < slots; ++b)
        HIP_TRY(hipEventCreateWithFlags(&st.done[b], hipEventDisableTiming));
    const auto fill = [=](psim_particle *dst, uint64_t first, uint64_t len) {
        if (mode == PSIM_GEN_HOLLOW) gen_hollow(dst, gcount32, first, len);
        else gen_filled(dst, kUniformSeed, first, len);
    };
    // built here, on this thread, where a failed allocation can still be reported
    if (mode == PSIM_GEN_FILLED) (void)psim_hostgen::XoshiroPowers::get();
    const unsigned hw = std::thread::hardware_concurrency();
    uint64_t nchunk = 0;
    for (uint64_t done = 0; done < count; done += chunk, ++nchunk) {
        const uint64_t m = (count - done) < chunk ? (count - done) : chunk;
        const int b = (int)(nchunk % (uint64_t)slots);
        psim_particle *host = st.host + (uint64_t)b * slot_particles;
        psim_particle *stage = h->aos_stage + (uint64_t)b * slot_particles;
        if (nchunk >= (uint64_t)slots) HIP_TRY(hipEventSynchronize(st.done[b]));   // slot free again
        // both states are index-addressed: split the chunk over a few host threads
        const uint64_t workers = m < 65536 ? 1 : (hw == 0 ? 4 : (hw > 16 ? 16 : hw));
        const uint64_t per = (m + workers - 1) / workers;
        std::vector<std::thread> pool;
        for (uint64_t w = 1; w < workers && w * per < m; ++w) {
            const uint64_t lo = w * per, len = (m - lo) < per ? (m - lo) : per;
            try {
                pool.emplace_back(fill, host + lo, gfirst + done + lo, len);
            } catch (...) {                                      // no thread: do it here
                fill(host + lo, gfirst + done + lo, len);
            }
        }
        fill(host, gfirst + done, per < m ? per : m);
        for (auto &t : pool) t.join();
        HIP_TRY(hipMemcpyAsync(stage, host, (size_t)m * sizeof(psim_particle),
                               hipMemcpyHostToDevice, h->stream));
        HIP_TRY(psim::launch_import_aos(h->soa, r.dst_first + (uint32_t)done, (uint32_t)m, stage, h->stream));
        HIP_TRY(hipEventRecord(st.done[b], h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));                    // before the staging goes away
    return PSIM_OK;
}

int upload_aos_chunked(psim_handle *h, const psim_particle *src, uint64_t first, uint64_t n)
{
    int rc = ensure_stage(h);
    if (rc != PSIM_OK) return rc;
    for (uint64_t done = 0; done < n; done += kStageParticles) {
        const uint32_t m = (uint32_t)((n - done) < kStageParticles ? (n - done) : kStageParticles);
        HIP_TRY(hipMemcpyAsync(h->aos_stage, src + done, (size_t)m * sizeof(psim_particle),
                               hipMemcpyHostToDevice, h->stream));
        HIP_TRY(psim::launch_import_aos(h->soa, (uint32_t)(first + done), m, h->aos_stage,
                                        h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return PSIM_OK;
}

int check_range(const psim_handle *h, uint64_t first, uint64_t n)
{
    if (first > h->count || n > h->count - first)
        return fail(PSIM_ERR_INVALID, "range [%llu, +%llu) exceeds particle count %u",
                    (unsigned long long)first, (unsigned long long)n, h->count);
    return PSIM_OK;
}

int use_device(const psim_handle *h)
{
    HIP_TRY(hipSetDevice(h->device));
    return PSIM_OK;
}

#define CHECK_HANDLE(h)                                              \
    do {                                                             \
        if (!(h)) return fail(PSIM_ERR_INVALID, "null handle");      \
        int rc_ = use_device(h);                                     \
        if (rc_ != PSIM_OK) return rc_;                              \
    } while (0)

int create_impl(psim_handle **out, uint32_t count, uint64_t global_first, uint64_t global_count,
                int generation_mode, int device_id)
{
    if (!out) return fail(PSIM_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (generation_mode < PSIM_GEN_HOLLOW || generation_mode > PSIM_GEN_UNIFORM)
        return fail(PSIM_ERR_INVALID, "unknown generation mode %d", generation_mode);
    if (global_count && (global_first > global_count || count > global_count - global_first))
        return fail(PSIM_ERR_INVALID, "shard [%llu, +%u) exceeds global count %llu",
                    (unsigned long long)global_first, count, (unsigned long long)global_count);
    if (count > kMaxParticles)
        return fail(PSIM_ERR_INVALID, "%u particles exceed the maximum of %u per handle", count,
                    kMaxParticles);
    if (global_count > 0xFFFFFFFFull && generation_mode != PSIM_GEN_UNIFORM)
        return fail(PSIM_ERR_INVALID, "the reference generators index particles with u32");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(PSIM_ERR_NO_DEVICE, "no HIP device visible (this backend has no CPU path)");
    if (device_id < 0 || device_id >= ndev)
        return fail(PSIM_ERR_INVALID, "device %d out of range (%d visible)", device_id, ndev);
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PSIM_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only",
                    device_id, prop.gcnArchName);

    psim_handle *h = new (std::nothrow) psim_handle();
    if (!h) return fail(PSIM_ERR_NOMEM, "host allocation failed");
    h->device = device_id;
    h->num_cus = prop.multiProcessorCount;
    h->count = count;
    h->global_first = global_first;
    h->global_count = global_count;
    h->generation_mode = generation_mode;
    h->overrides = read_overrides();
    int rc = PSIM_OK;
    hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        rc = fail(PSIM_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    } else {
        h->stream = h->own_stream;
        // set-up for psim_update_multi_host_noalloc, which may not do it itself
        e = psim::static_list_address(device_id, &h->static_list);
        if (e != hipSuccess)
            rc = fail(PSIM_ERR_HIP, "address of the static attractor list: %s", hipGetErrorString(e));
        if (rc == PSIM_OK) rc = allocate(h, round_capacity(count));
        if (rc == PSIM_OK) rc = generate(h, generation_mode);
        if (rc == PSIM_OK) {
            e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess)
                rc = fail(PSIM_ERR_HIP, "initial generation failed: %s", hipGetErrorString(e));
        }
    }
    if (rc != PSIM_OK) {
        psim_destroy(h);
        return rc;
    }
    *out = h;
    return PSIM_OK;
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

bool is_colour_mode(uint32_t m) { return m == 1u || m == 2u; }

// Colour bookkeeping after an update whose LAST step ran with `last_mode`, `wrote` telling
// whether any step of it computed colours.  Under PSIM_COLOR_RESET the compute shader's
// rule applies (compute.wgsl:75-77, 90-92): a step in any other mode leaves colour ==
// initial_colour.  The shader rewrites the colour every step; the visible state is the
// same when the copy is made only after colours have actually been changed.
int settle_colours(psim_handle *h, uint32_t last_mode, bool wrote)
{
    if (wrote) h->colours_changed = true;
    if (h->colour_semantics == PSIM_COLOR_RESET && !is_colour_mode(last_mode) &&
        h->colours_changed && h->count) {
        HIP_TRY(hipMemcpyAsync(h->soa.color, h->soa.initial_color, (size_t)h->count * 16u,
                               hipMemcpyDeviceToDevice, h->stream));
        h->colours_changed = false;
    }
    return PSIM_OK;
}

}  // namespace

// ---- lifecycle --------------------------------------------------------------

PSIM_API int psim_create(psim_handle **out, uint32_t count, int generation_mode, int device_id)
{
    return create_impl(out, count, 0, 0, generation_mode, device_id);
}

PSIM_API int psim_create_shard(psim_handle **out, uint32_t count, uint64_t global_first,
                               uint64_t global_count, int generation_mode, int device_id)
{
    if (global_count == 0) return fail(PSIM_ERR_INVALID, "global_count must be > 0 for a shard");
    return create_impl(out, count, global_first, global_count, generation_mode, device_id);
}

PSIM_API void psim_destroy(psim_handle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->ev_start) (void)hipEventDestroy(h->ev_start);
    if (h->ev_stop) (void)hipEventDestroy(h->ev_stop);
    if (h->aos_stage) (void)hipFree(h->aos_stage);
    release_shared_aos(h);                      // importers keep their own reference
    if (h->bounce) (void)hipHostFree(h->bounce);
    if (h->list_host) (void)hipHostFree(h->list_host);
    for (hipEvent_t e : h->list_copied)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->bounce_done)
        if (e) (void)hipEventDestroy(e);
    for (auto &d : h->downloads)
        if (d.done) (void)hipEventDestroy(d.done);
    if (h->slab) (void)hipFree(h->slab);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

// ---- the hot path -----------------------------------------------------------

PSIM_API int psim_update(psim_handle *h, const psim_sim_params *params)
{
    CHECK_HANDLE(h);
    if (!params) return fail(PSIM_ERR_INVALID, "null params");
    const int force = params->is_mouse_dragging > 0 ? 1 : 0;
    HIP_TRY(psim::launch_update(h->soa, h->count, *params, force, nullptr, nullptr, 0,
                                tuning_for(h->count, 1, h->overrides),
                                h->stream));
    return settle_colours(h, params->color_mode, is_colour_mode(params->color_mode));
}

PSIM_API int psim_update_batch(psim_handle *h, const psim_sim_params *params_host,
                               uint32_t n_steps)
{
    CHECK_HANDLE(h);
    if (n_steps == 0) return PSIM_OK;
    if (!params_host) return fail(PSIM_ERR_INVALID, "null params");
    for (uint32_t done = 0; done < n_steps; done += PSIM_BATCH_STEPS) {
        const uint32_t m = (n_steps - done) < PSIM_BATCH_STEPS ? (n_steps - done)
                                                                : PSIM_BATCH_STEPS;
        HIP_TRY(psim::launch_update_batch(h->soa, h->count, params_host + done, m,
                                          batch_tuning_for(h->count, m, h->overrides), h->stream));
    }
    bool wrote = false;
    for (uint32_t k = 0; k < n_steps; ++k) wrote = wrote || is_colour_mode(params_host[k].color_mode);
    return settle_colours(h, params_host[n_steps - 1].color_mode, wrote);
}

PSIM_API int psim_update_multi(psim_handle *h, const psim_sim_params *params,
                               const psim_attractor *attractors_device, uint32_t n_attractors)
{
    CHECK_HANDLE(h);
    if (!params) return fail(PSIM_ERR_INVALID, "null params");
    if (n_attractors > PSIM_MAX_ATTRACTORS)
        return fail(PSIM_ERR_INVALID, "%u attractors exceed the maximum of %u", n_attractors,
                    PSIM_MAX_ATTRACTORS);
    if (n_attractors && !attractors_device) return fail(PSIM_ERR_INVALID, "null attractor list");
    if (n_attractors && (reinterpret_cast<uintptr_t>(attractors_device) & 3u))
        return fail(PSIM_ERR_INVALID, "attractor list must be 4-byte aligned");
    const int force = n_attractors ? 2 : 0;
    HIP_TRY(psim::launch_update(h->soa, h->count, *params, force, attractors_device, nullptr,
                                n_attractors, tuning_for(h->count, n_attractors, h->overrides),
                                h->stream));
    return settle_colours(h, params->color_mode, is_colour_mode(params->color_mode));
}

PSIM_API int psim_update_multi_host(psim_handle *h, const psim_sim_params *params,
                                    const psim_attractor *attractors_host, uint32_t n_attractors)
{
    CHECK_HANDLE(h);
    if (!params) return fail(PSIM_ERR_INVALID, "null params");
    if (n_attractors > PSIM_MAX_ATTRACTORS)
        return fail(PSIM_ERR_INVALID, "%u attractors exceed the maximum of %u", n_attractors,
                    PSIM_MAX_ATTRACTORS);
    if (n_attractors && !attractors_host) return fail(PSIM_ERR_INVALID, "null attractor list");
    if (n_attractors > PSIM_KERNARG_ATTRACTORS) {
        // Too long for the kernel arguments: through one of a few page-locked slots into the
        // handle's own device scratch, in stream order.  A slot is reused only after its
        // earlier copy has left the host.
        static_assert(sizeof(h->list_copied) / sizeof(h->list_copied[0]) == kListSlots, "slots");
        if (!h->list_host)
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h->list_host),
                                  kListSlots * PSIM_MAX_ATTRACTORS * sizeof(psim_attractor),
                                  hipHostMallocDefault));
        const uint32_t slot = h->list_next++ % kListSlots;
        if (!h->list_copied[slot])
            HIP_TRY(hipEventCreateWithFlags(&h->list_copied[slot], hipEventDisableTiming));
        else
            HIP_TRY(hipEventSynchronize(h->list_copied[slot]));
        psim_attractor *host = h->list_host + slot * PSIM_MAX_ATTRACTORS;
        psim_attractor *dev = h->list_dev + slot * PSIM_MAX_ATTRACTORS;
        memcpy(host, attractors_host, n_attractors * sizeof(psim_attractor));
        HIP_TRY(hipMemcpyAsync(dev, host, n_attractors * sizeof(psim_attractor),
                               hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipEventRecord(h->list_copied[slot], h->stream));
        HIP_TRY(psim::launch_update(h->soa, h->count, *params, 2, dev, nullptr, n_attractors,
                                    tuning_for(h->count, n_attractors, h->overrides), h->stream));
        return settle_colours(h, params->color_mode, is_colour_mode(params->color_mode));
    }
    if (n_attractors == 1) {
        // One attractor is the reference's own case: its record becomes the mouse_* fields
        // and the single-attractor kernel runs (same arithmetic, 2 % faster than the list
        // kernel with a list of one).
        psim_sim_params p = *params;
        const psim_attractor &a = attractors_host[0];
        p.is_mouse_dragging = a.active > 0 ? 1u : 0u;
        p.mouse_position[0] = a.position[0];
        p.mouse_position[1] = a.position[1];
        p.mouse_position[2] = a.position[2];
        p.mouse_force = a.force;
        p.mouse_radius = a.radius;
        HIP_TRY(psim::launch_update(h->soa, h->count, p, (int)p.is_mouse_dragging, nullptr, nullptr,
                                    0, tuning_for(h->count, 1, h->overrides), h->stream));
        return settle_colours(h, params->color_mode, is_colour_mode(params->color_mode));
    }
    const int force = n_attractors ? 3 : 0;
    HIP_TRY(psim::launch_update(h->soa, h->count, *params, force, nullptr, attractors_host,
                                n_attractors, tuning_for(h->count, n_attractors, h->overrides),
                                h->stream));
    return settle_colours(h, params->color_mode, is_colour_mode(params->color_mode));
}

PSIM_API int psim_update_multi_host_noalloc(psim_handle *h, const psim_sim_params *params,
                                            const psim_attractor *attractors_host,
                                            uint32_t n_attractors)
{
    CHECK_HANDLE(h);
    if (!params) return fail(PSIM_ERR_INVALID, "null params");
    if (n_attractors > PSIM_MAX_ATTRACTORS)
        return fail(PSIM_ERR_INVALID,
                    "%u attractors exceed the %u records of the static device list", n_attractors,
                    PSIM_MAX_ATTRACTORS);
    if (n_attractors && !attractors_host) return fail(PSIM_ERR_INVALID, "null attractor list");
    if (!h->static_list) return fail(PSIM_ERR_HIP, "the static attractor list has no address");
    // Every record the kernel reads is written first, in stream order, so nothing of an
    // earlier, longer list is seen; nothing has to start at zero.
    HIP_TRY(psim::launch_store_attractors(h->static_list, attractors_host, n_attractors, h->stream));
    HIP_TRY(psim::launch_update(h->soa, h->count, *params, n_attractors ? 2 : 0, h->static_list,
                                nullptr, n_attractors,
                                tuning_for(h->count, n_attractors, h->overrides), h->stream));
    return settle_colours(h, params->color_mode, is_colour_mode(params->color_mode));
}

PSIM_API int psim_update_soa(float *px, float *py, float *pz, float *vx, float *vy, float *vz,
                             float *color, uint32_t count, const psim_sim_params *params,
                             void *hip_stream)
{
    if (!params) return fail(PSIM_ERR_INVALID, "null params");
    if (count == 0) return PSIM_OK;
    if (!px || !py || !pz || !vx || !vy || !vz) return fail(PSIM_ERR_INVALID, "null state array");
    const bool colours = params->color_mode == 1 || params->color_mode == 2;
    if (colours && !color)
        return fail(PSIM_ERR_INVALID, "color_mode %u needs a colour array", params->color_mode);
    psim::SoaPtrs s = {px, py, pz, vx, vy, vz, color, nullptr, nullptr};
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    const bool vec = aligned16(px) && aligned16(py) && aligned16(pz) && aligned16(vx) &&
                     aligned16(vy) && aligned16(vz) && (!colours || aligned16(color));
    if (vec) {
        const TuningOverrides overrides = read_overrides();
        psim::LaunchTuning t = tuning_for(count, 1, overrides);
        // The sc1 / nt sc1 stores write partial 128-byte lines through to memory; arrays
        // that are 16-byte but not line-aligned are faster with plain accesses (100 M
        // particles, arrays 16 bytes off a line: 5.99 k GB/s plain against 5.14 k).
        const uintptr_t any = reinterpret_cast<uintptr_t>(px) | reinterpret_cast<uintptr_t>(py) |
                              reinterpret_cast<uintptr_t>(pz) | reinterpret_cast<uintptr_t>(vx) |
                              reinterpret_cast<uintptr_t>(vy) | reinterpret_cast<uintptr_t>(vz) |
                              (colours ? reinterpret_cast<uintptr_t>(color) : 0);
        if ((any & 127u) != 0 && overrides.nontemporal < 0) t.access = psim::PSIM_ACCESS_PLAIN;
        HIP_TRY(psim::launch_update(s, count, *params, params->is_mouse_dragging > 0 ? 1 : 0,
                                    nullptr, nullptr, 0, t, stream));
    } else {
        const uintptr_t all = reinterpret_cast<uintptr_t>(px) | reinterpret_cast<uintptr_t>(py) |
                              reinterpret_cast<uintptr_t>(pz) | reinterpret_cast<uintptr_t>(vx) |
                              reinterpret_cast<uintptr_t>(vy) | reinterpret_cast<uintptr_t>(vz) |
                              (colours ? reinterpret_cast<uintptr_t>(color) : 0);
        if (all & 3u)
            return fail(PSIM_ERR_INVALID, "state and colour arrays must be 4-byte aligned");
        HIP_TRY(psim::launch_update_scalar(s, count, *params, 0, stream));
    }
    return PSIM_OK;
}

PSIM_API int psim_store_attractor(psim_attractor *dst_device, const psim_attractor *value_host,
                                  void *hip_stream)
{
    if (!dst_device || !value_host) return fail(PSIM_ERR_INVALID, "null attractor pointer");
    if (reinterpret_cast<uintptr_t>(dst_device) & 3u)
        return fail(PSIM_ERR_INVALID, "attractor destination must be 4-byte aligned");
    HIP_TRY(psim::launch_store_attractor(dst_device, *value_host,
                                         static_cast<hipStream_t>(hip_stream)));
    return PSIM_OK;
}

PSIM_API int psim_sync(psim_handle *h)
{
    CHECK_HANDLE(h);
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PSIM_OK;
}

// ---- state management -------------------------------------------------------

namespace {

// `resize_buffer` of the reference's CPU backend (src/simulation/cpu.rs:123-158): its
// particle vector only ever grows.  Growing past everything generated so far appends a
// FRESH set of (new_count - generated) particles -- a whole new sphere of that size, not a
// continuation of the old one (cpu.rs:138-140) -- and keeps every existing particle;
// shrinking only lowers the active count, and growing again within what was generated brings
// the same particles back as they were left.
int resize_append(psim_handle *h, uint32_t new_count, int generation_mode)
{
    if (new_count > h->watermark) {
        const uint32_t additional = new_count - h->watermark;
        if (new_count > h->capacity) {               // cpu.rs:142-147: a larger buffer, state kept
            const Storage before = storage_of(h);
            const psim::SoaPtrs old = h->soa;
            int rc = allocate(h, round_capacity(new_count));
            if (rc != PSIM_OK) return rc;            // old storage stays valid
            const size_t w = h->watermark;
            float *const from[6] = {old.px, old.py, old.pz, old.vx, old.vy, old.vz};
            float *const to[6] = {h->soa.px, h->soa.py, h->soa.pz, h->soa.vx, h->soa.vy, h->soa.vz};
            hipError_t e = hipSuccess;
            for (int k = 0; k < 6 && e == hipSuccess; ++k)
                e = hipMemcpyAsync(to[k], from[k], w * 4, hipMemcpyDeviceToDevice, h->stream);
            if (e == hipSuccess)
                e = hipMemcpyAsync(h->soa.color, old.color, w * 16, hipMemcpyDeviceToDevice, h->stream);
            if (e == hipSuccess)
                e = hipMemcpyAsync(h->soa.initial_color, old.initial_color, w * 16,
                                   hipMemcpyDeviceToDevice, h->stream);
            if (e == hipSuccess)
                e = hipMemcpyAsync(h->soa.padding, old.padding, w * 8, hipMemcpyDeviceToDevice,
                                   h->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) {                   // back to the old buffer, which is intact
                (void)hipStreamSynchronize(h->stream);
                (void)hipFree(h->slab);
                restore_storage(h, before);
                return fail(PSIM_ERR_HIP, "copying the particles into the larger buffer failed: %s",
                            hipGetErrorString(e));
            }
            (void)hipFree(before.slab);
        }
        const int rc = generate_range(h, generation_mode,
                                      GenRange{h->watermark, additional, 0, additional});
        if (rc != PSIM_OK) return rc;
        h->watermark = new_count;
    }
    h->count = new_count;                            // cpu.rs:150
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PSIM_OK;
}

}  // namespace

PSIM_API int psim_resize(psim_handle *h, uint32_t new_count, int generation_mode)
{
    CHECK_HANDLE(h);
    if (generation_mode < PSIM_GEN_HOLLOW || generation_mode > PSIM_GEN_UNIFORM)
        return fail(PSIM_ERR_INVALID, "unknown generation mode %d", generation_mode);
    if (new_count == h->count) {                     // compute.rs:153-157: mode stored, no-op
        h->generation_mode = generation_mode;
        return PSIM_OK;
    }
    if (new_count > kMaxParticles)
        return fail(PSIM_ERR_INVALID, "%u particles exceed the maximum of %u per handle",
                    new_count, kMaxParticles);
    if (h->global_count)
        return fail(PSIM_ERR_INVALID, "a shard cannot be resized; recreate the shards");
    HIP_TRY(hipStreamSynchronize(h->stream));
    // Count, watermark and generation mode change only once the new particles exist: a call
    // that fails leaves the handle describing what its storage really holds.
    if (h->resize_semantics == PSIM_RESIZE_APPEND) {
        const int rc = resize_append(h, new_count, generation_mode);
        if (rc == PSIM_OK) h->generation_mode = generation_mode;
        return rc;
    }
    const Storage before = storage_of(h);
    const uint32_t old_count = h->count;
    const bool grown = new_count > h->capacity;
    if (grown) {                                     // compute.rs:162-186: new buffer
        int rc = allocate(h, round_capacity(new_count));
        if (rc != PSIM_OK) return rc;                // old storage stays valid
    }
    h->count = new_count;
    int rc = generate(h, generation_mode);           // compute.rs:160: all particles
    if (rc == PSIM_OK && hipStreamSynchronize(h->stream) != hipSuccess)
        rc = fail(PSIM_ERR_HIP, "generating the resized set failed: %s",
                  hipGetErrorString(hipGetLastError()));
    if (rc != PSIM_OK) {
        // With a new buffer the old particles are untouched: back to them.  Without one the
        // generators wrote in place; the old count then covers partly regenerated particles.
        (void)hipStreamSynchronize(h->stream);
        if (grown) {
            (void)hipFree(h->slab);
            restore_storage(h, before);
        }
        h->count = old_count;
        return rc;
    }
    if (grown) (void)hipFree(before.slab);
    h->generation_mode = generation_mode;
    return PSIM_OK;
}

PSIM_API int psim_set_resize_semantics(psim_handle *h, int semantics)
{
    CHECK_HANDLE(h);
    if (semantics != PSIM_RESIZE_REGENERATE && semantics != PSIM_RESIZE_APPEND)
        return fail(PSIM_ERR_INVALID, "unknown resize semantics %d", semantics);
    if (semantics == PSIM_RESIZE_APPEND && h->global_count)
        return fail(PSIM_ERR_INVALID, "a shard cannot be resized; recreate the shards");
    h->resize_semantics = semantics;
    return PSIM_OK;
}

PSIM_API int psim_resize_semantics(const psim_handle *h)
{
    return h ? h->resize_semantics : PSIM_RESIZE_REGENERATE;
}

PSIM_API int psim_reset(psim_handle *h, int generation_mode)
{
    CHECK_HANDLE(h);
    if (generation_mode < PSIM_GEN_HOLLOW || generation_mode > PSIM_GEN_UNIFORM)
        return fail(PSIM_ERR_INVALID, "unknown generation mode %d", generation_mode);
    int rc = generate(h, generation_mode);           // compute.rs:213-215
    if (rc != PSIM_OK) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->generation_mode = generation_mode;            // compute.rs:212
    return PSIM_OK;
}

PSIM_API uint32_t psim_particle_count(const psim_handle *h) { return h ? h->count : 0; }

PSIM_API int psim_method(const psim_handle *h)
{
    (void)h;
    return PSIM_METHOD_HIP;
}

PSIM_API int psim_is_paused(const psim_handle *h) { return h && h->paused ? 1 : 0; }

PSIM_API void psim_set_paused(psim_handle *h, int paused)
{
    if (h) h->paused = paused != 0;
}

// ---- particle data in and out -------------------------------------------------

namespace {

// Transposes chunk after chunk into the staging buffer and copies it out, all in stream
// order (the next chunk's transposition waits for the previous chunk's copy by itself).
int enqueue_download_aos(psim_handle *h, psim_particle *dst_host, uint64_t first, uint64_t n)
{
    int rc = check_range(h, first, n);
    if (rc != PSIM_OK) return rc;
    if (n == 0) return PSIM_OK;
    if (!dst_host) return fail(PSIM_ERR_INVALID, "null destination");
    rc = ensure_stage(h);
    if (rc != PSIM_OK) return rc;
    for (uint64_t done = 0; done < n; done += kStageParticles) {
        const uint32_t m = (uint32_t)((n - done) < kStageParticles ? (n - done) : kStageParticles);
        HIP_TRY(psim::launch_export_aos(h->soa, (uint32_t)(first + done), m, h->aos_stage,
                                        h->stream));
        HIP_TRY(hipMemcpyAsync(dst_host + done, h->aos_stage, (size_t)m * sizeof(psim_particle),
                               hipMemcpyDeviceToHost, h->stream));
    }
    return PSIM_OK;
}

// True when `p` is ordinary (pageable) host memory, which the runtime copies out of the
// device at a fraction of the link rate.
bool is_pageable(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();                      // unknown to the runtime: plain malloc memory
        return true;
    }
    return a.type == hipMemoryTypeUnregistered;
}

void parallel_memcpy(void *dst, const void *src, size_t bytes)
{
    const size_t kThreads = 4, kMin = 4u << 20;
    if (bytes < kMin) {
        memcpy(dst, src, bytes);
        return;
    }
    const size_t per = (bytes / kThreads + 4095) & ~(size_t)4095;
    std::vector<std::thread> pool;
    for (size_t t = 1; t < kThreads && t * per < bytes; ++t) {
        const size_t lo = t * per, len = bytes - lo < per ? bytes - lo : per;
        try {
            pool.emplace_back(memcpy, (char *)dst + lo, (const char *)src + lo, len);
        } catch (...) {
            memcpy((char *)dst + lo, (const char *)src + lo, len);
        }
    }
    memcpy(dst, src, per < bytes ? per : bytes);
    for (auto &t : pool) t.join();
}

// The two page-locked bounce slots and their events, created on first use.
int ensure_bounce(psim_handle *h)
{
    if (!h->bounce)
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h->bounce),
                              2 * (size_t)kBounceParticles * sizeof(psim_particle),
                              hipHostMallocDefault));
    for (auto &e : h->bounce_done)
        if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return PSIM_OK;
}

// The way out through the bounce slots: the device copies slot after slot into the two
// page-locked slots at the link rate while `drain(slot, offset, m)` -- called with
// particles [offset, offset + m) of the request -- disposes of the previous one (a copy
// into pageable memory, a file write).  `drain` returns a status code.
template <typename Drain>
int download_aos_through_slots(psim_handle *h, uint64_t first, uint64_t n, Drain drain)
{
    if (n == 0) return PSIM_OK;
    int rc = ensure_stage(h);
    if (rc != PSIM_OK) return rc;
    const uint64_t slot = kBounceParticles < h->stage_particles / 2 ? kBounceParticles
                                                                     : (h->stage_particles + 1) / 2;
    rc = ensure_bounce(h);
    if (rc != PSIM_OK) return rc;
    uint64_t prev_off = 0, prev_m = 0, c = 0;
    for (uint64_t done = 0; done < n; done += slot, ++c) {
        const uint32_t m = (uint32_t)((n - done) < slot ? (n - done) : slot);
        const int b = (int)(c & 1);
        HIP_TRY(psim::launch_export_aos(h->soa, (uint32_t)(first + done), m,
                                        h->aos_stage + b * slot, h->stream));
        HIP_TRY(hipMemcpyAsync(h->bounce + (uint64_t)b * kBounceParticles, h->aos_stage + b * slot,
                               (size_t)m * sizeof(psim_particle), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipEventRecord(h->bounce_done[b], h->stream));
        if (c > 0) {                                   // the previous slot, while this one travels
            HIP_TRY(hipEventSynchronize(h->bounce_done[b ^ 1]));
            rc = drain(h->bounce + (uint64_t)(b ^ 1) * kBounceParticles, prev_off, prev_m);
            if (rc != PSIM_OK) {
                (void)hipStreamSynchronize(h->stream);
                return rc;
            }
        }
        prev_off = done, prev_m = m;
    }
    const int last = (int)((c - 1) & 1);
    HIP_TRY(hipEventSynchronize(h->bounce_done[last]));
    return drain(h->bounce + (uint64_t)last * kBounceParticles, prev_off, prev_m);
}

// The way in, the same way round: `fill(slot, offset, m)` fills a page-locked slot with
// particles [offset, offset + m) of the request while the previous slot travels to the
// device and is transposed there.
template <typename Fill>
int upload_aos_through_slots(psim_handle *h, uint64_t first, uint64_t n, Fill fill)
{
    if (n == 0) return PSIM_OK;
    int rc = ensure_stage(h);
    if (rc != PSIM_OK) return rc;
    const uint64_t slot = kBounceParticles < h->stage_particles / 2 ? kBounceParticles
                                                                     : (h->stage_particles + 1) / 2;
    rc = ensure_bounce(h);
    if (rc != PSIM_OK) return rc;
    uint64_t c = 0;
    for (uint64_t done = 0; done < n; done += slot, ++c) {
        const uint32_t m = (uint32_t)((n - done) < slot ? (n - done) : slot);
        const int b = (int)(c & 1);
        if (c >= 2) HIP_TRY(hipEventSynchronize(h->bounce_done[b]));      // slot consumed
        psim_particle *host = h->bounce + (uint64_t)b * kBounceParticles;
        rc = fill(host, done, (uint64_t)m);
        if (rc != PSIM_OK) {
            (void)hipStreamSynchronize(h->stream);
            return rc;
        }
        HIP_TRY(hipMemcpyAsync(h->aos_stage + b * slot, host, (size_t)m * sizeof(psim_particle),
                               hipMemcpyHostToDevice, h->stream));
        HIP_TRY(psim::launch_import_aos(h->soa, (uint32_t)(first + done), m,
                                        h->aos_stage + b * slot, h->stream));
        HIP_TRY(hipEventRecord(h->bounce_done[b], h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PSIM_OK;
}

// Pageable host memory on either side (the runtime's own pageable path reaches 10.7 GB/s
// down; these are bound by the host copy).
int download_aos_bounced(psim_handle *h, psim_particle *dst, uint64_t first, uint64_t n)
{
    return download_aos_through_slots(
        h, first, n, [dst](const psim_particle *slot, uint64_t off, uint64_t m) {
            parallel_memcpy(dst + off, slot, (size_t)m * sizeof(psim_particle));
            return (int)PSIM_OK;
        });
}

int upload_aos_bounced(psim_handle *h, const psim_particle *src, uint64_t first, uint64_t n)
{
    return upload_aos_through_slots(
        h, first, n, [src](psim_particle *slot, uint64_t off, uint64_t m) {
            parallel_memcpy(slot, src + off, (size_t)m * sizeof(psim_particle));
            return (int)PSIM_OK;
        });
}

// Plain device <-> pageable host copies (the SoA transfers) through the same two slots.
int bounce_bytes(psim_handle *h, void *host, void *dev, size_t bytes, bool to_host)
{
    int rc = ensure_bounce(h);
    if (rc != PSIM_OK) return rc;
    const size_t slot = (size_t)kBounceParticles * sizeof(psim_particle);
    char *pin = reinterpret_cast<char *>(h->bounce);
    size_t prev_off = 0, prev_len = 0, c = 0;
    for (size_t done = 0; done < bytes; done += slot, ++c) {
        const size_t len = bytes - done < slot ? bytes - done : slot;
        const int b = (int)(c & 1);
        if (to_host) {
            HIP_TRY(hipMemcpyAsync(pin + b * slot, (char *)dev + done, len, hipMemcpyDeviceToHost,
                                   h->stream));
            HIP_TRY(hipEventRecord(h->bounce_done[b], h->stream));
            if (c > 0) {
                HIP_TRY(hipEventSynchronize(h->bounce_done[b ^ 1]));
                parallel_memcpy((char *)host + prev_off, pin + (b ^ 1) * slot, prev_len);
            }
            prev_off = done, prev_len = len;
        } else {
            if (c >= 2) HIP_TRY(hipEventSynchronize(h->bounce_done[b]));
            parallel_memcpy(pin + b * slot, (const char *)host + done, len);
            HIP_TRY(hipMemcpyAsync((char *)dev + done, pin + b * slot, len, hipMemcpyHostToDevice,
                                   h->stream));
            HIP_TRY(hipEventRecord(h->bounce_done[b], h->stream));
        }
    }
    if (to_host && c > 0) {
        const int last = (int)((c - 1) & 1);
        HIP_TRY(hipEventSynchronize(h->bounce_done[last]));
        parallel_memcpy((char *)host + prev_off, pin + last * slot, prev_len);
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PSIM_OK;
}

}  // namespace

PSIM_API int psim_download_aos(psim_handle *h, psim_particle *dst_host, uint64_t first, uint64_t n)
{
    CHECK_HANDLE(h);
    if (n && dst_host && check_range(h, first, n) == PSIM_OK && is_pageable(dst_host))
        return download_aos_bounced(h, dst_host, first, n);
    int rc = enqueue_download_aos(h, dst_host, first, n);
    if (rc != PSIM_OK) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PSIM_OK;
}

PSIM_API int psim_download_aos_async(psim_handle *h, psim_particle *dst_pinned, uint64_t first,
                                     uint64_t n)
{
    CHECK_HANDLE(h);
    int rc = enqueue_download_aos(h, dst_pinned, first, n);
    if (rc != PSIM_OK || !dst_pinned) return rc;     // (an empty range may come without a buffer)
    // Remember where this download ends in the stream, under its destination: the entry of the
    // same destination, else a free one, else the oldest.
    psim_handle::PendingDownload *slot = nullptr;
    for (auto &d : h->downloads)
        if (!slot && d.dst == dst_pinned) slot = &d;
    for (auto &d : h->downloads)
        if (!slot && d.dst == nullptr) slot = &d;
    if (!slot) {
        slot = &h->downloads[0];
        for (auto &d : h->downloads)
            if (d.age < slot->age) slot = &d;
    }
    if (!slot->done) HIP_TRY(hipEventCreateWithFlags(&slot->done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(slot->done, h->stream));
    slot->dst = dst_pinned;
    slot->age = ++h->download_clock;
    return PSIM_OK;
}

PSIM_API int psim_download_aos_wait(psim_handle *h, const psim_particle *dst_pinned)
{
    CHECK_HANDLE(h);
    if (!dst_pinned) return fail(PSIM_ERR_INVALID, "null destination");
    for (auto &d : h->downloads)
        if (d.dst == dst_pinned) {
            HIP_TRY(hipEventSynchronize(d.done));
            d.dst = nullptr;
            return PSIM_OK;
        }
    return fail(PSIM_ERR_INVALID, "no asynchronous download into %p is pending (only the four "
                "most recent destinations are tracked)", (const void *)dst_pinned);
}

PSIM_API int psim_upload_aos(psim_handle *h, const psim_particle *src_host, uint64_t first,
                             uint64_t n)
{
    CHECK_HANDLE(h);
    int rc = check_range(h, first, n);
    if (rc != PSIM_OK) return rc;
    if (n == 0) return PSIM_OK;
    if (!src_host) return fail(PSIM_ERR_INVALID, "null source");
    h->colours_changed = true;                   // the caller's colours, whatever they are
    if (is_pageable(src_host)) return upload_aos_bounced(h, src_host, first, n);
    return upload_aos_chunked(h, src_host, first, n);
}

PSIM_API int psim_host_alloc(void **out, uint64_t bytes)
{
    if (!out) return fail(PSIM_ERR_INVALID, "null output pointer");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(PSIM_ERR_NO_DEVICE, "no HIP device: page-locked memory needs one");
    void *p = nullptr;
    HIP_TRY(hipHostMalloc(&p, bytes ? (size_t)bytes : 1, hipHostMallocPortable));
    *out = p;
    return PSIM_OK;
}

PSIM_API void psim_host_free(void *ptr)
{
    if (ptr) (void)hipHostFree(ptr);
}

PSIM_API int psim_download_soa(psim_handle *h, float *dst_host, uint64_t first, uint64_t n)
{
    CHECK_HANDLE(h);
    int rc = check_range(h, first, n);
    if (rc != PSIM_OK) return rc;
    if (n == 0) return PSIM_OK;
    if (!dst_host) return fail(PSIM_ERR_INVALID, "null destination");
    float *arr[6] = {h->soa.px, h->soa.py, h->soa.pz, h->soa.vx, h->soa.vy, h->soa.vz};
    if (is_pageable(dst_host)) {
        for (int k = 0; k < 6; ++k) {
            rc = bounce_bytes(h, dst_host + (uint64_t)k * n, arr[k] + first, n * sizeof(float), true);
            if (rc != PSIM_OK) return rc;
        }
        return PSIM_OK;
    }
    for (int k = 0; k < 6; ++k)
        HIP_TRY(hipMemcpyAsync(dst_host + (uint64_t)k * n, arr[k] + first, n * sizeof(float),
                               hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PSIM_OK;
}

PSIM_API int psim_upload_soa(psim_handle *h, const float *src_host, uint64_t first, uint64_t n)
{
    CHECK_HANDLE(h);
    int rc = check_range(h, first, n);
    if (rc != PSIM_OK) return rc;
    if (n == 0) return PSIM_OK;
    if (!src_host) return fail(PSIM_ERR_INVALID, "null source");
    float *arr[6] = {h->soa.px, h->soa.py, h->soa.pz, h->soa.vx, h->soa.vy, h->soa.vz};
    if (is_pageable(src_host)) {
        for (int k = 0; k < 6; ++k) {
            rc = bounce_bytes(h, const_cast<float *>(src_host) + (uint64_t)k * n, arr[k] + first,
                              n * sizeof(float), false);
            if (rc != PSIM_OK) return rc;
        }
        return PSIM_OK;
    }
    for (int k = 0; k < 6; ++k)
        HIP_TRY(hipMemcpyAsync(arr[k] + first, src_host + (uint64_t)k * n, n * sizeof(float),
                               hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PSIM_OK;
}

PSIM_API int psim_export_aos_device(psim_handle *h, psim_particle *dst_device, uint64_t first,
                                    uint64_t n)
{
    CHECK_HANDLE(h);
    int rc = check_range(h, first, n);
    if (rc != PSIM_OK) return rc;
    if (n == 0) return PSIM_OK;
    if (!dst_device || !aligned16(dst_device))
        return fail(PSIM_ERR_INVALID, "AoS device buffer must be non-null and 16-byte aligned");
    // The one destination whose size is known here: a caller that kept the pointer of the
    // shared record buffer across a resize that outgrew it must not write past its end.
    if (h->shared_aos) {
        const char *lo = reinterpret_cast<const char *>(h->shared_aos), *hi = lo + h->shared_bytes;
        const char *d = reinterpret_cast<const char *>(dst_device);
        if (d >= lo && d < hi && n * sizeof(psim_particle) > (uint64_t)(hi - d))
            return fail(PSIM_ERR_INVALID, "%llu records do not fit the shared record buffer from "
                        "there: call psim_shared_aos again after the set has grown",
                        (unsigned long long)n);
    }
    HIP_TRY(psim::launch_export_aos(h->soa, (uint32_t)first, (uint32_t)n, dst_device, h->stream));
    return PSIM_OK;
}

PSIM_API int psim_import_aos_device(psim_handle *h, const psim_particle *src_device,
                                    uint64_t first, uint64_t n)
{
    CHECK_HANDLE(h);
    int rc = check_range(h, first, n);
    if (rc != PSIM_OK) return rc;
    if (n == 0) return PSIM_OK;
    if (!src_device || !aligned16(src_device))
        return fail(PSIM_ERR_INVALID, "AoS device buffer must be non-null and 16-byte aligned");
    h->colours_changed = true;
    HIP_TRY(psim::launch_import_aos(h->soa, (uint32_t)first, (uint32_t)n, src_device, h->stream));
    return PSIM_OK;
}

namespace {

void release_shared_aos(psim_handle *h)
{
    if (h->shared_aos) {
        (void)hipMemUnmap(h->shared_aos, h->shared_bytes);
        (void)hipMemAddressFree(h->shared_aos, h->shared_bytes);
    }
    if (h->shared_memory) (void)hipMemRelease(h->shared_memory);
    h->shared_aos = nullptr;
    h->shared_memory = nullptr;
    h->shared_bytes = 0;
}

// hipMalloc serves small requests out of larger allocations, and a descriptor exported for
// such a range refers to the whole underlying allocation (an importer would see other data at
// offset 0).  So the shared buffer is a physical allocation of its own, mapped at an address
// range of its own.
int create_shared_aos(psim_handle *h)
{
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = h->device;
    prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    size_t granularity = 0;
    HIP_TRY(hipMemGetAllocationGranularity(&granularity, &prop, hipMemAllocationGranularityMinimum));
    if (granularity == 0) granularity = 2u << 20;
    const uint64_t want = (uint64_t)h->capacity * sizeof(psim_particle);
    const uint64_t bytes = (want + granularity - 1) / granularity * granularity;
    HIP_TRY(hipMemCreate(&h->shared_memory, (size_t)bytes, &prop, 0));
    void *va = nullptr;
    hipError_t e = hipMemAddressReserve(&va, (size_t)bytes, granularity, nullptr, 0);
    if (e == hipSuccess) {
        e = hipMemMap(va, (size_t)bytes, 0, h->shared_memory, 0);
        if (e != hipSuccess) (void)hipMemAddressFree(va, (size_t)bytes);
    }
    if (e != hipSuccess) {
        (void)hipMemRelease(h->shared_memory);
        h->shared_memory = nullptr;
        return fail(PSIM_ERR_HIP, "mapping the shared record buffer failed: %s", hipGetErrorString(e));
    }
    h->shared_aos = static_cast<psim_particle *>(va);
    h->shared_bytes = bytes;
    hipMemAccessDesc access;
    memset(&access, 0, sizeof(access));
    access.location = prop.location;
    access.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(va, (size_t)bytes, &access, 1);
    if (e != hipSuccess) {
        release_shared_aos(h);
        return fail(PSIM_ERR_HIP, "hipMemSetAccess on the shared record buffer failed: %s",
                    hipGetErrorString(e));
    }
    return PSIM_OK;
}

}  // namespace

PSIM_API int psim_shared_aos(psim_handle *h, psim_particle **device_ptr_out, int *dmabuf_fd_out,
                             uint64_t *bytes_out)
{
    CHECK_HANDLE(h);
    if (!device_ptr_out || !dmabuf_fd_out || !bytes_out)
        return fail(PSIM_ERR_INVALID, "null output pointer");
    *device_ptr_out = nullptr, *dmabuf_fd_out = -1, *bytes_out = 0;
    if (h->shared_aos && h->shared_bytes < (uint64_t)h->capacity * sizeof(psim_particle)) {
        HIP_TRY(hipStreamSynchronize(h->stream));    // the set has outgrown it; an export may still write it
        release_shared_aos(h);
    }
    if (!h->shared_aos) {
        const int rc = create_shared_aos(h);
        if (rc != PSIM_OK) return rc;
    }
    int fd = -1;
    HIP_TRY(hipMemExportToShareableHandle(&fd, h->shared_memory, hipMemHandleTypePosixFileDescriptor, 0));
    *device_ptr_out = h->shared_aos;
    *dmabuf_fd_out = fd;
    *bytes_out = h->shared_bytes;
    return PSIM_OK;
}

PSIM_API int psim_set_color_semantics(psim_handle *h, int semantics)
{
    CHECK_HANDLE(h);
    if (semantics != PSIM_COLOR_KEEP && semantics != PSIM_COLOR_RESET)
        return fail(PSIM_ERR_INVALID, "unknown colour semantics %d", semantics);
    h->colour_semantics = semantics;
    return PSIM_OK;
}

PSIM_API int psim_color_semantics(const psim_handle *h)
{
    return h ? h->colour_semantics : PSIM_COLOR_KEEP;
}

PSIM_API int psim_device_soa(psim_handle *h, psim_soa_view *view)
{
    if (!h || !view) return fail(PSIM_ERR_INVALID, "null argument");
    h->colours_changed = true;                   // the caller may write through these pointers
    view->px = h->soa.px; view->py = h->soa.py; view->pz = h->soa.pz;
    view->vx = h->soa.vx; view->vy = h->soa.vy; view->vz = h->soa.vz;
    view->color = h->soa.color;
    view->initial_color = h->soa.initial_color;
    view->padding = h->soa.padding;
    view->count = h->count;
    view->capacity = h->capacity;
    return PSIM_OK;
}

PSIM_API int psim_checksum(psim_handle *h, uint64_t out[6])
{
    CHECK_HANDLE(h);
    if (!out) return fail(PSIM_ERR_INVALID, "null output");
    // One 256-thread workgroup per CU reads fastest (100 M particles, GB/s of the 24 B read:
    // 128 workgroups 4.2 k, 256 6.8 k, 512 6.4 k, 4096 5.8 k).
    int blocks = psim::knob(psim::KNOB_CHECKSUM_BLOCKS);
    if (blocks < 1) blocks = h->num_cus > 0 ? h->num_cus : 256;
    HIP_TRY(psim::launch_checksum(h->soa, h->count, h->checksum_dev, (uint32_t)blocks, h->stream));
    HIP_TRY(hipMemcpyAsync(out, h->checksum_dev, 6 * sizeof(uint64_t), hipMemcpyDeviceToHost,
                           h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PSIM_OK;
}

// ---- checkpoints ----------------------------------------------------------------

namespace {

using psim_ckpt::File;
using psim_ckpt::kCheckpointMagic;
using psim_ckpt::read_all;
using psim_ckpt::read_header;
using psim_ckpt::record_offset;
using psim_ckpt::write_all;

}  // namespace

PSIM_API int psim_checkpoint_write(psim_handle *h, const char *path, uint64_t step)
{
    CHECK_HANDLE(h);
    if (!path) return fail(PSIM_ERR_INVALID, "null path");
    File f;
    f.fd = open(path, O_WRONLY | O_CREAT | O_CLOEXEC, 0644);
    if (f.fd < 0) return fail(PSIM_ERR_IO, "cannot open %s for writing: %s", path, strerror(errno));
    const bool shard = h->global_count != 0;
    const uint64_t total = shard ? h->global_count : (uint64_t)h->count;
    const uint64_t first = h->global_first;
    psim_checkpoint_header hdr;
    memset(&hdr, 0, sizeof(hdr));
    // No valid header while records change.  A shard does this too (it is idempotent, and all
    // shards do it before the barrier that precedes the commit): a caller that forgot
    // psim_checkpoint_begin cannot leave an older, valid header over half-rewritten records.
    int rc = write_all(f.fd, &hdr, sizeof(hdr), 0, path);
    if (rc != PSIM_OK) return rc;
    const int fd = f.fd;
    rc = download_aos_through_slots(
        h, 0, h->count, [fd, first, path](const psim_particle *slot, uint64_t off, uint64_t m) {
            return write_all(fd, slot, (size_t)m * sizeof(psim_particle), record_offset(first + off),
                             path);
        });
    if (rc != PSIM_OK) return rc;
    if (first + h->count == total && ftruncate(f.fd, (off_t)record_offset(total)) != 0)
        return fail(PSIM_ERR_IO, "cannot size %s: %s", path, strerror(errno));
    if (!shard) {                                   // shards: psim_checkpoint_commit, once all are in
        hdr = psim_ckpt::make_header(total, step, (uint32_t)h->generation_mode,
                                     (uint32_t)h->colour_semantics);
        rc = write_all(f.fd, &hdr, sizeof(hdr), 0, path);
        if (rc != PSIM_OK) return rc;
    }
    const int cfd = f.fd;
    f.fd = -1;
    if (close(cfd) != 0) return fail(PSIM_ERR_IO, "closing %s failed: %s", path, strerror(errno));
    return PSIM_OK;
}

PSIM_API int psim_checkpoint_begin(const char *path)
{
    if (!path) return fail(PSIM_ERR_INVALID, "null path");
    return psim_ckpt::begin_file(path);
}

PSIM_API int psim_checkpoint_commit(const psim_handle *h, const char *path, uint64_t step)
{
    if (!h) return fail(PSIM_ERR_INVALID, "null handle");
    if (!path) return fail(PSIM_ERR_INVALID, "null path");
    const uint64_t total = h->global_count ? h->global_count : (uint64_t)h->count;
    return psim_ckpt::commit_file(path, psim_ckpt::make_header(total, step,
                                                              (uint32_t)h->generation_mode,
                                                              (uint32_t)h->colour_semantics));
}

PSIM_API int psim_checkpoint_read(psim_handle *h, const char *path, uint64_t *step_out)
{
    CHECK_HANDLE(h);
    if (!path) return fail(PSIM_ERR_INVALID, "null path");
    File f;
    f.fd = open(path, O_RDONLY | O_CLOEXEC);
    if (f.fd < 0) return fail(PSIM_ERR_IO, "cannot open %s: %s", path, strerror(errno));
    psim_checkpoint_header hdr;
    int rc = read_header(f.fd, path, &hdr);
    if (rc != PSIM_OK) return rc;
    const uint64_t total = h->global_count ? h->global_count : (uint64_t)h->count;
    if (hdr.particle_count != total)
        return fail(PSIM_ERR_INVALID, "%s holds %llu particles, the simulation %llu", path,
                    (unsigned long long)hdr.particle_count, (unsigned long long)total);
    const int fd = f.fd;
    const uint64_t first = h->global_first;
    h->colours_changed = true;                   // the file's colours, whatever they are
    rc = upload_aos_through_slots(
        h, 0, h->count, [fd, first, path](psim_particle *slot, uint64_t off, uint64_t m) {
            return read_all(fd, slot, (size_t)m * sizeof(psim_particle), record_offset(first + off),
                            path);
        });
    if (rc != PSIM_OK) return rc;
    if (step_out) *step_out = hdr.step;
    return PSIM_OK;
}

PSIM_API int psim_checkpoint_info(const char *path, psim_checkpoint_header *out)
{
    if (!path || !out) return fail(PSIM_ERR_INVALID, "null argument");
    File f;
    f.fd = open(path, O_RDONLY | O_CLOEXEC);
    if (f.fd < 0) return fail(PSIM_ERR_IO, "cannot open %s: %s", path, strerror(errno));
    return read_header(f.fd, path, out);
}

// ---- streams ------------------------------------------------------------------

PSIM_API int psim_set_stream(psim_handle *h, void *hip_stream)
{
    CHECK_HANDLE(h);
    HIP_TRY(hipStreamSynchronize(h->stream));        // drain work queued on the old stream
    h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
    return PSIM_OK;
}

PSIM_API void *psim_get_stream(const psim_handle *h) { return h ? h->stream : nullptr; }

PSIM_API int psim_device(const psim_handle *h) { return h ? h->device : -1; }

PSIM_API int psim_device_count(void)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return ndev < 0 ? 0 : ndev;
}

// ---- timing -------------------------------------------------------------------

PSIM_API int psim_timer_start(psim_handle *h)
{
    CHECK_HANDLE(h);
    if (!h->ev_start) HIP_TRY(hipEventCreate(&h->ev_start));
    if (!h->ev_stop) HIP_TRY(hipEventCreate(&h->ev_stop));
    HIP_TRY(hipEventRecord(h->ev_start, h->stream));
    return PSIM_OK;
}

PSIM_API int psim_timer_stop(psim_handle *h, float *elapsed_ms)
{
    CHECK_HANDLE(h);
    if (!elapsed_ms) return fail(PSIM_ERR_INVALID, "null output");
    if (!h->ev_start || !h->ev_stop)
        return fail(PSIM_ERR_INVALID, "psim_timer_stop without psim_timer_start");
    HIP_TRY(hipEventRecord(h->ev_stop, h->stream));
    HIP_TRY(hipEventSynchronize(h->ev_stop));
    HIP_TRY(hipEventElapsedTime(elapsed_ms, h->ev_start, h->ev_stop));
    return PSIM_OK;
}

// ---- errors -------------------------------------------------------------------

PSIM_API int psim_launch_shape(uint32_t count, uint32_t n_attractors, psim_launch_shape_info *out)
{
    if (!out) return fail(PSIM_ERR_INVALID, "null output");
    if (n_attractors > PSIM_MAX_ATTRACTORS)
        return fail(PSIM_ERR_INVALID, "%u attractors exceed the maximum of %u", n_attractors,
                    PSIM_MAX_ATTRACTORS);
    const psim::LaunchTuning t = psim::choose_tuning(count, n_attractors);
    out->access = t.access;
    out->workgroup_size = PSIM_UPDATE_BLOCK;
    out->workgroups = psim::update_grid(count, t);
    out->reserve_lds = t.reserve_lds;
    out->reserve_lds_colour = t.reserve_lds_colour;
    out->resident_per_cu = t.reserve_lds ? 163840u / t.reserve_lds : 0u;
    return PSIM_OK;
}

PSIM_API int psim_launch_occupancy(psim_handle *h, uint32_t n_attractors, uint32_t color_mode,
                                   uint32_t *workgroups_per_cu, uint32_t *lds_bytes_per_workgroup)
{
    CHECK_HANDLE(h);
    if (!workgroups_per_cu) return fail(PSIM_ERR_INVALID, "null output");
    if (n_attractors > PSIM_MAX_ATTRACTORS)
        return fail(PSIM_ERR_INVALID, "%u attractors exceed the maximum of %u", n_attractors,
                    PSIM_MAX_ATTRACTORS);
    // 0 attractors: psim_update without dragging; 1: psim_update; more: psim_update_multi
    const int force = n_attractors == 0 ? 0 : (n_attractors == 1 ? 1 : 2);
    int blocks = 0;
    uint32_t lds = 0;
    HIP_TRY(psim::update_occupancy(color_mode, force, n_attractors,
                                   tuning_for(h->count, n_attractors ? n_attractors : 1, h->overrides),
                                   &blocks, &lds));
    *workgroups_per_cu = blocks < 0 ? 0u : (uint32_t)blocks;
    if (lds_bytes_per_workgroup) *lds_bytes_per_workgroup = lds;
    return PSIM_OK;
}

PSIM_API const char *psim_last_error(void) { return g_error; }

PSIM_API const char *psim_version(void) { return "psim-hip 0.3 (gfx950)"; }
