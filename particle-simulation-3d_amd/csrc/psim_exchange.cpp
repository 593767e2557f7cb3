// psim_exchange.cpp -- the multi-attractor step of a particle set that is sharded over one
// process per GPU: the psim_exchange_* entry points of include/psim.h.
//
// Every rank owns one attractor.  A step all-gathers the ranks' 32-byte records with RCCL
// (over xGMI between the GPUs of a node) on a communication stream of its own and runs the
// multi-attractor update kernel on the handle's stream, reading the gathered list straight
// from the collective's receive buffer.  The receive buffers form a ring of 2 x 8 lists, so the
// all-gather of step t + 1 runs while the kernel of step t does:
//
//   communication stream : [first step of a half of the ring: wait until the kernels that
//                          read that half, 9 to 16 steps ago, are done]
//                          | store own record -> list[t%16][rank]
//                          | all-gather in place over list[t%16] | record gathered[t%16]
//   handle's stream      : wait gathered[t%16] | update kernel reading list[t%16]
//                          | [last step of a half: record kernel_done[half]]
//
// Every packet in the handle's queue besides the kernel costs device time between two
// kernels (6.25 M particles: a wait plus a record per step add 8-11 us to a 42 us kernel);
// with the ring the record is needed once per 8 steps.  The wait goes too when the exchange
// is host-paced (psim_exchange_set_pacing): the calling thread waits for the gather, which has
// run under the previous kernel, and the kernel is enqueued directly behind its predecessor
// (44.7 us per step against 47.5 device-paced and 42.3 for the bare kernel).  That holds while
// the gather finds room beside a running kernel, which a collective as large as RCCL's device
// kernel does not (it starts when the kernel's grid runs dry: scripts/exp_collective_overlap.hip).
// Pipelined pacing is host pacing one call late: a call enqueues its gather FIRST, then waits
// for the previous step's gather and enqueues the previous step's kernel, so kernels still run
// back to back and every gather is already waiting when the kernel before it ends.
//
// The exchange moves 32 bytes per rank, so it is bound by latency, not by link bandwidth;
// what matters is that nothing in a step waits on the host.  The reference has exactly one
// attractor (src/simulation/mod.rs:65, src/simulation/cpu.rs:58) and a single process, so
// none of this has a counterpart there; with one rank the step equals psim_update with that
// rank's attractor.
//
// This file holds no GPU code of its own: records are stored and particles stepped through
// the per-handle entry points.  RCCL is loaded on first use (librccl.so.1 is half a gigabyte;
// callers that never exchange never map it), taking the copy a host program has already
// loaded when there is one.
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <mutex>
#include <new>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "psim.h"
#include "psim_tuning.h"

#define PSIM_API extern "C" __attribute__((visibility("default")))

static_assert(PSIM_EXCHANGE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id is RCCL's unique id");

// The error sink of the library (psim_api.hip).
int psim_fail(int code, const char *fmt, ...);

namespace {

// ---- RCCL, loaded on first use ----------------------------------------------------

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t,
                              hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    // The way out of a collective that will never finish (a rank died).  Optional: a library
    // without them still exchanges, it just cannot give up.
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;
    char why[256] = "";
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl()
{
    Rccl &r = g_rccl;
    // a copy that is already mapped (a host program that uses RCCL itself) wins
    const char *named = getenv("PSIM_RCCL_LIBRARY");
    if (named && *named) r.lib = dlopen(named, RTLD_NOW | RTLD_GLOBAL);
    if (!r.lib) r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!r.lib) r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!r.lib) r.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!r.lib) {
        snprintf(r.why, sizeof(r.why), "cannot load librccl.so.1: %s", dlerror());
        return;
    }
#define PSIM_RCCL_SYMBOL(field, name)                                              \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name));             \
    if (!r.field) {                                                                \
        snprintf(r.why, sizeof(r.why), "librccl.so.1 lacks %s", name);             \
        return;                                                                    \
    }
    PSIM_RCCL_SYMBOL(GetUniqueId, "ncclGetUniqueId")
    PSIM_RCCL_SYMBOL(CommInitRank, "ncclCommInitRank")
    PSIM_RCCL_SYMBOL(CommDestroy, "ncclCommDestroy")
    PSIM_RCCL_SYMBOL(AllGather, "ncclAllGather")
    PSIM_RCCL_SYMBOL(GetErrorString, "ncclGetErrorString")
#undef PSIM_RCCL_SYMBOL
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.lib, "ncclCommAbort"));
    r.CommGetAsyncError =
        reinterpret_cast<decltype(r.CommGetAsyncError)>(dlsym(r.lib, "ncclCommGetAsyncError"));
}

// Null (with the reason recorded) when RCCL is not usable.
const Rccl *rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    if (g_rccl.why[0]) {
        psim_fail(PSIM_ERR_COMM, "%s", g_rccl.why);
        return nullptr;
    }
    return &g_rccl;
}

#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess)                                                            \
            return psim_fail(e_ == hipErrorOutOfMemory ? PSIM_ERR_NOMEM : PSIM_ERR_HIP,  \
                             "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),      \
                             __FILE__, __LINE__);                                        \
    } while (0)

#define RCCL_TRY(r, expr)                                                                \
    do {                                                                                 \
        ncclResult_t n_ = (expr);                                                        \
        if (n_ != ncclSuccess)                                                           \
            return psim_fail(PSIM_ERR_COMM, "%s failed: %s (%s:%d)", #expr,              \
                             (r)->GetErrorString(n_), __FILE__, __LINE__);               \
    } while (0)

constexpr int kRecordWords = sizeof(psim_attractor) / 4;     // 8 four-byte words per record

}  // namespace

struct psim_exchange {
    psim_handle *h = nullptr;
    int device = 0;
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t comm_stream = nullptr;
    psim_attractor *ring = nullptr;                   // kRing lists of world records each
    hipEvent_t gathered[16] = {};                     // per list: its all-gather has finished
    hipEvent_t kernel_done[2] = {nullptr, nullptr};   // per half: its last kernel has finished
    uint64_t t = 0;                                   // steps taken
    int pacing = PSIM_EXCHANGE_DEVICE_PACED;          // see psim_exchange_set_pacing()
    bool pending = false;                             // pipelined: a step's kernel not yet enqueued
    psim_sim_params pending_params = {};              //   its parameters
    uint32_t pending_slot = 0;                        //   and the list its gather fills
    uint32_t timeout_ms = 0;                          // 0 = wait for ever, see psim_exchange_set_timeout()
    bool broken = false;                              // a step failed or timed out: unusable
    char why_broken[200] = "";
};

namespace {
constexpr uint32_t kHalf = 8, kRing = 2 * kHalf;
static_assert(sizeof(psim_exchange::gathered) / sizeof(hipEvent_t) == kRing, "one event per list");

int64_t now_ms()
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (int64_t)t.tv_sec * 1000 + t.tv_nsec / 1000000;
}

// The exchange is finished: the ranks are no longer in step, so nothing collective may follow.
// The communicator is aborted, which lets RCCL's kernels on the device leave their waits, so
// that the streams drain and the process can tear down; the lists they leave are garbage.
int give_up(psim_exchange *x, const char *why)
{
    if (!x->broken) {
        x->broken = true;
        snprintf(x->why_broken, sizeof(x->why_broken), "%s", why);
        if (x->comm && g_rccl.CommAbort) {
            (void)g_rccl.CommAbort(x->comm);          // frees the communicator
            x->comm = nullptr;
        }
    }
    return psim_fail(PSIM_ERR_COMM, "rank %d of %d: %s; the exchange cannot be used any more",
                     x->rank, x->world, x->why_broken);
}

// What the communicator reports about collectives already enqueued (a peer that went away,
// a network error): false when it reports a failure.
bool comm_healthy(const psim_exchange *x, char *why, size_t why_bytes)
{
    if (!x->comm || !g_rccl.CommGetAsyncError) return true;
    ncclResult_t state = ncclSuccess;
    const ncclResult_t asked = g_rccl.CommGetAsyncError(x->comm, &state);
    if (asked == ncclSuccess && (state == ncclSuccess || state == ncclInProgress)) return true;
    snprintf(why, why_bytes, "the communicator reports \"%s\"",
             g_rccl.GetErrorString(asked != ncclSuccess ? asked : state));
    return false;
}

// Waits until `ready()` (a hipEventQuery / hipStreamQuery) stops saying hipErrorNotReady.
// Without a timeout this is never called: the plain blocking calls are used.  With one, the
// calling thread polls -- the readiness, and what the communicator says about itself -- until
// the deadline, and gives the exchange up when it passes.
template <typename Ready>
int wait_with_deadline(psim_exchange *x, Ready ready, const char *what)
{
    const int64_t start = now_ms(), deadline = start + (int64_t)x->timeout_ms;
    for (uint32_t spins = 0;; ++spins) {
        const hipError_t e = ready();
        if (e == hipSuccess) return PSIM_OK;
        char why[160];
        if (e != hipErrorNotReady) {
            (void)hipGetLastError();
            snprintf(why, sizeof(why), "%s failed: %s", what, hipGetErrorString(e));
            return give_up(x, why);
        }
        (void)hipGetLastError();                      // NotReady is sticky in the runtime's last error
        if ((spins & 63u) == 63u && !comm_healthy(x, why, sizeof(why))) return give_up(x, why);
        if (now_ms() >= deadline) {
            snprintf(why, sizeof(why), "%s did not finish within %u ms (a rank has gone?)", what,
                     x->timeout_ms);
            return give_up(x, why);
        }
        if (spins > 2000) {                           // a gather takes tens of microseconds: spin
            const struct timespec nap = {0, 50000};   // first, then stop burning the core
            (void)nanosleep(&nap, nullptr);
        }
    }
}

int wait_event(psim_exchange *x, hipEvent_t e, const char *what)
{
    if (x->timeout_ms == 0) {
        HIP_TRY(hipEventSynchronize(e));
        return PSIM_OK;
    }
    return wait_with_deadline(x, [e] { return hipEventQuery(e); }, what);
}

int wait_stream(psim_exchange *x, hipStream_t s, const char *what)
{
    if (x->timeout_ms == 0) {
        HIP_TRY(hipStreamSynchronize(s));
        return PSIM_OK;
    }
    return wait_with_deadline(x, [s] { return hipStreamQuery(s); }, what);
}

}  // namespace

// ---- rendezvous ---------------------------------------------------------------------

PSIM_API int psim_exchange_unique_id(uint8_t id[PSIM_EXCHANGE_ID_BYTES])
{
    if (!id) return psim_fail(PSIM_ERR_INVALID, "null id");
    const Rccl *r = rccl();
    if (!r) return PSIM_ERR_COMM;
    // RCCL answers a machine without a GPU with a "fatal error" banner on the process's
    // output; say it here instead.
    int devices = 0;
    if (hipGetDeviceCount(&devices) != hipSuccess || devices <= 0) {
        (void)hipGetLastError();
        return psim_fail(PSIM_ERR_COMM, "no HIP device visible: RCCL cannot make a communicator id");
    }
    ncclUniqueId u;
    RCCL_TRY(r, r->GetUniqueId(&u));
    memcpy(id, u.internal, PSIM_EXCHANGE_ID_BYTES);
    return PSIM_OK;
}

// The id as a file: rank 0 makes it, writes it under a temporary name and renames it into
// place, so a reader never sees a partial id; the other ranks poll for the file.
PSIM_API int psim_exchange_rendezvous_file(const char *path, int rank,
                                           uint8_t id[PSIM_EXCHANGE_ID_BYTES],
                                           uint32_t timeout_ms)
{
    if (!path || !id) return psim_fail(PSIM_ERR_INVALID, "null argument");
    if (rank < 0) return psim_fail(PSIM_ERR_INVALID, "rank %d", rank);
    if (rank == 0) {
        const int rc = psim_exchange_unique_id(id);
        if (rc != PSIM_OK) return rc;
        char tmp[4096];
        if (snprintf(tmp, sizeof(tmp), "%s.%ld.tmp", path, (long)getpid()) >= (int)sizeof(tmp))
            return psim_fail(PSIM_ERR_INVALID, "path too long");
        const int fd = open(tmp, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0600);
        if (fd < 0) return psim_fail(PSIM_ERR_IO, "cannot create %s: %s", tmp, strerror(errno));
        const ssize_t w = write(fd, id, PSIM_EXCHANGE_ID_BYTES);
        const int closed = close(fd);
        if (w != PSIM_EXCHANGE_ID_BYTES || closed != 0 || rename(tmp, path) != 0) {
            const int err = errno;
            (void)unlink(tmp);
            return psim_fail(PSIM_ERR_IO, "cannot write %s: %s", path, strerror(err));
        }
        return PSIM_OK;
    }
    struct timespec start;
    clock_gettime(CLOCK_MONOTONIC, &start);
    for (;;) {
        const int fd = open(path, O_RDONLY | O_CLOEXEC);
        if (fd >= 0) {
            const ssize_t got = read(fd, id, PSIM_EXCHANGE_ID_BYTES);
            (void)close(fd);
            if (got == PSIM_EXCHANGE_ID_BYTES) return PSIM_OK;
        } else if (errno != ENOENT) {
            return psim_fail(PSIM_ERR_IO, "cannot open %s: %s", path, strerror(errno));
        }
        struct timespec now;
        clock_gettime(CLOCK_MONOTONIC, &now);
        const int64_t ms = (int64_t)(now.tv_sec - start.tv_sec) * 1000 +
                           (now.tv_nsec - start.tv_nsec) / 1000000;
        if (ms >= (int64_t)timeout_ms)
            return psim_fail(PSIM_ERR_COMM, "rank %d: no id at %s after %u ms", rank, path,
                             timeout_ms);
        const struct timespec nap = {0, 1000000};             // 1 ms
        (void)nanosleep(&nap, nullptr);
    }
}

// ---- lifecycle ------------------------------------------------------------------------

PSIM_API void psim_exchange_destroy(psim_exchange *x)
{
    if (!x) return;
    (void)hipSetDevice(x->device);
    // With a timeout the communication stream gets that long to drain; a collective still stuck
    // then is aborted, or the wait below would never return.
    if (x->timeout_ms && !x->broken && x->comm_stream)
        (void)wait_stream(x, x->comm_stream, "the communication stream");
    // Kernels on the handle's stream may still read the lists.  The handle is the caller's and
    // may already be gone, so it is not touched here: the whole device is waited for instead.
    (void)hipDeviceSynchronize();
    if (x->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(x->comm);   // null once aborted
    for (hipEvent_t e : x->gathered)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : x->kernel_done)
        if (e) (void)hipEventDestroy(e);
    if (x->ring) (void)hipFree(x->ring);
    if (x->comm_stream) (void)hipStreamDestroy(x->comm_stream);
    delete x;
}

namespace {

int create_impl(psim_exchange *x, const uint8_t *id)
{
    const Rccl *r = rccl();
    if (!r) return PSIM_ERR_COMM;
    HIP_TRY(hipSetDevice(x->device));
    // The communication stream runs at the priority of the handle's stream.  The highest
    // priority buys nothing measurable (one rank, 6.25 M particles: 47.3 / 44.6 us per step
    // device- / host-paced either way), and a high-priority queue that waits -- at the start of
    // each half of the ring -- for kernels of a normal-priority queue is a priority inversion:
    // with several rank processes sharing one GPU, one rank's waiting communication queue kept
    // another rank's kernels from being scheduled, and the ranks waited for each other for ever.
    int least = 0, greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    const int priority = psim::knob(psim::KNOB_EXCHANGE_COMM_PRIORITY) == 1 ? greatest : 0;
    HIP_TRY(hipStreamCreateWithPriority(&x->comm_stream, hipStreamNonBlocking, priority));
    void *buffers = nullptr;
    const size_t bytes = (size_t)kRing * (size_t)x->world * sizeof(psim_attractor);
    HIP_TRY(hipMalloc(&buffers, bytes));
    x->ring = static_cast<psim_attractor *>(buffers);
    HIP_TRY(hipMemsetAsync(buffers, 0, bytes, x->comm_stream));
    // kernel_done only orders the next writer of a list behind its last reader: no data
    // passes through it, so it needs no system-scope fence.  gathered hands the collective's
    // output (written by other GPUs' kernels too) to the update kernel and keeps the default.
    const int fence_knob = psim::knob(psim::KNOB_EXCHANGE_EVENT_FENCES);
    const int fences = fence_knob < 0 ? 1 : fence_knob;
    for (hipEvent_t &e : x->gathered)
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming |
                                        ((fences & 2) ? hipEventDisableSystemFence : 0)));
    for (hipEvent_t &e : x->kernel_done)
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming |
                                        ((fences & 1) ? hipEventDisableSystemFence : 0)));
    HIP_TRY(hipStreamSynchronize(x->comm_stream));
    ncclUniqueId u;
    memcpy(u.internal, id, PSIM_EXCHANGE_ID_BYTES);
    RCCL_TRY(r, r->CommInitRank(&x->comm, x->world, u, x->rank));
    return PSIM_OK;
}

}  // namespace

PSIM_API int psim_exchange_create(psim_exchange **out, psim_handle *h,
                                  const uint8_t id[PSIM_EXCHANGE_ID_BYTES], int rank, int world)
{
    if (!out) return psim_fail(PSIM_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (!h || !id) return psim_fail(PSIM_ERR_INVALID, "null argument");
    if (world < 1 || (uint32_t)world > PSIM_MAX_ATTRACTORS)
        return psim_fail(PSIM_ERR_INVALID, "%d ranks: one attractor per rank, at most %u", world,
                         PSIM_MAX_ATTRACTORS);
    if (rank < 0 || rank >= world)
        return psim_fail(PSIM_ERR_INVALID, "rank %d outside a world of %d", rank, world);
    psim_exchange *x = new (std::nothrow) psim_exchange();
    if (!x) return psim_fail(PSIM_ERR_NOMEM, "host allocation failed");
    x->h = h;
    x->device = psim_device(h);
    x->rank = rank;
    x->world = world;
    const int rc = create_impl(x, id);
    if (rc != PSIM_OK) {
        psim_exchange_destroy(x);
        return rc;
    }
    *out = x;
    return PSIM_OK;
}

// ---- the step ----------------------------------------------------------------------------

namespace {

// The kernel of the step whose all-gather fills list `slot`, behind that gather.  `wait`: how
// the kernel is ordered behind the gather -- the handle's stream waits (device pacing), or the
// calling thread does and the kernel goes in with nothing in front of it.
int enqueue_kernel(psim_exchange *x, const psim_sim_params *params, uint32_t slot, bool host_waits)
{
    hipStream_t compute = static_cast<hipStream_t>(psim_get_stream(x->h));
    psim_attractor *list = x->ring + (size_t)slot * (size_t)x->world;
    if (host_waits) {
        const int rc = wait_event(x, x->gathered[slot], "the all-gather of a step");
        if (rc != PSIM_OK) return rc;
    } else {
        HIP_TRY(hipStreamWaitEvent(compute, x->gathered[slot], 0));
    }
    const int rc = psim_update_multi(x->h, params, list, (uint32_t)x->world);
    if (rc != PSIM_OK) return rc;
    if (slot % kHalf == kHalf - 1) HIP_TRY(hipEventRecord(x->kernel_done[slot / kHalf], compute));
    return PSIM_OK;
}

// Pipelined pacing: the kernel that the previous call left for this one.
int flush_pending(psim_exchange *x)
{
    if (!x->pending) return PSIM_OK;
    x->pending = false;
    return enqueue_kernel(x, &x->pending_params, x->pending_slot, true);
}

int step_impl(psim_exchange *x, const psim_sim_params *params, const psim_attractor *mine)
{
    const Rccl *r = &g_rccl;                         // loaded: the exchange exists
    HIP_TRY(hipSetDevice(x->device));
    const uint32_t slot = (uint32_t)(x->t % kRing), half = slot / kHalf;
    psim_attractor *list = x->ring + (size_t)slot * (size_t)x->world;
    // With a timeout the host never runs more than a ring ahead of the collectives: the gather
    // that used this list 16 steps ago must have finished, or the exchange is given up here --
    // which is how a device-paced run notices that a rank has gone.
    if (x->timeout_ms && x->t >= kRing) {
        const int rc = wait_event(x, x->gathered[slot], "an all-gather enqueued 16 steps ago");
        if (rc != PSIM_OK) return rc;
    }
    if (slot % kHalf == 0 && x->t >= kRing)          // the kernels that read this half are done
        HIP_TRY(hipStreamWaitEvent(x->comm_stream, x->kernel_done[half], 0));
    int rc = psim_store_attractor(list + x->rank, mine, x->comm_stream);
    if (rc != PSIM_OK) return rc;
    RCCL_TRY(r, r->AllGather(list + x->rank, list, kRecordWords, ncclInt32, x->comm,
                             x->comm_stream));       // in place: own record already sits at [rank]
    HIP_TRY(hipEventRecord(x->gathered[slot], x->comm_stream));
    if (x->pacing == PSIM_EXCHANGE_PIPELINED) {
        // This step's gather is in its queue; only now the PREVIOUS step's kernel goes in, behind
        // a gather that has had a whole kernel to run under.  This step's kernel waits for the
        // next call (or psim_exchange_sync).
        rc = flush_pending(x);
        if (rc != PSIM_OK) return rc;
        x->pending = true;
        x->pending_params = *params;
        x->pending_slot = slot;
    } else {
        // Host-paced: the HOST waits for the gather (which runs under the previous step's
        // kernel), so the kernel goes into its queue with nothing in front of it but the
        // previous kernel.  Device-paced: the handle's stream waits.
        rc = enqueue_kernel(x, params, slot, x->pacing == PSIM_EXCHANGE_HOST_PACED);
        if (rc != PSIM_OK) return rc;
    }
    ++x->t;
    return PSIM_OK;
}

}  // namespace

PSIM_API int psim_exchange_step(psim_exchange *x, const psim_sim_params *params,
                                const psim_attractor *mine)
{
    if (!x) return psim_fail(PSIM_ERR_INVALID, "null exchange");
    if (!params || !mine) return psim_fail(PSIM_ERR_INVALID, "null argument");
    if (x->broken) return give_up(x, x->why_broken);
    const int rc = step_impl(x, params, mine);
    // A step that failed half-way leaves this rank out of step with the others (its collective
    // may or may not have been enqueued): whatever the reason, the exchange is finished.
    if (rc != PSIM_OK && !x->broken) {
        char why[200];
        snprintf(why, sizeof(why), "a step failed (%.150s)", psim_last_error());
        (void)give_up(x, why);                       // the error text now names both
    }
    return rc;
}

PSIM_API int psim_exchange_set_pacing(psim_exchange *x, int pacing)
{
    if (!x) return psim_fail(PSIM_ERR_INVALID, "null exchange");
    if (pacing != PSIM_EXCHANGE_DEVICE_PACED && pacing != PSIM_EXCHANGE_HOST_PACED &&
        pacing != PSIM_EXCHANGE_PIPELINED)
        return psim_fail(PSIM_ERR_INVALID, "unknown pacing %d", pacing);
    if (x->broken) return give_up(x, x->why_broken);
    if (x->pending && pacing != PSIM_EXCHANGE_PIPELINED) {   // nothing stays behind in the old mode
        HIP_TRY(hipSetDevice(x->device));
        const int rc = flush_pending(x);
        if (rc != PSIM_OK) {
            char why[200];
            snprintf(why, sizeof(why), "a step failed (%.150s)", psim_last_error());
            if (!x->broken) (void)give_up(x, why);
            return rc;
        }
    }
    x->pacing = pacing;
    return PSIM_OK;
}

PSIM_API int psim_exchange_set_timeout(psim_exchange *x, uint32_t timeout_ms)
{
    if (!x) return psim_fail(PSIM_ERR_INVALID, "null exchange");
    x->timeout_ms = timeout_ms;
    return PSIM_OK;
}

PSIM_API int psim_exchange_sync(psim_exchange *x)
{
    if (!x) return psim_fail(PSIM_ERR_INVALID, "null exchange");
    HIP_TRY(hipSetDevice(x->device));
    if (x->broken) return give_up(x, x->why_broken);
    int rc = flush_pending(x);                       // pipelined pacing: the last step's kernel
    if (rc != PSIM_OK) {
        char why[200];
        snprintf(why, sizeof(why), "a step failed (%.150s)", psim_last_error());
        if (!x->broken) (void)give_up(x, why);
        return rc;
    }
    rc = wait_stream(x, x->comm_stream, "the communication stream");
    if (rc != PSIM_OK) return rc;
    if (x->timeout_ms == 0) return psim_sync(x->h);
    return wait_stream(x, static_cast<hipStream_t>(psim_get_stream(x->h)), "the handle's stream");
}

PSIM_API int psim_exchange_last_list(psim_exchange *x, psim_attractor *out_host)
{
    if (!x || !out_host) return psim_fail(PSIM_ERR_INVALID, "null argument");
    if (x->t == 0) return psim_fail(PSIM_ERR_INVALID, "no step has been taken");
    const int rc = psim_exchange_sync(x);
    if (rc != PSIM_OK) return rc;
    HIP_TRY(hipMemcpy(out_host, x->ring + (size_t)((x->t - 1) % kRing) * (size_t)x->world,
                      (size_t)x->world * sizeof(psim_attractor), hipMemcpyDeviceToHost));
    return PSIM_OK;
}

PSIM_API int psim_exchange_world(const psim_exchange *x) { return x ? x->world : 0; }
PSIM_API int psim_exchange_rank(const psim_exchange *x) { return x ? x->rank : -1; }
