// fake_rccl.hip -- a TEST DOUBLE for librccl.so.1, for one purpose: RCCL refuses two ranks on
// one GPU, so a one-GPU machine cannot run the attractor exchange (psim_exchange_*) with a
// world larger than one.  This library implements the seven RCCL entry points the exchange
// binds, with the all-gather going through POSIX shared memory between the rank processes, so
// that the exchange's own logic for world > 1 (each rank's place in the receive lists, the ring
// of lists, the waits between the streams, giving up when a rank has gone) can be exercised by
// several processes sharing GPU 0.
//
// Like the real one, the all-gather is ASYNCHRONOUS, stream-ordered and a KERNEL: the call
// launches one single-wave kernel on the caller's stream and returns.  The kernel writes this
// rank's bytes into a page-locked mailbox, raises a flag, and waits -- as RCCL's kernels wait
// for their peers -- until a helper thread of this library has carried the bytes through
// shared memory to the other rank processes and brought theirs back; then it copies all ranks'
// bytes into the receive buffer.  So a missing wait between the exchange's streams shows as
// wrong data here as it would with RCCL, and a rank that has died leaves the others' streams
// stuck in that kernel until ncclCommAbort releases them.  The kernel's wait is bounded (30 s
// by the device's constant clock, and an iteration cap): every wave always reaches its end.
//
// It is selected only through PSIM_RCCL_LIBRARY by the tests; nothing in the library or the
// benchmark ever loads it.
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <new>
#include <thread>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

namespace {

constexpr int kSlots = 64, kMaxRanks = 64, kMaxWords = 16;   // per rank and gather
constexpr int kJoinSeconds = 90;        // ranks start at different times
constexpr int kSafetySeconds = 25;      // nobody aborted a collective a dead rank left stuck
constexpr uint64_t kDeviceTicks = 30ull * 100000000ull;      // 30 s of the 100 MHz constant clock
constexpr uint32_t kDeviceSpins = 40000000u;

// Between the rank processes.
struct Shared {
    std::atomic<uint32_t> joined;
    std::atomic<uint32_t> arrived[kSlots];                   // running count per slot
    uint32_t data[kSlots][kMaxRanks][kMaxWords];
};

// Between one rank's kernels and its helper thread: page-locked, coherent host memory.
struct Mailbox {
    uint32_t sent[kSlots];       // kernel -> helper: the collective's ticket once `out` is filled
    uint32_t done[kSlots];       // helper -> kernel: the ticket once `in` holds every rank's words
    uint32_t abort;              // host -> kernels: stop waiting
    uint32_t gave_up;            // kernels -> host: a wait ran into its bound
    uint32_t out[kSlots][kMaxWords];
    uint32_t in[kSlots][kMaxRanks * kMaxWords];
};

struct Comm {
    Shared *shared = nullptr;
    Mailbox *box = nullptr;
    char name[160] = "";
    int rank = 0, world = 1;
    uint32_t words = 0;                                      // per rank; the same in every call
    uint64_t seq = 0;
    std::atomic<int> stop{0};
    std::atomic<int> async_error{(int)ncclSuccess};
    std::thread helper;
};

double seconds_now()
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + t.tv_nsec * 1e-9;
}

// PSIM_FAKE_RCCL_TRACE=1: one line on stderr per stage of every collective.
bool tracing()
{
    static const bool on = [] {
        const char *e = getenv("PSIM_FAKE_RCCL_TRACE");
        return e && *e == '1';
    }();
    return on;
}

#define TRACE(...)                                             \
    do {                                                       \
        if (tracing()) {                                       \
            fprintf(stderr, "[fake rccl %d %.6f] ", (int)getpid(), seconds_now()); \
            fprintf(stderr, __VA_ARGS__);                      \
            fputc('\n', stderr);                               \
        }                                                      \
    } while (0)

void nap()
{
    const struct timespec t = {0, 20000};
    nanosleep(&t, nullptr);
}

bool wait_for(const std::atomic<uint32_t> &v, uint32_t at_least, int seconds,
              const std::atomic<int> *stop)
{
    const double start = seconds_now();
    for (uint32_t spins = 0; v.load(std::memory_order_acquire) < at_least; ++spins) {
        if (stop && stop->load(std::memory_order_acquire)) return false;
        if (seconds_now() - start > seconds) return false;
        if (spins > 200) nap();
    }
    return true;
}

// How long the helper thread spins for its kernel before it starts napping (a nap costs tens of
// microseconds of latency: fine for the tests, too slow for timing rehearsals, which raise it
// with PSIM_FAKE_RCCL_SPIN).
uint32_t patience()
{
    static const uint32_t spins = [] {
        const char *e = getenv("PSIM_FAKE_RCCL_SPIN");
        return e && *e ? (uint32_t)strtoul(e, nullptr, 10) : 2000u;
    }();
    return spins;
}

uint32_t load_acquire(const uint32_t *p)
{
    return __atomic_load_n(p, __ATOMIC_ACQUIRE);
}

// The helper thread: takes the collectives in order.  For each, waits until this rank's kernel
// has filled the mailbox, publishes the words in shared memory, waits until every rank has
// published its own for the same collective, collects them and releases the kernel.
void carry(Comm *c)
{
    Shared *sh = c->shared;
    Mailbox *mb = c->box;
    for (uint64_t seq = 0;; ++seq) {
        const int slot = (int)(seq % kSlots);
        const uint32_t ticket = (uint32_t)(seq + 1);
        for (uint32_t spins = 0; load_acquire(&mb->sent[slot]) != ticket; ++spins) {
            if (c->stop.load(std::memory_order_acquire)) return;
            if (spins > patience()) nap();
        }
        const uint32_t words = c->words;
        TRACE("rank %d: collective %llu has left the device", c->rank, (unsigned long long)seq);
        memcpy(sh->data[slot][c->rank], mb->out[slot], words * 4);
        sh->arrived[slot].fetch_add(1, std::memory_order_acq_rel);
        // the counter keeps running: this slot's (seq / kSlots + 1)-th use is complete at world x that
        const uint32_t all = (uint32_t)((seq / kSlots + 1) * (uint64_t)c->world);
        if (!wait_for(sh->arrived[slot], all, kSafetySeconds, &c->stop)) {
            c->async_error.store((int)ncclSystemError, std::memory_order_release);
            TRACE("rank %d: collective %llu is incomplete", c->rank, (unsigned long long)seq);
        }
        for (int r = 0; r < c->world; ++r)
            memcpy(mb->in[slot] + (size_t)r * words, sh->data[slot][r], words * 4);
        __atomic_store_n(&mb->done[slot], ticket, __ATOMIC_RELEASE);
        TRACE("rank %d: collective %llu goes back to the device", c->rank, (unsigned long long)seq);
    }
}

// One wave.  send may alias recv + rank * words (in place): it is read before recv is written.
__global__ void __launch_bounds__(64) gather_kernel(const uint32_t *send, uint32_t *recv,
                                                    Mailbox *mb, int slot, uint32_t ticket,
                                                    uint32_t words, uint32_t world)
{
    const uint32_t lane = threadIdx.x;
    if (lane < words) mb->out[slot][lane] = send[lane];
    __threadfence_system();
    __shared__ uint32_t released;
    if (lane == 0) {
        __hip_atomic_store(&mb->sent[slot], ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint64_t start = wall_clock64();
        uint32_t ok = 0;
        for (uint32_t spins = 0; spins < kDeviceSpins; ++spins) {
            if (__hip_atomic_load(&mb->done[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == ticket) {
                ok = 1;
                break;
            }
            if (__hip_atomic_load(&mb->abort, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
            if (wall_clock64() - start > kDeviceTicks) break;
            __builtin_amdgcn_s_sleep(64);
        }
        if (!ok) __hip_atomic_store(&mb->gave_up, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        released = ok;
    }
    __syncthreads();
    __threadfence_system();
    if (released)
        for (uint32_t i = lane; i < world * words; i += 64) recv[i] = mb->in[slot][i];
}

// The same collective with the resource footprint of RCCL's own device kernel on gfx950
// (ncclDevKernel_Generic_*: 37 664 B of LDS, 248 vector registers per lane, 512 threads), for
// rehearsing WHEN such a workgroup finds room beside the update kernel.  Wave 0 does the work
// of gather_kernel, the other seven only take their place.  PSIM_FAKE_RCCL_FAT=1 selects it.
__global__ void __launch_bounds__(512) gather_kernel_fat(const uint32_t *send, uint32_t *recv,
                                                         Mailbox *mb, int slot, uint32_t ticket,
                                                         uint32_t words, uint32_t world)
{
    __shared__ char ballast[37664 - 16];
    __shared__ uint32_t released;
    const uint32_t lane = threadIdx.x;
    ballast[lane] = (char)lane;
    asm volatile("" ::: "v247");
    if (lane < words) mb->out[slot][lane] = send[lane];
    __threadfence_system();
    if (lane == 0) {
        __hip_atomic_store(&mb->sent[slot], ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint64_t start = wall_clock64();
        uint32_t ok = 0;
        for (uint32_t spins = 0; spins < kDeviceSpins; ++spins) {
            if (__hip_atomic_load(&mb->done[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == ticket) {
                ok = 1;
                break;
            }
            if (__hip_atomic_load(&mb->abort, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
            if (wall_clock64() - start > kDeviceTicks) break;
            __builtin_amdgcn_s_sleep(64);
        }
        if (!ok) __hip_atomic_store(&mb->gave_up, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        released = ok + (uint32_t)(ballast[5] == 77);
    }
    __syncthreads();
    __threadfence_system();
    if (released && lane < 64)
        for (uint32_t i = lane; i < world * words; i += 64) recv[i] = mb->in[slot][i];
}

bool fat()
{
    static const bool on = [] {
        const char *e = getenv("PSIM_FAKE_RCCL_FAT");
        return e && *e == '1';
    }();
    return on;
}

}  // namespace

extern "C" {

__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id->internal, 0, sizeof(id->internal));
    const int fd = open("/dev/urandom", O_RDONLY);
    uint64_t token[2] = {0, 0};
    if (fd < 0 || read(fd, token, sizeof(token)) != (ssize_t)sizeof(token)) return ncclSystemError;
    close(fd);
    snprintf(id->internal, sizeof(id->internal), "/psim_fake_rccl_%016llx%016llx",
             (unsigned long long)token[0], (unsigned long long)token[1]);
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks,
                                                                     ncclUniqueId id, int rank)
{
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm *c = new (std::nothrow) Comm();
    if (!c) return ncclSystemError;
    c->rank = rank, c->world = nranks;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    if (hipHostMalloc(reinterpret_cast<void **>(&c->box), sizeof(Mailbox),
                      hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
        delete c;
        return ncclUnhandledCudaError;
    }
    memset(c->box, 0, sizeof(Mailbox));
    const int fd = shm_open(c->name, O_RDWR | O_CREAT, 0600);   // fresh shared memory is zero
    if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) {
        delete c;
        return ncclSystemError;
    }
    void *p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        delete c;
        return ncclSystemError;
    }
    c->shared = static_cast<Shared *>(p);
    // The kernel's code is loaded, and has run once, before any rank can be waiting in it.
    hipLaunchKernelGGL(gather_kernel, dim3(1), dim3(64), 0, nullptr, nullptr, nullptr, c->box, 0, 0u, 0u, 0u);
    if (fat())
        hipLaunchKernelGGL(gather_kernel_fat, dim3(1), dim3(512), 0, nullptr, nullptr, nullptr, c->box, 0, 0u, 0u, 0u);
    if (hipDeviceSynchronize() != hipSuccess) return ncclUnhandledCudaError;
    TRACE("rank %d of %d joins %s", rank, nranks, c->name);
    c->shared->joined.fetch_add(1, std::memory_order_acq_rel);
    if (!wait_for(c->shared->joined, (uint32_t)nranks, kJoinSeconds, nullptr)) return ncclSystemError;
    try {
        c->helper = std::thread(carry, c);
    } catch (...) {
        return ncclSystemError;
    }
    *out = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

// The caller has drained its streams (the exchange waits for the device first).
__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return ncclSuccess;
    c->stop.store(1, std::memory_order_release);
    if (c->helper.joinable()) c->helper.join();
    munmap(c->shared, sizeof(Shared));
    if (c->rank == 0) shm_unlink(c->name);
    (void)hipHostFree(c->box);
    delete c;
    return ncclSuccess;
}

// Releases the kernels that wait for ranks which will never come, and the helper thread.
// Kernels may still be enqueued on the caller's streams, so nothing they touch is freed: a
// test double may leak.
__attribute__((visibility("default"))) ncclResult_t ncclCommAbort(ncclComm_t comm)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return ncclSuccess;
    TRACE("rank %d: abort", c->rank);
    __atomic_store_n(&c->box->abort, 1u, __ATOMIC_RELEASE);
    c->stop.store(1, std::memory_order_release);
    if (c->helper.joinable()) c->helper.join();
    if (c->rank == 0) shm_unlink(c->name);
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommGetAsyncError(ncclComm_t comm,
                                                                          ncclResult_t *state)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || !state) return ncclInvalidArgument;
    *state = (ncclResult_t)c->async_error.load(std::memory_order_acquire);
    if (*state == ncclSuccess && load_acquire(&c->box->gave_up)) *state = ncclSystemError;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclAllGather(const void *send, void *recv,
                                                                  size_t count, ncclDataType_t type,
                                                                  ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (type != ncclInt32 && type != ncclFloat32 && type != ncclUint32) return ncclInvalidArgument;
    if (!c || count == 0 || count > (size_t)kMaxWords) return ncclInvalidArgument;
    if (c->words == 0) c->words = (uint32_t)count;
    if (c->words != (uint32_t)count) return ncclInvalidArgument;   // one shape per communicator
    const uint64_t seq = c->seq++;
    TRACE("rank %d: collective %llu enqueued", c->rank, (unsigned long long)seq);
    if (fat())
        hipLaunchKernelGGL(gather_kernel_fat, dim3(1), dim3(512), 0, stream,
                           static_cast<const uint32_t *>(send), static_cast<uint32_t *>(recv), c->box,
                           (int)(seq % kSlots), (uint32_t)(seq + 1), (uint32_t)count, (uint32_t)c->world);
    else
        hipLaunchKernelGGL(gather_kernel, dim3(1), dim3(64), 0, stream,
                           static_cast<const uint32_t *>(send), static_cast<uint32_t *>(recv), c->box,
                           (int)(seq % kSlots), (uint32_t)(seq + 1), (uint32_t)count, (uint32_t)c->world);
    return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

__attribute__((visibility("default"))) const char *ncclGetErrorString(ncclResult_t r)
{
    return r == ncclSuccess ? "no error" : "fake RCCL (test double) reported a failure";
}

}  // extern "C"
