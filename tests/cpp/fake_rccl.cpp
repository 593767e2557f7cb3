// fake_rccl.cpp -- a TEST DOUBLE for librccl.so.1, for one purpose: RCCL refuses two ranks on
// one GPU, so a one-GPU machine cannot run the attractor exchange (psim_exchange_*) with a
// world larger than one.  This library implements the five RCCL entry points the exchange
// uses -- the all-gather through POSIX shared memory between the rank processes, made
// stream-ordered by synchronising the stream -- so that the exchange's own logic for
// world > 1 (each rank's place in the receive lists, the ring of lists, the list length the
// kernel is given) can be exercised by several processes sharing GPU 0.  It is slow, it is
// selected only through PSIM_RCCL_LIBRARY by tests/test_gpu_exchange.py, and nothing in the
// library or the benchmark ever loads it.
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <new>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

namespace {

constexpr int kSlots = 64, kMaxRanks = 64, kMaxBytes = 64;   // per rank and gather

struct Shared {
    std::atomic<uint32_t> joined;
    std::atomic<uint32_t> arrived[kSlots];                   // running count per slot
    unsigned char data[kSlots][kMaxRanks][kMaxBytes];
};

struct Comm {
    Shared *shared = nullptr;
    char name[160] = "";
    int rank = 0, world = 1;
    uint64_t seq = 0;
};

bool wait_for(const std::atomic<uint32_t> &v, uint32_t at_least)
{
    struct timespec start;
    clock_gettime(CLOCK_MONOTONIC, &start);
    while (v.load(std::memory_order_acquire) < at_least) {
        struct timespec now;
        clock_gettime(CLOCK_MONOTONIC, &now);
        if (now.tv_sec - start.tv_sec > 30) return false;    // a rank died: do not hang the test
        const struct timespec nap = {0, 20000};
        nanosleep(&nap, nullptr);
    }
    return true;
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
    c->shared->joined.fetch_add(1, std::memory_order_acq_rel);
    if (!wait_for(c->shared->joined, (uint32_t)nranks)) return ncclSystemError;
    *out = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return ncclSuccess;
    munmap(c->shared, sizeof(Shared));
    if (c->rank == 0) shm_unlink(c->name);
    delete c;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclAllGather(const void *send, void *recv,
                                                                  size_t count, ncclDataType_t type,
                                                                  ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (type != ncclInt32 && type != ncclFloat32 && type != ncclUint32) return ncclInvalidArgument;
    const size_t bytes = count * 4;
    if (!c || bytes > (size_t)kMaxBytes) return ncclInvalidArgument;
    const uint64_t seq = c->seq++;
    const int slot = (int)(seq % kSlots);
    // everything enqueued before the collective has happened; what is enqueued after it comes
    // after these synchronous copies: stream order, the slow way
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->shared->data[slot][c->rank], send, bytes, hipMemcpyDeviceToHost) != hipSuccess)
        return ncclUnhandledCudaError;
    c->shared->arrived[slot].fetch_add(1, std::memory_order_acq_rel);
    // the counter keeps running: this slot's (seq / kSlots + 1)-th use is complete at world x that
    if (!wait_for(c->shared->arrived[slot], (uint32_t)((seq / kSlots + 1) * (uint64_t)c->world)))
        return ncclSystemError;
    for (int r = 0; r < c->world; ++r)
        if (hipMemcpy(static_cast<char *>(recv) + (size_t)r * bytes, c->shared->data[slot][r], bytes,
                      hipMemcpyHostToDevice) != hipSuccess)
            return ncclUnhandledCudaError;
    return ncclSuccess;
}

__attribute__((visibility("default"))) const char *ncclGetErrorString(ncclResult_t r)
{
    return r == ncclSuccess ? "no error" : "fake RCCL (test double) reported a failure";
}

}  // extern "C"
