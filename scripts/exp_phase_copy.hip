// exp_phase_copy.hip -- does separating reads from writes IN TIME buy HBM bandwidth?
//
// On the MI355X a read-only sweep runs at 7.1 TB/s, a write-only one at 7.1 TB/s, and any kernel
// that reads and writes in equal parts -- the particle update -- at 6.7 TB/s.  If the difference
// is what the memory pays for turning its buses round, a chip that reads for a few microseconds
// and then writes for a few microseconds should get some of it back.  This experiment scales an
// array in place with persistent waves: each wave loads R float4 registers per lane (R KiB per
// wave), scales, stores, and moves on.  In the phased form every wave issues its loads only in
// even windows and its stores only in odd windows of the chip-wide constant clock
// (s_memrealtime, 100 MHz); the free-running form is the same kernel without the waits.
// Every wait is bounded, so the grid drains whatever the clock does.
//
// build: hipcc -O3 --offload-arch=gfx950 -o exp_phase_copy exp_phase_copy.hip
// run:   ./exp_phase_copy [MiB]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                       \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Waits until the chip-wide clock is in a window of the wanted parity (at most ~`limit` polls).
__device__ __forceinline__ void wait_window(uint32_t window_ticks, uint32_t parity, uint32_t shift)
{
    for (int spin = 0; spin < 20000; ++spin) {
        const uint32_t now = (uint32_t)__builtin_amdgcn_s_memrealtime() + shift;
        if (((now / window_ticks) & 1u) == parity) return;
        __builtin_amdgcn_s_sleep(2);
    }
}

template <int R, bool PHASED>
__global__ void __launch_bounds__(64) phase_copy(f32x4 *data, uint64_t nquads, uint32_t window_ticks,
                                                 uint32_t store_shift)
{
    extern __shared__ char reserve[];                         // residency cap only
    (void)reserve;
    const uint32_t lane = threadIdx.x;
    const uint64_t per_round = (uint64_t)R * 64u;             // quads a wave moves per round
    const uint64_t stride = (uint64_t)gridDim.x * per_round;
    for (uint64_t base = (uint64_t)blockIdx.x * per_round; base + per_round <= nquads;
         base += stride) {
        f32x4 v[R];
        if (PHASED) wait_window(window_ticks, 0u, 0u);
#pragma unroll
        for (int j = 0; j < R; ++j)
            v[j] = __builtin_nontemporal_load(&data[base + (uint64_t)j * 64u + lane]);
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = v[j] * 1.0000001f;
        if (PHASED) wait_window(window_ticks, 1u, store_shift);
#pragma unroll
        for (int j = 0; j < R; ++j)
            __builtin_nontemporal_store(v[j], &data[base + (uint64_t)j * 64u + lane]);
    }
}

template <int R, bool PHASED>
float run(f32x4 *data, uint64_t nquads, int cus, int waves_per_cu, uint32_t window_ticks,
          uint32_t store_shift, int reps)
{
    const size_t lds = (size_t)(160 * 1024) / (size_t)waves_per_cu - 512;    // caps residency
    auto kernel = phase_copy<R, PHASED>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 grid((unsigned)(cus * waves_per_cu));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kernel, grid, dim3(64), lds, 0, data, nquads, window_ticks, store_shift);
    CHECK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(kernel, grid, dim3(64), lds, 0, data, nquads, window_ticks, store_shift);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return ms / (float)reps;
}

template <int R>
void sweep(f32x4 *data, uint64_t nquads, int cus)
{
    const double bytes = 2.0 * 16.0 * (double)(nquads / ((uint64_t)R * 64u) * ((uint64_t)R * 64u));
    for (int wpc : {4, 6, 8}) {
        if (((wpc + 3) / 4) * (4 * R + 16) > 512) continue;  // registers: R quads of data per lane
        const float free_ms = run<R, false>(data, nquads, cus, wpc, 0u, 0u, 5);
        printf("R=%3d (%3d KiB/wave) %d waves/CU  free-running          : %.4f ms  %7.1f GB/s\n", R, R,
               wpc, free_ms, bytes / free_ms / 1e6);
        // a window holds what the resident waves move at ~7 TB/s
        const double window_us = (double)cus * wpc * R * 1024.0 / 7.0e6;
        for (double scale : {0.75, 1.0, 1.25, 1.5}) {
            const uint32_t ticks = (uint32_t)(window_us * scale * 100.0 + 0.5);
            if (ticks < 20) continue;
            for (uint32_t shift : {0u, ticks / 8u, ticks / 4u}) {
                const float ms = run<R, true>(data, nquads, cus, wpc, ticks, shift, 5);
                printf("R=%3d (%3d KiB/wave) %d waves/CU  window %5.2f us early %4.2f us : %.4f ms  %7.1f GB/s\n",
                       R, R, wpc, ticks / 100.0, shift / 100.0, ms, bytes / ms / 1e6);
            }
        }
        fflush(stdout);
    }
}

int main(int argc, char **argv)
{
    const size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 2400;
    const uint64_t nquads = mib * 1024ull * 1024ull / 16ull;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    f32x4 *data;
    CHECK(hipMalloc(&data, nquads * 16ull));
    CHECK(hipMemset(data, 0x3f, nquads * 16ull));
    printf("in-place scale of %zu MiB, %d CUs, nt loads and stores, persistent waves\n", mib, cus);
    sweep<16>(data, nquads, cus);
    sweep<32>(data, nquads, cus);
    sweep<64>(data, nquads, cus);
    CHECK(hipFree(data));
    return 0;
}
