// exp_collective_overlap.hip -- can a collective's kernel run UNDER the update kernel?
//
// The exchange enqueues the all-gather of step t + 1 on its own stream while the update kernel
// of step t runs, so that the gather's latency hides under the kernel.  That only works if the
// collective's workgroup finds room on the chip.  RCCL's device kernel on gfx950
// (ncclDevKernel_Generic_*, read from the code object in librccl.so.1) asks for 37 664 B of
// LDS, 248-256 vector registers per lane and up to 512 threads per workgroup -- two waves per
// SIMD at that register count: nearly a whole CU -- while the update kernel keeps every CU full
// of single-wave workgroups that are replaced the moment one ends.  One GPU cannot run a real
// multi-rank collective, so this experiment launches a stand-in with the SAME resource
// footprint (`fat_kernel`) on a second stream right when an update kernel starts and stamps
// the chip-wide constant clock: does the stand-in start at once, or only when the update
// kernel's grid runs dry?  And what does it cost the update kernel to leave some CUs free (a
// stream with a CU mask), which gives the stand-in a place to run?
//
// What it showed (profiles/tuning/r03_collective_overlap.log): without free CUs the stand-in
// starts when the update kernel ends, whatever its stream's priority -- or at the boundary
// between two launches when the step is issued in two halves; with 8-32 CUs masked out of the
// update kernel's stream it runs under the kernel, and the 6.25 M-particle step does not slow
// down (the 100 M-particle step would, by 2.7 %).  So a collective must already be waiting when
// a kernel's grid runs dry: the exchange's device pacing, where the host runs ahead, does that.  The library does NOT mask its streams
// all the same: of six short programs that carried large device-to-host copies on a CU-masked
// stream and then destroyed it, three left the runtime hanging in a later hipStreamDestroy or at
// process exit (none of seven that ran only kernels and small copies on it did), and whether
// the real collective behaves like the stand-in needs several GPUs to see.
// A caller who wants to try hands the handle a masked stream of its own (psim_set_stream).
//
// build (from the repository root):
//   hipcc --offload-arch=gfx950 -O2 -Iinclude -o scripts/exp_collective_overlap \
//       scripts/exp_collective_overlap.hip -Lparticle-simulation-3d_amd/csrc -lpsim_hip \
//       -Wl,-rpath,'$ORIGIN/../particle-simulation-3d_amd/csrc'
// run: scripts/exp_collective_overlap
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <set>
#include <vector>

#include "psim.h"

#define TRY(expr)                                                                   \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s (line %d)\n", #expr, hipGetErrorString(e_), __LINE__); \
            return 1;                                                               \
        }                                                                           \
    } while (0)
#define PSIM_TRY(expr)                                                              \
    do {                                                                            \
        if ((expr) != PSIM_OK) {                                                    \
            fprintf(stderr, "%s: %s (line %d)\n", #expr, psim_last_error(), __LINE__); \
            return 1;                                                               \
        }                                                                           \
    } while (0)

// One stamp of the constant 100 MHz clock all CUs share.
__global__ void stamp_kernel(uint64_t *dst)
{
    if (threadIdx.x == 0) *dst = wall_clock64();
}

// A workgroup shaped like RCCL's: 512 threads, 37 664 B of LDS, 248 vector registers per lane
// (the clobber makes the compiler allocate that many).  It stamps the clock when it starts,
// stays for `ticks` of the clock (a small collective: ~10 us) and stamps again.
constexpr int kFatLds = 37664;
__global__ __launch_bounds__(512) void fat_kernel(uint64_t *start_end, uint32_t ticks)
{
    __shared__ char lds[kFatLds];
    const uint64_t t0 = wall_clock64();
    lds[threadIdx.x] = (char)threadIdx.x;
    asm volatile("" ::: "v247");
    __syncthreads();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        start_end[2 * blockIdx.x] = t0;
        start_end[2 * blockIdx.x + 1] = wall_clock64() + (uint64_t)(lds[7] == 99);
    }
}

// Where a workgroup runs: XCD and the CU inside it (hardware id registers).
__global__ void where_kernel(uint32_t *out, uint32_t ticks)
{
    uint32_t hw_id, xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) out[blockIdx.x] = ((xcc_id & 0xFu) << 8) | ((hw_id >> 8) & 0xFFu);
}

static double median(std::vector<double> v)
{
    std::sort(v.begin(), v.end());
    return v.empty() ? 0.0 : v[v.size() / 2];
}

struct Shape {
    const char *name;
    uint32_t particles;
    uint32_t attractors;   // 1: psim_update; more: psim_update_multi_host
};

int main()
{
    hipDeviceProp_t prop;
    TRY(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs\n", prop.gcnArchName, cus);
    hipFuncAttributes fa;
    TRY(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(fat_kernel)));
    printf("stand-in for the collective: %d vector registers, %zu B LDS, 512 threads per workgroup\n",
           fa.numRegs, fa.sharedSizeBytes);

    uint64_t *stamps = nullptr;            // page-locked: the device writes, the host reads
    TRY(hipHostMalloc(reinterpret_cast<void **>(&stamps), 64 * sizeof(uint64_t), hipHostMallocDefault));
    uint32_t *where = nullptr;
    const uint32_t where_blocks = 16384;
    TRY(hipHostMalloc(reinterpret_cast<void **>(&where), where_blocks * sizeof(uint32_t), hipHostMallocDefault));
    // The collective's stream, at normal or at the highest priority.  Never more than three
    // streams live at once (this one, the handle's own, the one under test): the runtime spreads
    // streams over four hardware queues (GPU_MAX_HW_QUEUES), and two streams that share a queue
    // run one after the other whatever the CUs could hold.
    int least = 0, greatest = 0;
    TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t side = nullptr;
    hipEvent_t started, t_begin, t_end;
    TRY(hipEventCreateWithFlags(&started, hipEventDisableTiming));
    TRY(hipEventCreate(&t_begin));
    TRY(hipEventCreate(&t_end));

    psim_sim_params params;
    memset(&params, 0, sizeof(params));
    params.delta_time = 0.016f, params.damping = 0.99f, params.mouse_force = 5.0f;
    params.mouse_radius = 10.0f, params.is_mouse_dragging = 1, params.max_dist_for_color = 50.0f;
    params.mouse_position[2] = 48.0f;
    psim_attractor list[8];
    memset(list, 0, sizeof(list));
    for (int a = 0; a < 8; ++a) {
        list[a].position[0] = -35.0f + 10.0f * a, list[a].position[1] = 5.0f * a - 20.0f;
        list[a].position[2] = 30.0f - 7.0f * a;
        list[a].force = 5.0f, list[a].radius = 10.0f, list[a].active = 1;
    }

    const Shape shapes[] = {{"6.25 M particles, 8 attractors (one GPU's share of the 50 M configuration)", 6250000u, 8u},
                            {"100 M particles, single attractor (the headline; no collective runs beside it)", 100000000u, 1u}};
    // -1: no mask, the stand-in's stream at the highest priority; -3 / -2: no mask, the step as
    // one launch / as two launches over the halves of the arrays (the stateless single-attractor
    // entry point on the handle's arrays): does the boundary between two launches, where the
    // first one's grid runs dry, give the stand-in its place?
    const int free_cus[] = {0, -1, -3, -2, 8, 16, 32};
    for (const Shape &shape : shapes) {
        printf("\n%s\n", shape.name);
        printf("%-28s %12s %14s %16s %16s %14s\n", "update kernel's stream", "us per step",
               "CUs it ran on", "stand-in starts", "stand-in ends", "kernel ends");
        psim_handle *h = nullptr;
        PSIM_TRY(psim_create(&h, shape.particles, PSIM_GEN_UNIFORM, 0));
        for (int leave : free_cus) {
            hipStream_t stream = nullptr;
            if (side) TRY(hipStreamDestroy(side));
            if (leave == -1) TRY(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, greatest));
            else TRY(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
            if (leave <= 0) {
                TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
            } else {
                // the highest `leave` bits of the mask cleared
                uint32_t mask[8];
                for (int w = 0; w < 8; ++w) mask[w] = 0xFFFFFFFFu;
                for (int b = cus - leave; b < cus; ++b) mask[b / 32] &= ~(1u << (b % 32));
                TRY(hipExtStreamCreateWithCUMask(&stream, (uint32_t)((cus + 31) / 32), mask));
            }
            PSIM_TRY(psim_set_stream(h, stream));
            psim_soa_view v;
            PSIM_TRY(psim_device_soa(h, &v));
            const uint32_t half = (shape.particles / 2u) & ~1023u;
            const auto update = [&]() {
                if (leave == -3)
                    return psim_update_soa(v.px, v.py, v.pz, v.vx, v.vy, v.vz, nullptr, shape.particles,
                                           &params, stream);
                if (leave == -2) {
                    const int rc = psim_update_soa(v.px, v.py, v.pz, v.vx, v.vy, v.vz, nullptr, half,
                                                   &params, stream);
                    if (rc != PSIM_OK) return rc;
                    return psim_update_soa(v.px + half, v.py + half, v.pz + half, v.vx + half,
                                           v.vy + half, v.vz + half, nullptr, shape.particles - half,
                                           &params, stream);
                }
                return shape.attractors == 1 ? psim_update(h, &params)
                                             : psim_update_multi_host(h, &params, list, shape.attractors);
            };
            // where the stream's workgroups land
            hipLaunchKernelGGL(where_kernel, dim3(where_blocks), dim3(64), 0, stream, where, 300u);
            TRY(hipStreamSynchronize(stream));
            std::set<uint32_t> places;
            int per_xcd[16] = {0};
            for (uint32_t b = 0; b < where_blocks; ++b)
                if (places.insert(where[b]).second) per_xcd[(where[b] >> 8) & 15]++;
            // the cost of the mask: steps back to back
            const int steps = shape.particles > 50000000u ? 60 : 400;
            for (int k = 0; k < 20; ++k) PSIM_TRY(update());
            TRY(hipEventRecord(t_begin, stream));
            for (int k = 0; k < steps; ++k) PSIM_TRY(update());
            TRY(hipEventRecord(t_end, stream));
            TRY(hipEventSynchronize(t_end));
            float ms = 0.0f;
            TRY(hipEventElapsedTime(&ms, t_begin, t_end));
            // the overlap: the stand-in becomes eligible when the update kernel starts
            std::vector<double> begins, ends, kernel;
            for (int trial = 0; trial < 40; ++trial) {
                hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, stream, stamps + 0);
                TRY(hipEventRecord(started, stream));
                PSIM_TRY(update());
                hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, stream, stamps + 1);
                TRY(hipStreamWaitEvent(side, started, 0));
                hipLaunchKernelGGL(fat_kernel, dim3(1), dim3(512), 0, side, stamps + 2, 1000u);
                TRY(hipStreamSynchronize(side));
                TRY(hipStreamSynchronize(stream));
                begins.push_back((double)(stamps[2] - stamps[0]) / 100.0);   // 100 MHz -> us
                ends.push_back((double)(stamps[3] - stamps[0]) / 100.0);
                kernel.push_back((double)(stamps[1] - stamps[0]) / 100.0);
            }
            char label[64], xcds[128] = "";
            if (leave == -3) snprintf(label, sizeof(label), "no mask, 1 attr., 1 launch");
            else if (leave == -2) snprintf(label, sizeof(label), "no mask, 1 attr., 2 launches");
            else if (leave < 0) snprintf(label, sizeof(label), "no mask, stand-in high prio");
            else snprintf(label, sizeof(label), leave ? "mask leaves %d CUs free" : "no mask", leave);
            for (int x = 0; x < 8; ++x) snprintf(xcds + strlen(xcds), sizeof(xcds) - strlen(xcds), "%s%d", x ? "+" : "", per_xcd[x]);
            printf("%-28s %12.2f %14s %13.1f us %13.1f us %11.1f us\n", label, 1e3 * ms / steps, xcds,
                   median(begins), median(ends), median(kernel));
            PSIM_TRY(psim_set_stream(h, nullptr));
            TRY(hipStreamDestroy(stream));
        }
        psim_destroy(h);
    }
    printf("\n(times after the stamp in front of the update kernel, medians of 40 trials; the stand-in "
           "stays 10 us; CUs per XCD the stream's workgroups were seen on)\n");
    return 0;
}
