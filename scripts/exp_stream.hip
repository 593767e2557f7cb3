// exp_stream.hip -- standalone experiment: which access shape moves a 12-stream in-place
// fp32 update (6 arrays read + written) fastest on MI355X?  Not part of the library.
//
// Variants: particle groups per lane (U = 1, 2, 4; a group = 4 consecutive particles =
// one 16-byte access per array), non-temporal or plain accesses, workgroup size, and the
// number of resident waves per CU (capped by reserving LDS).  A plain float4 copy
// (read A, write B) of the same byte count is timed as the reference ceiling.
//
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o exp_stream exp_stream.hip
// run:   ./exp_stream [particles]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <functional>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(e)                                                                      \
    do {                                                                              \
        hipError_t r_ = (e);                                                          \
        if (r_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s (line %d)\n", #e, hipGetErrorString(r_), __LINE__); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

struct Arrays {
    f32x4 *a[6];
};

template <bool NT>
__device__ __forceinline__ f32x4 ld(const f32x4 *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT>
__device__ __forceinline__ void st(f32x4 *p, f32x4 v)
{
#ifdef EXP_NT_SC1
    // build with -DEXP_NT_SC1: non-temporal stores carry nt + sc1 (every kernel here stores
    // last, so the compiler's wait counting is not disturbed by the hand-written store)
    if constexpr (NT) asm volatile("global_store_dwordx4 %0, %1, off nt sc1" ::"v"(p), "v"(v) : "memory");
#else
    if constexpr (NT) __builtin_nontemporal_store(v, p);
#endif
    else *p = v;
}

// the real per-particle arithmetic (single attractor), so VALU work is representative
__device__ __forceinline__ void step1(float &px, float &py, float &pz, float &vx, float &vy,
                                      float &vz)
{
    const float dt = 0.016f, two_r = 20.0f, force = 5.0f, damping = 0.99f;
    const float dx = 0.0f - px, dy = 0.0f - py, dz = 48.0f - pz;
    const float dist = sqrtf((dx * dx + dy * dy) + dz * dz);
    if (dist < two_r) {
        const float t = 1.0f - dist / two_r;
        const float ff = (t * t) * 2.0f;
        const float inv = 1.0f / dist;
        vx = vx + (((dx * inv) * force) * ff) * dt;
        vy = vy + (((dy * inv) * force) * ff) * dt;
        vz = vz + (((dz * inv) * force) * ff) * dt;
    }
    px = px + vx * dt; py = py + vy * dt; pz = pz + vz * dt;
    vx = vx * damping; vy = vy * damping; vz = vz * damping;
}

// One pass: the wave owns U * 64 consecutive groups; lane l takes groups l, l+64, ...
// so every instruction still covers 1 KiB of consecutive addresses.
template <int U, bool NT>
__global__ void update_u(Arrays s, uint32_t ngroups)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t base = wave * (64u * U) + lane;
    f32x4 r[U][6];
#pragma unroll
    for (int k = 0; k < U; ++k) {
        const uint32_t g = base + 64u * k;
        if (g < ngroups) {
#pragma unroll
            for (int c = 0; c < 6; ++c) r[k][c] = ld<NT>(s.a[c] + g);
        }
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
        const uint32_t g = base + 64u * k;
        if (g < ngroups) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = r[k][0][j], y = r[k][1][j], z = r[k][2][j];
                float u = r[k][3][j], v = r[k][4][j], w = r[k][5][j];
                step1(x, y, z, u, v, w);
                r[k][0][j] = x; r[k][1][j] = y; r[k][2][j] = z;
                r[k][3][j] = u; r[k][4][j] = v; r[k][5][j] = w;
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) st<NT>(s.a[c] + g, r[k][c]);
        }
    }
}

// Tiled layout: one buffer, the six components of a wave's 64 groups stored back to back
// (6 KiB contiguous per wave), so the in-place update is ONE read stream and ONE write
// stream over the same addresses instead of twelve.
template <bool NT>
__global__ void update_tiled(f32x4 *buf, uint32_t ngroups)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t g = wave * 64u + lane;
    if (g >= ngroups) return;
    f32x4 *p = buf + (size_t)wave * 384u + lane;
    f32x4 r[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) r[c] = ld<NT>(p + 64 * c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float x = r[0][j], y = r[1][j], z = r[2][j], u = r[3][j], v = r[4][j], w = r[5][j];
        step1(x, y, z, u, v, w);
        r[0][j] = x; r[1][j] = y; r[2][j] = z; r[3][j] = u; r[4][j] = v; r[5][j] = w;
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) st<NT>(p + 64 * c, r[c]);
}

// In-place float4 scale: the ceiling of a single read-modify-write stream.
template <bool NT>
__global__ void inplace4(f32x4 *a, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) st<NT>(a + i, ld<NT>(a + i) * 0.99f);
}

template <bool NT>
__global__ void copy4(const f32x4 *src, f32x4 *dst, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) st<NT>(dst + i, ld<NT>(src + i));
}

// benchmark-like contents: positions uniform in [-50, 50), velocities in [-1, 1)
__global__ void fill_random(float *a, size_t n, float scale, float offset, uint32_t salt)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull + salt;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    a[i] = (float)(z >> 40) * 0x1.0p-24f * scale + offset;
}

static float time_launches(hipStream_t s, int reps, const std::function<void()> &launch)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    std::vector<float> ms;
    for (int round = 0; round < 3; ++round) {
        CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) launch();
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        float t;
        CHECK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t / reps);
    }
    CHECK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return ms[1];
}

template <int U, bool NT>
static void run_update(hipStream_t s, const Arrays &arr, uint32_t ngroups, int block, int lds,
                       double bytes)
{
    const uint32_t waves = (ngroups + 64u * U - 1) / (64u * U);
    const uint32_t wpb = block / 64;
    const uint32_t grid = (waves + wpb - 1) / wpb;
    if (lds > 65536)
        CHECK(hipFuncSetAttribute((const void *)update_u<U, NT>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const float ms = time_launches(s, 20, [&] {
        hipLaunchKernelGGL((update_u<U, NT>), dim3(grid), dim3(block), lds, s, arr, ngroups);
    });
    const int per_cu = lds ? 163840 / lds : 99;
    printf("update U=%d nt=%d block=%4d lds=%6d (<=%2d wg/CU, %3d waves/CU) : %.4f ms  %7.1f GB/s\n",
           U, (int)NT, block, lds, per_cu, std::min(32, per_cu * (block / 64)), ms,
           bytes / ms / 1e6);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000000ull;
    const uint32_t ngroups = (uint32_t)(n / 4);
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    Arrays arr;
    for (int c = 0; c < 6; ++c) {
        CHECK(hipMalloc((void **)&arr.a[c], (size_t)ngroups * 16));
        CHECK(hipMemsetAsync(arr.a[c], 0, (size_t)ngroups * 16, s));   // zeros: never in range
    }
    const double bytes = 48.0 * 4.0 * ngroups;

    // reference ceiling: float4 copy moving the same number of bytes (24 B read + 24 B written)
    {
        const size_t n4 = (size_t)ngroups * 6;      // 96 B read + 96 B written per group of 4
        f32x4 *src, *dst;
        CHECK(hipMalloc((void **)&src, n4 * 16));
        CHECK(hipMalloc((void **)&dst, n4 * 16));
        CHECK(hipMemsetAsync(src, 0, n4 * 16, s));
        if (argc > 2)
            hipLaunchKernelGGL(fill_random, dim3((unsigned)((n4 * 4 + 255) / 256)), dim3(256), 0, s,
                               (float *)src, n4 * 4, 100.0f, -50.0f, 7u);
        for (int nt = 0; nt < 2; ++nt)
            for (int lds : {0, 23040, 65536}) {
                const int block = lds == 23040 ? 64 : 256;
                const float ms = time_launches(s, 20, [&] {
                    const uint32_t grid = (uint32_t)((n4 + block - 1) / block);
                    if (nt) hipLaunchKernelGGL(copy4<true>, dim3(grid), dim3(block), lds, s, src, dst, n4);
                    else hipLaunchKernelGGL(copy4<false>, dim3(grid), dim3(block), lds, s, src, dst, n4);
                });
                printf("copy   nt=%d block=%4d lds=%6d : %.4f ms  %7.1f GB/s\n", nt, block, lds, ms,
                       bytes / ms / 1e6);
            }
        CHECK(hipFree(src));
        CHECK(hipFree(dst));
    }

    // tiled layout and the in-place ceiling
    {
        const uint32_t waves = (ngroups + 63u) / 64u;
        const size_t n4 = (size_t)waves * 384u;
        f32x4 *buf;
        CHECK(hipMalloc((void **)&buf, n4 * 16));
        CHECK(hipMemsetAsync(buf, 0, n4 * 16, s));
        if (argc > 2)
            hipLaunchKernelGGL(fill_random, dim3((unsigned)((n4 * 4 + 255) / 256)), dim3(256), 0, s,
                               (float *)buf, n4 * 4, 100.0f, -50.0f, 9u);
        for (int nt = 0; nt < 2; ++nt)
            for (int lds : {0, 16384, 23040, 32768}) {
                const int block = lds ? 64 : 256;
                const float ms = time_launches(s, 20, [&] {
                    const uint32_t grid = (uint32_t)((n4 + block - 1) / block);
                    if (nt) hipLaunchKernelGGL(inplace4<true>, dim3(grid), dim3(block), lds, s, buf, n4);
                    else hipLaunchKernelGGL(inplace4<false>, dim3(grid), dim3(block), lds, s, buf, n4);
                });
                printf("inplace nt=%d block=%4d lds=%6d : %.4f ms  %7.1f GB/s\n", nt, block, lds, ms,
                       bytes / ms / 1e6);
            }
        for (int nt = 0; nt < 2; ++nt)
            for (int lds : {0, 12288, 16384, 20480, 23040, 26624, 32768, 40960}) {
                const int block = 64;
                const float ms = time_launches(s, 20, [&] {
                    if (nt) hipLaunchKernelGGL(update_tiled<true>, dim3(waves), dim3(block), lds, s, buf, ngroups);
                    else hipLaunchKernelGGL(update_tiled<false>, dim3(waves), dim3(block), lds, s, buf, ngroups);
                });
                printf("tiled  nt=%d block=%4d lds=%6d (%3d waves/CU) : %.4f ms  %7.1f GB/s\n", nt, block,
                       lds, lds ? std::min(32, 163840 / lds) : 32, ms, bytes / ms / 1e6);
                fflush(stdout);
            }
        CHECK(hipFree(buf));
    }

    for (int lds : {0, 16384, 23040, 26624, 32768, 40960, 53248})
        run_update<1, true>(s, arr, ngroups, 64, lds, bytes);
    if (argc > 2) {        // any second argument: repeat with benchmark-like random contents
        for (int c = 0; c < 6; ++c) {
            const size_t nf = (size_t)ngroups * 4;
            hipLaunchKernelGGL(fill_random, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, s,
                               (float *)arr.a[c], nf, c < 3 ? 100.0f : 2.0f, c < 3 ? -50.0f : -1.0f,
                               (uint32_t)c);
        }
        printf("random contents (about 2 %% of the particles in range):\n");
        for (int lds : {16384, 20480, 23040, 26624, 32768})
            run_update<1, true>(s, arr, ngroups, 64, lds, bytes);
        return 0;
    }
    for (int lds : {23040, 26624, 32768, 40960, 53248, 81920})
        run_update<2, true>(s, arr, ngroups, 64, lds, bytes);
    for (int lds : {32768, 40960, 53248, 81920, 131072})
        run_update<4, true>(s, arr, ngroups, 64, lds, bytes);
    for (int lds : {40960, 65536, 81920})
        run_update<2, true>(s, arr, ngroups, 128, lds, bytes);
    run_update<1, false>(s, arr, ngroups, 64, 23040, bytes);
    run_update<2, false>(s, arr, ngroups, 64, 40960, bytes);
    return 0;
}
