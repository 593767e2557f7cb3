// exp_cache_resident.hip -- standalone experiment: for a state that fits the 256 MiB
// Infinity Cache (6.25 M and 10 M particles), which mix of plain / non-temporal loads and
// stores and how many resident waves per CU move the in-place update fastest?
// Not part of the library.
//
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o exp_cache_resident exp_cache_resident.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
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

template <bool NTL, bool NTS>
__global__ void update(Arrays s, uint32_t ngroups)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    f32x4 r[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        if constexpr (NTL) r[c] = __builtin_nontemporal_load(s.a[c] + g);
        else r[c] = s.a[c][g];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float x = r[0][j], y = r[1][j], z = r[2][j], u = r[3][j], v = r[4][j], w = r[5][j];
        step1(x, y, z, u, v, w);
        r[0][j] = x; r[1][j] = y; r[2][j] = z; r[3][j] = u; r[4][j] = v; r[5][j] = w;
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        if constexpr (NTS) __builtin_nontemporal_store(r[c], s.a[c] + g);
        else s.a[c][g] = r[c];
    }
}

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

template <bool NTL, bool NTS>
static void run(hipStream_t s, const Arrays &arr, uint32_t ngroups, int block, int lds)
{
    const uint32_t grid = (ngroups + block - 1) / block;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i)
        hipLaunchKernelGGL((update<NTL, NTS>), dim3(grid), dim3(block), lds, s, arr, ngroups);
    std::vector<float> ms;
    for (int round = 0; round < 3; ++round) {
        CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < 200; ++i)
            hipLaunchKernelGGL((update<NTL, NTS>), dim3(grid), dim3(block), lds, s, arr, ngroups);
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        float t;
        CHECK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t / 200);
    }
    CHECK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    const int waves = lds ? std::min(32, (163840 / lds) * (block / 64)) : 32;
    printf("loads %s stores %s block %4d lds %6d (%2d waves/CU): %.4f ms  %7.1f GB/s\n",
           NTL ? "nt   " : "plain", NTS ? "nt   " : "plain", block, lds, waves, ms[1],
           192.0 * ngroups / ms[1] / 1e6);
    fflush(stdout);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

int main(int argc, char **argv)
{
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    std::vector<size_t> sizes = {6250000, 10000000, 14000000};
    if (argc > 1) sizes = {(size_t)strtoull(argv[1], nullptr, 10)};   // e.g. 100000000: streaming
    for (size_t n : sizes) {
        const uint32_t ngroups = (uint32_t)(n / 4);
        Arrays arr;
        for (int c = 0; c < 6; ++c) {
            CHECK(hipMalloc((void **)&arr.a[c], (size_t)ngroups * 16));
            hipLaunchKernelGGL(fill_random, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                               (float *)arr.a[c], n, c < 3 ? 100.0f : 2.0f, c < 3 ? -50.0f : -1.0f,
                               (uint32_t)c);
        }
        printf("== %zu particles (%.0f MB of state)\n", n, 24.0 * n / 1e6);
        for (int lds : {0, 8192, 11776, 16384, 23040, 32768}) {
            run<false, false>(s, arr, ngroups, 64, lds);
            run<false, true>(s, arr, ngroups, 64, lds);
            run<true, false>(s, arr, ngroups, 64, lds);
            run<true, true>(s, arr, ngroups, 64, lds);
        }
        run<false, false>(s, arr, ngroups, 256, 0);
        run<false, true>(s, arr, ngroups, 256, 0);
        for (int c = 0; c < 6; ++c) CHECK(hipFree(arr.a[c]));
    }
    return 0;
}
