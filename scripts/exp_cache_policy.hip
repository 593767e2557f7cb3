// exp_cache_policy.hip -- standalone experiment: the in-place 12-stream update with every
// combination of the gfx950 cache-policy bits (sc0, nt, sc1) on its loads and stores, through
// raw buffer accesses (aux bit 0 = sc0, bit 1 = nt, bit 4 = sc1).  Single-wave workgroups,
// 7 resident per CU, random contents.  Not part of the library.
//
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o exp_cache_policy exp_cache_policy.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(e)                                                                      \
    do {                                                                              \
        hipError_t r_ = (e);                                                          \
        if (r_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s (line %d)\n", #e, hipGetErrorString(r_), __LINE__); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

struct Arrays {
    float *a[6];
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

// XCD-contiguous variant: workgroups are dealt round-robin to the 8 XCDs, so workgroup b runs on
// XCD b % 8; here XCD k sweeps the k-th eighth of every array instead of every 8th KiB.
template <int LD, int ST>
__global__ void update_xcd(Arrays s, uint32_t ngroups, uint32_t bytes)
{
    const uint32_t per = (gridDim.x + 7u) / 8u;
    const uint32_t block = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    const uint32_t g = block * blockDim.x + threadIdx.x;
    if (block >= gridDim.x || g >= ngroups) return;
    const uint32_t off = g * 16u;
    f32x4 r[6];
    __amdgpu_buffer_rsrc_t rs[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        rs[c] = __builtin_amdgcn_make_buffer_rsrc(s.a[c], 0, (int)bytes, 0x00020000);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs[c], off, 0, LD);
        r[c] = __builtin_bit_cast(f32x4, v);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float x = r[0][j], y = r[1][j], z = r[2][j], u = r[3][j], v = r[4][j], w = r[5][j];
        step1(x, y, z, u, v, w);
        r[0][j] = x; r[1][j] = y; r[2][j] = z; r[3][j] = u; r[4][j] = v; r[5][j] = w;
    }
#pragma unroll
    for (int c = 0; c < 6; ++c)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r[c]), rs[c], off, 0, ST);
}

template <int LD, int ST>
__global__ void update(Arrays s, uint32_t ngroups, uint32_t bytes)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    const uint32_t off = g * 16u;
    f32x4 r[6];
    __amdgpu_buffer_rsrc_t rs[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        rs[c] = __builtin_amdgcn_make_buffer_rsrc(s.a[c], 0, (int)bytes, 0x00020000);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs[c], off, 0, LD);
        r[c] = __builtin_bit_cast(f32x4, v);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float x = r[0][j], y = r[1][j], z = r[2][j], u = r[3][j], v = r[4][j], w = r[5][j];
        step1(x, y, z, u, v, w);
        r[0][j] = x; r[1][j] = y; r[2][j] = z; r[3][j] = u; r[4][j] = v; r[5][j] = w;
    }
#pragma unroll
    for (int c = 0; c < 6; ++c)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r[c]), rs[c], off, 0, ST);
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

static const char *policy(int aux)
{
    switch (aux) {
    case 0: return "-        ";
    case 1: return "sc0      ";
    case 2: return "nt       ";
    case 3: return "sc0 nt   ";
    case 16: return "sc1      ";
    case 17: return "sc0 sc1  ";
    case 18: return "nt sc1   ";
    default: return "sc0 nt sc1";
    }
}

template <int LD, int ST>
static void run(hipStream_t s, const Arrays &arr, uint32_t ngroups, int lds)
{
    const uint32_t grid = (ngroups + 63) / 64;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((update<LD, ST>), dim3(grid), dim3(64), lds, s, arr, ngroups, ngroups * 16u);
    std::vector<float> ms;
    for (int round = 0; round < 3; ++round) {
        CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < 20; ++i)
            hipLaunchKernelGGL((update<LD, ST>), dim3(grid), dim3(64), lds, s, arr, ngroups,
                               ngroups * 16u);
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        float t;
        CHECK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t / 20);
    }
    CHECK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    printf("loads %s stores %s lds %5d: %.4f ms  %7.1f GB/s\n", policy(LD), policy(ST), lds, ms[1],
           192.0 * ngroups / ms[1] / 1e6);
    fflush(stdout);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

template <int LD>
static void run_stores(hipStream_t s, const Arrays &arr, uint32_t ngroups, int lds)
{
    run<LD, 0>(s, arr, ngroups, lds);
    run<LD, 1>(s, arr, ngroups, lds);
    run<LD, 2>(s, arr, ngroups, lds);
    run<LD, 3>(s, arr, ngroups, lds);
    run<LD, 16>(s, arr, ngroups, lds);
    run<LD, 17>(s, arr, ngroups, lds);
    run<LD, 18>(s, arr, ngroups, lds);
    run<LD, 19>(s, arr, ngroups, lds);
}

int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000000ull;
    const uint32_t ngroups = (uint32_t)(n / 4);
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    Arrays arr;
    for (int c = 0; c < 6; ++c) {
        // third argument: 1 = hipDeviceMallocUncached, 2 = hipDeviceMallocFinegrained
        const int kind = argc > 3 ? atoi(argv[3]) : 0;
        if (kind == 0) CHECK(hipMalloc((void **)&arr.a[c], (size_t)ngroups * 16));
        else CHECK(hipExtMallocWithFlags((void **)&arr.a[c], (size_t)ngroups * 16,
                                         kind == 1 ? hipDeviceMallocUncached
                                                   : hipDeviceMallocFinegrained));
        hipLaunchKernelGGL(fill_random, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, arr.a[c], n,
                           c < 3 ? 100.0f : 2.0f, c < 3 ? -50.0f : -1.0f, (uint32_t)c);
    }
    const int lds = argc > 2 ? atoi(argv[2]) : 23040;
    if (argc > 4) {            // fourth argument: compare the workgroup -> address mappings only
        const uint32_t grid = (ngroups + 63) / 64;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        for (int variant = 0; variant < 2; ++variant)
            for (int round = 0; round < 4; ++round) {
                CHECK(hipEventRecord(e0, s));
                for (int i = 0; i < 20; ++i) {
                    if (variant == 0)
                        hipLaunchKernelGGL((update<2, 18>), dim3(grid), dim3(64), lds, s, arr, ngroups,
                                           ngroups * 16u);
                    else
                        hipLaunchKernelGGL((update_xcd<2, 18>), dim3(grid), dim3(64), lds, s, arr,
                                           ngroups, ngroups * 16u);
                }
                CHECK(hipEventRecord(e1, s));
                CHECK(hipEventSynchronize(e1));
                float t;
                CHECK(hipEventElapsedTime(&t, e0, e1));
                printf("%s lds %5d: %.4f ms  %7.1f GB/s\n",
                       variant ? "XCD-contiguous eighths  " : "round-robin (index order)", lds, t / 20,
                       192.0 * ngroups / (t / 20) / 1e6);
            }
        CHECK(hipGetLastError());
        return 0;
    }
    printf("== %zu particles, single-wave workgroups, 7 per CU\n", n);
    run_stores<0>(s, arr, ngroups, lds);
    run_stores<1>(s, arr, ngroups, lds);
    run_stores<2>(s, arr, ngroups, lds);
    run_stores<3>(s, arr, ngroups, lds);
    run_stores<16>(s, arr, ngroups, lds);
    run_stores<17>(s, arr, ngroups, lds);
    run_stores<18>(s, arr, ngroups, lds);
    run_stores<19>(s, arr, ngroups, lds);
    return 0;
}
