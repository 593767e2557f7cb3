// Experiment: how fast can the step run IN PLACE on the 64-byte record layout (the buffer
// the reference's compute shader works on), touching only the 32 bytes of position and
// velocity per record?  Variants:
//   simple  lane = particle: two 16-byte loads / stores at a 64-byte stride
//   pair    lane pair = record: each wave instruction touches 32 records x 32 consecutive
//           bytes; neighbouring lanes swap halves so that every lane owns one particle
//   whole   linear 16-byte accesses over whole records (reads all 64 bytes), lanes whose
//           quad is position or velocity write back
// Build: hipcc --offload-arch=gfx950 -O3 -o exp_aos_inplace scripts/exp_aos_inplace.hip
// Run:   ./exp_aos_inplace [particles] [reserve_lds_bytes]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(e)                                                                          \
    do {                                                                                  \
        hipError_t r_ = (e);                                                              \
        if (r_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_));                       \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

template <bool NT>
__device__ __forceinline__ f32x4 ld(const f32x4 *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT>
__device__ __forceinline__ void st(f32x4 *p, f32x4 v)
{
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

__device__ __forceinline__ void step(f32x4 &q0, f32x4 &q1, float dt, float g, float damp)
{
    q1[1] = q1[1] - g * dt;
    q0[0] = q0[0] + q1[0] * dt;
    q0[1] = q0[1] + q1[1] * dt;
    q0[2] = q0[2] + q1[2] * dt;
    q1[0] = q1[0] * damp;
    q1[1] = q1[1] * damp;
    q1[2] = q1[2] * damp;
}

template <bool NTL, bool NTS>
__global__ void __launch_bounds__(64) simple_kernel(f32x4 *aos, uint32_t n, float dt, float g,
                                                    float damp)
{
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n) return;
    f32x4 q0 = ld<NTL>(aos + 4ull * i), q1 = ld<NTL>(aos + 4ull * i + 1);
    step(q0, q1, dt, g, damp);
    st<NTS>(aos + 4ull * i, q0);
    st<NTS>(aos + 4ull * i + 1, q1);
}

__device__ __forceinline__ f32x4 swap_neighbour(f32x4 v)
{
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        r[k] = __builtin_bit_cast(
            float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v[k]), 0xB1 /* quad_perm 1,0,3,2 */,
                                            0xF, 0xF, true));
    return r;
}

// n must be a multiple of 64 here (experiment only).
template <bool NTL, bool NTS>
__global__ void __launch_bounds__(64) pair_kernel(f32x4 *aos, uint32_t n, float dt, float g,
                                                  float damp)
{
    const uint32_t base = blockIdx.x * 64u;
    const uint32_t lane = threadIdx.x, odd = lane & 1u;
    if (base >= n) return;
    f32x4 *pa = aos + 4ull * (base + (lane >> 1)) + odd;
    f32x4 *pb = aos + 4ull * (base + 32u + (lane >> 1)) + odd;
    const f32x4 a = ld<NTL>(pa), b = ld<NTL>(pb);
    const f32x4 got = swap_neighbour(odd ? a : b);
    f32x4 q0 = odd ? got : a, q1 = odd ? b : got;
    step(q0, q1, dt, g, damp);
    const f32x4 back = swap_neighbour(odd ? q0 : q1);
    st<NTS>(pa, odd ? back : q0);
    st<NTS>(pb, odd ? q1 : back);
}

// Whole records, linear: lane t of instruction k touches quad 64k + t of the wave's 256
// quads = 64 records.  Position and velocity of a record sit in lanes 4r, 4r+1.
template <bool NTL, bool NTS>
__global__ void __launch_bounds__(64) whole_kernel(f32x4 *aos, uint32_t n, float dt, float g,
                                                   float damp)
{
    const uint32_t base = blockIdx.x * 64u;
    const uint32_t lane = threadIdx.x;
    if (base >= n) return;
    f32x4 *p = aos + 4ull * base + lane;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f32x4 v = ld<NTL>(p + 64 * k);
        const f32x4 other = swap_neighbour(v);
        const uint32_t field = lane & 3u;
        f32x4 q0 = field == 0 ? v : other, q1 = field == 0 ? other : v;
        step(q0, q1, dt, g, damp);
        if (field < 2) st<NTS>(p + 64 * k, field == 0 ? q0 : q1);
    }
}

template <typename K>
static void run(const char *name, K kernel, f32x4 *aos, uint32_t n, uint32_t reserve,
                double bytes_per_particle)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const uint32_t grid = (n + 63u) / 64u;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), reserve, 0, aos, n, 1e-3f, 9.8f, 0.999f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        for (int k = 0; k < 10; ++k)
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), reserve, 0, aos, n, 1e-3f, 9.8f, 0.999f);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / 10 < best) best = ms / 10;
    }
    printf("%-22s reserve %6u: %8.4f ms/step  %7.2f G updates/s  %7.1f GB/s of %g B\n", name,
           reserve, best, n / best / 1e6, bytes_per_particle * n / best / 1e6, bytes_per_particle);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)atol(argv[1]) & ~63u : 100000000u;
    f32x4 *aos;
    CHECK(hipMalloc(&aos, (size_t)n * 64));
    CHECK(hipMemset(aos, 0, (size_t)n * 64));
    const uint32_t reserves[] = {0, 6144, 11776, 16384, 23040, 26624};
    for (uint32_t r : reserves) {
        if (argc > 2 && (uint32_t)atol(argv[2]) != r) continue;
        run("simple plain/plain", simple_kernel<false, false>, aos, n, r, 64);
        run("simple nt/plain", simple_kernel<true, false>, aos, n, r, 64);
        run("simple nt/nt", simple_kernel<true, true>, aos, n, r, 64);
        run("pair plain/plain", pair_kernel<false, false>, aos, n, r, 64);
        run("pair nt/plain", pair_kernel<true, false>, aos, n, r, 64);
        run("pair nt/nt", pair_kernel<true, true>, aos, n, r, 64);
        run("whole plain/plain", whole_kernel<false, false>, aos, n, r, 96);
        run("whole nt/nt", whole_kernel<true, true>, aos, n, r, 96);
    }
    CHECK(hipFree(aos));
    return 0;
}
