// exp_valu_issue.hip -- what one vector instruction costs a SIMD of the MI355X when several
// waves share it: the yardstick of the kernels here that are bound by instruction issue
// (fused batches, long attractor lists).  Each kernel runs a loop of 64 independent
// instructions of one kind; the grid puts W waves on every SIMD (blocks of 256 threads = one
// wave per SIMD, 256 x W blocks), every wave times itself with s_memtime, and the host reports
// cycles per instruction per SIMD = wave cycles / instructions / ... x 1 (the W waves of a SIMD
// run side by side, so a wave's own time per instruction divided by W is the SIMD's cost).
//
// build: hipcc -O2 --offload-arch=gfx950 -o exp_valu_issue exp_valu_issue.hip
// run:   ./exp_valu_issue            (a table on stdout)
#include <hip/hip_runtime.h>

#include <algorithm>
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

constexpr int kUnroll = 64, kLoops = 2000;

// 16 accumulators a0..a15 and two operands b, c stay in registers; BODY is one instruction on
// accumulator `A`.
#define REPEAT16(I)                                                                       \
    I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7) I(8) I(9) I(10) I(11) I(12) I(13) I(14) I(15)

#define KERNEL(name, INSTR)                                                               \
    __global__ void __launch_bounds__(256) name(float *out, unsigned long long *cycles)   \
    {                                                                                     \
        float a[16];                                                                      \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = 1.0f + threadIdx.x * 1e-3f + i; \
        float b = 1.0000001f + blockIdx.x * 1e-9f, c = 1e-7f;                             \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                      \
        for (int l = 0; l < kLoops; ++l) {                                                \
            _Pragma("unroll") for (int r = 0; r < kUnroll / 16; ++r) { REPEAT16(INSTR) }  \
        }                                                                                 \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                      \
        float s = 0.0f;                                                                   \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) s += a[i];                         \
        out[blockIdx.x * 256 + threadIdx.x] = s + b + c;                                  \
        if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0; \
    }

#define I_FMA(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define I_MUL(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_ADD(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
#define I_MOV(k) asm volatile("v_mov_b32 %0, %1" : "+v"(a[k]) : "v"(b));
#define I_RCP(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
#define I_SQRT(k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k]));
#define I_CMP(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc");
#define I_CMP_S(k) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(a[k]), "v"(b) : "s20", "s21");
#define I_CND(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : "vcc");
#define I_CND_S(k) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a[k]) : "v"(b) : "s20", "s21");
#define I_DSCALE(k) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[k]) : "v"(b) : "vcc");
#define I_DFMAS(k) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c) : "vcc");
#define I_DFIX(k) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define I_DPP(k) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k]) : "v"(b));
#define I_ADD_U(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
#define I_CMP_CND(k) \
    asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : "vcc");

KERNEL(k_fma, I_FMA)
KERNEL(k_mul, I_MUL)
KERNEL(k_add, I_ADD)
KERNEL(k_mov, I_MOV)
KERNEL(k_rcp, I_RCP)
KERNEL(k_sqrt, I_SQRT)
KERNEL(k_cmp, I_CMP)
KERNEL(k_cmp_s, I_CMP_S)
KERNEL(k_cnd, I_CND)
KERNEL(k_cnd_s, I_CND_S)
KERNEL(k_dscale, I_DSCALE)
KERNEL(k_dfmas, I_DFMAS)
KERNEL(k_dfix, I_DFIX)
KERNEL(k_dpp, I_DPP)
KERNEL(k_add_u, I_ADD_U)
KERNEL(k_cmp_cnd, I_CMP_CND)

// packed fp32: two accumulators per instruction (even-aligned register pairs)
__global__ void __launch_bounds__(256) k_pk_mul(float *out, unsigned long long *cycles)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = f32x2{1.0f + threadIdx.x * 1e-3f + i, 2.0f + i};
    f32x2 b = {1.0000001f + blockIdx.x * 1e-9f, 0.9999999f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < kLoops; ++l) {
#pragma unroll
        for (int r = 0; r < kUnroll / 16; ++r) {
#define I_PK(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
            REPEAT16(I_PK)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i][0] + a[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

__global__ void __launch_bounds__(256) k_pk_fma(float *out, unsigned long long *cycles)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = f32x2{1.0f + threadIdx.x * 1e-3f + i, 2.0f + i};
    f32x2 b = {1.0000001f + blockIdx.x * 1e-9f, 0.9999999f}, c = {1e-7f, 2e-7f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < kLoops; ++l) {
#pragma unroll
        for (int r = 0; r < kUnroll / 16; ++r) {
#define I_PKF(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
            REPEAT16(I_PKF)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i][0] + a[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

typedef void (*Kernel)(float *, unsigned long long *);

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int max_w = 8;
    float *out;
    unsigned long long *cycles;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * cus * max_w));
    CHECK(hipMalloc(&cycles, sizeof(unsigned long long) * 4 * cus * max_w));
    struct Row { const char *name; Kernel k; int per_body; };
    const Row rows[] = {
        {"v_fma_f32", k_fma, 1}, {"v_mul_f32", k_mul, 1}, {"v_add_f32", k_add, 1},
        {"v_add_u32", k_add_u, 1}, {"v_mov_b32", k_mov, 1}, {"v_mov_b32_dpp row_shr:1", k_dpp, 1},
        {"v_pk_mul_f32", k_pk_mul, 1}, {"v_pk_fma_f32", k_pk_fma, 1},
        {"v_rcp_f32", k_rcp, 1}, {"v_sqrt_f32", k_sqrt, 1},
        {"v_cmp_lt_f32 vcc", k_cmp, 1}, {"v_cmp_lt_f32 s[20:21]", k_cmp_s, 1},
        {"v_cndmask_b32 vcc", k_cnd, 1}, {"v_cndmask_b32 s[20:21]", k_cnd_s, 1},
        {"v_cmp + v_cndmask (pair)", k_cmp_cnd, 2},
        {"v_div_scale_f32", k_dscale, 1}, {"v_div_fmas_f32", k_dfmas, 1},
        {"v_div_fixup_f32", k_dfix, 1},
    };
    printf("%d CUs; cycles per instruction per SIMD with W waves on every SIMD "
           "(median wave, s_memtime ticks)\n", cus);
    printf("%-28s", "instruction");
    for (int w : {1, 2, 4, 8}) printf("   W=%d  ", w);
    printf("\n");
    std::vector<unsigned long long> host(4 * cus * max_w);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (const Row &row : rows) {
        printf("%-28s", row.name);
        for (int w : {1, 2, 4, 8}) {
            const int blocks = cus * w;
            hipLaunchKernelGGL(row.k, dim3(blocks), dim3(256), 0, 0, out, cycles);      // warm
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(row.k, dim3(blocks), dim3(256), 0, 0, out, cycles);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipGetLastError());
            CHECK(hipDeviceSynchronize());
            float ms = 0.0f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(host.data(), cycles, sizeof(unsigned long long) * 4 * blocks,
                            hipMemcpyDeviceToHost));
            std::sort(host.begin(), host.begin() + 4 * blocks);
            const double wave = (double)host[2 * blocks];
            const double per = wave / ((double)kLoops * kUnroll * row.per_body) / w;
            printf(" %7.2f ", per);
            if (w == 8) {
                // the whole launch by the host's clock: instructions per SIMD per microsecond,
                // and what a tick of s_memtime is worth
                const double instr_per_simd = (double)kLoops * kUnroll * row.per_body * w;
                printf("  | W=8 launch %.3f ms: %.0f instr/us/SIMD, %.0f ticks/us", ms,
                       instr_per_simd / (ms * 1e3), wave / (ms * 1e3));
            }
        }
        printf("\n");
    }
    CHECK(hipFree(out));
    CHECK(hipFree(cycles));
    return 0;
}
