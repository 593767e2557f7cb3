// psim_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the particle-update path.
//
// The update is a pure HBM stream: 24 B read + 24 B written per particle
// (px,py,pz,vx,vy,vz as six fp32 arrays), < 1 flop/B, so there is no MFMA here.
// Each lane moves 16 B per array per access (global_load/store_dwordx4, 1 KiB
// per wave instruction); the step constants arrive as kernel arguments (SGPRs);
// the attractor list of the multi-attractor form is staged once per workgroup
// into LDS and read back as same-address broadcasts.
//
// Arithmetic follows the reference CPU path (src/simulation/cpu.rs:65-113) one
// rounding per source-level operation: this file is compiled with
// -ffp-contract=off (rustc never forms FMAs) and with hipcc's default correctly
// rounded fp32 divide and sqrt, denormals kept.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <type_traits>

#include "psim_kernels.h"
#include "psim_threshold.h"
#include "psim_tuning.h"

#pragma clang fp contract(off)

namespace psim {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- memory access helpers -------------------------------------------------

template <bool NT, typename T>
__device__ __forceinline__ T ld(const T *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

template <bool NT, typename T>
__device__ __forceinline__ void st(T *p, T v)
{
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// Store of one element (4, 8 or 16 bytes) per lane with explicit cache-policy bits: the wave
// writes elements first .. first+63 of `array`, lanes at or past element `limit` are dropped
// by the bounds check of the buffer descriptor.  The compiler emits neither `sc1` alone nor
// `nt sc1` for a plain or a non-temporal global store, and those are the fast ones.
// Measured on the in-place update (GB/s of the 48 B stream):
//   100 M particles (streaming, loads nt): stores plain 5.86 k | nt 6.46 k | sc0 sc1 6.51 k |
//                                          nt sc1 6.57 k
//   10 M / 6.25 M / 4 M (fit the Infinity Cache, loads plain): stores plain 7.03 / 6.83 /
//                                          6.44 k | sc1 7.15 / 7.10 / 6.62 k | nt 6.3 k
//   1 M (fits the L2s): stores plain 7.2 k | sc1 5.1 k
// `first` must be the same for every lane of the wave.
#define PSIM_BUFFER_WORD3 0x00020000   /* gfx9 family: raw buffer, 32-bit data format */
#define PSIM_POLICY_SC1 16             /* aux bits of the raw buffer intrinsics: 1 sc0, 2 nt, 16 sc1 */
#define PSIM_POLICY_NT_SC1 18

template <int POLICY, typename T>
__device__ __forceinline__ void st_policy(T *array, uint64_t first, uint64_t limit,
                                          uint32_t lane, T v)
{
    static_assert(sizeof(T) == 4 || sizeof(T) == 8 || sizeof(T) == 16, "dword, dwordx2, dwordx4");
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)first);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(first >> 32));
    const uint64_t f = ((uint64_t)hi << 32) | lo;
    const uint64_t left = limit > f ? limit - f : 0;
    const uint32_t bytes = (uint32_t)(left < 64 ? left : 64) * (uint32_t)sizeof(T);
    const __amdgpu_buffer_rsrc_t r =
        __builtin_amdgcn_make_buffer_rsrc(array + f, 0, (int)bytes, PSIM_BUFFER_WORD3);
    const uint32_t off = lane * (uint32_t)sizeof(T);
    if constexpr (sizeof(T) == 16)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, POLICY);
    else if constexpr (sizeof(T) == 8)
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, off, 0, POLICY);
    else
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, off, 0, POLICY);
}

// The streaming form used by the kernels that only ever stream (generator, transposes).
template <typename T>
__device__ __forceinline__ void st_stream(T *array, uint64_t first, uint64_t limit,
                                          uint32_t lane, T v)
{
    st_policy<PSIM_POLICY_NT_SC1>(array, first, limit, lane, v);
}

// One 16-byte quad per lane of the state arrays, by access mode.  `g0` is the wave's first
// quad, `g` = g0 + lane the lane's own; lanes past `limit` must not call the plain form.
template <int MODE>
__device__ __forceinline__ void st_state(f32x4 *array, uint32_t g0, uint32_t limit, uint32_t lane,
                                         f32x4 v)
{
    if constexpr (MODE == PSIM_ACCESS_STREAM) st_policy<PSIM_POLICY_NT_SC1>(array, g0, limit, lane, v);
    else if constexpr (MODE == PSIM_ACCESS_RESIDENT) st_policy<PSIM_POLICY_SC1>(array, g0, limit, lane, v);
    else array[g0 + lane] = v;
}

// ---- the per-particle step (src/simulation/cpu.rs:66-112) -------------------

// glam Vec3::length: sqrt((x*x + y*y) + z*z).
__device__ __forceinline__ float length3(float x, float y, float z)
{
    return sqrtf((x * x + y * y) + z * z);
}

// Force law of one attractor, cpu.rs:76-83.  `two_r` is radius * 2.0, `thr` =
// psim_in_range_threshold(two_r): `d2 < thr` is exactly the reference's `dist < two_r`
// (psim_threshold.h).
// The part behind the test: dir = (dx, dy, dz), d2 = its squared length.
__device__ __forceinline__ void attract_in_range(float &vx, float &vy, float &vz, float dx,
                                                 float dy, float dz, float d2, float aforce,
                                                 float two_r, float dt)
{
    const float dist = sqrtf(d2);                            // cpu.rs:77
    const float t = 1.0f - dist / two_r;
    const float ff = (t * t) * 2.0f;                         // cpu.rs:80 powi(2) * 2.0
    const float inv = 1.0f / dist;                           // normalize = dir * (1/len)
    const float fx = ((dx * inv) * aforce) * ff;             // cpu.rs:81
    const float fy = ((dy * inv) * aforce) * ff;
    const float fz = ((dz * inv) * aforce) * ff;
    vx = vx + fx * dt;                                       // cpu.rs:82
    vy = vy + fy * dt;
    vz = vz + fz * dt;
}

__device__ __forceinline__ void attract(float &vx, float &vy, float &vz, float px, float py,
                                        float pz, float ax, float ay, float az, float aforce,
                                        float two_r, float thr, float dt)
{
    const float dx = ax - px, dy = ay - py, dz = az - pz;    // cpu.rs:76
    const float d2 = (dx * dx + dy * dy) + dz * dz;
    if (d2 < thr)                                            // cpu.rs:79, see above
        attract_in_range(vx, vy, vz, dx, dy, dz, d2, aforce, two_r, dt);
}

// The single attractor of SimParams applied to a lane's four particles, for the kernels that
// are bound by instruction issue (the fused batches).  The in-range test is cheap, the force
// law behind it (attract_in_range: a correctly rounded square root and two divisions) costs
// about as much as everything else in a step, and a wave runs it once per particle SLOT in
// which any of its 64 lanes has a hit.  With r = 10 in the 100-unit cube 1.9 % of the
// particles are in range: 2.8 of the 4 slots have a hit somewhere in the wave, although hardly
// any lane has two.  So, when it pays, every lane first takes the FIRST of its particles that
// is in range through one common run of the force law (selecting its inputs and scattering
// its result costs about one more run), and what is left -- lanes with a second hit -- goes
// slot by slot.  Whether it pays is decided per wave and step from wave-wide ballots: the
// common run replaces (slots with a hit) - (slots still needed afterwards) slot runs and is
// taken when that is 2 or more.  Small radii (at most one slot with a hit) and large ones
// (every lane in range everywhere) therefore take the slot runs alone.  16 fused steps at 100 M
// particles, ms with / without: r = 5 1.26 / 1.26, r = 10 1.66 / 2.00, r = 15 2.13 / 2.45, r = 20
// and up equal.  Each particle gets exactly the operations of attract() either way.
// PSIM_FIRST_HIT = 0 builds the kernels without it.
#ifndef PSIM_FIRST_HIT
#define PSIM_FIRST_HIT 1
#endif
#ifndef PSIM_FIRST_HIT_SINGLE       /* the same in the one-step kernel, which is HBM-bound */
#define PSIM_FIRST_HIT_SINGLE 0
#endif
__device__ __forceinline__ void attract4_first_hits(float (&vx)[4], float (&vy)[4], float (&vz)[4],
                                                    const float (&px)[4], const float (&py)[4],
                                                    const float (&pz)[4], float ax, float ay,
                                                    float az, float aforce, float two_r, float thr,
                                                    float dt)
{
    float dx[4], dy[4], dz[4], d2[4];
    bool take[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        dx[j] = ax - px[j], dy[j] = ay - py[j], dz[j] = az - pz[j];      // cpu.rs:76
        d2[j] = (dx[j] * dx[j] + dy[j] * dy[j]) + dz[j] * dz[j];
        take[j] = d2[j] < thr;                                           // cpu.rs:79
    }
    // wave-uniform: slots with a hit, and slots that still have one once every lane's first
    // hit is dealt with
    const bool second1 = take[1] && take[0], second2 = take[2] && (take[0] || take[1]),
               second3 = take[3] && (take[0] || take[1] || take[2]);
    const int slots = (__builtin_amdgcn_ballot_w64(take[0]) != 0) + (__builtin_amdgcn_ballot_w64(take[1]) != 0) +
                      (__builtin_amdgcn_ballot_w64(take[2]) != 0) + (__builtin_amdgcn_ballot_w64(take[3]) != 0);
    const int left = (__builtin_amdgcn_ballot_w64(second1) != 0) + (__builtin_amdgcn_ballot_w64(second2) != 0) +
                     (__builtin_amdgcn_ballot_w64(second3) != 0);
    if (slots - left >= 2) {
        // one-hot: the lane's first particle in range
        const bool s0 = take[0], s1 = take[1] && !take[0], s2 = take[2] && !take[0] && !take[1],
                   s3 = take[3] && !take[0] && !take[1] && !take[2];
        if (s0 || s1 || s2 || s3) {
            // Values, not array elements, go into the selections: `c ? v[0] : v[1]` on array
            // elements is an lvalue -- a selected ADDRESS -- and would put the arrays into
            // scratch memory (measured: 3.3 times slower).
            const auto pick = [&](const float (&v)[4]) {
                const float e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3];
                return s0 ? e0 : (s1 ? e1 : (s2 ? e2 : e3));
            };
            const float sx = pick(dx), sy = pick(dy), sz = pick(dz), sd = pick(d2);
            float ux = pick(vx), uy = pick(vy), uz = pick(vz);
            attract_in_range(ux, uy, uz, sx, sy, sz, sd, aforce, two_r, dt);
            const auto put = [&](float (&v)[4], float u) {
                const float e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3];
                v[0] = s0 ? u : e0; v[1] = s1 ? u : e1; v[2] = s2 ? u : e2; v[3] = s3 ? u : e3;
            };
            put(vx, ux); put(vy, uy); put(vz, uz);
        }
        take[0] = false, take[1] = second1, take[2] = second2, take[3] = second3;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (take[j])
            attract_in_range(vx[j], vy[j], vz[j], dx[j], dy[j], dz[j], d2[j], aforce, two_r, dt);
}

template <int N>
__device__ __forceinline__ void apply_gravity(float (&vy)[N], const psim_sim_params &p)
{
    const float gdt = p.gravity * p.delta_time;
    if constexpr (N == 4) {
        // two particles per packed fp32 instruction; every element rounds as the scalar
        // operation does
        const f32x2 g2 = {gdt, gdt};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x2 r = f32x2{vy[2 * h], vy[2 * h + 1]} - g2;
            vy[2 * h] = r[0]; vy[2 * h + 1] = r[1];
        }
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) vy[j] = vy[j] - gdt;     // cpu.rs:72
    }
}

template <int N>
__device__ __forceinline__ void integrate(float (&px)[N], float (&py)[N], float (&pz)[N],
                                          float (&vx)[N], float (&vy)[N], float (&vz)[N],
                                          const psim_sim_params &p)
{
    const float dt = p.delta_time;
    if constexpr (N == 4) {
        const f32x2 dt2 = {dt, dt}, damp2 = {p.damping, p.damping};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int a = 2 * h, b = 2 * h + 1;
            f32x2 x = {px[a], px[b]}, y = {py[a], py[b]}, z = {pz[a], pz[b]};
            f32x2 u = {vx[a], vx[b]}, v = {vy[a], vy[b]}, w = {vz[a], vz[b]};
            x = x + u * dt2; y = y + v * dt2; z = z + w * dt2;           // cpu.rs:87
            u = u * damp2; v = v * damp2; w = w * damp2;                 // cpu.rs:90
            px[a] = x[0]; px[b] = x[1]; py[a] = y[0]; py[b] = y[1]; pz[a] = z[0]; pz[b] = z[1];
            vx[a] = u[0]; vx[b] = u[1]; vy[a] = v[0]; vy[b] = v[1]; vz[a] = w[0]; vz[b] = w[1];
        }
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            px[j] = px[j] + vx[j] * dt;                      // cpu.rs:87
            py[j] = py[j] + vy[j] * dt;
            pz[j] = pz[j] + vz[j] * dt;
            vx[j] = vx[j] * p.damping;                       // cpu.rs:90
            vy[j] = vy[j] * p.damping;
            vz[j] = vz[j] * p.damping;
        }
    }
}

// ---- attractor lists through a per-wave queue of in-range pairs -------------------
//
// With a list, the in-range test (9 VALU instructions per particle and attractor) is cheap
// and the force law behind it (a square root, two divisions: ~60 instructions) is dear.  With
// r = 10 in the 100-unit cube only 1.9 % of the particles are in range of an attractor, yet
// 71 % of the 64-lane wave instructions of the straightforward loop run the force law for
// at least one lane.  Here a wave first tests its 256 particles against a chunk of
// PSIM_PAIR_CHUNK attractors, remembering hits as bit masks; the lanes then write their hits
// one after the other into a queue in LDS (positions from a wave-wide prefix sum), the
// wave works the queue off with every lane busy, and each lane adds the resulting velocity
// terms to its particles in attractor order.  The force on a particle depends on positions
// only, which do not change before all attractors are applied, so every term has the value
// the loop would compute; the additions happen in the same order; the result is the same
// bit for bit.  A chunk with more hits than the queue holds (large radii) takes the loop.
// Sizes measured at 100 M particles, 64 attractors, r = 10 (ms per step; chunk / queue):
// 16 / 256 2.34 | 32 / 512 2.39 | 16 / 512 2.54 | 8 / 256 2.63 | 32 / 256 6.99 (overflows).
#define PSIM_PAIR_CHUNK 16u     /* attractors tested per round: the hit mask of a particle */
#define PSIM_PAIR_QUEUE 256u    /* in-range (particle, attractor) pairs a wave queues per round */

// Exclusive prefix sum of `x` over the 64 lanes (all active); `total` = the wave's sum, in
// a scalar register.  Seven data-parallel-primitive adds (row shifts, then the two row
// broadcasts of the gfx9 family); no LDS traffic.
__device__ __forceinline__ uint32_t wave_exclusive_scan(uint32_t x, uint32_t &total)
{
    // update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl): lanes that are masked
    // out or whose source lies outside the row keep `old` = 0.
#define PSIM_DPP(src, ctrl, rows, banks) \
    (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(src), ctrl, rows, banks, false)
    uint32_t v = x + PSIM_DPP(x, 0x111, 0xF, 0xF);           // row_shr:1
    v += PSIM_DPP(x, 0x112, 0xF, 0xF);                       // row_shr:2
    v += PSIM_DPP(x, 0x113, 0xF, 0xF);                       // row_shr:3  -> sums of 4 within a row
    v += PSIM_DPP(v, 0x114, 0xF, 0xE);                       // row_shr:4, banks 1-3
    v += PSIM_DPP(v, 0x118, 0xF, 0xC);                       // row_shr:8, banks 2-3 -> rows done
    v += PSIM_DPP(v, 0x142, 0xA, 0xF);                       // row_bcast:15 into rows 1 and 3
    v += PSIM_DPP(v, 0x143, 0xC, 0xF);                       // row_bcast:31 into rows 2 and 3
#undef PSIM_DPP
    total = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    return v - x;
}

// Orders one wave's LDS writes before its LDS reads that follow (LDS operations of a wave
// execute in order; this pins the compiler's ordering).
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The attractor loop of step<2, 4> for a whole wave (every lane must call; lanes without
// particles pass live = false).  `alist` is the list in LDS with thresholds in _pad[0],
// `queue` the wave's PSIM_PAIR_QUEUE quads of LDS.
__device__ __forceinline__ void attract_list_queued(float (&px)[4], float (&py)[4], float (&pz)[4],
                                                    float (&vx)[4], float (&vy)[4], float (&vz)[4],
                                                    const psim_attractor *alist,
                                                    const f32x4 *tests, uint32_t na,
                                                    float dt, f32x4 *queue, uint32_t lane,
                                                    bool live)
{
    for (uint32_t c0 = 0; c0 < na; c0 += PSIM_PAIR_CHUNK) {
        const uint32_t cn = (na - c0) < PSIM_PAIR_CHUNK ? (na - c0) : PSIM_PAIR_CHUNK;
        uint32_t mask[4] = {0u, 0u, 0u, 0u};
        // `tests` holds one 16-byte record per attractor, {position, threshold}, the
        // threshold 0 for an attractor that is switched off (no squared distance is below
        // it): one LDS read and no branch per attractor, so that the reads of the unrolled
        // loop below go out ahead of the arithmetic instead of one round trip at a time.
        const auto test = [&](uint32_t a) {                  // hits of attractor c0 + a
            const f32x4 at = tests[c0 + a];
            const float thr = at[3];
            // two particles per packed fp32 instruction (each lane of a packed operation
            // rounds like the scalar one; nothing is fused)
            const f32x2 ax = {at[0], at[0]}, ay = {at[1], at[1]}, az = {at[2], at[2]};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x2 dx = ax - f32x2{px[2 * h], px[2 * h + 1]},
                            dy = ay - f32x2{py[2 * h], py[2 * h + 1]},
                            dz = az - f32x2{pz[2 * h], pz[2 * h + 1]};
                const f32x2 d2 = (dx * dx + dy * dy) + dz * dz;
                mask[2 * h] |= ((live && d2[0] < thr) ? 1u : 0u) << a;           // cpu.rs:79
                mask[2 * h + 1] |= ((live && d2[1] < thr) ? 1u : 0u) << a;
            }
        };
        // The chunk's first attractor alone tells (nearly for free: four scalar population
        // counts) whether the chunk is likely to overflow the queue, as large radii do;
        // the other tests are then not spent.  A wrong guess costs time, not correctness.
        test(0);
        uint32_t first_hits = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            first_hits += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(mask[j] & 1u));
        uint32_t total = PSIM_PAIR_QUEUE + 1u, base = 0;
        if (first_hits * cn <= PSIM_PAIR_QUEUE) {
#pragma unroll 2
            for (uint32_t a = 1; a < cn; ++a) test(a);
            const uint32_t hits =
                (uint32_t)(__builtin_popcount(mask[0]) + __builtin_popcount(mask[1]) +
                           __builtin_popcount(mask[2]) + __builtin_popcount(mask[3]));
            base = wave_exclusive_scan(hits, total);
            if (total == 0) continue;
        }
        if (total > PSIM_PAIR_QUEUE) {                       // too many for the queue: the loop
            for (uint32_t a = 0; a < cn; ++a) {
                const psim_attractor at = alist[c0 + a];
                if (at.active > 0) {
                    const float two_r = at.radius * 2.0f;
                    const float thr = __uint_as_float(at._pad[0]);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (live)
                            attract(vx[j], vy[j], vz[j], px[j], py[j], pz[j], at.position[0],
                                    at.position[1], at.position[2], at.force, two_r, thr, dt);
                }
            }
            continue;
        }
        // the lane's hits, particle by particle and in attractor order, at base, base + 1, ...
        uint32_t k = base;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            for (uint32_t m = mask[j]; m; m &= m - 1u)
                queue[k++] = f32x4{px[j], py[j], pz[j],
                                   __uint_as_float(c0 + (uint32_t)__builtin_ctz(m))};
        wave_lds_sync();
        for (uint32_t q = lane; q < total; q += 64u) {      // cpu.rs:76-82 of one pair per lane
            const f32x4 e = queue[q];
            const psim_attractor &at = alist[__float_as_uint(e[3])];
            const float two_r = at.radius * 2.0f;
            const float dx = at.position[0] - e[0], dy = at.position[1] - e[1],
                        dz = at.position[2] - e[2];
            const float d2 = (dx * dx + dy * dy) + dz * dz;
            const float dist = sqrtf(d2);
            const float t = 1.0f - dist / two_r;
            const float ff = (t * t) * 2.0f;
            const float inv = 1.0f / dist;
            const float fx = ((dx * inv) * at.force) * ff;
            const float fy = ((dy * inv) * at.force) * ff;
            const float fz = ((dz * inv) * at.force) * ff;
            queue[q] = f32x4{fx * dt, fy * dt, fz * dt, 0.0f};
        }
        wave_lds_sync();
        k = base;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            for (uint32_t m = mask[j]; m; m &= m - 1u) {
                const f32x4 term = queue[k++];
                vx[j] = vx[j] + term[0];                     // cpu.rs:82
                vy[j] = vy[j] + term[1];
                vz[j] = vz[j] + term[2];
            }
        wave_lds_sync();                                     // the next round rewrites the queue
    }
}

// FORCE: 0 = no attractor, 1 = the single attractor of SimParams,
//        2 = attractor list in LDS, 3 = attractor list in kernel arguments,
//        4 = attractor list in LDS, worked off through the queue of in-range pairs
//            (attract_list_queued; step<4, N> itself is the loop of 2, for the tail).
//
// Steps N particles held in registers.  Particles are independent, so for a list the
// attractor loop is the OUTER loop: each attractor record is fetched once per N
// particles and the N divide/sqrt dependency chains interleave.  Per particle the
// attractors are still applied in index order, which is all the arithmetic depends on.
template <int FORCE, int N>
__device__ __forceinline__ void step(float (&px)[N], float (&py)[N], float (&pz)[N],
                                     float (&vx)[N], float (&vy)[N], float (&vz)[N],
                                     const psim_sim_params &p, const psim_attractor *alist,
                                     uint32_t na)
{
    const float dt = p.delta_time;
    apply_gravity<N>(vy, p);
    // The in-range threshold of each attractor rides in a padding word of the kernel's own
    // copy of the record (SimParams::_padding2, psim_attractor::_pad[0]); see with_threshold().
    if constexpr ((FORCE == 5 || (FORCE == 1 && PSIM_FIRST_HIT_SINGLE)) && N == 4) {   // first hits at once
        attract4_first_hits(vx, vy, vz, px, py, pz, p.mouse_position[0], p.mouse_position[1],
                        p.mouse_position[2], p.mouse_force, p.mouse_radius * 2.0f,
                        __uint_as_float(p._padding2), dt);
    } else if constexpr (FORCE == 1 || FORCE == 5) {
        const float two_r = p.mouse_radius * 2.0f;
        const float thr = __uint_as_float(p._padding2);
#pragma unroll
        for (int j = 0; j < N; ++j)
            attract(vx[j], vy[j], vz[j], px[j], py[j], pz[j], p.mouse_position[0],
                    p.mouse_position[1], p.mouse_position[2], p.mouse_force, two_r, thr, dt);
    } else if constexpr (FORCE >= 2) {
        for (uint32_t a = 0; a < na; ++a) {
            const psim_attractor at = alist[a];
            if (at.active > 0) {
                const float two_r = at.radius * 2.0f;
                const float thr = __uint_as_float(at._pad[0]);
#pragma unroll
                for (int j = 0; j < N; ++j)
                    attract(vx[j], vy[j], vz[j], px[j], py[j], pz[j], at.position[0],
                            at.position[1], at.position[2], at.force, two_r, thr, dt);
            }
        }
    }
    integrate<N>(px, py, pz, vx, vy, vz, p);
}

// Colour of one particle for modes 1 and 2 (cpu.rs:93-105), from the damped
// velocity / the new position.
template <int CM>
__device__ __forceinline__ f32x4 colour(float px, float py, float pz, float vx, float vy,
                                        float vz, const psim_sim_params &p)
{
    f32x4 c;
    if constexpr (CM == 1) {
        const float speed = length3(vx, vy, vz);
        const float ns = fminf(speed / 5.0f, 1.0f);          // f32::min ignores NaN
        c = f32x4{ns, 0.5f - ns * 0.5f, 1.0f - ns, 1.0f};
    } else {
        const float d = length3(px, py, pz);
        float nd = d / fmaxf(p.max_dist_for_color, 0.01f);
        nd = nd < 0.0f ? 0.0f : nd;                          // f32::clamp keeps NaN
        nd = nd > 1.0f ? 1.0f : nd;
        c = f32x4{nd, 0.0f, 1.0f - nd, 1.0f};
    }
    return c;
}

struct KernargAttractors {
    psim_attractor a[PSIM_KERNARG_ATTRACTORS];
};

// ---- the update kernel -----------------------------------------------------
//
// CM    colour mode 0 (colour untouched, 48 B/particle), 1 or 2 (+16 B store).
// FORCE see step().
// MODE  PSIM_ACCESS_*: a state larger than the 256 MiB Infinity Cache is streamed once per
//       step, so keeping its lines cached only evicts useful ones (nt loads, nt sc1
//       stores); a state that fits the Infinity Cache is fastest with plain loads and sc1
//       stores; one that fits the L2s with plain accesses.
//
// A lane owns a group of 4 consecutive particles: one 16-byte access per array, so each
// wave instruction covers 1 KiB of consecutive addresses.  Grid-stride over groups; a
// launch whose grid covers every group runs the loop once (the fastest shape measured:
// workgroups are dispatched in index order, so the resident ones sweep a narrow window
// of each array).  The last n % 4 particles are stepped one at a time by workgroup 0.
//
// Colour (modes 1/2): a lane's 4 RGBA values are 64 consecutive bytes, so storing them
// straight from the lane would write 16 B at a 64 B stride per instruction (measured:
// 2.4x slower with non-temporal stores, 15 % slower with plain ones).  Instead the wave
// transposes them through LDS and stores 1 KiB of consecutive bytes per instruction.
// LDS image per wave: [j][lane] with rows of 68 quads; the 4 writes are lane-linear and
// the reads (lane L, instruction k -> quad [L%4][16k + L/4]) touch 16 distinct 16-byte
// bank slots per 16-lane group, so neither side has a bank conflict.
#define PSIM_COLOUR_ROW 68   /* quads per LDS row: 64 + 4 of padding */

// Stores the wave's 256 colour quads (`col` of every lane with `live`) to the colour array
// of groups g0 .. g0+63 through the wave's LDS tile `t`, 1 KiB of consecutive bytes per
// store instruction.  Must be reached by the whole wave.
// The second half: the whole wave stores what its lanes have written into the tile
// (t[j][lane] = colour of the lane's particle j).
template <int MODE>
__device__ __forceinline__ void store_staged_colours(f32x4 (*t)[PSIM_COLOUR_ROW], f32x4 *col4,
                                                     uint32_t g0, uint32_t lane, uint32_t ngroups)
{
    // LDS operations of one wave execute in order; the fences only pin the compiler's
    // ordering of the lanes' writes against the reads below.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t q = k * 64u + lane;                        // quad within the wave's 4 KiB
        const uint32_t src = q >> 2;                              // lane that computed it
        if constexpr (MODE == PSIM_ACCESS_STREAM) {
            st_policy<PSIM_POLICY_NT_SC1>(col4, 4ull * g0 + 64u * k, 4ull * ngroups, lane,
                                          t[q & 3u][src]);
        } else if constexpr (MODE == PSIM_ACCESS_RESIDENT) {
            st_policy<PSIM_POLICY_SC1>(col4, 4ull * g0 + 64u * k, 4ull * ngroups, lane,
                                       t[q & 3u][src]);
        } else {
            if (g0 + src < ngroups) col4[4ull * g0 + q] = t[q & 3u][src];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                              // a later call rewrites the tile
}

template <int MODE>
__device__ __forceinline__ void store_colours_transposed(f32x4 (*t)[PSIM_COLOUR_ROW],
                                                         f32x4 *col4, uint32_t g0, uint32_t lane,
                                                         bool live, uint32_t ngroups,
                                                         const f32x4 (&col)[4])
{
    if (live) {
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j][lane] = col[j];
    }
    store_staged_colours<MODE>(t, col4, g0, lane, ngroups);
}

template <int CM, int FORCE, int MODE>
__global__ void __launch_bounds__(PSIM_UPDATE_BLOCK) update_kernel(SoaPtrs s, uint32_t n,
                                                                  psim_sim_params p,
                                                                  const psim_attractor *attr_dev,
                                                                  uint32_t na,
                                                                  KernargAttractors ka)
{
    constexpr bool NT = MODE == PSIM_ACCESS_STREAM;          // non-temporal loads
    const psim_attractor *alist = nullptr;
    [[maybe_unused]] f32x4 *lds_tests = nullptr;              // see attract_list_queued()
    if constexpr (FORCE == 4) {
        __shared__ f32x4 test_records[PSIM_MAX_ATTRACTORS];
        lds_tests = test_records;
    }
    if constexpr (FORCE == 2 || FORCE == 4) {
        // Stage the attractor list (<= 2 KiB) into LDS once per workgroup; the queued form
        // also takes a list that came in the kernel arguments (attr_dev == nullptr).
        __shared__ psim_attractor lds_attr[PSIM_MAX_ATTRACTORS];
        const uint32_t words = na * (sizeof(psim_attractor) / 4u);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(attr_dev);
        if constexpr (FORCE == 4)
            if (!attr_dev) src = reinterpret_cast<const uint32_t *>(ka.a);
        for (uint32_t w = threadIdx.x; w < words; w += PSIM_UPDATE_BLOCK)
            reinterpret_cast<uint32_t *>(lds_attr)[w] = src[w];
        __syncthreads();
        static_assert(PSIM_MAX_ATTRACTORS <= PSIM_UPDATE_BLOCK, "one lane per attractor");
        if (threadIdx.x < na) {                               // one lane per attractor
            psim_attractor &at = lds_attr[threadIdx.x];
            const float thr = psim_in_range_threshold(at.radius * 2.0f);
            at._pad[0] = __float_as_uint(thr);
            if constexpr (FORCE == 4)
                lds_tests[threadIdx.x] = f32x4{at.position[0], at.position[1], at.position[2],
                                               at.active > 0 ? thr : 0.0f};
        }
        __syncthreads();
        alist = lds_attr;
    } else if constexpr (FORCE == 3) {
        alist = ka.a;
    }

    f32x4 *px4 = reinterpret_cast<f32x4 *>(s.px), *py4 = reinterpret_cast<f32x4 *>(s.py),
          *pz4 = reinterpret_cast<f32x4 *>(s.pz), *vx4 = reinterpret_cast<f32x4 *>(s.vx),
          *vy4 = reinterpret_cast<f32x4 *>(s.vy), *vz4 = reinterpret_cast<f32x4 *>(s.vz);
    f32x4 *col4 = reinterpret_cast<f32x4 *>(s.color);

    const uint32_t ngroups = n >> 2;
    const uint32_t stride = gridDim.x * PSIM_UPDATE_BLOCK;   // groups per grid pass
    const uint32_t lane = threadIdx.x & 63u;
    // The trip count is the same for every lane of a wave (the colour transposition
    // needs the whole wave); lanes past the end are masked by `live`.
    for (uint32_t g0 = blockIdx.x * PSIM_UPDATE_BLOCK + (threadIdx.x & ~63u); g0 < ngroups;
         g0 += stride) {
        const uint32_t g = g0 + lane;
        const bool live = g < ngroups;
        f32x4 col[4];
        if constexpr (FORCE == 4) {
            // The queued list needs the whole wave in step: lanes past the end carry zeros.
            __shared__ f32x4 queue[PSIM_UPDATE_BLOCK / 64][PSIM_PAIR_QUEUE];
            f32x4 x = {}, y = {}, z = {}, u = {}, v = {}, w = {};
            if (live) {
                x = ld<NT>(px4 + g), y = ld<NT>(py4 + g), z = ld<NT>(pz4 + g);
                u = ld<NT>(vx4 + g), v = ld<NT>(vy4 + g), w = ld<NT>(vz4 + g);
            }
            float a[4], b[4], c[4], d[4], e[4], f[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = x[j]; b[j] = y[j]; c[j] = z[j];
                d[j] = u[j]; e[j] = v[j]; f[j] = w[j];
            }
            apply_gravity<4>(e, p);
            attract_list_queued(a, b, c, d, e, f, alist, lds_tests, na, p.delta_time,
                                queue[threadIdx.x >> 6], lane, live);
            integrate<4>(a, b, c, d, e, f, p);
            if (live) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    x[j] = a[j]; y[j] = b[j]; z[j] = c[j];
                    u[j] = d[j]; v[j] = e[j]; w[j] = f[j];
                }
                st_state<MODE>(px4, g0, ngroups, lane, x); st_state<MODE>(py4, g0, ngroups, lane, y);
                st_state<MODE>(pz4, g0, ngroups, lane, z); st_state<MODE>(vx4, g0, ngroups, lane, u);
                st_state<MODE>(vy4, g0, ngroups, lane, v); st_state<MODE>(vz4, g0, ngroups, lane, w);
                if constexpr (CM != 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        col[j] = colour<CM>(a[j], b[j], c[j], d[j], e[j], f[j], p);
                }
            }
        } else if (live) {
            f32x4 x = ld<NT>(px4 + g), y = ld<NT>(py4 + g), z = ld<NT>(pz4 + g);
            f32x4 u = ld<NT>(vx4 + g), v = ld<NT>(vy4 + g), w = ld<NT>(vz4 + g);
            float a[4], b[4], c[4], d[4], e[4], f[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = x[j]; b[j] = y[j]; c[j] = z[j];
                d[j] = u[j]; e[j] = v[j]; f[j] = w[j];
            }
            step<FORCE, 4>(a, b, c, d, e, f, p, alist, na);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                x[j] = a[j]; y[j] = b[j]; z[j] = c[j];
                u[j] = d[j]; v[j] = e[j]; w[j] = f[j];
            }
            st_state<MODE>(px4, g0, ngroups, lane, x); st_state<MODE>(py4, g0, ngroups, lane, y);
            st_state<MODE>(pz4, g0, ngroups, lane, z); st_state<MODE>(vx4, g0, ngroups, lane, u);
            st_state<MODE>(vy4, g0, ngroups, lane, v); st_state<MODE>(vz4, g0, ngroups, lane, w);
            if constexpr (CM != 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    col[j] = colour<CM>(a[j], b[j], c[j], d[j], e[j], f[j], p);
            }
        }
        if constexpr (CM != 0) {
            __shared__ f32x4 tile[PSIM_UPDATE_BLOCK / 64][4][PSIM_COLOUR_ROW];
            store_colours_transposed<MODE>(tile[threadIdx.x >> 6], col4, g0, lane, live, ngroups,
                                           col);
        }
    }

    // Tail: particles 4*ngroups .. n-1.
    if (blockIdx.x == 0 && threadIdx.x < (n & 3u)) {
        const uint32_t i = (ngroups << 2) + threadIdx.x;
        float x[1] = {s.px[i]}, y[1] = {s.py[i]}, z[1] = {s.pz[i]};
        float u[1] = {s.vx[i]}, v[1] = {s.vy[i]}, w[1] = {s.vz[i]};
        step<FORCE, 1>(x, y, z, u, v, w, p, alist, na);
        s.px[i] = x[0]; s.py[i] = y[0]; s.pz[i] = z[0];
        s.vx[i] = u[0]; s.vy[i] = v[0]; s.vz[i] = w[0];
        if constexpr (CM != 0) col4[i] = colour<CM>(x[0], y[0], z[0], u[0], v[0], w[0], p);
    }
}

// One particle per lane, 4-byte accesses: for caller-owned arrays that are not
// 16-byte aligned (psim_update_soa only).
template <int CM, int FORCE>
__global__ void __launch_bounds__(PSIM_BLOCK) update_kernel_scalar(SoaPtrs s, uint32_t n,
                                                                  psim_sim_params p)
{
    const uint64_t stride = (uint64_t)gridDim.x * PSIM_BLOCK;    // 64-bit: i + stride may pass 2^32
    for (uint64_t i = (uint64_t)blockIdx.x * PSIM_BLOCK + threadIdx.x; i < n; i += stride) {
        float x[1] = {s.px[i]}, y[1] = {s.py[i]}, z[1] = {s.pz[i]};
        float u[1] = {s.vx[i]}, v[1] = {s.vy[i]}, w[1] = {s.vz[i]};
        step<FORCE, 1>(x, y, z, u, v, w, p, nullptr, 0);
        s.px[i] = x[0]; s.py[i] = y[0]; s.pz[i] = z[0];
        s.vx[i] = u[0]; s.vy[i] = v[0]; s.vz[i] = w[0];
        if constexpr (CM != 0) {
            const f32x4 c = colour<CM>(x[0], y[0], z[0], u[0], v[0], w[0], p);
            float *dst = s.color + 4 * i;                    // may be only 4-byte aligned
            dst[0] = c[0]; dst[1] = c[1]; dst[2] = c[2]; dst[3] = c[3];
        }
    }
}

// ---- several steps in one pass ----------------------------------------------
//
// Particles never read other particles, so k consecutive steps of one particle can be
// taken back to back in registers: the state crosses HBM once instead of k times and the
// kernel turns from HBM-bound into VALU-bound (~35-60 instructions per particle and step).
// The k parameter sets travel in the kernel arguments (k <= PSIM_BATCH_STEPS = 64 fits the
// 4 KiB argument segment; they are wave-uniform, so they are read with scalar loads).
// The arithmetic of each step is exactly that of update_kernel, in the same order, so the
// result equals k separate launches bit for bit.  Colour is written once, from the state
// after step `colour_step` (the last step whose colour mode is 1 or 2; -1 = none): later
// steps with mode 0 keep it (src/simulation/cpu.rs:106), earlier ones are overwritten.
struct BatchParams {
    psim_sim_params p[PSIM_BATCH_STEPS];
};

// `staged` (N == 4): the lane's row of the wave's colour tile; the colours of step
// `colour_step` go there at once instead of staying in 16 registers for the rest of the
// batch (62 instead of 78 registers: 8 instead of 6 waves per SIMD).  `col` (N == 1) takes
// the tail particle's.
template <int N>
__device__ __forceinline__ void step_batch(float (&a)[N], float (&b)[N], float (&c)[N],
                                           float (&d)[N], float (&e)[N], float (&f)[N],
                                           f32x4 (*staged)[PSIM_COLOUR_ROW], uint32_t lane,
                                           f32x4 *col, const BatchParams &bp, uint32_t nsteps,
                                           int colour_step)
{
    for (uint32_t k = 0; k < nsteps; ++k) {
        const psim_sim_params &p = bp.p[k];
        if (p.is_mouse_dragging > 0) step<PSIM_FIRST_HIT ? 5 : 1, N>(a, b, c, d, e, f, p, nullptr, 0);
        else step<0, N>(a, b, c, d, e, f, p, nullptr, 0);
        if ((int)k == colour_step) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const f32x4 v = p.color_mode == 1 ? colour<1>(a[j], b[j], c[j], d[j], e[j], f[j], p)
                                                  : colour<2>(a[j], b[j], c[j], d[j], e[j], f[j], p);
                if constexpr (N == 4) staged[j][lane] = v;
                else col[j] = v;
            }
        }
    }
}

template <int MODE>
__global__ void __launch_bounds__(PSIM_UPDATE_BLOCK) update_batch_kernel(SoaPtrs s, uint32_t n,
                                                                        BatchParams bp,
                                                                        uint32_t nsteps,
                                                                        int colour_step)
{
    constexpr bool NT = MODE == PSIM_ACCESS_STREAM;
    __shared__ f32x4 tile[PSIM_UPDATE_BLOCK / 64][4][PSIM_COLOUR_ROW];
    f32x4 *px4 = reinterpret_cast<f32x4 *>(s.px), *py4 = reinterpret_cast<f32x4 *>(s.py),
          *pz4 = reinterpret_cast<f32x4 *>(s.pz), *vx4 = reinterpret_cast<f32x4 *>(s.vx),
          *vy4 = reinterpret_cast<f32x4 *>(s.vy), *vz4 = reinterpret_cast<f32x4 *>(s.vz);
    f32x4 *col4 = reinterpret_cast<f32x4 *>(s.color);

    const uint32_t ngroups = n >> 2;
    const uint32_t stride = gridDim.x * PSIM_UPDATE_BLOCK;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t g0 = blockIdx.x * PSIM_UPDATE_BLOCK + (threadIdx.x & ~63u); g0 < ngroups;
         g0 += stride) {
        const uint32_t g = g0 + lane;
        const bool live = g < ngroups;
        if (live) {
            f32x4 x = ld<NT>(px4 + g), y = ld<NT>(py4 + g), z = ld<NT>(pz4 + g);
            f32x4 u = ld<NT>(vx4 + g), v = ld<NT>(vy4 + g), w = ld<NT>(vz4 + g);
            float a[4], b[4], c[4], d[4], e[4], f[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = x[j]; b[j] = y[j]; c[j] = z[j];
                d[j] = u[j]; e[j] = v[j]; f[j] = w[j];
            }
            step_batch<4>(a, b, c, d, e, f, tile[threadIdx.x >> 6], lane, nullptr, bp, nsteps,
                          colour_step);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                x[j] = a[j]; y[j] = b[j]; z[j] = c[j];
                u[j] = d[j]; v[j] = e[j]; w[j] = f[j];
            }
            st_state<MODE>(px4, g0, ngroups, lane, x); st_state<MODE>(py4, g0, ngroups, lane, y);
            st_state<MODE>(pz4, g0, ngroups, lane, z); st_state<MODE>(vx4, g0, ngroups, lane, u);
            st_state<MODE>(vy4, g0, ngroups, lane, v); st_state<MODE>(vz4, g0, ngroups, lane, w);
        }
        if (colour_step >= 0)                                     // uniform over the grid
            store_staged_colours<MODE>(tile[threadIdx.x >> 6], col4, g0, lane, ngroups);
    }

    // Tail: particles 4*ngroups .. n-1.
    if (blockIdx.x == 0 && threadIdx.x < (n & 3u)) {
        const uint32_t i = (ngroups << 2) + threadIdx.x;
        float x[1] = {s.px[i]}, y[1] = {s.py[i]}, z[1] = {s.pz[i]};
        float u[1] = {s.vx[i]}, v[1] = {s.vy[i]}, w[1] = {s.vz[i]};
        f32x4 col[1];
        step_batch<1>(x, y, z, u, v, w, nullptr, 0u, col, bp, nsteps, colour_step);
        s.px[i] = x[0]; s.py[i] = y[0]; s.pz[i] = z[0];
        s.vx[i] = u[0]; s.vy[i] = v[0]; s.vz[i] = w[0];
        if (colour_step >= 0) col4[i] = col[0];
    }
}

hipError_t launch_update_batch(const SoaPtrs &s, uint32_t n, const psim_sim_params *params,
                               uint32_t nsteps, const LaunchTuning &t, hipStream_t stream)
{
    if (n == 0 || nsteps == 0) return hipSuccess;
    BatchParams bp;
    memset(&bp, 0, sizeof(bp));
    memcpy(bp.p, params, nsteps * sizeof(psim_sim_params));
    for (uint32_t k = 0; k < nsteps; ++k) {                  // see step(): threshold per step
        const float thr = psim_in_range_threshold(params[k].mouse_radius * 2.0f);
        memcpy(&bp.p[k]._padding2, &thr, sizeof(thr));
    }
    int colour_step = -1;
    for (uint32_t k = 0; k < nsteps; ++k)
        if (params[k].color_mode == 1 || params[k].color_mode == 2) colour_step = (int)k;
    const uint32_t grid = update_grid(n, t);
    constexpr uint32_t used = (PSIM_UPDATE_BLOCK / 64) * 4 * PSIM_COLOUR_ROW * 16;
    const uint32_t extra = t.reserve_lds > used ? t.reserve_lds - used : 0;
    switch (t.access) {
    case PSIM_ACCESS_STREAM:
        hipLaunchKernelGGL(update_batch_kernel<PSIM_ACCESS_STREAM>, dim3(grid),
                           dim3(PSIM_UPDATE_BLOCK), extra, stream, s, n, bp, nsteps, colour_step);
        break;
    case PSIM_ACCESS_RESIDENT:
        hipLaunchKernelGGL(update_batch_kernel<PSIM_ACCESS_RESIDENT>, dim3(grid),
                           dim3(PSIM_UPDATE_BLOCK), extra, stream, s, n, bp, nsteps, colour_step);
        break;
    default:
        hipLaunchKernelGGL(update_batch_kernel<PSIM_ACCESS_PLAIN>, dim3(grid),
                           dim3(PSIM_UPDATE_BLOCK), extra, stream, s, n, bp, nsteps, colour_step);
    }
    return hipGetLastError();
}

// ---- launch dispatch -------------------------------------------------------

// The kernels' own copy of a SimParams / attractor record carries the in-range threshold of
// its attractor in a padding word (the caller's records are never modified).
static psim_sim_params with_threshold(const psim_sim_params &p)
{
    psim_sim_params q = p;
    const float thr = psim_in_range_threshold(p.mouse_radius * 2.0f);
    memcpy(&q._padding2, &thr, sizeof(thr));
    return q;
}

static psim_attractor with_threshold(const psim_attractor &a)
{
    psim_attractor q = a;
    const float thr = psim_in_range_threshold(a.radius * 2.0f);
    memcpy(&q._pad[0], &thr, sizeof(thr));
    return q;
}

// Every instantiation of update_kernel has the same signature, so a launch is "pick the
// instantiation and its dynamic LDS, then launch through the pointer"; the occupancy query
// (update_occupancy) picks the same way.
using UpdateKernel = void (*)(SoaPtrs, uint32_t, psim_sim_params, const psim_attractor *, uint32_t,
                              KernargAttractors);

struct UpdateLaunch {
    UpdateKernel kernel;
    uint32_t dynamic_lds;
};

template <int CM, int FORCE>
static UpdateLaunch pick_nt(const LaunchTuning &t)
{
    // t.reserve_lds bytes of dynamic LDS are requested but never touched: they only
    // limit how many workgroups a CU holds at once (see choose_tuning()).
    // The reservation counts the LDS the kernel itself declares.
    constexpr uint32_t used = (CM != 0 ? (PSIM_UPDATE_BLOCK / 64) * 4 * PSIM_COLOUR_ROW * 16 : 0) +
                              (FORCE == 2 || FORCE == 4 ? PSIM_MAX_ATTRACTORS * sizeof(psim_attractor) : 0) +
                              (FORCE == 4 ? (PSIM_UPDATE_BLOCK / 64) * PSIM_PAIR_QUEUE * 16 +
                                                PSIM_MAX_ATTRACTORS * 16 : 0);
    uint32_t want = CM != 0 ? t.reserve_lds_colour : t.reserve_lds;
    // A list of one attractor in the kernel arguments carries the loop of the list kernels:
    // 7 workgroups per CU (0.741 ms at 100 M particles) rather than the 6 of the plain
    // kernel (0.783 ms).  The kernels that stage a list into LDS start with a barrier and
    // want at least 13 (one attractor, 100 M particles: 0.751 ms against 0.951 with 7;
    // 6.25 M: 43 us against 49).
    if (FORCE == 3 && want == 26624u) want = 23040u;
    if ((FORCE == 2 || FORCE == 4) && want > 11776u) want = 11776u;
    const uint32_t extra = want > used ? want - used : 0;
    switch (t.access) {
    case PSIM_ACCESS_STREAM: return {update_kernel<CM, FORCE, PSIM_ACCESS_STREAM>, extra};
    case PSIM_ACCESS_RESIDENT: return {update_kernel<CM, FORCE, PSIM_ACCESS_RESIDENT>, extra};
    default: return {update_kernel<CM, FORCE, PSIM_ACCESS_PLAIN>, extra};
    }
}

template <int FORCE>
static UpdateLaunch pick_cm(uint32_t color_mode, const LaunchTuning &t)
{
    switch (color_mode) {
    case 1: return pick_nt<1, FORCE>(t);
    case 2: return pick_nt<2, FORCE>(t);
    default: return pick_nt<0, FORCE>(t);
    }
}

// The host list length from which the queued form is used (KNOB_QUEUE_FROM: 0 = never).
static uint32_t queue_from()
{
    const int x = knob(KNOB_QUEUE_FROM);
    return x < 0 ? 2u : (x == 0 ? 0xFFFFFFFFu : (uint32_t)x);
}

// force as launch_update() takes it (0 none, 1 SimParams, 2 device list, 3 host list).
static UpdateLaunch pick_update(uint32_t color_mode, int force, uint32_t na, const LaunchTuning &t)
{
    // Lists of two attractors and more go through the queue of in-range pairs (100 M
    // particles, r = 10, ms per step with / without: 2 attractors in device memory 0.759 /
    // 0.811, 4: 0.751 / 0.858, 8: 0.768 / 1.00, 16: 0.899 / 1.96, 64: 2.40 / 7.59; lists that
    // came in the kernel arguments: 2-4 attractors equal, 8: 0.761 / 0.947).
    // A list in device memory is staged into LDS anyway and takes the queue from one
    // attractor on.
    const uint32_t from = queue_from();
    if (force >= 2 && na >= 1 && from != 0xFFFFFFFFu && (force == 2 || na >= from))
        return pick_cm<4>(color_mode, t);
    switch (force) {
    case 0: return pick_cm<0>(color_mode, t);
    case 1: return pick_cm<1>(color_mode, t);
    case 2: return pick_cm<2>(color_mode, t);
    default: return pick_cm<3>(color_mode, t);
    }
}

uint32_t update_grid(uint32_t n, const LaunchTuning &t)
{
    const uint32_t ngroups = n >> 2;
    uint32_t need = (ngroups + PSIM_UPDATE_BLOCK - 1u) / PSIM_UPDATE_BLOCK;
    if (need == 0) need = 1;                                  // the tail still needs block 0
    if (t.max_blocks > 0 && need > t.max_blocks) need = t.max_blocks;
    return need;
}

// Launch shape, from measurements on MI355X (GB/s of the 48 B/particle stream, 100 M
// particles unless noted).  Grid: one pass over the array beats a capped grid-stride grid
// (2048-8192 workgroups: 4.3-4.9 k).  Accesses: plain 5.36 k, non-temporal 5.53 k.
// Resident waves per CU, non-temporal, by workgroup size and LDS reservation:
//   256 threads: 32 waves (no cap) 5.5 k | 12 waves 5.99 k | 8 waves 6.22 k | 4 waves 5.90 k
//   128 threads: 10 waves 6.26 k | 6 waves 6.23-6.30 k | 4 waves 5.34 k
//    64 threads: 10 waves 6.35 k | 7 waves 6.39 k | 6 waves 6.34 k | 5 waves 5.35 k
// Fewer resident waves keep the twelve read/write streams inside a narrower address
// window, which HBM serves better, until too few bytes are in flight; single-wave
// workgroups give the finest control, so the update kernel runs 64-thread workgroups and
// reserves (never touches) LDS per workgroup: 23040 B lets 7 of them fit a CU's 160 KiB,
// 26624 B lets 6.  (Those figures are for non-temporal global stores; the nt sc1 buffer
// stores of st_stream() add 2-3 % at every occupancy and move the best point from 7 to 6.)
// By size (64 threads): 1 M particles plain without cap 7.3 k (cap 6.2 k); 6.25-14 M plain
// + cap 6.6-7.0 k (the state fits the 256 MiB Infinity Cache); from 20 M on
// streaming + cap 6.6 k (plain + cap 5.9 k).
//
// Attractor lists add ~80 VALU instructions per particle and attractor, in long
// divide/sqrt dependency chains that a few waves per SIMD cannot hide.  At 100 M
// particles (256-thread workgroups), ms per step with / without an 8-wave cap:
// 1-2 attractors 0.74-0.76 / 0.87-0.88, 4 attractors 0.88 / 0.87, 8 attractors 1.47 / 0.96.
LaunchTuning choose_tuning(uint32_t n, uint32_t n_attractors)
{
    const uint64_t state_bytes = 24ull * n;
    LaunchTuning t;
    t.max_blocks = 0;
    // By state size (GB/s of the 48 B stream, best access mode of plain / resident / stream):
    //   4 M particles 6.57 / 6.86 / -      6.25 M 6.92 / 7.06 / -     10 M 7.08 / 7.20 / 6.3
    //   11 M 7.12 / 7.27 / 6.51            12.5 M 6.57 / 6.07 / 6.14  14 M 6.61 / 6.24 / 6.54
    //   16 M 6.36 / 6.28 / 6.53            18 M 6.17 / 6.00 / 6.53    100 M 5.9 / - / 6.65
    // i.e. sc1 stores while the state fits the 256 MiB Infinity Cache, plain accesses in the
    // narrow band just above it, streaming from ~360 MB on.
    const bool stream = state_bytes > 360000000ull;
    const bool resident = state_bytes >= 96000000ull && state_bytes <= 268000000ull;
    t.access = stream ? PSIM_ACCESS_STREAM : (resident ? PSIM_ACCESS_RESIDENT : PSIM_ACCESS_PLAIN);
    const bool cap = state_bytes >= 96000000ull && n_attractors < 4;
    // Streaming (nt loads, nt sc1 stores): 6 workgroups per CU, 6.64-6.68 k against 6.54-6.58 k
    // for 7 at 100 M particles; the same holds with the colour store as a thirteenth stream
    // and at 500 M / 1 G particles.  Resident: 6 and 7 are equal (4-11 M particles).  Plain
    // band: 7 (6.57-6.61 k against 6.28-6.48 k for 6).
    t.reserve_lds = cap ? ((stream || resident) ? 26624u : 23040u) : 0u;
    t.reserve_lds_colour = t.reserve_lds;
    // A list of 2-4 attractors (worked off through the pair queue): 13 workgroups per CU
    // (ms per step at 100 M particles, 2 / 4 / 5 attractors: 13 resident 0.759 / 0.758 /
    // 0.760, 7 resident 1.01 / 1.06 / 1.07, 10 resident 0.843 / 0.864 / 0.877, uncapped
    // 0.759 / 0.751 / 0.794); 5-8 attractors 16 workgroups per CU, from 9 on no cap
    // (8 / 16 / 64 attractors: 0.768 / 0.899 / 2.40 uncapped, 0.830 / 1.14 / 3.25 with 13
    // resident).
    // States that do not stream run lists uncapped (6.25 M particles, 2 / 8 / 64 attractors,
    // us per step: 44 / 48 / 178 uncapped, 46 / 54 / 228 with 13 resident; 8 attractors
    // without the queue: 74).
    // Finer, streaming, ms per step at 13 / 16 / 20 resident / uncapped: 5 attractors
    // 0.767 / 0.757 / 0.762 / 0.765, 8: 0.810 / 0.754 / 0.763 / 0.781, 10: 0.883 / 0.789 /
    // 0.775 / 0.767, 16: 1.06 / 0.911 / 0.880 / 0.851.
    if (n_attractors >= 2)
        t.reserve_lds = t.reserve_lds_colour =
            !stream ? 0u : (n_attractors <= 4 ? 11776u : (n_attractors <= 8 ? 10240u : 0u));
    return t;
}

hipError_t launch_update(const SoaPtrs &s, uint32_t n, const psim_sim_params &params, int force,
                         const psim_attractor *attr_dev, const psim_attractor *attr_host,
                         uint32_t na, const LaunchTuning &t, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const psim_sim_params p = with_threshold(params);
    KernargAttractors ka;
    memset(&ka, 0, sizeof(ka));
    if (force == 3)
        for (uint32_t a = 0; a < na; ++a) ka.a[a] = with_threshold(attr_host[a]);
    const UpdateLaunch l = pick_update(p.color_mode, force, na, t);
    hipLaunchKernelGGL(l.kernel, dim3(update_grid(n, t)), dim3(PSIM_UPDATE_BLOCK), l.dynamic_lds,
                       stream, s, n, p, force == 2 ? attr_dev : nullptr, force >= 2 ? na : 0u, ka);
    return hipGetLastError();
}

// How many workgroups of the update psim_update / psim_update_multi would launch the
// runtime says one CU holds at once, dynamic LDS reservation included.
hipError_t update_occupancy(uint32_t color_mode, int force, uint32_t na, const LaunchTuning &t,
                            int *workgroups_per_cu, uint32_t *lds_bytes)
{
    const UpdateLaunch l = pick_update(color_mode, force, na, t);
    hipFuncAttributes attr;
    const hipError_t e = hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(l.kernel));
    if (e != hipSuccess) return e;
    *lds_bytes = (uint32_t)attr.sharedSizeBytes + l.dynamic_lds;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(
        workgroups_per_cu, reinterpret_cast<const void *>(l.kernel), PSIM_UPDATE_BLOCK,
        l.dynamic_lds);
}

hipError_t launch_update_scalar(const SoaPtrs &s, uint32_t n, const psim_sim_params &params,
                                uint32_t max_blocks, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const psim_sim_params p = with_threshold(params);
    uint32_t grid = (n + PSIM_BLOCK - 1u) / PSIM_BLOCK;
    if (max_blocks > 0 && grid > max_blocks) grid = max_blocks;
    const bool drag = p.is_mouse_dragging > 0;
#define PSIM_SCALAR(CM, F)                                                                      \
    hipLaunchKernelGGL((update_kernel_scalar<CM, F>), dim3(grid), dim3(PSIM_BLOCK), 0, stream, \
                       s, n, p)
    switch (p.color_mode) {
    case 1: if (drag) PSIM_SCALAR(1, 1); else PSIM_SCALAR(1, 0); break;
    case 2: if (drag) PSIM_SCALAR(2, 1); else PSIM_SCALAR(2, 0); break;
    default: if (drag) PSIM_SCALAR(0, 1); else PSIM_SCALAR(0, 0); break;
    }
#undef PSIM_SCALAR
    return hipGetLastError();
}

// ---- attractor staging -----------------------------------------------------

__global__ void store_attractor_kernel(psim_attractor *dst, psim_attractor value)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *dst = value;
}

hipError_t launch_store_attractor(psim_attractor *dst, const psim_attractor &value,
                                  hipStream_t stream)
{
    hipLaunchKernelGGL(store_attractor_kernel, dim3(1), dim3(64), 0, stream, dst, value);
    return hipGetLastError();
}

// ---- attractor list in a static device array (the entry point that allocates nothing) ----
//
// `scratch` is the device-side staging of a host attractor list for callers that may not
// allocate: 64 records = 2 KiB, held by every GPU the library runs on until the process
// ends.  In host code the array's name is only a stand-in; the address each GPU holds its
// copy at comes from hipGetSymbolAddress, once per device, and that pointer is what every
// kernel argument below uses.
static __device__ __align__(256) psim_attractor scratch[PSIM_MAX_ATTRACTORS];

#define PSIM_MAX_DEVICES 64
static psim_attractor *g_static_list[PSIM_MAX_DEVICES];   // by device ordinal, null = not asked yet
static std::mutex g_static_list_lock;

hipError_t static_list_address(int device, psim_attractor **out)
{
    *out = nullptr;
    if (device < 0 || device >= PSIM_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> hold(g_static_list_lock);
    if (!g_static_list[device]) {
        int current = -1;
        hipError_t e = hipGetDevice(&current);
        if (e != hipSuccess) return e;
        if (current != device) return hipErrorInvalidDevice;   // the caller makes it current
        void *ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_static_list[device] = static_cast<psim_attractor *>(ptr);
    }
    *out = g_static_list[device];
    return hipSuccess;
}

// Up to PSIM_KERNARG_ATTRACTORS records travel in the kernel arguments; one lane per word.
__global__ void store_attractors_kernel(psim_attractor *dst, KernargAttractors ka, uint32_t count)
{
    constexpr uint32_t kWords = sizeof(psim_attractor) / 4u;
    static_assert(PSIM_KERNARG_ATTRACTORS * kWords <= 64u, "one wave covers a chunk");
    const uint32_t w = threadIdx.x;
    if (blockIdx.x == 0 && w < count * kWords)
        reinterpret_cast<uint32_t *>(dst)[w] = reinterpret_cast<const uint32_t *>(ka.a)[w];
}

hipError_t launch_store_attractors(psim_attractor *dst, const psim_attractor *values_host,
                                   uint32_t count, hipStream_t stream)
{
    for (uint32_t done = 0; done < count; done += PSIM_KERNARG_ATTRACTORS) {
        const uint32_t m = (count - done) < PSIM_KERNARG_ATTRACTORS ? (count - done)
                                                                    : PSIM_KERNARG_ATTRACTORS;
        KernargAttractors ka;
        memset(&ka, 0, sizeof(ka));
        memcpy(ka.a, values_host + done, m * sizeof(psim_attractor));
        hipLaunchKernelGGL(store_attractors_kernel, dim3(1), dim3(64), 0, stream, dst + done, ka, m);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ---- synthetic uniform state -----------------------------------------------

// Counter-based SplitMix64: the value depends only on (seed, counter).
__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t counter)
{
    uint64_t z = seed + (counter + 1u) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ float unit_at(uint64_t seed, uint64_t counter)
{
    return (float)(uint32_t)(splitmix64_at(seed, counter) >> 40) * 0x1.0p-24f;
}

// Particle with global index g: position U[-50,50)^3, velocity U[-1,1)^3, colour
// (pos/50 + 1) * 0.5 with alpha 1 (the reference's colour rule,
// src/simulation/mod.rs:162-163), paddings 0.  One particle per lane.
template <bool NT>
__global__ void __launch_bounds__(PSIM_BLOCK) gen_uniform_kernel(SoaPtrs s, uint32_t n,
                                                                uint64_t global_first,
                                                                uint64_t seed)
{
    const uint64_t stride = (uint64_t)gridDim.x * PSIM_BLOCK;
    for (uint64_t i = (uint64_t)blockIdx.x * PSIM_BLOCK + threadIdx.x; i < n; i += stride) {
        const uint64_t c0 = 6ull * (global_first + i);
        float pos[3], vel[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            pos[c] = unit_at(seed, c0 + c) * 100.0f - 50.0f;
            vel[c] = unit_at(seed, c0 + 3 + c) * 2.0f - 1.0f;
        }
        const f32x4 col = {(pos[0] / 50.0f + 1.0f) * 0.5f, (pos[1] / 50.0f + 1.0f) * 0.5f,
                           (pos[2] / 50.0f + 1.0f) * 0.5f, 1.0f};
        if constexpr (NT) {
            const uint32_t lane = threadIdx.x & 63u;
            const uint64_t w0 = i - lane;                        // the wave's first particle
            st_stream(s.px, w0, n, lane, pos[0]); st_stream(s.py, w0, n, lane, pos[1]);
            st_stream(s.pz, w0, n, lane, pos[2]); st_stream(s.vx, w0, n, lane, vel[0]);
            st_stream(s.vy, w0, n, lane, vel[1]); st_stream(s.vz, w0, n, lane, vel[2]);
            st_stream(reinterpret_cast<f32x4 *>(s.color), w0, n, lane, col);
            st_stream(reinterpret_cast<f32x4 *>(s.initial_color), w0, n, lane, col);
            st_stream(reinterpret_cast<f32x2 *>(s.padding), w0, n, lane, f32x2{0.0f, 0.0f});
        } else {
            st<NT>(s.px + i, pos[0]); st<NT>(s.py + i, pos[1]); st<NT>(s.pz + i, pos[2]);
            st<NT>(s.vx + i, vel[0]); st<NT>(s.vy + i, vel[1]); st<NT>(s.vz + i, vel[2]);
            st<NT>(reinterpret_cast<f32x4 *>(s.color) + i, col);
            st<NT>(reinterpret_cast<f32x4 *>(s.initial_color) + i, col);
            st<NT>(reinterpret_cast<f32x2 *>(s.padding) + i, f32x2{0.0f, 0.0f});
        }
    }
}

// One pass over the particles (like the update, faster than a capped grid-stride grid);
// non-temporal stores and at most 3 resident workgroups per CU for states that do not fit
// the Infinity Cache.
hipError_t launch_gen_uniform(const SoaPtrs &s, uint32_t n, uint64_t global_first, uint64_t seed,
                              hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const uint32_t grid = (n + PSIM_BLOCK - 1u) / PSIM_BLOCK;
    if ((uint64_t)n * 64u > 400000000ull)
        hipLaunchKernelGGL(gen_uniform_kernel<true>, dim3(grid), dim3(PSIM_BLOCK), 53248, stream,
                           s, n, global_first, seed);
    else
        hipLaunchKernelGGL(gen_uniform_kernel<false>, dim3(grid), dim3(PSIM_BLOCK), 0, stream, s,
                           n, global_first, seed);
    return hipGetLastError();
}

// ---- AoS <-> SoA transposes ------------------------------------------------
//
// The 64-byte record (src/simulation/mod.rs:86-98) is four 16-byte quads:
//   q0 = position.xyz, padding1   q1 = velocity.xyz, padding2
//   q2 = color                    q3 = initial_color
// A workgroup of BLOCK threads (256, or one wave) moves BLOCK particles = BLOCK * 64 bytes
// of AoS.  The AoS side is
// accessed as a linear run of quads (lane t touches quad k*BLOCK + t, so a
// wave instruction covers 1 KiB of consecutive bytes); the transposition happens
// in LDS.  LDS holds quads at [field][particle]; the per-particle side is lane-linear,
// and the rows are padded so that the AoS-order side (lane t -> [t & 3][k*64 + t/4]) is
// free of bank conflicts too: by 4 quads for the 16-byte reads of the export (banks
// taken modulo 64 dwords), by 2 quads for the 16-byte writes of the import (modulo 32).

template <bool NT, int BLOCK>
__global__ void __launch_bounds__(BLOCK) export_aos_kernel(SoaPtrs s, uint32_t first,
                                                               uint32_t n, f32x4 *aos)
{
    __shared__ f32x4 tile[4][BLOCK + 4];
    const uint32_t base = blockIdx.x * BLOCK;            // first particle of the tile
    const uint32_t t = threadIdx.x;
    const uint32_t i = base + t;
    if (i < n) {
        const uint32_t src = first + i;
        const f32x2 pad = ld<NT>(reinterpret_cast<const f32x2 *>(s.padding) + src);
        tile[0][t] = f32x4{ld<NT>(s.px + src), ld<NT>(s.py + src), ld<NT>(s.pz + src), pad[0]};
        tile[1][t] = f32x4{ld<NT>(s.vx + src), ld<NT>(s.vy + src), ld<NT>(s.vz + src), pad[1]};
        tile[2][t] = ld<NT>(reinterpret_cast<const f32x4 *>(s.color) + src);
        tile[3][t] = ld<NT>(reinterpret_cast<const f32x4 *>(s.initial_color) + src);
    }
    __syncthreads();
    const uint32_t live = (n - base) < BLOCK ? (n - base) : BLOCK;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t q = k * BLOCK + t;                // quad index within the tile
        const uint32_t particle = q >> 2, field = q & 3u;
        if constexpr (NT) {
            st_stream(aos, 4ull * base + (q & ~63u), 4ull * n, t & 63u,
                      tile[field][particle < BLOCK ? particle : 0]);
        } else {
            if (particle < live) st<NT>(aos + 4ull * base + q, tile[field][particle]);
        }
    }
}

template <bool NT, int BLOCK>
__global__ void __launch_bounds__(BLOCK) import_aos_kernel(SoaPtrs s, uint32_t first,
                                                               uint32_t n, const f32x4 *aos)
{
    __shared__ f32x4 tile[4][BLOCK + 2];
    const uint32_t base = blockIdx.x * BLOCK;
    const uint32_t t = threadIdx.x;
    const uint32_t live = (n - base) < BLOCK ? (n - base) : BLOCK;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t q = k * BLOCK + t;
        const uint32_t particle = q >> 2, field = q & 3u;
        if (particle < live) tile[field][particle] = ld<NT>(aos + 4ull * base + q);
    }
    __syncthreads();
    const uint32_t i = base + t;
    if (i < n) {
        const uint32_t dst = first + i;
        const f32x4 q0 = tile[0][t], q1 = tile[1][t];
        if constexpr (NT) {
            const uint32_t lane = t & 63u;
            const uint64_t w0 = (uint64_t)dst - lane, end = (uint64_t)first + n;
            st_stream(s.px, w0, end, lane, q0[0]); st_stream(s.py, w0, end, lane, q0[1]);
            st_stream(s.pz, w0, end, lane, q0[2]); st_stream(s.vx, w0, end, lane, q1[0]);
            st_stream(s.vy, w0, end, lane, q1[1]); st_stream(s.vz, w0, end, lane, q1[2]);
            st_stream(reinterpret_cast<f32x2 *>(s.padding), w0, end, lane, f32x2{q0[3], q1[3]});
            st_stream(reinterpret_cast<f32x4 *>(s.color), w0, end, lane, tile[2][t]);
            st_stream(reinterpret_cast<f32x4 *>(s.initial_color), w0, end, lane, tile[3][t]);
        } else {
            st<NT>(s.px + dst, q0[0]); st<NT>(s.py + dst, q0[1]); st<NT>(s.pz + dst, q0[2]);
            st<NT>(s.vx + dst, q1[0]); st<NT>(s.vy + dst, q1[1]); st<NT>(s.vz + dst, q1[2]);
            st<NT>(reinterpret_cast<f32x2 *>(s.padding) + dst, f32x2{q0[3], q1[3]});
            st<NT>(reinterpret_cast<f32x4 *>(s.color) + dst, tile[2][t]);
            st<NT>(reinterpret_cast<f32x4 *>(s.initial_color) + dst, tile[3][t]);
        }
    }
}

// ---- the transposes with several particles per lane -------------------------------------
//
// Above, a lane owns one particle, so the SoA side moves 4 bytes per lane: 256 bytes per wave
// instruction on each of the six state arrays.  Here a single-wave workgroup moves
// P = 64 * PPL particles (PPL = 2 or 4) and a lane owns PPL consecutive ones, so that every
// access of either side covers 512 bytes or 1 KiB of consecutive addresses, like the update
// kernel's.  Positions, velocities and the two padding floats cross LDS as eight rows of
// floats, comp[c][particle]; the row length is 4 (mod 64) floats, which makes both the
// scattered 4-byte side (lane = quad of a record: banks 16 * field + particle) and the
// 16-byte side (lane = PPL consecutive particles) free of bank conflicts.  The two colours
// are 16-byte units on both sides and cross as quads, quad[field][particle].  Full tiles
// only: the launcher hands the last n % P particles to the one-particle-per-lane kernels.
template <int PPL> struct LaneVec;
template <> struct LaneVec<2> { typedef f32x2 type; };
template <> struct LaneVec<4> { typedef f32x4 type; };

template <bool NT, int PPL>
__global__ void __launch_bounds__(64) import_aos_wide_kernel(SoaPtrs s, uint32_t first,
                                                             const f32x4 *aos)
{
    typedef typename LaneVec<PPL>::type vec;
    constexpr int P = 64 * PPL, ROW = P + 4;
    __shared__ __align__(16) float comp[8][ROW];
    __shared__ f32x4 quad[2][P + 2];
    const uint32_t lane = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * P;          // tile's first particle, in the request
    const f32x4 *src = aos + 4 * base;
    f32x4 v[4 * PPL];
#pragma unroll
    for (int k = 0; k < 4 * PPL; ++k) v[k] = ld<NT>(src + k * 64 + lane);
#pragma unroll
    for (int k = 0; k < 4 * PPL; ++k) {
        const uint32_t q = k * 64 + lane, particle = q >> 2, field = q & 3u;
        if (field < 2) {
#pragma unroll
            for (int c = 0; c < 4; ++c) comp[4 * field + c][particle] = v[k][c];
        } else {
            quad[field - 2][particle] = v[k];
        }
    }
    wave_lds_sync();
    const uint64_t p0 = (uint64_t)first + base;              // in the handle's arrays
    const uint32_t g = lane * PPL;
    float *const arr[6] = {s.px, s.py, s.pz, s.vx, s.vy, s.vz};
    constexpr int row_of[6] = {0, 1, 2, 4, 5, 6};
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const vec x = *reinterpret_cast<const vec *>(&comp[row_of[a]][g]);
        if constexpr (NT) st_stream(reinterpret_cast<vec *>(arr[a]), p0 / PPL, p0 / PPL + 64, lane, x);
        else reinterpret_cast<vec *>(arr[a])[p0 / PPL + lane] = x;
    }
    // {padding1, padding2} pairs, two particles per 16 bytes
#pragma unroll
    for (int m = 0; m < PPL / 2; ++m) {
        const uint32_t u = m * 64 + lane;                     // 16-byte unit = particles 2u, 2u + 1
        const f32x2 a = *reinterpret_cast<const f32x2 *>(&comp[3][2 * u]);
        const f32x2 b = *reinterpret_cast<const f32x2 *>(&comp[7][2 * u]);
        const f32x4 pad = {a[0], b[0], a[1], b[1]};
        f32x4 *dst = reinterpret_cast<f32x4 *>(s.padding);
        if constexpr (NT) st_stream(dst, p0 / 2 + 64u * m, p0 / 2 + 64u * m + 64, lane, pad);
        else dst[p0 / 2 + u] = pad;
    }
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        f32x4 *col = reinterpret_cast<f32x4 *>(s.color), *ini = reinterpret_cast<f32x4 *>(s.initial_color);
        if constexpr (NT) {
            st_stream(col, p0 + 64u * k, p0 + 64u * k + 64, lane, quad[0][k * 64 + lane]);
            st_stream(ini, p0 + 64u * k, p0 + 64u * k + 64, lane, quad[1][k * 64 + lane]);
        } else {
            col[p0 + k * 64 + lane] = quad[0][k * 64 + lane];
            ini[p0 + k * 64 + lane] = quad[1][k * 64 + lane];
        }
    }
}

template <bool NT, int PPL>
__global__ void __launch_bounds__(64) export_aos_wide_kernel(SoaPtrs s, uint32_t first, f32x4 *aos)
{
    typedef typename LaneVec<PPL>::type vec;
    constexpr int P = 64 * PPL, ROW = P + 4;
    __shared__ __align__(16) float comp[8][ROW];
    __shared__ f32x4 quad[2][P + 4];
    const uint32_t lane = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * P;
    const uint64_t p0 = (uint64_t)first + base;
    const uint32_t g = lane * PPL;
    const float *const arr[6] = {s.px, s.py, s.pz, s.vx, s.vy, s.vz};
    constexpr int row_of[6] = {0, 1, 2, 4, 5, 6};
    vec x[6];
    f32x4 pad[PPL / 2], col[PPL], ini[PPL];
#pragma unroll
    for (int a = 0; a < 6; ++a) x[a] = ld<NT>(reinterpret_cast<const vec *>(arr[a]) + p0 / PPL + lane);
#pragma unroll
    for (int m = 0; m < PPL / 2; ++m)
        pad[m] = ld<NT>(reinterpret_cast<const f32x4 *>(s.padding) + p0 / 2 + m * 64 + lane);
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        col[k] = ld<NT>(reinterpret_cast<const f32x4 *>(s.color) + p0 + k * 64 + lane);
        ini[k] = ld<NT>(reinterpret_cast<const f32x4 *>(s.initial_color) + p0 + k * 64 + lane);
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) *reinterpret_cast<vec *>(&comp[row_of[a]][g]) = x[a];
#pragma unroll
    for (int m = 0; m < PPL / 2; ++m) {
        const uint32_t u = m * 64 + lane;
        *reinterpret_cast<f32x2 *>(&comp[3][2 * u]) = f32x2{pad[m][0], pad[m][2]};
        *reinterpret_cast<f32x2 *>(&comp[7][2 * u]) = f32x2{pad[m][1], pad[m][3]};
    }
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        quad[0][k * 64 + lane] = col[k];
        quad[1][k * 64 + lane] = ini[k];
    }
    wave_lds_sync();
    f32x4 *dst = aos + 4 * base;
#pragma unroll
    for (int k = 0; k < 4 * PPL; ++k) {
        const uint32_t q = k * 64 + lane, particle = q >> 2, field = q & 3u;
        f32x4 v;
        if (field < 2) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = comp[4 * field + c][particle];
        } else {
            v = quad[field - 2][particle];
        }
        if constexpr (NT) st_stream(dst, (uint64_t)k * 64, (uint64_t)k * 64 + 64, lane, v);
        else dst[q] = v;
    }
}

// Launch shape of the transposes, measured at 100 M particles (GB/s of the 128 B per
// particle moved): plain, 32 waves per CU 5.34 k (export) / 5.36 k (import); with streaming
// accesses and, through an LDS reservation, fewer resident waves per CU:
//   export, 256-thread workgroups + 36864 B (3 workgroups = 12 waves) 6.47-6.6 k;
//           64-thread workgroups + 6144 B (15 waves) 6.24-6.5 k, 3 % behind back to back
//   import, 256-thread workgroups + 20480 B (4 workgroups) 6.1 k;
//           64-thread workgroups + 6144 B / 7168 B (15 / 14 waves) 6.37-6.40 k,
//           4096 B (19 waves) 6.17 k, 8192 B (13 waves) 5.9 k, 16384 B (7 waves) 4.1 k
// The import gains from single-wave workgroups because no wave waits at a barrier for the
// other three waves' loads; the streaming export keeps 256 threads; at cache-resident sizes
// single waves are ahead for both (1 M particles, plain: export 6.50 -> 6.63 k, import
// 6.25 -> 6.77 k).  Plain accesses win while the SoA and AoS sides together fit the
// 256 MiB Infinity Cache (2 M particles: 7.0-7.1 k against 6.2 / 5.8 k streaming), streaming
// from 2.5 M particles on (6.2 / 5.9 k against 5.9 / 5.8 k).
// Overrides (psim_tuning.h): KNOB_AOS_RESERVE the LDS reservation of both transposes,
// KNOB_AOS_BLOCK (64 or 256) their workgroup size, KNOB_AOS_STREAM (0 / 1) the access form.
static bool aos_streaming(uint32_t n)
{
    const int v = knob(KNOB_AOS_STREAM);
    return v >= 0 ? v != 0 : (uint64_t)n * 128u > 288000000ull;
}

static uint32_t aos_reserve(uint32_t automatic)
{
    const int v = knob(KNOB_AOS_RESERVE);
    return v >= 0 ? (uint32_t)v : automatic;
}

static bool aos_single_wave(bool automatic)
{
    const int v = knob(KNOB_AOS_BLOCK);
    return v == 64 ? true : (v == (int)PSIM_BLOCK ? false : automatic);
}

template <bool NT, typename Pointer>
static void launch_transpose(const SoaPtrs &s, uint32_t first, uint32_t n,
                             Pointer aos, bool single_wave, uint32_t reserve,
                             uint32_t reserve_256, hipStream_t stream)
{
    if (aos_single_wave(single_wave)) {
        const dim3 grid((n + 63u) / 64u), block(64);
        if constexpr (std::is_const_v<std::remove_pointer_t<Pointer>>)
            hipLaunchKernelGGL((import_aos_kernel<NT, 64>), grid, block, aos_reserve(reserve),
                               stream, s, first, n, aos);
        else
            hipLaunchKernelGGL((export_aos_kernel<NT, 64>), grid, block, aos_reserve(reserve),
                               stream, s, first, n, aos);
    } else {
        const dim3 grid((n + PSIM_BLOCK - 1u) / PSIM_BLOCK), block(PSIM_BLOCK);
        if constexpr (std::is_const_v<std::remove_pointer_t<Pointer>>)
            hipLaunchKernelGGL((import_aos_kernel<NT, PSIM_BLOCK>), grid, block,
                               aos_reserve(reserve_256), stream, s, first, n, aos);
        else
            hipLaunchKernelGGL((export_aos_kernel<NT, PSIM_BLOCK>), grid, block,
                               aos_reserve(reserve_256), stream, s, first, n, aos);
    }
}

// KNOB_AOS_PPL = 1 (one particle per lane, the kernels above), 2 or 4.
static int aos_ppl(int automatic)
{
    const int v = knob(KNOB_AOS_PPL);
    return v == 1 || v == 2 || v == 4 ? v : automatic;
}

// The wide kernels over the full tiles of [first, first + n); returns how many particles they
// covered (0 when they do not apply: the SoA side must be 16-byte aligned at `first`).
template <bool IMPORT, typename Pointer>
static uint32_t launch_wide(const SoaPtrs &s, uint32_t first, uint32_t n, Pointer aos, bool streaming,
                            int ppl, uint32_t reserve, hipStream_t stream)
{
    if (ppl == 1 || (first & 3u) != 0) return 0;
    const uint32_t tile = 64u * (uint32_t)ppl, tiles = n / tile;
    if (tiles == 0) return 0;
    const dim3 grid(tiles), block(64);
    const uint32_t lds = aos_reserve(reserve);
#define PSIM_WIDE(NT, PPL)                                                                      \
    do {                                                                                        \
        if constexpr (IMPORT)                                                                   \
            hipLaunchKernelGGL((import_aos_wide_kernel<NT, PPL>), grid, block, lds, stream, s, first, aos); \
        else                                                                                    \
            hipLaunchKernelGGL((export_aos_wide_kernel<NT, PPL>), grid, block, lds, stream, s, first, aos); \
    } while (0)
    if (streaming) {
        if (ppl == 4) PSIM_WIDE(true, 4); else PSIM_WIDE(true, 2);
    } else {
        if (ppl == 4) PSIM_WIDE(false, 4); else PSIM_WIDE(false, 2);
    }
#undef PSIM_WIDE
    return tiles * tile;
}

hipError_t launch_export_aos(const SoaPtrs &s, uint32_t first, uint32_t n, psim_particle *aos,
                             hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    {
        const bool streaming = aos_streaming(n) && (reinterpret_cast<uintptr_t>(aos) & 127u) == 0 &&
                               (first & 31u) == 0;
        const uint32_t done = launch_wide<false>(s, first, n, reinterpret_cast<f32x4 *>(aos), streaming,
                                                 aos_ppl(1), 0, stream);
        if (done) {
            first += done, n -= done, aos += done;
            if (n == 0) return hipGetLastError();
        }
    }
    f32x4 *dst = reinterpret_cast<f32x4 *>(aos);
    // nt sc1 stores write partial 128-byte lines through to memory, which is slow: the
    // streaming form needs a line-aligned destination.
    if (aos_streaming(n) && (reinterpret_cast<uintptr_t>(aos) & 127u) == 0)
        launch_transpose<true>(s, first, n, dst, false, 6144, 36864, stream);
    else
        launch_transpose<false>(s, first, n, dst, true, n >= (1u << 20) ? 6144 : 0, 0, stream);
    return hipGetLastError();
}

hipError_t launch_import_aos(const SoaPtrs &s, uint32_t first, uint32_t n,
                             const psim_particle *aos, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    {
        const bool streaming = aos_streaming(n) && (first & 31u) == 0 &&
                               (reinterpret_cast<uintptr_t>(s.px) & 127u) == 0;
        // Streaming import: two particles per lane and 5 resident waves per CU (24320 B on top
        // of the kernel's 8384 B): 6.60 k GB/s at 100 M particles against 6.27 k for one
        // particle per lane, 6.43 against 6.26 k at 20 M, 6.25 against 5.95 k at 4 M (6 waves
        // 6.2-6.4 k, 4 waves equal, 3 waves 6.1 k; four particles per lane 6.3 k at best).  The
        // export gains nothing from wider lanes (6.35-6.46 k against 6.60-6.63 k).
        const uint32_t done = launch_wide<true>(s, first, n, reinterpret_cast<const f32x4 *>(aos),
                                                streaming, aos_ppl(streaming ? 2 : 1), 24320, stream);
        if (done) {
            first += done, n -= done, aos += done;
            if (n == 0) return hipGetLastError();
        }
    }
    const f32x4 *src = reinterpret_cast<const f32x4 *>(aos);
    // The SoA side stores 4 bytes per lane from particle `first` on: line-aligned only when
    // `first` is a multiple of 32 particles (the arrays themselves start 4 KiB-aligned).
    if (aos_streaming(n) && (first & 31u) == 0 && (reinterpret_cast<uintptr_t>(s.px) & 127u) == 0)
        launch_transpose<true>(s, first, n, src, true, 6144, 20480, stream);
    else
        launch_transpose<false>(s, first, n, src, true, n >= (1u << 20) ? 6144 : 0, 0, stream);
    return hipGetLastError();
}

// ---- checksums (wavefront reductions) ---------------------------------------

// Sum of a 64-bit value over the 64 lanes of a wave, result in lane 0.
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// out[k] += sum over particles of the bit pattern of component k.  Integer sums
// commute, so the result does not depend on the schedule.
template <bool NT>
__global__ void __launch_bounds__(PSIM_BLOCK) checksum_kernel(SoaPtrs s, uint32_t n,
                                                             unsigned long long *out)
{
    const float *arr[6] = {s.px, s.py, s.pz, s.vx, s.vy, s.vz};
    uint64_t acc[6] = {0, 0, 0, 0, 0, 0};
    // 16-byte loads over groups of 4 particles, then the n % 4 tail one by one
    const uint32_t ngroups = n >> 2;
    const uint64_t stride = (uint64_t)gridDim.x * PSIM_BLOCK;
    for (uint64_t g = (uint64_t)blockIdx.x * PSIM_BLOCK + threadIdx.x; g < ngroups; g += stride) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const f32x4 q = ld<NT>(reinterpret_cast<const f32x4 *>(arr[k]) + g);
            acc[k] += (uint64_t)__float_as_uint(q[0]) + __float_as_uint(q[1]) +
                      __float_as_uint(q[2]) + __float_as_uint(q[3]);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3u)) {
        const uint32_t i = (ngroups << 2) + threadIdx.x;
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k] += __float_as_uint(arr[k][i]);
    }
    __shared__ uint64_t partial[PSIM_BLOCK / 64][6];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const uint64_t w = wave_sum_u64(acc[k]);
        if (lane == 0) partial[wave][k] = w;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        uint64_t tot = 0;
#pragma unroll
        for (int w = 0; w < PSIM_BLOCK / 64; ++w) tot += partial[w][threadIdx.x];
        atomicAdd(&out[threadIdx.x], (unsigned long long)tot);
    }
}

hipError_t launch_checksum(const SoaPtrs &s, uint32_t n, uint64_t *out_dev, uint32_t max_blocks,
                           hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(out_dev, 0, 6 * sizeof(uint64_t), stream);
    if (e != hipSuccess || n == 0) return e;
    uint32_t grid = (n + PSIM_BLOCK - 1u) / PSIM_BLOCK;
    if (max_blocks > 0 && grid > max_blocks) grid = max_blocks;
    if ((uint64_t)n * 24u > 400000000ull)      // streaming: the state does not fit the caches
        hipLaunchKernelGGL(checksum_kernel<true>, dim3(grid), dim3(PSIM_BLOCK), 0, stream, s, n,
                           reinterpret_cast<unsigned long long *>(out_dev));
    else
        hipLaunchKernelGGL(checksum_kernel<false>, dim3(grid), dim3(PSIM_BLOCK), 0, stream, s, n,
                           reinterpret_cast<unsigned long long *>(out_dev));
    return hipGetLastError();
}

}  // namespace psim
