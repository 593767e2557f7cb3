/*
 * psim_oracle.c -- CPU restatement of the reference's particle update.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker and the CPU
 * baseline.  Nothing in the shipped GPU path links, loads or calls it; only the
 * test suite, the smoke check and the benchmark's CPU-baseline leg do.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference is Rust (it cannot be built
 * here: no rustc/cargo) and ships no tests, golden vectors or fixtures for this
 * path.  The restatement is pinned instead by (1) an independent op-by-op
 * numpy-float32 derivation (tests/gen_golden.py), compared on committed vectors
 * and on seeded random sweeps (tests/test_random_parity.py), (2) hand-derived
 * known answers (tests/test_oracle_golden.py), and (3) the reference's own second
 * implementation, the compute shader (src/shaders/compute.wgsl:31-99), restated
 * from its text and required to agree to 1e-5 wherever the two sources say the
 * same thing (tests/test_wgsl_twin.py) -- which pins the algorithm, not the bits.
 *
 * What is restated (paths relative to the reference root):
 *   src/simulation/cpu.rs:51-59    parameter hoist, dragging = is_mouse_dragging > 0
 *   src/simulation/cpu.rs:63       active slice = first particle_count particles
 *   src/simulation/cpu.rs:65-113   the per-particle body (order of operations kept)
 *   src/simulation/mod.rs:52-98    SimParams / Particle layouts
 *   src/simulation/mod.rs:145-192  initial-state generators
 * Third-party arithmetic that is not in the reference tree, restated from its
 * published behaviour: glam 0.30.10 `Vec3` (Cargo.lock:959-962) -- scalar
 * {x,y,z}; dot = (x*x' + y*y') + z*z'; length = sqrt(dot(v,v));
 * normalize = v * (1.0 / length(v)) (one reciprocal, three multiplies; the
 * zero-length assert is compiled out, so a zero vector normalises to NaN);
 * Rust core f32::{powi(2) = x*x, min/max = IEEE minNum/maxNum (NaN-ignoring),
 * clamp = two compares that let NaN through}.  rustc never contracts a*b+c into
 * an FMA, so this file must be compiled with -ffp-contract=off and without
 * -ffast-math; every +,-,*,/ and sqrt below is one correctly rounded fp32
 * operation.  rayon's par_iter_mut (cpu.rs:65) only schedules: particles are
 * independent, so any partition (here an OpenMP static one) gives identical
 * results.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/psim.h"

#if defined(__FAST_MATH__)
#error "psim_oracle.c must not be compiled with -ffast-math"
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* ---- glam Vec3 restatement ------------------------------------------------ */

typedef struct { float x, y, z; } vec3;

static inline float v3_dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float v3_length(vec3 a) { return sqrtf(v3_dot(a, a)); }
static inline vec3 v3_sub(vec3 a, vec3 b) { vec3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
static inline vec3 v3_add(vec3 a, vec3 b) { vec3 r = { a.x + b.x, a.y + b.y, a.z + b.z }; return r; }
static inline vec3 v3_scale(vec3 a, float s) { vec3 r = { a.x * s, a.y * s, a.z * s }; return r; }
static inline vec3 v3_normalize(vec3 a) { return v3_scale(a, 1.0f / v3_length(a)); }

/* Rust f32::clamp(0.0, 1.0): NaN passes through. */
static inline float clamp01(float x)
{
    if (x < 0.0f) x = 0.0f;
    if (x > 1.0f) x = 1.0f;
    return x;
}

/* ---- the force law, cpu.rs:76-83 ------------------------------------------ */

static inline vec3 apply_attractor(vec3 velocity, vec3 position, vec3 apos, float aforce,
                                   float aradius, float dt)
{
    vec3 dir = v3_sub(apos, position);                       /* cpu.rs:76 */
    float dist = v3_length(dir);                             /* cpu.rs:77 */
    if (dist < aradius * 2.0f) {                             /* cpu.rs:79 */
        float t = 1.0f - dist / (aradius * 2.0f);
        float force_factor = (t * t) * 2.0f;                 /* cpu.rs:80 */
        vec3 force = v3_scale(v3_scale(v3_normalize(dir), aforce), force_factor); /* :81 */
        velocity = v3_add(velocity, v3_scale(force, dt));    /* cpu.rs:82 */
    }
    return velocity;
}

/* One particle, cpu.rs:66-112.  `color` is read and written in place (mode 0
 * keeps the current colour, cpu.rs:106).  When `attr` is non-null the single
 * mouse attractor is replaced by the list attr[0..na) applied in index order
 * (the multi-attractor extension; with one entry equal to the mouse fields the
 * arithmetic is the same). */
static inline void step_one(vec3 *pos, vec3 *vel, float color[4], const psim_sim_params *p,
                            const psim_attractor *attr, uint32_t na)
{
    const float dt = p->delta_time;
    vec3 position = *pos, velocity = *vel;

    velocity.y -= p->gravity * dt;                           /* cpu.rs:72 */

    if (attr) {
        for (uint32_t a = 0; a < na; ++a) {
            if (attr[a].active > 0) {
                vec3 ap = { attr[a].position[0], attr[a].position[1], attr[a].position[2] };
                velocity = apply_attractor(velocity, position, ap, attr[a].force,
                                           attr[a].radius, dt);
            }
        }
    } else if (p->is_mouse_dragging > 0) {                   /* cpu.rs:55,75 */
        vec3 mp = { p->mouse_position[0], p->mouse_position[1], p->mouse_position[2] };
        velocity = apply_attractor(velocity, position, mp, p->mouse_force, p->mouse_radius, dt);
    }

    position = v3_add(position, v3_scale(velocity, dt));     /* cpu.rs:87 */
    velocity = v3_scale(velocity, p->damping);               /* cpu.rs:90 */

    if (p->color_mode == 1) {                                /* cpu.rs:94-99 */
        float speed = v3_length(velocity);
        float ns = fminf(speed / 5.0f, 1.0f);
        color[0] = ns;
        color[1] = 0.5f - ns * 0.5f;
        color[2] = 1.0f - ns;
        color[3] = 1.0f;
    } else if (p->color_mode == 2) {                         /* cpu.rs:100-105 */
        float d = v3_length(position);
        float nd = clamp01(d / fmaxf(p->max_dist_for_color, 0.01f));
        color[0] = nd;
        color[1] = 0.0f;
        color[2] = 1.0f - nd;
        color[3] = 1.0f;
    }                                                        /* else keep, cpu.rs:106 */

    *pos = position;
    *vel = velocity;
}

/* ---- public: update ------------------------------------------------------- */

/* AoS update in the reference's own layout (cpu.rs:63-113).  threads <= 0 means
 * "all available". */
ORACLE_API void psim_oracle_update_aos(psim_particle *particles, uint64_t n,
                                       const psim_sim_params *params,
                                       const psim_attractor *attr, uint32_t na, int threads)
{
    (void)threads;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(threads)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        psim_particle *q = &particles[i];
        vec3 pos = { q->position[0], q->position[1], q->position[2] };
        vec3 vel = { q->velocity[0], q->velocity[1], q->velocity[2] };
        step_one(&pos, &vel, q->color, params, attr, na);
        q->position[0] = pos.x; q->position[1] = pos.y; q->position[2] = pos.z;
        q->velocity[0] = vel.x; q->velocity[1] = vel.y; q->velocity[2] = vel.z;
    }
}

/* Same arithmetic over six SoA arrays (+ optional interleaved RGBA colour,
 * which may be null when color_mode is 0). */
ORACLE_API void psim_oracle_update_soa(float *px, float *py, float *pz, float *vx, float *vy,
                                       float *vz, float *color, uint64_t n,
                                       const psim_sim_params *params,
                                       const psim_attractor *attr, uint32_t na, int threads)
{
    (void)threads;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(threads)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        vec3 pos = { px[i], py[i], pz[i] };
        vec3 vel = { vx[i], vy[i], vz[i] };
        float keep[4] = { 0, 0, 0, 0 };
        step_one(&pos, &vel, color ? &color[4 * i] : keep, params, attr, na);
        px[i] = pos.x; py[i] = pos.y; pz[i] = pos.z;
        vx[i] = vel.x; vy[i] = vel.y; vz[i] = vel.z;
    }
}

/* ---- public: synthetic uniform state -------------------------------------- */

/* Counter-based SplitMix64: the value depends only on (seed, counter). */
static inline uint64_t splitmix64_at(uint64_t seed, uint64_t counter)
{
    uint64_t z = seed + (counter + 1u) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* 24 random bits -> [0,1), exact in fp32. */
static inline float unit_at(uint64_t seed, uint64_t counter)
{
    return (float)(uint32_t)(splitmix64_at(seed, counter) >> 40) * 0x1.0p-24f;
}

#define PSIM_UNIFORM_SEED 69ull   /* the reference's own fixed seed, mod.rs:170 */

/* Particle with GLOBAL index g of the synthetic benchmark state: position
 * U[-50,50)^3 (the reference's world scale, sphere radius 50, mod.rs:147),
 * velocity U[-1,1)^3, colour = (pos/50 + 1) * 0.5, alpha 1 (mod.rs:162-163),
 * paddings 0.  Component c of particle g uses counter 6*g + c. */
static inline void uniform_particle(uint64_t g, psim_particle *q)
{
    for (int c = 0; c < 3; ++c) {
        q->position[c] = unit_at(PSIM_UNIFORM_SEED, 6u * g + (uint64_t)c) * 100.0f - 50.0f;
        q->velocity[c] = unit_at(PSIM_UNIFORM_SEED, 6u * g + 3u + (uint64_t)c) * 2.0f - 1.0f;
    }
    q->padding1 = 0.0f;
    q->padding2 = 0.0f;
    for (int c = 0; c < 3; ++c) {
        float col = (q->position[c] / 50.0f + 1.0f) * 0.5f;
        q->color[c] = col;
        q->initial_color[c] = col;
    }
    q->color[3] = 1.0f;
    q->initial_color[3] = 1.0f;
}

ORACLE_API void psim_oracle_gen_uniform_aos(psim_particle *out, uint64_t global_first, uint64_t n)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i) uniform_particle(global_first + (uint64_t)i, &out[i]);
}

ORACLE_API void psim_oracle_gen_uniform_soa(float *px, float *py, float *pz, float *vx, float *vy,
                                            float *vz, uint64_t global_first, uint64_t n)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        psim_particle q;
        uniform_particle(global_first + (uint64_t)i, &q);
        px[i] = q.position[0]; py[i] = q.position[1]; pz[i] = q.position[2];
        vx[i] = q.velocity[0]; vy[i] = q.velocity[1]; vz[i] = q.velocity[2];
    }
}

/* ---- public: the reference's generators (mod.rs:145-192) ------------------- */

/* Hollow: Fibonacci sphere, mod.rs:150-167.  `count` is the size of the whole
 * set (it enters the y spacing); particles [first, first+n) are produced.
 * cos/sin are the C library's cosf/sinf, which is what Rust's f32::cos/sin call
 * on Linux.  count == 1 gives 0/0 = NaN for y, as in the reference (mod.rs:153). */
ORACLE_API void psim_oracle_gen_hollow_aos(psim_particle *out, uint32_t count, uint64_t first,
                                           uint64_t n)
{
    const float sphere_radius = 50.0f;                                 /* mod.rs:147 */
    const float golden_angle = 3.14159265358979323846f * (3.0f - sqrtf(5.0f)); /* :151 */
    const uint32_t cm = count > 1u ? count : 1u;                       /* count.max(1) */
    const float denom = (float)(cm - 1u);
    for (uint64_t k = 0; k < n; ++k) {
        uint32_t i = (uint32_t)(first + k);
        float fi = (float)i;
        float y = 1.0f - (fi / denom) * 2.0f;                          /* mod.rs:153 */
        float radius_at_y = sqrtf(1.0f - y * y);                       /* mod.rs:154 */
        float theta = golden_angle * fi;                               /* mod.rs:155 */
        float x = cosf(theta) * radius_at_y;                           /* mod.rs:157 */
        float z = sinf(theta) * radius_at_y;                           /* mod.rs:158 */
        psim_particle *q = &out[k];
        q->position[0] = x * sphere_radius;                            /* mod.rs:160 */
        q->position[1] = y * sphere_radius;
        q->position[2] = z * sphere_radius;
        q->padding1 = 0.0f;
        q->velocity[0] = q->velocity[1] = q->velocity[2] = 0.0f;       /* mod.rs:161 */
        q->padding2 = 0.0f;
        for (int c = 0; c < 3; ++c) {                                  /* mod.rs:162-163 */
            float col = (q->position[c] / sphere_radius + 1.0f) * 0.5f;
            q->color[c] = col;
            q->initial_color[c] = col;
        }
        q->color[3] = 1.0f;
        q->initial_color[3] = 1.0f;
    }
}

/* rand 0.9.2 SmallRng on 64-bit targets = Xoshiro256++ (restated from the
 * published algorithm; the crate source is not in the reference tree):
 * seed_from_u64 fills the 4-word state with consecutive SplitMix64 outputs;
 * next_u64 = rotl(s0 + s3, 23) + s0, then the xoshiro state update;
 * next_u32 = upper 32 bits of next_u64; random::<f32>() = (next_u32 >> 8) * 2^-24. */
typedef struct { uint64_t s[4]; } xoshiro256pp;

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

static void xoshiro_seed_from_u64(xoshiro256pp *r, uint64_t state)
{
    for (int i = 0; i < 4; ++i) {
        state += 0x9E3779B97F4A7C15ull;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        r->s[i] = z ^ (z >> 31);
    }
}

static inline uint64_t xoshiro_next_u64(xoshiro256pp *r)
{
    uint64_t *s = r->s;
    uint64_t result = rotl64(s[0] + s[3], 23) + s[0];
    uint64_t t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return result;
}

static inline float xoshiro_next_f32(xoshiro256pp *r)
{
    uint32_t u = (uint32_t)(xoshiro_next_u64(r) >> 32);
    return (float)(u >> 8) * 0x1.0p-24f;
}

/* Filled: seeded uniform ball, mod.rs:168-188.  The stream is sequential, so the
 * first `first` particles are drawn and discarded to reach the requested range. */
ORACLE_API void psim_oracle_gen_filled_aos(psim_particle *out, uint64_t first, uint64_t n)
{
    const float sphere_radius = 50.0f;
    const float pi = 3.14159265358979323846f;
    xoshiro256pp rng;
    xoshiro_seed_from_u64(&rng, 69);                                   /* mod.rs:170 */
    for (uint64_t k = 0; k < first + n; ++k) {
        float r = sphere_radius * cbrtf(xoshiro_next_f32(&rng));       /* mod.rs:173 */
        float theta = xoshiro_next_f32(&rng) * 2.0f * pi;              /* mod.rs:174 */
        float phi = acosf(xoshiro_next_f32(&rng) * 2.0f - 1.0f);       /* mod.rs:175 */
        if (k < first) continue;
        psim_particle *q = &out[k - first];
        q->position[0] = r * sinf(phi) * cosf(theta);                  /* mod.rs:177 */
        q->position[1] = r * cosf(phi);                                /* mod.rs:178 */
        q->position[2] = r * sinf(phi) * sinf(theta);                  /* mod.rs:179 */
        q->padding1 = 0.0f;
        q->velocity[0] = q->velocity[1] = q->velocity[2] = 0.0f;
        q->padding2 = 0.0f;
        for (int c = 0; c < 3; ++c) {                                  /* mod.rs:183-184 */
            float col = (q->position[c] / sphere_radius + 1.0f) * 0.5f;
            q->color[c] = col;
            q->initial_color[c] = col;
        }
        q->color[3] = 1.0f;
        q->initial_color[3] = 1.0f;
    }
}

/* Raw generator outputs for known-answer tests: `out[0..n)` = successive
 * next_u64 values of Xoshiro256++ started from state `s` (copied, not updated). */
ORACLE_API void psim_oracle_xoshiro256pp(const uint64_t s[4], uint64_t *out, uint64_t n)
{
    xoshiro256pp r;
    memcpy(r.s, s, sizeof(r.s));
    for (uint64_t i = 0; i < n; ++i) out[i] = xoshiro_next_u64(&r);
}

/* State words produced by seed_from_u64(seed). */
ORACLE_API void psim_oracle_xoshiro_seed(uint64_t seed, uint64_t s_out[4])
{
    xoshiro256pp r;
    xoshiro_seed_from_u64(&r, seed);
    memcpy(s_out, r.s, sizeof(r.s));
}

/* ---- public: checksums ---------------------------------------------------- */

/* out[k] = sum of the uint32 bit patterns of component k over n particles,
 * as uint64 mod 2^64 (order independent, exact). */
ORACLE_API void psim_oracle_checksum_soa(const float *px, const float *py, const float *pz,
                                         const float *vx, const float *vy, const float *vz,
                                         uint64_t n, uint64_t out[6])
{
    const float *arr[6] = { px, py, pz, vx, vy, vz };
    for (int k = 0; k < 6; ++k) {
        uint64_t s = 0;
        const float *a = arr[k];
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(+ : s)
#endif
        for (int64_t i = 0; i < (int64_t)n; ++i) {
            uint32_t bits;
            memcpy(&bits, &a[i], 4);
            s += bits;
        }
        out[k] = s;
    }
}

/* ---- public: misc ---------------------------------------------------------- */

ORACLE_API int psim_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Layout self-check used by the tests: sizes/offsets as the reference declares. */
ORACLE_API int psim_oracle_layout_ok(void)
{
    return sizeof(psim_sim_params) == 48 && offsetof(psim_sim_params, color_mode) == 8 &&
           offsetof(psim_sim_params, is_mouse_dragging) == 20 &&
           offsetof(psim_sim_params, mouse_position) == 32 && sizeof(psim_particle) == 64 &&
           offsetof(psim_particle, velocity) == 16 && offsetof(psim_particle, color) == 32 &&
           offsetof(psim_particle, initial_color) == 48 && sizeof(psim_attractor) == 32;
}
