// psim_hostgen.h -- the reference's two initial states, generated on the host.
//
// cos/sin/cbrt/acos of fp32 arguments differ between a device math library and the host
// C library that the reference's Rust std calls, so these states are produced on the host
// (same libm as the reference on this platform) and uploaded.  They run at creation /
// reset only.  Host-only code without any GPU call, so that it can also be built and
// checked on a machine without one.
#ifndef PSIM_HOSTGEN_H
#define PSIM_HOSTGEN_H

#include <math.h>
#include <stdint.h>

#include <vector>

#include "psim.h"

namespace psim_hostgen {

// Fibonacci sphere, src/simulation/mod.rs:150-167.  `count` is the size of the
// whole set; particles [first, first + n) are written.
inline void gen_hollow(psim_particle *out, uint32_t count, uint64_t first, uint64_t n)
{
    const float radius = 50.0f;
    const float golden = 3.14159265358979323846f * (3.0f - sqrtf(5.0f));
    const float denom = (float)((count > 1u ? count : 1u) - 1u);
    for (uint64_t k = 0; k < n; ++k) {
        const float fi = (float)(uint32_t)(first + k);
        const float y = 1.0f - (fi / denom) * 2.0f;
        const float ring = sqrtf(1.0f - y * y);
        const float theta = golden * fi;
        const float x = cosf(theta) * ring;
        const float z = sinf(theta) * ring;
        psim_particle &q = out[k];
        q.position[0] = x * radius;
        q.position[1] = y * radius;
        q.position[2] = z * radius;
        q.padding1 = 0.0f;
        q.velocity[0] = q.velocity[1] = q.velocity[2] = 0.0f;
        q.padding2 = 0.0f;
        for (int c = 0; c < 3; ++c)
            q.color[c] = q.initial_color[c] = (q.position[c] / radius + 1.0f) * 0.5f;
        q.color[3] = q.initial_color[3] = 1.0f;
    }
}

// The generator behind the reference's `SmallRng::seed_from_u64(69)`
// (src/simulation/mod.rs:170): Xoshiro256++ with SplitMix64 seeding; an f32 draw
// is the top 24 bits of the upper half of a 64-bit output, times 2^-24.
struct Xoshiro256pp {
    uint64_t s[4];
    explicit Xoshiro256pp(uint64_t seed)
    {
        for (auto &w : s) {
            seed += 0x9E3779B97F4A7C15ull;
            uint64_t z = seed;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            w = z ^ (z >> 31);
        }
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next()
    {
        const uint64_t r = rotl(s[0] + s[3], 23) + s[0];
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return r;
    }
    float next_f32() { return (float)((uint32_t)(next() >> 32) >> 8) * 0x1.0p-24f; }
    inline void jump(uint64_t steps);
};

// The state transition of Xoshiro256 is linear over GF(2), so `steps` calls of next()
// equal one multiplication of the 256-bit state by T^steps.  XoshiroPowers holds the
// images of the 256 basis states under T^(2^k), k = 0..39, built once by repeated
// squaring; a jump applies the powers named by the bits of `steps`.  This lets every
// host thread (and every shard) start at its own particle without drawing the numbers
// before it.
struct XoshiroPowers {
    static constexpr int kBits = 40;        // jumps are < 3 * 2^32 draws
    struct State { uint64_t w[4]; };
    std::vector<State> col;                        // [k * 256 + bit]

    static State apply(const State *m, const State &v)
    {
        State r{{0, 0, 0, 0}};
        for (int word = 0; word < 4; ++word)
            for (uint64_t bits = v.w[word]; bits; bits &= bits - 1) {
                const State &c = m[word * 64 + __builtin_ctzll(bits)];
                r.w[0] ^= c.w[0], r.w[1] ^= c.w[1], r.w[2] ^= c.w[2], r.w[3] ^= c.w[3];
            }
        return r;
    }
    XoshiroPowers() : col((size_t)kBits * 256)
    {
        for (int bit = 0; bit < 256; ++bit) {      // T itself: one step of each basis state
            Xoshiro256pp g(0);
            g.s[0] = g.s[1] = g.s[2] = g.s[3] = 0;
            g.s[bit / 64] = 1ull << (bit % 64);
            g.next();
            col[bit] = State{{g.s[0], g.s[1], g.s[2], g.s[3]}};
        }
        for (int k = 1; k < kBits; ++k)            // T^(2^k) = (T^(2^(k-1)))^2
            for (int bit = 0; bit < 256; ++bit)
                col[(size_t)k * 256 + bit] =
                    apply(&col[(size_t)(k - 1) * 256], col[(size_t)(k - 1) * 256 + bit]);
    }
    static const XoshiroPowers &get()
    {
        static const XoshiroPowers p;              // thread-safe one-time construction
        return p;
    }
};

inline void Xoshiro256pp::jump(uint64_t steps)
{
    if (steps < 4096) {                            // short hops: just draw
        for (uint64_t k = 0; k < steps; ++k) next();
        return;
    }
    const XoshiroPowers &p = XoshiroPowers::get();
    XoshiroPowers::State v{{s[0], s[1], s[2], s[3]}};
    for (int k = 0; k < XoshiroPowers::kBits; ++k)
        if (steps >> k & 1) v = XoshiroPowers::apply(&p.col[(size_t)k * 256], v);
    s[0] = v.w[0], s[1] = v.w[1], s[2] = v.w[2], s[3] = v.w[3];
}

// Seeded uniform ball, src/simulation/mod.rs:168-188.  One sequential random stream,
// three draws per particle: particles [first, first + n) of the whole set are written,
// the generator jumping to draw 3 * first.
inline void gen_filled(psim_particle *out, uint64_t seed, uint64_t first, uint64_t n)
{
    const float radius = 50.0f;
    const float pi = 3.14159265358979323846f;
    Xoshiro256pp rng(seed);
    rng.jump(3 * first);
    for (uint64_t k = 0; k < n; ++k) {
        const float r = radius * cbrtf(rng.next_f32());
        const float theta = rng.next_f32() * 2.0f * pi;
        const float phi = acosf(rng.next_f32() * 2.0f - 1.0f);
        psim_particle &q = out[k];
        q.position[0] = r * sinf(phi) * cosf(theta);
        q.position[1] = r * cosf(phi);
        q.position[2] = r * sinf(phi) * sinf(theta);
        q.padding1 = 0.0f;
        q.velocity[0] = q.velocity[1] = q.velocity[2] = 0.0f;
        q.padding2 = 0.0f;
        for (int c = 0; c < 3; ++c)
            q.color[c] = q.initial_color[c] = (q.position[c] / radius + 1.0f) * 0.5f;
        q.color[3] = q.initial_color[3] = 1.0f;
    }
}

}  // namespace psim_hostgen

#endif  // PSIM_HOSTGEN_H
