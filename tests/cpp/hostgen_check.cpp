// hostgen_check.cpp -- CPU check of the library's host-side generators
// (particle-simulation-3d_amd/csrc/psim_hostgen.h), built with a plain host compiler under
// AddressSanitizer + UBSan: the GF(2) jump-ahead of the Xoshiro256++ stream against drawing
// one by one, and slices of both initial states written to files for comparison with the
// oracle.
//
// usage: hostgen_check <out_dir>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <cmath>

#include "psim_hostgen.h"
#include "psim_threshold.h"

using psim_hostgen::Xoshiro256pp;

static int fail(const char *what)
{
    std::fprintf(stderr, "FAILED: %s\n", what);
    return 1;
}

static bool write_particles(const std::string &path, const std::vector<psim_particle> &v)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const size_t put = std::fwrite(v.data(), sizeof(psim_particle), v.size(), f);
    return std::fclose(f) == 0 && put == v.size();
}

int main(int argc, char **argv)
{
    if (argc != 2) return fail("usage: hostgen_check <out_dir>");
    const std::string dir = argv[1];

    // published first outputs of xoshiro256++ from the state {1, 2, 3, 4}
    {
        Xoshiro256pp g(0);
        g.s[0] = 1, g.s[1] = 2, g.s[2] = 3, g.s[3] = 4;
        const uint64_t want[3] = {41943041ull, 58720359ull, 3588806011781223ull};
        for (uint64_t w : want)
            if (g.next() != w) return fail("xoshiro256++ known answers");
    }

    // jump(k) == k calls of next(), across the short-hop threshold and for large k
    const uint64_t distances[] = {0, 1, 2, 4095, 4096, 4097, 65536, 1000003, 3ull * 5000000 + 1,
                                  (1ull << 32) + 12345, 3ull * 4000000000ull};
    for (uint64_t d : distances) {
        Xoshiro256pp a(69), b(69);
        a.jump(d);
        if (d <= 20000000ull) {
            for (uint64_t k = 0; k < d; ++k) b.next();
        } else {                                    // compose two jumps instead of drawing 1e10 numbers
            b.jump(d / 3);
            b.jump(d - d / 3);
        }
        if (std::memcmp(a.s, b.s, sizeof(a.s)) != 0) return fail("jump != sequential draws");
    }
    // additivity: jump(a) then jump(b) == jump(a + b)
    {
        Xoshiro256pp a(7), b(7);
        a.jump(123456789ull);
        a.jump(987654321ull);
        b.jump(123456789ull + 987654321ull);
        if (std::memcmp(a.s, b.s, sizeof(a.s)) != 0) return fail("jump is not additive");
    }

    // psim_in_range_threshold: `d2 < T` must equal `sqrtf(d2) < two_r` for every d2, checked
    // on the floats around T, around two_r^2, and at the special values, for many radii
    {
        Xoshiro256pp g(11);
        std::vector<float> radii = {0.0f, -1.0f, 1e-30f, 1e-23f, 1e-20f, 3e-19f, 0.5f, 1.0f, 2.0f, 20.0f,
                                    100.0f, 1.8e19f, 1.9e19f, 3e38f, INFINITY, NAN};
        for (int i = 0; i < 20000; ++i) {
            union { uint32_t u; float f; } r;
            r.u = (uint32_t)(g.next() >> 33);               // any positive bit pattern
            radii.push_back(r.f);
            radii.push_back(g.next_f32() * 100.0f);         // the slider range, doubled
        }
        for (float two_r : radii) {
            const float T = psim_in_range_threshold(two_r);
            std::vector<float> probes = {0.0f, 1.4e-45f, 1.0f, 3.4028235e38f, INFINITY, NAN};
            for (float centre : {T, two_r * two_r}) {
                union { float f; uint32_t u; } b;
                b.f = centre;
                if (centre == centre && centre > 0.0f && centre < INFINITY)
                    for (int d = -3; d <= 3; ++d) {
                        if ((int64_t)b.u + d < 0) continue;
                        union { float f; uint32_t u; } p;
                        p.u = (uint32_t)((int64_t)b.u + d);
                        probes.push_back(p.f);
                    }
            }
            for (float d2 : probes)
                if ((d2 < T) != (sqrtf(d2) < two_r)) {
                    std::fprintf(stderr, "two_r %a T %a d2 %a\n", two_r, T, d2);
                    return fail("d2 < T differs from sqrtf(d2) < two_r");
                }
        }
    }

    // slices of both states: one call over the range == many calls over its pieces
    const uint32_t count = 300001;
    std::vector<psim_particle> whole(count), pieces(count);
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 0) psim_hostgen::gen_hollow(whole.data(), count, 0, count);
        else psim_hostgen::gen_filled(whole.data(), 69, 0, count);
        for (uint64_t lo = 0; lo < count; lo += 70001) {
            const uint64_t len = count - lo < 70001 ? count - lo : 70001;
            if (mode == 0) psim_hostgen::gen_hollow(pieces.data() + lo, count, lo, len);
            else psim_hostgen::gen_filled(pieces.data() + lo, 69, lo, len);
        }
        if (std::memcmp(whole.data(), pieces.data(), count * sizeof(psim_particle)) != 0)
            return fail("a state generated in pieces differs from the whole");
        if (!write_particles(dir + (mode == 0 ? "/hollow.bin" : "/filled.bin"), whole))
            return fail("cannot write the state");
    }
    // a slice far into the filled stream (first particle 200 000 000)
    std::vector<psim_particle> far(1000);
    psim_hostgen::gen_filled(far.data(), 69, 200000000ull, far.size());
    if (!write_particles(dir + "/filled_far.bin", far)) return fail("cannot write the far slice");
    std::printf("hostgen ok\n");
    return 0;
}
