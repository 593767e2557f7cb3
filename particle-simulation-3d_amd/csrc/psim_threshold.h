// psim_threshold.h -- the exact squared-distance form of the reference's range test.
//
// The reference tests `dist < two_r` with dist = sqrt(d2) correctly rounded
// (src/simulation/cpu.rs:77-79).  x -> RN(sqrt(x)) is monotonic, so that test equals
// `d2 < T` with
//     T = the smallest float x >= 0 whose correctly rounded square root is >= two_r,
// exactly, for every d2 (a sum of squares: >= 0, +inf or NaN) and every two_r (NaN and
// two_r <= 0: never in range, T = NaN / 0; two_r * two_r overflowing: T = +inf).  T depends
// on the attractor only, so it is computed once per attractor (on the host for the SimParams
// attractor and host lists, by one lane per attractor while a device list is staged) and
// the ~20-instruction correctly rounded square root of the kernels runs only where some
// lane is in range.  Usable from host and device code (sqrtf is correctly rounded on both).
#ifndef PSIM_THRESHOLD_H
#define PSIM_THRESHOLD_H

#include <math.h>
#include <stdint.h>

#ifdef __HIPCC__
#define PSIM_HOST_DEVICE __host__ __device__
#else
#define PSIM_HOST_DEVICE
#endif

PSIM_HOST_DEVICE inline float psim_in_range_threshold(float two_r)
{
    if (!(two_r > 0.0f)) return two_r != two_r ? two_r : 0.0f;
    const float c = two_r * two_r;
    if (c > 3.402823466e+38f) return c;                      // +inf: every finite d2 is in range
    union { float f; uint32_t u; } b;
    b.f = c;
    while (sqrtf(b.f) >= two_r) b.u -= 1u;                   // stops at 0 at the latest: two_r > 0
    while (sqrtf(b.f) < two_r) b.u += 1u;                    // stops at +inf at the latest
    return b.f;
}

#endif  // PSIM_THRESHOLD_H
