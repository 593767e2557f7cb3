/* exchange_loop.c -- one rank of a one-process-per-GPU multi-attractor run, written in
 * plain C against include/psim.h alone: what a host in any language does with the
 * attractor exchange.  The ranks meet through a file (psim_exchange_rendezvous_file), every
 * rank moves its own attractor along a simple path and calls psim_exchange_step per frame.
 *
 * It reports the host time of a step (how long the calls take to return) next to the time a
 * step takes on the device, and the checksum of the final state, which the test-suite
 * compares with the CPU oracle's.
 *
 * usage: exchange_loop <rank> <world> <device> <particles_per_rank> <steps> <id_file>
 *                      [host|device|pipelined] [key=value ...]
 *   force=F radius=R   every attractor's force and radius (default 40, 25)
 *   timeout_ms=N       psim_exchange_set_timeout: give up instead of hanging when a rank has gone
 *   nap_us=N           the host sleeps N microseconds per step (a frame loop with other work)
 *   die_after=K        this rank leaves the run (exit status 0, nothing torn down) after K steps,
 *                      as a rank that crashed would
 * prints: host_us_per_step device_us_per_step checksum[6]
 * exit status: 0 done, 1 a call failed, 2 usage, 3 the exchange gave up (PSIM_ERR_COMM: the
 * message goes to stderr with the seconds the failing call took)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "psim.h"

static double now_us(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}

static psim_handle *g_handle = NULL;
static psim_exchange *g_exchange = NULL;

/* A failed call ends the program through an orderly tear-down: after PSIM_ERR_COMM the
 * exchange has aborted its communicator, and destroying it must not hang either. */
#define CHECK(call)                                                          \
    do {                                                                     \
        const double t_call_ = now_us();                                     \
        const int rc_ = (call);                                              \
        if (rc_ != PSIM_OK) {                                                \
            fprintf(stderr, "%s: %s [status %d after %.3f s]\n", #call,      \
                    psim_last_error(), rc_, (now_us() - t_call_) * 1e-6);    \
            psim_exchange_destroy(g_exchange);                               \
            psim_destroy(g_handle);                                          \
            fprintf(stderr, "torn down\n");                                  \
            return rc_ == PSIM_ERR_COMM ? 3 : 1;                             \
        }                                                                    \
    } while (0)

/* attractor of `rank` at step t: exactly representable steps, so that a checker in another
 * language reproduces it without rounding questions */
static float g_force = 40.0f, g_radius = 25.0f;

static psim_attractor attractor_at(int rank, int t)
{
    psim_attractor a = {{0.0f, 0.0f, 0.0f}, 0.0f, 0.0f, 1u, {0u, 0u}};
    a.force = g_force;
    a.radius = g_radius;
    a.position[0] = -40.0f + 10.0f * (float)rank + 0.25f * (float)(t % 64);
    a.position[1] = 30.0f - 0.5f * (float)(t % 32);
    a.position[2] = -20.0f + 5.0f * (float)rank;
    return a;
}

int main(int argc, char **argv)
{
    if (argc < 7) {
        fprintf(stderr, "usage: %s rank world device particles steps id_file [host|device|pipelined] "
                        "[force=F] [radius=R] [timeout_ms=N] [nap_us=N] [die_after=K]\n", argv[0]);
        return 2;
    }
    unsigned timeout_ms = 0;
    int die_after = -1, pacing = PSIM_EXCHANGE_DEVICE_PACED;
    unsigned nap_us = 0;
    for (int k = 7; k < argc; ++k) {
        if (strcmp(argv[k], "host") == 0) pacing = PSIM_EXCHANGE_HOST_PACED;
        else if (strcmp(argv[k], "device") == 0) pacing = PSIM_EXCHANGE_DEVICE_PACED;
        else if (strcmp(argv[k], "pipelined") == 0) pacing = PSIM_EXCHANGE_PIPELINED;
        else if (sscanf(argv[k], "force=%f", &g_force) == 1) {}
        else if (sscanf(argv[k], "radius=%f", &g_radius) == 1) {}
        else if (sscanf(argv[k], "timeout_ms=%u", &timeout_ms) == 1) {}
        else if (sscanf(argv[k], "die_after=%d", &die_after) == 1) {}
        else if (sscanf(argv[k], "nap_us=%u", &nap_us) == 1) {}
        else {
            fprintf(stderr, "unknown argument %s\n", argv[k]);
            return 2;
        }
    }
    const int rank = atoi(argv[1]), world = atoi(argv[2]), device = atoi(argv[3]);
    const uint32_t n = (uint32_t)strtoul(argv[4], NULL, 10);
    const int steps = atoi(argv[5]);
    psim_handle *h = NULL;
    psim_exchange *x = NULL;
    uint8_t id[PSIM_EXCHANGE_ID_BYTES];
    CHECK(psim_create_shard(&h, n, (uint64_t)rank * n, (uint64_t)world * n, PSIM_GEN_UNIFORM, device));
    g_handle = h;
    CHECK(psim_exchange_rendezvous_file(argv[6], rank, id, 60000));
    CHECK(psim_exchange_create(&x, h, id, rank, world));
    g_exchange = x;
    CHECK(psim_exchange_set_pacing(x, pacing));
    if (timeout_ms) CHECK(psim_exchange_set_timeout(x, timeout_ms));

    psim_sim_params p = {0.016f, 0.25f, 0u, 5.0f, 10.0f, 0u, 0.99f, 50.0f, {0.0f, 0.0f, 0.0f}, 0u};
    double host_us = 0.0;
    float device_ms = 0.0f;
    /* first the steps whose result is checked, then the same number again, timed */
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) CHECK(psim_timer_start(h));
        const double t0 = now_us();
        for (int t = 0; t < steps; ++t) {
            if (pass * steps + t == die_after) {
                CHECK(psim_sync(h));                /* what was enqueued has run: then gone */
                fprintf(stderr, "rank %d leaves after %d steps\n", rank, die_after);
                _exit(0);
            }
            const psim_attractor mine = attractor_at(rank, pass * steps + t);
            CHECK(psim_exchange_step(x, &p, &mine));
            if (nap_us) {
                const struct timespec nap = {nap_us / 1000000u, (long)(nap_us % 1000000u) * 1000L};
                (void)nanosleep(&nap, NULL);
            }
        }
        if (pass == 1) {
            host_us = (now_us() - t0) / steps;
            CHECK(psim_timer_stop(h, &device_ms));
        }
        CHECK(psim_exchange_sync(x));
    }
    uint64_t sum[6];
    CHECK(psim_checksum(h, sum));
    printf("%.2f %.2f", host_us, device_ms * 1e3 / steps);
    for (int k = 0; k < 6; ++k) printf(" %llu", (unsigned long long)sum[k]);
    printf("\n");
    psim_exchange_destroy(x);
    psim_destroy(h);
    return 0;
}
