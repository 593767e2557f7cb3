/*
 * selftest.c -- runs the CPU oracle's entry points under AddressSanitizer and
 * UndefinedBehaviorSanitizer (host code only; built by `make selftest`).
 * TEST INFRASTRUCTURE: it checks the checker, nothing shipped depends on it.
 */
#include <stdio.h>
#include <stdlib.h>

#include "psim_oracle.c"

static psim_sim_params params(float radius, uint32_t color_mode, uint32_t dragging)
{
    psim_sim_params p = { 0.016f, 0.5f, color_mode, 5.0f, radius, dragging, 0.99f, 50.0f,
                          { 0.0f, 0.0f, 48.0f }, 0 };
    return p;
}

int main(void)
{
    if (!psim_oracle_layout_ok()) return 1;
    const uint64_t n = 10007;
    psim_particle *aos = malloc(n * sizeof(*aos));
    float *soa = malloc(6 * n * sizeof(float));
    float *col = malloc(4 * n * sizeof(float));
    if (!aos || !soa || !col) return 2;

    psim_oracle_gen_uniform_aos(aos, 123456789012ull, n);
    psim_oracle_gen_uniform_soa(soa, soa + n, soa + 2 * n, soa + 3 * n, soa + 4 * n, soa + 5 * n,
                                123456789012ull, n);
    for (uint64_t i = 0; i < 4 * n; ++i) col[i] = 0.25f;

    psim_attractor list[3] = {
        { { 1.0f, 2.0f, 3.0f }, 50.0f, 30.0f, 1, { 0, 0 } },
        { { -20.0f, 0.0f, 10.0f }, 5.0f, 10.0f, 0, { 0, 0 } },
        { { 0.0f, 0.0f, 48.0f }, 100.0f, 50.0f, 1, { 0, 0 } },
    };
    for (uint32_t mode = 0; mode < 3; ++mode) {
        psim_sim_params p = params(25.0f, mode, 1);
        for (int s = 0; s < 10; ++s) {
            psim_oracle_update_aos(aos, n, &p, NULL, 0, 0);
            psim_oracle_update_aos(aos, n, &p, list, 3, 2);
            psim_oracle_update_soa(soa, soa + n, soa + 2 * n, soa + 3 * n, soa + 4 * n,
                                   soa + 5 * n, mode ? col : NULL, n, &p, NULL, 0, 0);
        }
    }
    /* a particle at the attractor: NaN must propagate without tripping anything */
    aos[0].position[0] = 0.0f; aos[0].position[1] = 0.0f; aos[0].position[2] = 48.0f;
    psim_sim_params p = params(10.0f, 2, 1);
    psim_oracle_update_aos(aos, n, &p, NULL, 0, 1);
    if (aos[0].position[0] == aos[0].position[0]) return 3;       /* must be NaN */

    psim_oracle_gen_hollow_aos(aos, 100000, 90000, n);
    psim_oracle_gen_hollow_aos(aos, 1, 0, 1);
    psim_oracle_gen_filled_aos(aos, 500, n);
    uint64_t sums[6];
    psim_oracle_checksum_soa(soa, soa + n, soa + 2 * n, soa + 3 * n, soa + 4 * n, soa + 5 * n, n,
                             sums);
    uint64_t st[4], out[8];
    psim_oracle_xoshiro_seed(69, st);
    psim_oracle_xoshiro256pp(st, out, 8);
    printf("oracle selftest OK (%llu)\n", (unsigned long long)(sums[0] ^ out[7]));
    free(aos); free(soa); free(col);
    return 0;
}
