// psim_group.cpp -- one particle set over several GPUs, driven from one process: the
// psim_group_* entry points of include/psim.h.  Built on the per-handle entry points alone
// (every shard is a psim_handle made by psim_create_shard), so this file contains no GPU
// code of its own.
#include <errno.h>
#include <string.h>

#include <new>
#include <string>
#include <thread>
#include <vector>

#include "psim_checkpoint_file.h"

#define PSIM_API extern "C" __attribute__((visibility("default")))

// ---- several GPUs from one process ------------------------------------------------

struct psim_group {
    uint64_t total = 0;
    std::vector<psim_handle *> shards;
    std::vector<uint64_t> first;     // first[k] = first particle of shard k; first[n] = total
};

namespace {

// Calls fn(k) for k = 0 .. n-1 on up to 16 host threads (k, k + threads, ... on one thread),
// so that the shards' synchronous entry points -- generators, record transfers, checksums,
// checkpoint I/O -- run on all devices at once instead of device after device.  Returns the
// status of the lowest k that failed, with its message as this thread's last error.
template <typename Fn>
int for_each_concurrently(size_t n, Fn fn)
{
    if (n == 0) return PSIM_OK;
    if (n == 1) return fn((size_t)0);
    const size_t threads = n < 16 ? n : 16;
    std::vector<int> status(n, PSIM_OK);
    std::vector<std::string> message(n);
    const auto work = [&](size_t t) {
        for (size_t k = t; k < n; k += threads) {
            try {
                status[k] = fn(k);
                if (status[k] != PSIM_OK) message[k] = psim_last_error();   // thread-local text
            } catch (...) {
                status[k] = PSIM_ERR_NOMEM;
            }
        }
    };
    try {
        std::vector<std::thread> pool;
        pool.reserve(threads);
        size_t started = 1;
        try {
            for (; started < threads; ++started) pool.emplace_back(work, started);
        } catch (...) {                              // no more threads: do their share here
            for (size_t t = started; t < threads; ++t) work(t);
        }
        work(0);
        for (auto &th : pool) th.join();
    } catch (...) {
        return psim_fail(PSIM_ERR_NOMEM, "host allocation failed");
    }
    for (size_t k = 0; k < n; ++k)
        if (status[k] != PSIM_OK)
            return psim_fail(status[k], "%s",
                             message[k].empty() ? "host allocation failed" : message[k].c_str());
    return PSIM_OK;
}

// For every shard that overlaps [first, first + n): call fn(shard, first_in_shard, count,
// offset_in_request), the shards at the same time.
template <typename Fn>
int for_overlapping_shards(psim_group *g, uint64_t first, uint64_t n, Fn fn)
{
    return for_each_concurrently(g->shards.size(), [&](size_t k) {
        const uint64_t lo = first > g->first[k] ? first : g->first[k];
        const uint64_t hi = first + n < g->first[k + 1] ? first + n : g->first[k + 1];
        if (lo >= hi) return (int)PSIM_OK;
        return fn(g->shards[k], lo - g->first[k], hi - lo, lo - first);
    });
}

int check_group_range(const psim_group *g, uint64_t first, uint64_t n)
{
    if (first > g->total || n > g->total - first)
        return psim_fail(PSIM_ERR_INVALID, "range [%llu, +%llu) exceeds particle count %llu",
                    (unsigned long long)first, (unsigned long long)n,
                    (unsigned long long)g->total);
    return PSIM_OK;
}

}  // namespace

PSIM_API int psim_group_create(psim_group **out, uint64_t total_count, const int *device_ids,
                               uint32_t n_shards, int generation_mode)
{
    if (!out) return psim_fail(PSIM_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (!device_ids || n_shards == 0)
        return psim_fail(PSIM_ERR_INVALID, "a group needs at least one device");
    if (n_shards > 1024) return psim_fail(PSIM_ERR_INVALID, "%u shards: at most 1024", n_shards);
    if (total_count / n_shards + (total_count % n_shards ? 1u : 0u) > PSIM_MAX_PARTICLES)
        return psim_fail(PSIM_ERR_INVALID, "%llu particles over %u shards exceed %u per shard",
                    (unsigned long long)total_count, n_shards, PSIM_MAX_PARTICLES);
    psim_group *g = new (std::nothrow) psim_group();
    if (!g) return psim_fail(PSIM_ERR_NOMEM, "host allocation failed");
    g->total = total_count;
    // k * total / n without overflowing 64 bits: total = q * n + r
    const uint64_t q = total_count / n_shards, r = total_count % n_shards;
    try {
        g->first.reserve(n_shards + 1u);
    } catch (...) {
        delete g;
        return psim_fail(PSIM_ERR_NOMEM, "host allocation failed");
    }
    for (uint32_t k = 0; k <= n_shards; ++k) g->first.push_back(q * k + r * k / n_shards);
    try {
        g->shards.assign(n_shards, nullptr);
    } catch (...) {
        delete g;
        return psim_fail(PSIM_ERR_NOMEM, "host allocation failed");
    }
    // every shard allocates and generates on its own device, all devices at once
    const int rc = for_each_concurrently(n_shards, [&](size_t k) {
        const uint32_t count = (uint32_t)(g->first[k + 1] - g->first[k]);
        // a set of one shard still is a shard of itself: global_count must be non-zero
        return total_count == 0
                   ? psim_create(&g->shards[k], 0, generation_mode, device_ids[k])
                   : psim_create_shard(&g->shards[k], count, g->first[k], total_count,
                                       generation_mode, device_ids[k]);
    });
    if (rc != PSIM_OK) {
        const std::string why = psim_last_error();   // destroying must not lose the reason
        psim_group_destroy(g);
        return psim_fail(rc, "%s", why.c_str());
    }
    *out = g;
    return PSIM_OK;
}

PSIM_API void psim_group_destroy(psim_group *g)
{
    if (!g) return;
    for (psim_handle *h : g->shards) psim_destroy(h);
    delete g;
}

PSIM_API int psim_group_update(psim_group *g, const psim_sim_params *params)
{
    if (!g) return psim_fail(PSIM_ERR_INVALID, "null group");
    for (psim_handle *h : g->shards) {
        const int rc = psim_update(h, params);
        if (rc != PSIM_OK) return rc;
    }
    return PSIM_OK;
}

PSIM_API int psim_group_update_batch(psim_group *g, const psim_sim_params *params_host,
                                     uint32_t n_steps)
{
    if (!g) return psim_fail(PSIM_ERR_INVALID, "null group");
    for (psim_handle *h : g->shards) {
        const int rc = psim_update_batch(h, params_host, n_steps);
        if (rc != PSIM_OK) return rc;
    }
    return PSIM_OK;
}

PSIM_API int psim_group_update_multi_host(psim_group *g, const psim_sim_params *params,
                                          const psim_attractor *attractors_host,
                                          uint32_t n_attractors)
{
    if (!g) return psim_fail(PSIM_ERR_INVALID, "null group");
    for (psim_handle *h : g->shards) {
        const int rc = psim_update_multi_host(h, params, attractors_host, n_attractors);
        if (rc != PSIM_OK) return rc;
    }
    return PSIM_OK;
}

PSIM_API int psim_group_sync(psim_group *g)
{
    if (!g) return psim_fail(PSIM_ERR_INVALID, "null group");
    for (psim_handle *h : g->shards) {
        const int rc = psim_sync(h);
        if (rc != PSIM_OK) return rc;
    }
    return PSIM_OK;
}

PSIM_API int psim_group_reset(psim_group *g, int generation_mode)
{
    if (!g) return psim_fail(PSIM_ERR_INVALID, "null group");
    return for_each_concurrently(g->shards.size(), [&](size_t k) {
        return psim_reset(g->shards[k], generation_mode);
    });
}

PSIM_API uint64_t psim_group_particle_count(const psim_group *g) { return g ? g->total : 0; }

PSIM_API uint32_t psim_group_shards(const psim_group *g)
{
    return g ? (uint32_t)g->shards.size() : 0;
}

PSIM_API psim_handle *psim_group_shard(psim_group *g, uint32_t index)
{
    return g && index < g->shards.size() ? g->shards[index] : nullptr;
}

PSIM_API int psim_group_download_aos(psim_group *g, psim_particle *dst_host, uint64_t first,
                                     uint64_t n)
{
    if (!g) return psim_fail(PSIM_ERR_INVALID, "null group");
    int rc = check_group_range(g, first, n);
    if (rc != PSIM_OK) return rc;
    if (n && !dst_host) return psim_fail(PSIM_ERR_INVALID, "null destination");
    return for_overlapping_shards(g, first, n,
                                  [dst_host](psim_handle *h, uint64_t f, uint64_t m, uint64_t off) {
                                      return psim_download_aos(h, dst_host + off, f, m);
                                  });
}

PSIM_API int psim_group_upload_aos(psim_group *g, const psim_particle *src_host, uint64_t first,
                                   uint64_t n)
{
    if (!g) return psim_fail(PSIM_ERR_INVALID, "null group");
    int rc = check_group_range(g, first, n);
    if (rc != PSIM_OK) return rc;
    if (n && !src_host) return psim_fail(PSIM_ERR_INVALID, "null source");
    return for_overlapping_shards(g, first, n,
                                  [src_host](psim_handle *h, uint64_t f, uint64_t m, uint64_t off) {
                                      return psim_upload_aos(h, src_host + off, f, m);
                                  });
}

PSIM_API int psim_group_checksum(psim_group *g, uint64_t out[6])
{
    if (!g) return psim_fail(PSIM_ERR_INVALID, "null group");
    if (!out) return psim_fail(PSIM_ERR_INVALID, "null output");
    for (int c = 0; c < 6; ++c) out[c] = 0;
    std::vector<uint64_t> part;
    try {
        part.assign(6 * g->shards.size(), 0);
    } catch (...) {
        return psim_fail(PSIM_ERR_NOMEM, "host allocation failed");
    }
    const int rc = for_each_concurrently(g->shards.size(), [&](size_t k) {
        return psim_checksum(g->shards[k], &part[6 * k]);
    });
    if (rc != PSIM_OK) return rc;
    for (size_t k = 0; k < g->shards.size(); ++k)
        for (int c = 0; c < 6; ++c) out[c] += part[6 * k + c];
    return PSIM_OK;
}

PSIM_API int psim_group_checkpoint_write(psim_group *g, const char *path, uint64_t step)
{
    if (!g) return psim_fail(PSIM_ERR_INVALID, "null group");
    if (!path) return psim_fail(PSIM_ERR_INVALID, "null path");
    int rc = psim_checkpoint_begin(path);        // no valid header while the records change
    if (rc != PSIM_OK) return rc;
    rc = for_each_concurrently(g->shards.size(), [&](size_t k) {
        return psim_checkpoint_write(g->shards[k], path, step);
    });
    if (rc != PSIM_OK) return rc;
    return psim_checkpoint_commit(g->shards[0], path, step);
}

PSIM_API int psim_group_checkpoint_read(psim_group *g, const char *path, uint64_t *step_out)
{
    if (!g) return psim_fail(PSIM_ERR_INVALID, "null group");
    if (!path) return psim_fail(PSIM_ERR_INVALID, "null path");
    std::vector<uint64_t> steps;
    try {
        steps.assign(g->shards.size(), 0);
    } catch (...) {
        return psim_fail(PSIM_ERR_NOMEM, "host allocation failed");
    }
    const int rc = for_each_concurrently(g->shards.size(), [&](size_t k) {
        return psim_checkpoint_read(g->shards[k], path, &steps[k]);
    });
    if (rc != PSIM_OK) return rc;
    if (step_out && !steps.empty()) *step_out = steps[0];
    return PSIM_OK;
}
