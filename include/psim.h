/*
 * psim.h -- C ABI of the MI355X (gfx950) particle-update backend.
 *
 * This is the boundary a host program binds to replace the reference's
 * `trait ParticleSimulation` (reference: src/simulation/mod.rs:21-50) for the
 * per-frame particle update.  Every entry point is `extern "C"`, takes plain
 * pointers / integers only, returns a status code (0 = OK) and never throws or
 * aborts.  A handle is NOT thread-safe: one host thread drives one handle, which
 * is the reference's calling model (the UI thread, src/app.rs:306).
 *
 * Device layout: structure-of-arrays.  Six fp32 arrays px,py,pz,vx,vy,vz hold the
 * live physics state (24 B/particle); colour (RGBA), initial colour (RGBA) and the
 * two padding floats of the reference's 64-byte record live in side arrays that
 * the update kernel touches only when color_mode is 1 or 2.  The reference's
 * 64-byte AoS `Particle` is the import/export format (psim_upload_aos /
 * psim_download_aos).
 */
#ifndef PSIM_H
#define PSIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- plain-old-data records shared with the reference --------------------- */

/* Field-for-field the reference's `#[repr(C)] struct SimParams`
 * (src/simulation/mod.rs:52-67); 48 bytes, align 4. */
typedef struct psim_sim_params {
    float    delta_time;          /* @0  */
    float    gravity;             /* @4  */
    uint32_t color_mode;          /* @8   0 keep colour, 1 speed, 2 distance */
    float    mouse_force;         /* @12 */
    float    mouse_radius;        /* @16 */
    uint32_t is_mouse_dragging;   /* @20  attractor active when > 0 */
    float    damping;             /* @24 */
    float    max_dist_for_color;  /* @28 */
    float    mouse_position[3];   /* @32  attractor position */
    uint32_t _padding2;           /* @44 */
} psim_sim_params;

/* Field-for-field the reference's `#[repr(C)] struct Particle`
 * (src/simulation/mod.rs:86-98); 64 bytes. */
typedef struct psim_particle {
    float position[3];       /* @0  */
    float padding1;          /* @12 */
    float velocity[3];       /* @16 */
    float padding2;          /* @28 */
    float color[4];          /* @32 */
    float initial_color[4];  /* @48 */
} psim_particle;

/* One attractor of the multi-attractor extension (no counterpart in the
 * reference, which has exactly one attractor: the mouse_* fields of SimParams,
 * src/simulation/mod.rs:58-65).  With one active attractor whose fields equal
 * mouse_position / mouse_force / mouse_radius the result is identical to
 * psim_update.  32 bytes. */
typedef struct psim_attractor {
    float    position[3];  /* @0  */
    float    force;        /* @12 */
    float    radius;       /* @16 */
    uint32_t active;       /* @20  applied when > 0 */
    uint32_t _pad[2];      /* @24 */
} psim_attractor;

#define PSIM_MAX_ATTRACTORS 64u

/* Raw device pointers of a handle's SoA storage (valid until the next
 * psim_resize / psim_destroy).  Replaces `get_particle_buffer`
 * (src/simulation/mod.rs:44) for consumers that read device memory directly. */
typedef struct psim_soa_view {
    float *px, *py, *pz;     /* count floats each */
    float *vx, *vy, *vz;     /* count floats each */
    float *color;            /* 4*count floats, RGBA interleaved */
    float *initial_color;    /* 4*count floats, RGBA interleaved */
    float *padding;          /* 2*count floats: {padding1, padding2} */
    uint32_t count;          /* active particles */
    uint32_t capacity;       /* allocated particles (>= count) */
} psim_soa_view;

typedef struct psim_handle psim_handle;

/* ---- enums ---------------------------------------------------------------- */

/* Status codes.  psim_last_error() holds the text of the most recent failure on
 * the calling thread. */
enum {
    PSIM_OK            = 0,
    PSIM_ERR_INVALID   = 1,   /* bad argument (null, out of range, misaligned) */
    PSIM_ERR_HIP       = 2,   /* a HIP runtime call failed */
    PSIM_ERR_NOMEM     = 3,   /* device or host allocation failed */
    PSIM_ERR_NO_DEVICE = 4,   /* no usable gfx950 device */
    PSIM_ERR_IO        = 5,   /* a file could not be opened, read or written */
    PSIM_ERR_COMM      = 6    /* RCCL could not be loaded, or a collective call failed */
};

/* `enum SphereGeneration` (src/simulation/mod.rs:15-19) plus the synthetic
 * benchmark distribution. */
enum {
    PSIM_GEN_HOLLOW  = 0,  /* Fibonacci sphere, r = 50 (mod.rs:150-167) */
    PSIM_GEN_FILLED  = 1,  /* seeded uniform ball, r = 50 (mod.rs:168-188) */
    PSIM_GEN_UNIFORM = 2   /* positions U[-50,50)^3, velocities U[-1,1)^3,
                              counter-based SplitMix64, seed 69 (generated on the
                              device; index-addressed so shards are reproducible) */
};

/* `enum SimulationMethod` (src/simulation/mod.rs:9-13) plus this backend. */
enum {
    PSIM_METHOD_CPU            = 0,
    PSIM_METHOD_COMPUTE_SHADER = 1,
    PSIM_METHOD_HIP            = 2
};

/* What a step with color_mode other than 1 or 2 does to the colour.  The reference's
 * two backends disagree: the CPU path keeps the current colour
 * (src/simulation/cpu.rs:106), the compute shader resets it to the initial colour
 * (src/shaders/compute.wgsl:75-77, 90-92).  They differ only after modes 1/2 have been
 * visited. */
enum {
    PSIM_COLOR_KEEP  = 0,  /* CPU-path semantics (default: the CPU path is the oracle) */
    PSIM_COLOR_RESET = 1   /* compute-shader semantics */
};

/* What `resize_buffer` does.  The reference's two backends disagree here as well: the
 * compute backend regenerates all particles for the new count
 * (src/simulation/compute.rs:146-193), the CPU backend keeps what it has, appends a fresh set
 * of particles when it grows beyond everything generated so far and merely lowers the
 * active count when it shrinks (src/simulation/cpu.rs:123-158). */
enum {
    PSIM_RESIZE_REGENERATE = 0,  /* compute-backend semantics (default) */
    PSIM_RESIZE_APPEND     = 1   /* CPU-backend semantics */
};

/* ---- lifecycle ------------------------------------------------------------ */

/* Replaces `ParticleSimulation::new` (src/simulation/mod.rs:22-29; call sites
 * src/app.rs:106-125, 182-195).  The wgpu device / surface-format arguments are
 * dropped; `device_id` selects the GPU.  Allocates SoA storage for `count`
 * particles (at most PSIM_MAX_PARTICLES), fills it with `generation_mode`, creates the
 * handle's stream. */
#define PSIM_MAX_PARTICLES 0xFFFFFC00u   /* counts are u32 like the reference's; storage is
                                            rounded up to multiples of 1024 particles */
int psim_create(psim_handle **out, uint32_t count, int generation_mode, int device_id);

/* Same, for one shard of a larger particle array that is partitioned across
 * GPUs: this handle owns global particles [global_first, global_first + count)
 * of `global_count`.  Generators address particles by global index, so the union
 * of the shards equals the single-GPU array.  No counterpart in the reference
 * (single device). */
int psim_create_shard(psim_handle **out, uint32_t count, uint64_t global_first,
                      uint64_t global_count, int generation_mode, int device_id);

/* Drop of the simulation object. */
void psim_destroy(psim_handle *h);

/* ---- the hot path --------------------------------------------------------- */

/* Replaces `ParticleSimulation::update` (src/simulation/mod.rs:30-36; call site
 * src/app.rs:306-307).  One semi-implicit Euler step for all active particles
 * with the arithmetic of src/simulation/cpu.rs:65-113.  Asynchronous like the
 * reference's compute backend (src/simulation/compute.rs:124-144): the kernel is
 * enqueued on the handle's stream and the call returns; `*params` is copied
 * before returning.  Like the reference, the paused flag is NOT consulted here
 * (the caller gates on is_paused, src/app.rs:283).  The call only enqueues kernels whose
 * arguments travel by value, so it may be made while the handle's stream (psim_set_stream)
 * is being captured into a HIP graph; replaying the graph repeats the recorded steps.
 * (Under PSIM_COLOR_RESET the colour copy is decided when the call is made, that is at
 * capture: capture the steps of a graph in the colour state they will be replayed in.) */
int psim_update(psim_handle *h, const psim_sim_params *params);

/* Multi-attractor step: gravity, then for a = 0..n_attractors-1 in index order
 * the reference's force law (src/simulation/cpu.rs:76-83) with attractor a's
 * position / force / radius, then integrate / damp / colour as psim_update.
 * `attractors_device` is DEVICE memory (e.g. the receive buffer of an RCCL
 * all-gather enqueued earlier on the same stream); it is read when the kernel
 * runs.  The mouse_* fields and is_mouse_dragging of `params` are ignored.
 * n_attractors <= PSIM_MAX_ATTRACTORS.  The kernel tests all (particle, attractor) pairs
 * first and evaluates the force law only for the pairs in range, gathered wave by wave;
 * the velocity terms are added per particle in index order, so the result is that of the
 * plain loop, bit for bit. */
int psim_update_multi(psim_handle *h, const psim_sim_params *params,
                      const psim_attractor *attractors_device, uint32_t n_attractors);

/* Same with the attractor list in HOST memory: copied by value into the launch,
 * no device buffer or synchronisation needed.  Lists of more than 8 (up to
 * PSIM_MAX_ATTRACTORS) do not fit the kernel arguments: they are copied through one of four
 * page-locked slots of the handle into its device scratch, in stream order; the call
 * returns at once unless all four slots still wait for their copies. */
int psim_update_multi_host(psim_handle *h, const psim_sim_params *params,
                           const psim_attractor *attractors_host, uint32_t n_attractors);

/* psim_update_multi_host for callers that cannot allocate: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference.
 * The call allocates nothing, on the host or the device, and moves the list without a copy
 * engine: it (at most PSIM_MAX_ATTRACTORS records; more is PSIM_ERR_INVALID) is written by small kernels
 * that carry it in their arguments into a static device array of the library -- 2 KiB that
 * every GPU the library runs on holds until the process ends -- and the update kernel stages
 * it from there into LDS.  The result is that of psim_update_multi_host, bit for bit.  All
 * handles on one GPU share that array: calls on one stream follow each other safely, calls
 * that overlap on different streams of one GPU race for it -- and so do two graphs holding
 * such calls when they are captured or replayed at the same time on one GPU.  Under
 * PSIM_COLOR_RESET a step in a mode other than 1 / 2 that follows changed colours also
 * enqueues one device-to-device copy of count x 16 bytes (the colour reset of psim_update),
 * and whether it does is decided when the call is made, i.e. at capture: capture a graph in
 * the colour state it will be replayed in.  Asynchronous; the host list is copied before
 * returning. */
int psim_update_multi_host_noalloc(psim_handle *h, const psim_sim_params *params,
                                   const psim_attractor *attractors_host, uint32_t n_attractors);

/* Applies n_steps consecutive updates, params_host[0] first, with the state crossing
 * HBM once per 64 steps instead of once per step: each lane loads its particles,
 * steps them in registers with one parameter set after the other, and stores them.
 * Particles never read other particles (src/simulation/cpu.rs:65-113), so the result
 * is identical, bit for bit, to n_steps calls of psim_update with the same
 * parameters.  No counterpart in the reference, whose caller learns a frame's
 * delta_time only when the frame starts (src/app.rs:204-207); meant for replay and
 * headless runs whose parameter stream is known ahead.  Asynchronous; params_host
 * is copied before returning. */
int psim_update_batch(psim_handle *h, const psim_sim_params *params_host, uint32_t n_steps);

/* Stateless form of psim_update over caller-owned device arrays (each `count`
 * floats; `color` may be null when params->color_mode is 0, else 4*count
 * floats), enqueued on `hip_stream` (null = the default stream) of the current
 * device.  For callers that keep particle state in their own allocations. */
int psim_update_soa(float *px, float *py, float *pz, float *vx, float *vy, float *vz,
                    float *color, uint32_t count, const psim_sim_params *params,
                    void *hip_stream);

/* Writes one attractor record into device memory at `dst_device` (4-byte
 * aligned, on the current device) from a host value, asynchronously on
 * `hip_stream` (null = the default stream).  The value travels in the kernel
 * arguments, so the host copy may be reused immediately and no pinned memory is
 * needed.  Used to stage this rank's attractor into the send buffer of the
 * all-gather on the communication stream. */
int psim_store_attractor(psim_attractor *dst_device, const psim_attractor *value_host,
                         void *hip_stream);

/* Blocks until everything enqueued on the handle's stream has finished.  The
 * reference's equivalent is the caller's queue.submit + present (src/app.rs:313). */
int psim_sync(psim_handle *h);

/* ---- state management ----------------------------------------------------- */

/* Replaces `resize_buffer` (src/simulation/mod.rs:37-43; call site
 * src/app.rs:479-491) with the compute backend's semantics
 * (src/simulation/compute.rs:146-193): records the generation mode; no-op when
 * new_count equals the current count; otherwise regenerates ALL particles for
 * new_count (state is lost); storage grows when needed and is kept on shrink. */
int psim_resize(psim_handle *h, uint32_t new_count, int generation_mode);

/* Selects what psim_resize does from now on (PSIM_RESIZE_*).  Under PSIM_RESIZE_APPEND the
 * handle behaves like the CPU backend's particle vector: growing past everything generated
 * so far appends generate(new_count - generated, mode) -- a fresh set of that many
 * particles -- behind the existing ones, whose state is kept (storage grows with a device
 * copy); shrinking lowers the active count only; growing again within what was generated
 * brings the same particles back as they were left.  psim_reset regenerates the active
 * count and forgets the rest (cpu.rs:172-186).  Not available on shards. */
int psim_set_resize_semantics(psim_handle *h, int semantics);
int psim_resize_semantics(const psim_handle *h);

/* Replaces `reset` (src/simulation/mod.rs:47; call site src/app.rs:336-343):
 * regenerates the current count of particles with `generation_mode`. */
int psim_reset(psim_handle *h, int generation_mode);

/* `get_particle_count` (src/simulation/mod.rs:46). */
uint32_t psim_particle_count(const psim_handle *h);

/* `get_method` (src/simulation/mod.rs:45); always PSIM_METHOD_HIP. */
int psim_method(const psim_handle *h);

/* `is_paused` / `set_paused` (src/simulation/mod.rs:48-49): a plain flag the
 * caller honours. */
int  psim_is_paused(const psim_handle *h);
void psim_set_paused(psim_handle *h, int paused);

/* Selects PSIM_COLOR_KEEP (src/simulation/cpu.rs:106) or PSIM_COLOR_RESET
 * (src/shaders/compute.wgsl:75-77, 90-92) for the updates that follow.  Under RESET the
 * colour array equals the initial colours after every step whose mode is not 1 or 2 -- as
 * with the shader -- but the copy is made only when colours were actually changed since
 * (by a mode 1/2 step, an upload, or a caller holding the raw pointers of
 * psim_device_soa), so steady mode-0 stepping still moves 48 B per particle.  The stateless
 * psim_update_soa has no initial colours and always keeps. */
int psim_set_color_semantics(psim_handle *h, int semantics);
int psim_color_semantics(const psim_handle *h);

/* ---- particle data in and out (replaces `get_particle_buffer`,
 *      src/simulation/mod.rs:44, whose consumer is the renderer's 64-byte
 *      instance layout, src/renderer.rs:32-67) ------------------------------- */

/* Copies particles [first, first+n) to / from HOST memory in the reference's
 * 64-byte AoS layout (transposed on the device).  Synchronous. */
int psim_download_aos(psim_handle *h, psim_particle *dst_host, uint64_t first, uint64_t n);
int psim_upload_aos(psim_handle *h, const psim_particle *src_host, uint64_t first, uint64_t n);

/* Page-locked host memory for the copies above.  The CPU backend of the
 * reference keeps a host-side `Vec<Particle>` that it hands to the renderer every
 * frame (src/simulation/cpu.rs:8, :116-120); a caller that stages frames through
 * host memory the same way should take that buffer from here: copies into and out
 * of page-locked memory run at the full host-link rate and can be asynchronous;
 * pageable memory is accepted everywhere too and is moved through page-locked bounce
 * slots inside the library (slower, synchronous, and it costs host threads).  Free
 * with psim_host_free (NULL is allowed). */
int  psim_host_alloc(void **out, uint64_t bytes);
void psim_host_free(void *ptr);

/* psim_download_aos without the wait: enqueues the transposition and the copy on the
 * handle's stream and returns.  `dst_pinned` must come from psim_host_alloc and stay
 * untouched until psim_sync; updates enqueued afterwards run behind the copy and do not
 * disturb it.  This is how a frame loop that stages through host memory
 * (src/simulation/cpu.rs:116-120) overlaps the transfer of frame k with its own work. */
int psim_download_aos_async(psim_handle *h, psim_particle *dst_pinned, uint64_t first, uint64_t n);

/* Waits for the most recent psim_download_aos_async into `dst_pinned` -- and for nothing that
 * was enqueued after it.  This is what lets a staged frame loop run one frame ahead: per
 * frame it enqueues psim_update and the download of frame k into one of two page-locked
 * buffers, THEN waits for the download of frame k-1 into the other and hands that to the
 * renderer (queue.write_buffer, src/simulation/cpu.rs:116-120) while the device is already
 * working on frame k; the renderer sees every frame one frame late.  The handle tracks its
 * four most recent destinations; PSIM_ERR_INVALID when none into `dst_pinned` is pending. */
int psim_download_aos_wait(psim_handle *h, const psim_particle *dst_pinned);

/* Copies positions and velocities of particles [first, first+n) to / from HOST
 * memory as six consecutive arrays of n floats: px | py | pz | vx | vy | vz.
 * Synchronous. */
int psim_download_soa(psim_handle *h, float *dst_host, uint64_t first, uint64_t n);
int psim_upload_soa(psim_handle *h, const float *src_host, uint64_t first, uint64_t n);

/* Transposes particles [first, first+n) into / out of a DEVICE buffer in the
 * 64-byte AoS layout (e.g. a vertex buffer shared with a renderer), on the
 * handle's stream, asynchronously. */
int psim_export_aos_device(psim_handle *h, psim_particle *dst_device, uint64_t first, uint64_t n);
int psim_import_aos_device(psim_handle *h, const psim_particle *src_device, uint64_t first, uint64_t n);

/* Raw SoA device pointers. */
int psim_device_soa(psim_handle *h, psim_soa_view *view);

/* The device-resident replacement of `get_particle_buffer` (src/simulation/mod.rs:44).  The
 * reference's compute backend keeps the particle buffer on the device and the renderer draws
 * straight from it (src/simulation/compute.rs:28-34, src/custom_renderer.rs:36-38, instance
 * layout src/renderer.rs:32-67): no frame crosses the host link.  This call gives a renderer
 * in another API (Vulkan under wgpu, or another process) the same thing: a device buffer of
 * the handle's capacity in 64-byte `Particle` records, owned by the handle, together with a
 * NEW dma-buf file descriptor that refers to it (the caller closes it after importing; Vulkan
 * imports it with VK_EXT_external_memory_dma_buf, HIP with hipImportExternalMemory as an
 * opaque descriptor).  Fill it whenever a frame is wanted with
 *     psim_export_aos_device(h, *device_ptr_out, 0, psim_particle_count(h))
 * which runs on the handle's stream behind the updates; the importer must not read before that
 * has finished (psim_sync, or an event on psim_get_stream).  The buffer lives until
 * psim_destroy or until the set outgrows it: after a psim_resize that enlarged the storage,
 * call again and import the new descriptor (an importer's mapping of the old memory stays
 * valid, and stale, until it lets go).  The buffer is a physical allocation of its own
 * (hipMemCreate), so the descriptor refers to exactly these records from offset 0;
 * *bytes_out is its size, the records rounded up to the allocation granularity.
 * PSIM_ERR_HIP when the runtime cannot allocate or export such memory. */
int psim_shared_aos(psim_handle *h, psim_particle **device_ptr_out, int *dmabuf_fd_out,
                    uint64_t *bytes_out);

/* Order-independent checksums of the active state, computed on the device with
 * wavefront reductions: out[k] = sum over particles of the uint32 bit pattern of
 * component k (k = 0..5 : px,py,pz,vx,vy,vz) as a uint64 (mod 2^64).
 * Synchronous.  No counterpart in the reference; lets a caller verify a
 * multi-gigabyte state without downloading it. */
int psim_checksum(psim_handle *h, uint64_t out[6]);

/* ---- checkpoints ---------------------------------------------------------- */

/* A checkpoint file is this 64-byte header followed by the particles of the whole set as
 * 64-byte `Particle` records (src/simulation/mod.rs:86-98) in index order -- the layout
 * every consumer of the reference's particle buffer already reads.  All fields
 * little-endian.  The reference has no persistence of its own; this is what its record
 * layout makes cheap. */
typedef struct psim_checkpoint_header {
    char     magic[8];          /* "PSIMCKP1" */
    uint32_t version;           /* 1 */
    uint32_t record_bytes;      /* 64 */
    uint64_t particle_count;    /* particles in the whole set (all shards) */
    uint64_t step;              /* the caller's step counter, stored and returned as is */
    uint32_t generation_mode;   /* the mode the handle last generated with */
    uint32_t color_semantics;   /* PSIM_COLOR_* of the writing handle */
    uint8_t  reserved[24];      /* zero */
} psim_checkpoint_header;

/* Writes the handle's particles to `path` (created when missing), streaming them through
 * page-locked memory while the file writes proceed.  A plain handle invalidates the header
 * first and writes it last, so an interrupted write leaves no valid file, and truncates the
 * file to its exact size.  A shard invalidates the header too, then writes only its own
 * records, at their place in the whole-set file, and never a valid header (`step` is
 * ignored); the shard that owns the last particle sets the file size.  Shards -- in one
 * process or many -- follow
 *     psim_checkpoint_begin(path)            once, before any shard writes
 *     psim_checkpoint_write(shard, path, 0)  every shard, any order, also at the same time
 *     psim_checkpoint_commit(shard, path, step)   once, after every shard has returned
 * with the caller's own barrier between the stages, so that a run interrupted anywhere
 * before the commit leaves a file every reader rejects (psim_group_checkpoint_write does
 * exactly this).  Data goes through the page cache; no fsync.  Synchronous. */
int psim_checkpoint_write(psim_handle *h, const char *path, uint64_t step);

/* Creates `path` when missing and invalidates its header.  Needs no device. */
int psim_checkpoint_begin(const char *path);

/* Writes the header of the whole set `h` belongs to (any of its shards, or a plain handle)
 * after checking that the file has exactly the size of that set; fails with PSIM_ERR_IO
 * when it has not, i.e. when the shard owning the last particle has not written.  The size
 * is a sanity check only -- a file that existed before already has it: that EVERY shard has
 * written is what the caller's barrier before this call guarantees.  Touches no device
 * memory. */
int psim_checkpoint_commit(const psim_handle *h, const char *path, uint64_t step);

/* Restores the handle's particles from `path`: the header's particle_count must equal the
 * handle's whole-set count (so a file written by one sharding restores into any other).
 * `step_out` (optional) receives the stored step counter.  The handle's generation mode and
 * colour semantics are left as they are.  Stepping on from a restored state gives the
 * results the writing run would have got.  Synchronous. */
int psim_checkpoint_read(psim_handle *h, const char *path, uint64_t *step_out);

/* Reads and validates the header of `path` only; needs no device. */
int psim_checkpoint_info(const char *path, psim_checkpoint_header *out);

/* ---- streams -------------------------------------------------------------- */

/* Makes the handle enqueue on a caller-owned hipStream_t (e.g. the stream an
 * RCCL collective is ordered on).  Null restores the handle's own stream. */
int   psim_set_stream(psim_handle *h, void *hip_stream);
void *psim_get_stream(const psim_handle *h);

/* Device ordinal the handle lives on. */
int psim_device(const psim_handle *h);

/* HIP devices this process sees (0 when there is none): the ordinals a handle, a shard or
 * the device_ids of a group may name. */
int psim_device_count(void);

/* ---- timing ---------------------------------------------------------------- */

/* Device-side timer around work enqueued on the handle's stream.  The reference
 * times `update` with a host wall clock (src/app.rs:303-315), which for its GPU
 * backend measures command encoding only; because psim_update is asynchronous the
 * equivalent here is a pair of stream events.  psim_timer_start records the start
 * event; psim_timer_stop records the stop event, waits for it and returns the
 * milliseconds the stream spent between the two. */
int psim_timer_start(psim_handle *h);
int psim_timer_stop(psim_handle *h, float *elapsed_ms);

/* The launch shape the library picks for an update of `count` particles with
 * `n_attractors` attractors (1 for psim_update).  The reference hard-codes one shape,
 * `dispatch_workgroups(count.div_ceil(256))` of 256-thread workgroups
 * (src/simulation/compute.rs:142-143, src/shaders/compute.wgsl:31); here it depends on
 * whether the state fits the L2s, the Infinity Cache or neither.  Needs no device. */
typedef struct psim_launch_shape_info {
    int      access;              /* 0 plain accesses, 1 plain loads + sc1 stores
                                     (cache-resident), 2 nt loads + nt sc1 stores (streaming) */
    uint32_t workgroup_size;      /* threads per workgroup */
    uint32_t workgroups;          /* grid size */
    uint32_t reserve_lds;         /* LDS bytes reserved per workgroup to cap residency, colour mode 0 */
    uint32_t reserve_lds_colour;  /* the same for colour modes 1 / 2 */
    uint32_t resident_per_cu;     /* workgroups a CU's 160 KiB of LDS then holds (0 = not capped) */
} psim_launch_shape_info;
int psim_launch_shape(uint32_t count, uint32_t n_attractors, psim_launch_shape_info *out);

/* What the runtime says about that shape on the handle's device: how many workgroups of
 * the update kernel psim_update (n_attractors 0 or 1) or psim_update_multi (more) would
 * launch for the handle's particle count, with `color_mode`, one CU holds at once
 * (hipOccupancyMaxActiveBlocksPerMultiprocessor with the LDS reservation).  The residency
 * cap is worth a fifth of the update's bandwidth, so a caller -- or a test -- can check that
 * a driver update or a part with another LDS size has not undone it: for a capped shape the
 * result must equal psim_launch_shape's resident_per_cu.  lds_bytes_per_workgroup (optional)
 * receives the LDS one workgroup takes: what the kernel declares plus the reservation.  (A
 * kernel that declares more than the reservation asks for -- colour modes 1 / 2 with a list
 * of 5 to 8 attractors: 11520 B against 10240 -- is held by its own LDS: 14 per CU, not 16.) */
int psim_launch_occupancy(psim_handle *h, uint32_t n_attractors, uint32_t color_mode,
                          uint32_t *workgroups_per_cu, uint32_t *lds_bytes_per_workgroup);

/* ---- several GPUs from one process ---------------------------------------- */

/* A group is one particle set partitioned over several shards, each a psim_handle on a
 * device of the caller's choice, driven by ONE host thread: what a single-process caller
 * like the reference application (one `Box<dyn ParticleSimulation>`, src/app.rs:18) needs
 * to use all GPUs of a node.  (Programs that run one process per GPU create their shard
 * with psim_create_shard instead.)  Shard k owns the contiguous range
 * [k * total / n, (k + 1) * total / n) and lives on device_ids[k]; a device may be named
 * more than once.  Particles never read other particles (src/simulation/cpu.rs:65-113),
 * so a step is one kernel per shard and no data moves between devices.  total_count may
 * exceed 2^32 with PSIM_GEN_UNIFORM; the reference's two generators index with u32. */
typedef struct psim_group psim_group;

int psim_group_create(psim_group **out, uint64_t total_count, const int *device_ids,
                      uint32_t n_shards, int generation_mode);
void psim_group_destroy(psim_group *g);

/* `update` on every shard: the kernels are enqueued device after device and run
 * concurrently; returns without waiting (psim_group_sync waits for all). */
int psim_group_update(psim_group *g, const psim_sim_params *params);
int psim_group_update_batch(psim_group *g, const psim_sim_params *params_host, uint32_t n_steps);
/* The multi-attractor step on every shard with one host list: in a single process every
 * attractor is known to the host, so the several-GPU multi-attractor configuration needs
 * no exchange between devices at all. */
int psim_group_update_multi_host(psim_group *g, const psim_sim_params *params,
                                 const psim_attractor *attractors_host, uint32_t n_attractors);
int psim_group_sync(psim_group *g);

/* `reset` / `get_particle_count` of the whole set. */
int psim_group_reset(psim_group *g, int generation_mode);
uint64_t psim_group_particle_count(const psim_group *g);

/* The shards, for everything per-shard (streams, raw pointers, colour semantics ...).
 * The handles belong to the group: do not destroy or resize them. */
uint32_t psim_group_shards(const psim_group *g);
psim_handle *psim_group_shard(psim_group *g, uint32_t index);

/* Particles [first, first + n) of the whole set as 64-byte records, to / from host memory,
 * gathered from / scattered to the shards that hold them. */
int psim_group_download_aos(psim_group *g, psim_particle *dst_host, uint64_t first, uint64_t n);
int psim_group_upload_aos(psim_group *g, const psim_particle *src_host, uint64_t first, uint64_t n);

/* The checksum of the whole set (sum of the shards' checksums) and a checkpoint of it in
 * the one-file format above (every shard writes / reads its own records). */
int psim_group_checksum(psim_group *g, uint64_t out[6]);
int psim_group_checkpoint_write(psim_group *g, const char *path, uint64_t step);
int psim_group_checkpoint_read(psim_group *g, const char *path, uint64_t *step_out);

/* ---- one process per GPU: the attractor exchange ------------------------------ */

/* The multi-attractor step of a set that is sharded over one process per GPU (each rank a
 * psim_create_shard handle): every rank owns ONE attractor, a step all-gathers the ranks'
 * 32-byte records with RCCL -- over xGMI between the GPUs of a node -- and applies all of
 * them, rank 0's first, with psim_update_multi.  No counterpart in the reference, which has
 * one attractor and one process (src/simulation/mod.rs:65, src/simulation/cpu.rs:58); with
 * one rank a step equals psim_update with that rank's attractor.  The single-attractor
 * configurations need none of this: their 48-byte SimParams is replicated by the hosts.
 *
 * The exchange is 32 bytes per rank, bound by latency, so it runs on a stream of its own
 * into a ring of 16 receive lists and overlaps the previous steps' kernels; a step only
 * enqueues work and returns (about 15 us of host time).  librccl.so.1 is loaded on the first call of this section
 * (PSIM_RCCL_LIBRARY names another file), never by the rest of the library. */
#define PSIM_EXCHANGE_ID_BYTES 128
typedef struct psim_exchange psim_exchange;

/* A fresh communicator id (RCCL's ncclUniqueId).  ONE rank makes it; the caller hands the
 * same 128 bytes to every rank over whatever channel it has (MPI, a socket, a file). */
int psim_exchange_unique_id(uint8_t id[PSIM_EXCHANGE_ID_BYTES]);

/* That channel for callers that have none: rank 0 makes an id and publishes it as the file
 * `path` (written under another name, then renamed, so it is never seen incomplete); every
 * other rank waits up to timeout_ms for the file and reads it.  `path` must be fresh for
 * every run (a file left by an earlier run would be taken for this run's) and visible to
 * all ranks, e.g. under /dev/shm for the ranks of one node. */
int psim_exchange_rendezvous_file(const char *path, int rank,
                                  uint8_t id[PSIM_EXCHANGE_ID_BYTES], uint32_t timeout_ms);

/* Joins the communicator `id` as `rank` of `world` (at most PSIM_MAX_ATTRACTORS) on the
 * device of `h`; collective: returns once every rank has called it (RCCL's ncclCommInitRank
 * waits without a limit: a launcher that can lose a rank before this point keeps its own
 * watchdog; psim_exchange_set_timeout covers everything after it).  The handle stays the
 * caller's and must outlive the exchange. */
int psim_exchange_create(psim_exchange **out, psim_handle *h,
                         const uint8_t id[PSIM_EXCHANGE_ID_BYTES], int rank, int world);
void psim_exchange_destroy(psim_exchange *x);

/* One step: this rank's attractor `mine` (host memory, copied before returning) joins the
 * other ranks'; all `world` of them are applied to the handle's particles as
 * psim_update_multi does (mouse_* of `params` ignored).  Collective and asynchronous. */
int psim_exchange_step(psim_exchange *x, const psim_sim_params *params,
                       const psim_attractor *mine);

/* How a step's kernel is ordered behind its all-gather.
 * PSIM_EXCHANGE_DEVICE_PACED (default): the handle's stream waits for the gather's event; the
 * call returns at once, and the wait costs a few microseconds of device time per step.
 * PSIM_EXCHANGE_HOST_PACED: the calling thread waits for the gather -- which runs under the
 * previous step's kernel -- and then enqueues the kernel with nothing in front of it, so kernels
 * run back to back; the call returns once the gather has finished (the kernel still runs
 * asynchronously).  Pays when a step's kernel takes longer than the gather plus the host's
 * work, i.e. for shards of millions of particles -- measured on one rank, where no collective
 * kernel runs.  On several GPUs the collective's workgroup (RCCL's wants most of a CU) only
 * finds room at the moment an update kernel's grid runs dry; device pacing, with the host
 * running ahead, has the next gather waiting at that moment, host pacing enqueues it about
 * when the next kernel starts.  Measure both before switching.  All ranks may choose
 * independently. */
/* PSIM_EXCHANGE_PIPELINED: host pacing one call late, for runs that step on and on (headless
 * ranks): a call enqueues this step's all-gather FIRST, then waits for the PREVIOUS step's
 * gather -- which has had a whole kernel to run under -- and enqueues the previous step's
 * kernel with nothing in front of it.  Kernels run back to back as with host pacing, and, as
 * with device pacing, every gather is already in its queue when the kernel before it ends (with
 * an all-gather kernel of RCCL's size in the test double: 46.8 us per step against 48.4
 * device-paced and 66.1 host-paced at 6.25 M particles; over the real RCCL on one rank 44.8
 * against 47.5 and 44.6).  The price: after psim_exchange_step the handle is ONE
 * STEP BEHIND -- the step's kernel is enqueued by the next psim_exchange_step, by
 * psim_exchange_sync / psim_exchange_last_list, or when the pacing is changed.  Call
 * psim_exchange_sync before touching the handle through its own entry points (psim_sync,
 * transfers, checksums, checkpoints); psim_exchange_destroy does not apply a step that is still
 * outstanding. */
enum {
    PSIM_EXCHANGE_DEVICE_PACED = 0,
    PSIM_EXCHANGE_HOST_PACED   = 1,
    PSIM_EXCHANGE_PIPELINED    = 2
};
int psim_exchange_set_pacing(psim_exchange *x, int pacing);

/* A way out of a collective that will never finish.  By default (0) the exchange waits for
 * ever, like RCCL itself: when a rank dies, the others hang in their next wait.  With a timeout
 * every host-side wait of the exchange -- the gather of a host-paced step, psim_exchange_sync,
 * psim_exchange_last_list, psim_exchange_destroy -- polls instead of blocking, asks the
 * communicator for asynchronous errors (ncclCommGetAsyncError) while it does, and after
 * timeout_ms without completion aborts the communicator (ncclCommAbort) and returns
 * PSIM_ERR_COMM; a device-paced step additionally waits for the gather enqueued 16 steps
 * earlier, so the host never runs more than the ring ahead of a collective that hangs.  After
 * PSIM_ERR_COMM from any call of this section -- a timeout, or a step that failed half-way for
 * any reason -- the ranks are out of step: the exchange is unusable (every further call
 * returns PSIM_ERR_COMM; destroy it, and tear the other ranks down too) and the particles are
 * defined only as far as the steps that had been applied before (a host-paced step that times
 * out has not enqueued its kernel; a device-paced run may have applied kernels to lists the
 * aborted collective never filled: restore from a checkpoint).  The reference's only failure
 * policy is to abort the process (`panic = "abort"`, Cargo.toml:49); a run sharded over
 * processes needs to be able to report instead.  Each rank sets its own timeout. */
int psim_exchange_set_timeout(psim_exchange *x, uint32_t timeout_ms);

/* Enqueues the kernel of a step that pipelined pacing still holds back, then waits for
 * everything the exchange and its handle have enqueued. */
int psim_exchange_sync(psim_exchange *x);

/* The `world` records the most recent step applied, in rank order, into host memory (waits
 * for that step): lets a caller check what the collective delivered. */
int psim_exchange_last_list(psim_exchange *x, psim_attractor *out_host);

int psim_exchange_world(const psim_exchange *x);
int psim_exchange_rank(const psim_exchange *x);

/* ---- debugging -------------------------------------------------------------- */

/* Overrides one of the measured launch-shape choices of this process, for kernel experiments
 * and for tests that must force a shape the library would not pick at their size.  The
 * reference hard-codes its one dispatch shape (src/simulation/compute.rs:142-143) and has no
 * counterpart.  `name` is one of "nt", "reserve_lds", "max_blocks", "skew_bytes" (copied when a
 * handle is created), "queue_from", "aos_stream", "aos_reserve", "aos_block", "aos_ppl",
 * "checksum_blocks", "exchange_event_fences", "exchange_comm_priority" (read at each launch /
 * when an exchange is created); value -1 restores the automatic choice.  This call -- or a process started with
 * the variable PSIM_EXPERIMENT set to "1", which reads PSIM_<NAME> variables once -- is the
 * ONLY way to change a launch shape: no other environment variable is consulted.  Results
 * never depend on the shape, only speed does.  Not thread-safe against launches in flight on
 * other threads. */
int psim_debug_set_tuning(const char *name, int value);

/* ---- errors --------------------------------------------------------------- */

/* Text of the most recent failure on the calling thread ("" if none). */
const char *psim_last_error(void);

/* Library version string. */
const char *psim_version(void);

#ifdef __cplusplus
}
#endif

#endif /* PSIM_H */
