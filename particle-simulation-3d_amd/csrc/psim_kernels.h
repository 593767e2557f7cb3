// psim_kernels.h -- host-callable launchers of the gfx950 kernels (internal).
#ifndef PSIM_KERNELS_H
#define PSIM_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/psim.h"

#ifndef PSIM_BLOCK
#define PSIM_BLOCK 256               /* threads per workgroup = 4 wave64 (all but the update) */
#endif
#ifndef PSIM_UPDATE_BLOCK
#define PSIM_UPDATE_BLOCK 64         /* the update kernel: single-wave workgroups */
#endif
#define PSIM_BATCH_STEPS 64u          /* parameter sets that fit one launch's kernel arguments */
#define PSIM_KERNARG_ATTRACTORS 8u   /* attractor list size that fits the kernel arguments */

namespace psim {

// Device pointers of one particle set (structure of arrays).
struct SoaPtrs {
    float *px, *py, *pz, *vx, *vy, *vz;
    float *color;          // 4 floats per particle (may be null for colour mode 0)
    float *initial_color;  // 4 floats per particle (may be null for the update)
    float *padding;        // 2 floats per particle (may be null for the update)
};

// How a kernel touches the particle state.
enum : int {
    PSIM_ACCESS_PLAIN = 0,      // global loads and stores (the state fits the L2s)
    PSIM_ACCESS_RESIDENT = 1,   // plain loads, sc1 stores (the state fits the Infinity Cache)
    PSIM_ACCESS_STREAM = 2      // nt loads, nt sc1 stores (the state is streamed from HBM)
};

// Launch shape of the update kernel.
struct LaunchTuning {
    uint32_t max_blocks;   // 0 = one pass (grid covers every group), else grid-stride cap
    int access;            // PSIM_ACCESS_*
    uint32_t reserve_lds;  // untouched dynamic LDS bytes per workgroup: caps workgroups per CU
    uint32_t reserve_lds_colour;   // the same for colour modes 1/2 (one more write stream)
};

uint32_t update_grid(uint32_t n, const LaunchTuning &t);

// The measured-best launch shape for n particles and an attractor list of the given length.
LaunchTuning choose_tuning(uint32_t n, uint32_t n_attractors);

// force: 0 none, 1 the SimParams attractor, 2 list in device memory (staged in
// LDS), 3 list copied from host memory into the kernel arguments.
hipError_t launch_update(const SoaPtrs &s, uint32_t n, const psim_sim_params &p, int force,
                         const psim_attractor *attr_dev, const psim_attractor *attr_host,
                         uint32_t na, const LaunchTuning &t, hipStream_t stream);

// Workgroups of that update (same picking of the kernel instantiation and of its dynamic LDS)
// that one CU holds at once according to the runtime, and the LDS one of them takes (what the
// kernel declares plus the reservation); needs a current device.
hipError_t update_occupancy(uint32_t color_mode, int force, uint32_t na, const LaunchTuning &t,
                            int *workgroups_per_cu, uint32_t *lds_bytes);

// nsteps <= PSIM_BATCH_STEPS consecutive single-attractor steps in one pass over the state.
hipError_t launch_update_batch(const SoaPtrs &s, uint32_t n, const psim_sim_params *params,
                               uint32_t nsteps, const LaunchTuning &t, hipStream_t stream);

hipError_t launch_update_scalar(const SoaPtrs &s, uint32_t n, const psim_sim_params &p,
                                uint32_t max_blocks, hipStream_t stream);

hipError_t launch_store_attractor(psim_attractor *dst, const psim_attractor &value,
                                  hipStream_t stream);

// The address, on `device` (which must be current), of the library's static device array
// of PSIM_MAX_ATTRACTORS attractor records; asked once per device and remembered.
hipError_t static_list_address(int device, psim_attractor **out);

// Writes `count` records from host memory into device memory at `dst`, carried in the
// arguments of ceil(count / 8) small kernels: no copy engine, no page-locked staging.
hipError_t launch_store_attractors(psim_attractor *dst, const psim_attractor *values_host,
                                   uint32_t count, hipStream_t stream);

hipError_t launch_gen_uniform(const SoaPtrs &s, uint32_t n, uint64_t global_first, uint64_t seed,
                              hipStream_t stream);

hipError_t launch_export_aos(const SoaPtrs &s, uint32_t first, uint32_t n, psim_particle *aos,
                             hipStream_t stream);
hipError_t launch_import_aos(const SoaPtrs &s, uint32_t first, uint32_t n,
                             const psim_particle *aos, hipStream_t stream);

hipError_t launch_checksum(const SoaPtrs &s, uint32_t n, uint64_t *out_dev, uint32_t max_blocks,
                           hipStream_t stream);

}  // namespace psim

#endif
