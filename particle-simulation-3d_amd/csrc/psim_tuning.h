// psim_tuning.h -- overrides of the measured launch shapes, for experiments and for tests that
// must force a shape the library would not pick at their size (internal).
//
// Nothing here is read from the environment of an ordinary caller: every override is -1
// ("automatic") unless the process sets it through psim_debug_set_tuning(), or starts with
// PSIM_EXPERIMENT=1 in its environment, in which case the PSIM_<NAME> variables are read ONCE,
// on first use, into the same table.
#ifndef PSIM_TUNING_H
#define PSIM_TUNING_H

namespace psim {

enum Knob : int {
    KNOB_NT = 0,               // update kernels: 0 plain | 1 streaming | 2 cache-resident accesses
    KNOB_RESERVE_LDS,          // update kernels: bytes of LDS reserved per workgroup
    KNOB_MAX_BLOCKS,           // update kernels: grid cap (0 = one pass)
    KNOB_SKEW_BYTES,           // bytes between consecutive state arrays of a new slab
    KNOB_QUEUE_FROM,           // host list length from which the pair queue is used (0 = never)
    KNOB_AOS_STREAM,           // transposes: 0 plain | 1 streaming accesses
    KNOB_AOS_RESERVE,          // transposes: bytes of LDS reserved per workgroup
    KNOB_AOS_BLOCK,            // transposes: 64 or 256 threads per workgroup
    KNOB_AOS_PPL,              // transposes: 1, 2 or 4 particles per lane
    KNOB_CHECKSUM_BLOCKS,      // checksum: workgroups
    KNOB_EXCHANGE_EVENT_FENCES,// exchange: bit 0 kernel_done, bit 1 gathered without system fence
    KNOB_EXCHANGE_COMM_PRIORITY,// exchange: 1 = communication stream at the highest priority (automatic: normal)
    KNOB_COUNT
};

// The override, or -1 for "automatic".
int knob(Knob k);

// Sets the override called `name` (the lower-case enumerator without KNOB_, e.g. "aos_ppl");
// -1 restores automatic.  Returns false for an unknown name.
bool set_knob(const char *name, int value);

}  // namespace psim

#endif
