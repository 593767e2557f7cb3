//! Raw bindings of `libpsim_hip.so`, the MI355X (gfx950) backend of the particle update.
//!
//! GENERATED from `include/psim.h` by `scripts/gen_rust_sys.py`; do not edit by hand.
//! One declaration per prototype of the header, with the same names.  The records are the
//! reference's own `#[repr(C)]` layouts (`src/simulation/mod.rs:52-67, 86-98`), so
//! `&SimParams` / `&[Particle]` of the application cross the boundary by pointer cast.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

macro_rules! opaque {
    ($name:ident) => {
        #[repr(C)]
        pub struct $name {
            _private: [u8; 0],
        }
    };
}
opaque!(PsimHandle);
opaque!(PsimGroup);
opaque!(PsimExchange);

/// `struct SimParams`, 48 bytes (`src/simulation/mod.rs:52-67`).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct PsimSimParams {
    pub delta_time: f32,
    pub gravity: f32,
    pub color_mode: u32,
    pub mouse_force: f32,
    pub mouse_radius: f32,
    pub is_mouse_dragging: u32,
    pub damping: f32,
    pub max_dist_for_color: f32,
    pub mouse_position: [f32; 3],
    pub _padding2: u32,
}

/// `struct Particle`, 64 bytes (`src/simulation/mod.rs:86-98`).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct PsimParticle {
    pub position: [f32; 3],
    pub padding1: f32,
    pub velocity: [f32; 3],
    pub padding2: f32,
    pub color: [f32; 4],
    pub initial_color: [f32; 4],
}

/// One attractor of the multi-attractor extension, 32 bytes.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct PsimAttractor {
    pub position: [f32; 3],
    pub force: f32,
    pub radius: f32,
    pub active: u32,
    pub _pad: [u32; 2],
}

/// Raw device pointers of a handle's structure-of-arrays storage.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PsimSoaView {
    pub px: *mut f32,
    pub py: *mut f32,
    pub pz: *mut f32,
    pub vx: *mut f32,
    pub vy: *mut f32,
    pub vz: *mut f32,
    pub color: *mut f32,
    pub initial_color: *mut f32,
    pub padding: *mut f32,
    pub count: u32,
    pub capacity: u32,
}

/// The 64 bytes at the start of a checkpoint file.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PsimCheckpointHeader {
    pub magic: [c_char; 8],
    pub version: u32,
    pub record_bytes: u32,
    pub particle_count: u64,
    pub step: u64,
    pub generation_mode: u32,
    pub color_semantics: u32,
    pub reserved: [u8; 24],
}

/// The launch shape the library picks for a particle count.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PsimLaunchShapeInfo {
    pub access: c_int,
    pub workgroup_size: u32,
    pub workgroups: u32,
    pub reserve_lds: u32,
    pub reserve_lds_colour: u32,
    pub resident_per_cu: u32,
}

const _: () = assert!(std::mem::size_of::<PsimSimParams>() == 48);
const _: () = assert!(std::mem::size_of::<PsimParticle>() == 64);
const _: () = assert!(std::mem::size_of::<PsimAttractor>() == 32);
const _: () = assert!(std::mem::size_of::<PsimCheckpointHeader>() == 64);

// ---- constants -------------------------------------------------
pub const PSIM_MAX_ATTRACTORS: u32 = 64;
pub const PSIM_MAX_PARTICLES: u32 = 0xFFFFFC00;
pub const PSIM_EXCHANGE_ID_BYTES: u32 = 128;
pub const PSIM_OK: c_int = 0;
pub const PSIM_ERR_INVALID: c_int = 1;
pub const PSIM_ERR_HIP: c_int = 2;
pub const PSIM_ERR_NOMEM: c_int = 3;
pub const PSIM_ERR_NO_DEVICE: c_int = 4;
pub const PSIM_ERR_IO: c_int = 5;
pub const PSIM_ERR_COMM: c_int = 6;
pub const PSIM_GEN_HOLLOW: c_int = 0;
pub const PSIM_GEN_FILLED: c_int = 1;
pub const PSIM_GEN_UNIFORM: c_int = 2;
pub const PSIM_METHOD_CPU: c_int = 0;
pub const PSIM_METHOD_COMPUTE_SHADER: c_int = 1;
pub const PSIM_METHOD_HIP: c_int = 2;
pub const PSIM_COLOR_KEEP: c_int = 0;
pub const PSIM_COLOR_RESET: c_int = 1;
pub const PSIM_RESIZE_REGENERATE: c_int = 0;
pub const PSIM_RESIZE_APPEND: c_int = 1;
pub const PSIM_EXCHANGE_DEVICE_PACED: c_int = 0;
pub const PSIM_EXCHANGE_HOST_PACED: c_int = 1;
pub const PSIM_EXCHANGE_PIPELINED: c_int = 2;

#[link(name = "psim_hip")]
extern "C" {
    pub fn psim_create(out: *mut *mut PsimHandle, count: u32, generation_mode: c_int, device_id: c_int) -> c_int;
    pub fn psim_create_shard(out: *mut *mut PsimHandle, count: u32, global_first: u64, global_count: u64, generation_mode: c_int, device_id: c_int) -> c_int;
    pub fn psim_destroy(h: *mut PsimHandle);
    pub fn psim_update(h: *mut PsimHandle, params: *const PsimSimParams) -> c_int;
    pub fn psim_update_multi(h: *mut PsimHandle, params: *const PsimSimParams, attractors_device: *const PsimAttractor, n_attractors: u32) -> c_int;
    pub fn psim_update_multi_host(h: *mut PsimHandle, params: *const PsimSimParams, attractors_host: *const PsimAttractor, n_attractors: u32) -> c_int;
    pub fn psim_update_multi_host_noalloc(h: *mut PsimHandle, params: *const PsimSimParams, attractors_host: *const PsimAttractor, n_attractors: u32) -> c_int;
    pub fn psim_update_batch(h: *mut PsimHandle, params_host: *const PsimSimParams, n_steps: u32) -> c_int;
    pub fn psim_update_soa(px: *mut f32, py: *mut f32, pz: *mut f32, vx: *mut f32, vy: *mut f32, vz: *mut f32, color: *mut f32, count: u32, params: *const PsimSimParams, hip_stream: *mut c_void) -> c_int;
    pub fn psim_store_attractor(dst_device: *mut PsimAttractor, value_host: *const PsimAttractor, hip_stream: *mut c_void) -> c_int;
    pub fn psim_sync(h: *mut PsimHandle) -> c_int;
    pub fn psim_resize(h: *mut PsimHandle, new_count: u32, generation_mode: c_int) -> c_int;
    pub fn psim_set_resize_semantics(h: *mut PsimHandle, semantics: c_int) -> c_int;
    pub fn psim_resize_semantics(h: *const PsimHandle) -> c_int;
    pub fn psim_reset(h: *mut PsimHandle, generation_mode: c_int) -> c_int;
    pub fn psim_particle_count(h: *const PsimHandle) -> u32;
    pub fn psim_method(h: *const PsimHandle) -> c_int;
    pub fn psim_is_paused(h: *const PsimHandle) -> c_int;
    pub fn psim_set_paused(h: *mut PsimHandle, paused: c_int);
    pub fn psim_set_color_semantics(h: *mut PsimHandle, semantics: c_int) -> c_int;
    pub fn psim_color_semantics(h: *const PsimHandle) -> c_int;
    pub fn psim_download_aos(h: *mut PsimHandle, dst_host: *mut PsimParticle, first: u64, n: u64) -> c_int;
    pub fn psim_upload_aos(h: *mut PsimHandle, src_host: *const PsimParticle, first: u64, n: u64) -> c_int;
    pub fn psim_host_alloc(out: *mut *mut c_void, bytes: u64) -> c_int;
    pub fn psim_host_free(ptr: *mut c_void);
    pub fn psim_download_aos_async(h: *mut PsimHandle, dst_pinned: *mut PsimParticle, first: u64, n: u64) -> c_int;
    pub fn psim_download_aos_wait(h: *mut PsimHandle, dst_pinned: *const PsimParticle) -> c_int;
    pub fn psim_download_soa(h: *mut PsimHandle, dst_host: *mut f32, first: u64, n: u64) -> c_int;
    pub fn psim_upload_soa(h: *mut PsimHandle, src_host: *const f32, first: u64, n: u64) -> c_int;
    pub fn psim_export_aos_device(h: *mut PsimHandle, dst_device: *mut PsimParticle, first: u64, n: u64) -> c_int;
    pub fn psim_import_aos_device(h: *mut PsimHandle, src_device: *const PsimParticle, first: u64, n: u64) -> c_int;
    pub fn psim_device_soa(h: *mut PsimHandle, view: *mut PsimSoaView) -> c_int;
    pub fn psim_shared_aos(h: *mut PsimHandle, device_ptr_out: *mut *mut PsimParticle, dmabuf_fd_out: *mut c_int, bytes_out: *mut u64) -> c_int;
    pub fn psim_checksum(h: *mut PsimHandle, out: *mut u64) -> c_int;
    pub fn psim_checkpoint_write(h: *mut PsimHandle, path: *const c_char, step: u64) -> c_int;
    pub fn psim_checkpoint_begin(path: *const c_char) -> c_int;
    pub fn psim_checkpoint_commit(h: *const PsimHandle, path: *const c_char, step: u64) -> c_int;
    pub fn psim_checkpoint_read(h: *mut PsimHandle, path: *const c_char, step_out: *mut u64) -> c_int;
    pub fn psim_checkpoint_info(path: *const c_char, out: *mut PsimCheckpointHeader) -> c_int;
    pub fn psim_set_stream(h: *mut PsimHandle, hip_stream: *mut c_void) -> c_int;
    pub fn psim_get_stream(h: *const PsimHandle) -> *mut c_void;
    pub fn psim_device(h: *const PsimHandle) -> c_int;
    pub fn psim_device_count() -> c_int;
    pub fn psim_timer_start(h: *mut PsimHandle) -> c_int;
    pub fn psim_timer_stop(h: *mut PsimHandle, elapsed_ms: *mut f32) -> c_int;
    pub fn psim_launch_shape(count: u32, n_attractors: u32, out: *mut PsimLaunchShapeInfo) -> c_int;
    pub fn psim_launch_occupancy(h: *mut PsimHandle, n_attractors: u32, color_mode: u32, workgroups_per_cu: *mut u32, lds_bytes_per_workgroup: *mut u32) -> c_int;
    pub fn psim_group_create(out: *mut *mut PsimGroup, total_count: u64, device_ids: *const c_int, n_shards: u32, generation_mode: c_int) -> c_int;
    pub fn psim_group_destroy(g: *mut PsimGroup);
    pub fn psim_group_update(g: *mut PsimGroup, params: *const PsimSimParams) -> c_int;
    pub fn psim_group_update_batch(g: *mut PsimGroup, params_host: *const PsimSimParams, n_steps: u32) -> c_int;
    pub fn psim_group_update_multi_host(g: *mut PsimGroup, params: *const PsimSimParams, attractors_host: *const PsimAttractor, n_attractors: u32) -> c_int;
    pub fn psim_group_sync(g: *mut PsimGroup) -> c_int;
    pub fn psim_group_reset(g: *mut PsimGroup, generation_mode: c_int) -> c_int;
    pub fn psim_group_particle_count(g: *const PsimGroup) -> u64;
    pub fn psim_group_shards(g: *const PsimGroup) -> u32;
    pub fn psim_group_shard(g: *mut PsimGroup, index: u32) -> *mut PsimHandle;
    pub fn psim_group_download_aos(g: *mut PsimGroup, dst_host: *mut PsimParticle, first: u64, n: u64) -> c_int;
    pub fn psim_group_upload_aos(g: *mut PsimGroup, src_host: *const PsimParticle, first: u64, n: u64) -> c_int;
    pub fn psim_group_checksum(g: *mut PsimGroup, out: *mut u64) -> c_int;
    pub fn psim_group_checkpoint_write(g: *mut PsimGroup, path: *const c_char, step: u64) -> c_int;
    pub fn psim_group_checkpoint_read(g: *mut PsimGroup, path: *const c_char, step_out: *mut u64) -> c_int;
    pub fn psim_exchange_unique_id(id: *mut u8) -> c_int;
    pub fn psim_exchange_rendezvous_file(path: *const c_char, rank: c_int, id: *mut u8, timeout_ms: u32) -> c_int;
    pub fn psim_exchange_create(out: *mut *mut PsimExchange, h: *mut PsimHandle, id: *const u8, rank: c_int, world: c_int) -> c_int;
    pub fn psim_exchange_destroy(x: *mut PsimExchange);
    pub fn psim_exchange_step(x: *mut PsimExchange, params: *const PsimSimParams, mine: *const PsimAttractor) -> c_int;
    pub fn psim_exchange_set_pacing(x: *mut PsimExchange, pacing: c_int) -> c_int;
    pub fn psim_exchange_set_timeout(x: *mut PsimExchange, timeout_ms: u32) -> c_int;
    pub fn psim_exchange_sync(x: *mut PsimExchange) -> c_int;
    pub fn psim_exchange_last_list(x: *mut PsimExchange, out_host: *mut PsimAttractor) -> c_int;
    pub fn psim_exchange_world(x: *const PsimExchange) -> c_int;
    pub fn psim_exchange_rank(x: *const PsimExchange) -> c_int;
    pub fn psim_debug_set_tuning(name: *const c_char, value: c_int) -> c_int;
    pub fn psim_last_error() -> *const c_char;
    pub fn psim_version() -> *const c_char;
}
