//! `src/simulation/hip.rs` -- the MI355X backend of the application, beside `cpu.rs` and
//! `compute.rs`: `trait ParticleSimulation` (`src/simulation/mod.rs:21-50`) over the C ABI of
//! `libpsim_hip.so` (crate `psim-sys`).  Not compiled in this repository (no Rust toolchain
//! in its environment); every call below is a declaration of `psim-sys/src/lib.rs`, and the
//! call sequence of `update` is the one `tests/cpp/staged_frames.cpp` runs and checks.
//!
//! To wire it in:
//!   * `src/simulation/mod.rs:9-13`: `pub enum SimulationMethod { Cpu, ComputeShader, Hip }`
//!     and `pub mod hip;` beside `pub mod cpu; pub mod compute;`
//!   * `src/app.rs:106-125` and `:182-195`: one more arm,
//!     `SimulationMethod::Hip => Box::new(HipParticleSimulation::new(device, count, format, mode))`
//!   * `src/app.rs:87-93`: push `SimulationMethod::Hip` when `psim_device_count() > 0`
//! Everything else goes through the trait object (`app.rs:283, 306, 339, 346-348, 485, 602`).
use super::{Particle, ParticleSimulation, SimParams, SimulationMethod, SphereGeneration};
use psim_sys::*;
use std::os::raw::{c_int, c_void};

fn generation(mode: SphereGeneration) -> c_int {
    match mode {
        SphereGeneration::Hollow => PSIM_GEN_HOLLOW,
        SphereGeneration::Filled => PSIM_GEN_FILLED,
    }
}

fn check(rc: c_int) {
    if rc != PSIM_OK {
        let msg = unsafe { std::ffi::CStr::from_ptr(psim_last_error()) };
        // the reference's backends panic on wgpu validation errors; release builds abort
        panic!("psim error {rc}: {}", msg.to_string_lossy());
    }
}

/// The renderer draws from a wgpu vertex buffer in the 64-byte Particle layout
/// (src/renderer.rs:32-67).  Until that buffer is shared with HIP, frames are staged
/// through host memory as the CPU backend does (src/simulation/cpu.rs:116-120) -- through
/// TWO page-locked buffers from psim_host_alloc, so that the device can fill one with frame k
/// while the other, holding frame k-1, is handed to wgpu.  `update` therefore never waits for
/// the work it has just enqueued; the price is that the renderer sees every frame one frame
/// late (and, while paused, the frame before the last one computed).
pub struct HipParticleSimulation {
    handle: *mut PsimHandle,
    staging: [*mut Particle; 2],
    staging_len: usize,
    /// staging buffer with a download in flight: frame `frame - 1`
    in_flight: Option<usize>,
    frame: u64,
    particle_buffer: wgpu::Buffer,
}

fn alloc_staging(n: usize) -> *mut Particle {
    let mut p: *mut c_void = std::ptr::null_mut();
    // a set of 0 particles still gets a (64-byte) buffer: nothing below special-cases it
    check(unsafe { psim_host_alloc(&mut p, (n.max(1) as u64) * 64) });
    p as *mut Particle
}

fn vertex_buffer(device: &wgpu::Device, count: u32) -> wgpu::Buffer {
    device.create_buffer(&wgpu::BufferDescriptor {
        label: Some("HIP Particle Buffer"),
        size: (count.max(1) as u64) * 64,
        usage: wgpu::BufferUsages::COPY_DST | wgpu::BufferUsages::VERTEX,
        mapped_at_creation: false,
    })
}

impl HipParticleSimulation {
    /// The current particles into the vertex buffer, waiting for them: what the reference's
    /// backends do in `new`, `resize_buffer` and `reset` (cpu.rs:24-28, :143-157, :181-185;
    /// compute.rs:28-34, :187-189, :214-215), so that the renderer has the new set even while
    /// the simulation is paused (app.rs:283 skips `update` then).
    fn upload_now(&mut self, queue: &wgpu::Queue) {
        self.in_flight = None; // whatever was on its way belongs to the old set
        if self.staging_len == 0 {
            return;
        }
        let staging = self.staging[0] as *mut PsimParticle;
        check(unsafe { psim_download_aos(self.handle, staging, 0, self.staging_len as u64) });
        let frame = unsafe { std::slice::from_raw_parts(self.staging[0], self.staging_len) };
        queue.write_buffer(&self.particle_buffer, 0, bytemuck::cast_slice(frame));
    }

    fn resize_staging(&mut self, device: &wgpu::Device, new_count: u32) {
        if new_count as usize != self.staging_len {
            check(unsafe { psim_sync(self.handle) }); // no download may still write the old buffers
            for s in self.staging.iter_mut() {
                unsafe { psim_host_free(*s as *mut c_void) };
                *s = alloc_staging(new_count as usize);
            }
            self.staging_len = new_count as usize;
            self.particle_buffer = vertex_buffer(device, new_count);
        }
    }
}

impl ParticleSimulation for HipParticleSimulation {
    fn new(
        device: &wgpu::Device,
        initial_particle_count: u32,
        _surface_format: wgpu::TextureFormat,
        generation_mode: SphereGeneration,
    ) -> Self {
        let mut handle = std::ptr::null_mut();
        check(unsafe {
            psim_create(&mut handle, initial_particle_count, generation(generation_mode), 0)
        });
        // `new` has no queue (mod.rs:22-29), so the first frame reaches the vertex buffer with
        // the first `update`, `resize_buffer` or `reset`; a caller that may start paused adds
        // `sim.reset(device, queue, mode)` after `new` (the reference's backends fill their
        // buffer with create_buffer_init, which needs no queue: cpu.rs:24-28).
        Self {
            handle,
            staging: [
                alloc_staging(initial_particle_count as usize),
                alloc_staging(initial_particle_count as usize),
            ],
            staging_len: initial_particle_count as usize,
            in_flight: None,
            frame: 0,
            particle_buffer: vertex_buffer(device, initial_particle_count),
        }
    }

    fn update(
        &mut self,
        _device: &wgpu::Device,
        queue: &wgpu::Queue,
        _encoder: &mut wgpu::CommandEncoder,
        params: &SimParams,
    ) {
        // SimParams is #[repr(C)] with the layout of psim_sim_params (mod.rs:52-67)
        let params = params as *const SimParams as *const PsimSimParams;
        check(unsafe { psim_update(self.handle, params) }); // asynchronous
        if self.staging_len == 0 {
            return;
        }
        // frame k: transposed and copied out behind the update, into the buffer the renderer
        // is not being fed from
        let slot = (self.frame & 1) as usize;
        let n = self.staging_len as u64;
        check(unsafe {
            psim_download_aos_async(self.handle, self.staging[slot] as *mut PsimParticle, 0, n)
        });
        // frame k-1 has had a whole frame's time to arrive: hand it to the renderer while the
        // device works on frame k
        if let Some(prev) = self.in_flight {
            check(unsafe {
                psim_download_aos_wait(self.handle, self.staging[prev] as *const PsimParticle)
            });
            let frame = unsafe { std::slice::from_raw_parts(self.staging[prev], self.staging_len) };
            queue.write_buffer(&self.particle_buffer, 0, bytemuck::cast_slice(frame));
        }
        self.in_flight = Some(slot);
        self.frame += 1;
    }

    fn resize_buffer(
        &mut self,
        device: &wgpu::Device,
        queue: &wgpu::Queue,
        new_count: u32,
        generation_mode: SphereGeneration,
    ) {
        check(unsafe { psim_resize(self.handle, new_count, generation(generation_mode)) });
        self.resize_staging(device, new_count);
        self.upload_now(queue); // cpu.rs:153-157, compute.rs:187-189
    }

    fn get_particle_buffer(&self) -> &wgpu::Buffer {
        &self.particle_buffer
    }

    fn get_method(&self) -> SimulationMethod {
        SimulationMethod::Hip
    }

    fn get_particle_count(&self) -> u32 {
        unsafe { psim_particle_count(self.handle) }
    }

    fn reset(
        &mut self,
        _device: &wgpu::Device,
        queue: &wgpu::Queue,
        generation_mode: SphereGeneration,
    ) {
        check(unsafe { psim_reset(self.handle, generation(generation_mode)) });
        self.upload_now(queue); // cpu.rs:181-185, compute.rs:214-215
    }

    fn is_paused(&self) -> bool {
        unsafe { psim_is_paused(self.handle) != 0 }
    }

    fn set_paused(&mut self, paused: bool) {
        unsafe { psim_set_paused(self.handle, paused as c_int) }
    }
}

impl Drop for HipParticleSimulation {
    fn drop(&mut self) {
        unsafe {
            psim_sync(self.handle); // a download may still be writing the staging buffers
            for s in self.staging {
                psim_host_free(s as *mut c_void);
            }
            psim_destroy(self.handle)
        }
    }
}

/// One rank of a headless one-process-per-GPU run with a moving attractor per rank: the
/// shard, the RCCL exchange and the per-frame call, nothing else.  The ranks meet through a
/// file that rank 0 publishes (`psim_exchange_rendezvous_file`).
pub struct ShardedRank {
    handle: *mut PsimHandle,
    exchange: *mut PsimExchange,
}

impl ShardedRank {
    pub fn new(rank: i32, world: i32, device: i32, particles_per_rank: u32, id_file: &str) -> Self {
        let mut handle = std::ptr::null_mut();
        let n = particles_per_rank as u64;
        check(unsafe {
            psim_create_shard(&mut handle, particles_per_rank, rank as u64 * n, world as u64 * n,
                              PSIM_GEN_UNIFORM, device)
        });
        let path = std::ffi::CString::new(id_file).expect("path without NUL");
        let mut id = [0u8; PSIM_EXCHANGE_ID_BYTES as usize];
        check(unsafe { psim_exchange_rendezvous_file(path.as_ptr(), rank, id.as_mut_ptr(), 60_000) });
        let mut exchange = std::ptr::null_mut();
        check(unsafe { psim_exchange_create(&mut exchange, handle, id.as_ptr(), rank, world) });
        // a rank that dies must not leave the others hanging in their next collective
        check(unsafe { psim_exchange_set_timeout(exchange, 30_000) });
        Self { handle, exchange }
    }

    /// One frame: this rank's attractor joins the others', all are applied (asynchronous).
    /// Panics (psim error 6) when the exchange has given up on a rank that has gone.
    pub fn step(&mut self, params: &PsimSimParams, mine: &PsimAttractor) {
        check(unsafe { psim_exchange_step(self.exchange, params, mine) });
    }

    /// A headless rank has nothing else to do between frames: let the thread wait for each
    /// gather, so that the kernels run back to back (`PSIM_EXCHANGE_HOST_PACED`).  Faster on one
    /// rank; on several GPUs measure first: the default pacing keeps the gathers a step ahead,
    /// which is what lets them find room beside the kernels.
    pub fn pace_from_host(&mut self) {
        check(unsafe { psim_exchange_set_pacing(self.exchange, PSIM_EXCHANGE_HOST_PACED) });
    }

    /// The pacing for a rank of a multi-GPU run that steps on and on: host pacing one call
    /// late (`PSIM_EXCHANGE_PIPELINED`), so that kernels run back to back AND every all-gather
    /// is already waiting when the kernel before it ends.  The particles are one step behind
    /// until `sync`.
    pub fn pace_pipelined(&mut self) {
        check(unsafe { psim_exchange_set_pacing(self.exchange, PSIM_EXCHANGE_PIPELINED) });
    }

    pub fn sync(&mut self) {
        check(unsafe { psim_exchange_sync(self.exchange) });
    }
}

impl Drop for ShardedRank {
    fn drop(&mut self) {
        unsafe {
            psim_exchange_destroy(self.exchange);
            psim_destroy(self.handle)
        }
    }
}
