"""MI355X (gfx950) backend for the particle-update path of particle-simulation-3d.

This package is a thin ctypes binding over ``csrc/libpsim_hip.so`` (the C ABI of
``include/psim.h``) plus :class:`HipParticleSimulation`, a host-side mirror of the
reference's ``trait ParticleSimulation`` (reference: src/simulation/mod.rs:21-50)
with the wgpu arguments dropped.  All particle arithmetic runs in hand-written HIP
kernels; there is no CPU fallback, and a missing library is an error.

The directory name contains hyphens, so import it with
``importlib.import_module("particle-simulation-3d_amd")``.
"""
from __future__ import annotations

import ctypes
import os
import sys
from ctypes import POINTER, byref, c_char_p, c_float, c_int, c_uint32, c_uint64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PSIM_LIBRARY selects another build of the same library (kernel experiments).
LIBRARY_PATH = os.environ.get("PSIM_LIBRARY") or os.path.join(_HERE, "csrc", "libpsim_hip.so")

# ---- records shared with the reference ---------------------------------------

# `#[repr(C)] struct Particle`, 64 bytes (reference: src/simulation/mod.rs:86-98).
PARTICLE_DTYPE = np.dtype(
    [
        ("position", "<f4", (3,)),
        ("padding1", "<f4"),
        ("velocity", "<f4", (3,)),
        ("padding2", "<f4"),
        ("color", "<f4", (4,)),
        ("initial_color", "<f4", (4,)),
    ]
)
assert PARTICLE_DTYPE.itemsize == 64

# One attractor of the multi-attractor extension, 32 bytes (include/psim.h).
ATTRACTOR_DTYPE = np.dtype(
    [
        ("position", "<f4", (3,)),
        ("force", "<f4"),
        ("radius", "<f4"),
        ("active", "<u4"),
        ("_pad", "<u4", (2,)),
    ]
)
assert ATTRACTOR_DTYPE.itemsize == 32


class SimParams(ctypes.Structure):
    """`#[repr(C)] struct SimParams`, 48 bytes (reference: src/simulation/mod.rs:52-67).

    The defaults are `impl Default for SimParams` (mod.rs:69-84).
    """

    _fields_ = [
        ("delta_time", c_float),
        ("gravity", c_float),
        ("color_mode", c_uint32),
        ("mouse_force", c_float),
        ("mouse_radius", c_float),
        ("is_mouse_dragging", c_uint32),
        ("damping", c_float),
        ("max_dist_for_color", c_float),
        ("mouse_position", c_float * 3),
        ("_padding2", c_uint32),
    ]

    def __init__(self, delta_time=0.016, gravity=0.0, color_mode=0, mouse_force=5.0,
                 mouse_radius=10.0, is_mouse_dragging=0, damping=0.99,
                 max_dist_for_color=50.0, mouse_position=(0.0, 0.0, 0.0)):
        super().__init__()
        self.delta_time = delta_time
        self.gravity = gravity
        self.color_mode = color_mode
        self.mouse_force = mouse_force
        self.mouse_radius = mouse_radius
        self.is_mouse_dragging = is_mouse_dragging
        self.damping = damping
        self.max_dist_for_color = max_dist_for_color
        self.mouse_position = (c_float * 3)(*mouse_position)
        self._padding2 = 0


assert ctypes.sizeof(SimParams) == 48


class SoaView(ctypes.Structure):
    """`psim_soa_view`: raw device pointers of a handle's storage."""

    _fields_ = [(n, c_void_p) for n in
                ("px", "py", "pz", "vx", "vy", "vz", "color", "initial_color", "padding")] + [
        ("count", c_uint32), ("capacity", c_uint32)]


# `enum SphereGeneration` (mod.rs:15-19) + the synthetic benchmark distribution.
GEN_HOLLOW, GEN_FILLED, GEN_UNIFORM = 0, 1, 2
# `enum SimulationMethod` (mod.rs:9-13) + this backend.
METHOD_CPU, METHOD_COMPUTE_SHADER, METHOD_HIP = 0, 1, 2
COLOR_KEEP, COLOR_RESET = 0, 1      # cpu.rs:106 / compute.wgsl:75-77
RESIZE_REGENERATE, RESIZE_APPEND = 0, 1   # compute.rs:146-193 / cpu.rs:123-158
MAX_ATTRACTORS = 64
MAX_HOST_ATTRACTORS = 64     # lists of up to 8 travel in the kernel arguments, longer ones are staged

# Every symbol include/psim.h declares: name -> (restype, argtypes).
class LaunchShapeInfo(ctypes.Structure):
    """`psim_launch_shape_info`: the launch shape picked for a particle count."""

    _fields_ = [("access", c_int), ("workgroup_size", c_uint32), ("workgroups", c_uint32),
                ("reserve_lds", c_uint32), ("reserve_lds_colour", c_uint32),
                ("resident_per_cu", c_uint32)]


ACCESS_PLAIN, ACCESS_RESIDENT, ACCESS_STREAM = 0, 1, 2


class CheckpointHeader(ctypes.Structure):
    """`psim_checkpoint_header`: the 64 bytes at the start of a checkpoint file."""

    _fields_ = [("magic", ctypes.c_char * 8), ("version", c_uint32), ("record_bytes", c_uint32),
                ("particle_count", c_uint64), ("step", c_uint64), ("generation_mode", c_uint32),
                ("color_semantics", c_uint32), ("reserved", ctypes.c_uint8 * 24)]


assert ctypes.sizeof(CheckpointHeader) == 64

_P = c_void_p
ABI = {
    "psim_launch_shape": (c_int, [c_uint32, c_uint32, POINTER(LaunchShapeInfo)]),
    "psim_launch_occupancy": (c_int, [_P, c_uint32, c_uint32, POINTER(c_uint32), POINTER(c_uint32)]),
    "psim_create": (c_int, [POINTER(_P), c_uint32, c_int, c_int]),
    "psim_create_shard": (c_int, [POINTER(_P), c_uint32, c_uint64, c_uint64, c_int, c_int]),
    "psim_destroy": (None, [_P]),
    "psim_update": (c_int, [_P, POINTER(SimParams)]),
    "psim_update_batch": (c_int, [_P, POINTER(SimParams), c_uint32]),
    "psim_update_multi": (c_int, [_P, POINTER(SimParams), _P, c_uint32]),
    "psim_update_multi_host": (c_int, [_P, POINTER(SimParams), _P, c_uint32]),
    "psim_update_multi_host_noalloc": (c_int, [_P, POINTER(SimParams), _P, c_uint32]),
    "psim_update_soa": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_uint32, POINTER(SimParams), _P]),
    "psim_store_attractor": (c_int, [_P, _P, _P]),
    "psim_sync": (c_int, [_P]),
    "psim_resize": (c_int, [_P, c_uint32, c_int]),
    "psim_reset": (c_int, [_P, c_int]),
    "psim_particle_count": (c_uint32, [_P]),
    "psim_method": (c_int, [_P]),
    "psim_is_paused": (c_int, [_P]),
    "psim_set_paused": (None, [_P, c_int]),
    "psim_set_color_semantics": (c_int, [_P, c_int]),
    "psim_color_semantics": (c_int, [_P]),
    "psim_set_resize_semantics": (c_int, [_P, c_int]),
    "psim_resize_semantics": (c_int, [_P]),
    "psim_download_aos": (c_int, [_P, _P, c_uint64, c_uint64]),
    "psim_upload_aos": (c_int, [_P, _P, c_uint64, c_uint64]),
    "psim_download_aos_async": (c_int, [_P, _P, c_uint64, c_uint64]),
    "psim_download_aos_wait": (c_int, [_P, _P]),
    "psim_host_alloc": (c_int, [POINTER(_P), c_uint64]),
    "psim_host_free": (None, [_P]),
    "psim_download_soa": (c_int, [_P, _P, c_uint64, c_uint64]),
    "psim_upload_soa": (c_int, [_P, _P, c_uint64, c_uint64]),
    "psim_export_aos_device": (c_int, [_P, _P, c_uint64, c_uint64]),
    "psim_import_aos_device": (c_int, [_P, _P, c_uint64, c_uint64]),
    "psim_device_soa": (c_int, [_P, POINTER(SoaView)]),
    "psim_shared_aos": (c_int, [_P, POINTER(_P), POINTER(c_int), POINTER(c_uint64)]),
    "psim_checksum": (c_int, [_P, POINTER(c_uint64)]),
    "psim_checkpoint_write": (c_int, [_P, c_char_p, c_uint64]),
    "psim_checkpoint_begin": (c_int, [c_char_p]),
    "psim_checkpoint_commit": (c_int, [_P, c_char_p, c_uint64]),
    "psim_checkpoint_read": (c_int, [_P, c_char_p, POINTER(c_uint64)]),
    "psim_checkpoint_info": (c_int, [c_char_p, POINTER(CheckpointHeader)]),
    "psim_group_create": (c_int, [POINTER(_P), c_uint64, POINTER(c_int), c_uint32, c_int]),
    "psim_group_destroy": (None, [_P]),
    "psim_group_update": (c_int, [_P, POINTER(SimParams)]),
    "psim_group_update_batch": (c_int, [_P, POINTER(SimParams), c_uint32]),
    "psim_group_update_multi_host": (c_int, [_P, POINTER(SimParams), _P, c_uint32]),
    "psim_group_sync": (c_int, [_P]),
    "psim_group_reset": (c_int, [_P, c_int]),
    "psim_group_particle_count": (c_uint64, [_P]),
    "psim_group_shards": (c_uint32, [_P]),
    "psim_group_shard": (_P, [_P, c_uint32]),
    "psim_group_download_aos": (c_int, [_P, _P, c_uint64, c_uint64]),
    "psim_group_upload_aos": (c_int, [_P, _P, c_uint64, c_uint64]),
    "psim_group_checksum": (c_int, [_P, POINTER(c_uint64)]),
    "psim_group_checkpoint_write": (c_int, [_P, c_char_p, c_uint64]),
    "psim_group_checkpoint_read": (c_int, [_P, c_char_p, POINTER(c_uint64)]),
    "psim_exchange_unique_id": (c_int, [_P]),
    "psim_exchange_rendezvous_file": (c_int, [c_char_p, c_int, _P, c_uint32]),
    "psim_exchange_create": (c_int, [POINTER(_P), _P, _P, c_int, c_int]),
    "psim_exchange_destroy": (None, [_P]),
    "psim_exchange_step": (c_int, [_P, POINTER(SimParams), _P]),
    "psim_exchange_set_pacing": (c_int, [_P, c_int]),
    "psim_exchange_set_timeout": (c_int, [_P, c_uint32]),
    "psim_exchange_sync": (c_int, [_P]),
    "psim_exchange_last_list": (c_int, [_P, _P]),
    "psim_exchange_world": (c_int, [_P]),
    "psim_exchange_rank": (c_int, [_P]),
    "psim_set_stream": (c_int, [_P, _P]),
    "psim_get_stream": (_P, [_P]),
    "psim_device": (c_int, [_P]),
    "psim_device_count": (c_int, []),
    "psim_timer_start": (c_int, [_P]),
    "psim_timer_stop": (c_int, [_P, POINTER(c_float)]),
    "psim_debug_set_tuning": (c_int, [c_char_p, c_int]),
    "psim_last_error": (c_char_p, []),
    "psim_version": (c_char_p, []),
}

_lib = None


class PsimError(RuntimeError):
    """A C-ABI call returned a non-zero status."""

    def __init__(self, code, message):
        super().__init__(f"psim error {code}: {message}")
        self.code = code


def load_library():
    """Loads ``csrc/libpsim_hip.so`` and declares every entry point.

    Raises ``RuntimeError`` if the library has not been built; nothing else can
    compute the update.
    """
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBRARY_PATH):
        raise RuntimeError(
            f"{LIBRARY_PATH} is missing: build it with `make -C {os.path.dirname(LIBRARY_PATH)}` "
            "(or __graft_entry__.build()); this backend has no CPU fallback")
    lib = ctypes.CDLL(LIBRARY_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (restype, argtypes) in ABI.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise PsimError(rc, load_library().psim_last_error().decode())


def _as_attractors(attractors):
    arr = np.ascontiguousarray(attractors, dtype=ATTRACTOR_DTYPE)
    return arr


class HipParticleSimulation:
    """Host-side mirror of the reference's ``trait ParticleSimulation``
    (src/simulation/mod.rs:21-50) over the C ABI.

    Method names and argument meaning follow the trait; the wgpu device / queue /
    encoder / surface-format arguments are dropped.  ``update`` is asynchronous,
    as in the reference's compute backend (src/simulation/compute.rs:124-144);
    call :meth:`sync` (or any download) to wait.
    """

    def __init__(self, handle):
        self._h = handle
        self._lib = load_library()

    # -- `fn new(device, initial_particle_count, surface_format, generation_mode)`
    @classmethod
    def new(cls, initial_particle_count, generation_mode=GEN_HOLLOW, device_id=0):
        lib = load_library()
        h = c_void_p()
        _check(lib.psim_create(byref(h), initial_particle_count, generation_mode, device_id))
        return cls(h)

    @classmethod
    def new_shard(cls, count, global_first, global_count, generation_mode=GEN_UNIFORM,
                  device_id=0):
        """One shard [global_first, global_first + count) of a `global_count` set."""
        lib = load_library()
        h = c_void_p()
        _check(lib.psim_create_shard(byref(h), count, global_first, global_count,
                                     generation_mode, device_id))
        return cls(h)

    def close(self):
        if self._h:
            self._lib.psim_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- `fn update(&mut self, device, queue, encoder, params: &SimParams)`
    def update(self, params):
        _check(self._lib.psim_update(self._h, byref(params)))

    def update_batch(self, params_list):
        """Applies consecutive updates, one per SimParams in `params_list`, with the state
        crossing HBM once per 64 steps; bit-identical to calling update() for each."""
        arr = (SimParams * len(params_list))(*params_list)
        _check(self._lib.psim_update_batch(self._h, arr, len(params_list)))

    def update_multi_device(self, params, attractors_device_ptr, n_attractors):
        """Multi-attractor step with the list in DEVICE memory (integer address)."""
        _check(self._lib.psim_update_multi(self._h, byref(params),
                                           c_void_p(attractors_device_ptr), n_attractors))

    def update_multi(self, params, attractors):
        """Multi-attractor step with a host list (ATTRACTOR_DTYPE array, <= 64)."""
        arr = _as_attractors(attractors)
        _check(self._lib.psim_update_multi_host(self._h, byref(params),
                                                arr.ctypes.data_as(c_void_p), len(arr)))

    def update_multi_noalloc(self, params, attractors):
        """update_multi through the library's static device list: allocates nothing."""
        arr = _as_attractors(attractors)
        _check(self._lib.psim_update_multi_host_noalloc(self._h, byref(params),
                                                        arr.ctypes.data_as(c_void_p), len(arr)))

    def store_attractor(self, dst_device_ptr, attractor, stream=None):
        """Writes one attractor record to device memory on `stream` (default: the
        handle's stream)."""
        store_attractor(dst_device_ptr, attractor,
                        self.get_stream() if stream is None else stream)

    def sync(self):
        _check(self._lib.psim_sync(self._h))

    # -- `fn resize_buffer(&mut self, device, queue, new_count, generation_mode)`
    def resize_buffer(self, new_count, generation_mode):
        _check(self._lib.psim_resize(self._h, new_count, generation_mode))

    # -- `fn get_method(&self) -> SimulationMethod`
    def get_method(self):
        return self._lib.psim_method(self._h)

    # -- `fn get_particle_count(&self) -> u32`
    def get_particle_count(self):
        return self._lib.psim_particle_count(self._h)

    # -- `fn reset(&mut self, device, queue, generation_mode)`
    def reset(self, generation_mode):
        _check(self._lib.psim_reset(self._h, generation_mode))

    # -- `fn is_paused(&self) -> bool` / `fn set_paused(&mut self, paused: bool)`
    def is_paused(self):
        return bool(self._lib.psim_is_paused(self._h))

    def set_paused(self, paused):
        self._lib.psim_set_paused(self._h, 1 if paused else 0)

    # -- colour of a step in a mode other than 1/2: keep (cpu.rs:106) or reset to the
    #    initial colour (compute.wgsl:75-77, 90-92)
    def set_color_semantics(self, semantics):
        _check(self._lib.psim_set_color_semantics(self._h, int(semantics)))

    def get_color_semantics(self):
        return self._lib.psim_color_semantics(self._h)

    # -- `fn get_particle_buffer(&self) -> &wgpu::Buffer`: the device pointers
    def get_particle_buffer(self):
        view = SoaView()
        _check(self._lib.psim_device_soa(self._h, byref(view)))
        return view

    # -- particle data in and out
    def download_aos(self, first=0, n=None, out=None):
        """Particles [first, first + n) as 64-byte records; `out` (for instance an array
        from pinned_empty) is filled and returned when given."""
        n = self.get_particle_count() - first if n is None else n
        if out is None:
            out = np.empty(n, dtype=PARTICLE_DTYPE)
        elif out.dtype != PARTICLE_DTYPE or out.ndim != 1 or len(out) < n \
                or not out.flags.c_contiguous:
            raise ValueError("out must be a contiguous PARTICLE_DTYPE array of at least n records")
        _check(self._lib.psim_download_aos(self._h, out.ctypes.data_as(c_void_p), first, n))
        return out[:n]

    def download_aos_async(self, out, first=0, n=None):
        """Enqueues the download into `out` (an array from pinned_empty) and returns; the
        records are valid after sync()."""
        n = self.get_particle_count() - first if n is None else n
        if out.dtype != PARTICLE_DTYPE or out.ndim != 1 or len(out) < n \
                or not out.flags.c_contiguous:
            raise ValueError("out must be a contiguous PARTICLE_DTYPE array of at least n records")
        _check(self._lib.psim_download_aos_async(self._h, out.ctypes.data_as(c_void_p), first, n))
        return out[:n]

    def download_aos_wait(self, out):
        """Waits for the most recent download_aos_async into `out`, and for nothing enqueued
        after it (a staged frame loop runs one frame ahead this way)."""
        _check(self._lib.psim_download_aos_wait(self._h, out.ctypes.data_as(c_void_p)))

    def upload_aos(self, particles, first=0):
        arr = np.ascontiguousarray(particles, dtype=PARTICLE_DTYPE)
        _check(self._lib.psim_upload_aos(self._h, arr.ctypes.data_as(c_void_p), first, len(arr)))

    def download_soa(self, first=0, n=None):
        """Returns a (6, n) float32 array: px, py, pz, vx, vy, vz."""
        n = self.get_particle_count() - first if n is None else n
        out = np.empty((6, n), dtype=np.float32)
        _check(self._lib.psim_download_soa(self._h, out.ctypes.data_as(c_void_p), first, n))
        return out

    def upload_soa(self, soa, first=0):
        arr = np.ascontiguousarray(soa, dtype=np.float32)
        if arr.ndim != 2 or arr.shape[0] != 6:
            raise ValueError("expected a (6, n) float32 array")
        _check(self._lib.psim_upload_soa(self._h, arr.ctypes.data_as(c_void_p), first,
                                         arr.shape[1]))

    def export_aos_device(self, dst_device_ptr, first=0, n=None):
        n = self.get_particle_count() - first if n is None else n
        _check(self._lib.psim_export_aos_device(self._h, c_void_p(dst_device_ptr), first, n))

    def import_aos_device(self, src_device_ptr, first=0, n=None):
        n = self.get_particle_count() - first if n is None else n
        _check(self._lib.psim_import_aos_device(self._h, c_void_p(src_device_ptr), first, n))

    def shared_aos(self):
        """(device address, new dma-buf file descriptor, bytes) of the handle's device buffer
        of 64-byte records for a renderer in another API or process; fill it with
        export_aos_device(address).  The caller closes the descriptor."""
        ptr, fd, size = c_void_p(), c_int(-1), c_uint64(0)
        _check(self._lib.psim_shared_aos(self._h, byref(ptr), byref(fd), byref(size)))
        return int(ptr.value), int(fd.value), int(size.value)

    def checksum(self):
        out = (c_uint64 * 6)()
        _check(self._lib.psim_checksum(self._h, out))
        return [int(v) for v in out]

    def set_resize_semantics(self, semantics):
        """RESIZE_REGENERATE (compute backend, default) or RESIZE_APPEND (CPU backend)."""
        _check(self._lib.psim_set_resize_semantics(self._h, semantics))

    def get_resize_semantics(self):
        return self._lib.psim_resize_semantics(self._h)

    # -- checkpoints
    def save_checkpoint(self, path, step=0):
        """Writes this handle's particles into the checkpoint file `path` (a shard writes
        its own records of the whole-set file)."""
        _check(self._lib.psim_checkpoint_write(self._h, os.fsencode(path), int(step)))

    def commit_checkpoint(self, path, step=0):
        """Shards only: writes the header once every shard has saved its records
        (checkpoint_begin -> every shard's save_checkpoint -> commit_checkpoint)."""
        _check(self._lib.psim_checkpoint_commit(self._h, os.fsencode(path), int(step)))

    def load_checkpoint(self, path):
        """Restores the particles from `path`; returns the stored step counter."""
        step = c_uint64(0)
        _check(self._lib.psim_checkpoint_read(self._h, os.fsencode(path), byref(step)))
        return int(step.value)

    # -- streams
    def set_stream(self, hip_stream_ptr):
        _check(self._lib.psim_set_stream(self._h, c_void_p(hip_stream_ptr)))

    def get_stream(self):
        return self._lib.psim_get_stream(self._h)

    def device(self):
        return self._lib.psim_device(self._h)

    # -- device timer around enqueued work
    def timer_start(self):
        _check(self._lib.psim_timer_start(self._h))

    def timer_stop(self):
        """Waits for the enqueued work and returns the device milliseconds."""
        ms = c_float()
        _check(self._lib.psim_timer_stop(self._h, byref(ms)))
        return float(ms.value)

    def launch_occupancy(self, n_attractors=1, color_mode=0):
        """(workgroups of this handle's update kernel one CU holds at once, asked of the
        runtime; LDS bytes one workgroup takes)."""
        out, lds = c_uint32(), c_uint32()
        _check(self._lib.psim_launch_occupancy(self._h, n_attractors, color_mode, byref(out),
                                               byref(lds)))
        return int(out.value), int(lds.value)

    @property
    def handle(self):
        return self._h


class HipParticleGroup:
    """One particle set over several shards driven from one process (`psim_group_*`):
    shard k lives on `device_ids[k]` and owns a contiguous index range."""

    def __init__(self, total_count, device_ids, generation_mode=GEN_UNIFORM):
        self._lib = load_library()
        ids = (c_int * len(device_ids))(*device_ids)
        g = c_void_p()
        _check(self._lib.psim_group_create(byref(g), int(total_count), ids, len(device_ids),
                                           generation_mode))
        self._g = g

    def close(self):
        if self._g:
            self._lib.psim_group_destroy(self._g)
            self._g = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update(self, params):
        _check(self._lib.psim_group_update(self._g, byref(params)))

    def update_batch(self, params_list):
        arr = (SimParams * len(params_list))(*params_list)
        _check(self._lib.psim_group_update_batch(self._g, arr, len(params_list)))

    def update_multi(self, params, attractors):
        arr = _as_attractors(attractors)
        _check(self._lib.psim_group_update_multi_host(self._g, byref(params),
                                                      arr.ctypes.data_as(c_void_p), len(arr)))

    def sync(self):
        _check(self._lib.psim_group_sync(self._g))

    def reset(self, generation_mode):
        _check(self._lib.psim_group_reset(self._g, generation_mode))

    def get_particle_count(self):
        return int(self._lib.psim_group_particle_count(self._g))

    def shard_count(self):
        return int(self._lib.psim_group_shards(self._g))

    def shard(self, index):
        """Shard `index` as a HipParticleSimulation that does NOT own its handle."""
        h = self._lib.psim_group_shard(self._g, index)
        if not h:
            raise IndexError(index)
        sim = HipParticleSimulation(c_void_p(h))
        sim.close = lambda: None                 # the group owns the handle
        return sim

    def download_aos(self, first=0, n=None):
        n = self.get_particle_count() - first if n is None else n
        out = np.empty(n, dtype=PARTICLE_DTYPE)
        _check(self._lib.psim_group_download_aos(self._g, out.ctypes.data_as(c_void_p), first, n))
        return out

    def upload_aos(self, particles, first=0):
        arr = np.ascontiguousarray(particles, dtype=PARTICLE_DTYPE)
        _check(self._lib.psim_group_upload_aos(self._g, arr.ctypes.data_as(c_void_p), first,
                                               len(arr)))

    def checksum(self):
        out = (c_uint64 * 6)()
        _check(self._lib.psim_group_checksum(self._g, out))
        return [int(v) for v in out]

    def save_checkpoint(self, path, step=0):
        _check(self._lib.psim_group_checkpoint_write(self._g, os.fsencode(path), int(step)))

    def load_checkpoint(self, path):
        step = c_uint64(0)
        _check(self._lib.psim_group_checkpoint_read(self._g, os.fsencode(path), byref(step)))
        return int(step.value)


EXCHANGE_ID_BYTES = 128
EXCHANGE_DEVICE_PACED, EXCHANGE_HOST_PACED, EXCHANGE_PIPELINED = 0, 1, 2


def exchange_unique_id():
    """A fresh communicator id (128 bytes) for HipAttractorExchange; ONE rank makes it and
    the caller hands it to all ranks."""
    buf = (ctypes.c_uint8 * EXCHANGE_ID_BYTES)()
    _check(load_library().psim_exchange_unique_id(buf))
    return bytes(buf)


def exchange_rendezvous_file(path, rank, timeout_ms=60_000):
    """The id through a file: rank 0 makes and publishes it at `path` (fresh per run), the
    other ranks wait for it.  Returns the 128 bytes."""
    buf = (ctypes.c_uint8 * EXCHANGE_ID_BYTES)()
    _check(load_library().psim_exchange_rendezvous_file(os.fsencode(path), int(rank), buf,
                                                        int(timeout_ms)))
    return bytes(buf)


class HipAttractorExchange:
    """`psim_exchange_*`: the multi-attractor step of a set sharded over one process per GPU.
    Every rank owns one attractor; step() all-gathers the ranks' records with RCCL on a stream
    of its own (into a ring of receive lists, overlapping the previous steps' kernels) and applies them all
    to the particles of `sim`; the data path is the library's own."""

    def __init__(self, sim, unique_id, rank, world):
        self._lib = load_library()
        if len(unique_id) != EXCHANGE_ID_BYTES:
            raise ValueError("the id is 128 bytes")
        self._sim = sim                              # must outlive the exchange
        ident = (ctypes.c_uint8 * EXCHANGE_ID_BYTES).from_buffer_copy(unique_id)
        x = c_void_p()
        _check(self._lib.psim_exchange_create(byref(x), sim.handle, ident, int(rank), int(world)))
        self._x = x
        self.rank, self.world = int(rank), int(world)

    def step(self, params, mine):
        arr = _as_attractors(mine).reshape(-1)[:1]
        _check(self._lib.psim_exchange_step(self._x, byref(params), arr.ctypes.data_as(c_void_p)))

    def step_at(self, params_ref, mine_address):
        """step() for tight loops: `params_ref` = ctypes.byref(params) made once,
        `mine_address` = the address of a 32-byte attractor record in host memory."""
        rc = self._lib.psim_exchange_step(self._x, params_ref, mine_address)
        if rc:
            _check(rc)

    def set_pacing(self, pacing):
        """EXCHANGE_DEVICE_PACED (default), EXCHANGE_HOST_PACED or EXCHANGE_PIPELINED (the
        handle is then one step behind until sync()): see include/psim.h."""
        _check(self._lib.psim_exchange_set_pacing(self._x, int(pacing)))

    def set_timeout(self, timeout_ms):
        """0 (default): wait for ever.  Otherwise every host-side wait gives the exchange up
        after `timeout_ms` (PsimError with code 6 = PSIM_ERR_COMM): see include/psim.h."""
        _check(self._lib.psim_exchange_set_timeout(self._x, int(timeout_ms)))

    def sync(self):
        _check(self._lib.psim_exchange_sync(self._x))

    def last_list(self):
        """The records the most recent step applied, in rank order."""
        out = np.zeros(self.world, dtype=ATTRACTOR_DTYPE)
        _check(self._lib.psim_exchange_last_list(self._x, out.ctypes.data_as(c_void_p)))
        return out

    def close(self):
        if self._x:
            self._lib.psim_exchange_destroy(self._x)
            self._x = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def update_soa(px, py, pz, vx, vy, vz, color, count, params, stream=0):
    """Stateless update over caller-owned DEVICE arrays (integer addresses)."""
    lib = load_library()
    _check(lib.psim_update_soa(c_void_p(px), c_void_p(py), c_void_p(pz), c_void_p(vx),
                               c_void_p(vy), c_void_p(vz), c_void_p(color), count,
                               byref(params), c_void_p(stream)))


def store_attractor(dst_device_ptr, attractor, stream=0):
    """Writes one ATTRACTOR_DTYPE record to DEVICE memory, asynchronously on `stream`."""
    arr = _as_attractors(attractor).reshape(-1)[:1]
    _check(load_library().psim_store_attractor(c_void_p(dst_device_ptr),
                                               arr.ctypes.data_as(c_void_p), c_void_p(stream)))


class _PinnedBlock:
    """Owns one psim_host_alloc allocation; freed when the last array over it goes."""

    def __init__(self, nbytes):
        self._lib = load_library()
        ptr = c_void_p()
        _check(self._lib.psim_host_alloc(byref(ptr), nbytes))
        self.ptr = ptr.value

    def __del__(self):
        if getattr(self, "ptr", None):
            self._lib.psim_host_free(c_void_p(self.ptr))
            self.ptr = None


def pinned_empty(n, dtype=PARTICLE_DTYPE):
    """An uninitialised 1-D numpy array of `n` records in page-locked host memory
    (psim_host_alloc): download_aos(out=...) / upload_aos() run at the host link's rate."""
    dtype = np.dtype(dtype)
    nbytes = int(n) * dtype.itemsize
    block = _PinnedBlock(nbytes)
    raw = (ctypes.c_char * max(nbytes, 1)).from_address(block.ptr)
    raw._owner = block                       # the numpy array keeps `raw`, `raw` keeps the block
    return np.frombuffer(raw, dtype=np.uint8, count=nbytes).view(dtype)


def launch_shape(count, n_attractors=1):
    """The launch shape the library picks for `count` particles (no device needed)."""
    info = LaunchShapeInfo()
    _check(load_library().psim_launch_shape(count, n_attractors, byref(info)))
    return {name: getattr(info, name) for name, _ in LaunchShapeInfo._fields_}


def checkpoint_begin(path):
    """Creates `path` when missing and invalidates its header: the first stage of a
    checkpoint written by several shards (no device needed)."""
    _check(load_library().psim_checkpoint_begin(os.fsencode(path)))


def checkpoint_info(path):
    """Header fields of a checkpoint file (no device needed)."""
    hdr = CheckpointHeader()
    _check(load_library().psim_checkpoint_info(os.fsencode(path), byref(hdr)))
    return {"version": hdr.version, "record_bytes": hdr.record_bytes,
            "particle_count": hdr.particle_count, "step": hdr.step,
            "generation_mode": hdr.generation_mode, "color_semantics": hdr.color_semantics}


CHECKPOINT_HEADER_BYTES = 64


def read_checkpoint_records(path, first=0, n=None):
    """Particles [first, first + n) of a checkpoint file as a read-only PARTICLE_DTYPE
    memory map (plain file access; for tools that inspect a checkpoint without a GPU)."""
    info = checkpoint_info(path)
    count = info["particle_count"]
    if n is None:
        n = count - first
    if first < 0 or n < 0 or first + n > count:
        raise ValueError(f"range [{first}, +{n}) exceeds the {count} particles of {path}")
    if n == 0:
        return np.empty(0, dtype=PARTICLE_DTYPE)
    return np.memmap(path, dtype=PARTICLE_DTYPE, mode="r",
                     offset=CHECKPOINT_HEADER_BYTES + first * PARTICLE_DTYPE.itemsize, shape=(n,))


def debug_set_tuning(name, value):
    """Overrides a launch-shape choice of this process (psim_debug_set_tuning: experiments and
    shape-forcing tests); value -1 restores the automatic choice."""
    _check(load_library().psim_debug_set_tuning(name.encode(), int(value)))


def device_count():
    """HIP devices this process sees."""
    return int(load_library().psim_device_count())


def version():
    return load_library().psim_version().decode()


sys.modules.setdefault("psim_amd", sys.modules[__name__])
