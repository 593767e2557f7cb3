"""Sharding of the particle array across the GPUs of one node, one process per GPU.

Particles never read other particles (reference: src/simulation/cpu.rs:65-113 has no
pairwise term), so the array is cut into contiguous index ranges and every rank steps
its own range.  With the reference's single attractor the 48-byte SimParams is
replicated by the host and NO communication happens.  Only the multi-attractor
extension (one moving attractor owned by each rank; the reference has exactly one)
exchanges data: a 32-byte record per rank per step, all-gathered over RCCL (xGMI between
the GPUs) by the library's own exchange (NativeMultiAttractorRunner over psim_exchange_*).
The exchange is latency-bound (256 B on 8 ranks), so it runs on its own stream into a ring
of receive lists and overlaps the previous step's kernel.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np

from . import ATTRACTOR_DTYPE, GEN_UNIFORM, HipParticleSimulation

ATTRACTOR_FLOATS = ATTRACTOR_DTYPE.itemsize // 4      # 8 four-byte words per record


def shard_range(total, rank, world):
    """Contiguous range of rank `rank`: (first, count).  Ranges tile [0, total) and
    differ in size by at most one particle."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    first = rank * total // world
    last = (rank + 1) * total // world
    return first, last - first


def attractor_at(step, index, force=5.0, radius=10.0):
    """Attractor `index` at step `step`: a closed-form Lissajous path inside the
    50-unit world cube (evaluated in float64, rounded once to fp32), so every rank and
    the CPU oracle can reproduce any attractor at any step without communication."""
    return attractor_table(step, 1, index, force, radius)


def attractor_table(first_step, count, index, force=5.0, radius=10.0):
    """Attractor `index` at steps first_step .. first_step + count - 1 (one record each)."""
    a = np.zeros(count, dtype=ATTRACTOR_DTYPE)
    pos = a["position"]
    wx, wy, wz = 1.0 + 0.25 * index, 1.5 + 0.125 * index, 0.75 + 0.0625 * index
    px, py, pz = 0.7 * index, 1.3 * index, 0.4 * index
    for k in range(count):
        t = 0.01 * (first_step + k)
        pos[k, 0] = 40.0 * math.sin(wx * t + px)
        pos[k, 1] = 40.0 * math.sin(wy * t + py)
        pos[k, 2] = 40.0 * math.cos(wz * t + pz)
    a["force"] = force
    a["radius"] = radius
    a["active"] = 1
    return a


def attractors_at(step, world, force=5.0, radius=10.0):
    """All `world` attractors at `step` (what the all-gather must deliver)."""
    return np.concatenate([attractor_at(step, i, force, radius) for i in range(world)])


class NativeMultiAttractorRunner:
    """One rank of the sharded multi-attractor simulation through the library's own exchange
    (`psim_exchange_*`: RCCL all-gather on a communication stream into a ring of receive
    lists, then the multi-attractor kernel).  `unique_id` is the 128-byte communicator id every rank received from the one that
    made it (HipParticleSimulation-side helpers: exchange_unique_id, exchange_rendezvous_file).
    Attractor `rank` follows the closed-form path of attractor_at().  `pacing` "host" makes
    the calling thread wait for each gather so that the kernels run back to back, "pipelined"
    does the same one call late, so that every gather is already waiting when the kernel before
    it ends -- the particles are then one step behind until finish() (see
    psim_exchange_set_pacing in include/psim.h); `timeout_ms` > 0 bounds every wait of the
    exchange (psim_exchange_set_timeout)."""

    def __init__(self, particles_per_rank, rank, world, local_rank, params, unique_id,
                 force=5.0, radius=10.0, first=None, total=None, pacing="device", timeout_ms=0):
        from . import (HipAttractorExchange, EXCHANGE_DEVICE_PACED, EXCHANGE_HOST_PACED,
                       EXCHANGE_PIPELINED)
        self.rank, self.world, self.params = rank, world, params
        self.force, self.radius = force, radius
        # this rank's shard [first, first + particles_per_rank) of `total` (default: equal shards)
        self.sim = HipParticleSimulation.new_shard(
            particles_per_rank, rank * particles_per_rank if first is None else first,
            world * particles_per_rank if total is None else total,
            GEN_UNIFORM, device_id=local_rank)
        self.exchange = HipAttractorExchange(self.sim, unique_id, rank, world)
        self.exchange.set_pacing({"device": EXCHANGE_DEVICE_PACED, "host": EXCHANGE_HOST_PACED,
                                  "pipelined": EXCHANGE_PIPELINED}[pacing])
        if timeout_ms:                 # give up (PsimError, code 6) instead of hanging on a lost rank
            self.exchange.set_timeout(timeout_ms)
        self.t = 0
        self._path = None          # own attractor records for steps [_path_first, +len)
        self._path_first = 0
        self._path_address = 0
        self._params_ref = ctypes.byref(params)

    def step(self):
        # The host side of a step must stay well under the kernel's 40-odd microseconds: the
        # records come from a table refilled 256 steps at a time and are handed over by address.
        k = self.t - self._path_first
        if self._path is None or not 0 <= k < len(self._path):
            self._path_first, k = self.t, 0
            self._path = attractor_table(self.t, 256, self.rank, self.force, self.radius)
            self._path_address = self._path.ctypes.data
        self.exchange.step_at(self._params_ref, self._path_address + k * ATTRACTOR_DTYPE.itemsize)
        self.t += 1

    def finish(self):
        self.exchange.sync()

    def close(self):
        self.exchange.close()
        self.sim.close()
