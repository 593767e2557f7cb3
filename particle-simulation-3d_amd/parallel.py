"""Sharding of the particle array across the GPUs of one node, one process per GPU.

Particles never read other particles (reference: src/simulation/cpu.rs:65-113 has no
pairwise term), so the array is cut into contiguous index ranges and every rank steps
its own range.  With the reference's single attractor the 48-byte SimParams is
replicated by the host and NO communication happens.  Only the multi-attractor
extension (one moving attractor owned by each rank; the reference has exactly one)
exchanges data: a 32-byte record per rank per step, all-gathered over RCCL (xGMI between
the GPUs) -- by the library's own exchange (NativeMultiAttractorRunner over psim_exchange_*:
no torch in the data path) or, as first built, with torch.distributed (MultiAttractorRunner:
backend "nccl" = RCCL on the GPUs, "gloo" in CPU tests).  The exchange is latency-bound
(256 B on 8 ranks), so it runs on its own stream into rotating receive buffers and overlaps
the previous step's kernel.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np

from . import ATTRACTOR_DTYPE, GEN_UNIFORM, HipParticleSimulation, store_attractor

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


class AttractorExchange:
    """All-gather of one attractor record per rank with torch.distributed.

    Backend-agnostic: tensors live on `device` ("cpu" with gloo, "cuda" with nccl).
    `gather(record)` returns a (world, 8) float32 tensor holding every rank's record
    bit for bit (the uint32 `active` word travels as its float32 bit pattern).
    """

    def __init__(self, device, group=None):
        import torch
        import torch.distributed as dist
        self._torch, self._dist, self._group = torch, dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = device

    def gather(self, record):
        torch, dist = self._torch, self._dist
        words = np.ascontiguousarray(record, dtype=ATTRACTOR_DTYPE).view(np.float32).reshape(-1)
        send = torch.from_numpy(words.copy()).to(self.device)
        if self.device == "cpu" or str(self.device) == "cpu":
            parts = [torch.empty_like(send) for _ in range(self.world)]
            dist.all_gather(parts, send, group=self._group)      # gloo
            return torch.stack(parts)
        recv = torch.empty(self.world * ATTRACTOR_FLOATS, dtype=torch.float32, device=self.device)
        dist.all_gather_into_tensor(recv, send, group=self._group)
        return recv.view(self.world, ATTRACTOR_FLOATS)


def gathered_to_records(gathered):
    """(world, 8) float32 tensor -> ATTRACTOR_DTYPE array (bit-preserving)."""
    arr = np.ascontiguousarray(gathered.detach().cpu().numpy(), dtype=np.float32)
    return arr.view(ATTRACTOR_DTYPE).reshape(-1)


class NativeMultiAttractorRunner:
    """One rank of the sharded multi-attractor simulation through the library's own exchange
    (`psim_exchange_*`: RCCL all-gather on a communication stream into a ring of receive
    lists, then the multi-attractor kernel): the same steps as MultiAttractorRunner without torch in the data
    path.  `unique_id` is the 128-byte communicator id every rank received from the one that
    made it (HipParticleSimulation-side helpers: exchange_unique_id, exchange_rendezvous_file).
    Attractor `rank` follows the closed-form path of attractor_at().  `pacing` "host" makes
    the calling thread wait for each gather so that the kernels run back to back (see
    psim_exchange_set_pacing in include/psim.h)."""

    def __init__(self, particles_per_rank, rank, world, local_rank, params, unique_id,
                 force=5.0, radius=10.0, first=None, total=None, pacing="device"):
        from . import HipAttractorExchange, EXCHANGE_DEVICE_PACED, EXCHANGE_HOST_PACED
        self.rank, self.world, self.params = rank, world, params
        self.force, self.radius = force, radius
        # this rank's shard [first, first + particles_per_rank) of `total` (default: equal shards)
        self.sim = HipParticleSimulation.new_shard(
            particles_per_rank, rank * particles_per_rank if first is None else first,
            world * particles_per_rank if total is None else total,
            GEN_UNIFORM, device_id=local_rank)
        self.exchange = HipAttractorExchange(self.sim, unique_id, rank, world)
        self.exchange.set_pacing({"device": EXCHANGE_DEVICE_PACED,
                                  "host": EXCHANGE_HOST_PACED}[pacing])
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


class MultiAttractorRunner:
    """One rank of the sharded multi-attractor simulation on a GPU.

    Per step t (buffers indexed t % 2):
      comm stream    : wait until the kernel of step t-2 has read recv[t%2];
                       write this rank's attractor(t) into send[t%2] (value carried in
                       kernel arguments: no host synchronisation, no pinned memory);
                       all-gather send[t%2] -> recv[t%2]
      compute stream : wait for that all-gather; multi-attractor update kernel reading
                       recv[t%2] (staged into LDS by every workgroup)
    so the all-gather of step t+1 overlaps the kernel of step t.

    `lookahead` = K > 1: the attractor paths are known ahead (closed form), so one
    collective carries K steps of every rank's attractor (K x 32 B per rank, staged through
    page-locked memory), a small device copy reorders the gathered [rank][step] records
    into one list per step, and the K update kernels that follow need no host work besides
    their launches.  The particles come out identical; only the number of collectives
    changes (one per K steps).
    """

    def __init__(self, particles_per_rank, rank, world, local_rank, params, group=None,
                 force=5.0, radius=10.0, lookahead=1, first=None, total=None):
        import torch
        import torch.distributed as dist
        self._torch, self._dist, self._group = torch, dist, group
        self.rank, self.world, self.params = rank, world, params
        self.force, self.radius = force, radius
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised (backend nccl)")
        self.sim = HipParticleSimulation.new_shard(
            particles_per_rank, rank * particles_per_rank if first is None else first,
            world * particles_per_rank if total is None else total,
            GEN_UNIFORM, device_id=local_rank)
        dev = torch.device("cuda", local_rank)
        self.comm = torch.cuda.Stream(device=dev)
        self.compute = torch.cuda.Stream(device=dev)
        self.sim.set_stream(self.compute.cuda_stream)
        self.send = [torch.zeros(ATTRACTOR_FLOATS, dtype=torch.float32, device=dev)
                     for _ in range(2)]
        self.recv = [torch.zeros(world * ATTRACTOR_FLOATS, dtype=torch.float32, device=dev)
                     for _ in range(2)]
        self.gathered = [torch.cuda.Event() for _ in range(2)]
        self.kernel_done = [torch.cuda.Event() for _ in range(2)]
        self.send_ptr = [t.data_ptr() for t in self.send]
        self.recv_ptr = [t.data_ptr() for t in self.recv]
        self.comm_ptr = self.comm.cuda_stream
        self.t = 0
        # The collective is ordered on torch's CURRENT stream, so the communication stream
        # is made current for the life of the runner (cheaper than entering a stream
        # context every step; the kernels of this library take their stream explicitly).
        self._previous_stream = torch.cuda.current_stream(dev)
        torch.cuda.set_stream(self.comm)
        self._path = None          # own attractor records for steps [_path_first, +len)
        self._path_first = 0
        self.lookahead = int(lookahead)
        if self.lookahead < 1:
            raise ValueError("lookahead must be >= 1")
        if self.lookahead > 1:
            k = self.lookahead
            self.host = [torch.empty(k * ATTRACTOR_FLOATS, dtype=torch.float32).pin_memory()
                         for _ in range(2)]
            self.send = [torch.zeros(k * ATTRACTOR_FLOATS, dtype=torch.float32, device=dev)
                         for _ in range(2)]
            self.recv = [torch.zeros(world * k * ATTRACTOR_FLOATS, dtype=torch.float32, device=dev)
                         for _ in range(2)]
            self.lists = [torch.zeros(k * world * ATTRACTOR_FLOATS, dtype=torch.float32,
                                      device=dev) for _ in range(2)]
            self.lists_ptr = [t.data_ptr() for t in self.lists]

    def _own_attractor(self, t):
        """This rank's attractor at step t, from a table refilled 256 steps at a time."""
        if self._path is None or not self._path_first <= t < self._path_first + len(self._path):
            self._path_first = t
            self._path = attractor_table(t, 256, self.rank, self.force, self.radius)
        return self._path[t - self._path_first:t - self._path_first + 1]

    def _gather_ahead(self, cycle):
        """Exchanges the attractors of steps cycle*K .. cycle*K + K-1 in one collective."""
        k, w, b = self.lookahead, self.world, cycle % 2
        if cycle >= 2:
            self.gathered[b].synchronize()     # host: the staging of cycle-2 has been consumed
            self.comm.wait_event(self.kernel_done[b])      # device: its lists have been read
        table = attractor_table(cycle * k, k, self.rank, self.force, self.radius)
        self.host[b].numpy()[:] = table.view(np.float32).reshape(-1)
        self.send[b].copy_(self.host[b], non_blocking=True)          # on the comm stream
        self._dist.all_gather_into_tensor(self.recv[b], self.send[b], group=self._group)
        self.lists[b].view(k, w, ATTRACTOR_FLOATS).copy_(
            self.recv[b].view(w, k, ATTRACTOR_FLOATS).transpose(0, 1))
        self.gathered[b].record(self.comm)
        self.compute.wait_event(self.gathered[b])

    def _step_ahead(self):
        k = self.lookahead
        cycle, s = divmod(self.t, k)
        if s == 0:
            self._gather_ahead(cycle)
        b = cycle % 2
        self.sim.update_multi_device(
            self.params, self.lists_ptr[b] + s * self.world * ATTRACTOR_DTYPE.itemsize, self.world)
        if s == k - 1:
            self.kernel_done[b].record(self.compute)
        self.t += 1

    def step(self):
        if self.lookahead > 1:
            return self._step_ahead()
        b = self.t % 2
        if self.t >= 2:                        # the kernel of step t-2 read recv[b]
            self.comm.wait_event(self.kernel_done[b])
        store_attractor(self.send_ptr[b], self._own_attractor(self.t), self.comm_ptr)
        self._dist.all_gather_into_tensor(self.recv[b], self.send[b], group=self._group)
        self.gathered[b].record(self.comm)
        self.compute.wait_event(self.gathered[b])
        self.sim.update_multi_device(self.params, self.recv_ptr[b], self.world)
        self.kernel_done[b].record(self.compute)
        self.t += 1

    def close(self):
        self.finish()
        self._torch.cuda.set_stream(self._previous_stream)
        self.sim.close()

    def finish(self):
        self.compute.synchronize()
        self.comm.synchronize()
