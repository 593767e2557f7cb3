"""The attractor exchange as first built, over torch.distributed: TEST SCAFFOLDING.

The product's exchange is the library's own (psim_exchange_*, driven from Python by
parallel.NativeMultiAttractorRunner).  These classes stay for two jobs the library's exchange
cannot do on a CPU-only or one-GPU machine: the world_size-2/3 gloo job of
tests/test_sharding_gloo.py, which checks shard ranges, attractor paths and the gathered lists
with the CPU oracle standing in for the kernel, and the real-RCCL cross-check of
tests/test_gpu_multi.py.  Nothing in the package or in bench.py imports this file.
"""
from __future__ import annotations

import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

psim = importlib.import_module("particle-simulation-3d_amd")
par = importlib.import_module("particle-simulation-3d_amd.parallel")

ATTRACTOR_DTYPE, GEN_UNIFORM = psim.ATTRACTOR_DTYPE, psim.GEN_UNIFORM
HipParticleSimulation, store_attractor = psim.HipParticleSimulation, psim.store_attractor
ATTRACTOR_FLOATS = par.ATTRACTOR_FLOATS
attractor_table = par.attractor_table


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
