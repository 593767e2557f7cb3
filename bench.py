#!/usr/bin/env python3
"""Benchmark of the particle-update hot path on MI355X.

A step is one update (one kernel launch) of this rank's particle shard.  At N = 1 the
workload is the headline configuration: 100 M particles, single attractor, fp32,
synthetic uniform-random state generated on the device.  With N > 1 every rank holds its
own 100 M-particle shard of an N x 100 M array (weak scaling); the single-attractor path
needs no communication, so torch.distributed is used only for the barriers and the
max-over-ranks of the timing.

Prints ONE JSON line on rank 0: whole-job particle-updates/s over the timed steps, plus
`roofline` (algorithmic bytes per launch / HIP-event launch duration against the HBM peak,
and the HBM traffic measured with rocprofv3 counters when a summary for this size is
present under profiles/) and, at N = 1, `cpu_baseline` (the CPU restatement of the
reference's update timed on the host cores).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGORITHMIC_BYTES_PER_UPDATE = 48      # 6 fp32 read + 6 fp32 written, colour mode 0
HBM_PEAK_GBPS = 8000.0                 # MI355X HBM3E datasheet peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--particles", type=int, default=100_000_000,
                    help="particles per GPU (default: the 100 M headline configuration)")
    ap.add_argument("--workload", choices=["single", "multi"], default="single",
                    help="single: one attractor, no communication (headline); "
                         "multi: one moving attractor per rank, all-gathered every step")
    ap.add_argument("--lookahead", type=int, default=1,
                    help="multi workload: steps of attractor state exchanged per all-gather "
                         "(1 = one collective per step)")
    ap.add_argument("--cpu-sample", type=int, default=20_000_000,
                    help="particles of the same workload stepped by the CPU baseline")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(sample, seconds):
    """The CPU restatement of the reference's update (oracle), on the reference's own
    64-byte AoS layout, OpenMP over the host cores this process may use."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import psim_testlib as T
    lib = T.load_oracle()
    threads = int(lib.psim_oracle_max_threads())
    params = T.bench_params()
    particles = T.oracle_gen_uniform_aos(0, sample)
    import ctypes
    ptr = particles.ctypes.data_as(ctypes.c_void_p)
    lib.psim_oracle_update_aos(ptr, sample, ctypes.byref(params), None, 0, threads)   # warm-up
    steps = 0
    t0 = time.perf_counter()
    while True:
        lib.psim_oracle_update_aos(ptr, sample, ctypes.byref(params), None, 0, threads)
        steps += 1
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    return {
        "value": sample * steps / dt,
        "unit": "particle-updates/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{sample} particles (first {sample} of the same synthetic state, AoS 64 B) x "
                  f"{steps} steps in {dt:.2f} s; update loop only, no upload copy",
    }


def load_traffic(particles):
    """HBM bytes per launch from the committed PMC summary, if it covers this size."""
    path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        with open(path) as f:
            s = json.load(f)
        if int(s.get("particles", -1)) == particles:
            return float(s["hbm_bytes_per_launch"])
    except (OSError, ValueError, KeyError):
        pass
    return None


def main():
    args = parse_args()
    # Native libraries (RCCL's start-up banner) write to file descriptor 1; keep it for the
    # single JSON line and send everything else to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch                       # first: the C-ABI library then shares torch's HIP runtime
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         f"python -m torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("no GPU visible: this benchmark has no CPU path")
    # PSIM_BENCH_REHEARSAL=1 (tests only): several ranks share the GPUs that exist and talk
    # over gloo, so the N > 1 control flow can be run on a one-GPU machine.  The numbers of
    # such a run mean nothing.
    rehearsal = os.environ.get("PSIM_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run (RANK set) the process group is used even for one rank
    distributed = world > 1 or args.workload == "multi" or "RANK" in os.environ
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if distributed:
            if rehearsal:
                dist.barrier()
            else:
                dist.barrier(device_ids=[local_rank])

    psim = importlib.import_module("particle-simulation-3d_amd")
    n = args.particles
    params = psim.SimParams(delta_time=0.016, gravity=0.0, color_mode=0, mouse_force=5.0,
                            mouse_radius=10.0, is_mouse_dragging=1, damping=0.99,
                            max_dist_for_color=50.0, mouse_position=(0.0, 0.0, 48.0))

    if args.workload == "multi":
        par = importlib.import_module("particle-simulation-3d_amd.parallel")
        runner = par.MultiAttractorRunner(n, rank, world, local_rank, params,
                                          lookahead=args.lookahead)
        step, finish, sim = runner.step, runner.finish, runner.sim
    else:
        sim = psim.HipParticleSimulation.new_shard(n, rank * n, world * n, psim.GEN_UNIFORM,
                                                   device_id=local_rank)
        step = lambda: sim.update(params)      # noqa: E731
        finish = lambda: None                  # noqa: E731

    for _ in range(args.warmup):
        step()
    finish()
    sim.sync()

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.timer_start()
    for _ in range(args.steps):
        step()
    finish()
    device_ms = sim.timer_stop()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    barrier()

    if rank == 0:
        total_updates = world * n * args.steps
        launch_s = device_ms / 1e3 / args.steps
        bytes_per_launch = ALGORITHMIC_BYTES_PER_UPDATE * n
        achieved = bytes_per_launch / launch_s / 1e9
        out = {
            "metric": "particle-updates/sec @100M particles, 1/2/4/8 GPU; % HBM roofline",
            "value": total_updates / elapsed,
            "unit": "particle-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"{n} particles per GPU, single attractor, SoA fp32, dt 0.016, "
                             "damping 0.99, attractor (0,0,48) r 10 f 5"
                             if args.workload == "single" else
                             f"{n} particles per GPU, {world} moving attractors all-gathered "
                             "over RCCL every step"),
                "particles_per_gpu": n,
                "total_particles": world * n,
                "sharding": f"contiguous index ranges x{world}, no data-path collective"
                            if args.workload == "single" else
                            (f"contiguous index ranges x{world}, all-gather of 32 B per rank per step"
                             if args.lookahead == 1 else
                             f"contiguous index ranges x{world}, all-gather of {args.lookahead} x 32 B "
                             f"per rank every {args.lookahead} steps"),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": ("psim::update_kernel<colour 0, single attractor>"
                           if args.workload == "single" else
                           "psim::update_kernel<colour 0, attractor list staged in LDS> "
                           "(launch_ms here also holds the waits for the per-step all-gather)"),
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "launch_ms": launch_s * 1e3,
                "traffic": load_traffic(n) if args.workload == "single" else None,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.cpu_seconds)
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    sim.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
