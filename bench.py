#!/usr/bin/env python3
"""Benchmark of the particle-update hot path on MI355X.

A step is one update (one kernel launch) of this rank's particle shard.  The default, and
the workload the headline metric is quoted on, is `--config 3`: 100 M particles per GPU,
single attractor, fp32, synthetic uniform-random state generated on the device.  With N > 1
every rank holds its own shard (weak scaling: the per-GPU count is fixed; `--scaling strong`:
the total is fixed and divided).  The single-attractor path needs no communication, so
torch.distributed is used only for the barriers and the max-over-ranks of the timing.

  --config 1   100 k particles, single attractor: the reference's own CPU-runnable case (on the
               GPU a launch-bound 3 us step; the CPU restatement is timed beside it)
  --config 2   10 M particles, single attractor (fits the Infinity Cache: cache-assisted)
  --config 3   100 M particles per GPU, single attractor            (headline, default)
  --config 4   1 B particles over 8 GPUs = 125 M per GPU, single attractor, no collective
  --config 5   50 M particles over 8 GPUs = 6.25 M per GPU, one moving attractor per rank,
               all-gathered every step over RCCL by the library's own exchange

Prints ONE JSON line on rank 0: whole-job particle-updates/s over the timed steps, plus
`roofline` (algorithmic bytes per launch / HIP-event launch duration against the HBM peak,
and the HBM traffic measured with rocprofv3 counters when the committed summary under
profiles/ was taken from this very build) and, at N = 1, `cpu_baseline` (the CPU restatement
of the reference's update timed on the host cores) and `sustained` (the same step held for
seconds after the timed steps: the rate once clocks have settled).  At N > 1 the single-attractor
configurations append `exchange_check`: a short run of the config-5 step over all ranks,
outside the timed region, so that one multi-GPU run also shows what the collective delivers
and costs.
"""
from __future__ import annotations

import argparse
import hashlib
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGORITHMIC_BYTES_PER_UPDATE = 48      # 6 fp32 read + 6 fp32 written, colour mode 0
HBM_PEAK_GBPS = 8000.0                 # MI355X HBM3E datasheet peak
KERNEL_SOURCES = ("psim_kernels.hip", "psim_kernels.h", "psim_threshold.h", "psim_tuning.h",
                  "psim_tuning.cpp", "Makefile")

# name, particles per GPU as written (the 8-GPU configurations divided by 8), total when the
# particle count is held fixed instead (--scaling strong), workload
CONFIGS = {
    1: ("configs[0]: 100k particles, single point attractor (the reference's CPU-runnable case, "
        "here on the GPU with the CPU restatement timed beside it; launch-bound at this size)",
        100_000, 100_000, "single"),
    2: ("configs[1]: 10M particles, single attractor", 10_000_000, 10_000_000, "single"),
    3: ("configs[2]: 100M particles, single attractor", 100_000_000, 100_000_000, "single"),
    4: ("configs[3]: 1B particles over 8 GPUs (125M per GPU), single attractor, no collective",
        125_000_000, 1_000_000_000, "single"),
    5: ("configs[4]: 50M particles over 8 GPUs (6.25M per GPU), one moving attractor per rank, "
        "RCCL all-gather of attractor state every step", 6_250_000, 50_000_000, "multi"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=3,
                    help="which configuration of the baseline to run (default 3: 100 M per GPU)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: the per-GPU particle count is fixed; strong: the total is")
    ap.add_argument("--particles", type=int, default=None,
                    help="particles per GPU (weak) or in total (strong); overrides --config's count")
    ap.add_argument("--workload", choices=["single", "multi"], default=None,
                    help="single: one attractor, no communication; multi: one moving attractor "
                         "per rank, all-gathered every step (default: what --config says)")
    ap.add_argument("--pacing", choices=["host", "device", "pipelined"], default=None,
                    help="multi workload: who waits for a step's all-gather -- the calling thread "
                         "(host: kernels run back to back; the default on one rank), the kernel's "
                         "stream (device: the host runs ahead), or the calling thread one call late "
                         "(pipelined, the default on several ranks: kernels back to back AND every "
                         "all-gather already waiting when the previous kernel's grid runs dry, the "
                         "one moment a collective's workgroup finds room beside the update kernel; "
                         "see psim_exchange_set_pacing)")
    ap.add_argument("--cpu-sample", type=int, default=None,
                    help="particles stepped by the CPU baseline (default: the GPU's count)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustained-seconds", type=float, default=None,
                    help="after the timed steps, keep stepping for about this long and report "
                         "the rate held (`sustained`; 0: leave it out; default 6, and 0 in a "
                         "rehearsal of several ranks on one GPU)")
    ap.add_argument("--no-exchange-check", action="store_true")
    ap.add_argument("--no-copy-yardstick", action="store_true",
                    help="leave out roofline.device_copy (the runtime's device-to-device copy of "
                         "the same bytes, timed in the same run at N = 1)")
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
        "sample": f"{sample} particles (the same synthetic state from particle 0 on, AoS 64 B) x "
                  f"{steps} steps in {dt:.2f} s; update loop only, no upload copy",
    }


def copy_yardstick(torch, moved_bytes, reps=20):
    """A yardstick taken in the same run on the same GPU: the runtime's own device-to-device
    copy (torch's tensor copy) moving the bytes one update moves, half read and half written.
    Nothing of the product runs here; it only says what a plain copy reaches on this box."""
    words = max(moved_bytes // 2 // 4, 1)
    src = torch.rand(words, dtype=torch.float32, device="cuda")
    dst = torch.empty_like(src)
    for _ in range(3):
        dst.copy_(src)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for _ in range(reps):
        dst.copy_(src)
    stop.record()
    torch.cuda.synchronize()
    ms = start.elapsed_time(stop) / reps
    del src, dst
    torch.cuda.empty_cache()
    return {"what": "torch device-to-device copy of a float32 tensor, bytes read + written = the "
                    "update's algorithmic bytes per launch", "bytes": words * 8,
            "ms": ms, "achieved": words * 8 / (ms / 1e3) / 1e9, "unit": "GB/s"}


def host_memory_allows(particles):
    """The CPU leg holds `particles` 64-byte records; keep well inside the free memory."""
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    return particles * 64 * 2 < int(line.split()[1]) * 1024
    except OSError:
        pass
    return particles <= 20_000_000


def sha256_of(paths):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            for block in iter(lambda: f.read(1 << 20), b""):
                h.update(block)
    return h.hexdigest()


def build_identity(psim):
    csrc = os.path.dirname(psim.LIBRARY_PATH)
    return {"library_sha256": sha256_of([psim.LIBRARY_PATH]),
            "kernel_sources_sha256": sha256_of([os.path.join(csrc, n) for n in KERNEL_SOURCES])}


def load_traffic(particles, identity):
    """HBM bytes per launch from the committed counter summary, when it was taken at this
    size from this build: the same library file, or one built from the same kernel sources."""
    path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    info = {"traffic": None, "traffic_profiled_git_sha": None, "traffic_build_match": "no summary"}
    try:
        with open(path) as f:
            s = json.load(f)
    except (OSError, ValueError):
        return info
    info["traffic_profiled_git_sha"] = s.get("git_sha")
    if int(s.get("particles", -1)) != particles:
        info["traffic_build_match"] = f"summary is for {s.get('particles')} particles"
    elif s.get("library_sha256") == identity["library_sha256"]:
        info["traffic_build_match"] = "library"
    elif s.get("kernel_sources_sha256") == identity["kernel_sources_sha256"]:
        info["traffic_build_match"] = "kernel sources"
    else:
        info["traffic_build_match"] = "none: profiled build differs from this one"
    if info["traffic_build_match"] in ("library", "kernel sources"):
        info["traffic"] = float(s["hbm_bytes_per_launch"])
    return info


# The extra multi-GPU run after the headline: the exchange itself gives up on a collective
# that does not complete (psim_exchange_set_timeout), the watchdog is the net under that.
# (The variable shortens the watchdog for the test that rehearses an abandoned check.)
EXCHANGE_CHECK_SECONDS = float(os.environ.get("PSIM_BENCH_EXCHANGE_CHECK_SECONDS", "120"))
EXCHANGE_TIMEOUT_MS = 30_000
EXIT_EXCHANGE_CHECK_FAILED = 3


class Watchdog:
    """Gives up on the code it guards after `seconds`, or when the launcher sends SIGTERM
    (a sibling rank died): calls `give_up(reason)` on a thread of its own -- the main thread may
    be blocked inside a collective that will never finish -- and ends the process with
    `status`, which is not 0: what was guarded did not complete."""

    def __init__(self, seconds, give_up, status):
        import select
        import signal
        import threading
        self._signal = signal
        self._sig_r, sig_w = os.pipe()
        os.set_blocking(sig_w, False)
        self._stop_r, self._stop_w = os.pipe()
        self._old_fd = signal.set_wakeup_fd(sig_w, warn_on_full_buffer=False)
        self._old_handler = signal.signal(signal.SIGTERM, lambda *a: None)

        def watch():
            ready, _, _ = select.select([self._sig_r, self._stop_r], [], [], seconds)
            if self._stop_r in ready:
                return
            give_up("terminated by the launcher" if ready else f"no result after {seconds:.0f} s")
            os._exit(status)

        self._thread = threading.Thread(target=watch, daemon=True)
        self._thread.start()

    def cancel(self):
        os.write(self._stop_w, b"x")
        self._thread.join()
        self._signal.signal(self._signal.SIGTERM, self._old_handler)
        self._signal.set_wakeup_fd(self._old_fd)


def exchange_check(psim, par, torch, dist, rank, world, local_rank, params, steps=200, warmup=20,
                   rehearsal=False):
    """The config-5 step (6.25 M particles per rank, one moving attractor per rank, RCCL
    all-gather every step through psim_exchange_*) over all ranks, untimed by the headline:
    what the collective delivers (bit for bit against the closed-form paths) and what a step
    costs against the same kernel fed from a host list."""
    import numpy as np
    n = 6_250_000
    dev = "cpu" if rehearsal else "cuda"             # a rehearsal talks over gloo
    ident = torch.zeros(psim.EXCHANGE_ID_BYTES, dtype=torch.uint8, device=dev)
    if rank == 0:
        ident.copy_(torch.frombuffer(bytearray(psim.exchange_unique_id()), dtype=torch.uint8))
    dist.broadcast(ident, 0)
    runner = par.NativeMultiAttractorRunner(n, rank, world, local_rank, params,
                                            bytes(ident.cpu().numpy().tobytes()),
                                            timeout_ms=EXCHANGE_TIMEOUT_MS)
    for _ in range(warmup):
        runner.step()
    runner.finish()
    if rehearsal:
        dist.barrier()
    else:
        dist.barrier(device_ids=[local_rank])
    runner.sim.timer_start()
    t0 = time.perf_counter()
    for _ in range(steps):
        runner.step()
    device_ms = runner.sim.timer_stop()
    runner.finish()
    wall = time.perf_counter() - t0
    # the same again with the calling thread waiting for each gather (kernels back to back)
    runner.exchange.set_pacing(psim.EXCHANGE_HOST_PACED)
    for _ in range(warmup):
        runner.step()
    runner.finish()
    barrier = dist.barrier if rehearsal else (lambda: dist.barrier(device_ids=[local_rank]))
    barrier()
    runner.sim.timer_start()
    for _ in range(steps):
        runner.step()
    host_paced_ms = runner.sim.timer_stop()
    runner.finish()
    # and one call late (kernels back to back, every gather enqueued ahead of the kernel before)
    runner.exchange.set_pacing(psim.EXCHANGE_PIPELINED)
    for _ in range(warmup):
        runner.step()
    runner.finish()
    barrier()
    runner.sim.timer_start()
    for _ in range(steps):
        runner.step()
    pipelined_ms = runner.sim.timer_stop()
    runner.finish()
    last = 3 * (warmup + steps) - 1
    ok = runner.exchange.last_list().tobytes() == par.attractors_at(last, world).tobytes()
    # the same kernel without the exchange: the list handed over by the host
    table = par.attractors_at(last, world)
    for _ in range(warmup):
        runner.sim.update_multi(params, table)
    runner.sim.sync()
    runner.sim.timer_start()
    for _ in range(steps):
        runner.sim.update_multi(params, table)
    kernel_ms = runner.sim.timer_stop()
    runner.close()
    t = torch.tensor([wall / steps * 1e6, device_ms / steps * 1e3, kernel_ms / steps * 1e3,
                      0.0 if ok else 1.0, host_paced_ms / steps * 1e3, pipelined_ms / steps * 1e3],
                     dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    w, d, k, bad, hp, pl = (float(v) for v in t.tolist())
    return {
        "workload": f"{n} particles per GPU, {world} moving attractors, psim_exchange_step "
                    f"(RCCL all-gather of 32 B per rank into a ring of receive lists) x {steps} steps",
        "lists_match_closed_form": bad == 0.0,
        "us_per_step_wall": w,
        "us_per_step_device": d,
        "us_per_step_device_host_paced": hp,
        "us_per_step_device_pipelined": pl,
        "us_per_step_kernel_only": k,
        "aggregation": "max over ranks",
    }


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
    config_name, per_gpu, fixed_total, workload = CONFIGS[args.config]
    if args.workload:
        workload = args.workload
    if args.pacing is None:
        args.pacing = "host" if world == 1 else "pipelined"
    # PSIM_BENCH_REHEARSAL=1 (tests only): several ranks share the GPUs that exist and talk
    # over gloo, so the N > 1 control flow can be run on a one-GPU machine.  The numbers of
    # such a run mean nothing.
    rehearsal = os.environ.get("PSIM_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run (RANK set) the process group is used even for one rank
    distributed = world > 1 or "RANK" in os.environ
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
    par = importlib.import_module("particle-simulation-3d_amd.parallel")
    # this rank's shard: [first, first + n) of `total`
    if args.scaling == "strong":
        total = args.particles if args.particles is not None else fixed_total
        first, n = par.shard_range(total, rank, world)
    else:
        n = args.particles if args.particles is not None else per_gpu
        total, first = world * n, rank * n
    params = psim.SimParams(delta_time=0.016, gravity=0.0, color_mode=0, mouse_force=5.0,
                            mouse_radius=10.0, is_mouse_dragging=1, damping=0.99,
                            max_dist_for_color=50.0, mouse_position=(0.0, 0.0, 48.0))

    if workload == "multi":
        if rehearsal and world > 1 and not os.environ.get("PSIM_RCCL_LIBRARY"):
            raise SystemExit("RCCL cannot put two ranks on one GPU: a rehearsal of the exchange "
                             "needs a stand-in named by PSIM_RCCL_LIBRARY")
        if world > 1:
            ident = torch.zeros(psim.EXCHANGE_ID_BYTES, dtype=torch.uint8,
                                device="cpu" if rehearsal else "cuda")
            if rank == 0:
                ident.copy_(torch.frombuffer(bytearray(psim.exchange_unique_id()),
                                             dtype=torch.uint8))
            dist.broadcast(ident, 0)
            unique_id = ident.cpu().numpy().tobytes()
        else:
            unique_id = psim.exchange_unique_id()
        runner = par.NativeMultiAttractorRunner(n, rank, world, local_rank, params, unique_id,
                                                first=first, total=total, pacing=args.pacing)
        step, finish, sim = runner.step, runner.finish, runner.sim
    else:
        runner = None
        sim = psim.HipParticleSimulation.new_shard(n, first, total, psim.GEN_UNIFORM,
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
    device_ms = sim.timer_stop()
    finish()
    torch.cuda.synchronize()
    own_elapsed = elapsed = time.perf_counter() - t0
    per_rank = [[own_elapsed, device_ms, float(n)]]
    if distributed:
        dev = "cpu" if rehearsal else "cuda"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor(per_rank[0], dtype=torch.float64, device=dev)
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        per_rank = [[float(v) for v in r.tolist()] for r in everyone]
    barrier()

    # After the K timed steps: the same step for seconds on end, untimed by the headline.  K
    # steps take milliseconds; this says what the device holds once clocks and temperature
    # have settled (and keeps the GPU busy long enough for an outside sampler to see it).
    sustained = None
    hold_seconds = args.sustained_seconds
    if hold_seconds is None:
        hold_seconds = 0.0 if rehearsal else 6.0
    if hold_seconds > 0:
        # The number of steps comes from `elapsed`, which is the same on every rank after the
        # reduction above: with one attractor per rank a step is a collective, and ranks that
        # took different numbers of steps would wait for each other for ever.
        k = int(hold_seconds / max(elapsed / args.steps, 1e-7))
        k = max(args.steps, min(k, 1_000_000))
        t1 = time.perf_counter()
        sim.timer_start()
        for _ in range(k):
            step()
        held_ms = sim.timer_stop()
        finish()
        sim.sync()
        held_s = time.perf_counter() - t1
        slowest = [held_s, held_ms]
        if distributed:
            t = torch.tensor(slowest, dtype=torch.float64, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            slowest = [float(v) for v in t.tolist()]
        sustained = {"steps": k, "seconds": slowest[0], "ms_per_step": slowest[0] / k * 1e3,
                     "launch_ms": slowest[1] / k, "value": total * k / slowest[0],
                     "unit": "particle-updates/s", "aggregation": "max over ranks"}
        barrier()

    yardstick = None
    if world == 1 and not rehearsal and not args.no_copy_yardstick:
        yardstick = copy_yardstick(torch, ALGORITHMIC_BYTES_PER_UPDATE * n)

    out = None
    if rank == 0:
        total_updates = total * args.steps
        wall_ms = [r[0] / args.steps * 1e3 for r in per_rank]
        launch_ms = [r[1] / args.steps for r in per_rank]
        # the roofline is a per-GPU statement: the slowest rank's launch and its own shard
        slow = max(range(world), key=lambda r: launch_ms[r])
        bytes_per_launch = ALGORITHMIC_BYTES_PER_UPDATE * int(per_rank[slow][2])
        achieved = bytes_per_launch / (launch_ms[slow] / 1e3) / 1e9
        identity = build_identity(psim)
        traffic = load_traffic(n, identity) if workload == "single" else \
            {"traffic": None, "traffic_profiled_git_sha": None,
             "traffic_build_match": "counters are collected for the single-attractor kernel"}
        if workload == "single":
            sharding = f"contiguous index ranges x{world}, no data-path collective"
            kernel = "psim::update_kernel<colour 0, single attractor>"
            what = "single attractor, SoA fp32, dt 0.016, damping 0.99, attractor (0,0,48) r 10 f 5"
        else:
            sharding = (f"contiguous index ranges x{world}, all-gather of 32 B per rank every step "
                        f"(psim_exchange_step: RCCL behind the C ABI, {args.pacing}-paced)")
            kernel = ("psim::update_kernel<colour 0, attractor list staged in LDS> "
                      "(launch_ms here also holds the waits for the per-step all-gather)")
            what = f"{world} moving attractors (one per rank, r 10 f 5) all-gathered over RCCL"
        out = {
            "metric": "particle-updates/sec @100M particles, 1/2/4/8 GPU; % HBM roofline",
            "value": total_updates / elapsed,
            "unit": "particle-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{config_name}: {n} particles per GPU x {world} GPU, {what}",
                "baseline_config": args.config,
                "particles_per_gpu": n,
                "total_particles": total,
                "sharding": sharding,
            },
            "per_rank_ms_per_step": {"min": min(wall_ms), "max": max(wall_ms)},
            "roofline": {
                "bound": "hbm",
                "kernel": kernel,
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "launch_ms": launch_ms[slow],
                "per_rank_launch_ms": {"min": min(launch_ms), "max": max(launch_ms)},
                "per": "GPU (the slowest rank's launches over its own shard)",
                **traffic,
                **identity,
            },
        }
        if sustained is not None:
            out["sustained"] = sustained
        if yardstick is not None:
            out["roofline"]["device_copy"] = yardstick
        if world == 1 and not args.no_cpu_baseline:
            sample = args.cpu_sample
            note = ""
            if sample is None:
                sample = n if host_memory_allows(n) else 20_000_000
                if sample != n:
                    note = f"; {n} records do not fit the free host memory"
            out["cpu_baseline"] = cpu_baseline(sample, args.cpu_seconds)
            out["cpu_baseline"]["sample"] += note

    def emit(extra=None):
        if rank == 0:
            line = dict(out, exchange_check=extra) if extra is not None else out
            os.write(json_fd, (json.dumps(line) + "\n").encode())

    # (a rehearsal shares one GPU between the ranks, which RCCL refuses: it runs the check only
    # when PSIM_RCCL_LIBRARY names a stand-in, as the tests do)
    if world > 1 and workload == "single" and not args.no_exchange_check and \
            (not rehearsal or os.environ.get("PSIM_RCCL_LIBRARY")):
        # The headline numbers are final at this point.  Whatever happens to the extra run --
        # an error, a hang, a rank killed by the launcher -- the line still goes out.
        sim.close()
        sim = None
        def abandoned(why):
            print(f"rank {rank}: exchange check abandoned: {why}", file=sys.stderr, flush=True)
            emit({"error": f"exchange check abandoned: {why}"})

        guard = Watchdog(EXCHANGE_CHECK_SECONDS, abandoned, EXIT_EXCHANGE_CHECK_FAILED)
        failed = False
        try:
            check = exchange_check(psim, par, torch, dist, rank, world, local_rank, params,
                                   rehearsal=rehearsal)
        except Exception as e:                       # noqa: BLE001 -- reported in the line
            check = {"error": f"{type(e).__name__}: {e}"}
            failed = True
            print(f"rank {rank}: exchange check failed: {check['error']}", file=sys.stderr,
                  flush=True)
        guard.cancel()
        if not failed and not check.get("lists_match_closed_form", False):
            failed = True
            print(f"rank {rank}: exchange check: the gathered lists differ from the closed-form "
                  "paths", file=sys.stderr, flush=True)
        emit(check)
        # The ranks left the check together (its last step is a reduction over all of them), or
        # some of them never will: either way each ends here on its own -- no further barrier
        # that a rank which has gone could hold up, no tear-down of two communicators.  The
        # headline line is out in every case; the exit status says whether the check completed.
        sys.stderr.flush()
        os._exit(EXIT_EXCHANGE_CHECK_FAILED if failed else 0)
    else:
        emit()

    if runner is not None:
        runner.close()
    elif sim is not None:
        sim.close()
    if distributed:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
