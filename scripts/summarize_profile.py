#!/usr/bin/env python3
"""Condenses gpurun_out/prof/<tag>/ (written by scripts/profile.sh) into profiles/:

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, verbatim
  profiles/<tag>_pmc.csv            per-kernel mean of FETCH_SIZE / WRITE_SIZE
  profiles/<tag>_summary.md         the numbers side by side
  profiles/pmc_summary.json         HBM bytes per launch of the update kernel (bench.py
                                    reports it as roofline.traffic)

gfx950 corrections applied to the counters: both are in KiB; FETCH_SIZE counts a wide
coalesced read stream at half its bytes, so it is doubled; WRITE_SIZE is exact for
16-byte-per-lane streaming stores.
"""
import csv
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mean_counter(path, counter, kernel_substr):
    vals = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter and kernel_substr in row["Kernel_Name"]:
                vals.append(float(row["Counter_Value"]))
    return statistics.mean(vals), len(vals)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", "prof", tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"),
                os.path.join(dst, f"{tag}_kernel_stats.csv"))
    bench = json.load(open(os.path.join(src, "trace_bench.json")))
    n = bench["config"]["particles_per_gpu"]

    kernel = "update_kernel"
    stats = {}
    with open(os.path.join(src, "trace", "trace_kernel_stats.csv")) as f:
        for row in csv.DictReader(f):
            if kernel in row["Name"]:
                stats = row
    fetch_kib, nf = mean_counter(os.path.join(src, "pmc_FETCH_SIZE", "pmc_counter_collection.csv"),
                                 "FETCH_SIZE", kernel)
    write_kib, nw = mean_counter(os.path.join(src, "pmc_WRITE_SIZE", "pmc_counter_collection.csv"),
                                 "WRITE_SIZE", kernel)
    read_bytes = fetch_kib * 1024 * 2          # gfx950: wide coalesced reads count half
    write_bytes = write_kib * 1024
    total = read_bytes + write_bytes
    with open(os.path.join(dst, f"{tag}_pmc.csv"), "w") as f:
        f.write("kernel,counter,launches,mean_value_KiB,bytes_per_launch_corrected\n")
        f.write(f"{kernel},FETCH_SIZE,{nf},{fetch_kib:.3f},{read_bytes:.0f}\n")
        f.write(f"{kernel},WRITE_SIZE,{nw},{write_kib:.3f},{write_bytes:.0f}\n")
    # Which build the counters belong to: the hashes bench.py computed on the GPU box in the
    # traced run, and the commit of the tree they were taken from (summarised right after).
    import subprocess
    try:
        git_sha = subprocess.check_output(["git", "rev-parse", "HEAD"], cwd=ROOT, text=True).strip()
        if subprocess.check_output(["git", "status", "--porcelain", "--", "particle-simulation-3d_amd",
                                    "include"], cwd=ROOT, text=True).strip():
            git_sha += "+uncommitted"
    except (OSError, subprocess.CalledProcessError):
        git_sha = None
    summary = {
        "tag": tag,
        "git_sha": git_sha,
        "library_sha256": bench["roofline"].get("library_sha256"),
        "kernel_sources_sha256": bench["roofline"].get("kernel_sources_sha256"),
        "particles": n,
        "kernel": stats.get("Name", kernel),
        "launches_traced": int(stats.get("Calls", 0)),
        "avg_launch_ns": float(stats.get("AverageNs", 0)),
        "hbm_read_bytes_per_launch": read_bytes,
        "hbm_write_bytes_per_launch": write_bytes,
        "hbm_bytes_per_launch": total,
        "algorithmic_bytes_per_launch": 48 * n,
        "bench_launch_ms": bench["roofline"]["launch_ms"],
        "bench_achieved_GBps": bench["roofline"]["achieved"],
    }
    json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
    avg_ms = summary["avg_launch_ns"] / 1e6
    with open(os.path.join(dst, f"{tag}_summary.md"), "w") as f:
        f.write(f"# {tag}: `python bench.py --steps {bench['steps']} --warmup {bench['warmup']}` "
                f"under rocprofv3 (MI355X, {n} particles)\n\n")
        f.write("The profiled command stops after the timed steps (`--no-cpu-baseline "
                "--sustained-seconds 0 --no-copy-yardstick`), so every traced launch of the update "
                "kernel is a warm-up or a timed step.\n\n")
        f.write("| quantity | value |\n|---|---|\n")
        f.write(f"| update kernel launches traced | {summary['launches_traced']} |\n")
        f.write(f"| rocprofv3 average launch duration | {avg_ms:.4f} ms |\n")
        f.write(f"| bench.py HIP-event launch duration (same run) | {summary['bench_launch_ms']:.4f} ms |\n")
        f.write(f"| algorithmic bytes per launch (48 B x N) | {48 * n:.4e} |\n")
        f.write(f"| achieved (algorithmic / rocprof duration) | {48 * n / avg_ms / 1e6:.1f} GB/s "
                f"= {48 * n / avg_ms / 1e6 / 8000:.3f} of 8 TB/s |\n")
        f.write(f"| FETCH_SIZE mean | {fetch_kib:.1f} KiB -> x1024 x2 = {read_bytes:.4e} B read |\n")
        f.write(f"| WRITE_SIZE mean | {write_kib:.1f} KiB -> x1024 = {write_bytes:.4e} B written |\n")
        f.write(f"| HBM traffic per launch | {total:.4e} B = {total / (48 * n):.4f} x algorithmic |\n")
    print(open(os.path.join(dst, f"{tag}_summary.md")).read())


if __name__ == "__main__":
    main()
