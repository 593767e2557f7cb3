#!/usr/bin/env python3
"""Joins the rocprofv3 output of scripts/profile_ops.py (run by scripts/profile.sh) into
profiles/<tag>_ops_summary.md: for every kernel of the path the average duration of its
last launches, the HBM bytes read and written per launch from the FETCH_SIZE / WRITE_SIZE
passes (gfx950: KiB units; FETCH_SIZE doubled, see summarize_profile.py) and the ratio
to the kernel's algorithmic bytes."""
import csv
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def squeeze(s):
    return s.replace(" ", "")


def last_rows(path, name_col, fragment, count):
    rows = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if squeeze(fragment) in squeeze(row[name_col]):
                rows.append(row)
    return rows[-count:]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", "prof", tag)
    cfg = json.load(open(os.path.join(src, "ops_config.json")))
    n, launches = cfg["particles"], cfg["launches"]
    trace = os.path.join(src, "ops_trace", "trace_kernel_trace.csv")
    lines = []
    for k in cfg["kernels"]:
        rows = last_rows(trace, "Kernel_Name", k["name"], launches)
        if not rows:
            continue
        dur = statistics.mean(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
        vals = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            path = os.path.join(src, f"ops_pmc_{counter}", "pmc_counter_collection.csv")
            got = [r for r in last_rows(path, "Kernel_Name", k["name"], 3)
                   if r["Counter_Name"] == counter]
            vals[counter] = statistics.mean(float(r["Counter_Value"]) for r in got) if got else None
        alg = k["algorithmic_bytes"]
        rd = vals["FETCH_SIZE"] * 1024 * 2 if vals["FETCH_SIZE"] is not None else None
        wr = vals["WRITE_SIZE"] * 1024 if vals["WRITE_SIZE"] is not None else None
        lines.append((k, rows[0], dur, rd, wr, alg))
    out = os.path.join(ROOT, "profiles", f"{tag}_ops_summary.md")
    with open(out, "w") as f:
        f.write(f"# {tag}: every kernel of the path under rocprofv3 (`scripts/profile_ops.py`, "
                f"MI355X, {n} particles)\n\n")
        f.write("Durations: `--kernel-trace`, mean of the last launches of each kernel. HBM bytes: "
                "separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, KiB x 1024, FETCH_SIZE x 2 "
                "(gfx950 counts wide coalesced reads at half).\n\n")
        f.write("| kernel | configuration | launch ms | algorithmic GB/s | of 8 TB/s | HBM read B | "
                "HBM written B | traffic / algorithmic | VGPRs | static LDS B |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for k, row, dur, rd, wr, alg in lines:
            ms = dur / 1e6
            gbps = alg / dur
            total = (rd or 0) + (wr or 0)
            # a kernel bound by instruction issue is priced in <tag>_valu_summary.md, not here
            frac = f"{gbps / 8000:.3f}" if k.get("bound", "hbm") == "hbm" else \
                f"not HBM-bound: `{tag}_valu_summary.md`"
            f.write(f"| `{k['name']}` | {k['what']} | {ms:.4f} | {gbps:.0f} | {frac} | "
                    f"{rd:.4e} | {wr:.4e} | {total / alg:.4f} | {row['VGPR_Count']} | "
                    f"{row['LDS_Block_Size']} |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
