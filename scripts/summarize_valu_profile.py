#!/usr/bin/env python3
"""Condenses gpurun_out/prof/<tag>_valu/ (scripts/profile_valu.sh) into
profiles/<tag>_valu_summary.md: for the kernels that are bound by vector-instruction issue
rather than by HBM -- fused batches, long attractor lists -- the instructions they issue, the
cycles the chip spent, and the fraction of the fp32 vector issue peak that is.

Issue model (MI355X: 256 CUs x 4 SIMDs; a SIMD issues a 64-lane vector instruction over
2 cycles, a transcendental one -- v_rcp / v_rsq / v_sqrt, the cores of the correctly rounded
divide and square root -- over 8): a launch cannot take fewer than
(2 x ordinary + 8 x transcendental instructions) / 1024 SIMD cycles.  The clock is the one the
chip held during the launch: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / duration."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS = 256 * 4
OPS = (("single", "one step per launch (the HBM-bound headline kernel, for comparison)"),
       ("batch16", "16 fused steps, `update_batch_kernel`"),
       ("batch64", "64 fused steps, `update_batch_kernel`"),
       ("list8", "8 attractors r = 10 through the pair queue"),
       ("list64", "64 attractors r = 10 through the pair queue"))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", "prof", f"{tag}_valu")
    rows = []
    for op, what in OPS:
        cfg = json.load(open(os.path.join(src, op, "config.json")))
        stem = cfg["kernel"].split("<")[0]
        c = collections.defaultdict(list)
        for p in ("pmc1", "pmc2"):
            with open(os.path.join(src, op, p, "pmc_counter_collection.csv")) as f:
                for r in csv.DictReader(f):
                    if stem in r["Kernel_Name"]:
                        c[r["Counter_Name"]].append(float(r["Counter_Value"]))
        m = {k: sum(v) / len(v) for k, v in c.items()}
        dur = []
        with open(os.path.join(src, op, "trace", "trace_kernel_trace.csv")) as f:
            for r in csv.DictReader(f):
                if stem in r["Kernel_Name"]:
                    dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        ms = sum(dur) / len(dur) / 1e6
        n, steps = cfg["particles"], cfg["steps_per_launch"]
        valu, trans = m["SQ_INSTS_VALU"], m["SQ_INSTS_VALU_TRANS_F32"]
        cycles = m["GRBM_GUI_ACTIVE"] / 8                      # per XCD = chip cycles of the launch
        # the counter pass and the trace are different runs: the clock from the counter run's
        # own duration is not available, so the trace duration stands in
        ghz = cycles / (ms * 1e6)
        floor_cycles = (2 * (valu - trans) + 8 * trans) / SIMDS
        rows.append({
            "op": op, "what": what, "ms": ms, "ms_events": cfg["hip_event_ms_per_launch"],
            "valu": valu, "trans": trans, "salu": m["SQ_INSTS_SALU"], "lds": m["SQ_INSTS_LDS"],
            "per_particle_step": valu * 64 / (n * steps), "trans_per_particle_step": trans * 64 / (n * steps),
            "cycles": cycles, "ghz": ghz, "floor_cycles": floor_cycles, "frac": floor_cycles / cycles,
            "waves_per_simd": m["SQ_WAVE_CYCLES"] * 4 / (cycles * SIMDS),
            "wait_inst": m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
            "wait_any": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
            "hbm_frac": 48 * n / (ms * 1e-3) / 8e12,
            "fma": m["SQ_INSTS_VALU_FMA_F32"], "add": m["SQ_INSTS_VALU_ADD_F32"], "mul": m["SQ_INSTS_VALU_MUL_F32"],
            "updates_per_s": n * steps / (ms * 1e-3),
        })
    out = os.path.join(ROOT, "profiles", f"{tag}_valu_summary.md")
    with open(out, "w") as f:
        f.write(f"# {tag}: the kernels bound by vector-instruction issue, under rocprofv3 counters "
                f"(`scripts/profile_valu.sh`, MI355X, {rows[0]['op'] and json.load(open(os.path.join(src, 'single', 'config.json')))['particles']} particles)\n\n")
        f.write("Counters: two `--pmc` passes of 8 SQ counters (+ `GRBM_GUI_ACTIVE`), one operation per "
                "process; durations from a separate `--kernel-trace` run.  `SQ_WAVES` = 390 625 in every "
                "pass (the grid), so the SQ counters are sums over the whole chip.\n\n"
                "Issue floor: a SIMD issues a 64-lane vector instruction over 2 cycles, a transcendental "
                "one (`v_rcp/v_rsq/v_sqrt`: the cores of the correctly rounded divide and square root) over "
                "8; 1024 SIMDs.  `of issue peak` = floor cycles / cycles the launch took (`GRBM_GUI_ACTIVE` "
                "/ 8 XCDs).  For comparison the last column prices the same launch against HBM.  The floor "
                "is a lower bound of the cost: measured one kind at a time (`scripts/exp_valu_issue.hip`, "
                "`profiles/tuning/r02_valu_issue_costs.log`), compares, selects, the `v_div_*` helpers, "
                "data-parallel primitives and packed fp32 forms take two issue slots each and `v_rcp / "
                "v_sqrt` 3.5; the fused batch kernel's code averages 1.55 slots per instruction, so its "
                "0.59 here is 0.8-0.9 of what the vector unit can issue of that mix.\n\n")
        f.write("| operation | launch ms | particle-updates/s | VALU instr / particle-step | of them transcendental | "
                "clock GHz | issue-floor cycles | cycles taken | **of issue peak** | waves / SIMD | "
                "wave time stalled on issue | wave time parked (waitcnt) | of 8 TB/s (48 B/particle/launch) |\n")
        f.write("|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            f.write(f"| {r['what']} | {r['ms']:.3f} | {r['updates_per_s']:.3e} | {r['per_particle_step']:.1f} | "
                    f"{r['trans_per_particle_step']:.2f} | {r['ghz']:.2f} | {r['floor_cycles']:.3e} | "
                    f"{r['cycles']:.3e} | **{r['frac']:.3f}** | {r['waves_per_simd']:.1f} | "
                    f"{r['wait_inst']:.2f} | {r['wait_any']:.2f} | {r['hbm_frac']:.3f} |\n")
        f.write("\nInstruction mix per launch (wave instructions): \n\n"
                "| operation | VALU | fma f32 | add f32 | mul f32 | transcendental | SALU | LDS |\n|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            f.write(f"| {r['op']} | {r['valu']:.4e} | {r['fma']:.3e} | {r['add']:.3e} | {r['mul']:.3e} | "
                    f"{r['trans']:.3e} | {r['salu']:.3e} | {r['lds']:.3e} |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
