#!/bin/bash
# On the GPU box (from the repo root): the VALU-bound kernels under rocprofv3, one operation per
# process -- a kernel trace for the durations, then two counter passes (8 SQ counters each).
# Raw output: gpurun_out/prof/<tag>_valu/<op>/; scripts/summarize_valu_profile.py condenses it.
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/${TAG}_valu
cd /tmp && export TMPDIR=/tmp
PASS1="SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
PASS2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
for OP in single batch16 batch64 list8 list64; do
    mkdir -p "$OUT/$OP"
    timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$OP/trace" -o trace \
        -- python3 "$REPO/scripts/profile_valu.py" --op $OP > "$OUT/$OP/config.json" 2> "$OUT/$OP/trace.err" \
        || { echo "$OP: kernel-trace pass failed"; tail -5 "$OUT/$OP/trace.err"; exit 1; }
    timeout -s KILL 120 rocprofv3 --pmc $PASS1 --output-format csv -d "$OUT/$OP/pmc1" -o pmc \
        -- python3 "$REPO/scripts/profile_valu.py" --op $OP > /dev/null 2> "$OUT/$OP/pmc1.err" \
        || { echo "$OP: counter pass 1 failed"; tail -5 "$OUT/$OP/pmc1.err"; exit 1; }
    timeout -s KILL 120 rocprofv3 --pmc $PASS2 --output-format csv -d "$OUT/$OP/pmc2" -o pmc \
        -- python3 "$REPO/scripts/profile_valu.py" --op $OP > /dev/null 2> "$OUT/$OP/pmc2.err" \
        || { echo "$OP: counter pass 2 failed"; tail -5 "$OUT/$OP/pmc2.err"; exit 1; }
    echo "$OP done"
done
