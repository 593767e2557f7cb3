#!/bin/bash
# Runs on the GPU box (from the repo root): kernel-trace statistics of the default
# bench command, then HBM traffic counters in separate passes.  Raw output goes to
# gpurun_out/prof/<tag>/; scripts/summarize_profile.py condenses it into profiles/.
set -u
TAG=${1:-r01}
STEPS=${2:-100}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace \
    -- python3 "$REPO/bench.py" --steps "$STEPS" --warmup 10 --no-cpu-baseline --sustained-seconds 0 --no-copy-yardstick \
    > "$OUT/trace_bench.json" 2> "$OUT/trace.err" || { echo "kernel-trace pass failed"; tail -5 "$OUT/trace.err"; exit 1; }
echo "kernel-trace pass done"

for C in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -o pmc \
        -- python3 "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --sustained-seconds 0 --no-copy-yardstick \
        > "$OUT/pmc_${C}_bench.json" 2> "$OUT/pmc_$C.err" || { echo "pmc pass $C failed"; tail -5 "$OUT/pmc_$C.err"; exit 1; }
    echo "pmc pass $C done"
done

# every kernel of the path once each (scripts/profile_ops.py): durations, then counters
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ops_trace" -o trace \
    -- python3 "$REPO/scripts/profile_ops.py" \
    > "$OUT/ops_config.json" 2> "$OUT/ops_trace.err" || { echo "ops kernel-trace pass failed"; tail -5 "$OUT/ops_trace.err"; exit 1; }
echo "ops kernel-trace pass done"
for C in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d "$OUT/ops_pmc_$C" -o pmc \
        -- python3 "$REPO/scripts/profile_ops.py" --launches 3 \
        > /dev/null 2> "$OUT/ops_pmc_$C.err" || { echo "ops pmc pass $C failed"; tail -5 "$OUT/ops_pmc_$C.err"; exit 1; }
    echo "ops pmc pass $C done"
done
find "$OUT" -name "*.csv" | head -30
