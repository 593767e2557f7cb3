#!/bin/bash
# One rank of the 50 M configuration (bench.py --config 5) with the exchange's communication
# stream at the highest and at normal priority (-> profiles/tuning/r03_exchange_comm_stream_priority.log).
cd "$(dirname "$0")/.."
for rep in 1 2; do
for pacing in device host; do
  for prio in high normal; do
    if [ $prio = high ]; then export PSIM_EXPERIMENT=1 PSIM_EXCHANGE_COMM_PRIORITY=1; else unset PSIM_EXPERIMENT PSIM_EXCHANGE_COMM_PRIORITY; fi
    timeout -k 10 120 python bench.py --config 5 --steps 3000 --warmup 300 --pacing $pacing --no-cpu-baseline 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$pacing-paced, comm priority $prio: %.2f us per step (launch %.2f us)' % (d['ms_per_step']*1e3, d['roofline']['launch_ms']*1e3))" || exit 1
  done
done
done
