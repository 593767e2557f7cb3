#!/bin/bash
# One rank of the 50 M configuration (bench.py --config 5: 6.25 M particles, the exchange every
# step) over the RCCL test double, whose all-gather kernel is either a single wave or has the
# resource footprint of RCCL's own device kernel (PSIM_FAKE_RCCL_FAT=1: 37 664 B of LDS, 248
# vector registers, 512 threads -- a workgroup that only finds room when the update kernel's
# grid runs dry), with the step paced from the host and from the device.
export PSIM_RCCL_LIBRARY=$(pwd)/tests/cpp/libfake_rccl.so
export PSIM_FAKE_RCCL_SPIN=100000000      # the double's helper thread never naps: a gather then takes ~10 us
for fat in 0 1; do
  for pacing in host device pipelined; do
    PSIM_FAKE_RCCL_FAT=$fat timeout -k 10 120 python bench.py --config 5 --steps 3000 --warmup 300 \
        --pacing $pacing --no-cpu-baseline --sustained-seconds 0 --no-copy-yardstick 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('gather kernel %s, %s-paced: %.2f us per step (device %.2f)' % ('RCCL-sized' if $fat else 'single wave', '$pacing', d['ms_per_step']*1e3, d['roofline']['launch_ms']*1e3))" || exit 1
  done
done
