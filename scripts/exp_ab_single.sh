for n in 100000000 10000000 6250000 1000000; do
  for v in base rounds; do
    if [ $v = rounds ]; then export PSIM_LIBRARY=$PWD/scratch_libs/libpsim_hip_single_rounds.so; else unset PSIM_LIBRARY; fi
    echo "== $v n=$n"
    timeout -k 10 200 python -u scripts/tune.py --particles $n --nt -1 --lds -1 --steps 200 --rounds 5 | tail -1 || exit 1
    timeout -k 10 200 python -u scripts/tune.py --particles $n --nt -1 --lds -1 --steps 200 --rounds 5 --radius 50 | tail -1 || exit 1
  done
done
