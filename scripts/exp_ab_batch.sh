for rep in 1 2; do
for v in base bfi; do
  export PSIM_LIBRARY=$PWD/scratch_libs/libpsim_$v.so
  for op in batch16 batch64; do echo "$v"; timeout -k 10 100 python scripts/profile_valu.py --op $op || exit 1; done
done
done
