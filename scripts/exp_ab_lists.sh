# A/B of library builds on the attractor-list kernels: 100 M particles (8 and 64 attractors) and
# 6.25 M particles (8 attractors, one GPU's share of the 50 M configuration).
for v in "$@"; do
  export PSIM_LIBRARY=$PWD/scratch_libs/libpsim_$v.so
  for rep in 1 2; do
    for op in list8 list64; do echo "$v 100M"; timeout -k 10 100 python scripts/profile_valu.py --op $op || exit 1; done
    echo "$v 6.25M"; timeout -k 10 100 python scripts/profile_valu.py --op list8 --particles 6250000 --launches 300 || exit 1
  done
done
