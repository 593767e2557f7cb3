set -u
REPO=$GRAFT_REPO_ROOT
mkdir -p $REPO/gpurun_out/prof; cd /tmp && export TMPDIR=/tmp
for v in rounds slots; do
  if [ $v = slots ]; then export PSIM_LIBRARY=$REPO/scratch_libs/libpsim_hip_slot.so; else unset PSIM_LIBRARY; fi
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $REPO/gpurun_out/prof/r02_ab_$v -o pmc -- python3 $REPO/scripts/profile_valu.py --op batch16 > /dev/null 2> $REPO/gpurun_out/prof/r02_ab_$v.err || exit 1
done
