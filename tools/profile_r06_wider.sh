#!/bin/bash
# Round 6: counters of the fused kernel's block sweep on the wider systems (separate --pmc passes, as tools/profile_r06.sh): gpurun_out/prof_r06w/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_r06w; rm -rf $OUT; mkdir -p $OUT
pmc_passes() {      # $1 = sub-directory, rest = command
  local sub=$1; shift; local i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$sub/p$i -o p -- "$@" > $OUT/$sub.p$i.log 2>&1 || echo "$sub pass $i ($grp) failed"
  done
  python tools/pmc_summary.py $OUT/$sub $OUT/pmc_$sub.json
  rm -rf $OUT/$sub
}
pmc_passes wider_rocket_hs python tools/dev/wider_one.py ROCKETLANDING HERMITE_SIMPSON 4096 30 2
pmc_passes wider_cartpole_twin_hs python tools/dev/wider_one.py CARTPOLE_ELASTIC HERMITE_SIMPSON 4096 60 2
pmc_passes wider_rocket_twin_hs python tools/dev/wider_one.py ROCKETLANDING_ELASTIC HERMITE_SIMPSON 4096 30 2
bash tools/dev/exp/exp98.sh > /dev/null 2>&1; cp gpurun_out/exp98/times.txt $OUT/wider_systems_times.txt
ls $OUT; head -c 1500 $OUT/pmc_wider_cartpole_twin_hs.json
# kernel statistics (rocprofv3 --kernel-trace --stats) of the three wide systems' Hermite-Simpson solves, one file
for sys in ROCKETLANDING CARTPOLE_ELASTIC ROCKETLANDING_ELASTIC; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$sys -o kt -- python tools/dev/wider_one.py $sys HERMITE_SIMPSON 4096 30 3 > $OUT/kt_$sys.log 2>&1
  f=$(find $OUT/kt_$sys -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && { echo "# $sys HERMITE_SIMPSON N=100 B=4096, 30 iterations, 3 solves"; grep -E "Name|hs_solve" $f; } >> $OUT/wider_systems_kernel_stats.csv
  rm -rf $OUT/kt_$sys
done
