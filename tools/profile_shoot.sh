#!/bin/bash
# config 3 (VANDERPOL single shooting 1 x 50, B = 8192 per GPU): kernel trace + PMC passes of the shooting wavefront kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_shoot; rm -rf $OUT; mkdir -p $OUT
python tools/dev/cfg3.py 8192 2>/dev/null | tail -1 > $OUT/config3_1gpu.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python tools/dev/cfg3.py 8192 > $OUT/kt.log 2>&1 || echo "kernel trace failed"
i=0
for grp in "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p -- python tools/dev/cfg3.py 8192 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python tools/pmc_summary.py $OUT $OUT/pmc_shoot.json
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/config3_kernel_stats.csv
