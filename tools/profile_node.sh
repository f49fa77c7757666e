#!/bin/bash
# PMC passes for config 5 (NODE): MFMA instruction counts and busy cycles of the solve kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_node; rm -rf $OUT; mkdir -p $OUT
python tools/dev/node_bench.py 1024 2>/dev/null > $OUT/config5_1gpu.json
i=0
for grp in "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p -- python tools/dev/node_bench.py 1024 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python tools/pmc_summary.py $OUT $OUT/pmc_node.json
