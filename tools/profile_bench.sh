#!/bin/bash
# Profile `python bench.py` (default workload) on the GPU box: kernel-trace stats, then PMC passes (each its own run,
# --kernel-trace only beside --pmc).  Output: gpurun_out/prof_<tag>/ ; summaries to copy into profiles/<round>/.
#   usage: tools/profile_bench.sh <tag>
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
python bench.py > $OUT/bench_n1_default.json 2> $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python bench.py --cpu-budget 0 --no-other-configs > $OUT/bench_n1_under_rocprof.json 2> $OUT/stats.log
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/bench_n1_kernel_stats.csv 2>/dev/null
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p -- python bench.py --cpu-budget 0 --no-other-configs --steps 3 --warmup 1 > $OUT/p$i.log 2>&1 || echo "pass $i ($grp) failed"
done
python tools/pmc_summary.py $OUT $OUT/pmc_bench_n1.json
