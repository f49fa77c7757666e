#!/bin/bash
# Round-6 measurement pass on one GPU box: gpurun_out/prof_r06/ (summaries are copied into profiles/r06/ by hand).
#   bench line; the same command under rocprofv3 --kernel-trace --stats; PMC passes of the bench (separate runs); configs 3 and 5 under
#   --kernel-trace --stats AND their PMC passes (counters of the final build for every number the documents quote); batch sweeps.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_r06; rm -rf $OUT; mkdir -p $OUT
python bench.py > $OUT/bench_n1_default.json 2> $OUT/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python bench.py --cpu-budget 0 --no-other-configs > $OUT/bench_n1_under_rocprof.json 2> $OUT/stats.log
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/bench_n1_kernel_stats.csv 2>/dev/null
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
pmc_passes bench_n1 python bench.py --cpu-budget 0 --no-other-configs --steps 3 --warmup 1
# configs 3 and 5 of the round-6 build: kernel trace + counters
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt3 -o kt -- python tools/dev/cfg3.py 8192 > $OUT/config3.log 2>&1
cp $(find $OUT/kt3 -name "*kernel_stats.csv" | head -1) $OUT/config3_kernel_stats.csv 2>/dev/null
pmc_passes config3_shoot python tools/dev/cfg3.py 8192
python tools/dev/node_bench.py 128 256 300 512 1024 2048 2>/dev/null | grep config > $OUT/config5_1gpu.jsonl
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt5 -o kt -- python tools/dev/node_bench.py 128 1024 > $OUT/config5.log 2>&1
cp $(find $OUT/kt5 -name "*kernel_stats.csv" | head -1) $OUT/config5_kernel_stats.csv 2>/dev/null
pmc_passes config5_node python tools/dev/node_bench.py 1024
pmc_passes config5_node_b128 python tools/dev/node_bench.py 128
# batch sweep of the headline
for B in 128 256 512 1024 1536 2048 3072 8192; do python bench.py --batch $B --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 >> $OUT/batch_sweep.jsonl; done
python bench.py --scaling strong --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 > $OUT/bench_n1_strong.json
rm -rf $OUT/stats $OUT/kt3 $OUT/kt5
tail -c 300 $OUT/bench_n1_default.json; echo; cat $OUT/config5_1gpu.jsonl; cat $OUT/pmc_config5_node.json | head -c 600
# round 6: the two-level sweep against the plain recursion on the headline draw, the strong split's expectation
timeout 600 python tools/dev/twolevel/agree.py CARTPOLE:100:512 CARTPOLE:25:64 CARTPOLE:5:8 VANDERPOL:40:32 CANCERTREATMENT:20:16 2>&1 | grep waves > $OUT/two_level_agreement.txt
timeout 1200 python tools/projected_scaling.py $OUT/projected_scaling.json > $OUT/projected_scaling.log 2>&1
