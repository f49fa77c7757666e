#!/bin/bash
# Round-4 measurement pass on one GPU box: gpurun_out/prof_r04/ (summaries are copied into profiles/r04/ by hand).
#   bench line, the same command under rocprofv3 --kernel-trace --stats, PMC passes (separate runs), batch sweep, configs 3 / 5 under
#   --kernel-trace --stats, the N = 1 weak / strong lines and the 2-rank rehearsal of the N > 1 path on one device.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_r04; rm -rf $OUT; mkdir -p $OUT
python bench.py > $OUT/bench_n1_default.json 2> $OUT/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python bench.py --cpu-budget 0 --no-other-configs > $OUT/bench_n1_under_rocprof.json 2> $OUT/stats.log
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/bench_n1_kernel_stats.csv 2>/dev/null
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p -- python bench.py --cpu-budget 0 --no-other-configs --steps 3 --warmup 1 > $OUT/p$i.log 2>&1 || echo "pass $i ($grp) failed"
done
python tools/pmc_summary.py $OUT $OUT/pmc_bench_n1.json
# batch sweep (W = 2 up to two trajectories per CU)
for B in 256 512 1024 1536 2048 3072 8192; do python bench.py --batch $B --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 >> $OUT/batch_sweep.jsonl; done
for B in 256 512; do MYRIAD_FUSED_WAVES=1 python bench.py --batch $B --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 >> $OUT/batch_sweep_w1.jsonl; done
# configs 3 and 5: kernel trace of the round-4 build
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt3 -o kt -- python tools/dev/cfg3.py 8192 > $OUT/config3.log 2>&1
cp $(find $OUT/kt3 -name "*kernel_stats.csv" | head -1) $OUT/config3_kernel_stats.csv 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt5 -o kt -- python tools/dev/node_bench.py 128 1024 > $OUT/config5.log 2>&1
cp $(find $OUT/kt5 -name "*kernel_stats.csv" | head -1) $OUT/config5_kernel_stats.csv 2>/dev/null
# N = 1 lines of both scalings, and the N > 1 path rehearsed with two ranks on ONE device (RCCL group of two on device 0, if RCCL allows it)
python bench.py --scaling strong --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 > $OUT/bench_n1_strong.json
timeout 600 python bench.py --gpus 2 --scaling strong --cpu-budget 0 --no-other-configs > $OUT/rehearsal_2ranks_strong.json 2> $OUT/rehearsal_2ranks_strong.err; echo "rc $?" >> $OUT/rehearsal_2ranks_strong.err
timeout 600 python bench.py --gpus 2 --scaling weak --batch 2048 --cpu-budget 0 --no-other-configs > $OUT/rehearsal_2ranks_weak.json 2> $OUT/rehearsal_2ranks_weak.err; echo "rc $?" >> $OUT/rehearsal_2ranks_weak.err
rm -rf $OUT/stats $OUT/p[0-9] $OUT/kt3 $OUT/kt5
tail -c 400 $OUT/bench_n1_default.json; echo; cat $OUT/batch_sweep.jsonl | python -c "
import sys, json
for l in sys.stdin:
  d = json.loads(l); print(d['config']['global_batch'], round(d['value']), d['ms_per_step'], d['solver_kernel']['avg_ms'])"
tail -3 $OUT/rehearsal_2ranks_strong.err
