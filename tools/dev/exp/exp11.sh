#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp11; O=gpurun_out/exp11
export PYTHONUNBUFFERED=1
variants/mfma_exec > $O/mfma_exec.txt 2>&1; cat $O/mfma_exec.txt
variants/riccati_scan_probe > $O/riccati_scan_probe.txt 2>&1; cat $O/riccati_scan_probe.txt
timeout 1500 python -m pytest tests -q -m gpu > $O/suite.log 2>&1; tail -3 $O/suite.log
# speculative second rung (wavefront 1 sweeps the next inertia candidate): correct under the new gate?  faster?
for c in "TIMBERHARVEST HS 6 3" "TIMBERHARVEST TRAP 6 3" "MOULDFUNGICIDE HS 100 3" "BIOREACTOR HS 20 3" "CANCERTREATMENT HS 100 3" "CARTPOLE HS 100 8"; do
  MYRIAD_HIP_LIB=$PWD/variants/lib_spec.so timeout 300 python tools/dev/fresh_stats.py $c 8 "MYRIAD_FUSED_WAVES=1,MYRIAD_FUSED_WAVES=2" >> $O/spec.log 2>&1
done
grep -h "distinct\|fault" $O/spec.log
for B in 256 512; do MYRIAD_HIP_LIB=$PWD/variants/lib_spec.so python bench.py --batch $B --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 >> $O/spec_batch.jsonl; python bench.py --batch $B --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 >> $O/default_batch.jsonl; done
python -c "
import json
for f in ('spec_batch','default_batch'):
  for l in open('$O/'+f+'.jsonl'):
    d=json.loads(l); print(f, d['config']['global_batch'], round(d['value']), d['solver_kernel']['avg_ms'])
"
