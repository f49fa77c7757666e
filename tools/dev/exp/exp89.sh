#!/bin/bash
# exp89: counters of the two-wavefront kernel (two-level sweep) at B = 512 against the one-wavefront kernel at the same batch size
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/exp89; rm -rf $OUT; mkdir -p $OUT
pmc_passes() {
  local sub=$1; shift; local i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$sub/p$i -o p -- "$@" > $OUT/$sub.p$i.log 2>&1 || echo "$sub pass $i ($grp) failed"
  done
  python tools/pmc_summary.py $OUT/$sub $OUT/pmc_$sub.json
  rm -rf $OUT/$sub
}
pmc_passes w2_b512 python tools/dev/one_solve.py 512
MYRIAD_FUSED_WAVES=1 pmc_passes w1_b512 python tools/dev/one_solve.py 512
python - <<'PY'
import json
for f in ("w2_b512","w1_b512"):
  d=json.load(open(f"gpurun_out/exp89/pmc_{f}.json"))
  for k,v in d.items():
    if "solve" in k:
      a=v["per_dispatch_avg"]; wc=a["SQ_WAVE_CYCLES"]
      print(f, k, "wave cycles %.3g active %.1f%% wait %.1f%% wait_inst %.1f%% busy %.3g | valu %.3g salu %.3g lds %.3g vmem rd %.3g wr %.3g mfma %.3g | fetch x2 %.2f GB write %.2f GB" % (wc, 100*a["SQ_ACTIVE_INST_ANY"]/wc, 100*a["SQ_WAIT_ANY"]/wc, 100*a["SQ_WAIT_INST_ANY"]/wc, a["SQ_BUSY_CYCLES"], a["SQ_INSTS_VALU"], a["SQ_INSTS_SALU"], a["SQ_INSTS_LDS"], a["SQ_INSTS_VMEM_RD"], a["SQ_INSTS_VMEM_WR"], a["SQ_INSTS_MFMA"], v["derived_traffic_bytes"]["fetch_x2"]/1e9, v["derived_traffic_bytes"]["write"]/1e9), v["launch"])
PY
