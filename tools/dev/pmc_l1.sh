#!/bin/bash
# dev tool: L1 / address-unit counters of the solver kernel (separate passes; kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_l1; rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WAVEFRONTS_sum" \
           "TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TD_TD_BUSY_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p -- python tools/dev/one_solve.py 4096 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python tools/pmc_summary.py $OUT $OUT/summary.json
