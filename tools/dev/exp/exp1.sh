#!/bin/bash
# round 4, experiment 1: where do W = 2 and W = 1 part?  (committed build + four diagnostic variants)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp1; O=gpurun_out/exp1
export PYTHONUNBUFFERED=1
FAIL="MOULDFUNGICIDE:HS:6:1,MOULDFUNGICIDE:HS:100:1,CANCERTREATMENT:HS:100:3,TIMBERHARVEST:HS:6:1,MOULDFUNGICIDE:TRAP:6:1,CANCERTREATMENT:HS:50:1"
W12="MYRIAD_FUSED_WAVES=1,MYRIAD_FUSED_WAVES=2,MYRIAD_FUSED_WAVES=1+MYRIAD_POISON=nan,MYRIAD_FUSED_WAVES=1+MYRIAD_POISON=big,MYRIAD_FUSED_WAVES=2+MYRIAD_POISON=nan,MYRIAD_FUSED_WAVES=2+MYRIAD_POISON=big,MYRIAD_SOLVE_MODE=wave1,MYRIAD_SOLVE_MODE=wave1+MYRIAD_POISON=nan"
echo "== A: default build, failing cases, W=1/W=2 x poison, twice" > $O/a.log
for i in 1 2; do WPROBE_VERBOSE=1 timeout 600 python tools/dev/wprobe.py $FAIL $W12 >> $O/a.log 2>&1; done
echo "== B: inline sweep for W=1 too" > $O/b.log
MYRIAD_HIP_LIB=$PWD/variants/lib_inl.so WPROBE_VERBOSE=1 timeout 600 python tools/dev/wprobe.py $FAIL MYRIAD_FUSED_WAVES=1,MYRIAD_FUSED_WAVES=2 >> $O/b.log 2>&1
echo "== C: W=2 calls the sweep" > $O/c.log
MYRIAD_HIP_LIB=$PWD/variants/lib_callw.so WPROBE_VERBOSE=1 timeout 600 python tools/dev/wprobe.py $FAIL MYRIAD_FUSED_WAVES=1,MYRIAD_FUSED_WAVES=2 >> $O/c.log 2>&1
echo "== D: traces" > $O/d.log
for c in MOULDFUNGICIDE:HS:6:1 CANCERTREATMENT:HS:100:1 TIMBERHARVEST:HS:6:1; do
  for w in 1 2; do
    echo "--- $c W=$w (trace build)" >> $O/d.log
    MYRIAD_HIP_LIB=$PWD/variants/lib_trace.so WPROBE_MAX_ITER=60 timeout 300 python tools/dev/wprobe.py $c MYRIAD_FUSED_WAVES=$w >> $O/d.log 2>&1
  done
  echo "--- $c W=1 (trace + inline build)" >> $O/d.log
  MYRIAD_HIP_LIB=$PWD/variants/lib_trace_inl.so WPROBE_MAX_ITER=60 timeout 300 python tools/dev/wprobe.py $c MYRIAD_FUSED_WAVES=1 >> $O/d.log 2>&1
done
echo "== E: whole probe, W=1 vs W=2 vs poisoned W=1 (default build)" > $O/e.log
timeout 2400 python tools/dev/wprobe.py all "MYRIAD_FUSED_WAVES=1,MYRIAD_FUSED_WAVES=2,MYRIAD_FUSED_WAVES=1+MYRIAD_POISON=nan,MYRIAD_SOLVE_MODE=wave1,MYRIAD_SOLVE_MODE=wave1+MYRIAD_POISON=nan" >> $O/e.log 2>&1
tail -3 $O/a.log $O/b.log $O/c.log $O/e.log
