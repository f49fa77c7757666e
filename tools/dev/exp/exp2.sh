#!/bin/bash
# round 4, experiment 2: BIOREACTOR, the two cases where W = 2 and W = 1 part on the round-4 build; GPU suite + bench of the build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp2; O=gpurun_out/exp2
export PYTHONUNBUFFERED=1
C="BIOREACTOR:HS:100:3,BIOREACTOR:HS:50:3,BIOREACTOR:HS:100:8,BIOREACTOR:TRAP:100:3"
W="MYRIAD_FUSED_WAVES=1,MYRIAD_FUSED_WAVES=2,MYRIAD_FUSED_WAVES=2+MYRIAD_POISON=nan,MYRIAD_FUSED_WAVES=2+MYRIAD_POISON=big,MYRIAD_FUSED_WAVES=1+MYRIAD_POISON=nan,MYRIAD_SOLVE_MODE=wave1,MYRIAD_SOLVE_MODE=lane"
for i in 1 2; do WPROBE_VERBOSE=1 timeout 600 python tools/dev/wprobe.py $C $W >> $O/a.log 2>&1; done
MYRIAD_HIP_LIB=$PWD/variants/lib_inl.so WPROBE_VERBOSE=1 timeout 600 python tools/dev/wprobe.py $C MYRIAD_FUSED_WAVES=1,MYRIAD_FUSED_WAVES=2 >> $O/b_inline.log 2>&1
MYRIAD_HIP_LIB=$PWD/variants/lib_callw.so WPROBE_VERBOSE=1 timeout 600 python tools/dev/wprobe.py $C MYRIAD_FUSED_WAVES=1,MYRIAD_FUSED_WAVES=2 >> $O/c_callw.log 2>&1
for w in 1 2; do
  MYRIAD_HIP_LIB=$PWD/variants/lib_trace.so WPROBE_VERBOSE=1 WPROBE_MAX_ITER=40 timeout 300 python tools/dev/wprobe.py BIOREACTOR:HS:100:3 MYRIAD_FUSED_WAVES=$w > $O/trace_w$w.log 2>&1
done
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
