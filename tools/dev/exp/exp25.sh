#!/bin/bash
# exp25: automatic variables initialised to ZERO against initialised to the PATTERN (NaN): the same code shape, different garbage -- any difference in the
# results is a read of an uninitialised local.  (exp24 compared pattern against the regular build: the lane kernel bit-identical in 320 of 320 cases, the
# wavefront kernels differ in the last bits in 1 case of 5 -- different code shape, different fused multiply-adds.)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp25
for m in wave wave1 lane; do
  for l in autozero autoinit; do
    MYRIAD_SOLVE_MODE=$m MYRIAD_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib$l.so WPROBE_VERBOSE=1 python tools/dev/wprobe.py all "" 2>/dev/null | grep -v "^compared" > gpurun_out/exp25/${l}_$m.txt
  done
  diff gpurun_out/exp25/autozero_$m.txt gpurun_out/exp25/autoinit_$m.txt > gpurun_out/exp25/diff_$m.txt
  echo "$m: $(wc -l < gpurun_out/exp25/autozero_$m.txt) cases, differing lines: $(grep -c '^<' gpurun_out/exp25/diff_$m.txt)"; head -6 gpurun_out/exp25/diff_$m.txt
done
# the W = 2 form and the network kernel
for l in autozero autoinit; do
  MYRIAD_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib$l.so python -m pytest -q tests/test_gpu_poison.py -k "node or NODE or fresh or headline" 2>&1 | tail -2
done
