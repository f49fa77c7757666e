#!/bin/bash
# exp24: every automatic variable of every kernel initialised with clang's pattern (-ftrivial-auto-var-init=pattern: all-ones words, a NaN for every double): the same
# bits must come out as from the regular build (a difference = a read of an uninitialised local)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp24
WPROBE_VERBOSE=1 python tools/dev/wprobe.py all "" 2>/dev/null | grep -v "^compared" > gpurun_out/exp24/regular.txt
MYRIAD_HIP_LIB=$GRAFT_REPO_ROOT/variants/libautoinit.so WPROBE_VERBOSE=1 python tools/dev/wprobe.py all "" 2>/dev/null | grep -v "^compared" > gpurun_out/exp24/autoinit.txt
wc -l gpurun_out/exp24/regular.txt gpurun_out/exp24/autoinit.txt
diff gpurun_out/exp24/regular.txt gpurun_out/exp24/autoinit.txt > gpurun_out/exp24/diff.txt; echo "differing lines: $(grep -c '^<' gpurun_out/exp24/diff.txt)"; head -20 gpurun_out/exp24/diff.txt
for m in wave1 lane; do
  MYRIAD_SOLVE_MODE=$m WPROBE_VERBOSE=1 python tools/dev/wprobe.py all "" 2>/dev/null | grep -v "^compared" > gpurun_out/exp24/regular_$m.txt
  MYRIAD_SOLVE_MODE=$m MYRIAD_HIP_LIB=$GRAFT_REPO_ROOT/variants/libautoinit.so WPROBE_VERBOSE=1 python tools/dev/wprobe.py all "" 2>/dev/null | grep -v "^compared" > gpurun_out/exp24/autoinit_$m.txt
  diff gpurun_out/exp24/regular_$m.txt gpurun_out/exp24/autoinit_$m.txt > gpurun_out/exp24/diff_$m.txt; echo "$m: $(wc -l < gpurun_out/exp24/regular_$m.txt) cases, differing lines: $(grep -c '^<' gpurun_out/exp24/diff_$m.txt)"; head -10 gpurun_out/exp24/diff_$m.txt
done
