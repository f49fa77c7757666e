#!/bin/bash
# Read-before-write check of the AUTOMATIC variables of every kernel (the part MYRIAD_POISON cannot reach: it overwrites LDS and the global scratch).
# clang's -ftrivial-auto-var-init gives every local without an initialiser a value: `pattern` = all-ones words (a NaN for every double, 0xAAAAAAAA for
# integers), `zero` = zeros.  A kernel that reads a local before writing it returns other bits under the two builds (or NaNs under `pattern`).
#   step 1 (CPU, ~13 min):  tools/dev/autoinit_check.sh build        -> variants/libautoinit.so, variants/libautozero.so   (all 26 system objects)
#   step 2 (GPU, ~8 min):   tools/dev/autoinit_check.sh probe        -> gpurun_out/autoinit/
# Findings of round 4: profiles/r04/autoinit_check.txt.  (-ftrivial-auto-var-init-stop-after=N bisects a difference down to one variable: exp26.sh.)
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd $ROOT
if [ "$1" = build ]; then
  ALL=$(python -c "import __graft_entry__ as g; print(','.join(g._systems()))")
  MYR_VARIANT_SYS=$ALL tools/dev/build_variant.sh "" libautoinit.so -ftrivial-auto-var-init=pattern $EXTRA
  MYR_VARIANT_SYS=$ALL tools/dev/build_variant.sh "" libautozero.so -ftrivial-auto-var-init=zero $EXTRA
elif [ "$1" = probe ]; then
  mkdir -p gpurun_out/autoinit
  MYRIAD_HIP_LIB=$ROOT/variants/libautoinit.so python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee gpurun_out/autoinit/suite_pattern.txt
  for m in wave wave1 lane; do
    for l in autozero autoinit; do
      MYRIAD_SOLVE_MODE=$m MYRIAD_HIP_LIB=$ROOT/variants/lib$l.so WPROBE_VERBOSE=1 python tools/dev/wprobe.py all "" 2>/dev/null | grep -v "^compared" > gpurun_out/autoinit/${l}_$m.txt
    done
    diff gpurun_out/autoinit/autozero_$m.txt gpurun_out/autoinit/autoinit_$m.txt > gpurun_out/autoinit/diff_$m.txt
    echo "MYRIAD_SOLVE_MODE=$m: $(wc -l < gpurun_out/autoinit/autozero_$m.txt) cases, zero- and pattern-initialised builds differ in $(grep -c '^<' gpurun_out/autoinit/diff_$m.txt)"
  done | tee gpurun_out/autoinit/summary.txt
else
  echo "usage: $0 build|probe   (EXTRA=-ffp-contract=off for the contraction-free pair)"
fi
