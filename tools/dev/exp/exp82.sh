#!/bin/bash
# exp82: where an iteration of the two-wavefront form goes (phase-timing build, cycles of wavefront 0): B = 512
O=gpurun_out/exp82; mkdir -p $O
for w in 1 2; do MYRIAD_HIP_LIB=$PWD/xv/libpt.so MYRIAD_FUSED_WAVES=$w timeout 300 python tools/dev/one_solve.py 512 2>&1 | grep -E "^traj|converged" | tail -6 | tee $O/pt_w$w.txt; done
