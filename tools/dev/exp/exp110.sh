#!/bin/bash
# exp110: the headline kernel at more than one wavefront per SIMD: bound multipliers in global scratch (-DMYR_ZLU_FORCE=1: 40.8 -> 24.8 KB of LDS, six workgroups
# per CU by LDS) and 256 registers per lane (-DMYR_FUSED_OCC=2), each alone (controls) and together, against the regular build
cd /root/repo; O=gpurun_out/exp110; mkdir -p $O
for lib in myriad_amd/libmyriad_hip.so xv/libzlu.so xv/libocc2.so xv/libocc2zlu.so; do
  for B in 4096 8192 2048; do
    MYRIAD_HIP_LIB=$PWD/$lib MYRIAD_DEBUG_PTRS=1 timeout 300 python tools/dev/one_solve.py $B 2>&1 | grep -E "converged|fused W" | tail -2 | cut -c1-90 | tr '\n' ' ' | sed "s|^|$lib B=$B |"; echo
  done
done | tee $O/times.txt
