#!/bin/bash
# exp97: the wide systems' Hermite-Simpson forms with the bound multipliers in global scratch (-DMYR_ZLU_GLOBAL_WIDE=1: four one-wavefront workgroups per CU
# instead of two) against the two-wavefront form at two workgroups per CU (xv/libbase.so: the dispatch rule of exp96)
cd /root/repo; O=gpurun_out/exp97; mkdir -p $O
for lib in xv/libbase.so xv/libzg.so; do
  for sys in ROCKETLANDING CARTPOLE_ELASTIC ROCKETLANDING_ELASTIC; do
    for w in 0 1 2; do
      MYRIAD_HIP_LIB=$PWD/$lib MYRIAD_FUSED_WAVES=$w MYRIAD_DEBUG_PTRS=1 timeout 300 python tools/dev/wider_one.py $sys HERMITE_SIMPSON 4096 30 2 2>&1 | grep -E "solver kernels|fused W" | tail -2 | cut -c1-100 | tr '\n' ' ' | sed "s|^|$lib W=$w |"; echo
    done
  done
done | tee $O/times.txt
