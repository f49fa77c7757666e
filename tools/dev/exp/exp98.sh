#!/bin/bash
# exp98: the wide systems after the round's three changes (sequential forward recursion, two wavefronts per trajectory where the LDS allows two workgroups per
# CU only, bound multipliers in global scratch where that doubles the workgroups per CU): both schemes, B = 4096 and 256, the library's own choice of form
cd /root/repo; O=gpurun_out/exp98; mkdir -p $O
for sys in ROCKETLANDING CARTPOLE_ELASTIC ROCKETLANDING_ELASTIC BEARPOPULATIONS; do
  for rule in HERMITE_SIMPSON TRAPEZOIDAL; do
    for B in 4096 256; do
      MYRIAD_DEBUG_PTRS=1 timeout 300 python tools/dev/wider_one.py $sys $rule $B 30 2 2>&1 | grep -E "solver kernels|fused W" | tail -2 | cut -c10-75 | tr '\n' ' '; echo
    done
  done
done | tee $O/times.txt
