#!/bin/bash
# exp96: systems whose solver LDS allows two workgroups per CU only (ROCKETLANDING: 60 KB; the twins) leave two SIMDs idle in the one-wavefront form:
# the two-wavefront form at every batch size?
cd /root/repo; O=gpurun_out/exp96; mkdir -p $O
for sys in ROCKETLANDING CARTPOLE_ELASTIC ROCKETLANDING_ELASTIC BEARPOPULATIONS; do
  for rule in HERMITE_SIMPSON TRAPEZOIDAL; do
    for w in 1 2; do
      MYRIAD_FUSED_WAVES=$w MYRIAD_DEBUG_PTRS=1 timeout 300 python tools/dev/wider_one.py $sys $rule 4096 30 2 2>&1 | grep -E "solver kernels|fused W" | tail -2 | cut -c1-140 | sed "s|^|W=$w |"
    done
  done
done | tee $O/times.txt
