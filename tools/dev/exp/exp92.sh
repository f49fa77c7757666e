#!/bin/bash
# exp92: where an iteration of ROCKETLANDING's fused kernel goes (phase-timing build of the system's Hermite-Simpson object, cycles per phase over a whole
# solve, trajectories 0..3 of B = 4096 and of B = 256 -- four per CU against one per CU: how much of the time is the other three wavefronts' spill traffic)
cd /root/repo; O=gpurun_out/exp92; mkdir -p $O
for B in 4096 256; do
  MYRIAD_HIP_LIB=$PWD/xv/librk_timing.so timeout 300 python tools/dev/wider_one.py ROCKETLANDING HERMITE_SIMPSON $B 30 1 2>&1 | grep -E "^traj|solver kernels" | tee $O/pt_b$B.txt
done
