#!/bin/bash
# exp95: the phases of an iteration of the wide systems with the sequential forward recursion in (phase-timing build of the Hermite-Simpson objects)
cd /root/repo; O=gpurun_out/exp95; mkdir -p $O
for sys in ROCKETLANDING CARTPOLE_ELASTIC ROCKETLANDING_ELASTIC; do
  MYRIAD_HIP_LIB=$PWD/xv/libpt.so MYRIAD_PARK_ITER=0 timeout 300 python tools/dev/wider_one.py $sys HERMITE_SIMPSON 4096 30 1 2>&1 | grep -E "^traj [0-3] |solver kernels" | head -5 | tee $O/pt_$sys.txt
done
