#!/bin/bash
# exp103: the backward pass in parts (phase-timing build: linearisation of the points | elimination and adjoint maps of the stage | adjoint scan + multipliers):
# ROCKETLANDING (six states) and the headline system, B = 4096, whole solves
cd /root/repo; O=gpurun_out/exp103; mkdir -p $O
MYRIAD_HIP_LIB=$PWD/xv/libpt.so MYRIAD_PARK_ITER=0 timeout 300 python tools/dev/wider_one.py ROCKETLANDING HERMITE_SIMPSON 4096 30 1 2>&1 | grep -E "^traj [0-3] |workgroup 0" | head -4 | tee $O/rocket.txt
MYRIAD_HIP_LIB=$PWD/xv/libpt.so MYRIAD_PARK_ITER=0 timeout 300 python tools/dev/one_solve.py 4096 2>&1 | grep -E "^traj [0-3] |workgroup 0" | tail -4 | tee $O/cartpole.txt
