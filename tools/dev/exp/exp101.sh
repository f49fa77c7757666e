#!/bin/bash
# exp101: the phases of an iteration of the headline kernel (one wavefront per trajectory, B = 4096, whole solves) and of the trapezoidal literal
cd /root/repo; O=gpurun_out/exp101; mkdir -p $O
MYRIAD_HIP_LIB=$PWD/xv/libpt.so MYRIAD_PARK_ITER=0 timeout 300 python tools/dev/one_solve.py 4096 2>&1 | grep -E "^traj [0-3] " | tail -4 | tee $O/pt_w1_b4096.txt
MYRIAD_HIP_LIB=$PWD/xv/libpt.so MYRIAD_PARK_ITER=0 MYRIAD_FUSED_WAVES=1 timeout 300 python tools/dev/one_solve.py 256 2>&1 | grep -E "^traj [0-3] " | tail -4 | tee $O/pt_w1_b256.txt
# ... and the sequential forward recursion at five knot variables, measured properly (exp94's library linked the variant object BEHIND the regular ones: the
# regular kernels ran): the headline workload and its trapezoidal twin through the regular library and through xv/libfseq5.so, three repetitions each
for lib in myriad_amd/libmyriad_hip.so xv/libfseq5.so myriad_amd/libmyriad_hip.so xv/libfseq5.so; do
  for B in 4096 512; do MYRIAD_HIP_LIB=$PWD/$lib timeout 300 python tools/dev/one_solve.py $B 2>&1 | tail -1 | sed "s|^|$lib B=$B |"; done
done | tee $O/cartpole_nw5.txt
