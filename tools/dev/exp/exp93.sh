#!/bin/bash
# exp93: the recursions of the wide systems' forward and backward phases run sequentially over stored maps (HsFused::FSEQ, BSEQ) against the wave scans: ROCKETLANDING,
# both schemes (xv/libfseq.so: forward only; xv/libfbseq.so: both), then the phases of an iteration again (exp92's build with the change)
cd /root/repo; O=gpurun_out/exp93; mkdir -p $O
for lib in myriad_amd/libmyriad_hip.so xv/libfseq.so xv/libfbseq.so; do
  for rule in HERMITE_SIMPSON TRAPEZOIDAL; do
    for B in 4096 256; do MYRIAD_HIP_LIB=$PWD/$lib timeout 300 python tools/dev/wider_one.py ROCKETLANDING $rule $B 30 2 2>&1 | grep "solver kernels" | tail -1 | sed "s|^|$lib |"; done
  done
done | tee $O/times.txt
MYRIAD_HIP_LIB=$PWD/xv/librk_timing.so timeout 300 python tools/dev/wider_one.py ROCKETLANDING HERMITE_SIMPSON 4096 30 1 2>&1 | grep -E "^traj|solver kernels" | tee $O/pt_b4096.txt
MYRIAD_HIP_LIB=$PWD/xv/libfbseq.so timeout 900 python -m pytest tests/test_gpu_systems.py tests/test_gpu_poison.py -x -q -k "ROCKET or wider" 2>&1 | tail -5
