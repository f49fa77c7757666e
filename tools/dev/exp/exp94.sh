#!/bin/bash
# exp94: the sequential forward recursion (HsFused::FSEQ) on the elastic twins (CARTPOLE's: 9 knot variables, ROCKETLANDING's: 14), and whether the headline
# (VOID for the headline-system part: xv/libfseq5.so linked the variant object of the whole system BEHIND the regular part objects, so the regular kernels ran; redone in exp101.sh)
# system (5 knot variables) would gain from it (xv/libfseq5.so: SysCARTPOLE built with -DMYR_FWD_SEQ_NW=5)
cd /root/repo; O=gpurun_out/exp94; mkdir -p $O
for lib in myriad_amd/libmyriad_hip.so xv/libfseq.so; do
  for sys in CARTPOLE_ELASTIC ROCKETLANDING_ELASTIC; do
    for rule in HERMITE_SIMPSON TRAPEZOIDAL; do
      MYRIAD_HIP_LIB=$PWD/$lib timeout 300 python tools/dev/wider_one.py $sys $rule 4096 30 2 2>&1 | grep "solver kernels" | tail -1 | sed "s|^|$lib |"
    done
  done
done | tee $O/times.txt
for lib in myriad_amd/libmyriad_hip.so xv/libfseq5.so; do
  for B in 4096 512; do MYRIAD_HIP_LIB=$PWD/$lib timeout 300 python tools/dev/one_solve.py $B 2>&1 | tail -1 | sed "s|^|$lib B=$B |"; done
done | tee $O/cartpole_nw5.txt
MYRIAD_HIP_LIB=$PWD/xv/libfseq.so timeout 1200 python -m pytest tests/test_gpu_elastic.py tests/test_gpu_systems.py -x -q 2>&1 | tail -5
