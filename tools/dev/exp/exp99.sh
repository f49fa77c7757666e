#!/bin/bash
# exp99: the phases of an iteration of the network kernel as of the end of round 6 (phase-timing build), B = 128 (four wavefronts + a helper workgroup) and B = 1024
cd /root/repo; O=gpurun_out/exp99; mkdir -p $O
for B in 128 1024; do MYRIAD_VARIANT_LIB=xv/libnodept.so timeout 300 python tools/dev/node_phase_timing.py $B 2>&1 | grep -E "^traj [0-3] wave 0|workgroup 0|converged" | head -12 | tee $O/pt_b$B.txt; done
