#!/bin/bash
# exp104: where a tile of the network passes goes (phase-timing build: cycles of workgroup 0, wavefront 0 per MODE and segment), B = 1024 (two wavefronts per
# trajectory, no helpers) and B = 128 (four wavefronts + a helper workgroup)
cd /root/repo; O=gpurun_out/exp104; mkdir -p $O
for B in 1024 128; do MYRIAD_VARIANT_LIB=xv/libnodept.so timeout 300 python tools/dev/node_phase_timing.py $B 2>&1 | grep -E "workgroup 0|MODE . segments|converged" | head -9 | tee $O/seg_b$B.txt; done
