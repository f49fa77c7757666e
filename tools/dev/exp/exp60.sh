#!/bin/bash
# exp60 (round 5): per-pass cycles of the final build (headline kernel W = 1 at B = 4096 and W = 2 at B = 256; network kernel at B = 128 / 1024) + a whole CPU baseline solve
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp60
(python bench.py --cpu-full gpurun_out/exp60/cpu_baseline_full.json > gpurun_out/exp60/cpu_full.log 2>&1 &)
for B in 4096 256; do MYRIAD_VARIANT_LIB=variants/libtiming.so python tools/dev/phase_timing.py $B 2>&1 | grep -E "traj [0-3] it|status" | awk '!s[$1,$2]++' > gpurun_out/exp60/phase_cartpole_b$B.txt; done
for B in 1024 128; do MYRIAD_VARIANT_LIB=variants/libtiming.so python tools/dev/node_phase_timing.py $B 2>&1 | grep -E "traj [0-3] it|workgroup 0|converged" | awk '!s[$1,$2,$3]++' | head -n 12 > gpurun_out/exp60/phase_node_b$B.txt; done
cat gpurun_out/exp60/phase_*.txt | cut -c1-200
wait; sleep 1
while pgrep -f "cpu-full" > /dev/null; do sleep 5; done
tail -n 3 gpurun_out/exp60/cpu_full.log | cut -c1-300
