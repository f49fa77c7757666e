#!/bin/bash
# exp52 (round 5): error text of the failing network-kernel cases
cd $GRAFT_REPO_ROOT
MYRIAD_DEBUG_PTRS=1 python -m pytest tests/test_gpu_poison.py -x -q -m gpu -s -k "test_network_dynamics_four_wavefront_kernel and 100-300" > gpurun_out/exp52_c.txt 2>&1
grep -E "myriad\]|Error|rocm-smi|passed|failed" gpurun_out/exp52_c.txt | cut -c1-260 | tail -n 20
rocm-smi --showmeminfo vram 2>&1 | tail -n 5
