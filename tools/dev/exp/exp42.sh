#!/bin/bash
# exp42: the two-phase launch under the network kernel (config 5)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_poison.py -q -x -k "network" 2>&1 | grep -E "passed|failed|Error|assert" | head
for k in 0 -1 8 12 16; do echo "== MYRIAD_PARK_ITER=$k"; MYRIAD_PARK_ITER=$k python tools/dev/node_bench.py 128 512 1024 2048 2>/dev/null | tail -4; done
