#!/bin/bash
# exp91: the network kernel's two forms (two wavefronts per trajectory, two trajectories per CU / four wavefronts, one per CU) at batch sizes between one and
# eight trajectories per CU -- is the switch at B = #CU still in the right place now that the sweep is two-level?
cd /root/repo
for w in 0 4 2; do
  echo "== MYRIAD_FUSED_WAVES=$w"
  MYRIAD_FUSED_WAVES=$w timeout 300 python tools/dev/node_bench.py 256 300 384 512 768 1024 1536 2048 2>/dev/null | grep config | python -c "
import sys, json
for l in sys.stdin:
  d = json.loads(l); print(d['B'], round(d['kernel_ms'], 2), 'ms', round(d['solves_per_s_kernel']), 'solves/s', d['status'])"
done
