#!/bin/bash
# exp90: closed-form systems, four wavefronts per trajectory: two chunks x two rungs of the inertia ladder (HsFused::TLS) -- agreement and rates at B <= 256
O=gpurun_out/exp90; mkdir -p $O
AGREE_WAVES=1,2,4 timeout 600 python tools/dev/twolevel/agree.py CARTPOLE:100:256 CARTPOLE:25:64 CARTPOLE:5:8 CARTPOLE:3:4 VANDERPOL:40:32 TIMBERHARVEST:6:8 2>&1 | grep waves | cut -c1-250 | tee $O/agree.txt
for w in 2 4; do for B in 128 256; do MYRIAD_FUSED_WAVES=$w timeout 300 python bench.py --batch $B --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('waves $w B', d['config']['global_batch'], round(d['value']), 'solves/s kernel', d['solver_kernel']['avg_ms'], d['iterations'])"; done; done | tee $O/rates.txt
