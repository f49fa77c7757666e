#!/bin/bash
# exp86: the interfaces' terminal weight: 4 x |diag P| (default build), 1 x |diag P|, fixed 1e4 -- agreement with the plain recursion, network-kernel time
O=gpurun_out/exp86; mkdir -p $O
for v in default c1 fixed; do
  if [ $v = default ]; then unset MYRIAD_HIP_LIB; else export MYRIAD_HIP_LIB=$PWD/xv/lib$v.so; fi
  echo "== $v" | tee -a $O/agree.txt
  timeout 600 python tools/dev/twolevel/agree.py CARTPOLE:100:128 HIVTREATMENT:50:8 BIOREACTOR:50:8 CANCERTREATMENT:100:8 TUMOUR:20:3 2>&1 | grep -v instance | grep waves | cut -c1-260 | tee -a $O/agree.txt
  timeout 300 python tools/dev/node_bench.py 128 1024 2>&1 | grep config | cut -c1-200 | tee -a $O/node.txt
  timeout 300 python bench.py --batch 512 --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B512', round(d['value']), d['solver_kernel']['avg_ms'])"
done
