#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do MYRIAD_PARK_ITER=12 MYRIAD_BENCH_TRACE=1 python bench.py --cpu-budget 0 --no-other-configs 2>&1 >/dev/null | grep "step" | cut -c1-220; done
