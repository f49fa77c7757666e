#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python bench.py --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), 'kernel', round(d['solver_kernel']['avg_ms'],3), 'eval', round(d['roofline']['avg_ms'],4), 'no-download', round(d['download']['value_without_download']), round(d['download']['ms_per_step_without_download'],3))"; }
for k in 0 12 10 0 12 10; do run MYRIAD_PARK_ITER=$k; done
