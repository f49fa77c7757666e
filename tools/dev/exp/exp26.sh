#!/bin/bash
# exp26: which automatic variable is read before it is written?  -ftrivial-auto-var-init-stop-after=N, zero against pattern, SysTUMOUR translation unit
cd $GRAFT_REPO_ROOT
for N in $BIS_NS; do
  for t in z p; do
    h=$(MYRIAD_HIP_LIB=$GRAFT_REPO_ROOT/variants/libt_${t}_$N.so WPROBE_VERBOSE=1 python tools/dev/wprobe.py ${BIS_CASE:-TUMOUR:HS:6:1} "" 2>/dev/null | grep -o "z#[0-9a-f]*" | tr '\n' ' ')
    echo "N=$N $t $h"
  done
done
