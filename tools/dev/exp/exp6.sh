#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp6; O=gpurun_out/exp6
L=variants/wg_global_order
for i in 1 2 3; do
  for cfg in "64 256 256" "256 512 256" "16 64 1024" "512 2048 64" "2 8 4096" "64 256 1"; do
    $L $cfg 0 >> $O/litmus.log 2>&1
    $L $cfg 1 >> $O/litmus.log 2>&1
  done
done
cat $O/litmus.log
