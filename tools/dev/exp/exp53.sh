#!/bin/bash
# exp53 (round 5): activation / tangent streams of the network passes with and without non-temporal accesses, B = 1024 (every CU busy) and B = 128
cd $GRAFT_REPO_ROOT
for lib in libnodetiming.so libnodetiming_plain.so; do
  for B in 128 1024; do
    echo "== $lib B=$B"
    MYRIAD_VARIANT_LIB=variants/$lib python tools/dev/node_phase_timing.py $B 2>&1 | grep -E "traj 1 it|workgroup 0|kernel ms" | awk '!seen[$1,$2,$3]++' | head -n 4
  done
done
