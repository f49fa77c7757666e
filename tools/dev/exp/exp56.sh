#!/bin/bash
# exp56 (round 5): state of every workgroup of a launch that does not end (pinned-memory trace words, dumped by a host thread after 8 s)
cd $GRAFT_REPO_ROOT
MYRIAD_VARIANT_LIB=variants/libnodetrace2.so MYRIAD_NODE_HELPERS=1 timeout 25 python tools/dev/node_phase_timing.py 8 > gpurun_out/exp56_trace.txt 2>&1; echo "rc=$?"; cat gpurun_out/exp56_trace.txt | cut -c1-160 | head -n 40
