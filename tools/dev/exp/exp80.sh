#!/bin/bash
# exp80: two-level sweep against the plain recursion, iteration by iteration (trace build) and after 1 / 2 / 3 / 5 iterations
O=gpurun_out/exp80; mkdir -p $O
for w in 1 2; do
  MYRIAD_HIP_LIB=$PWD/xv/libtrace.so MYRIAD_FUSED_WAVES=$w AGREE_WAVES=$w timeout 300 python tools/dev/twolevel/agree.py CARTPOLE:100:1 2>&1 | grep -E "^(T|S|L) b0" > $O/trace_w$w.txt
done
paste -d'\n' <(grep "^T b0 w0" $O/trace_w1.txt | cut -c1-400) <(grep "^T b0 w0" $O/trace_w2.txt | cut -c1-400) | awk '{print substr($0, 1, 60) " ... " $(NF-3), $(NF-2), $(NF-1), $NF}' | head -60
grep "^L" $O/trace_w2.txt | head -60
for mi in 1 2 3 5 1000; do echo "max_iter $mi"; AGREE_MAX_ITER=$mi timeout 300 python tools/dev/twolevel/agree.py CARTPOLE:100:64 CARTPOLE:10:16 CARTPOLE:2:4 2>&1 | grep -v Warn | tail -12; done 2>&1 | tee $O/agree_iters.txt
