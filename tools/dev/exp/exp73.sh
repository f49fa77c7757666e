#!/bin/bash
# exp73: the one-time 5-6 ms host stall inside an asynchronous-copy enqueue (bench.py: "slow enqueue of the download"): where it falls for different
# warm-up counts, and whether priming the copy path before the steps removes it from the run.
for w in 3 6 10; do echo "warmup $w, no priming"; MYRIAD_BENCH_PRIME_MS=0 MYRIAD_BENCH_TRACE=1 python bench.py --warmup $w --cpu-budget 0 --no-other-configs 2>&1 | grep "slow enqueue\|per-step" | cut -c1-200; done
# (a COUNT of priming copies -- 16, 32, 64 rounds of three -- did not remove it: the stall came at step 27, 24, 21 of 40: earlier the longer the priming took)
for p in 0 300 700 1200; do echo "warmup 3, download path exercised for $p ms before the steps"; MYRIAD_BENCH_PRIME_MS=$p MYRIAD_BENCH_TRACE=1 python bench.py --steps 60 --cpu-budget 0 --no-other-configs 2>&1 | grep "slow enqueue\|per-step\|primed" | cut -c1-420; done
