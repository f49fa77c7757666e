#!/usr/bin/env python3
"""STREAM-style probe of the box's HBM ('peak to be confirmed on the box'): write-only fill, copy and
read-only reduction of buffers of the eval kernel's size, timed with device events; prints one JSON line."""
import json
import torch
dev = torch.device("cuda", 0)
n = 394_000_000 // 8                     # the eval kernel's 394 MB per launch, as doubles
x = torch.empty(n, dtype=torch.float64, device=dev); y = torch.empty_like(x)
def timed(fn, reps=50):
  for _ in range(5): fn()
  torch.cuda.synchronize()
  a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(reps): fn()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / reps * 1e-3
t_fill = timed(lambda: x.fill_(1.0))
t_copy = timed(lambda: y.copy_(x))
t_sum = timed(lambda: x.sum())
B = n * 8
print(json.dumps({"bytes": B, "fill_us": t_fill * 1e6, "fill_GBps": B / t_fill / 1e9, "copy_us": t_copy * 1e6, "copy_GBps": 2 * B / t_copy / 1e9,
                  "sum_us": t_sum * 1e6, "sum_GBps": B / t_sum / 1e9}))
