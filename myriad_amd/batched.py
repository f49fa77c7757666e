"""Sharding of independent problem instances over the GPUs of one node (one process per GPU) and the path's single
collective: the final gather of solutions over RCCL/xGMI (torch.distributed backend "nccl"; "gloo" in CPU tests).
The reference has no multi-device code at all; instances are independent NLPs, so there is no
data-path collective -- only this gather of z*, cost, status (to every rank, or to one rank: `dst`) (<= 33 MB total at the BASELINE sizes)."""
from __future__ import annotations

from typing import Dict, Sequence, Tuple


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
  """Contiguous chunk [lo, hi) of `total` instances owned by `rank` (sizes differ by at most one)."""
  if not (0 <= rank < world):
    raise ValueError("rank out of range")
  base, rem = divmod(total, world)
  lo = rank * base + min(rank, rem)
  return lo, lo + base + (1 if rank < rem else 0)


def gather_solutions(local: Dict[str, "torch.Tensor"], counts: Sequence[int], group=None, dst=None) -> Dict[str, "torch.Tensor"]:
  """Collect the per-rank result tensors (first dim = local batch, possibly ragged across ranks: `counts[r]` rows from
  rank r), concatenated in rank order.  dst=None: all_gather, every rank gets the result.  dst=r: gather to rank r only
  (1/world of the all_gather bytes: what a driver that post-processes on one rank needs); other ranks get {}."""
  import torch
  import torch.distributed as dist
  world = dist.get_world_size(group)
  rank = dist.get_rank(group)
  assert len(counts) == world and local[next(iter(local))].shape[0] == counts[rank]
  mx = max(counts)
  out = {}
  equal = all(c == mx for c in counts)
  fused = dst is None and equal and dist.get_backend(group) == "nccl"      # RCCL: one flat all-gather straight into the result
  for k, t in local.items():
    if fused:
      res = t.new_empty((world * mx,) + tuple(t.shape[1:]))
      dist.all_gather_into_tensor(res, t.contiguous(), group=group)
      out[k] = res
      continue
    pad = t
    if t.shape[0] < mx:   # (all_)gather needs equal shapes: pad the short ranks
      pad = torch.cat([t, t.new_zeros((mx - t.shape[0],) + tuple(t.shape[1:]))], dim=0)
    pad = pad.contiguous()
    if dst is None:
      bufs = [torch.empty_like(pad) for _ in range(world)]
      dist.all_gather(bufs, pad, group=group)
    else:
      if equal:     # receive straight into the slices of the result (no per-rank buffers, no concatenation pass)
        res = t.new_empty((world * mx,) + tuple(t.shape[1:])) if rank == dst else None
        dist.gather(pad, [res[r * mx:(r + 1) * mx] for r in range(world)] if rank == dst else None, dst=dst, group=group)
        if rank == dst:
          out[k] = res
        continue
      bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
      dist.gather(pad, bufs, dst=dst, group=group)
      if rank != dst:
        continue
    out[k] = torch.cat([b[:counts[r]] for r, b in enumerate(bufs)], dim=0)
  return out
