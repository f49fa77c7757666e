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
  stage = dist.get_backend(group) == "gloo"       # gloo moves host memory: device tensors are staged through it (results stay on the host)
  for k, t in local.items():
    if stage and t.is_cuda:
      t = t.cpu()
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


def fan_out_solve(engines: Sequence, z0, lb, ub, params=None, opts=None) -> Dict:
  """One batch over several handles (one per device) beneath an unchanged caller: contiguous `shard_range` chunks, one host
  thread per engine (the C-ABI call releases the GIL and every handle binds its own device and stream), results concatenated
  in instance order.  The reference's API solves one instance per call on one device (trajectory_optimizers/base.py:69-93);
  this is the internal fan-out SURVEY.md 8(b) "Threading" asks for.  `engines[i].solve(z0, lb, ub, params=, opts=)` must
  return a dict of arrays with the batch in the first dimension.  An engine may appear twice (two shards queue on one
  device)."""
  import numpy as np
  z0 = np.asarray(z0, dtype=np.float64)
  if z0.ndim == 1:
    z0 = z0[None]
  B = z0.shape[0]
  lb = np.broadcast_to(np.asarray(lb, dtype=np.float64), z0.shape)
  ub = np.broadcast_to(np.asarray(ub, dtype=np.float64), z0.shape)
  p = None if params is None else np.asarray(params, dtype=np.float64)
  def call(e, lo, hi):
    pr = p if (p is None or p.ndim == 1) else p[lo:hi]
    return e.solve(z0[lo:hi], lb[lo:hi], ub[lo:hi], params=pr, opts=opts)

  return fan_out(engines, B, call)


def fan_out_solve_x0(engines: Sequence, x0s, g0, g1, lb, ub, params=None, opts=None) -> Dict:
  """fan_out_solve for instances that differ in their start state only (`engine.solve_x0`: guess and bounds are expanded on the
  device from the templates g0, g1, lb, ub [n]); the shards carry x0s [B,ns] and, if given per instance, params."""
  import numpy as np
  x0s = np.asarray(x0s, dtype=np.float64)
  p = None if params is None else np.asarray(params, dtype=np.float64)

  def call(e, lo, hi):
    pr = p if (p is None or p.ndim == 1) else p[lo:hi]
    return e.solve_x0(x0s[lo:hi], g0, g1, lb, ub, params=pr, opts=opts)

  return fan_out(engines, x0s.shape[0], call)


def fan_out(engines: Sequence, B: int, call) -> Dict:
  """`call(engine, lo, hi)` on contiguous shards of B instances, one host thread per engine; dict results concatenated."""
  import threading
  import numpy as np
  world = max(1, min(len(engines), B))
  if world == 1:
    return call(engines[0], 0, B)
  out = [None] * world
  err = [None] * world
  locks = {}
  for e in engines[:world]:
    locks.setdefault(id(e), threading.Lock())        # a handle owns one scratch buffer and one stream: one call at a time

  def work(r):
    lo, hi = shard_range(B, r, world)
    try:
      with locks[id(engines[r])]:
        out[r] = call(engines[r], lo, hi)
    except BaseException as e:   # re-raised in the caller's thread
      err[r] = e

  ts = [threading.Thread(target=work, args=(r,), name=f"myriad-dev{r}") for r in range(world)]
  for t in ts:
    t.start()
  for t in ts:
    t.join()
  for e in err:
    if e is not None:
      raise e
  return {k: np.concatenate([o[k] for o in out], axis=0) for k in out[0]}
