"""Multi-GPU plumbing: one process per GPU, segments sharded by contiguous row
blocks, profiles replicated, ONE exchange step - the all-gather of the [N/G, 192] embeddings that
the global segment-segment affinity / clustering stage needs.  torch.distributed is the transport
(backend "nccl" is RCCL on ROCm; the same code runs on "gloo" for the CPU tests).

xGMI is point-to-point (7 links x ~153 GB/s per GPU): the gather is issued as ONE
all_gather_into_tensor over the whole shard so RCCL can drive all links at once - never a ring of
per-row messages."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous [lo, hi) row blocks, sizes differing by at most one (first n % world get +1)."""
    q, r = divmod(n, world)
    out, lo = [], 0
    for k in range(world):
        hi = lo + q + (1 if k < r else 0)
        out.append((lo, hi))
        lo = hi
    return out


def shard_range(n: int, rank: Optional[int] = None, world: Optional[int] = None) -> Tuple[int, int]:
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    return shard_bounds(n, world)[rank]


def _gather_into(out: torch.Tensor, send: torch.Tensor, group=None) -> None:
    """all_gather_into_tensor; the gloo transport cannot move device tensors for this collective, so a gloo group with
    tensors on a GPU (rehearsals of the N > 1 path on a single-GPU box) is staged through host memory."""
    if send.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, send.cpu().contiguous(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, send.contiguous(), group=group)


def all_gather_rows(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Gather row shards produced under shard_bounds(n_total, world) into the full [n_total, d]
    tensor on every rank.  Shards are padded to the largest block so a single collective moves
    everything; padding rows are dropped after the gather."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    bounds = shard_bounds(n_total, world)
    lo, hi = bounds[rank]
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank}: shard has {local.shape[0]} rows, expected {hi - lo}")
    width = max(h - l for l, h in bounds)
    if width == 0:
        return local.new_zeros((0,) + tuple(local.shape[1:]))
    send = local if local.shape[0] == width else torch.cat([local, local.new_zeros((width - local.shape[0],) + tuple(local.shape[1:]))])
    out = local.new_empty((world * width,) + tuple(local.shape[1:]))
    _gather_into(out, send, group)
    if all(h - l == width for l, h in bounds):
        return out
    return torch.cat([out[k * width: k * width + (h - l)] for k, (l, h) in enumerate(bounds)])


def all_reduce_sum(x: torch.Tensor, group=None) -> torch.Tensor:
    dist.all_reduce(x, op=dist.ReduceOp.SUM, group=group)
    return x
