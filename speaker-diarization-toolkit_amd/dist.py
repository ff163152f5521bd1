"""Multi-GPU plumbing: one process per GPU, segments sharded by contiguous row
blocks, profiles replicated, ONE exchange step - the all-gather of the [N/G, 192] embeddings that
the global segment-segment affinity / clustering stage needs.  torch.distributed is the transport
(backend "nccl" is RCCL on ROCm; the same code runs on "gloo" for the CPU tests).

xGMI is point-to-point (7 links x ~153 GB/s per GPU): the gather is issued as ONE
all_gather_into_tensor over the whole shard so RCCL can drive all links at once - never a ring of
per-row messages.  Which algorithm RCCL then picks is its own choice; $SDK_ALLGATHER=direct (round 5) issues the
same exchange as world - 1 PAIRWISE send / receive transfers in one batch instead (every pair's direct link, all at
once: floor 0.63 ms for 8 x 96-MB shards against 4.4 ms for a ring) - same result, and bench.py --gpus N times
both forms so the first run on a multi-GPU node says which one to keep."""
from __future__ import annotations

import os
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


def allgather_mode() -> str:
    """"auto" (default: one all_gather_into_tensor, the library's own algorithm) or "direct" (pairwise send / receive), from $SDK_ALLGATHER"""
    m = os.environ.get("SDK_ALLGATHER", "auto")
    if m not in ("auto", "direct"):
        raise ValueError(f"SDK_ALLGATHER={m}: expected auto or direct")
    return m


def _gather_direct(out: torch.Tensor, send: torch.Tensor, group=None) -> None:
    """out[k * rows : (k + 1) * rows] = rank k's `send`, as world - 1 pairwise transfers issued together (batch_isend_irecv: one RCCL group)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    rows = send.shape[0]
    out[rank * rows:(rank + 1) * rows].copy_(send)
    if world == 1:
        return
    ops = []
    for d in range(1, world):
        to, frm = (rank + d) % world, (rank - d) % world
        ops.append(dist.P2POp(dist.isend, send, dist.get_global_rank(group, to) if group is not None else to, group))
        ops.append(dist.P2POp(dist.irecv, out[frm * rows:(frm + 1) * rows], dist.get_global_rank(group, frm) if group is not None else frm, group))
    for w in dist.batch_isend_irecv(ops):
        w.wait()


def _gather_into(out: torch.Tensor, send: torch.Tensor, group=None, mode: Optional[str] = None) -> None:
    """The embedding exchange: all_gather_into_tensor, or the pairwise form (mode / $SDK_ALLGATHER = "direct").  The gloo transport cannot move
    device tensors, so a gloo group with tensors on a GPU (rehearsals of the N > 1 path on a single-GPU box) is staged through host memory."""
    mode = mode or allgather_mode()
    send = send.contiguous()
    staged = send.is_cuda and dist.get_backend(group) == "gloo"
    src = send.cpu() if staged else send
    dst = torch.empty(out.shape, dtype=out.dtype) if staged else out
    if mode == "direct":
        _gather_direct(dst, src, group)
    else:
        dist.all_gather_into_tensor(dst, src, group=group)
    if staged:
        out.copy_(dst)


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
