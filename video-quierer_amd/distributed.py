"""Multi-GPU plumbing for the path's two exchange steps.  One
process per GPU; ``torch.distributed`` is the transport (backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests) — tensors in, tensors out,
no compute here.

* ingest: frames shard by contiguous ranges (keeps ``f"{video_id}_{i}"`` order,
  reference src/video_search_system.py:164-166); ONE all-gather of the
  per-rank ``[n_rank, D]`` embeddings puts all rows, in frame order, on every rank.
* search over a row-sharded matrix: every rank scans its shard and emits its
  local top-k with GLOBAL row ids; ONE all-gather of ``[Q, k]`` (distance, id)
  pairs, then a k-way merge in the reference's ``(distance, id)`` order
  (reference src/indexes/hnsw.py:269).  The score matrix is never exchanged.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of n items for this rank (first n % world ranks get one extra)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_rows(local: torch.Tensor, counts=None) -> torch.Tensor:
    """[n_rank, D] per rank → [sum n_rank, D] on every rank, in rank (= frame) order.
    Equal shard sizes use one all_gather_into_tensor; ragged shards pad to the largest."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    if counts is None:
        c = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        allc = torch.empty(world, dtype=torch.int64, device=local.device)
        dist.all_gather_into_tensor(allc, c)
        counts = [int(v) for v in allc.tolist()]
    if len(set(counts)) == 1:
        out = torch.empty((world * counts[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    buf = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, pad)
    return torch.cat([buf[r * mx: r * mx + counts[r]] for r in range(world)], dim=0)


def merge_topk(all_ids: torch.Tensor, all_dist: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """[W, Q, k] candidate lists (global ids, -1 = empty slot with distance +inf) → exact [Q, k]
    in (distance asc, id asc) order."""
    w, q, kk = all_ids.shape
    d = all_dist.permute(1, 0, 2).reshape(q, w * kk)
    i = all_ids.permute(1, 0, 2).reshape(q, w * kk).to(torch.int64)
    # lexicographic (distance, id): sort by id first, then a stable sort by distance
    o1 = torch.argsort(torch.where(i < 0, torch.iinfo(torch.int64).max, i), dim=1, stable=True)
    d1, i1 = torch.gather(d, 1, o1), torch.gather(i, 1, o1)
    o2 = torch.argsort(d1, dim=1, stable=True)[:, :k]
    return torch.gather(i1, 1, o2).to(torch.int32), torch.gather(d1, 1, o2)


def sharded_topk(local_ids: torch.Tensor, local_dist: torch.Tensor, row_offset: int, k: int):
    """Local [Q, k] (shard-local row ids) → global exact [Q, k] on every rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return (local_ids if row_offset == 0 else torch.where(local_ids >= 0, local_ids + int(row_offset), local_ids)), local_dist
    gids = torch.where(local_ids >= 0, local_ids + int(row_offset), local_ids)
    world = dist.get_world_size()
    q, kk = gids.shape
    all_ids = torch.empty((world * q, kk), dtype=gids.dtype, device=gids.device)     # concatenated along dim 0
    all_d = torch.empty((world * q, kk), dtype=local_dist.dtype, device=local_dist.device)
    dist.all_gather_into_tensor(all_ids, gids.contiguous())
    dist.all_gather_into_tensor(all_d, local_dist.contiguous())
    return merge_topk(all_ids.view(world, q, kk), all_d.view(world, q, kk), k)
