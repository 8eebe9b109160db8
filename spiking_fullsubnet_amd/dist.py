"""Multi-GPU execution of the hot path: independent clips shard across ranks (one process per GPU), the scan
needs no data-path collective, and the only exchange is the all-gather of the enhanced magnitudes -- the analogue
of the reference's ``accelerator.gather_for_metrics(step_output)`` (audiozen/trainer.py:511,555).

Backend-agnostic on purpose: ``nccl`` (= RCCL over xGMI) on the GPU node, ``gloo`` in the CPU tests.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_clips: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced partition of ``n_clips`` over ``world`` ranks: rank r owns [lo, hi).
    The first ``n_clips % world`` ranks own one clip more (same convention as torch.tensor_split)."""
    if not (0 <= rank < world) or n_clips < 0:
        raise ValueError(f"bad shard request: n_clips={n_clips}, rank={rank}, world={world}")
    base, extra = divmod(n_clips, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_clips(local: torch.Tensor, n_clips: int, group=None) -> torch.Tensor:
    """All-gather per-rank results ``local[b_local, ...]`` into clip order ``[n_clips, ...]`` on every rank.

    Equal shards use one ``all_gather_into_tensor`` (a single RCCL collective over the fully connected xGMI
    fabric); ragged shards pad to the largest shard first (the reference's ``gather_for_metrics`` likewise
    truncates the padded tail).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(n_clips, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {local.shape[0]} clips, its shard is [{lo}, {hi})")
    local = local.contiguous()
    if n_clips % world == 0:
        out = torch.empty((n_clips,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    big = -(-n_clips // world)
    padded = torch.zeros((big,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    buf = torch.empty((world * big,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    parts = []
    for r in range(world):
        a, b = shard_bounds(n_clips, r, world)
        parts.append(buf[r * big: r * big + (b - a)])
    return torch.cat(parts, 0)
