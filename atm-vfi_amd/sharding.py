"""Frame-batch data parallelism over the GPUs of one node.

Every frame pair is independent (no cross-sample op in ``forward``), so the path shards
with no data-path collective: rank r interpolates pairs ``r::world``.  The only exchange
is one all-gather of the per-rank output frames (RCCL over xGMI when the backend is
"nccl"; gloo on CPU in the tests).  Weights are replicated (206 MB for base)."""
from __future__ import annotations

from typing import Callable, List, Sequence

import torch
import torch.distributed as dist


def shard_indices(n_pairs: int, rank: int, world: int) -> List[int]:
    return list(range(rank, n_pairs, world))


def gather_frames(local: torch.Tensor, n_pairs: int, rank: int, world: int) -> torch.Tensor:
    """all-gather ``[n_local,3,H,W]`` outputs back into pair order ``[n_pairs,3,H,W]``.
    Ranks may hold different counts (ragged tail): shorter ranks are padded for the
    collective and the padding is dropped afterwards."""
    if world == 1:
        return local
    per = (n_pairs + world - 1) // world
    shape = (per,) + tuple(local.shape[1:])
    send = local.new_zeros(shape)
    send[:local.shape[0]] = local
    parts = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(parts, send.contiguous())
    out = local.new_empty((n_pairs,) + tuple(local.shape[1:]))
    for r in range(world):
        idx = shard_indices(n_pairs, r, world)
        if idx:
            out[idx] = parts[r][:len(idx)]
    return out


def interpolate_sharded(forward: Callable, im0: torch.Tensor, im1: torch.Tensor, rank: int, world: int,
                        micro_batch: int = 1) -> torch.Tensor:
    """Run ``forward(im0[i], im1[i])['I_t']`` for this rank's pairs and all-gather the frames."""
    n = im0.shape[0]
    mine = shard_indices(n, rank, world)
    outs = []
    for s in range(0, len(mine), micro_batch):
        idx = mine[s:s + micro_batch]
        outs.append(forward(im0[idx], im1[idx])["I_t"])
    local = torch.cat(outs, 0) if outs else im0.new_zeros((0,) + tuple(im0.shape[1:]))
    return gather_frames(local, n, rank, world)
