"""Multi-GPU form of compute(crc, buf): SURVEY.md section 8(e).

A buffer is split into contiguous ranges, one per rank (one process per GPU).
Each rank reduces its own range on its own GPU with no data-path collective;
the only exchange is one (state, length) pair per rank -- 16 bytes -- gathered
with torch.distributed (RCCL over xGMI on GPUs, gloo in the CPU tests), after
which every rank stitches the pairs with sift_crc32_concat:

    compute(s, A|B) = compute(s, A) * x^(8|B|)  +  compute(0, B)

The reference is single-process (crc.rs has no counterpart for this).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch
import torch.distributed as dist

from . import crc as _crc


def stitch(pairs: Sequence[Sequence[int]]) -> int:
    """Fold [(state_r, len_r)] in rank order; state_0 already carries the
    incoming crc, every later state_r started from 0."""
    total = int(pairs[0][0]) & 0xFFFFFFFF
    for state, length in pairs[1:]:
        total = _crc.concat(total, int(state) & 0xFFFFFFFF, int(length))
    return total


def local_state_async(crc: int, local: torch.Tensor, rank: int, pair: torch.Tensor,
                      workspace: Optional[torch.Tensor] = None, stream: Optional[int] = None) -> None:
    """Launch this rank's reduction on the current stream.  `pair` is a
    2-element int64 device tensor the caller initialised to [0, local.numel()];
    the kernel stores the u32 state into the low half of pair[0] (little
    endian).  Nothing else is launched, so a step is the reduction alone."""
    _crc.compute_async(crc if rank == 0 else 0, local, pair[0:1].view(torch.int32), workspace, stream)


def sharded_compute(crc: int, local, group=None,
                    local_compute: Optional[Callable[[int, object], int]] = None) -> int:
    """compute(crc, concat of every rank's `local` in rank order) on every rank.

    `local_compute(state, buf) -> state` defaults to the GPU path; the CPU
    (gloo) tests pass the oracle here to exercise the exchange and stitch.
    """
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    n = local.numel() if isinstance(local, torch.Tensor) else len(local)
    if local_compute is None:
        dev = local.device
        pair = torch.tensor([0, n], dtype=torch.int64, device=dev)
        local_state_async(crc, local, rank, pair)
    else:
        state = local_compute(crc if rank == 0 else 0, local)
        dev = local.device if isinstance(local, torch.Tensor) else torch.device("cpu")
        pair = torch.tensor([state & 0xFFFFFFFF, n], dtype=torch.int64, device=dev)
    gathered = torch.empty(2 * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(gathered, pair, group=group)
    pairs = gathered.view(world, 2).cpu().tolist()
    return stitch([(s & 0xFFFFFFFF, l) for s, l in pairs])


def exchange_and_stitch(pairs: torch.Tensor, group=None) -> list:
    """`pairs` is this rank's [steps, 2] int64 tensor of (state, length) for
    `steps` independent buffers, each split over the ranks in rank order.  One
    all-gather moves all of them (16 bytes per rank per buffer); returns the
    `steps` stitched registers, identical on every rank."""
    world = dist.get_world_size(group)
    steps = pairs.shape[0]
    flat = torch.empty(world * steps * 2, dtype=torch.int64, device=pairs.device)
    dist.all_gather_into_tensor(flat, pairs.reshape(-1).contiguous(), group=group)
    g = flat.view(world, steps, 2).cpu().tolist()
    return [stitch([(g[r][i][0] & 0xFFFFFFFF, g[r][i][1]) for r in range(world)])
            for i in range(steps)]
