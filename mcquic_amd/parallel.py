"""Multi-GPU layer of the Compressor path: one process per GPU, images sharded, no data-path collective.

Every image is independent in encode / decode, so a batch is cut into contiguous per-rank
slices and each rank runs the whole model on its slice.  Collectives (RCCL over xGMI with backend "nccl", gloo
in the CPU tests) appear only where validation statistics are combined:
  * all_gather of per-image statistics rows (e.g. [psnr, ms_ssim, bits]) -- what the reference's validator
    accumulates on rank 0 only (mcquic/validate/validator.py:40-58);
  * all_reduce(sum) of per-level code histograms [m, k_l] -- what `IdealBPP` counts
    (mcquic/validate/handlers.py:110-187) and what EntropyCoder.forward all-reduces in training
    (mcquic/modules/entropyCoder.py:28-44).  The three levels travel in ONE flat buffer (86 KB for qp=2): the
    message is latency-bound, so one collective instead of three.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) slice of `n` images for `rank`; the first n % world ranks get one extra."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_image_stats(local: torch.Tensor, group=None) -> torch.Tensor:
    """all_gather of per-image rows [n_local, f] -> [n_total, f] in rank order (ragged shards allowed)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    counts = torch.zeros(world, dtype=torch.int64, device=local.device)
    counts[dist.get_rank(group)] = local.shape[0]
    dist.all_reduce(counts, group=group)
    cap = int(counts.max())
    padded = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    return torch.cat([o[: int(c)] for o, c in zip(out, counts)], 0)


def code_histograms(codes: Sequence[torch.Tensor], ks: Sequence[int], group=None) -> List[torch.Tensor]:
    """Per-level code counts [m, k_l] (int64) summed over all images of all ranks with ONE all_reduce."""
    flat = []
    for code, k in zip(codes, ks):
        n, m = code.shape[0], code.shape[1]
        idx = code.permute(1, 0, 2, 3).reshape(m, -1) + (torch.arange(m, device=code.device) * k)[:, None]
        # scatter_add instead of torch.bincount: bincount reads its maximum back to the host (a sync, and illegal
        # inside a captured hipGraph)
        idx = idx.reshape(-1)
        flat.append(torch.zeros(m * k, dtype=torch.int64, device=code.device).scatter_add_(0, idx, torch.ones_like(idx)))
    buf = torch.cat(flat)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, group=group)
    out, off = [], 0
    for code, k in zip(codes, ks):
        m = code.shape[1]
        out.append(buf[off: off + m * k].reshape(m, k))
        off += m * k
    return out
