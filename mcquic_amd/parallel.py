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


def _staged(t: torch.Tensor, group=None) -> torch.Tensor:
    """The tensor a collective of `group` should run on: device tensors as they are under RCCL ("nccl"); under gloo (the CPU
    tests, and the two-processes-on-one-GPU test: RCCL refuses two ranks on one device) a host copy."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        return t.cpu()
    return t


def gather_image_stats(local: torch.Tensor, group=None) -> torch.Tensor:
    """all_gather of per-image rows [n_local, f] -> [n_total, f] in rank order (ragged shards allowed)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    home = local.device
    local = _staged(local, group)
    counts = torch.zeros(world, dtype=torch.int64, device=local.device)
    counts[dist.get_rank(group)] = local.shape[0]
    dist.all_reduce(counts, group=group)
    cap = int(counts.max())
    padded = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    return torch.cat([o[: int(c)] for o, c in zip(out, counts)], 0).to(home)


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
        staged = _staged(buf, group)
        dist.all_reduce(staged, group=group)
        buf = staged.to(buf.device)
    out, off = [], 0
    for code, k in zip(codes, ks):
        m = code.shape[1]
        out.append(buf[off: off + m * k].reshape(m, k))
        off += m * k
    return out


def data_parallel(model: torch.nn.Module, device: torch.device, group=None, **kwargs):
    """torch DistributedDataParallel around the training-mode Compressor: gradients produced by the HIP kernels
    (mcquic_amd.autograd) are all-reduced bucket by bucket -- over RCCL / xGMI with backend "nccl" (the reference's
    `torchrun` + DDP set-up, mcquic/train/ddp.py:79-95).  Under gloo with device parameters (tests: two ranks sharing one GPU)
    nothing relies on gloo's device support: the initial state broadcast and the gradient buckets are staged through the host
    (a communication hook), buffers are constants of the layers and are not re-broadcast."""
    from torch.nn.parallel import DistributedDataParallel
    staged = kwargs.pop("stage_through_host", None)
    if staged is None:
        staged = device.type == "cuda" and dist.get_backend(group) == "gloo"
    if staged:
        with torch.no_grad():                                  # rank 0's parameters and buffers, like DDP's own init sync
            for t in list(model.parameters()) + list(model.buffers()):
                host = t.detach().cpu()
                dist.broadcast(host, dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                if dist.get_rank(group) != 0:
                    t.copy_(host)
        kwargs.setdefault("init_sync", False)
        kwargs.setdefault("broadcast_buffers", False)
    ddp = DistributedDataParallel(model, device_ids=[device.index] if device.type == "cuda" else None, process_group=group, **kwargs)
    if staged:
        world = dist.get_world_size(group)

        def staged_allreduce(state, bucket):
            buf = bucket.buffer()
            host = buf.detach().cpu()
            dist.all_reduce(host, group=group)
            buf.copy_(host.div_(world))
            fut = torch.futures.Future()
            fut.set_result(buf)
            return fut
        ddp.register_comm_hook(None, staged_allreduce)
    return ddp
