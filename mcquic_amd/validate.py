"""Validation-side helpers around the Compressor (reference: mcquic/validate/validator.py:40-97,
mcquic/validate/metrics.py:264-274, mcquic/validate/handlers.py).  Caller-level code: statistics only.

`speed` is the reference's throughput protocol verbatim (random 10x3x768x512 batch, one warm-up, 50 `compress` then 50
`decompress` calls timed with events on the current stream, Mpps = 50*10*768*512/1000/ms) -- byte streams included,
exactly what the README's 25.45 / 22.03 Mpps (RTX 3090) were measured with.  `validate` restores a batch and returns
per-image PSNR (on the de-transformed uint8 images) and bits per pixel; with torch.distributed initialised the rows of
all ranks are gathered (parallel.gather_image_stats), images being sharded by the caller (parallel.shard_range).
"""
from __future__ import annotations

from typing import Tuple

import torch

from . import ops, parallel


@torch.inference_mode()
def speed(model, iters: int = 50, batch: int = 10, height: int = 768, width: int = 512) -> Tuple[float, float]:
    device = next(model.parameters()).device
    tensor = torch.rand(batch, 3, height, width).to(device)
    startEvent = torch.cuda.Event(enable_timing=True)
    endEvent = torch.cuda.Event(enable_timing=True)
    codes, binaries, headers = model.compress(tensor)          # warm up
    model.decompress(binaries, headers)
    startEvent.record()
    for _ in range(iters):
        codes, binaries, headers = model.compress(tensor)
    endEvent.record()
    torch.cuda.synchronize()
    encoderMs = startEvent.elapsed_time(endEvent)
    startEvent.record()
    for _ in range(iters):
        model.decompress(binaries, headers)
    endEvent.record()
    torch.cuda.synchronize()
    decoderMs = startEvent.elapsed_time(endEvent)
    mpx = iters * batch * height * width / 1000
    return mpx / encoderMs, mpx / decoderMs


def psnr(x_u8: torch.Tensor, y_u8: torch.Tensor) -> torch.Tensor:
    """Per-image PSNR of uint8 batches, float64, upper bound 255 (metrics.py:264-274)."""
    mse = ((x_u8.double() - y_u8.double()) ** 2).mean(dim=(1, 2, 3))
    return 10.0 * (255.0 ** 2 / (mse + 1e-4)).log10()


@torch.inference_mode()
def validate(model, images: torch.Tensor, group=None) -> torch.Tensor:
    """images: this rank's shard, fp32 [n, 3, h, w] in [-1, 1].  Returns rows [psnr_db, bpp] for ALL images (rank order)."""
    codes, binaries, headers = model.compress(images)
    restored = model.decompress(binaries, headers)
    p = psnr(ops.detransform(images.contiguous()), ops.detransform(restored.contiguous()))
    pixels = images.shape[-2] * images.shape[-1]
    bpp = torch.tensor([sum(len(s) for s in b) * 8 / pixels for b in binaries], dtype=torch.float64, device=images.device)
    return parallel.gather_image_stats(torch.stack([p, bpp], 1), group)
