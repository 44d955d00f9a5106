"""Validation-side helpers around the Compressor (reference: mcquic/validate/validator.py:40-97,
mcquic/validate/metrics.py:264-274, mcquic/validate/handlers.py).  Caller-level code: statistics only.

`speed` is the reference's throughput protocol verbatim (random 10x3x768x512 batch, one warm-up, 50 `compress` then 50
`decompress` calls timed with events on the current stream, Mpps = 50*10*768*512/1000/ms) -- byte streams included,
exactly what the README's 25.45 / 22.03 Mpps (RTX 3090) were measured with.  `validate` restores a batch and returns
per-image PSNR and MS-SSIM (on the de-transformed uint8 images, HIP kernels of csrc/metrics.hip) and bits per pixel;
with torch.distributed initialised the rows of all ranks are gathered (parallel.gather_image_stats), images being
sharded by the caller (parallel.shard_range).  `ideal_bpp` is the handlers' entropy estimate over code histograms.
"""
from __future__ import annotations

from typing import Sequence, Tuple

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
    """Per-image PSNR of uint8 batches, float64, upper bound 255 (metrics.py:264-274).  The squared-error sum is taken
    exactly in integers on the device, so the float64 mean squared error equals the reference's bit for bit."""
    mse = ops.sqdiff_sum(x_u8, y_u8).double() / float(x_u8[0].numel())
    return 10.0 * (255.0 ** 2 / (mse + 1e-4)).log10()


def ms_ssim_db(x_u8: torch.Tensor, y_u8: torch.Tensor) -> torch.Tensor:
    """Per-image MS-SSIM in dB as the reference's handler reports it (handlers.py:14-27: Decibel(1.0) of the MsSSIM
    module's 1 - ms_ssim, i.e. -10 log10(1 - value)); float32 like the reference."""
    return -10.0 * (1.0 - ops.ms_ssim(x_u8, y_u8)).log10()


def ideal_bpp(histograms: Sequence[torch.Tensor], total_pixels: int) -> float:
    """Entropy-coded size estimate of a validation run (handlers.py:110-187 `IdealBPP.Result`): for every level and group
    the empirical entropy of the accumulated code histogram [m, k] (e.g. parallel.code_histograms summed over the run)
    times the number of codes of that group, summed, per image pixel."""
    total = 0.0
    for usage in histograms:
        usage = usage.float()
        count = usage.sum(-1)
        prob = usage / count[:, None]
        ent = prob.log2()
        ent[ent == float("-inf")] = 0
        total += float((-(prob * ent).sum(-1) * count).sum())
    return total / float(total_pixels)


@torch.inference_mode()
def validate(model, images: torch.Tensor, group=None, msssim: bool = True, gather: bool = True) -> torch.Tensor:
    """images: this rank's shard, fp32 [n, 3, h, w] in [-1, 1].  Returns float64 rows [psnr_db, ms_ssim_db, bpp] for ALL
    images (rank order) -- the [psnr, ms_ssim, bits] statistics of the reference's validator (validator.py:40-58).
    `msssim=False` fills the MS-SSIM column with NaN (the metric is undefined for sides <= 160 pixels); `gather=False`
    returns this rank's rows only."""
    if images.shape[0] == 0:                                  # an empty shard still takes part in the gather
        rows = torch.empty((0, 3), dtype=torch.float64, device=images.device)
        return parallel.gather_image_stats(rows, group) if gather else rows
    codes, binaries, headers = model.compress(images)
    restored = model.decompress(binaries, headers)
    a, b = ops.detransform(images.contiguous()), ops.detransform(restored.contiguous())
    p = psnr(a, b)
    s = ms_ssim_db(a, b).double() if msssim else torch.full_like(p, float("nan"))
    pixels = images.shape[-2] * images.shape[-1]
    bpp = torch.tensor([sum(len(s_) for s_ in bi) * 8 / pixels for bi in binaries], dtype=torch.float64, device=images.device)
    rows = torch.stack([p, s, bpp], 1)
    return parallel.gather_image_stats(rows, group) if gather else rows
