"""CPU oracle for the validation metrics either side of the Compressor path (SURVEY.md §8 "next" row 3):
MS-SSIM, PSNR and the ideal (entropy) bits-per-pixel of a code set.

TEST INFRASTRUCTURE ONLY (same rule as oracle/mcquic_ref.py): only `tests/`, `__graft_entry__.smoke()` and
bench-side checkers import it; the product computes these metrics with the HIP kernels in
`mcquic_amd/csrc/metrics.hip`.

Parity status: pinned against the reference's own `mcquic/validate/metrics.py` (`MsSSIM`, `PSNR`) and
`mcquic/validate/handlers.py` (`IdealBPP`), imported unmodified from /root/reference by
`tests/golden/make_golden.py`; the captured values are `tests/golden/f7_metrics.npz`, and
`tests/test_oracle_vs_reference.py` repeats the comparison live when the reference tree is present.

The restatement is written with explicit shifted-slice sums (taps added in index order, multiply and add rounded
separately) rather than library convolutions, so that the HIP kernel can follow the SAME fp32 operation order; what
remains between the two is the order of the final means only.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import numpy as np
import torch

MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)      # mcquic/validate/metrics.py:19
WIN_SIZE, WIN_SIGMA = 11, 1.5                                # metrics.py:222 (MsSSIM defaults)
K1, K2 = 0.01, 0.03                                          # metrics.py:222


def gauss_window(size: int = WIN_SIZE, sigma: float = WIN_SIGMA) -> torch.Tensor:
    """metrics.py:22-37 `_fspecial_gauss_1d`: exp(-(i - size//2)^2 / (2 sigma^2)), normalised, float32 throughout."""
    coords = torch.arange(size).float() - (size // 2)
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def _blur_valid(x: torch.Tensor, win: torch.Tensor) -> torch.Tensor:
    """metrics.py:40-66 `_gaussian_filter`: 'valid' separable blur, rows (H) first, then columns (W).
    A dimension shorter than the window is left unfiltered (the reference warns and skips it)."""
    t = win.numel()
    out = x
    if out.shape[-2] >= t:
        ho = out.shape[-2] - t + 1
        acc = win[0] * out[..., 0:ho, :]
        for i in range(1, t):
            acc = acc + win[i] * out[..., i:i + ho, :]
        out = acc
    if out.shape[-1] >= t:
        wo = out.shape[-1] - t + 1
        acc = win[0] * out[..., :, 0:wo]
        for i in range(1, t):
            acc = acc + win[i] * out[..., :, i:i + wo]
        out = acc
    return out


def ssim_and_cs(x: torch.Tensor, y: torch.Tensor, win: torch.Tensor, data_range: float = 255.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """metrics.py:69-104 `_ssim`: per-(image, channel) means of the SSIM map and of its contrast-structure factor."""
    c1 = (K1 * data_range) ** 2
    c2 = (K2 * data_range) ** 2
    mu1, mu2 = _blur_valid(x, win), _blur_valid(y, win)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = _blur_valid(x * x, win) - mu1_sq
    s2 = _blur_valid(y * y, win) - mu2_sq
    s12 = _blur_valid(x * y, win) - mu12
    cs_map = (2 * s12 + c2) / (s1 + s2 + c2)
    ssim_map = ((2 * mu12 + c1) / (mu1_sq + mu2_sq + c1)) * cs_map
    return ssim_map.flatten(2).mean(-1), cs_map.flatten(2).mean(-1)


def _halve(x: torch.Tensor) -> torch.Tensor:
    """metrics.py:177-179: avg_pool2d(kernel 2, stride 2, padding = size % 2), zero padding counted in the divisor.
    Window (i, j) covers rows 2i - ph, 2i - ph + 1 and columns 2j - pw, 2j - pw + 1; the four taps are added in
    row-major order."""
    h, w = x.shape[-2:]
    ph, pw = h % 2, w % 2
    ho, wo = (h + 2 * ph - 2) // 2 + 1, (w + 2 * pw - 2) // 2 + 1
    p = torch.zeros(x.shape[:-2] + (2 * ho, 2 * wo), dtype=x.dtype)
    hh, ww = min(h, 2 * ho - ph), min(w, 2 * wo - pw)
    p[..., ph:ph + hh, pw:pw + ww] = x[..., :hh, :ww]
    return (((p[..., 0::2, 0::2] + p[..., 0::2, 1::2]) + p[..., 1::2, 0::2]) + p[..., 1::2, 1::2]) / 4


def ms_ssim(x_u8: torch.Tensor, y_u8: torch.Tensor) -> torch.Tensor:
    """metrics.py:142-193 `ms_ssim` as the validator calls it (handlers.py:14-27: uint8 images `.float()`, data range
    255, five levels, per-image result).  Returns the MS-SSIM VALUE in [0, 1], shape [N], float32
    (the reference's module returns 1 - value; see `ms_ssim_db`)."""
    x, y = x_u8.float(), y_u8.float()
    if min(x.shape[-2:]) <= (WIN_SIZE - 1) * 16:
        raise ValueError("image side must exceed 160 pixels for the four down-samplings")    # metrics.py:163-166
    win = gauss_window()
    weights = torch.tensor(MS_WEIGHTS)
    levels = []
    for lv in range(5):
        s, cs = ssim_and_cs(x, y, win)
        if lv < 4:
            levels.append(torch.relu(cs))
            x, y = _halve(x), _halve(y)
        else:
            levels.append(torch.relu(s))
    stack = torch.stack(levels, dim=1)                                  # [N, level, C]
    return torch.prod(stack ** weights.view(1, -1, 1), dim=1).mean(1)


def ms_ssim_db(value: torch.Tensor) -> torch.Tensor:
    """handlers.py:18,26 + validate/utils.py:6-12: Decibel(1.0) of the module output 1 - ms_ssim -> -10 log10(1 - v)."""
    return -10 * (1.0 - value).log10()


def psnr_u8(x_u8: torch.Tensor, y_u8: torch.Tensor) -> torch.Tensor:
    """metrics.py:264-274 `PSNR.forward`: float64 mean squared error per image, 10 log10(255^2 / (mse + 1e-4))."""
    mse = ((x_u8.double() - y_u8.double()) ** 2).mean(dim=(1, 2, 3))
    return 10.0 * (255.0 ** 2 / (mse + 1e-4)).log10()


def ideal_bpp(histograms: Sequence[torch.Tensor], code_counts: Sequence[torch.Tensor], total_pixels: int) -> float:
    """handlers.py:110-187 `IdealBPP.Result`: per level and group the empirical entropy (bits) of the accumulated code
    histogram [m, k] times the number of codes of that group [m], summed, per image pixel.  float32 like the handler."""
    total = 0.0
    for usage, count in zip(histograms, code_counts):
        usage = usage.float()
        prob = usage / usage.sum(-1, keepdim=True)
        ent = prob.log2()
        ent[ent == float("-inf")] = 0
        ent = -(prob * ent).sum(-1)
        total += float((ent * count.float()).sum())
    return total / float(total_pixels)


def make_u8_pair(seed: int, n: int, h: int, w: int, noise: float = 12.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """Seeded image-like uint8 pair [n, 3, h, w]: smooth random fields plus texture, and a noisy, slightly blurred copy."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    img = np.zeros((n, 3, h, w))
    for i in range(n):
        for c in range(3):
            f = np.zeros((h, w))
            for _ in range(6):
                fy, fx, ph = rng.uniform(0.005, 0.08), rng.uniform(0.005, 0.08), rng.uniform(0, 2 * math.pi)
                f += rng.uniform(10, 40) * np.sin(2 * math.pi * (fy * yy + fx * xx) + ph)
            img[i, c] = 128 + f + rng.normal(0, 6, (h, w))
    x = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    soft = img.copy()
    soft[..., 1:-1, 1:-1] = 0.5 * img[..., 1:-1, 1:-1] + 0.125 * (img[..., :-2, 1:-1] + img[..., 2:, 1:-1] +
                                                                   img[..., 1:-1, :-2] + img[..., 1:-1, 2:])
    y = np.clip(np.rint(soft + rng.normal(0, noise, img.shape)), 0, 255).astype(np.uint8)
    return torch.from_numpy(x), torch.from_numpy(y)


CODE_BATCH_KS = (8192, 2048, 512)


def make_code_batches(seed: int = 77, batches: int = 2) -> List[List[torch.Tensor]]:
    """Seeded skewed (geometric) code batches of the qp=2 geometry: per batch three levels [3, 2, h_l, w_l] int64."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(batches):
        out.append([torch.from_numpy(np.minimum(rng.geometric(8.0 / k, (3, 2, hh, ww)) - 1, k - 1).astype(np.int64))
                    for k, (hh, ww) in zip(CODE_BATCH_KS, ((48, 32), (24, 16), (12, 8)))])
    return out
