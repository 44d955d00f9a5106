#!/usr/bin/env python3
"""Throughput of the validation-metric kernels (csrc/metrics.hip) on the BASELINE batch geometry.

    python tools/bench_metrics.py [--batch 32] [--iters 20]

Reports per call: MS-SSIM and PSNR time for uint8 [batch, 3, 768, 512] pairs, the algorithmic HBM bytes
(every pyramid level read once per image pair + the pooled levels written once) and the fp32 operation count of the
separable blur (5 moments x 2 passes x 11 taps x 2 ops per output pixel), against the HBM and VALU peaks.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcquic_amd import ops  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    x = torch.randint(0, 256, (a.batch, 3, 768, 512), generator=g, dtype=torch.uint8).to(dev)
    y = (x.long() + torch.randint(-9, 10, x.shape, generator=g).to(dev)).clamp(0, 255).to(torch.uint8)
    planes = a.batch * 3
    h, w, px, byts = 768, 512, 0, 0
    for lv in range(5):
        elem = 1 if lv == 0 else 4
        byts += 2 * planes * h * w * elem * (2 if lv < 4 else 1)      # ssim pass + pooling pass read the level
        px += planes * (h - 10) * (w - 10)
        if lv < 4:
            h, w = (h + 1) // 2, (w + 1) // 2
            byts += 2 * planes * h * w * 4                             # pooled level written
    flops = px * 5 * 2 * 11 * 2
    ms = timed(lambda: ops.ms_ssim(x, y), a.iters)
    ps = timed(lambda: ops.sqdiff_sum(x, y), a.iters)
    print(f"ms_ssim  batch {a.batch}: {ms:.3f} ms/call  {a.batch / ms * 1e3:.0f} img/s  "
          f"{byts / ms / 1e6:.0f} GB/s algorithmic (HBM peak 8000)  {flops / ms / 1e9:.2f} TFLOP/s fp32 VALU (no-FMA peak ~39)")
    print(f"sqdiff   batch {a.batch}: {ps:.3f} ms/call  {2 * x.numel() / ps / 1e6:.0f} GB/s (HBM peak 8000)")


if __name__ == "__main__":
    main()
