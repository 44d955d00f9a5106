#!/usr/bin/env python3
"""Isolated timings of the soft assignment's logits kernels at the training step's level-0 shape (8 x 2 x 16x16 vectors, k = 8192,
d = 64: a 134 MB logits tensor) -- `mcq_vq_logits_f32` (forward), `mcq_vq_inner_f32` (backward's raw inner products):
    [MCQUIC_AMD_LIB=variant.so] python tools/probes/time_logits.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mcquic_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (n, m, d, h, w, k) in ((8, 2, 64, 16, 16, 8192), (8, 2, 64, 8, 8, 2048), (8, 2, 64, 4, 4, 512)):
    x = (torch.randn((n, m * d, h, w), generator=g) * 0.1).to(dev)
    cb = ops.PackedCodebook((torch.randn((m, k, d), generator=g) * (2 / (5 * d)) ** 0.5).to(dev))
    t = torch.ones(m, device=dev)

    def timed(fn, iters=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) * 1e3 / iters
    mb = n * m * h * w * k * 4 / 1e6
    a = timed(lambda: ops.vq_logits(x, cb, t, 1e-6))
    b = timed(lambda: ops.vq_inner(x, cb))
    print(f"k={k} {h}x{w}: logits {a:7.1f} us ({mb / a * 1e-0:.2f} MB/us = TB/s)   inner {b:7.1f} us ({mb / b:.2f} TB/s)   tensor {mb:.0f} MB")
