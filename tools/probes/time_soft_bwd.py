#!/usr/bin/env python3
"""Event timing of the soft assignment's kernels at the three levels of the config-#5 training geometry (8 x 256x256 crops)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mcquic_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, iters=20):
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


for (n, m, d, h, w, k) in ((8, 2, 64, 16, 16, 8192), (8, 2, 64, 8, 8, 2048), (8, 2, 64, 4, 4, 512)):
    x = torch.randn((n, m * d, h, w), device=dev)
    ddeq = torch.randn_like(x)
    cb = ops.PackedCodebook(torch.randn((m, k, d), device=dev))
    temperature = torch.ones((m,), device=dev)
    freq = torch.full((m, k), 1.0 / k, device=dev)
    expo = torch.tensor([1.0], device=dev)
    ud, ug = torch.rand((n, m, h, w, k), device=dev), torch.rand((n, m, h, w, k), device=dev)
    logits = ops.vq_logits(x, cb, temperature, 1e-6)
    t_logits = timed(lambda: ops.vq_logits(x, cb, temperature, 1e-6))
    work = logits.clone()
    t_sample = timed(lambda: ops.vq_gumbel_sample(work, ud, ug, freq, expo))
    code, index, hot = ops.vq_gumbel_sample(logits, ud, ug, freq, expo)
    ds = ops.vq_inner(ddeq, cb)
    t_inner = timed(lambda: ops.vq_inner(ddeq, cb))
    keep = ds.clone()
    t_sm = timed(lambda: ops.vq_softmax_bwd(logits, ug, keep, temperature, 1e-6))
    rowsum, dtrow = ops.vq_softmax_bwd(logits, ug, ds, temperature, 1e-6)
    t_bwd = timed(lambda: ops.vq_soft_bwd(ds, rowsum, x, ddeq, index, hot, cb))
    print(f"{n}x{m}x{d} {h}x{w} k={k}: logits {t_logits:7.1f} us | sample {t_sample:7.1f} | inner {t_inner:7.1f} | softmax_bwd {t_sm:7.1f} | "
          f"dx + dC (+ 2 transposes) {t_bwd:7.1f}", flush=True)
