#!/usr/bin/env python3
"""BASELINE config #4: the VQ distance + argmin kernel in isolation (M=4 codebooks, K=4096, D=256, 49152 latent
vectors per codebook; SURVEY.md section 8(d) inputs), plus the three qp=2 level shapes.  Prints one JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcquic_amd import ops  # noqa: E402

PEAK = 157.3


def run(m, k, d, n, h, w, iters=10):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    x = (torch.randn((n, m * d, h, w), generator=g) * 0.1).to(dev)
    cb = ops.PackedCodebook((torch.randn((m, k, d), generator=g) * (2 / (5 * d)) ** 0.5).to(dev))
    for _ in range(2):
        ops.vq_assign(x, cb)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        codes = ops.vq_assign(x, cb)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    flops = 2.0 * m * n * h * w * k * d
    return {"shape": f"m={m} k={k} d={d} vectors/codebook={n * h * w}", "ms": round(ms, 4), "gflop": round(flops / 1e9, 2),
            "tflops": round(flops / ms / 1e9, 2), "frac_of_fp32_mfma_peak": round(flops / ms / 1e9 / PEAK, 4),
            "assignments_per_s": round(m * n * h * w / ms * 1e3, 1), "code_checksum": int(codes.sum())}


if __name__ == "__main__":
    out = {"metric": "VQ distance/argmin kernel in isolation (fp32 MFMA)", "peak_tflops": PEAK,
           "config4": run(4, 4096, 256, 32, 48, 32),
           "qp2_levels": [run(2, 8192, 64, 32, 48, 32), run(2, 2048, 64, 32, 24, 16), run(2, 512, 64, 32, 12, 8)],
           # the reference's model No. 12: twelve codebooks of 16-dimensional codewords (a 64-MFMA tile against a 128-distance epilogue)
           "model12_levels": [run(12, 8192, 16, 16, 48, 32), run(12, 2048, 16, 16, 24, 16), run(12, 512, 16, 16, 12, 8)]}
    print(json.dumps(out))
