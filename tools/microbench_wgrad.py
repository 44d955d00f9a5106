#!/usr/bin/env python3
"""Weight-gradient kernels at the shapes of the config-#5 training step (8 x 256 x 256 crops): the NCHW row-walk kernel
(csrc/wgrad_rows.hip) against the NHWC kernel + its transposes (csrc/train_ops.hip).  Kernel tuning aid."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcquic_amd import ops  # noqa: E402

SHAPES = [(8, 128, 128, 128, 128), (8, 128, 128, 64, 64), (8, 128, 512, 64, 64), (8, 128, 128, 32, 32), (8, 128, 128, 16, 16),
          (8, 128, 128, 8, 8), (8, 128, 12, 128, 128)]


def timeit(fn, iters=20):
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


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", type=int, default=-1, help="index into SHAPES (default: all)")
    ap.add_argument("--only", default="", help="rows | nhwc")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    print("lib:", os.environ.get("MCQUIC_AMD_LIB", "default"))
    for n, cin, cout, h, w in (SHAPES if args.shape < 0 else [SHAPES[args.shape]]):
        xs = [torch.randn(n, cin, h, w, device=dev) for _ in range(4)]
        dys = [torch.randn(n, cout, h, w, device=dev) for _ in range(4)]
        flops = 2.0 * n * h * w * cout * cin * 9
        row = []
        for rows in ((True, False) if not args.only else (args.only == 'rows',)):
            ops._WGRAD_ROWS = rows
            i = [0]

            def fn():
                i[0] += 1
                return ops.conv2d_wgrad(xs[i[0] % 4], dys[i[0] % 4], 3, 1, want_bias=True)
            us = timeit(fn)
            row.append(f"{'rows' if rows else 'nhwc'} {us:8.1f} us {flops / us / 1e6:6.1f} TF")
        ops._WGRAD_ROWS = True
        print(f"{n}x{cin}->{cout} {h}x{w}: " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
