#!/usr/bin/env python3
"""Per-shape conv timings on the GPU for forced wave tiles / split-K factors (kernel tuning aid).

    python tools/microbench_conv.py                 # default shape list
    MCQUIC_AMD_LIB=path/to/variant.so python tools/microbench_conv.py

Each measurement cycles through `--nweights` distinct weight tensors so that a launch never finds its own
filter bank hot in L2 (as in the real network, where consecutive convs use different weights).
"""
import argparse
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcquic_amd import ops  # noqa: E402

SHAPES = [  # n, cin, cout, h, w, ks, stride
    (32, 128, 128, 384, 256, 3, 1),
    (32, 128, 128, 192, 128, 3, 1),
    (32, 128, 128, 96, 64, 3, 1),
    (32, 128, 128, 48, 32, 3, 1),
    (32, 128, 128, 24, 16, 3, 1),
    (32, 128, 128, 12, 8, 3, 1),
    (32, 128, 128, 12, 8, 1, 1),
    (32, 128, 128, 24, 16, 1, 1),
    (32, 128, 128, 48, 32, 1, 1),
    (32, 128, 128, 96, 64, 1, 1),
    (32, 128, 128, 192, 128, 1, 1),
    (32, 128, 128, 384, 256, 1, 1),
]
TILES = [0, 0x42, 0x142, 0x242, 0x342, 0x41, 0x141, 0x241, 0x341, 0x22, 0x122, 0x222, 0x322, 0x21, 0x121, 0x221, 0x11, 0x311]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--nweights", type=int, default=12)
    ap.add_argument("--flags", default="res")
    ap.add_argument("--small", action="store_true", help="only the three small levels")
    ap.add_argument("--big", action="store_true", help="only the two largest levels")
    ap.add_argument("--k1", action="store_true", help="only the 1x1 shapes")
    ap.add_argument("--winograd", action="store_true", help="the opt-in Winograd forms beside the direct one (3x3 stride-1 shapes): columns direct | F(2,3) 128-row | F(2x2,3x3)")
    ap.add_argument("--nprob", type=int, default=1, help="problems per launch (ops.conv2d_multi)")
    ap.add_argument("--batch1", action="store_true", help="the 3x3 shapes of one 768x512 image (batch-1 latency)")
    ap.add_argument("--c192", action="store_true", help="the 3x3 shapes of model No. 12 (channel 192) at 16 images: 64-row bands (0x22) against the 128-row tile (0x42)")
    ap.add_argument("--train", action="store_true", help="the 3x3 shapes of the config-#5 training step (8 images of 128x128 ... 4x4)")
    ap.add_argument("--neon", action="store_true", help="the 3x3 shapes of Neon(32, ...) at 4 x 512x512 (widths 32 / 64 / 8 at full resolution ... 32x32 maps): "
                                                    "32- and 64-row tiles, one / two / four pixel blocks per wave")
    ap.add_argument("--only3232", action="store_true", help="with --neon: only the 32 -> 32 shapes")
    ap.add_argument("--tiles", default=None, help="comma-separated hex tile codes instead of the full list, e.g. 0,0x1f,0x11")
    args = ap.parse_args()
    tiles = TILES if args.tiles is None else [int(t, 16) for t in args.tiles.split(",")]
    dev = torch.device("cuda:0")
    print("lib:", os.environ.get("MCQUIC_AMD_LIB", "default"))
    train = [(8, 128, 128, s, s, 3, 1) for s in (128, 64, 32, 16, 8, 4)]
    batch1 = [(1, 128, 128, hh, ww, 3, 1) for hh, ww in ((384, 256), (192, 128), (96, 64), (48, 32), (24, 16), (12, 8))]
    c192 = [(16, 192, 192, hh, ww, 3, 1) for hh, ww in ((384, 256), (192, 128), (96, 64), (48, 32), (24, 16), (12, 8))]
    neon = [(4, 32, 32, s_, s_, 3, 1) for s_ in (512, 256, 128, 64)] + [(4, 32, 64, 64, 64, 3, 1), (4, 64, 64, 64, 64, 3, 1), (4, 64, 8, 64, 64, 3, 1),
                                                                      (4, 8, 32, 64, 64, 3, 1), (4, 32, 32, 32, 32, 3, 1), (4, 32, 32, 16, 16, 3, 1), (4, 3, 32, 512, 512, 3, 1),
                                                                      (4, 32, 3, 512, 512, 3, 1)]
    if args.only3232:
        neon = [sh for sh in neon if sh[1] == 32 and sh[2] == 32 and sh[3] >= 64]
    if args.neon and args.tiles is None:
        tiles = [0, 0x11, 0x12, 0x14, 0x21, 0x22, 0x111, 0x112, 0x121]
    for (n, cin, cout, h, w, ks, stride) in neon if args.neon else c192 if args.c192 else train if args.train else batch1 if args.batch1 else (SHAPES[3:6] if args.small else SHAPES[:2] if args.big else SHAPES[6:] if args.k1 else SHAPES[:7]):
        x = torch.randn(n, cin, h, w, device=dev)
        res = torch.randn(n, cout, h // stride, w // stride, device=dev)
        packs = [ops.PackedConv(torch.randn(cout, cin, ks, ks, device=dev) * 0.03, torch.randn(cout, device=dev), winograd=2 if args.winograd else None)
                 for _ in range(args.nweights)]
        flops = 2.0 * n * (h // stride) * (w // stride) * cout * cin * ks * ks
        row = []
        for tile in tiles:
            if h * w > 100 * 64 and tile not in (0, 0x42, 0x41, 0x22, 0x11) and not args.train and not args.batch1 and not args.neon and args.tiles is None:
                continue
            kw = dict(tile=tile)
            if args.winograd:
                if ks != 3 or stride != 1 or tile not in (0, 0x42, 0x22):
                    continue
                kw["winograd"] = False if tile == 0 else 1 if tile == 0x42 else 2      # columns: direct | F(2,3) 128-row | F(2x2,3x3)
                kw["tile"] = 0
            if args.flags == "res":
                kw.update(res=res, dual_silu=True)
            elif args.flags == "resonly":
                kw.update(res=res)
            elif args.flags == "dual":
                kw.update(dual_silu=True)
            elif args.flags == "gdn":
                kw.update(square_in=True, gdn_mul=x)
            elif args.flags == "silu_out":
                kw.update(silu_out=True)
            elif args.flags == "dsilu":                       # the input-gradient launch of a ResidualBlock: * silu'(x) + dy
                kw.update(dsilu_mul=res, res=res)
            elif args.flags == "dsilu_only":
                kw.update(dsilu_mul=res)
            def launch(i):
                if args.nprob == 1:
                    ops.conv2d(x, packs[i % len(packs)], stride, **kw)
                else:
                    shared = {k_: v for k_, v in kw.items() if not torch.is_tensor(v)}
                    pp = {k_: v for k_, v in kw.items() if torch.is_tensor(v)}
                    ops.conv2d_multi([x] * args.nprob, [packs[(i + j) % len(packs)] for j in range(args.nprob)], stride,
                                     per_problem=[pp] * args.nprob, **shared)
            try:
                for i in range(3):
                    launch(i)
            except RuntimeError:                              # (a forced kernel that does not take this shape)
                continue
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(args.iters):
                launch(i)
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) * 1e3 / args.iters
            row.append(f"{tile:#05x}:{us:8.1f}us {flops * args.nprob / us / 1e6:6.1f}TF")
        print(f"{n}x{cin}->{cout} {h}x{w} k{ks}s{stride}: " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
