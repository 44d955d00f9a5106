#!/usr/bin/env python3
"""The fused 1x1 layers (MCQ_CONV_POST_GDN / _IGDN / _GATE) against their two-launch forms, isolated, on the shapes of the
32-image qp=2 step (and one image with --batch1):  python tools/bench_post.py [--batch1] [--iters 20]

Columns: the 3x3 launch alone | + the 1x1 launch (two launches) | fused (128 x 32 wave tiles)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcquic_amd import ops  # noqa: E402
from mcquic_amd.nn import GenDivNorm, InvGenDivNorm  # noqa: E402
from mcquic_amd.nn.convs import Conv2d  # noqa: E402


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
    return s.elapsed_time(e) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch1", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    n = 1 if args.batch1 else 32
    C = 128
    torch.manual_seed(0)
    rows = []
    for (h, w) in ((768, 512), (384, 256), (192, 128)):           # GDN behind a stride-2 convolution (the 768x512 one is the 3-channel stem's: not fused)
        if h == 768:
            continue
        x = torch.randn(n, C, h, w, device=dev)
        conv = Conv2d(C, C, 3, 2).to(dev).eval()
        gdn = GenDivNorm(C).to(dev).eval()
        pk, post = conv.packed(), gdn.packed_post()
        a = timed(lambda: ops.conv2d(x, pk, 2), args.iters)
        b = timed(lambda: gdn(ops.conv2d(x, pk, 2)), args.iters)
        f = [timed(lambda: ops.conv2d(x, pk, 2, post_gdn=post, tile=t), args.iters) for t in (0x41,)]
        rows.append((f"gdn  {n}x128 {h}x{w} s2", a, b, f))
    for (h, w) in ((192, 128), (96, 64)):                           # IGDN behind a pixelShuffle3x3
        x = torch.randn(n, C, h, w, device=dev)
        conv = Conv2d(C, 4 * C, 3, 1).to(dev).eval()
        igdn = InvGenDivNorm(C).to(dev).eval()
        pk, pks, post = conv.packed(), conv.packed_subpixel(), igdn.packed_post()
        a = timed(lambda: ops.conv2d(x, pk, 1, shuffle2=True), args.iters)
        b = timed(lambda: igdn(ops.conv2d(x, pk, 1, shuffle2=True)), args.iters)
        f = [timed(lambda: ops.conv2d(x, pks, 1, shuffle2=True, post_igdn=post, tile=t), args.iters) for t in (0x41,)]
        rows.append((f"igdn {n}x128 {h}x{w} up", a, b, f))
    for (h, w) in ((192, 128),):                                     # the AttentionBlock's conv1x1 + gate behind its side stack
        x = torch.randn(n, C, h, w, device=dev)
        res, am, xid = torch.randn_like(x), torch.randn_like(x), torch.randn_like(x)
        conv = Conv2d(C, C, 3, 1).to(dev).eval()
        c1 = Conv2d(C, C, 1, 1).to(dev).eval()
        pk, pk1, post = conv.packed(), c1.packed(), c1.packed_post()
        a = timed(lambda: ops.conv2d(x, pk, 1, res=res, dual_silu=True), args.iters)
        b = timed(lambda: ops.conv2d(ops.conv2d(x, pk, 1, res=res, dual_silu=True), pk1, 1, gate_mul=am, gate_id=xid, dual_silu=True), args.iters)
        f = [timed(lambda: ops.conv2d(x, pk, 1, res=res, post_gate=post, gate_mul=am, gate_id=xid, dual_silu=True, tile=t), args.iters) for t in (0x41,)]
        rows.append((f"gate {n}x128 {h}x{w}", a, b, f))
    print(f"{'case':28s} {'3x3 alone':>10s} {'two launches':>13s} {'fused':>13s}   (us)")
    for name, a, b, f in rows:
        print(f"{name:28s} {a:10.1f} {b:13.1f} {f[0]:13.1f}")


if __name__ == "__main__":
    main()
