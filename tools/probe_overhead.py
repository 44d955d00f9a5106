#!/usr/bin/env python3
"""Fixed per-launch cost of the conv kernel: time vs input channels at one spatial size (kernel tuning aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcquic_amd import ops  # noqa: E402


def time_conv(x, packs, iters=30, **kw):
    for i in range(5):
        ops.conv2d(x, packs[i % len(packs)], 1, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        ops.conv2d(x, packs[i % len(packs)], 1, **kw)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def main():
    dev = torch.device("cuda:0")
    n, cout = 32, 128
    for (h, w) in ((48, 32), (24, 16), (96, 64)):
        res = torch.randn(n, cout, h, w, device=dev)
        for cin in (16, 32, 64, 128, 256):
            x = torch.randn(n, cin, h, w, device=dev)
            packs = [ops.PackedConv(torch.randn(cout, cin, 3, 3, device=dev) * 0.03, torch.randn(cout, device=dev)) for _ in range(4)]
            row = []
            for tile in (0x42, 0x242, 0x122):
                for name, kw in (("plain", {}), ("res+twin", dict(res=res, dual_silu=True))):
                    row.append(f"{tile:#05x} {name}:{time_conv(x, packs, tile=tile, **kw):7.1f}")
            print(f"{h}x{w} cin={cin:3d}: " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
