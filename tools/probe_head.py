#!/usr/bin/env python3
"""Time the image head conv (128 -> 12 channels + PixelShuffle(2) at 384x256, batch 32) -- kernel tuning aid."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcquic_amd import ops
dev = torch.device("cuda:0")
x = torch.randn(32, 128, 384, 256, device=dev)
packs = [ops.PackedConv(torch.randn(12, 128, 3, 3, device=dev) * 0.03, torch.randn(12, device=dev)) for _ in range(4)]
for i in range(3):
    y = ops.conv2d(x, packs[i], 1, shuffle2=True)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for i in range(20):
    ops.conv2d(x, packs[i % 4], 1, shuffle2=True)
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / 20
print(f"lib {os.environ.get('MCQUIC_AMD_LIB', 'default')}: head conv {us:.1f} us, {2.0 * 32 * 384 * 256 * 128 * 9 * 12 / us / 1e6:.1f} TFLOP/s (12 real rows), checksum {float(y.double().sum()):.6f}")
