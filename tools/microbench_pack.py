#!/usr/bin/env python3
"""Time of the weight re-pack an optimizer step makes necessary (operand streams of the forward and the input-gradient
convolution): sixteen 128x128x3x3 weights per launch, as Conv2d.repack_stale groups them.
    [MCQUIC_AMD_LIB=variant.so] python tools/microbench_pack.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mcquic_amd import ops
dev = torch.device("cuda:0")
print("lib:", os.environ.get("MCQUIC_AMD_LIB", "default"))
for (co, ci, n) in [(128, 128, 16), (128, 128, 4), (512, 128, 4), (128, 512, 2)]:
    ws = [torch.randn(co, ci, 3, 3, device=dev) for _ in range(n)]
    for dgrad in (False, True):
        for _ in range(3):
            ops.pack_convs(ws, None, dgrad=dgrad)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            pk = ops.pack_convs(ws, None, dgrad=dgrad)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        mb = n * pk[0].wp.numel() * 4 / 1e6
        print(f"{n:2d} x {co}x{ci}x3x3 {'dgrad' if dgrad else 'fwd  '}: {us:7.1f} us per launch group, {mb:6.1f} MB written, {mb / us / 1e3 * 1e3:6.2f} GB/ms")
