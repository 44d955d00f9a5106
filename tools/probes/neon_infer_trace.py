#!/usr/bin/env python3
"""Probe: encode + decode of Neon(32, 4096, [16, 8, 4, 2, 2], denseNorm) on 8 x 3 x 512 x 512, a few times -- run under
`rocprofv3 --kernel-trace` to see what GroupNorm costs beside the convolutions (python tools/probes/neon_infer_trace.py [0|1])."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from mcquic_amd import Neon  # noqa: E402

dense = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = Neon(32, 4096, [16, 8, 4, 2, 2], dense).to(dev).eval()
x = (torch.rand(8, 3, 512, 512) * 2 - 1).to(dev)
with torch.no_grad():
    for _ in range(5):
        codes = model.encode(x)
        y = model.decode(codes)
torch.cuda.synchronize()
print("ok", tuple(y.shape))
