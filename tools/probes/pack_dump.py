"""Packed operand streams of a few weights, hashed: run once per library (MCQUIC_AMD_LIB) and compare the lines.
    MCQUIC_AMD_LIB=old.so python tools/probes/pack_dump.py; python tools/probes/pack_dump.py"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mcquic_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(5)
for (co, ci, ks, stride) in [(128, 128, 3, 1), (128, 128, 3, 2), (512, 128, 3, 1), (128, 512, 3, 1), (12, 128, 3, 1), (128, 3, 3, 2), (40, 72, 3, 1), (128, 64, 3, 1), (128, 128, 1, 1), (96, 50, 3, 1)]:
    w = torch.randn(co, ci, ks, ks, device=dev)
    b = torch.randn(co, device=dev)
    pk = ops.PackedConv(w, b)
    h = [hashlib.sha256(pk.wp.cpu().numpy().tobytes()).hexdigest()[:16]]
    try:
        dg = ops.PackedConv.dgrad(w, stride, 2.0)
    except NotImplementedError:
        dg = None
    if dg is not None:
        h.append(hashlib.sha256(dg.wp.cpu().numpy().tobytes()).hexdigest()[:16])
    print(co, ci, ks, stride, *h)
ws = [torch.randn(128, 128, 3, 3, device=dev) for _ in range(5)]
for dgrad in (False, True):
    pks = ops.pack_convs(ws, None, dgrad=dgrad)
    print("multi", dgrad, *[hashlib.sha256(p.wp.cpu().numpy().tobytes()).hexdigest()[:16] for p in pks])
