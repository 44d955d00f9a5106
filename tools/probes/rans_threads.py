"""Host rANS coder of a batch: time of EntropyCoder.compress / decompress (CPU tensors: the coder alone) against the number of
pool threads (MCQUIC_AMD_RANS_THREADS), batch 10 (the reference's speed protocol) and batch 32 (the bench's).
    python tools/probes/rans_threads.py            (on the GPU box: its 16 host cores are what the protocol runs on)
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from mcquic_amd.modules.entropyCoder import EntropyCoder

coder = EntropyCoder(2, [8192, 2048, 512])
g = torch.Generator().manual_seed(0)
print("cores", len(os.sched_getaffinity(0)))
for n in (10, 32):
    codes = [torch.randint(0, k, (n, 2, h, w), generator=g) for k, (h, w) in zip([8192, 2048, 512], [(48, 32), (24, 16), (12, 8)])]
    for thr in (1, 2, 4, 8, 16, 32):
        os.environ["MCQUIC_AMD_RANS_THREADS"] = str(thr)
        b, cs = coder.compress(codes)
        coder.decompress(b, cs)
        t = time.perf_counter()
        for _ in range(30):
            b, cs = coder.compress(codes)
        enc = (time.perf_counter() - t) / 30 * 1e3
        t = time.perf_counter()
        for _ in range(30):
            coder.decompress(b, cs)
        dec = (time.perf_counter() - t) / 30 * 1e3
        print(f"n={n:3d} threads={thr:2d}  compress {enc:6.3f} ms  decompress {dec:6.3f} ms")
