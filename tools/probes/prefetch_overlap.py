"""Which copy of the host-buffer pipeline fails to hide behind the kernels (bench.py secondary.host_buffers): batch 32 x 768x512,
encode + decode; (a) resident, (b) H2D through parallel.prefetch only, (c) D2H on a side stream only, (d) both, (e) both unpipelined.
    python tools/probes/prefetch_overlap.py          (on the GPU box)
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from mcquic_amd import Compressor, parallel

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = Compressor(128, 2, [8192, 2048, 512]).eval().to(dev)
x = (torch.rand((32, 3, 768, 512)) * 2 - 1).to(dev)
xh = torch.empty(x.shape, pin_memory=True).copy_(x)
yhs = [torch.empty(x.shape, pin_memory=True) for _ in range(2)]
out_stream = torch.cuda.Stream(dev)


def run(n, h2d, d2h, piped=True, keep=False, side_out=True):
    main = torch.cuda.current_stream(dev)
    freed = [None, None]
    alive = [None, None]
    src = parallel.prefetch([xh] * n, dev) if (h2d and piped) else ([xh] * n if h2d else [x] * n)
    for i, xd in enumerate(src):
        if h2d and not piped:
            xd = xd.to(dev, non_blocking=True)
        y = model.decode(model.encode(xd))
        if not d2h:
            continue
        if not piped or not side_out:
            yhs[0].copy_(y, non_blocking=True)
            continue
        ready = torch.cuda.Event()
        ready.record(main)
        if freed[i % 2] is not None:
            freed[i % 2].synchronize()
        with torch.cuda.stream(out_stream):
            out_stream.wait_event(ready)
            yhs[i % 2].copy_(y, non_blocking=True)
            if keep:
                alive[i % 2] = y                             # (held until this slot's copy is known to be over: no record_stream)
            else:
                y.record_stream(out_stream)
            freed[i % 2] = torch.cuda.Event()
            freed[i % 2].record(out_stream)


for name, kw in (("resident", dict(h2d=False, d2h=False)), ("h2d prefetched", dict(h2d=True, d2h=False)),
                 ("d2h side stream", dict(h2d=False, d2h=True)), ("both piped", dict(h2d=True, d2h=True)),
                 ("both serial", dict(h2d=True, d2h=True, piped=False)),
                 ("h2d pref, d2h main", dict(h2d=True, d2h=True, side_out=False)),
                 ("d2h side, kept", dict(h2d=False, d2h=True, keep=True)), ("both piped, kept", dict(h2d=True, d2h=True, keep=True))):
    run(3, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(8, **kw)
    torch.cuda.synchronize()
    print(f"{name:18s} {(time.perf_counter() - t0) / 8 * 1e3:8.3f} ms per batch")
