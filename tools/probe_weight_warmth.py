import sys, os, torch
sys.path.insert(0, ".")
from mcquic_amd import ops
dev = torch.device("cuda:0")
def run(n, h, w, nweights=40, mode="cold"):
    x = torch.randn(n, 128, h, w, device=dev)
    res = torch.randn(n, 128, h, w, device=dev)
    packs = [ops.PackedConv(torch.randn(128, 128, 3, 3, device=dev) * 0.03, torch.randn(128, device=dev)) for _ in range(nweights)]
    iters = 40
    for i in range(3): ops.conv2d(x, packs[i], res=res, dual_silu=True)
    torch.cuda.synchronize()
    tot = 0.0
    evs = []
    for i in range(iters):
        pk = packs[i % nweights]
        if mode == "touch":
            pk.wp.sum()                      # reads the packed weights through some XCDs' L2 -> MALL
        elif mode == "self":
            ops.conv2d(x, pk, res=res, dual_silu=True)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); ops.conv2d(x, pk, res=res, dual_silu=True); e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    us = sorted(s.elapsed_time(e) * 1e3 for s, e in evs)
    return us[len(us) // 2]
for (n, h, w) in [(32, 12, 8), (32, 24, 16), (8, 16, 16), (8, 8, 8), (8, 4, 4), (1, 12, 8), (1, 24, 16)]:
    print(f"{n}x128 {h}x{w}: cold {run(n,h,w,40,'cold'):6.1f} us | after a reduction over the weights {run(n,h,w,40,'touch'):6.1f} | right after the same conv {run(n,h,w,40,'self'):6.1f}")
