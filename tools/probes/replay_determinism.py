"""Replays of a captured forward + backward over fixed inputs must be bit-identical.  Between replays the host synchronises
`nsync` times (the trigger tools/probes/nan_hunt4.py isolated).  Part A: this package's training graph (main graph only, fixed
uniforms, no update); part B: a torch-only CNN of similar launch count -- is the defect ours or the runtime's?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = torch.device("cuda:0")
part = sys.argv[1]
nsync = int(sys.argv[2]) if len(sys.argv) > 2 else 666
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8


def trigger():
    for _ in range(nsync):
        torch.cuda.synchronize()


if part == "A":
    from mcquic_amd import Compressor, parallel
    torch.manual_seed(3407)
    ks = [8192, 2048, 512]
    model = Compressor(128, 2, ks).to(dev).train()
    n, hw = 8, 256
    x = (torch.rand((n, 3, hw, hw), generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
    g = torch.Generator().manual_seed(5)
    us = []
    for lv, k in enumerate(ks):
        s = hw // 16 // (2 ** lv)
        us.append((torch.rand((n, 2, s, s, k), generator=g).to(dev), torch.rand((n, 2, s, s, k), generator=g).to(dev)))
    step = parallel.GraphedTrainStep(model, torch.optim.SGD(model.parameters(), lr=0.0), x, forward_kwargs={"uniforms": us}, capture_post=False)
    names = [nm for nm, p in model.named_parameters() if p.requires_grad and p.grad is not None]
    graph, flat, params = step.graphs[0], step.flat, step.live
else:
    torch.manual_seed(1)
    layers = []
    for i in range(60):
        layers += [torch.nn.Conv2d(64 if i else 3, 64, 3, padding=1), torch.nn.GroupNorm(8, 64), torch.nn.SiLU()]
    model = torch.nn.Sequential(*layers).to(dev)
    x = torch.randn((8, 3, 64, 64), device=dev)
    params = list(model.parameters())
    names = [nm for nm, _ in model.named_parameters()]
    flat = torch.empty(sum(p.numel() for p in params), device=dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            for p in params:
                p.grad = None
            model(x).square().mean().backward()
    torch.cuda.current_stream().wait_stream(s)
    for p in params:
        p.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        model(x).square().mean().backward()
        torch.cat([p.grad.reshape(-1) for p in params], out=flat)

first = None
for i in range(reps):
    flat.zero_()
    graph.replay()
    trigger()
    torch.cuda.synchronize()
    if first is None:
        first = flat.clone()
        print(part, i, "reference replay, finite", bool(torch.isfinite(first).all()), flush=True)
        continue
    off, bad = 0, []
    for nm, p in zip(names, params):
        a, b = flat[off: off + p.numel()], first[off: off + p.numel()]
        err = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
        if not err <= 1e-3:                                   # (atomic accumulation orders differ between replays: ~1e-6)
            bad.append((nm, err))
        off += p.numel()
    print(part, i, "parameters whose gradient differs from replay 0 by > 1e-3 of its largest entry:", len(bad), bad[:2], bad[-2:], flush=True)
