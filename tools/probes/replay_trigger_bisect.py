import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mcquic_amd import Compressor, parallel, ops
dev = torch.device("cuda:0")
torch.manual_seed(3407)
model = Compressor(128, 2, [8192, 2048, 512]).to(dev).train()
x = (torch.rand((8, 3, 256, 256), generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
what = sys.argv[1]
step = parallel.GraphedTrainStep(model, torch.optim.SGD(model.parameters(), lr=1e-6), x)
out = []
for i in range(6):
    loss = step(x)
    if "S" in what:
        torch.cuda.synchronize()
    if "L" in what:
        out.append(float(loss))
    if "F" in what:
        out.append(bool(torch.isfinite(step.flat).all()))
    if "P" in what:
        out.append(sum(int(not torch.isfinite(p).all()) for p in model.parameters()))
    if "G" in what:
        out.append(sum(int(not torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None))
    if "1" in what:          # read every parameter, no host sync
        for p in model.parameters():
            torch.isfinite(p.detach())
    if "2" in what:          # 666 host syncs on an unrelated tensor
        t = torch.ones(4, device=dev)
        for _ in range(666):
            bool(t.all())
    if "3" in what:          # the same allocations without touching the parameters
        for p in model.parameters():
            torch.empty(p.numel(), dtype=torch.bool, device=dev).fill_(True)
    if "4" in what:          # one host sync + sleep
        torch.cuda.synchronize(); import time; time.sleep(0.5)
    if "R" in what:
        out.append(int(ops._rng_states[0][1]))
torch.cuda.synchronize()
print(what, out, "final loss", float(step(x)), "flat finite", bool(torch.isfinite(step.flat).all()))
