"""Does a training step of the Neon family survive hipGraph capture?  (forward only / forward + backward)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MCQUIC_AMD_BRANCH_STREAMS"] = "0"
import torch
from mcquic_amd import Neon
dev = torch.device("cuda:0")
mode = sys.argv[1]
torch.manual_seed(1)
m = Neon(32, 256, [8, 4, 2, 2], False).to(dev).train()
x = (torch.rand((2, 3, 128, 128)) * 2 - 1).to(dev)
def fb(backward):
    for p in m.parameters(): p.grad = None
    out = m(x)
    loss = torch.nn.functional.mse_loss(out[0], x)
    if backward: loss.backward()
    return loss
for _ in range(2): fb(True)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    l = fb(mode == "fb")
g.replay(); torch.cuda.synchronize()
print(mode, "captured and replayed, loss", float(l))
