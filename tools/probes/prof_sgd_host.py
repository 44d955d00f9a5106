import os, sys, cProfile, pstats, io
sys.path.insert(0, "/root/repo")
os.environ["MCQUIC_AMD_BRANCH_STREAMS"] = "0"
import torch
from mcquic_amd import Compressor
dev = torch.device("cuda:0")
torch.manual_seed(3407)
model = Compressor(128, 2, [8192, 2048, 512]).to(dev).train()
x = (torch.rand((8, 3, 256, 256)) * 2 - 1).to(dev)
opt = torch.optim.SGD(model.parameters(), lr=1e-6)
def step():
    opt.zero_grad(set_to_none=True)
    xHat, yHat, codes, logits = model(x)
    loss = torch.nn.functional.mse_loss(xHat, x)
    loss.backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
