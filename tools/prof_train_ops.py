import os, sys
sys.path.insert(0, "/root/repo")
os.environ["MCQUIC_AMD_BRANCH_STREAMS"] = "0"
import torch
from torch.profiler import profile, ProfilerActivity
from mcquic_amd import Compressor
dev = torch.device("cuda:0")
torch.manual_seed(3407)
model = Compressor(128, 2, [8192, 2048, 512]).to(dev).train()
x = (torch.rand((8, 3, 256, 256)) * 2 - 1).to(dev)
def step():
    for p in model.parameters(): p.grad = None
    xHat, yHat, codes, logits = model(x)
    loss = torch.nn.functional.mse_loss(xHat, x)
    loss.backward()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
rows = prof.key_averages()
rows = sorted(rows, key=lambda r: -r.count)
print(f"{'op':60s} {'count':>6} {'cuda_us':>10}")
for r in rows[:70]:
    name = r.key[:60]
    if name.startswith("void") or "kernel" in name.lower() or name.startswith("(anonymous"):
        continue
    print(f"{name:60s} {r.count:6d} {getattr(r, 'device_time_total', getattr(r, 'cuda_time_total', 0)):10.1f}")
