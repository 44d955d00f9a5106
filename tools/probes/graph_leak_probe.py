"""Which tensors still carry an autograd graph after a finished training iteration (they keep AccumulateGrad nodes alive, which
breaks hipGraph capture of the next iteration)."""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MCQUIC_AMD_BRANCH_STREAMS"] = "0"
import torch
from mcquic_amd import Neon, Compressor
dev = torch.device("cuda:0")
for name, m, hw in (("Compressor", Compressor(32, 2, [64, 32, 16]), 64), ("Neon", Neon(32, 256, [8, 4, 2, 2], False), 128)):
    m = m.to(dev).train()
    x = (torch.rand((2, 3, hw, hw)) * 2 - 1).to(dev)
    for _ in range(2):
        for p in m.parameters(): p.grad = None
        out = m(x)
        torch.nn.functional.mse_loss(out[0], x).backward()
        del out
    gc.collect()
    alive = [o for o in gc.get_objects() if torch.is_tensor(o) and o.grad_fn is not None]
    print(name, len(alive), [(tuple(o.shape), type(o.grad_fn).__name__) for o in alive[:12]])
    import collections
    for o in alive[:3]:
        refs = [type(r).__name__ + (":" + ",".join(k for k, v in r.items() if v is o)[:80] if isinstance(r, dict) else "") for r in gc.get_referrers(o)][:6]
        print("   held by", refs)
