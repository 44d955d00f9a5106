import torch, sys
sys.path.insert(0, '.')
import torch.nn.functional as F
from mcquic_amd import ops
from tests.test_gpu_ops import _ulp_err
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(5)
x = torch.cat([torch.linspace(-30.0, 30.0, 4000001), torch.randn(2000000, generator=g) * 3.0]).float()
want = x.double() * torch.sigmoid(x.double())
got = ops.silu(x.to(dev)).cpu()
e = _ulp_err(got, want); a = _ulp_err(F.silu(x), want); t = _ulp_err(F.silu(x.to(dev)).cpu(), want)
for name, v in (("ours", e), ("aten cpu", a), ("aten gpu", t)):
    print(name, "max %.2f mean %.3f p99.9 %.2f" % (v.max().item(), v.mean().item(), v.quantile(0.999).item() if v.numel() < 16e6 else -1), "argmax x", x[v.argmax()].item())
for lo, hi in ((-30,-10),(-10,-3),(-3,0),(0,3),(3,30)):
    m = (x>=lo)&(x<hi)
    print(lo,hi,"ours max %.2f mean %.3f | aten cpu max %.2f mean %.3f | torch gpu max %.2f" % (e[m].max(), e[m].mean(), a[m].max(), a[m].mean(), t[m].max()))
