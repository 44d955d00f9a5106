"""CPU: the plain-C restatement (oracle/mcq_oracle.c) agrees with the PyTorch restatement (oracle/mcquic_ref.py),
which is the one pinned against the real reference."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import c_oracle
from oracle import mcquic_ref as R


def test_c_conv_matches_torch():
    g = torch.Generator().manual_seed(0)
    for (n, cin, cout, h, w, ks, stride) in [(1, 8, 8, 9, 7, 3, 1), (2, 3, 16, 12, 10, 3, 2), (1, 16, 8, 6, 5, 1, 1)]:
        x = torch.rand((n, cin, h, w), generator=g) * 2 - 1
        wt = (torch.rand((cout, cin, ks, ks), generator=g) * 2 - 1) / np.sqrt(cin * ks * ks)
        b = torch.rand((cout,), generator=g) * 0.1
        want = F.conv2d(x, wt, b, stride=stride, padding=ks // 2).numpy()
        np.testing.assert_allclose(c_oracle.conv2d(x.numpy(), wt.numpy(), b.numpy(), stride), want, rtol=0, atol=2e-6)
        np.testing.assert_allclose(c_oracle.conv2d(x.numpy(), wt.numpy(), b.numpy(), stride, wide=True), want, rtol=0, atol=2e-6)


def test_c_vq_matches_torch_with_tie_audit():
    g = torch.Generator().manual_seed(1)
    m, k, d, n, h, w = 2, 512, 64, 2, 5, 6
    cb = torch.randn((m, k, d), generator=g) * np.sqrt(2 / (5 * d))
    x = torch.randn((n, m * d, h, w), generator=g) * 0.1
    codes, gap = c_oracle.vq_assign(x.numpy(), cb.numpy())
    want = R.vq_encode(x, cb).numpy()
    bad = codes != want
    assert (gap[bad] < 1e-6).all()
    assert np.array_equal(c_oracle.vq_gather(want, cb.numpy()), R.vq_decode(torch.from_numpy(want), cb).numpy())
