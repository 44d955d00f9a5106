"""CPU: the row-band geometry of ops._conv2d_banded (layers beyond the kernels' 2 GiB-per-image addressing, reference:
mcquic/modules/compressor.py:67-117 has no size limit) against torch's own convolution -- every band's kept rows must be the
unsplit convolution's rows bit for bit, for 3x3 / 1x1, stride 1 / 2, odd and even heights, any band height."""
import pytest
import torch
import torch.nn.functional as F

from mcquic_amd import ops


@pytest.mark.parametrize("h", [1, 2, 5, 8, 17, 40])
@pytest.mark.parametrize("ksize,stride", [(3, 1), (3, 2), (1, 1)])
@pytest.mark.parametrize("rows", [1, 2, 3, 7, 64])
def test_band_plan_reproduces_the_unsplit_rows(h, ksize, stride, rows):
    g = torch.Generator().manual_seed(h * 100 + rows)
    x = torch.randn((2, 3, h, 6), generator=g)
    wt = torch.randn((4, 3, ksize, ksize), generator=g)
    pad = ksize // 2
    whole = F.conv2d(x, wt, None, stride=stride, padding=pad)
    ho = whole.shape[2]
    out = torch.full_like(whole, float("nan"))
    plan = ops._band_plan(h, ksize, stride, rows)
    assert plan[0][0] == 0 and plan[-1][1] == ho and all(a[1] == b[0] for a, b in zip(plan, plan[1:]))     # a partition of the output rows
    for o0, o1, b0, b1, g0 in plan:
        assert 0 <= b0 < b1 <= h and b0 % stride == 0 and o1 - o0 <= rows
        local = F.conv2d(x[:, :, b0:b1], wt, None, stride=stride, padding=pad)       # the band's own conv (zero padding at ITS edges)
        assert g0 + local.shape[2] <= ho + 1                                           # (at most the one row past a clipped bottom edge)
        out[:, :, o0:o1] = local[:, :, o0 - g0:o1 - g0]
    assert torch.equal(out, whole)


def test_band_rows_only_past_the_limit():
    class W:
        cout, cin, ksize = 128, 128, 3
    x = torch.empty((1, 128, 64, 64))
    assert ops._band_rows(x, W, 1) == 0                        # 2.6 MB: one launch
    prev = ops.set_slab_limit(1 << 20)
    try:
        r = ops._band_rows(x, W, 1)
        assert 0 < r < 64 and (128 + 32) * (r + 2) * 64 * 4 < (1 << 20)      # a band with its halo fits the limit
    finally:
        ops.set_slab_limit(prev)
    assert ops.set_slab_limit(None) == prev
