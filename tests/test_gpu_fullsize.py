"""GPU, BASELINE's full sizes (batch 32 of 768x512, qp=2 model): size-independent properties, since the CPU oracle
would take minutes here.  (The oracle itself is compared at full image size on two images by bench.py's parity leg.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qp2(dev):
    from mcquic_amd import Compressor
    torch.manual_seed(3407)
    return Compressor(128, 2, [8192, 2048, 512]).eval().to(dev)


def test_full_batch_encode_decode_properties(dev, qp2):
    g = torch.Generator().manual_seed(3407)
    x = (torch.rand((32, 3, 768, 512), generator=g) * 2 - 1).to(dev)
    codes = qp2.encode(x)
    assert [tuple(c.shape) for c in codes] == [(32, 2, 48, 32), (32, 2, 24, 16), (32, 2, 12, 8)]
    for c, k in zip(codes, (8192, 2048, 512)):
        assert c.dtype == torch.int64 and int(c.min()) >= 0 and int(c.max()) < k
    # images are independent: a slice of the batch encodes to the same codes (different wave tiles / split-K choices)
    for sl in (slice(5, 6), slice(16, 20)):
        part = qp2.encode(x[sl])
        for a, b in zip(codes, part):
            assert torch.equal(a[sl], b)
    rec = qp2.decode(codes)
    assert tuple(rec.shape) == (32, 3, 768, 512) and torch.isfinite(rec).all()
    part = qp2.decode([c[7:9] for c in codes])
    assert (rec[7:9] - part).abs().max().item() < 1e-5
    # run-to-run determinism (no atomics anywhere on the path)
    assert all(torch.equal(a, b) for a, b in zip(codes, qp2.encode(x)))
    assert torch.equal(rec, qp2.decode(codes))


def test_conv_is_exactly_linear_under_power_of_two_scaling(dev):
    """conv(2 x) == 2 conv(x) bit for bit when there is no bias: every product and partial sum just gains an exponent."""
    from mcquic_amd import ops
    g = torch.Generator().manual_seed(1)
    x = (torch.rand((32, 128, 192, 128), generator=g) * 2 - 1).to(dev)
    w = ((torch.rand((128, 128, 3, 3), generator=g) * 2 - 1) / 34.0).to(dev)
    pk = ops.PackedConv(w, None)
    y1 = ops.conv2d(x, pk)
    y2 = ops.conv2d(x * 2.0, pk)
    assert torch.equal(y2, y1 * 2.0)
    # and additive over a split of the input channels (exact regrouping is not guaranteed; tolerance of a few ulp)
    xa, xb = x.clone(), x.clone()
    xa[:, 64:] = 0
    xb[:, :64] = 0
    err = (ops.conv2d(xa, pk) + ops.conv2d(xb, pk) - y1).abs().max().item()
    assert err < 1e-5


def test_vq_codes_are_the_nearest_codewords_at_full_size(dev):
    """At config-#4 size: the returned code minimises the distance recomputed by the gather + a plain reduction."""
    from mcquic_amd import ops
    m, k, d, n, h, w = 4, 4096, 256, 32, 48, 32
    g = torch.Generator().manual_seed(0)
    x = (torch.randn((n, m * d, h, w), generator=g) * 0.1).to(dev)
    cb = ops.PackedCodebook((torch.randn((m, k, d), generator=g) * (2 / (5 * d)) ** 0.5).to(dev))
    codes = ops.vq_assign(x, cb)
    chosen = ops.vq_gather(codes, cb)                                     # [n, m*d, h, w]
    dmin = ((x - chosen) ** 2).reshape(n, m, d, h, w).sum(2)              # distance to the chosen codeword
    # compare with a handful of random other codewords per vector: none may be closer (beyond rounding)
    for trial in range(4):
        other = torch.randint(0, k, codes.shape, generator=torch.Generator().manual_seed(trial)).to(dev)
        dother = ((x - ops.vq_gather(other, cb)) ** 2).reshape(n, m, d, h, w).sum(2)
        assert bool((dmin <= dother + 1e-5).all())
    # and exhaustively for one image-group against torch's own argmin of the reference formula
    xv = x[3, :d].reshape(d, -1).t()                                      # group 0 of image 3: [hw, d]
    c0 = cb.codebook[0]
    dist = (xv ** 2).sum(1, keepdim=True) + (c0 ** 2).sum(1)[None] - 2 * xv @ c0.t()
    ref = dist.argmin(1).reshape(h, w)
    bad = ref != codes[3, 0]
    if bad.any():
        top2 = torch.topk(dist, 2, dim=1, largest=False).values
        gap = (top2[:, 1] - top2[:, 0]).reshape(h, w)[bad]
        assert gap.max().item() < 1e-5
