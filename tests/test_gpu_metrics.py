"""GPU: validation metrics (csrc/metrics.hip) through the C-ABI against the CPU oracle (oracle/metrics_ref.py) and the
golden values captured from the reference's validate code (tests/golden/f7_metrics.npz).

Tolerances: MS-SSIM is float32 arithmetic; the kernel follows the oracle's operation order up to the final map means
(float64 tree on the device, float32 pairwise sums in torch), so values agree to a few 1e-7 -- asserted at 2e-6 against
the oracle and 5e-6 against the reference (whose library convolutions add taps in a different order).  The squared-error
sums behind PSNR are exact integers; the dB value goes through the device's float64 log10 (1e-13)."""
import os

import numpy as np
import pytest
import torch

from oracle import metrics_ref as M

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_ms_ssim_matches_oracle_and_reference(dev):
    from mcquic_amd import ops, validate
    z = np.load(os.path.join(G, "f7_metrics.npz"))
    for i, (seed, n, h, w) in enumerate(z["cases"].tolist()):
        x, y = M.make_u8_pair(seed, n, h, w)
        got = ops.ms_ssim(x.to(dev), y.to(dev)).cpu()
        np.testing.assert_allclose(got.numpy(), M.ms_ssim(x, y).numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(got.numpy(), z[f"msssim_{i}"], rtol=0, atol=5e-6)
        db = validate.ms_ssim_db(x.to(dev), y.to(dev)).cpu()
        np.testing.assert_allclose(db.numpy(), z[f"msssim_db_{i}"], rtol=0, atol=3e-3)
        p = validate.psnr(x.to(dev), y.to(dev)).cpu()
        np.testing.assert_allclose(p.numpy(), z[f"psnr_{i}"], rtol=1e-13, atol=0)      # exact MSE; log10 on the device


@pytest.mark.parametrize("shape", [(1, 3, 161, 161), (2, 1, 175, 389), (1, 4, 322, 163), (5, 3, 224, 256)])
def test_ms_ssim_ragged_shapes(dev, shape):
    """Odd sides at every pyramid level (zero-padded pooling), single-tile and multi-tile levels, C != 3."""
    from mcquic_amd import ops
    n, c, h, w = shape
    x3, y3 = M.make_u8_pair(h * 1000 + w, n * ((c + 2) // 3), h, w)
    x = x3.reshape(-1, h, w)[: n * c].reshape(n, c, h, w).contiguous()
    y = y3.reshape(-1, h, w)[: n * c].reshape(n, c, h, w).contiguous()
    # the oracle's level weights / channel mean are channel-count agnostic
    got = ops.ms_ssim(x.to(dev), y.to(dev)).cpu()
    np.testing.assert_allclose(got.numpy(), M.ms_ssim(x, y).numpy(), rtol=0, atol=2e-6)


def test_ms_ssim_properties(dev):
    from mcquic_amd import ops
    x, y = M.make_u8_pair(11, 3, 200, 300)
    xd, yd = x.to(dev), y.to(dev)
    a = ops.ms_ssim(xd, yd)
    assert torch.equal(ops.ms_ssim(yd, xd), a)                            # symmetric, bit for bit
    assert torch.equal(ops.ms_ssim(xd, xd).cpu(), torch.ones(3))          # identical images -> exactly 1
    assert torch.equal(ops.ms_ssim(xd[1:2], yd[1:2]), a[1:2])             # batch-invariant
    assert torch.equal(ops.ms_ssim(xd, yd), a)                            # deterministic
    worse = ops.ms_ssim(xd, (255 - yd))
    assert (worse < a).all() and (worse >= 0).all()
    with pytest.raises(ValueError):
        ops.ms_ssim(xd[..., :160, :], yd[..., :160, :])
    with pytest.raises(TypeError):
        ops.ms_ssim(xd.float(), yd.float())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.ms_ssim(x, y)


def test_sqdiff_sum_is_exact(dev):
    from mcquic_amd import ops
    g = torch.Generator().manual_seed(3)
    for shape in ((1, 3, 7, 5), (4, 3, 200, 331), (2, 3, 768, 512)):
        x = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
        y = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
        want = ((x.long() - y.long()) ** 2).flatten(1).sum(1)
        assert torch.equal(ops.sqdiff_sum(x.to(dev), y.to(dev)).cpu(), want)
    x = torch.zeros(1, 3, 300, 300, dtype=torch.uint8)
    assert torch.equal(ops.sqdiff_sum(x.to(dev), (x + 255).to(dev)).cpu(), torch.tensor([255 * 255 * 3 * 300 * 300]))


def test_ideal_bpp_matches_reference(dev):
    """handlers.py IdealBPP over two accumulated batches; histograms from parallel.code_histograms on the device."""
    from mcquic_amd import parallel, validate
    z = np.load(os.path.join(G, "f7_metrics.npz"))
    ks, batches = list(M.CODE_BATCH_KS), M.make_code_batches()
    acc = None
    for codes in batches:
        h = parallel.code_histograms([c.to(dev) for c in codes], ks)
        acc = h if acc is None else [a + b for a, b in zip(acc, h)]
    got = validate.ideal_bpp(acc, 2 * 3 * 768 * 512)
    assert abs(got - float(z["ideal_bpp"][0])) <= 1e-6 * float(z["ideal_bpp"][0])


def test_ms_ssim_full_size_batch(dev):
    """BASELINE batch geometry (32 x 3 x 768 x 512): two images against the oracle, the rest through invariances."""
    from mcquic_amd import ops
    x2, y2 = M.make_u8_pair(21, 2, 768, 512)
    g = torch.Generator().manual_seed(8)
    x = torch.randint(0, 256, (32, 3, 768, 512), generator=g, dtype=torch.uint8)
    x[:2] = x2
    noise = torch.randint(-9, 10, x.shape, generator=g)
    y = (x.long() + noise).clamp(0, 255).to(torch.uint8)
    y[:2] = y2
    xd, yd = x.to(dev), y.to(dev)
    got = ops.ms_ssim(xd, yd)
    np.testing.assert_allclose(got[:2].cpu().numpy(), M.ms_ssim(x2, y2).numpy(), rtol=0, atol=2e-6)
    assert torch.equal(ops.ms_ssim(yd, xd), got)
    assert torch.equal(ops.ms_ssim(xd[7:9], yd[7:9]), got[7:9])
    assert ((got > 0) & (got < 1)).all()


def test_metrics_random_shapes(dev):
    """20 seeded random (batch, height, width) image pairs from the smallest size the five-level pyramid admits (161) up: MS-SSIM
    within 2e-6 of the oracle, the squared-error sums behind PSNR exact."""
    import random
    from mcquic_amd import ops
    rng = random.Random(19)
    for it in range(20):
        n = rng.randint(1, 4)
        h, w = rng.choice([161, 162, 176, 255, 256, 257, rng.randint(161, 600)]), rng.choice([161, 163, 192, 320, 321, rng.randint(161, 600)])
        x, y = M.make_u8_pair(700 + it, n, h, w)
        got = ops.ms_ssim(x.to(dev), y.to(dev)).cpu()
        np.testing.assert_allclose(got.numpy(), M.ms_ssim(x, y).numpy(), rtol=0, atol=2e-6, err_msg=f"#{it} n{n} {h}x{w}")
        sq = ops.sqdiff_sum(x.to(dev), y.to(dev)).cpu()
        want = ((x.to(torch.int64) - y.to(torch.int64)) ** 2).flatten(1).sum(1)
        assert torch.equal(sq, want), f"#{it} n{n} {h}x{w}"
