"""GPU: the HIP path against the golden vectors captured from the real reference (tests/golden/*.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle import mcquic_ref as R

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_blocks_against_reference_vectors(dev):
    from mcquic_amd import nn as N
    z = np.load(os.path.join(G, "f1_blocks_c8.npz"))
    c = 8
    x = torch.from_numpy(z["x"]).to(dev)
    for name, ctor, mk in [("ResidualBlock", lambda: N.ResidualBlock(c, c), R._rb),
                           ("ResidualBlockWithStride", lambda: N.ResidualBlockWithStride(c, c), R._rb_stride),
                           ("ResidualBlockShuffle", lambda: N.ResidualBlockShuffle(c, c), R._rb_shuffle),
                           ("AttentionBlock", lambda: N.AttentionBlock(c), R._attn)]:
        sd = {}
        mk(sd, "", c, 11)
        mod = ctor()
        mod.load_state_dict(sd, strict=True)
        got = mod.to(dev).eval()(x).cpu().numpy()
        np.testing.assert_allclose(got, z[name], rtol=0, atol=5e-6, err_msg=name)
    for name, cls in [("GenDivNorm", N.GenDivNorm), ("InvGenDivNorm", N.InvGenDivNorm)]:
        sd = {}
        R._gdn_params(sd, "", c, 12)
        mod = cls(c)
        mod.load_state_dict(sd, strict=True)
        np.testing.assert_allclose(mod.to(dev).eval()(x * 2).cpu().numpy(), z[name], rtol=0, atol=5e-6, err_msg=name)


def test_vq_against_reference_vectors(dev):
    from mcquic_amd import ops
    z = np.load(os.path.join(G, "f2_vq.npz"))
    for tag in ("qp2_l2", "qp2_l1", "small"):
        m, k, d, n, h, w = [int(v) for v in z[tag + "_shape"]]
        g = torch.Generator().manual_seed(7)
        cb = torch.randn((m, k, d), generator=g) * np.sqrt(2 / (5 * d))
        x = torch.randn((n, m * d, h, w), generator=g) * 0.1
        code = ops.vq_assign(x.to(dev), ops.PackedCodebook(cb.to(dev))).cpu()
        want = torch.from_numpy(z[tag + "_code"].astype(np.int64))
        bad = (code != want).numpy()
        assert (z[tag + "_gap"][bad] < 1e-5).all(), f"{tag}: {bad.sum()} mismatches away from near-ties"


@pytest.mark.parametrize("tag", ["pad", "aligned"])
def test_small_model_against_reference_vectors(dev, tag):
    from mcquic_amd import Compressor
    z = np.load(os.path.join(G, "f4_small_model.npz"))
    n, h, w = [int(v) for v in z[tag + "_shape"]]
    sd = R.make_state_dict(8, 2, [32, 16, 8], seed=1)
    model = Compressor(8, 2, [32, 16, 8]).eval()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    codes = model.encode(R.make_images(n, h, w).to(dev))
    want = [torch.from_numpy(z[f"{tag}_code{lv}"].astype(np.int64)) for lv in range(3)]
    for lv, (c, wc) in enumerate(zip(codes, want)):
        assert torch.equal(c.cpu(), wc), f"level {lv}: {(c.cpu() != wc).sum()} mismatches"
    rec = model.decode([c.to(dev) for c in want]).cpu()
    np.testing.assert_allclose(rec[..., ::4, ::4].numpy(), z[tag + "_rec_strided"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(rec[..., 32:96, 32:96].numpy(), z[tag + "_rec_crop"], rtol=0, atol=1e-4)


def test_qp2_model_against_reference_vectors(dev):
    from mcquic_amd import Compressor
    z = np.load(os.path.join(G, "f5_qp2_model.npz"))
    sd = R.make_state_dict(128, 2, [8192, 2048, 512], seed=0)
    model = Compressor(128, 2, [8192, 2048, 512]).eval()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    codes = model.encode(R.make_images(1, 256, 384).to(dev))
    want = [torch.from_numpy(z[f"code{lv}"].astype(np.int64)) for lv in range(3)]
    for lv, (c, wc) in enumerate(zip(codes, want)):
        assert torch.equal(c.cpu(), wc), f"level {lv}: {(c.cpu() != wc).sum()} mismatches"
    rec = model.decode([c.to(dev) for c in want]).cpu()
    np.testing.assert_allclose(rec[:, :, 96:160, 160:224].numpy(), z["rec_crop"], rtol=0, atol=1e-4)
    assert abs(rec.abs().mean().item() - float(z["rec_mean_abs"][0])) < 1e-5


def test_blocks_c128_against_reference_vectors(dev):
    """F1 at the network's width: every block type at C = 128 against the reference's outputs."""
    from mcquic_amd import nn as N
    z = np.load(os.path.join(G, "f1b_blocks_c128.npz"))
    c = 128
    x = torch.from_numpy(z["x"]).to(dev)
    for name, ctor, mk in [("ResidualBlock", lambda: N.ResidualBlock(c, c), R._rb),
                           ("ResidualBlockWithStride", lambda: N.ResidualBlockWithStride(c, c), R._rb_stride),
                           ("ResidualBlockShuffle", lambda: N.ResidualBlockShuffle(c, c), R._rb_shuffle),
                           ("AttentionBlock", lambda: N.AttentionBlock(c), R._attn)]:
        sd = {}
        mk(sd, "", c, 21)
        mod = ctor()
        mod.load_state_dict(sd, strict=True)
        got = mod.to(dev).eval()(x).cpu().numpy()
        np.testing.assert_allclose(got, z[name], rtol=0, atol=2e-5, err_msg=name)
    for name, cls in [("GenDivNorm", N.GenDivNorm), ("InvGenDivNorm", N.InvGenDivNorm)]:
        sd = {}
        R._gdn_params(sd, "", c, 22)
        mod = cls(c)
        mod.load_state_dict(sd, strict=True)
        np.testing.assert_allclose(mod.to(dev).eval()(x * 2).cpu().numpy(), z[name], rtol=0, atol=2e-5, err_msg=name)


def _audit_codes(got, want, gaps, tie=1e-5):
    """Index parity with the near-tie protocol of DESIGN section 6: a code may differ from the reference's only where the
    reference's OWN top-2 distance gap at that vector is below `tie` (fp32 reassociation noise is ~1e-6 of distances of
    order 1: which of two codewords that close wins depends on the summation order of the conv stack, here as between
    two CPU BLAS builds).  A flipped code changes the residual the deeper levels quantize, so an image is only compared
    down to its first excused flip.  Returns (flips, images cut short)."""
    n = want[0].shape[0]
    alive = torch.ones(n, dtype=torch.bool)
    flips = 0
    for lv, (g, w_, gap) in enumerate(zip(got, want, gaps)):
        bad = (g != w_) & alive[:, None, None, None]
        if bad.any():
            worst = float(torch.from_numpy(gap)[bad].max())
            assert worst < tie, f"level {lv}: {int(bad.sum())} code mismatches, reference gap there up to {worst:.3e}"
            flips += int(bad.sum())
            alive &= ~bad.flatten(1).any(1)
    return flips, int((~alive).sum())


@pytest.mark.parametrize("tag", ["kodak", "sample"])
def test_qp2_fullsize_against_reference_vectors(dev, tag):
    """F5 at BASELINE's sizes, straight against the reference (no oracle in between): 2 x 3 x 768 x 512 (configs[1]'s
    geometry) and 1 x 3 x 1152 x 2048 (configs[0]: assets/sample.png's geometry) -- code indices exact up to audited
    near-ties (the fixture carries the reference's own top-2 gaps), pixels from the reference's codes within 1e-4."""
    from mcquic_amd import Compressor
    z = np.load(os.path.join(G, "f5b_qp2_fullsize.npz"))
    n, h, w, seed = [int(v) for v in z[tag + "_shape"]]
    sd = R.make_state_dict(128, 2, [8192, 2048, 512], seed=0)
    model = Compressor(128, 2, [8192, 2048, 512]).eval()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    codes = [c.cpu() for c in model.encode(R.make_images(n, h, w, seed=seed).to(dev))]
    want = [torch.from_numpy(z[f"{tag}_code{lv}"].astype(np.int64)) for lv in range(3)]
    flips, cut = _audit_codes(codes, want, [z[f"{tag}_gap{lv}"] for lv in range(3)])
    total = sum(c.numel() for c in want)
    assert flips <= max(1, total // 20000), f"{flips} audited near-tie flips in {total} codes: more than fp32 noise explains"
    rec = model.decode([c.to(dev) for c in want]).cpu()
    np.testing.assert_allclose(rec[..., ::16, ::16].numpy(), z[tag + "_rec_strided"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(rec[:, :, h // 2 - 32:h // 2 + 32, w // 2 - 32:w // 2 + 32].numpy(), z[tag + "_rec_crop"], rtol=0, atol=1e-4)
    assert abs(rec.abs().mean().item() - float(z[tag + "_rec_mean_abs"][0])) < 1e-5
