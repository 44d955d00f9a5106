"""CPU: the oracle (oracle/mcquic_ref.py) against golden vectors captured from the real reference
(tests/golden/make_golden.py).  Floats are compared at 1e-6 (a different host may pick different oneDNN kernels);
indices are exact except where the reference's own recorded top-2 gap is below 1e-5."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import mcquic_ref as R

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _sha(t):
    return np.frombuffer(hashlib.sha256(t.contiguous().numpy().tobytes()).digest(), dtype=np.uint8)


def test_blocks_match_reference():
    z = np.load(os.path.join(G, "f1_blocks_c8.npz"))
    c = 8
    x = torch.from_numpy(z["x"])
    assert torch.equal(x, _rand((2, c, 13, 18), 101))
    for name, mk, fn in [("ResidualBlock", R._rb, R.residual_block),
                         ("ResidualBlockWithStride", R._rb_stride, R.residual_block_with_stride),
                         ("ResidualBlockShuffle", R._rb_shuffle, R.residual_block_shuffle),
                         ("AttentionBlock", R._attn, R.attention_block)]:
        sd = {}
        mk(sd, "", c, 11)
        got = fn(sd, "", x.clone())
        np.testing.assert_allclose(got.numpy(), z[name], rtol=0, atol=1e-6, err_msg=name)
    for name, inv in [("GenDivNorm", False), ("InvGenDivNorm", True)]:
        sd = {}
        R._gdn_params(sd, "", c, 12)
        np.testing.assert_allclose(R.gdn(sd, "", x.clone() * 2, inv).numpy(), z[name], rtol=0, atol=1e-6, err_msg=name)


def test_vq_matches_reference():
    z = np.load(os.path.join(G, "f2_vq.npz"))
    for tag in ("qp2_l2", "qp2_l1", "small"):
        m, k, d, n, h, w = [int(v) for v in z[tag + "_shape"]]
        g = torch.Generator().manual_seed(7)
        cb = torch.randn((m, k, d), generator=g) * np.sqrt(2 / (5 * d))
        x = torch.randn((n, m * d, h, w), generator=g) * 0.1
        code = R.vq_encode(x, cb)
        want = torch.from_numpy(z[tag + "_code"].astype(np.int64))
        bad = (code != want).numpy()
        assert (z[tag + "_gap"][bad] < 1e-5).all(), f"{tag}: mismatch away from a near-tie"
        deq = R.vq_decode(want, cb)
        assert (_sha(deq) == z[tag + "_deq_sha"]).all()


@pytest.mark.parametrize("tag", ["config4", "qp2_l0"])
def test_vq_full_size_fixture_oracle_side(tag):
    """F2b (BASELINE configs[3] at its own size, captured from the reference): the oracle's codes for the first two images hash to
    the reference's, and the fixture's inputs are the tensors mcquic_amd.utils.synthetic.vq_case makes."""
    import hashlib
    from mcquic_amd.utils import synthetic as S
    z = np.load(os.path.join(G, "f2b_vq_fullsize.npz"))
    lat, cb = S.vq_case(tag)
    sha2 = hashlib.sha256(lat.contiguous().numpy().tobytes()).digest() + hashlib.sha256(cb.contiguous().numpy().tobytes()).digest()
    assert sha2 == z[tag + "_input_sha"].tobytes()
    code = R.vq_encode(lat[:2], cb)
    for i in range(2):
        assert S.code_hash(code[i]) == z[tag + "_code_hash"][i].tobytes()
    near = z[tag + "_near"]
    assert (z[tag + "_near_gap"] < float(z["near_threshold"][0])).all() and (near[:, 4] != near[:, 5]).all()


@pytest.mark.parametrize("tag", ["pad", "aligned"])
def test_small_model_matches_reference(tag):
    z = np.load(os.path.join(G, "f4_small_model.npz"))
    n, h, w = [int(v) for v in z[tag + "_shape"]]
    sd = R.make_state_dict(8, 2, [32, 16, 8], seed=1)
    x = R.make_images(n, h, w)
    assert (_sha(R.aligned_padding(x)) == z[tag + "_padded_sha"]).all()
    codes = R.encode(sd, x)
    for lv, c in enumerate(codes):
        assert torch.equal(c, torch.from_numpy(z[f"{tag}_code{lv}"].astype(np.int64))), f"level {lv}"
    rec = R.decode(sd, codes)
    np.testing.assert_allclose(rec[..., ::4, ::4].numpy(), z[tag + "_rec_strided"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(rec[..., 32:96, 32:96].numpy(), z[tag + "_rec_crop"], rtol=0, atol=1e-6)


def test_qp2_model_matches_reference():
    z = np.load(os.path.join(G, "f5_qp2_model.npz"))
    sd = R.make_state_dict(128, 2, [8192, 2048, 512], seed=0)
    assert len(sd) == int(z["n_state_dict_entries"][0]) == 718
    x = R.make_images(1, 256, 384)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    codes = R.encode(sd, x)
    for lv, c in enumerate(codes):
        assert torch.equal(c, torch.from_numpy(z[f"code{lv}"].astype(np.int64))), f"level {lv}"
    rec = R.decode(sd, codes)
    np.testing.assert_allclose(rec[:, :, 96:160, 160:224].numpy(), z["rec_crop"], rtol=0, atol=1e-6)
    assert abs(rec.abs().mean().item() - float(z["rec_mean_abs"][0])) < 1e-6


def test_detransform_and_psnr_formulas():
    x = torch.tensor([[-1.0, -0.9999, 0.0, 0.5, 0.99999, 1.0, 1.2, -3.0]]).reshape(1, 1, 2, 4)
    u = R.detransform(x)
    assert u.dtype == torch.uint8
    assert u.flatten().tolist() == [0, 0, 127, 191, 255, 255, 255, 0]
    a = torch.zeros(1, 3, 4, 4, dtype=torch.uint8)
    assert abs(float(R.psnr(a, a)) - 10 * np.log10(255.0 ** 2 / 1e-4)) < 1e-9


def _train_case():
    ch, m, ks = 8, 2, [32, 16, 8]
    sd = R.make_state_dict(ch, m, ks, seed=2)
    g = torch.Generator().manual_seed(3)
    for lv, k in enumerate(ks):
        f = torch.rand((m, k), generator=g) ** 3 + 1e-3
        sd[f"_quantizer._entropyCoder._freqEMA.{lv}"] = f / f.sum(-1, keepdim=True)
    x = R.make_images(2, 128, 128, seed=4)
    shapes = [(2, m, 8, 8, 32), (2, m, 4, 4, 16), (2, m, 2, 2, 8)]
    us = [(torch.rand(sh, generator=g), torch.rand(sh, generator=g)) for sh in shapes]
    return sd, x, us


def test_training_forward_matches_repaired_reference():
    z = np.load(os.path.join(G, "f6_train_forward.npz"))
    sd, x, us = _train_case()
    with torch.no_grad():
        xHat, yHat, codes, logits, counts = R.forward_train(sd, x, us)
    np.testing.assert_allclose(xHat[..., ::2, ::2].numpy(), z["xHat_strided"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(yHat.numpy(), z["yHat"], rtol=0, atol=1e-6)
    for lv in range(3):
        assert torch.equal(codes[lv], torch.from_numpy(z[f"code{lv}"].astype(np.int64)))
        np.testing.assert_allclose(logits[lv].numpy(), z[f"logit{lv}"], rtol=1e-6, atol=1e-6)
        ema = R.freq_ema_update(sd[f"_quantizer._entropyCoder._freqEMA.{lv}"], counts[lv])
        np.testing.assert_allclose(ema.numpy(), z[f"ema{lv}"], rtol=0, atol=1e-7)


def test_metrics_match_reference():
    """MS-SSIM / PSNR / IdealBPP restatements (oracle/metrics_ref.py) against values from the reference's validate code."""
    from oracle import metrics_ref as M
    z = np.load(os.path.join(G, "f7_metrics.npz"))
    for i, (seed, n, h, w) in enumerate(z["cases"].tolist()):
        x, y = M.make_u8_pair(seed, n, h, w)
        assert np.array_equal(np.concatenate([_sha(x), _sha(y)]), z[f"sha_{i}"]), "generator drifted"
        v = M.ms_ssim(x, y)
        np.testing.assert_allclose(v.numpy(), z[f"msssim_{i}"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(M.ms_ssim_db(v).numpy(), z[f"msssim_db_{i}"], rtol=0, atol=2e-3)
        np.testing.assert_allclose(M.psnr_u8(x, y).numpy(), z[f"psnr_{i}"], rtol=1e-14, atol=0)
    ks, batches = list(M.CODE_BATCH_KS), M.make_code_batches()
    hist = [torch.zeros(2, k) for k in ks]
    count = [torch.zeros(2) for _ in ks]
    for codes in batches:
        for lv, (c, k) in enumerate(zip(codes, ks)):
            for g in range(2):
                hist[lv][g] += torch.bincount(c[:, g].flatten(), minlength=k)
                count[lv][g] += c[:, g].numel()
    got = M.ideal_bpp(hist, count, 2 * 3 * 768 * 512)
    assert abs(got - float(z["ideal_bpp"][0])) <= 1e-6 * float(z["ideal_bpp"][0]), (got, z["ideal_bpp"])


def test_ms_ssim_rejects_small_images():
    from oracle import metrics_ref as M
    x = torch.zeros(1, 3, 160, 300, dtype=torch.uint8)
    with pytest.raises(ValueError):
        M.ms_ssim(x, x)


def test_blocks_c128_match_reference():
    """F1 at the network's width (C = 128)."""
    z = np.load(os.path.join(G, "f1b_blocks_c128.npz"))
    c = 128
    x = torch.from_numpy(z["x"])
    for name, mk, fn in [("ResidualBlock", R._rb, R.residual_block),
                         ("ResidualBlockWithStride", R._rb_stride, R.residual_block_with_stride),
                         ("ResidualBlockShuffle", R._rb_shuffle, R.residual_block_shuffle),
                         ("AttentionBlock", R._attn, R.attention_block)]:
        sd = {}
        mk(sd, "", c, 21)
        np.testing.assert_allclose(fn(sd, "", x.clone()).numpy(), z[name], rtol=0, atol=2e-6, err_msg=name)
    for name, inv in [("GenDivNorm", False), ("InvGenDivNorm", True)]:
        sd = {}
        R._gdn_params(sd, "", c, 22)
        np.testing.assert_allclose(R.gdn(sd, "", x.clone() * 2, inv).numpy(), z[name], rtol=0, atol=2e-6, err_msg=name)


def test_qp2_model_kodak_batch_matches_reference():
    """F5 at BASELINE's geometry: 2 x 3 x 768 x 512 through the qp=2 model, every code index against the reference."""
    z = np.load(os.path.join(G, "f5b_qp2_fullsize.npz"))
    n, h, w, seed = [int(v) for v in z["kodak_shape"]]
    sd = R.make_state_dict(128, 2, [8192, 2048, 512], seed=0)
    codes = R.encode(sd, R.make_images(n, h, w, seed=seed))
    want = [torch.from_numpy(z[f"kodak_code{lv}"].astype(np.int64)) for lv in range(3)]
    for lv, (c, wc) in enumerate(zip(codes, want)):
        assert torch.equal(c, wc), f"level {lv}: {(c != wc).sum()} mismatches"
    rec = R.decode(sd, want)
    np.testing.assert_allclose(rec[..., ::16, ::16].numpy(), z["kodak_rec_strided"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(rec[:, :, h // 2 - 32:h // 2 + 32, w // 2 - 32:w // 2 + 32].numpy(), z["kodak_rec_crop"], rtol=0, atol=2e-6)
