"""GPU: training-mode forward (BASELINE config #5, forward half) against the oracle and the golden vectors of the
repaired reference.  Uniform draws are inputs on both sides."""
import os

import numpy as np
import pytest
import torch

from oracle import mcquic_ref as R

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _vq_case(m, k, d, n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    cb = torch.randn((m, k, d), generator=g) * np.sqrt(2 / (5 * d))
    x = torch.randn((n, m * d, h, w), generator=g) * 0.1
    return x, cb, g


@pytest.mark.parametrize("shape", [(2, 8192, 64, 1, 4, 6), (2, 512, 64, 2, 3, 5), (2, 32, 4, 2, 8, 8), (3, 200, 10, 1, 5, 7)])
def test_logits(dev, shape):
    from mcquic_amd import ops
    m, k, d, n, h, w = shape
    x, cb, g = _vq_case(m, k, d, n, h, w, 11)
    temp = torch.rand((m, 1, 1, 1), generator=g) + 0.5
    want = R.vq_logit(x, cb, temp, torch.tensor([R.EPS]))
    got = ops.vq_logits(x.to(dev), ops.PackedCodebook(cb.to(dev)), temp.to(dev), R.EPS).cpu()
    assert got.shape == want.shape
    err = (got - want).abs().max().item()
    assert err <= 2e-6 * max(1.0, want.abs().max().item()), f"logits max abs err {err:.3e}"


@pytest.mark.parametrize("k", [8192, 2048, 512, 200, 32, 8])
def test_logit_division_is_the_ieee_quotient(dev, k):
    """-dist / sqrt(k) is computed as two fused steps around a precomputed reciprocal (csrc/vq_train.hip: div_by_constant); the
    result must be the correctly rounded float32 quotient the reference's `/` gives.  A zero latent against codewords with a single
    non-zero component makes dist = fl(v^2) exactly, so every logit is one known quotient: 16 x k of them per case."""
    from mcquic_amd import ops
    m, d = 16, 4
    g = torch.Generator().manual_seed(k)
    v = (torch.rand((m, k), generator=g) * 2 - 1) * torch.exp(torch.rand((m, k), generator=g) * 12 - 6)      # magnitudes 2.5e-3 .. 400
    cb = torch.zeros((m, k, d))
    cb[:, :, 1] = v
    x = torch.zeros((1, m * d, 1, 1))
    temp = torch.ones((m, 1, 1, 1))
    got = ops.vq_logits(x.to(dev), ops.PackedCodebook(cb.to(dev)), temp.to(dev), R.EPS).cpu().numpy().reshape(m, k)
    dist = (v.numpy() * v.numpy()).astype(np.float32)
    want = (-dist) / np.float32(np.sqrt(np.float64(k)))
    assert want.dtype == np.float32
    assert np.array_equal(got, want), f"{(got != want).sum()} of {got.size} quotients differ"


def test_gumbel_sample_and_soft_dequant(dev):
    from mcquic_amd import ops
    m, k, d, n, h, w = 2, 512, 64, 2, 4, 5
    x, cb, g = _vq_case(m, k, d, n, h, w, 12)
    logit0 = R.vq_logit(x, cb, torch.ones(m, 1, 1, 1), torch.tensor([R.EPS]))
    freq = torch.rand((m, k), generator=g) ** 3 + 1e-3
    freq = freq / freq.sum(-1, keepdim=True)
    u1, u2 = torch.rand(logit0.shape, generator=g), torch.rand(logit0.shape, generator=g)
    want_logit = R.random_drop(logit0, freq, u1)
    want_sample, y_soft, want_index = R.gumbel_softmax_hard(want_logit, u2)
    want_code = want_logit.argmax(-1)
    bits = np.log2(k)
    usage = (freq > R.EPS).float().mean().clamp(0., 1.)
    expo = (-(bits - 1) * (usage ** 2) + bits)
    lg = logit0.clone().to(dev)
    code, index, hot = ops.vq_gumbel_sample(lg, u1.to(dev), u2.to(dev), freq.to(dev), expo.to(dev))
    # random drop: identical except where u ** p sits on the threshold (audited)
    diff = (lg.cpu() - want_logit).abs() > 1e-3
    if diff.any():
        margin = ((u1 ** expo) - freq[:, None, None, :]).abs()[diff]
        assert margin.max().item() < 1e-6, "random-drop mask differs away from the threshold"
    assert not diff.any() or diff.sum() < 3
    if not diff.any():
        assert torch.equal(code.cpu(), want_code)
        assert torch.equal(index.cpu(), want_index[..., 0])
        want_hot = torch.gather(want_sample, -1, want_index)[..., 0]
        assert (hot.cpu() - want_hot).abs().max().item() < 1e-6
        # the sample is zero away from its arg-max, so the dense bmm equals the scaled gather
        deq = ops.vq_dequant_soft(index, hot, ops.PackedCodebook(cb.to(dev))).cpu()
        assert (deq - R.dequant_soft(want_sample, cb)).abs().max().item() < 1e-6


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
    return ch, m, ks, sd, x, us


def test_training_forward_against_reference_vectors(dev):
    from mcquic_amd import Compressor
    z = np.load(os.path.join(G, "f6_train_forward.npz"))
    ch, m, ks, sd, x, us = _train_case()
    model = Compressor(ch, m, ks)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).train()
    out = model(x.to(dev), uniforms=[(a.to(dev), b.to(dev)) for a, b in us])
    xHat, yHat, codes, logits = out
    for lv in range(3):
        assert torch.equal(codes[lv].cpu(), torch.from_numpy(z[f"code{lv}"].astype(np.int64))), f"codes level {lv}"
        np.testing.assert_allclose(logits[lv].detach().cpu().numpy(), z[f"logit{lv}"], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(model._quantizer._entropyCoder._freqEMA[lv].detach().cpu().numpy(), z[f"ema{lv}"],
                                   rtol=0, atol=1e-6)
    np.testing.assert_allclose(yHat.detach().cpu().numpy(), z["yHat"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(xHat.detach().cpu()[..., ::2, ::2].numpy(), z["xHat_strided"], rtol=0, atol=1e-4)
    assert model.eval()(x.to(dev)) is None
    # the same forward without autograd (fused inference kernels + the (index, hot) sample form)
    model.load_state_dict(sd, strict=True)
    with torch.no_grad():
        x2, y2, c2, l2 = model.train()(x.to(dev), uniforms=[(a.to(dev), b.to(dev)) for a, b in us])
    assert not x2.requires_grad
    for lv in range(3):
        assert torch.equal(c2[lv], codes[lv])
    assert (x2 - xHat.detach()).abs().max().item() < 1e-4


def test_training_forward_draws_its_own_uniforms(dev):
    from mcquic_amd import Compressor
    ch, m, ks, sd, x, _ = _train_case()
    model = Compressor(ch, m, ks)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).train()
    torch.manual_seed(0)
    xHat, yHat, codes, logits = model(x.to(dev))
    assert tuple(xHat.shape) == (2, 3, 128, 128) and tuple(yHat.shape) == (2, ch, 16, 16)
    assert [tuple(c.shape) for c in codes] == [(2, m, 8, 8), (2, m, 4, 4), (2, m, 2, 2)]
    assert [tuple(l.shape) for l in logits] == [(2, m, 8, 8, 32), (2, m, 4, 4, 16), (2, m, 2, 2, 8)]
    assert all(torch.isfinite(t).all() for t in (xHat, yHat))
    assert xHat.requires_grad


def test_logits_and_sample_random_shapes(dev):
    """50 seeded random (codebooks, codewords, vector length, batch, map) shapes through the logits kernel and the row kernels of
    the Gumbel sample (k from 8 to 8192, row widths 64 / 256 / 1024 threads and the tails between them): logits within 2e-6,
    codes / sample indices / straight-through values equal to the oracle's except where a random-drop decision sits on its
    threshold (audited as in the fixed case above)."""
    import random
    from mcquic_amd import ops
    rng = random.Random(13)
    flipped_cases = 0
    for it in range(50):
        m = rng.choice([1, 2, 2, 3, 12])
        k = rng.choice([8, 31, 32, 64, 100, 200, 256, 512, 1000, 2048, 8192])
        d = rng.choice([1, 4, 8, 10, 16, 64])
        n, h, w = rng.randint(1, 3), rng.randint(1, 8), rng.randint(1, 8)
        if n * m * h * w * k > 3e6:
            n, h = 1, min(h, 2)
        x, cb, g = _vq_case(m, k, d, n, h, w, 9000 + it)
        temp = torch.rand((m, 1, 1, 1), generator=g) + 0.5
        what = f"#{it} m{m} k{k} d{d} n{n} {h}x{w}"
        logit0 = R.vq_logit(x, cb, temp, torch.tensor([R.EPS]))
        pk = ops.PackedCodebook(cb.to(dev))
        lg = ops.vq_logits(x.to(dev), pk, temp.to(dev), R.EPS)
        err = (lg.cpu() - logit0).abs().max().item()
        assert err <= 2e-6 * max(1.0, logit0.abs().max().item()), f"{what}: logits max abs err {err:.3e}"
        freq = torch.rand((m, k), generator=g) ** 3 + 1e-3
        freq = freq / freq.sum(-1, keepdim=True)
        u1, u2 = torch.rand(logit0.shape, generator=g), torch.rand(logit0.shape, generator=g)
        bits = np.log2(k)
        usage = (freq > R.EPS).float().mean().clamp(0., 1.)
        expo = (-(bits - 1) * (usage ** 2) + bits)
        lg = logit0.clone().to(dev)                                  # (the oracle's logits in: the sample is compared on equal inputs)
        want_logit = R.random_drop(logit0, freq, u1)
        want_sample, _, want_index = R.gumbel_softmax_hard(want_logit, u2)
        code, index, hot = ops.vq_gumbel_sample(lg, u1.to(dev), u2.to(dev), freq.to(dev), expo.to(dev))
        diff = (lg.cpu() - want_logit).abs() > 1e-3
        if diff.any():
            margin = ((u1 ** expo) - freq[:, None, None, :]).abs()[diff]
            assert margin.max().item() < 1e-6 and diff.sum() < 3, f"{what}: random-drop mask differs away from the threshold"
            flipped_cases += 1
            continue
        assert torch.equal(code.cpu(), want_logit.argmax(-1)), what
        assert torch.equal(index.cpu(), want_index[..., 0]), what
        want_hot = torch.gather(want_sample, -1, want_index)[..., 0]
        assert (hot.cpu() - want_hot).abs().max().item() < 1e-6, what
        deq = ops.vq_dequant_soft(index, hot, pk).cpu()
        assert (deq - R.dequant_soft(want_sample, cb)).abs().max().item() < 1e-6, what
    assert flipped_cases <= 5, f"{flipped_cases} of 50 cases had a threshold decision"
