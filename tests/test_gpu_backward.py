"""GPU: backward kernels of the training step against torch.autograd on the CPU (the same formulas the reference
gets from autograd over nn.Conv2d / SiLU / GDN / sigmoid)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import mcquic_ref as R
from _record import record

pytestmark = pytest.mark.gpu

# Worst relative gradient error (per parameter tensor, against its largest entry) allowed per case = 4x what the round-4 GPU run
# measured (profiles/r04_gradient_errors.json, written by tests/_record.py).
# measured (MI355X, round 4): 2.6e-6 / 6.1e-6 / 3.6e-6 for the three training steps, 3.9e-6 / 2.4e-6 with the logits term; against
# the oracle in float64 the HIP gradients are 1.7e-6 / 2.8e-6 off where the float32 CPU path is 2.4e-6 / 3.5e-6.
GRAD_BARS = {
    "full_training_step[8-32-2x128]": 1.1e-5, "full_training_step[128-512-1x128]": 2.5e-5, "full_training_step[128-8192-8x256]": 1.5e-5,
    "full_training_step[192-512-1x128]": 1.5e-5,      # (measured 3.5e-6: model No. 12's width, twelve codebooks of d = 16)
    "logits_gradient[8-32-2x128]": 1.6e-5, "logits_gradient[128-512-1x128]": 1.0e-5,
}


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _close(got, want, tol, what):
    got = got.detach().cpu()
    err = (got - want).abs().max().item()
    ref = max(want.abs().max().item(), 1e-3)
    assert torch.isfinite(got).all(), f"{what}: non-finite"
    assert err <= tol * ref, f"{what}: max abs err {err:.3e} (ref max {ref:.3e})"


@pytest.mark.parametrize("case", [(2, 128, 128, 12, 16, 3, 1), (1, 128, 128, 9, 7, 3, 1), (2, 128, 128, 12, 16, 3, 2),
                                  (1, 128, 128, 9, 7, 3, 2), (2, 128, 128, 10, 6, 1, 1), (2, 3, 128, 16, 12, 3, 2),
                                  (1, 8, 8, 11, 13, 3, 1), (2, 128, 12, 8, 8, 3, 1),
                                  (1, 128, 128, 87, 167, 3, 1),   # 14529 pixels: 227 ranges of 66, the last seven empty
                                  (1, 64, 40, 1, 300, 3, 1), (3, 16, 24, 33, 1, 1, 1)])     # one-pixel-wide / -high maps
def test_conv_backward(dev, case):
    from mcquic_amd.nn import Conv2d
    n, cin, cout, h, w, ks, stride = case
    x = _rand((n, cin, h, w), 1)
    conv = Conv2d(cin, cout, ks, stride)
    wt, b = conv.weight.detach().clone(), conv.bias.detach().clone()
    xr = x.clone().requires_grad_()
    wr, br = wt.clone().requires_grad_(), b.clone().requires_grad_()
    y = F.conv2d(xr, wr, br, stride=stride, padding=ks // 2)
    gy = _rand(tuple(y.shape), 2)
    y.backward(gy)
    conv = conv.to(dev).train()
    xd = x.to(dev).requires_grad_()
    yd = conv(xd)
    _close(yd, y.detach(), 2e-6, "forward")
    yd.backward(gy.to(dev))
    _close(xd.grad, xr.grad, 3e-6, f"dx {case}")
    _close(conv.weight.grad, wr.grad, 3e-6, f"dW {case}")
    _close(conv.bias.grad, br.grad, 3e-6, f"db {case}")


@pytest.mark.parametrize("case", [(2, 128, 128, 16, 16), (8, 128, 128, 32, 32), (1, 128, 512, 8, 16), (2, 128, 12, 16, 24),
                                  (3, 40, 72, 8, 8), (1, 128, 128, 64, 64), (2, 128, 128, 128, 128), (5, 32, 32, 24, 8),
                                  (8, 128, 128, 8, 8), (1, 3, 128, 40, 32)])
def test_wgrad_rows_kernel(dev, case):
    """The NCHW weight-gradient kernel (csrc/wgrad_rows.hip: 3x3, stride 1, H and W multiples of 8) against CPU autograd,
    bit-identical when repeated (deterministic reduction), and equal to the NHWC kernel it replaces up to summation order."""
    from mcquic_amd import ops
    n, cin, cout, h, w = case
    assert ops._lib.load().mcq_conv2d_wgrad_nchw_workspace_floats(n, cin, h, w, cout) > 0
    x, gy = _rand((n, cin, h, w), 11), _rand((n, cout, h, w), 12)
    wr = torch.zeros((cout, cin, 3, 3), requires_grad=True)
    br = torch.zeros((cout,), requires_grad=True)
    F.conv2d(x, wr, br, padding=1).backward(gy)
    dw, db = ops.conv2d_wgrad(x.to(dev), gy.to(dev), 3, 1, want_bias=True)
    _close(dw, wr.grad, 3e-6, f"dW {case}")
    _close(db, br.grad, 3e-6, f"db {case}")
    dw2, db2 = ops.conv2d_wgrad(x.to(dev), gy.to(dev), 3, 1, want_bias=True)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    assert torch.equal(ops.conv2d_wgrad(x.to(dev), gy.to(dev), 3, 1), dw)
    ops._WGRAD_ROWS = False
    try:
        old = ops.conv2d_wgrad(x.to(dev), gy.to(dev), 3, 1)
    finally:
        ops._WGRAD_ROWS = True
    _close(dw, old.cpu(), 3e-6, f"rows vs NHWC kernel {case}")


@pytest.mark.parametrize("case", [(8, 3, 128, 256, 256), (8, 128, 128, 128, 128), (2, 128, 128, 32, 32), (3, 40, 72, 16, 48), (1, 128, 128, 16, 16),
                                  (2, 8, 8, 4, 16), (5, 32, 16, 12, 16)])
def test_wgrad_rows_stride2_kernel(dev, case):
    """The NCHW weight-gradient kernel for 3x3 stride-2 convolutions (ResidualBlockWithStride, the stem) against CPU autograd,
    repeatable bit for bit, and equal to the NHWC kernel it replaces up to summation order."""
    from mcquic_amd import ops
    n, cin, cout, h, w = case
    assert ops._lib.load().mcq_conv2d_wgrad_s2_nchw_workspace_floats(n, cin, h, w, cout) > 0
    x, gy = _rand((n, cin, h, w), 41), _rand((n, cout, h // 2, w // 2), 42)
    wr = torch.zeros((cout, cin, 3, 3), requires_grad=True)
    br = torch.zeros((cout,), requires_grad=True)
    F.conv2d(x, wr, br, stride=2, padding=1).backward(gy)
    dw, db = ops.conv2d_wgrad(x.to(dev), gy.to(dev), 3, 2, want_bias=True)
    _close(dw, wr.grad, 3e-6, f"dW {case}")
    _close(db, br.grad, 3e-6, f"db {case}")
    dw2, db2 = ops.conv2d_wgrad(x.to(dev), gy.to(dev), 3, 2, want_bias=True)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    ops._WGRAD_ROWS = False
    try:
        old = ops.conv2d_wgrad(x.to(dev), gy.to(dev), 3, 2)
    finally:
        ops._WGRAD_ROWS = True
    _close(dw, old.cpu(), 3e-6, f"rows vs NHWC kernel {case}")


@pytest.mark.parametrize("case", [(8, 128, 128, 64, 64, False), (8, 128, 128, 16, 16, True), (2, 128, 128, 8, 8, False), (3, 40, 72, 8, 24, True),
                                  (1, 8, 32, 16, 16, False), (8, 32, 32, 32, 32, True), (2, 128, 128, 6, 8, False)])
def test_wgrad_rows_1x1_kernel(dev, case):
    """The NCHW weight-gradient kernel for 1x1 convolutions (gate conv; GDN's gamma with the squared operand) against CPU
    autograd, repeatable bit for bit, and equal to the NHWC kernel it replaces up to summation order."""
    from mcquic_amd import ops
    n, cin, cout, h, w, sq = case
    assert ops._lib.load().mcq_conv2d_wgrad1x1_nchw_workspace_floats(n, cin, h, w, cout) > 0
    x, gy = _rand((n, cin, h, w), 31), _rand((n, cout, h, w), 32)
    wr = torch.zeros((cout, cin, 1, 1), requires_grad=True)
    br = torch.zeros((cout,), requires_grad=True)
    F.conv2d(x * x if sq else x, wr, br).backward(gy)
    dw, db = ops.conv2d_wgrad(x.to(dev), gy.to(dev), 1, 1, square_x=sq, want_bias=True)
    _close(dw, wr.grad, 3e-6, f"dW {case}")
    _close(db, br.grad, 3e-6, f"db {case}")
    dw2, db2 = ops.conv2d_wgrad(x.to(dev), gy.to(dev), 1, 1, square_x=sq, want_bias=True)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    assert torch.equal(ops.conv2d_wgrad(x.to(dev), gy.to(dev), 1, 1, square_x=sq), dw)
    ops._WGRAD_ROWS = False
    try:
        old = ops.conv2d_wgrad(x.to(dev), gy.to(dev), 1, 1, square_x=sq)
    finally:
        ops._WGRAD_ROWS = True
    _close(dw, old.cpu(), 3e-6, f"rows vs NHWC kernel {case}")


@pytest.mark.parametrize("case", [(8, 128, 128, 4, 4), (3, 40, 24, 4, 4), (11, 16, 16, 4, 8), (2, 128, 128, 6, 6), (8, 128, 128, 2, 2),
                                  (1, 8, 8, 1, 1)])
def test_wgrad_small_map_kernel(dev, case):
    """Maps the strip walk does not take (the 4x4 level of a 256x256 crop): the one-launch LDS kernel against CPU autograd."""
    from mcquic_amd import ops
    n, cin, cout, h, w = case
    assert ops._lib.load().mcq_conv2d_wgrad_nchw_workspace_floats(n, cin, h, w, cout) == 1
    x, gy = _rand((n, cin, h, w), 21), _rand((n, cout, h, w), 22)
    wr = torch.zeros((cout, cin, 3, 3), requires_grad=True)
    br = torch.zeros((cout,), requires_grad=True)
    F.conv2d(x, wr, br, padding=1).backward(gy)
    dw, db = ops.conv2d_wgrad(x.to(dev), gy.to(dev), 3, 1, want_bias=True)
    _close(dw, wr.grad, 3e-6, f"dW {case}")
    _close(db, br.grad, 3e-6, f"db {case}")
    # (a grouped launch may cut the rows into other ranges than the single launch -- its wave budget is shared by the
    #  convolutions -- so the sums agree to float32 rounding, not bit for bit; two problems of one launch are cut alike)
    (dw2, db2), (dw3, _) = ops.conv2d_wgrad_group([x.to(dev), x.to(dev)], [gy.to(dev), gy.to(dev)], want_bias=True)
    _close(dw2, wr.grad, 3e-6, f"grouped dW {case}")
    _close(db2, br.grad, 3e-6, f"grouped db {case}")
    assert torch.equal(dw2, dw3)


@pytest.mark.parametrize("case", [(3, 2, 128, 128, 16, 16), (16, 8, 128, 128, 8, 8), (19, 1, 32, 64, 8, 16),
                                  # the training step's own groups (whole-launch plans: several rounds of long walks, several
                                  # columns per wave, fewer than 1024 waves) and a ragged one
                                  (12, 8, 128, 128, 64, 64), (14, 8, 128, 128, 32, 32), (2, 8, 128, 128, 128, 128),
                                  (16, 8, 128, 128, 16, 16), (5, 3, 40, 72, 24, 16)])
def test_wgrad_rows_grouped_launch(dev, case):
    """Several convolutions of one shape in one launch pair: every (dW, db) equals the single-conv launch to float32 rounding
    (the grouped plan shares the launch's wave budget between the convolutions, i.e. sums the rows in other ranges), and the
    launch is deterministic: the same call twice gives the same bits."""
    from mcquic_amd import ops
    k, n, cin, cout, h, w = case
    xs = [_rand((n, cin, h, w), 100 + i).to(dev) for i in range(k)]
    dys = [_rand((n, cout, h, w), 200 + i).to(dev) for i in range(k)]
    got = ops.conv2d_wgrad_group(xs, dys, want_bias=True)
    assert len(got) == k
    for i, (dw, db) in enumerate(got):
        dw1, db1 = ops.conv2d_wgrad(xs[i], dys[i], 3, 1, want_bias=True)
        _close(dw, dw1.cpu(), 3e-6, f"dW {i}")
        _close(db, db1.cpu(), 3e-6, f"db {i}")
    again = ops.conv2d_wgrad_group(xs, dys, want_bias=True)
    assert all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(got, again))
    nb = ops.conv2d_wgrad_group(xs[:2], dys[:2], want_bias=False)
    assert nb[1][1] is None and nb[1][0].shape == got[1][0].shape
    _close(nb[1][0], got[1][0].cpu(), 3e-6, "without bias")


def test_pixel_shuffle_conv_backward(dev):
    from mcquic_amd.nn import pixelShuffle3x3
    n, c, h, w = 2, 128, 6, 10
    mod = pixelShuffle3x3(c, c, 2)
    wt, b = mod[0].weight.detach().clone().requires_grad_(), mod[0].bias.detach().clone().requires_grad_()
    x = _rand((n, c, h, w), 3)
    xr = x.clone().requires_grad_()
    y = F.pixel_shuffle(F.conv2d(xr, wt, b, padding=1), 2)
    gy = _rand(tuple(y.shape), 4)
    y.backward(gy)
    mod = mod.to(dev).train()
    xd = x.to(dev).requires_grad_()
    yd = mod(xd)
    _close(yd, y.detach(), 2e-6, "forward")
    yd.backward(gy.to(dev))
    _close(xd.grad, xr.grad, 3e-6, "dx")
    _close(mod[0].weight.grad, wt.grad, 3e-6, "dW")
    _close(mod[0].bias.grad, b.grad, 3e-6, "db")


@pytest.mark.parametrize("c", [8, 128])
def test_blocks_backward(dev, c):
    """Each block in training mode: forward and every gradient against CPU autograd through the oracle's functions."""
    from mcquic_amd import nn as N
    x = _rand((2, c, 8, 12), 5)
    cases = [(N.ResidualBlock(c, c), R._rb, R.residual_block), (N.ResidualBlockWithStride(c, c), R._rb_stride, R.residual_block_with_stride),
             (N.ResidualBlockShuffle(c, c), R._rb_shuffle, R.residual_block_shuffle), (N.AttentionBlock(c), R._attn, R.attention_block)]
    for mod, mk, fn in cases:
        sd = {}
        mk(sd, "", c, 9)
        mod.load_state_dict(sd, strict=True)
        params = {k: v.clone().requires_grad_() if v.is_floating_point() and v.dim() > 0 and "reparam" not in k else v for k, v in sd.items()}
        xr = x.clone().requires_grad_()
        y = fn(params, "", xr)
        gy = _rand(tuple(y.shape), 6)
        y.backward(gy)
        mod = mod.to(dev).train()
        xd = x.to(dev).requires_grad_()
        yd = mod(xd)
        _close(yd, y.detach(), 5e-6, f"{type(mod).__name__} forward")
        yd.backward(gy.to(dev))
        _close(xd.grad, xr.grad, 2e-5, f"{type(mod).__name__} dx")
        for name, p in mod.named_parameters():
            want = params[name].grad
            assert want is not None, name
            _close(p.grad, want, 2e-5, f"{type(mod).__name__} d{name}")


def _train_setup(ch, m, ks, n, hw, seed):
    sd = R.make_state_dict(ch, m, ks, seed=seed)
    g = torch.Generator().manual_seed(seed + 100)
    for lv, k in enumerate(ks):
        f = torch.rand((m, k), generator=g) ** 3 + 1e-3
        sd[f"_quantizer._entropyCoder._freqEMA.{lv}"] = f / f.sum(-1, keepdim=True)
        sd[f"_quantizer._encoders.{lv}._quantizer._temperature"] = torch.rand((m, 1, 1, 1), generator=g) + 0.5
    x = R.make_images(n, hw, hw, seed=seed + 1)
    us = []
    for lv, k in enumerate(ks):
        s = hw // 16 // (2 ** lv)
        us.append((torch.rand((n, m, s, s, k), generator=g), torch.rand((n, m, s, s, k), generator=g)))
    return sd, x, us


@pytest.mark.parametrize("cfg", [(8, 2, [32, 16, 8], 2, 128), (128, 2, [512, 64, 16], 1, 128),
                                 (192, 12, [512, 64, 16], 1, 128),          # model No. 12's width and codebook count (d = 16)
                                 (128, 2, [8192, 2048, 512], 8, 256)])      # BASELINE configs[4] itself: qp=2 codebooks, 8 x 256 x 256
def test_full_training_step_gradients(dev, cfg):
    """forward + backward of Compressor in training mode: every parameter gradient against CPU autograd through the
    oracle's forward_train (same weights, same uniform draws, loss = <xHat, G>).  The last case is config #5's own
    workload (8192-codeword soft assignment, logits [8, 2, 16, 16, 8192]; ~20 s of CPU autograd)."""
    from mcquic_amd import Compressor
    ch, m, ks, n, hw = cfg
    sd, x, us = _train_setup(ch, m, ks, n, hw, 21)
    leaf = {k: (v.clone().requires_grad_() if v.is_floating_point() and "reparam" not in k and "_bound" not in k and "_freqEMA" not in k else v)
            for k, v in sd.items()}
    # the codebook is one tensor under three names
    for lv in range(len(ks)):
        cb = leaf[f"_quantizer._encoders.{lv}._quantizer._codebook"]
        leaf[f"_quantizer._encoders.{lv}._dequantizer._codebook"] = cb
        leaf[f"_quantizer._decoders.{lv}._dequantizer._codebook"] = cb
    xHat, yHat, codes, logits, _ = R.forward_train(leaf, x, us)
    G = torch.rand(xHat.shape, generator=torch.Generator().manual_seed(5)) - 0.5
    (xHat * G).sum().backward()

    model = Compressor(ch, m, ks)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).train()
    out = model(x.to(dev), uniforms=[(a.to(dev), b.to(dev)) for a, b in us])
    for lv in range(len(ks)):
        assert torch.equal(out[2][lv].cpu(), codes[lv]), f"codes level {lv}"
    _close(out[0], xHat.detach(), 1e-4, "xHat")
    (out[0] * G.to(dev)).sum().backward()
    worst = ("", 0.0)
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        want = leaf[name].grad
        assert want is not None, f"oracle has no grad for {name}"
        assert p.grad is not None, f"no grad for {name}"
        got = p.grad.detach().cpu()
        rel = (got - want).abs().max().item() / max(want.abs().max().item(), 1e-6)
        if rel > worst[1]:
            worst = (name, rel)
    bar = GRAD_BARS[f"full_training_step[{ch}-{ks[0]}-{n}x{hw}]"]
    record(f"full_training_step[{ch}-{ks[0]}-{n}x{hw}]", worst_rel_grad_err=worst[1], at=worst[0], bar=bar)
    assert worst[1] < bar, f"worst gradient mismatch {worst[1]:.3e} at {worst[0]}"


@pytest.mark.parametrize("cfg", [(8, 2, [32, 16, 8], 2, 128), (128, 2, [512, 64, 16], 1, 128)])
def test_training_step_gradients_against_float64(dev, cfg):
    """The same step with the oracle in FLOAT64 as the truth: the HIP path (exact-fp32 MFMA contractions, v_log_f32 / v_exp_f32
    and Markstein division in the soft assignment, csrc/vq_train.hip) must sit as close to it as the float32 CPU path does --
    per parameter, error relative to the tensor's largest gradient; the discrete decisions (codes, random drop, hard sample) of
    the float64 run must be the float32 run's, or the comparison is meaningless (then the case is skipped, not failed)."""
    from mcquic_amd import Compressor
    ch, m, ks, n, hw = cfg
    sd, x, us = _train_setup(ch, m, ks, n, hw, 21)

    def leaves(dtype):
        leaf = {k: ((v.to(dtype) if v.is_floating_point() else v).clone().requires_grad_()
                    if v.is_floating_point() and "reparam" not in k and "_bound" not in k and "_freqEMA" not in k
                    else (v.to(dtype) if v.is_floating_point() else v)) for k, v in sd.items()}
        for lv in range(len(ks)):
            cb = leaf[f"_quantizer._encoders.{lv}._quantizer._codebook"]
            leaf[f"_quantizer._encoders.{lv}._dequantizer._codebook"] = cb
            leaf[f"_quantizer._decoders.{lv}._dequantizer._codebook"] = cb
        return leaf
    G = torch.rand((n, 3, hw, hw), generator=torch.Generator().manual_seed(5)) - 0.5
    l32, l64 = leaves(torch.float32), leaves(torch.float64)
    o32 = R.forward_train(l32, x, us)
    (o32[0] * G).sum().backward()
    o64 = R.forward_train(l64, x.double(), [(a.double(), b.double()) for a, b in us])
    (o64[0] * G.double()).sum().backward()
    if not all(torch.equal(a, b) for a, b in zip(o32[2], o64[2])):
        pytest.skip("the float64 run takes other discrete decisions than the float32 run on this seed")
    model = Compressor(ch, m, ks)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).train()
    out = model(x.to(dev), uniforms=[(a.to(dev), b.to(dev)) for a, b in us])
    (out[0] * G.to(dev)).sum().backward()
    worst_gpu, worst_cpu = ("", 0.0), ("", 0.0)
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        truth = l64[name].grad
        scale = max(truth.abs().max().item(), 1e-6)
        eg = (p.grad.detach().cpu().double() - truth).abs().max().item() / scale
        ec = (l32[name].grad.double() - truth).abs().max().item() / scale
        if eg > worst_gpu[1]:
            worst_gpu = (name, eg)
        if ec > worst_cpu[1]:
            worst_cpu = (name, ec)
    record(f"training_step_vs_float64[{ch}-{ks[0]}-{n}x{hw}]", hip_worst_rel_err=worst_gpu[1], hip_at=worst_gpu[0],
           cpu_f32_worst_rel_err=worst_cpu[1], cpu_at=worst_cpu[0])
    assert worst_gpu[1] <= max(4.0 * worst_cpu[1], 2e-5), (f"HIP gradients are {worst_gpu[1]:.3e} from the float64 truth at {worst_gpu[0]}; "
                                                          f"the float32 CPU path is {worst_cpu[1]:.3e} ({worst_cpu[0]})")


@pytest.mark.parametrize("cfg", [(8, 2, [32, 16, 8], 2, 128), (128, 2, [512, 64, 16], 1, 128)])
def test_logits_carry_their_gradient(dev, cfg):
    """The returned logits are differentiable like the reference's (mcquic/modules/quantizer.py:181-183,232-239): a loss on
    them -- <logits_l, W_l>, every entry including the randomly dropped ones -- reaches latents, codebooks, temperatures and
    everything upstream.  All parameter gradients of <xHat, G> + sum_l <logits_l, W_l> against CPU autograd through the oracle."""
    from mcquic_amd import Compressor
    ch, m, ks, n, hw = cfg
    sd, x, us = _train_setup(ch, m, ks, n, hw, 23)
    leaf = {k: (v.clone().requires_grad_() if v.is_floating_point() and "reparam" not in k and "_bound" not in k and "_freqEMA" not in k else v)
            for k, v in sd.items()}
    for lv in range(len(ks)):
        cb = leaf[f"_quantizer._encoders.{lv}._quantizer._codebook"]
        leaf[f"_quantizer._encoders.{lv}._dequantizer._codebook"] = cb
        leaf[f"_quantizer._decoders.{lv}._dequantizer._codebook"] = cb
    xHat, yHat, codes, logits, _ = R.forward_train(leaf, x, us)
    gen = torch.Generator().manual_seed(6)
    G = torch.rand(xHat.shape, generator=gen) - 0.5
    Ws = [(torch.rand(lg.shape, generator=gen) - 0.5) * 0.05 for lg in logits]
    ((xHat * G).sum() + sum((lg * w).sum() for lg, w in zip(logits, Ws))).backward()

    model = Compressor(ch, m, ks)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).train()
    out = model(x.to(dev), uniforms=[(a.to(dev), b.to(dev)) for a, b in us])
    assert all(lg.requires_grad for lg in out[3])
    ((out[0] * G.to(dev)).sum() + sum((lg * w.to(dev)).sum() for lg, w in zip(out[3], Ws))).backward()
    worst = ("", 0.0)
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        want, got = leaf[name].grad, p.grad.detach().cpu()
        rel = (got - want).abs().max().item() / max(want.abs().max().item(), 1e-6)
        if rel > worst[1]:
            worst = (name, rel)
    bar = GRAD_BARS[f"logits_gradient[{ch}-{ks[0]}-{n}x{hw}]"]
    record(f"logits_gradient[{ch}-{ks[0]}-{n}x{hw}]", worst_rel_grad_err=worst[1], at=worst[0], bar=bar)
    assert worst[1] < bar, f"worst gradient mismatch {worst[1]:.3e} at {worst[0]}"
    # the logits term alone (no gradient on xHat at all): the quantizer-side parameters still receive theirs
    for p in model.parameters():
        p.grad = None
    out = model(x.to(dev), uniforms=[(a.to(dev), b.to(dev)) for a, b in us])
    (out[3][0] * Ws[0].to(dev)).sum().backward()
    assert model._quantizer._encoders[0]._quantizer._temperature.grad is not None
    assert model._encoder[0].weight.grad is not None and float(model._encoder[0].weight.grad.abs().max()) > 0


def test_training_graph_survives_a_second_backward(dev):
    """the lockstep heads keep their activations through ctx.save_for_backward, so two backward passes over one
    graph (retain_graph=True: two losses, gradient penalties) work and accumulate like autograd's own nodes."""
    from mcquic_amd import Compressor
    sd, x, us = _train_setup(8, 2, [32, 16, 8], 2, 128, 25)
    model = Compressor(8, 2, [32, 16, 8])
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).train()
    out = model(x.to(dev), uniforms=[(a.to(dev), b.to(dev)) for a, b in us])
    G = (torch.rand(out[0].shape, generator=torch.Generator().manual_seed(7)) - 0.5).to(dev)
    loss = (out[0] * G).sum()
    loss.backward(retain_graph=True)
    once = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    loss.backward()
    for n, p in model.named_parameters():
        if n in once:
            assert torch.allclose(p.grad, 2 * once[n], rtol=1e-6, atol=1e-7), n


def test_model_pickles(dev):
    """no local lambdas on the module (torch.save(model) / multiprocessing spawn)."""
    import io
    from mcquic_amd import Compressor
    model = Compressor(8, 2, [32, 16, 8]).to(dev).eval()
    x = R.make_images(1, 128, 128).to(dev)
    want = model.encode(x)
    buf = io.BytesIO()
    torch.save(model, buf)
    buf.seek(0)
    again = torch.load(buf, weights_only=False)
    assert all(torch.equal(a, b) for a, b in zip(again.encode(x), want))


@pytest.mark.parametrize("case", [(128, 128, 3, 1, 5), (128, 128, 3, 2, 3), (128, 128, 1, 1, 17), (40, 24, 3, 1, 2), (12, 128, 3, 1, 3),
                                  (128, 3, 3, 2, 1)])
def test_grouped_pack_equals_single_packs(dev, case):
    """ops.pack_convs (up to 16 weights of one shape per launch) against one PackedConv / PackedConv.dgrad per weight: the same
    operand streams bit for bit, forward and input-gradient form (flip / transpose, sub-pixel form for stride 2)."""
    from mcquic_amd import ops
    cout, cin, ks, stride, n = case
    ws = [_rand((cout, cin, ks, ks), 300 + i, 0.1).to(dev) for i in range(n)]
    bs = [_rand((cout,), 400 + i).to(dev) if i % 2 == 0 else None for i in range(n)]
    got = ops.pack_convs(ws, bs)
    for i in range(n):
        one = ops.PackedConv(ws[i], bs[i])
        assert torch.equal(got[i].wp, one.wp), f"forward stream {i}"
        assert (got[i].cout, got[i].cin, got[i].ksize) == (one.cout, one.cin, one.ksize)
        assert (got[i].bias is None) == (bs[i] is None) and (bs[i] is None or torch.equal(got[i].bias, bs[i]))
    got = ops.pack_convs(ws, dgrad=True, stride=stride, scale=1.0)
    for i in range(n):
        one = ops.PackedConv.dgrad(ws[i], stride)
        assert torch.equal(got[i].wp, one.wp), f"input-gradient stream {i}"
        assert (got[i].cout, got[i].cin, got[i].ksize, got[i].bias) == (one.cout, one.cin, one.ksize, None)


def test_repack_stale_after_parameter_update(dev):
    """The grouped re-pack a training step starts with (Compressor._repackStale): after an in-place update of every parameter
    all forward streams and all input-gradient streams that had been built are fresh again, bit-equal to packing one by one,
    and a second call finds nothing to do."""
    from mcquic_amd import Compressor, ops
    from mcquic_amd.nn.convs import Conv2d
    torch.manual_seed(3)
    model = Compressor(8, 2, [16, 8, 4]).to(dev).train()
    x = _rand((2, 3, 64, 64), 9).to(dev)
    model(x)[0].sum().backward()
    convs = [m for m in model.modules() if isinstance(m, Conv2d)]
    assert Conv2d.repack_stale(convs) == 0
    with_dgrad = [c for c in convs if "_dgradCache" in c.__dict__]
    assert len(with_dgrad) > len(convs) // 2
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.01 * torch.randn_like(p))
    assert Conv2d.repack_stale(convs) == len(convs) + len(with_dgrad)
    for c in convs:
        fresh = ops.PackedConv(c.weight, c.bias)
        assert c._packedKey == c._key() and torch.equal(c._packed.wp, fresh.wp)
        assert (c.bias is None) or torch.equal(c._packed.bias, c.bias)
    for c in with_dgrad:
        cache = c.__dict__["_dgradCache"]
        assert cache.key == (ops.tensor_version(c.weight), c.weight.data_ptr(), c.stride)
        assert torch.equal(cache.packed.wp, ops.PackedConv.dgrad(c.weight, c.stride).wp)
    assert Conv2d.repack_stale(convs) == 0
    out = model(x)                                   # and the step after the update runs on them
    assert torch.isfinite(out[0]).all()


def test_winograd_input_gradient_pack(dev):
    """Opt-in: the stride-1 input-gradient convolution of a 3x3 layer in the Winograd F(2, 3) form, packed straight from the
    layer's OIHW weight (mcq_pack_conv_dgrad_weight_winograd_f32), against torch.autograd's conv2d input gradient."""
    from mcquic_amd import ops
    x = _rand((2, 128, 18, 22), 1).requires_grad_()
    wt = _rand((64, 128, 3, 3), 2, 0.05)
    dy = _rand((2, 64, 18, 22), 3)
    F.conv2d(x, wt, None, padding=1).backward(dy)
    back = ops.PackedConv.dgrad(wt.to(dev), 1, winograd=True)
    assert back.wino is not None and (back.cout, back.cin) == (128, 64)
    _close(ops.conv2d(dy.to(dev), back, winograd=True), x.grad, 1e-5, "winograd input gradient")
    _close(ops.conv2d(dy.to(dev), back, winograd=False), x.grad, 2e-6, "direct input gradient")


@pytest.mark.parametrize("case", [(8, 2, 64, 16, 16, 8192),     # config #5, first level: both MFMA forms
                                  (8, 2, 64, 8, 8, 2048),       # second level: both MFMA forms
                                  (8, 2, 64, 4, 4, 512),        # third level: dC on the MFMA form, dx on the lane-per-channel one
                                  (2, 3, 24, 4, 16, 512),       # d < 32: one channel block empty, the other partly
                                  (1, 2, 40, 8, 8, 1024),       # d between the blocks
                                  (3, 2, 16, 3, 5, 48)])        # nothing fits: both lane-per-channel forms
def test_soft_assignment_backward_contractions(dev, case):
    """mcq_vq_soft_bwd_f32 (the MFMA forms of csrc/vq_bwd_mfma.hip and the forms they replace) against the two contractions
    written out in float64: dx = 2 x rowsum - 2 ddist C, dC = 2 C colsum - 2 ddist^T x + scatter(hot dDeq)."""
    import numpy as np
    from mcquic_amd import ops
    n, m, d, h, w, k = case
    hw = h * w
    ddist = _rand((n, m, h, w, k), 1, 1e-3)
    x = _rand((n, m * d, h, w), 2)
    ddeq = _rand((n, m * d, h, w), 3)
    cb = _rand((m, k, d), 4)
    g = torch.Generator().manual_seed(5)
    index = torch.randint(0, k, (n, m, h, w), generator=g)
    index[0, 0, 0, :] = 7 % k                                    # several vectors on one codeword: the scatter adds them up
    hot = 1.0 + _rand((n, m, h, w), 6, 1e-3)
    rowsum = ddist.double().sum(-1).float()
    dd = ddist.double().numpy().reshape(n, m, hw, k)
    xx = x.double().numpy().reshape(n, m, d, hw)
    dq = ddeq.double().numpy().reshape(n, m, d, hw)
    cc = cb.double().numpy()
    want_dx = 2 * xx * rowsum.double().numpy().reshape(n, m, 1, hw) - 2 * np.einsum("ngvk,gkj->ngjv", dd, cc)
    want_dc = 2 * cc * dd.sum((0, 2))[:, :, None] - 2 * np.einsum("ngvk,ngjv->gkj", dd, xx)
    idx, hv = index.numpy().reshape(n, m, hw), hot.double().numpy().reshape(n, m, hw)
    for a in range(n):
        for b in range(m):
            np.add.at(want_dc[b], idx[a, b], (hv[a, b][None, :] * dq[a, b]).T)
    pk = ops.PackedCodebook(cb.to(dev))
    dx, dcb = ops.vq_soft_bwd(ddist.to(dev), rowsum.to(dev), x.to(dev), ddeq.to(dev), index.to(dev), hot.to(dev), pk)
    _close(dx, torch.from_numpy(want_dx.reshape(n, m * d, h, w)).float(), 2e-6, f"dx {case}")
    _close(dcb, torch.from_numpy(want_dc).float(), 2e-6, f"dcodebook {case}")
    dx2, dcb2 = ops.vq_soft_bwd(ddist.to(dev), rowsum.to(dev), x.to(dev), ddeq.to(dev), index.to(dev), hot.to(dev), pk)
    assert torch.equal(dx, dx2) and torch.equal(dcb, dcb2), "the contractions are deterministic"


@pytest.mark.parametrize("case", [(8, 128, 128, 8, 8), (8, 128, 128, 4, 4), (1, 32, 64, 16, 16), (3, 64, 32, 6, 6), (11, 16, 48, 2, 10), (2, 128, 128, 10, 12),
                                  (5, 48, 16, 2, 2), (2, 128, 128, 16, 16), (3, 32, 32, 8, 16), (3, 16, 16, 12, 12)])
def test_wgrad_few_pixels_kernel(dev, case):
    """csrc/wgrad_t16.h: weight gradients over <= 512 pixels (16 x 16 tiles on v_mfma_f32_16x16x4_f32, one pass, whole images staged
    in LDS in groups of <= 256 pixels): 3x3 (single and grouped launches), 1x1 and 1x1 over x^2, against CPU autograd -- image
    counts below / above a group, maps that are not powers of two, rectangular tiles of the weight."""
    from mcquic_amd import ops
    n, cin, cout, h, w = case
    lib = ops._lib.load()
    assert lib.mcq_conv2d_wgrad_nchw_workspace_floats(n, cin, h, w, cout) == 1 and lib.mcq_conv2d_wgrad1x1_nchw_workspace_floats(n, cin, h, w, cout) == 1
    xs = [_rand((n, cin, h, w), 31 + i) for i in range(3)]
    gys = [_rand((n, cout, h, w), 41 + i) for i in range(3)]
    wants = []
    for x, gy in zip(xs, gys):
        wr = torch.zeros((cout, cin, 3, 3), requires_grad=True)
        br = torch.zeros((cout,), requires_grad=True)
        F.conv2d(x, wr, br, padding=1).backward(gy)
        wants.append((wr.grad, br.grad))
    dw, db = ops.conv2d_wgrad(xs[0].to(dev), gys[0].to(dev), 3, 1, want_bias=True)
    _close(dw, wants[0][0], 3e-6, f"dW {case}")
    _close(db, wants[0][1], 3e-6, f"db {case}")
    outs = ops.conv2d_wgrad_group([x.to(dev) for x in xs], [g.to(dev) for g in gys], want_bias=True)
    for (gw, gb), (ww, wb) in zip(outs, wants):
        _close(gw, ww, 3e-6, f"grouped dW {case}")
        _close(gb, wb, 3e-6, f"grouped db {case}")
    assert torch.equal(outs[0][0], dw) and torch.equal(outs[0][1], db), "a convolution's result does not depend on its launch"
    for sq in (False, True):
        w1 = torch.zeros((cout, cin, 1, 1), requires_grad=True)
        b1 = torch.zeros((cout,), requires_grad=True)
        F.conv2d(xs[1] * xs[1] if sq else xs[1], w1, b1).backward(gys[1])
        dw1, db1 = ops.conv2d_wgrad(xs[1].to(dev), gys[1].to(dev), 1, 1, square_x=sq, want_bias=True)
        _close(dw1, w1.grad, 3e-6, f"1x1 dW {case} sq={sq}")
        _close(db1, b1.grad, 3e-6, f"1x1 db {case} sq={sq}")


def test_training_graph_producers_hand_silu_twins_on(dev):
    """Round 4: in the training graph the closing convolution of a strided / shuffle block (and the conv3x3 in the middle of a
    head) writes silu(out) in the same launch, so the block that follows starts from the twin instead of a stand-alone SiLU
    launch; the three gradient paths meeting at an AttentionBlock's input are one 3-way add (mcq_add3_f32).  Twins must be
    silu(out) exactly as the SiLU kernel gives it, and carry no autograd graph."""
    from mcquic_amd import nn as N, ops
    c = 32
    x = _rand((2, c, 8, 12), 5).to(dev).requires_grad_()
    for blk in (N.ResidualBlockWithStride(c, c), N.ResidualBlockShuffle(c, c)):
        blk = blk.to(dev).train()
        y = blk(x)
        tw = ops.silu_twin(y)
        assert tw is not None and tw.grad_fn is None and not tw.requires_grad
        assert torch.equal(tw, ops.silu(y.detach()))
        y.sum().backward()
    a, b, cc = (_rand((3, 5, 7, 9), s).to(dev) for s in (1, 2, 3))
    assert torch.equal(ops.add3(a, b, cc), (a + b) + cc)
    t = _rand((1031,), 4).to(dev)                                   # a tail that is no multiple of four
    assert torch.equal(ops.add3(t, t, t), (t + t) + t)


@pytest.mark.parametrize("case", [(2, 128, 24, 16), (1, 32, 9, 7), (8, 128, 64, 64)])
@pytest.mark.parametrize("inverse", [False, True])
def test_gdn_backward_epilogue_is_the_two_launch_form(dev, case, inverse):
    """MCQ_CONV_GDN_BWD / _IGDN_BWD: the element-wise gradients of y = x f(beta + gamma x^2) as the epilogue of the launch that
    recomputes s -- bit-equal to the round-3 form (s stored by one launch, mcq_gdn_bwd_prep_f32 behind it)."""
    from mcquic_amd import ops
    n, c, h, w = case
    x = _rand((n, c, h, w), 3, 2.0).to(dev)
    dy = _rand((n, c, h, w), 4).to(dev)
    gamma = (_rand((c, c, 1, 1), 5).abs() * 0.1 + torch.eye(c)[..., None, None] * 0.1)
    beta = _rand((c,), 6).abs() + 1.0
    pk = ops.PackedConv(gamma.to(dev), beta.to(dev))
    s = ops.conv2d(x, pk, square_in=True)
    want_dxd, want_ds = ops.gdn_bwd_prep(x, s, dy, inverse)
    dxd, ds = ops.conv2d_gdn_bwd(x, pk, dy, inverse)
    assert torch.isfinite(dxd).all() and torch.isfinite(ds).all()
    assert torch.equal(dxd, want_dxd) and torch.equal(ds, want_ds)


@pytest.mark.parametrize("case", [(8, 128, 128, 16, 16, 12), (2, 128, 128, 32, 32, 1), (8, 128, 128, 64, 64, 2), (3, 64, 96, 24, 16, 5)])
def test_wgrad_grouped_launches_are_deterministic(dev, case):
    """Grouped weight-gradient launches (row walk + fixed-order reduce pass, no atomics): repeated launches give bit-identical
    results, equal to torch.autograd's within float32 reassociation.  (Round 4 also ran this against a last-arriver tail in
    place of the reduce pass: correct, deterministic, and 1.4 ms slower per training step -- removed, see csrc/wgrad_rows.hip.)"""
    from mcquic_amd import ops
    n, cin, cout, h, w, k = case
    xs = [_rand((n, cin, h, w), 300 + i).to(dev) for i in range(k)]
    dys = [_rand((n, cout, h, w), 400 + i).to(dev) for i in range(k)]
    first = [(a.clone(), b.clone()) for a, b in ops.conv2d_wgrad_group(xs, dys, want_bias=True)]
    for it in range(6):
        for (a, b), (c, d) in zip(first, ops.conv2d_wgrad_group(xs, dys, want_bias=True)):
            assert torch.equal(a, c) and torch.equal(b, d), f"launch {it} differs from the first"
    for i in (0, k - 1):
        wr = torch.zeros((cout, cin, 3, 3), requires_grad=True)
        br = torch.zeros((cout,), requires_grad=True)
        F.conv2d(xs[i].cpu(), wr, br, padding=1).backward(dys[i].cpu())
        _close(first[i][0], wr.grad, 2e-5, f"dW of problem {i}")
        _close(first[i][1], br.grad, 2e-5, f"db of problem {i}")


def test_in_kernel_uniform_generator_statistics(dev):
    """csrc/vq_train.hip: rng_uniform -- the counter-based generator behind the soft assignment's draws when no tensors are given
    (the reference draws torch.rand_like(logit) twice per level, quantizer.py:194-230; no RNG-stream parity exists on either
    side, the draws only have to be i.i.d. uniforms on float32's [0, 1) grid).  4 M draws per stream: range, moments, a 256-bin
    chi-square, independence of the two streams, of neighbouring elements and of consecutive snapshots; reproducible from the
    snapshot alone."""
    from mcquic_amd import ops
    ops.seed_rng(1234, dev)
    a = ops.rng_snapshot(dev)
    b = ops.rng_snapshot(dev)
    assert int(a[0]) == 1234 and int(b[1]) == int(a[1]) + 1
    n = 1 << 22
    u0, u1 = ops.hash_uniform(a, 0, (n,)), ops.hash_uniform(a, 1, (n,))
    v0 = ops.hash_uniform(b, 0, (n,))
    assert torch.equal(u0, ops.hash_uniform(a.clone(), 0, (n,)))                      # a function of the snapshot alone
    for u in (u0, u1, v0):
        assert float(u.min()) >= 0.0 and float(u.max()) < 1.0
        assert bool(((u * 16777216.0) == (u * 16777216.0).round()).all())           # multiples of 2^-24
        d = u.double()
        assert abs(float(d.mean()) - 0.5) < 6e-4 and abs(float(d.var()) - 1.0 / 12.0) < 3e-4
        hist = torch.histc(u, bins=256, min=0.0, max=1.0).double()
        chi2 = float(((hist - n / 256) ** 2 / (n / 256)).sum())
        assert chi2 < 400.0, chi2                                                      # 255 degrees of freedom: mean 255, sd 22.6

    def corr(p, q):
        p, q = p.double() - 0.5, q.double() - 0.5
        return float((p * q).mean() / (p.std() * q.std()))
    assert abs(corr(u0, u1)) < 3e-3 and abs(corr(u0, v0)) < 3e-3 and abs(corr(u0[:-1], u0[1:])) < 3e-3
    assert abs(corr(u0[:-8192], u0[8192:])) < 3e-3                                    # the same column of neighbouring rows


@pytest.mark.parametrize("mk", [(2, 8192), (2, 2048), (3, 512), (1, 20000)])
def test_soft_assignment_with_in_kernel_draws_equals_given_tensors(dev, mk):
    """The draws made inside the kernels are exactly mcq_hash_uniform_f32's: the soft assignment run from a generator snapshot
    (no u_drop / u_gumbel tensors) gives bit-identical codes, samples, logits and gradients to the same call on the materialised
    tensors -- for every kernel variant (rows of 64 / 256 / 1024 threads, the wave-per-row form beyond 8192 codewords)."""
    from mcquic_amd import ops
    from mcquic_amd.autograd import SoftQuantizeFn
    m, k = mk
    d, n, h, w = 16, 2, 4, 4
    g = torch.Generator().manual_seed(k)
    cbp = (torch.randn((m, k, d), generator=g) * 0.3).to(dev).requires_grad_()
    x0 = (torch.randn((n, m * d, h, w), generator=g) * 0.3).to(dev)
    temp = (torch.rand((m, 1, 1, 1), generator=g) + 0.5).to(dev).requires_grad_()
    f = torch.rand((m, k), generator=g) ** 3 + 1e-3
    freq = (f / f.sum(-1, keepdim=True)).to(dev)
    expo = torch.tensor([5.0], device=dev)
    gd = torch.randn((n, m * d, h, w), generator=g).to(dev)
    rng = ops.rng_snapshot(dev)
    shape = (n, m, h, w, k)
    ud, ug = ops.hash_uniform(rng, 0, shape), ops.hash_uniform(rng, 1, shape)
    outs = []
    for args in ((None, None, rng), (ud, ug, None), (ud, None, rng)):
        x = x0.clone().requires_grad_()
        cbp.grad = temp.grad = None
        pk = ops.PackedCodebook(cbp)
        deq, code, logit, _sdeq = SoftQuantizeFn.apply(x, cbp, temp, freq, args[0], args[1], expo, pk, 1e-6, args[2])
        (deq * gd).sum().backward()
        outs.append((deq.detach(), code, logit.detach(), x.grad.clone(), cbp.grad.clone(), temp.grad.clone()))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)
    assert float(outs[0][3].abs().max()) > 0
    # the generator moves on: the next snapshot gives another sample
    rng2 = ops.rng_snapshot(dev)
    assert int(rng2[1]) == int(rng[1]) + 1
    assert not torch.equal(ops.hash_uniform(rng2, 1, shape), ug)


def test_conv_backward_random_shapes(dev):
    """100 seeded random convolution layers (channel counts 1 ... 256 that divide nothing, maps 1 x 1 ... 48 x 48, batches 1 ... 6,
    3x3 stride 1 / 2 and 1x1) forward + backward through nn.Conv2d's HIP path against torch.autograd in float64: the fixed cases
    above pin the network's shapes, this sweeps which weight-gradient / input-gradient kernel a shape is routed to (row walk,
    few-pixel tiles, NHWC fallback, stride-2 ring, ragged channel tails, grouped plans)."""
    import random
    from mcquic_amd.nn import Conv2d
    rng = random.Random(4)
    chans = [1, 2, 3, 5, 8, 12, 16, 17, 24, 32, 33, 48, 64, 65, 96, 128, 129, 192, 256]
    worst = 0.0
    for it in range(100):
        cin, cout = rng.choice(chans), rng.choice(chans)
        ks = rng.choice([3, 3, 3, 1])
        stride = rng.choice([1, 1, 2]) if ks == 3 else 1
        n, h, w = rng.randint(1, 6), rng.randint(1, 48), rng.randint(1, 48)
        if rng.random() < 0.3:                                   # the row-walk kernel's own domain: multiples of 8
            h, w = 8 * rng.randint(1, 6), 8 * rng.randint(1, 6)
        x = _rand((n, cin, h, w), 5000 + it)
        conv = Conv2d(cin, cout, ks, stride)
        wr, br = conv.weight.detach().double().requires_grad_(), conv.bias.detach().double().requires_grad_()
        xr = x.double().requires_grad_()
        y = F.conv2d(xr, wr, br, stride=stride, padding=ks // 2)
        gy = _rand(tuple(y.shape), 6000 + it)
        y.backward(gy.double())
        conv = conv.to(dev).train()
        xd = x.to(dev).requires_grad_()
        yd = conv(xd)
        what = f"#{it} n{n} {cin}->{cout} {h}x{w} k{ks}s{stride}"
        # float32 sums of K terms against float64: the bars of the fixed cases (K = 1152 products, <= 4096 pixels), widened by
        # sqrt(K / that) for longer sums
        grow = lambda terms, base: max(1.0, (terms / base) ** 0.5)                                   # noqa: E731
        _close(yd, y.detach().float(), 2e-6 * grow(cin * ks * ks, 1152), what + " forward")
        yd.backward(gy.to(dev))
        _close(xd.grad, xr.grad.float(), 3e-6 * grow(cout * ks * ks, 1152), what + " dx")
        _close(conv.weight.grad, wr.grad.float(), 3e-6 * grow(n * h * w, 4096), what + " dW")
        _close(conv.bias.grad, br.grad.float(), 3e-6 * grow(n * h * w, 4096), what + " db")
        worst = max(worst, float((conv.weight.grad.cpu() - wr.grad.float()).abs().max()) / max(float(wr.grad.abs().max()), 1e-3))
    record("conv_backward_random_shapes", worst_dW_relative=worst)


def test_soft_assignment_backward_random_shapes(dev):
    """40 seeded random (batch, codebooks, vector length, map, codewords) shapes through mcq_vq_soft_bwd_f32 against the two
    contractions in float64: which of the MFMA / tiled / lane-per-channel forms takes a shape depends on d, k, the map and the row
    count (csrc/vq_bwd_mfma.hip: mcq_vq_dc_mfma_ok / mcq_vq_dx_mfma_ok) -- the fixed cases pin one shape per form, this sweeps
    their borders."""
    import random
    import numpy as np
    from mcquic_amd import ops
    rng = random.Random(11)
    for it in range(40):
        m = rng.choice([1, 2, 2, 3, 12])
        d = rng.choice([1, 4, 5, 8, 16, 24, 40, 64, 64])
        k = rng.choice([8, 32, 48, 100, 128, 256, 512, 1024, 2048])
        n, h, w = rng.randint(1, 8), rng.randint(1, 16), rng.randint(1, 16)
        if rng.random() < 0.4:
            h, w = rng.choice([2, 4, 8, 16]), rng.choice([2, 4, 8, 16])
        if n * m * h * w * k > 6e6:
            n = 1
        hw = h * w
        ddist = _rand((n, m, h, w, k), 8000 + it, 1e-3)
        x = _rand((n, m * d, h, w), 8100 + it)
        ddeq = _rand((n, m * d, h, w), 8200 + it)
        cb = _rand((m, k, d), 8300 + it)
        index = torch.randint(0, k, (n, m, h, w), generator=torch.Generator().manual_seed(8400 + it))
        hot = 1.0 + _rand((n, m, h, w), 8500 + it, 1e-3)
        rowsum = ddist.double().sum(-1).float()
        dd = ddist.double().numpy().reshape(n, m, hw, k)
        xx = x.double().numpy().reshape(n, m, d, hw)
        dq = ddeq.double().numpy().reshape(n, m, d, hw)
        cc = cb.double().numpy()
        want_dx = 2 * xx * rowsum.double().numpy().reshape(n, m, 1, hw) - 2 * np.einsum("ngvk,gkj->ngjv", dd, cc)
        want_dc = 2 * cc * dd.sum((0, 2))[:, :, None] - 2 * np.einsum("ngvk,ngjv->gkj", dd, xx)
        idx, hv = index.numpy().reshape(n, m, hw), hot.double().numpy().reshape(n, m, hw)
        for a in range(n):
            for b in range(m):
                np.add.at(want_dc[b], idx[a, b], (hv[a, b][None, :] * dq[a, b]).T)
        pk = ops.PackedCodebook(cb.to(dev))
        dx, dcb = ops.vq_soft_bwd(ddist.to(dev), rowsum.to(dev), x.to(dev), ddeq.to(dev), index.to(dev), hot.to(dev), pk)
        what = f"#{it} n{n} m{m} d{d} {h}x{w} k{k}"
        _close(dx, torch.from_numpy(want_dx.reshape(n, m * d, h, w)).float(), 2e-6, "dx " + what)
        _close(dcb, torch.from_numpy(want_dc).float(), 3e-6, "dcodebook " + what)


@pytest.mark.parametrize("which", [pytest.param("sample", id="sample"), pytest.param("all", id="all", marks=pytest.mark.sweep)])
def test_blocks_backward_random_shapes(dev, which):
    """The four block types in training mode at 14 seeded random (channels, batch, map) shapes -- channel counts that are no
    multiple of 32 (GDN's 1x1 launches, the 16-row tiles), odd maps (stride-2 blocks round up, pixel-shuffle blocks double) --
    forward and every gradient against CPU autograd through the oracle's functions.  `-m gpu`: every second shape;
    `-m "gpu and sweep"`: all fourteen."""
    import random
    from mcquic_amd import nn as N
    rng = random.Random(17)
    for it in range(14):
        c = rng.choice([8, 12, 20, 32, 48, 64, 128, 192])
        n, h, w = rng.randint(1, 4), rng.randint(2, 20), rng.randint(2, 20)
        if which == "sample" and it % 2:
            continue
        x = _rand((n, c, h, w), 9500 + it)
        cases = [(N.ResidualBlock(c, c), R._rb, R.residual_block), (N.ResidualBlockWithStride(c, c), R._rb_stride, R.residual_block_with_stride),
                 (N.ResidualBlockShuffle(c, c), R._rb_shuffle, R.residual_block_shuffle), (N.AttentionBlock(c), R._attn, R.attention_block)]
        for mod, mk, fn in cases:
            sd = {}
            mk(sd, "", c, 9600 + it)
            mod.load_state_dict(sd, strict=True)
            params = {k: v.clone().requires_grad_() if v.is_floating_point() and v.dim() > 0 and "reparam" not in k else v for k, v in sd.items()}
            xr = x.clone().requires_grad_()
            y = fn(params, "", xr)
            gy = _rand(tuple(y.shape), 9700 + it)
            y.backward(gy)
            mod = mod.to(dev).train()
            xd = x.to(dev).requires_grad_()
            yd = mod(xd)
            what = f"#{it} {type(mod).__name__} c{c} n{n} {h}x{w}"
            _close(yd, y.detach(), 5e-6, what + " forward")
            yd.backward(gy.to(dev))
            _close(xd.grad, xr.grad, 2e-5, what + " dx")
            for name, p in mod.named_parameters():
                want = params[name].grad
                assert want is not None, name
                _close(p.grad, want, 2e-5, what + " d" + name)
