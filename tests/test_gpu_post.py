"""The 1x1 layer behind a 3x3 convolution INSIDE that convolution's launch (MCQ_CONV_POST_GDN / _IGDN / _GATE, round 6): GenDivNorm
after a strided convolution, InvGenDivNorm after a pixelShuffle3x3, the AttentionBlock's conv1x1 + gate after its side stack
(reference: mcquic/nn/gdn.py:67-91, mcquic/nn/blocks.py:98-122,141-159,281-288).

Each fused launch against (a) the CPU oracle's arithmetic in float64 of the same ops and (b) the two-launch form of this library
(the only difference: the 1x1 layer's summation order over the 128 channels and where its bias enters the sum), with the 128 x 32 wave
tile forced on small maps, ragged map sizes, and the blocks end to end on maps large enough for the library to
fuse on its own."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import mcquic_ref as R
from test_gpu_ops import _close, _rand

pytestmark = pytest.mark.gpu
C = 128


def _gdn_module(dev, inverse, seed=3):
    from mcquic_amd.nn import GenDivNorm, InvGenDivNorm
    sd = {}
    R._gdn_params(sd, "g.", C, seed=seed)
    mod = (InvGenDivNorm if inverse else GenDivNorm)(C)
    mod.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=True)
    return mod.to(dev).eval(), sd


@pytest.mark.parametrize("geom", [(2, 24, 32, 2), (1, 13, 37, 2), (2, 14, 18, 1), (3, 9, 70, 1)])
@pytest.mark.parametrize("silu_in", [False, True])
def test_conv_then_gdn_in_one_launch(dev, geom, silu_in, tile=0x41):
    from mcquic_amd import ops
    n, h, w, stride = geom
    gdn, sd = _gdn_module(dev, False)
    x = _rand((n, C, h, w), 1, 2.0)
    wt = _rand((C, C, 3, 3), 2, 1.0 / np.sqrt(C * 9))
    b = _rand((C,), 3, 0.1)
    pk = ops.PackedConv(wt.to(dev), b.to(dev))
    v = F.conv2d((F.silu(x) if silu_in else x).double(), wt.double(), b.double(), stride=stride, padding=1)
    want = R.gdn({k: t.double() for k, t in sd.items()}, "g.", v, False)
    got = ops.conv2d(x.to(dev), pk, stride, silu_in=silu_in, post_gdn=gdn.packed_post(), tile=tile)
    assert got.shape == want.shape
    _close(got, want.float(), 4e-6, f"conv+gdn {geom} tile={tile:#x}")
    two = gdn(ops.conv2d(x.to(dev), pk, stride, silu_in=silu_in))
    _close(got, two.cpu(), 2e-6, f"conv+gdn vs two launches {geom} tile={tile:#x}")


@pytest.mark.parametrize("geom", [(2, 12, 16), (1, 7, 21), (3, 5, 40)])
def test_shuffle_conv_then_igdn_in_one_launch(dev, geom, tile=0x41):
    """pixelShuffle3x3 -> InvGenDivNorm: the convolution's row tiles in sub-pixel-major order (Conv2d.packed_subpixel), the store to
    pixel (2 y + dy, 2 x + dx)."""
    from mcquic_amd import ops
    from mcquic_amd.nn.convs import pixelShuffle3x3
    n, h, w = geom
    igdn, sd = _gdn_module(dev, True, seed=5)
    up = pixelShuffle3x3(C, C, 2).to(dev).eval()
    x = _rand((n, C, h, w), 4, 2.0)
    wt, b = up[0].weight.detach().cpu(), up[0].bias.detach().cpu()
    v = F.pixel_shuffle(F.conv2d(F.silu(x).double(), wt.double(), b.double(), padding=1), 2)
    want = R.gdn({k: t.double() for k, t in sd.items()}, "g.", v, True)
    got = ops.conv2d(x.to(dev), up[0].packed_subpixel(), 1, silu_in=True, shuffle2=True, post_igdn=igdn.packed_post(), tile=tile)
    assert tuple(got.shape) == (n, C, 2 * h, 2 * w)
    _close(got, want.float(), 4e-6, f"shuffle+igdn {geom} tile={tile:#x}")
    two = igdn(up(x.to(dev), silu_in=True))
    _close(got, two.cpu(), 2e-6, f"shuffle+igdn vs two launches {geom} tile={tile:#x}")


@pytest.mark.parametrize("geom", [(2, 24, 32), (1, 13, 37)])
def test_conv_then_gate_in_one_launch(dev, geom, tile=0x41):
    """out = a * sigmoid(conv1x1(conv3x3(t) + b)) + x with its SiLU twin: the side stack's last convolution, the 1x1 layer and the
    gate of an AttentionBlock."""
    from mcquic_amd import ops
    n, h, w = geom
    t = _rand((n, C, h, w), 11)
    bres = _rand((n, C, h, w), 12)
    a = _rand((n, C, h, w), 13)
    xid = _rand((n, C, h, w), 14)
    w3 = _rand((C, C, 3, 3), 15, 1.0 / np.sqrt(C * 9))
    b3 = _rand((C,), 16, 0.1)
    w1 = _rand((C, C, 1, 1), 17, 1.0 / np.sqrt(C))
    b1 = _rand((C,), 18, 0.1)
    v = F.conv2d(t.double(), w3.double(), b3.double(), padding=1) + bres.double()
    want = a.double() * torch.sigmoid(F.conv2d(v, w1.double(), b1.double())) + xid.double()
    pk3 = ops.PackedConv(w3.to(dev), b3.to(dev))
    post = ops.PackedPost(w1.to(dev), b1.to(dev))
    got = ops.conv2d(t.to(dev), pk3, 1, res=bres.to(dev), post_gate=post, gate_mul=a.to(dev), gate_id=xid.to(dev), dual_silu=True, tile=tile)
    _close(got, want.float(), 4e-6, f"conv+gate {geom} tile={tile:#x}")
    _close(ops.silu_twin(got), F.silu(want).float(), 4e-6, f"conv+gate twin {geom} tile={tile:#x}")
    pk1 = ops.PackedConv(w1.to(dev), b1.to(dev))
    bb = ops.conv2d(t.to(dev), pk3, 1, res=bres.to(dev))
    two = ops.conv2d(bb, pk1, 1, gate_mul=a.to(dev), gate_id=xid.to(dev), dual_silu=True)
    _close(got, two.cpu(), 2e-6, f"conv+gate vs two launches {geom} tile={tile:#x}")


def test_post_refusals_and_the_size_rule(dev):
    """What the fused form does not take is refused, not mis-run: other channel counts, 1x1 producers, several flags at once; small
    maps (the 3x3 layer is split over waves there) are declined by mcq_conv2d_post_ok and the blocks then launch the 1x1 layer."""
    from mcquic_amd import ops, _lib
    lib = _lib.load()
    G, I, T, SH = ops.CONV_POST_GDN, ops.CONV_POST_IGDN, ops.CONV_POST_GATE, ops.CONV_SHUFFLE2
    assert lib.mcq_conv2d_post_ok(32, 128, 384, 256, 128, 3, 2, G) == 32 * 192 * 128 // 32      # 128 x 32 wave tiles
    assert lib.mcq_conv2d_post_ok(32, 128, 192, 128, 512, 3, 1, I | SH) == 4 * 32 * 192 * 128 // 32
    assert lib.mcq_conv2d_post_ok(32, 128, 192, 128, 128, 3, 1, T) == 32 * 192 * 128 // 32
    assert lib.mcq_conv2d_post_ok(1, 128, 192, 128, 512, 3, 1, I | SH) == 3072       # one image: 768 pixel blocks x 4 row tiles
    assert lib.mcq_conv2d_post_ok(32, 128, 24, 16, 128, 3, 2, G) == 96               # a 12x8 map: far below ops._POST_MIN_WAVES
    assert lib.mcq_conv2d_post_ok(32, 192, 384, 256, 192, 3, 2, G) == 0              # model No. 12: 192 channels do not fit one wave tile
    assert lib.mcq_conv2d_post_ok(32, 128, 384, 256, 128, 1, 1, G) == 0
    assert lib.mcq_conv2d_post_ok(32, 128, 384, 256, 128, 3, 2, G | T) == 0
    assert lib.mcq_conv2d_post_ok(32, 128, 192, 128, 512, 3, 1, G | SH) == 0
    pkx = ops.PackedConv(_rand((C, C, 3, 3), 2, 0.03).to(dev), None)
    assert ops.post_ok(torch.empty((32, C, 384, 256), device=dev), pkx, 2, "gdn") == 1        # the library's own tile
    assert ops.post_ok(torch.empty((1, C, 384, 256), device=dev), pkx, 2, "gdn") == 0x41      # 768 tiles: forced, one image's large maps
    assert ops.post_ok(torch.empty((1, C, 96, 64), device=dev), pkx, 2, "gdn") == 0
    gdn, _ = _gdn_module(dev, False)
    x = _rand((1, C, 8, 8), 1).to(dev)
    pk = ops.PackedConv(_rand((C, C, 3, 3), 2, 0.03).to(dev), None)
    with pytest.raises(RuntimeError):
        ops.conv2d(x, pk, 1, post_gdn=gdn.packed_post())                          # too small a map and no forced tile
    with pytest.raises(RuntimeError):
        ops.conv2d(x, pk, 1, post_gdn=gdn.packed_post(), silu_out=True, tile=0x41)
    with pytest.raises(ValueError):
        ops.PackedPost(_rand((64, 64, 1, 1), 3).to(dev), None)


@pytest.mark.parametrize("kind", ["stride", "shuffle", "attention"])
def test_blocks_fuse_on_large_maps_and_match_the_oracle(dev, kind, monkeypatch):
    """The blocks end to end at sizes where ops.post_ok says yes (the launches really are the fused ones: a spy on ops.conv2d sees
    the post_* option), against the CPU oracle and against the same block with MCQUIC_AMD_FUSE_POST off."""
    from mcquic_amd import nn as N, ops
    cases = {"stride": (N.ResidualBlockWithStride(C, C), R._rb_stride, R.residual_block_with_stride, (8, C, 192, 192), "post_gdn"),
             "shuffle": (N.ResidualBlockShuffle(C, C), R._rb_shuffle, R.residual_block_shuffle, (4, C, 64, 64), "post_igdn"),
             "attention": (N.AttentionBlock(C), R._attn, R.attention_block, (8, C, 96, 96), "post_gate")}
    mod, mk, fn, shape, opt = cases[kind]
    sd = {}
    mk(sd, "b.", C, 7)
    mod.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=True)
    mod = mod.to(dev).eval()
    x = _rand(shape, 41)
    seen = []
    real = ops.conv2d

    def spy(xx, ww, stride=1, **fused):
        seen.extend(k for k in fused if k.startswith("post_"))
        return real(xx, ww, stride, **fused)
    monkeypatch.setattr(ops, "conv2d", spy)
    got = mod(x.to(dev))
    assert seen == [opt], f"expected one fused launch ({opt}), saw {seen}"
    monkeypatch.setattr(ops, "_FUSE_POST", set())
    seen.clear()
    plain = mod(x.to(dev))
    assert seen == []
    _close(got, plain.cpu(), 4e-6, f"{kind}: fused vs separate launches")
    want = fn(sd, "b.", x[:2])
    _close(got[:2], want, 6e-6, f"{kind}: fused vs oracle")
    if ops.silu_twin(plain) is not None:
        assert ops.silu_twin(got) is not None
        _close(ops.silu_twin(got), ops.silu_twin(plain).cpu(), 4e-6, f"{kind}: twin")
