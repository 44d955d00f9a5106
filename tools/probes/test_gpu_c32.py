"""conv_c32_kernel (csrc/conv_c32.h, round 6): the 3x3 stride-1 convolution between 32 channels -- Neon's width
(mcquic/modules/compressor.py:181-241) -- with the filter bank in registers and the input patch in LDS, against float64
F.conv2d and, bit for bit, against the general kernel's unsplit 32 x 32 tile (same k-order, same epilogue order)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_gpu_ops import _close, _rand

pytestmark = pytest.mark.gpu
C = 32
FORCE = 0x320             # mcq_conv_desc.tile: conv_c32_kernel on any map size

GEOMS = [(1, 16, 32), (2, 37, 45), (3, 64, 64), (1, 5, 3), (2, 16, 33), (1, 130, 70)]
FLAGS = ["plain", "silu_out", "res", "res_minus", "res_twin", "dsilu", "dsilu_res", "nobias"]     # (no input prologue: the patch reaches LDS by DMA)


@pytest.mark.parametrize("geom", GEOMS)
@pytest.mark.parametrize("flags", FLAGS)
def test_c32_against_float64_and_the_general_tile(dev, geom, flags):
    from mcquic_amd import ops
    n, h, w = geom
    x = _rand((n, C, h, w), 1, 2.0)
    wt = _rand((C, C, 3, 3), 2, 1.0 / np.sqrt(C * 9))
    b = None if flags == "nobias" else _rand((C,), 3, 0.1)
    side = _rand((n, C, h, w), 4)
    side2 = _rand((n, C, h, w), 5)
    kw = {}
    xin = x.double()
    if "silu_in" in flags:
        kw["silu_in"] = True
        xin = F.silu(xin)
    want = F.conv2d(xin, wt.double(), None if b is None else b.double(), padding=1)
    if "dsilu" in flags:
        kw["dsilu_mul"] = side2.to(dev)
        sg = torch.sigmoid(side2.double())
        want = want * (sg * (1 + side2.double() * (1 - sg)))
    if "res" in flags:
        kw["res"] = side.to(dev)
        if flags == "res_minus":
            kw["res_scale"] = -1.0
            want = want - side.double()
        else:
            want = want + side.double()
    if flags == "silu_out":
        kw["silu_out"] = True
        want = F.silu(want)
    if "twin" in flags:
        kw["dual_silu"] = True
    pk = ops.PackedConv(wt.to(dev), None if b is None else b.to(dev))
    got = ops.conv2d(x.to(dev), pk, 1, tile=FORCE, **kw)
    _close(got, want.float(), 3e-6, f"c32 {geom} {flags}")
    ref = ops.conv2d(x.to(dev), pk, 1, tile=0x11, **kw)
    assert torch.equal(got, ref), f"c32 {geom} {flags}: differs from the general kernel's 32 x 32 tile"
    if "twin" in flags:
        assert torch.equal(ops.silu_twin(got), ops.silu_twin(ref))
        _close(ops.silu_twin(got), F.silu(want).float(), 3e-6, f"c32 twin {geom} {flags}")


def test_c32_is_what_large_maps_run_and_small_ones_do_not(dev):
    """The library's own choice (tile 0): from 256 tiles of 16 x 32 pixels up; the results do not depend on the choice."""
    from mcquic_amd import ops
    wt = _rand((C, C, 3, 3), 2, 1.0 / np.sqrt(C * 9))
    pk = ops.PackedConv(wt.to(dev), _rand((C,), 3, 0.1).to(dev))
    for n, h, w in ((4, 256, 256), (1, 64, 64)):
        x = _rand((n, C, h, w), 7).to(dev)
        auto = ops.conv2d(x, pk, 1, res=x, dual_silu=True)
        forced = ops.conv2d(x, pk, 1, res=x, dual_silu=True, tile=FORCE)
        assert torch.equal(auto, forced) or n == 1            # (a small map may take a split tile: another summation order)
        _close(auto, forced.cpu(), 2e-6, f"auto vs forced {n}x{h}x{w}")
    with pytest.raises(RuntimeError):                          # other widths never take it
        ops.conv2d(_rand((1, 64, 16, 16), 1).to(dev), ops.PackedConv(_rand((64, 64, 3, 3), 2, 0.05).to(dev), None), 1, tile=FORCE)
