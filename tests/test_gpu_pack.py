"""The packed operand stream of a 3x3 weight (include/mcquic_hip.h: mcq_pack_conv_weight_f32 and its grouped / masked forms)
against a numpy restatement of the layout the conv kernel reads (csrc/conv_mfma.hip: [tile][k-step][lane][band] per copy,
k-step = channel pair * 9 + tap, lane = 32 * (ci & 1) + (co & 31); 16 zero steps behind every copy), and the masked re-pack a
captured training step uses (parallel.GraphedTrainStep): only the named copies change."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TAIL = 32       # zero k-steps after each copy (MCQ_TAIL_STEPS; 16 until ABI 8)


def _expected_sections(w):
    """[(offset, array)] of the 128- / 64- / 32-row copies of w [Cout, Cin, 3, 3] (forward mode)."""
    cout, cin = w.shape[:2]
    S = (cin + 1) // 2
    TP = S * 9
    wz = np.zeros((((cout + 127) // 128) * 128, 2 * S, 9), dtype=np.float32)
    wz[:cout, :cin] = w.reshape(cout, cin, 9)
    out, off = [], 0
    for b in (4, 2, 1):
        ntile = (cout + 32 * b - 1) // (32 * b)
        sec = np.zeros((ntile * TP + TAIL, 64, b), dtype=np.float32)
        for tile in range(ntile):
            for q in range(b):
                rows = wz[tile * 32 * b + 32 * q: tile * 32 * b + 32 * q + 32]             # [32 co, 2S ci, 9]
                blk = rows.reshape(32, S, 2, 9).transpose(1, 3, 2, 0).reshape(TP, 64)      # [s * 9 + tap][32 * (ci & 1) + co]
                sec[tile * TP: (tile + 1) * TP, :, q] = blk
        out.append((off, sec.reshape(-1)))
        off += sec.size
    return out, off


@pytest.mark.parametrize("shape", [(128, 128), (40, 72), (512, 128), (128, 64), (96, 50)])
def test_packed_stream_layout(dev, shape):
    from mcquic_amd import ops
    cout, cin = shape
    w = torch.randn(cout, cin, 3, 3, generator=torch.Generator().manual_seed(cout + cin))
    pk = ops.PackedConv(w.to(dev), None)
    got = pk.wp.cpu().numpy()
    want, end = _expected_sections(w.numpy())
    for off, sec in want:
        assert np.array_equal(got[off: off + sec.size], sec)
    # what follows the three copies is the 16x16-tile order of the small-launch kernel (layers with 64 / 128 input channels)
    assert got.size >= end
    multi = ops.pack_convs([w.to(dev), (2 * w).to(dev)], None)
    assert torch.equal(multi[0].wp, pk.wp)
    assert torch.equal(multi[1].wp, ops.PackedConv((2 * w).to(dev), None).wp)


def test_masked_repack_touches_only_named_copies(dev):
    from mcquic_amd import _lib, ops
    lib = _lib.load()
    cout = cin = 128
    g = torch.Generator().manual_seed(3)
    w_old, w_new = torch.randn(cout, cin, 3, 3, generator=g).to(dev), torch.randn(cout, cin, 3, 3, generator=g).to(dev)
    old, new = ops.PackedConv(w_old, None).wp, ops.PackedConv(w_new, None).wp
    want, end = _expected_sections(w_new.cpu().numpy())
    bounds = [off for off, _ in want] + [end, old.numel()]                     # four copies: [b4, b2, b1, t16]
    for mask in (1, 2, 4, 8, 5, 15, 0):
        buf = old.clone()
        src = (ctypes.c_void_p * 1)(w_new.data_ptr())
        dst = (ctypes.c_void_p * 1)(buf.data_ptr())
        mk = (ctypes.c_uint8 * 1)(mask)
        rc = lib.mcq_pack_conv_weight_multi_masked_f32(src, dst, mk, 1, cout, cin, 3, 0, 1, 1.0, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        eff = mask if mask else 15                                            # 0 = unknown = everything
        for bit in range(4):
            lo, hi = bounds[bit], bounds[bit + 1]
            ref = new if (eff >> bit) & 1 else old
            assert torch.equal(buf[lo:hi], ref[lo:hi]), (mask, bit)


def test_section_trace_records_the_copy_a_launch_reads(dev):
    from mcquic_amd import ops
    w = torch.randn(128, 128, 3, 3, generator=torch.Generator().manual_seed(1)).to(dev)
    pk = ops.PackedConv(w, None)
    ops.section_trace(True)
    try:
        ops.conv2d(torch.randn(8, 128, 64, 64, device=dev), pk)                 # a big map: one of the 32-row-band copies
        big = ops.sections_used(pk)
        ops.conv2d(torch.randn(1, 128, 4, 4, device=dev), pk)                   # a tiny map: the small-launch kernel's order
        both = ops.sections_used(pk)
    finally:
        ops.section_trace(False)
    assert big in (1, 2, 4) and both == (big | 8), (big, both)
    ops.section_trace(True)
    ops.section_trace(False)
    assert ops.sections_used(pk) == 0                                          # starting a trace clears the records
