#!/usr/bin/env python3
"""Split-bf16 prototype of ONE layer: 128 -> 128 3x3 stride 1 on 32 x 192 x 128 -- see bf16x3_conv.hip.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -mllvm -pragma-unroll-threshold=1000000 tools/probes/bf16x3/bf16x3_conv.hip \
          -o tools/probes/bf16x3/libbf16x3_probe.so      (without the unroll threshold the staging ring is demoted to LDS)
    python tools/probes/bf16x3/bf16x3_probe.py

Reports: (1) accuracy against a float64 convolution on a small case -- max and mean error of the split product beside the
errors of the product path's exact-fp32 kernel (mcq_conv2d_f32) and of the x0 w0-only (plain bf16) product; (2) time of the
six-term kernel, of the one-term kernel (the loop's own speed of light: same loads per MFMA, a sixth of the MFMAs), of the split
pass, and of the direct fp32 kernel on the same tensors."""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
from mcquic_amd import ops  # noqa: E402

lib = ctypes.CDLL(os.path.join(HERE, "libbf16x3_probe.so"))
lib.mcq_probe_split_bf16x3.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
lib.mcq_probe_conv3x3_bf16x3.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 6 + [ctypes.c_void_p]


def split3(t):
    h = t.bfloat16()
    r1 = t - h.float()
    m = r1.bfloat16()
    l = (r1 - m.float()).bfloat16()
    return h, m, l


def pack_weights(w):
    """[Cout, Cin, 3, 3] float32 -> ws[plane][T][s][tap][band][lane = 32 khalf + row][8] bf16 (as int16)."""
    cout, cin = w.shape[:2]
    planes = []
    for part in split3(w):
        v = part.reshape(cout // 128, 4, 32, cin // 16, 2, 8, 9)          # T, band, row, s, khalf, e, tap
        planes.append(v.permute(0, 3, 6, 1, 4, 2, 5).contiguous())          # T, s, tap, band, khalf, row, e
    return torch.stack(planes).view(torch.int16).contiguous()


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def split_acts(x):
    n, c, h, w = x.shape
    out = torch.empty((3, n, c // 8, h, w, 8), dtype=torch.int16, device=x.device)
    assert lib.mcq_probe_split_bf16x3(x.data_ptr(), out.data_ptr(), n, c, h * w, stream()) == 0
    return out


def conv(xs, ws, bias, shape, cout, terms=6):
    n, c, h, w = shape
    y = torch.empty((n, cout, h, w), dtype=torch.float32, device=xs.device)
    assert lib.mcq_probe_conv3x3_bf16x3(xs.data_ptr(), ws.data_ptr(), bias.data_ptr(), y.data_ptr(), n, c, h, w, cout, terms, stream()) == 0
    return y


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    # ---- accuracy, small case, against float64 --------------------------------------------------------------------
    n, c, h, w, cout = 2, 128, 16, 64, 128
    x = (torch.randn((n, c, h, w), generator=g)).to(dev)
    wt = (torch.randn((cout, c, 3, 3), generator=g) * 0.03).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    want = torch.nn.functional.conv2d(x.double().cpu(), wt.double().cpu(), b.double().cpu(), padding=1)
    xs_t = torch.stack([p.reshape(n, c // 8, 8, h, w).permute(0, 1, 3, 4, 2) for p in split3(x)]).contiguous().view(torch.int16)
    xs = split_acts(x)
    print("split kernel == torch split:", bool(torch.equal(xs, xs_t)))
    ws = pack_weights(wt)
    scale = float(want.abs().max())
    for name, y in (("bf16 x 3 (six terms), v2 patch+weights in LDS", conv(xs, ws, b, x.shape, cout, 6)), ("bf16 x 3 (six terms), v1 weights in LDS", conv(xs, ws, b, x.shape, cout, 61)), ("bf16 x 3 (six terms), v0", conv(xs, ws, b, x.shape, cout, 60)),
                    ("bf16 x0 w0 only", conv(xs, ws, b, x.shape, cout, 1)),
                    ("exact fp32 MFMA (mcq_conv2d_f32)", ops.conv2d(x, ops.PackedConv(wt, b)))):
        err = (y.double().cpu() - want).abs()
        print(f"{name:36s} max |err| {float(err.max()):.3e}  mean |err| {float(err.mean()):.3e}   (output scale {scale:.2f})")
    # ---- speed, the layer itself ----------------------------------------------------------------------------------------
    n, c, h, w, cout = 32, 128, 192, 128, 128
    x = torch.randn((n, c, h, w), device=dev)
    wt = torch.randn((cout, c, 3, 3), device=dev) * 0.03
    b = torch.randn(cout, device=dev)
    ws = pack_weights(wt)
    xs = split_acts(x)
    pk = ops.PackedConv(wt, b)
    flops = 2.0 * n * h * w * cout * c * 9
    t6 = timed(lambda: conv(xs, ws, b, x.shape, cout, 6))
    t1 = timed(lambda: conv(xs, ws, b, x.shape, cout, 1))
    t60 = timed(lambda: conv(xs, ws, b, x.shape, cout, 60))
    t10 = timed(lambda: conv(xs, ws, b, x.shape, cout, 10))
    t61 = timed(lambda: conv(xs, ws, b, x.shape, cout, 61))
    t11 = timed(lambda: conv(xs, ws, b, x.shape, cout, 11))
    print(f"  v0 (all operands from global memory): six terms {t60:8.1f} us, one term {t10:8.1f} us")
    print(f"  v1 (weights through LDS)            : six terms {t61:8.1f} us, one term {t11:8.1f} us")
    ts = timed(lambda: split_acts(x))
    td = timed(lambda: ops.conv2d(x, pk))
    print(f"layer 32 x 128 -> 128, 192 x 128: direct fp32 {td:8.1f} us ({flops / td / 1e6:6.1f} TFLOP/s)")
    for code, what in ((601, "MFMAs only inside the loop"), (602, "no workgroup barrier"), (603, "no global loads in the loop")):
        print(f"  v2 ablation, {what:30s}: {timed(lambda: conv(xs, ws, b, x.shape, cout, code)):8.1f} us   (results wrong by construction)")
    print(f"  v2 (input patch + weights in LDS):")
    print(f"  bf16 x 3, six terms   {t6:8.1f} us  = {td / t6:4.2f}x direct  ({6 * flops / t6 / 1e6:7.1f} TFLOP/s of bf16 MFMA work, peak ~2500)")
    print(f"  one term (x0 w0)      {t1:8.1f} us  ({flops / t1 / 1e6:7.1f} TFLOP/s)")
    print(f"  split / re-pack pass  {ts:8.1f} us  ({(4 + 6) * x.numel() / ts / 1e6:6.2f} TB/s)  -> with it {td / (t6 + ts):4.2f}x direct")


if __name__ == "__main__":
    main()
