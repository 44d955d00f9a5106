#!/usr/bin/env python3
"""Where the training step's MFMA time goes, launch shape by launch shape (kernel tuning aid for BASELINE configs[4]).

One forward + backward of Compressor(128, 2, [8192, 2048, 512]) on 8 x 256x256 crops is traced at the op layer: every
conv launch (single / multi-problem), weight-gradient launch (single / grouped) with its geometry and flag set.  Each
DISTINCT signature is then replayed in isolation -- 20 back-to-back launches between two events, cycling through distinct
weight tensors -- and listed with its count per step, its time, the fp32-MFMA ideal of its FLOPs and the time it loses
per step against that ideal, worst first.

    python tools/train_conv_census.py [--batch 8 --crop 256 --iters 20 --eval]     (--eval: the inference encode+decode instead)
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PEAK = 157.3e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--crop", type=int, default=256)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--eval", action="store_true", help="trace inference encode+decode instead of the training step")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    os.environ["MCQUIC_AMD_BRANCH_STREAMS"] = "0"
    from mcquic_amd import Compressor, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(3407)
    model = Compressor(128, 2, [8192, 2048, 512]).to(dev)
    h, w = (a.height or a.crop), (a.width or a.crop)
    x = (torch.rand((a.batch, 3, h, w)) * 2 - 1).to(dev)

    sigs = collections.Counter()
    orig_multi, orig_conv, orig_wg, orig_wgg, orig_desc = ops.conv2d_multi, ops.conv2d, ops.conv2d_wgrad, ops.conv2d_wgrad_group, ops._conv_desc
    pending = []

    def desc(xx, wt, stride=1, **kw):
        out = orig_desc(xx, wt, stride, **kw)
        d = out[0]
        pending.append(("conv", d.N, d.Cin, d.H, d.W, d.Cout, d.ksize, d.stride, int(d.flags), float(d.res_scale)))
        return out

    def conv(xx, wt, stride=1, **kw):
        del pending[:]
        y = orig_conv(xx, wt, stride, **kw)
        sigs[pending[-1] + (1,)] += 1
        return y

    def multi(xs, ws, stride=1, per_problem=None, **shared):
        ops.conv2d = orig_conv
        del pending[:]
        try:
            ys = orig_multi(xs, ws, stride, per_problem=per_problem, **shared)
        finally:
            ops.conv2d = conv
        cap = 4
        for lo in range(0, len(xs), cap):
            sigs[pending[lo] + (min(cap, len(xs) - lo),)] += 1
        return ys

    def wg(xx, dy, ksize, stride, square_x=False, want_bias=False):
        sigs[("wgrad", xx.shape[0], xx.shape[1], xx.shape[2], xx.shape[3], dy.shape[1], ksize, stride, int(square_x), 0.0, 1)] += 1
        return orig_wg(xx, dy, ksize, stride, square_x=square_x, want_bias=want_bias)

    def wgg(xs, dys, want_bias=True):
        ops.conv2d_wgrad = orig_wg
        try:
            out = orig_wgg(xs, dys, want_bias=want_bias)
        finally:
            ops.conv2d_wgrad = wg
        xx, dy = xs[0], dys[0]
        for lo in range(0, len(xs), 16):
            sigs[("wgrad", xx.shape[0], xx.shape[1], xx.shape[2], xx.shape[3], dy.shape[1], 3, 1, 0, 0.0, min(16, len(xs) - lo))] += 1
        return out

    def one_step():
        if a.eval:
            with torch.no_grad():
                model.decode(model.encode(x))
        else:
            for p in model.parameters():
                p.grad = None
            xHat, _, _, _ = model(x)
            torch.nn.functional.mse_loss(xHat, x).backward()

    model.eval() if a.eval else model.train()
    one_step()                                              # warm-up (packs)
    ops._conv_desc, ops.conv2d, ops.conv2d_multi, ops.conv2d_wgrad, ops.conv2d_wgrad_group = desc, conv, multi, wg, wgg
    one_step()
    ops._conv_desc, ops.conv2d, ops.conv2d_multi, ops.conv2d_wgrad, ops.conv2d_wgrad_group = orig_desc, orig_conv, orig_multi, orig_wg, orig_wgg
    torch.cuda.synchronize()

    FL = {0x1: "silu_in", 0x2: "sq_in", 0x4: "silu_out", 0x8: "res", 0x10: "gdn", 0x20: "igdn", 0x40: "gate", 0x80: "shuf", 0x100: "twin",
          0x200: "mul", 0x400: "dsilu", 0x800: "wino", 0x1000: "wino2d", 0x20000: "4taps"}
    rows = []
    for sig, count in sigs.items():
        kind, n, cin, hh, ww, cout, ks, stride, flags, res_scale, nprob = sig
        pad = ks // 2
        ho, wo = (hh + 2 * pad - ks) // stride + 1, (ww + 2 * pad - ks) // stride + 1
        flops = 2.0 * n * ho * wo * cout * cin * ks * ks * nprob
        nw = 6
        if kind == "conv":
            xs = [torch.randn((n, cin, hh, ww), device=dev) for _ in range(nprob)]
            packs = [ops.PackedConv(torch.randn((cout, cin, ks, ks), device=dev) * 0.03, torch.randn(cout, device=dev)) for _ in range(nw * nprob)]
            if flags & 0x20000:                # MCQ_CONV_TAPS_LR (the input-gradient stream of a stride-2 layer): 4 of the 9 taps carry work
                flops *= 4.0 / 9.0
                for pk in packs:
                    pk.lr_taps = True
            shuf = bool(flags & 0x80)
            oshape = (n, cout // 4, 2 * ho, 2 * wo) if shuf else (n, cout, ho, wo)
            side = torch.randn(oshape, device=dev)
            kw = dict(silu_in=bool(flags & 1), square_in=bool(flags & 2), silu_out=bool(flags & 4), shuffle2=shuf, dual_silu=bool(flags & 0x100))
            pp = {}
            if flags & 0x8:
                pp["res"] = side
                kw["res_scale"] = res_scale
            for bit, key in ((0x10, "gdn_mul"), (0x20, "igdn_mul"), (0x40, "gate_mul"), (0x200, "mul"), (0x400, "dsilu_mul")):
                if flags & bit:
                    pp[key] = side
            if flags & 0x40:
                pp["gate_id"] = side

            def run(i):
                if nprob == 1:
                    ops.conv2d(xs[0], packs[i % nw], stride, **kw, **pp)
                else:
                    ops.conv2d_multi(xs, packs[(i % nw) * nprob:(i % nw + 1) * nprob], stride, per_problem=[pp] * nprob, **kw)
        else:
            xs = [torch.randn((n, cin, hh, ww), device=dev) for _ in range(nprob)]
            dys = [torch.randn((n, cout, ho, wo), device=dev) for _ in range(nprob)]

            def run(i):
                if nprob == 1:
                    ops.conv2d_wgrad(xs[0], dys[0], ks, stride, square_x=bool(flags), want_bias=True)
                else:
                    ops.conv2d_wgrad_group(xs, dys, want_bias=True)
        for i in range(3):
            run(i)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(a.iters):
            run(i)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / a.iters
        ideal = flops / PEAK * 1e6
        names = "+".join(v for b, v in FL.items() if flags & b) if kind == "conv" else ("sq" if flags else "")
        rows.append((count * (us - ideal), count, us, ideal, f"{kind} x{nprob} n{n} {cin}->{cout} {hh}x{ww} k{ks}s{stride} {names}"))
    rows.sort(reverse=True)
    tot_t = sum(r[1] * r[2] for r in rows)
    tot_i = sum(r[1] * r[3] for r in rows)
    print(f"# {len(rows)} distinct launch signatures, {sum(r[1] for r in rows)} launches per step; isolated time {tot_t / 1e3:.2f} ms, "
          f"fp32-MFMA ideal {tot_i / 1e3:.2f} ms ({tot_i / tot_t:.3f})")
    print(f"{'lost_us':>9} {'count':>5} {'us':>9} {'ideal_us':>9} {'eff':>6}  signature")
    for lost, count, us, ideal, name in rows[:a.top]:
        print(f"{lost:9.1f} {count:5d} {us:9.1f} {ideal:9.1f} {ideal / us:6.3f}  {name}")


if __name__ == "__main__":
    main()
