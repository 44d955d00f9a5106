#!/usr/bin/env python3
"""HISTORICAL probe (rounds 1-2): the `MCQ_STAMPS` hooks it reads were removed from csrc/conv_mfma.hip in round 3 -- check out
commit 0aa82a9 to build a library that carries them; the findings are in DESIGN.md section 4.
Where a conv tile's time goes outside the k-loop (kernel tuning aid; needs a library built with -DMCQ_STAMPS=1):

    hipcc ... -DMCQ_STAMPS=1 -o tools/variants/stamps.so mcquic_amd/csrc/*.hip mcquic_amd/csrc/rans.cpp
    MCQUIC_AMD_LIB=tools/variants/stamps.so python tools/probe_stamps.py

Every wave records s_memrealtime (100 MHz) at entry, with its operand ring requested, after its last MFMA and after its
last store was issued, plus HW_ID / XCC_ID.  Printed: the medians of those phases, the gap between two waves that follow
each other on one wave slot, and per SIMD how long 0 / 1 / 2 of its resident waves were inside the k-loop.
"""
import ctypes
import os
import sys
from collections import defaultdict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcquic_amd import ops, _lib  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = ctypes.CDLL(os.path.abspath(os.environ["MCQUIC_AMD_LIB"]))
    buf = torch.zeros(1 + 9 * (1 << 20), dtype=torch.int64, device=dev)
    assert lib.mcq_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
    shapes = [(32, 128, 128, 192, 128)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    for (n, cin, cout, h, w) in shapes:
        x = torch.randn(n, cin, h, w, device=dev)
        res = torch.randn(n, cout, h, w, device=dev)
        wino = os.environ.get("PROBE_WINOGRAD", "0") == "1"          # the opt-in Winograd instance instead of the direct form
        packs = [ops.PackedConv(torch.randn(cout, cin, 3, 3, device=dev) * 0.03, torch.randn(cout, device=dev), winograd=wino) for _ in range(3)]
        for flags in ("plain", "res"):
            kw = dict(res=res, dual_silu=True) if flags == "res" else {}
            kw["winograd"] = wino
            for i in range(3):
                ops.conv2d(x, packs[i], 1, **kw)
            torch.cuda.synchronize()
            buf[0] = 0
            torch.cuda.synchronize()
            ops.conv2d(x, packs[0], 1, **kw)
            torch.cuda.synchronize()
            cnt = int(buf[0].item())
            rec = buf[1:1 + 9 * cnt].cpu().numpy().reshape(cnt, 9)
            hw, xcc, t0, t1, t2, t3, da, db, dc = (rec[:, i] for i in range(9))
            tick = 0.01    # us
            print(f"\n== {n}x{cin}->{cout} {h}x{w} {flags}: {cnt} waves, kernel span {(t3.max() - t0.min()) * tick:.1f} us")
            for name, d in (("entry -> ring requested", t1 - t0), ("k-loop", t2 - t1), ("epilogue (to last store issued)", t3 - t2)):
                print(f"  {name:34s} median {np.median(d) * tick:8.2f} us   p10 {np.percentile(d, 10) * tick:8.2f}   p90 {np.percentile(d, 90) * tick:8.2f}")
            if dc.any():
                print(f"  band epilogue phases, summed over the bands: side loads issued {np.median(da) * tick:.2f} us, arithmetic (incl. waiting for them) "
                      f"{np.median(db) * tick:.2f} us, stores issued {np.median(dc) * tick:.2f} us")
            slot = (xcc.astype(np.int64) & 0xF) << 16 | (hw.astype(np.int64) & 0xFFFF)
            simd = slot >> 4
            by_slot = defaultdict(list)
            for i in range(cnt):
                by_slot[int(slot[i])].append(i)
            gaps = []
            for k, idx in by_slot.items():
                idx.sort(key=lambda i: t0[i])
                for a, b in zip(idx[:-1], idx[1:]):
                    gaps.append(t0[b] - t3[a])
            gaps = np.array(gaps)
            print(f"  wave slots seen {len(by_slot)}; waves per slot {cnt / len(by_slot):.2f}")
            print(f"  gap last store issued -> next wave's entry on the slot: median {np.median(gaps) * tick:.2f} us  p10 {np.percentile(gaps, 10) * tick:.2f}  p90 {np.percentile(gaps, 90) * tick:.2f}")
            by_simd = defaultdict(list)
            for i in range(cnt):
                by_simd[int(simd[i])].append(i)
            occ = np.zeros(8)
            span_total = 0.0
            for k, idx in by_simd.items():
                ev = []
                for i in idx:
                    ev.append((t1[i], 1))
                    ev.append((t2[i], -1))
                ev.sort()
                lo, hi = min(t0[i] for i in idx), max(t3[i] for i in idx)
                cur, last = 0, lo
                for (t, d) in ev:
                    occ[min(cur, 7)] += t - last
                    last, cur = t, cur + d
                occ[min(cur, 7)] += hi - last
                span_total += hi - lo
            print(f"  SIMDs seen {len(by_simd)}; share of SIMD time with k waves inside the k-loop: " +
                  "  ".join(f"{k}: {occ[k] / span_total * 100:.1f}%" for k in range(4)))
            # resident waves per SIMD at the same time (by [t0, t3] overlap)
            starts = np.sort(t0)
            print(f"  launch wave-front: first {((starts[:2048] - starts[0]) * tick).max():.1f} us for the first 2048 waves")


if __name__ == "__main__":
    main()
