#!/usr/bin/env python3
"""VERDICT r4 "next" #1a asks for ONE launch holding a block's input-gradient convolution and the weight-gradient row walk of the same
dy, each filling the other's single-round tail.  What such a launch could return is bounded by what the hardware does with the two
kernels resident TOGETHER -- which two streams show without writing the merged kernel: N pairs (input-gradient conv, row walk) on
8 x 128 x S x S maps, one stream against the walks on a side stream (eager: no capture effects).  Prints ms per pair."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mcquic_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream()
    for S in (128, 64):
        x = torch.randn(8, 128, S, S, device=dev)
        dy = torch.randn(8, 128, S, S, device=dev)
        w = torch.randn(128, 128, 3, 3, device=dev) * 0.03
        pack = ops.pack_convs([w], dgrad=True)[0]
        N = 8

        def dgrad():
            return ops.conv2d(dy, pack, dsilu_mul=x)

        def wgrad():
            with ops.wgrad_now():
                return ops.conv2d_wgrad(x, dy, 3, 1, want_bias=True)

        def only_d():
            for _ in range(N):
                dgrad()

        def only_w():
            for _ in range(N):
                wgrad()

        def serial():
            for _ in range(N):
                dgrad()
                wgrad()

        def forked():
            main_s = torch.cuda.current_stream()
            side.wait_stream(main_s)
            for _ in range(N):
                dgrad()
                with torch.cuda.stream(side):
                    wgrad()
            main_s.wait_stream(side)

        def timed(fn):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / 5 / N * 1000

        flops = 2 * 8 * S * S * 128 * 128 * 9
        ideal = flops / 157.3e12 * 1e6
        print(f"8x128x{S}x{S}: ideal {ideal:.1f} us per kernel | dgrad {timed(only_d):.1f} us | wgrad {timed(only_w):.1f} us | "
              f"pair, one stream {timed(serial):.1f} us | pair, walk on a side stream {timed(forked):.1f} us", flush=True)


if __name__ == "__main__":
    main()
