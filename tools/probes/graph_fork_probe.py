#!/usr/bin/env python3
"""Does ONE fork / join inside a captured hipGraph overlap a chain of tiny launches with a few fat ones?  (Round 2 found that
per-block forks -- hundreds of them, nested -- replay slower than one stream or crash the capture; this asks the question for
a single long-lived side stream.)  Chain = 120 dependent 3x3 convs on 8 x 128 x 8 x 8 maps (~12 us each, ~5 % of the chip);
fat = 8 grouped weight-gradient launches on 8 x 128 x 64 x 64 maps (~280 us each).  Prints eager / graph times for: chain,
fat, both on one stream, both with the fat launches on a side stream."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mcquic_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    x = torch.randn(8, 128, 8, 8, device=dev)
    packs = [ops.PackedConv(torch.randn(128, 128, 3, 3, device=dev) * 0.03, torch.randn(128, device=dev)) for _ in range(6)]
    xs = [torch.randn(8, 128, 64, 64, device=dev) for _ in range(4)]
    dys = [torch.randn(8, 128, 64, 64, device=dev) for _ in range(4)]
    side = torch.cuda.Stream()

    def chain():
        t = x
        for i in range(120):
            t = ops.conv2d(t, packs[i % 6], res=x)
        return t

    def fat():
        out = None
        for _ in range(8):
            out = ops.conv2d_wgrad_group(xs, dys, want_bias=True)
        return out

    def both_serial():
        return chain(), fat()

    def both_forked():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            f = fat()
        c = chain()
        main.wait_stream(side)
        return c, f

    def timed(fn, graph):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        if graph:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                keep = fn()
            run = g.replay
        else:
            run = fn
        run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            run()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / 5

    for name, fn in (("chain", chain), ("fat", fat), ("both, one stream", both_serial), ("both, fat on a side stream", both_forked)):
        print(f"{name:30s} eager {timed(fn, False):8.3f} ms   graph {timed(fn, True):8.3f} ms", flush=True)


if __name__ == "__main__":
    main()
