#!/usr/bin/env python3
"""Why does `secondary.batch1` of the bench line (6.8-6.9 ms) differ from `bench.py --batch 1 --graphs` alone (5.9 ms)?
(VERDICT r3 weak #4.)  One process, the bench's model and image; the same 30-replay measurement at several points:

    fresh            right after building the model (what the stand-alone run measures)
    after_b32        after 3 eager encode+decode steps of the 32-image batch (allocator high-water mark, hot chip)
    after_wino       after the opt-in Winograd round trip of the secondary (set_winograd(2), 3 steps, set_winograd(0))
    recaptured       graphs dropped and captured again in that state
    cooled           the same graphs after 10 s of idle

each with rocm-smi's current sclk beside it."""
import json
import subprocess
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from mcquic_amd import ops
from mcquic_amd.utils import synthetic


def sclk():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out)
        card = next(iter(d.values()))
        return {k: v for k, v in card.items() if "sclk" in k.lower() or "mclk" in k.lower()}
    except Exception as exc:       # noqa: BLE001
        return repr(exc)[:80]


def timed(fn, steps=30, warmup=3):
    for _ in range(warmup):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) / steps, 3), round((time.perf_counter() - t0) / steps * 1e3, 3)


def main():
    dev = torch.device("cuda", 0)
    model = synthetic.bench_model().to(dev)
    x = synthetic.bench_images(0, 32).to(dev)
    x1 = x[:1].contiguous()
    res = {}

    def b1():
        return model.decode(model.encode(x1))
    model.enableGraphs(True)
    res["fresh"] = (timed(b1), sclk())
    res["fresh_again"] = (timed(b1, 100), sclk())
    old = (model._graphs, model._graphStamp)
    model.enableGraphs(False)
    for _ in range(3):
        model.decode(model.encode(x))
    torch.cuda.synchronize()
    model._graphs, model._graphStamp = old
    res["old_graphs_after_b32"] = (timed(b1), sclk())
    model.enableGraphs(True)
    res["after_b32"] = (timed(b1), sclk())
    model.enableGraphs(False)
    ops.set_winograd(2)
    for _ in range(3):
        model.decode(model.encode(x))
    torch.cuda.synchronize()
    ops.set_winograd(0)
    model.enableGraphs(True)
    res["after_wino"] = (timed(b1), sclk())
    res["after_wino_100"] = (timed(b1, 100), sclk())
    model.enableGraphs(False)
    torch.cuda.empty_cache()
    model.enableGraphs(True)
    res["recaptured_after_empty_cache"] = (timed(b1), sclk())
    time.sleep(10)
    res["cooled_10s"] = (timed(b1), sclk())
    # eager (no graphs) for reference
    model.enableGraphs(False)
    res["eager"] = (timed(b1), sclk())
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
