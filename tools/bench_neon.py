#!/usr/bin/env python3
"""`secondary.neon` of bench.py on its own (one JSON line), for both denseNorm settings and a batch sweep of the training step's
peak memory -- the measurement behind the `checkpoint_wrapper` decision (mcquic/modules/compressor.py:230-231).
    python tools/bench_neon.py [--side 512] [--train-batches 4,8,16]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", type=int, default=512)
    ap.add_argument("--train-batches", default="4,8")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    out = {}
    for dense in (True, False):
        for tb in [int(v) for v in a.train_batches.split(",")]:
            r = bench.neon_figures(dev, dense=dense, train_batch=tb, infer_batch=max(8, tb), side=a.side)
            out[f"denseNorm={dense},train_batch={tb}"] = r
            print(json.dumps({f"denseNorm={dense},train_batch={tb}": r}), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/bench_neon.json", "w"), indent=1)


if __name__ == "__main__":
    main()
