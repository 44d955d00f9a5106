#!/usr/bin/env python3
"""Calibration of the FETCH_SIZE / WRITE_SIZE PMC counters against kernels whose HBM traffic is known.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/fs_f -o fs -- python tools/probe_fetch_size.py run
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/fs_w -o fs -- python tools/probe_fetch_size.py run
    python tools/probe_fetch_size.py report gpurun_out/fs_f/fs_results.db gpurun_out/fs_w/fs_results.db

Each case streams tensors much larger than the 256 MB MALL exactly once, so algorithmic bytes = HBM bytes.
"""
import os
import sqlite3
import sys
from collections import defaultdict

CASES = [  # name, kernel-name fragment, algorithmic read bytes, write bytes
    ("add_f32 (global dwordx4 loads) 2 x 1.6 GB -> 1.6 GB", "add_kernel", 2 * 32 * 128 * 384 * 256 * 4, 32 * 128 * 384 * 256 * 4),
    ("conv 1x1 128->32 plain @384x256 (buffer_load_dword per lane)", "conv_mfma_kernel<1", 32 * 128 * 384 * 256 * 4, 32 * 32 * 384 * 256 * 4),
    ("conv 3x3 128->128 plain @384x256", "conv_mfma_kernel<4, 2, 0, 9", 32 * 128 * 384 * 256 * 4, 32 * 128 * 384 * 256 * 4),
]


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mcquic_amd import ops
    dev = torch.device("cuda:0")
    x = torch.randn(32, 128, 384, 256, device=dev)
    y = torch.randn(32, 128, 384, 256, device=dev)
    p1 = ops.PackedConv(torch.randn(32, 128, 1, 1, device=dev) * 0.05, torch.zeros(32, device=dev))
    p3 = ops.PackedConv(torch.randn(128, 128, 3, 3, device=dev) * 0.03, torch.zeros(128, device=dev))
    for _ in range(3):
        ops.add(x, y)
        ops.conv2d(x, p1)
        ops.conv2d(x, p3)
    torch.cuda.synchronize()


def report(paths):
    vals = defaultdict(lambda: defaultdict(list))
    for path in paths:
        cur = sqlite3.connect(path).cursor()
        per = defaultdict(float)
        names = {}
        for did, name, cname, val in cur.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection"):
            per[(did, cname)] += val
            names[did] = name
        for (did, cname), v in per.items():
            vals[names[did]][cname].append(v)
    for label, frag, rd, wr in CASES:
        for name, cs in vals.items():
            if frag in name and "pack" not in name:
                f = cs.get("FETCH_SIZE", [float("nan")])
                w = cs.get("WRITE_SIZE", [float("nan")])
                fb, wb = sum(f) / len(f) * 1024, sum(w) / len(w) * 1024
                print(f"{label}\n    FETCH_SIZE {fb / 1e6:9.1f} MB vs {rd / 1e6:9.1f} MB algorithmic reads  -> ratio {fb / rd:.3f}"
                      f"\n    WRITE_SIZE {wb / 1e6:9.1f} MB vs {wr / 1e6:9.1f} MB algorithmic writes -> ratio {wb / wr:.3f}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2:])
