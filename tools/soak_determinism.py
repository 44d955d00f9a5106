#!/usr/bin/env python3
"""Soak: the same inputs through encode / decode (eager at batch 4 and 32, hipGraph replays at batch 1) and through the captured
training graph, thousands of times; every result must be bit-identical to the first (integer codes AND float reconstructions and
gradients: no launch of the path has an order-dependent reduction except the two-addend atomic of the soft assignment's input
gradient, whose sum does not depend on the order).  A race in a split-K reduction, an LDS hand-over or a graph replay shows up as a
mismatch count.  Prints one JSON line.

    python tools/soak_determinism.py [--scale 1.0] [--out profiles/r04_soak_determinism.json]
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0, help="multiplies every iteration count")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from mcquic_amd import Compressor, parallel
    from oracle import mcquic_ref as R
    dev = torch.device("cuda:0")
    ks = [8192, 2048, 512]
    model = Compressor(128, 2, ks).eval()
    model.load_state_dict(R.make_state_dict(128, 2, ks, seed=0), strict=True)
    model = model.to(dev)
    out = {}
    t_all = time.perf_counter()

    def soak(tag, x, iters, graphs=False):
        model.enableGraphs(graphs)
        with torch.no_grad():
            codes0 = [c.clone() for c in model.encode(x)]
            rec0 = model.decode(codes0).clone()
            bad = torch.zeros(2, dtype=torch.int64, device=dev)
            t0 = time.perf_counter()
            for _ in range(iters):
                codes = model.encode(x)
                rec = model.decode(codes)
                bad[0] += sum((a != b).any().to(torch.int64) for a, b in zip(codes, codes0))
                bad[1] += (rec != rec0).any().to(torch.int64)
            torch.cuda.synchronize()
        out[tag] = {"iterations": iters, "code_mismatching_runs": int(bad[0]), "pixel_mismatching_runs": int(bad[1]),
                    "seconds": round(time.perf_counter() - t0, 1)}
        model.enableGraphs(False)

    soak("encode_decode_b4_eager", R.make_images(4, 768, 512, seed=1).to(dev), int(1500 * args.scale))
    soak("encode_decode_b32_eager", R.make_images(32, 768, 512, seed=2).to(dev), int(150 * args.scale))
    soak("encode_decode_b1_graphs", R.make_images(1, 768, 512, seed=3).to(dev), int(3000 * args.scale), graphs=True)
    soak("encode_decode_ragged_b3_eager", R.make_images(3, 200, 136, seed=4).to(dev), int(1500 * args.scale))

    # the captured training graph over fixed draws, no update: gradients bit-identical on every replay
    torch.manual_seed(3407)
    tm = Compressor(128, 2, ks).to(dev).train()
    n, hw = 8, 256
    x = (torch.rand((n, 3, hw, hw), generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
    g = torch.Generator().manual_seed(5)
    us = []
    for lv, k in enumerate(ks):
        s = hw // 16 // (2 ** lv)
        us.append((torch.rand((n, 2, s, s, k), generator=g).to(dev), torch.rand((n, 2, s, s, k), generator=g).to(dev)))
    step = parallel.GraphedTrainStep(tm, torch.optim.SGD(tm.parameters(), lr=0.0), x, forward_kwargs={"uniforms": us}, capture_post=False)
    step.graphs[0].replay()
    first = step.flat.clone()
    loss0 = step.loss.clone()
    iters = int(1000 * args.scale)
    bad = torch.zeros(2, dtype=torch.int64, device=dev)
    t0 = time.perf_counter()
    for _ in range(iters):
        step.graphs[0].replay()
        bad[0] += (step.flat != first).any().to(torch.int64)
        bad[1] += (step.loss != loss0).to(torch.int64)
    torch.cuda.synchronize()
    out["train_graph_b8_256"] = {"iterations": iters, "gradient_mismatching_replays": int(bad[0]), "loss_mismatching_replays": int(bad[1]),
                                 "finite": bool(torch.isfinite(first).all()), "seconds": round(time.perf_counter() - t0, 1)}
    step.close()
    out["total_seconds"] = round(time.perf_counter() - t_all, 1)
    out["all_identical"] = all(v == 0 for d in out.values() if isinstance(d, dict) for k, v in d.items() if "mismatching" in k)
    line = json.dumps(out)
    print(line)
    if args.out:
        with open(args.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
