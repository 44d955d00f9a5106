"""Worker of tests/test_gpu_graph_replay.py: the captured training step replayed with EAGER work between the replays (what a
training loop does: logging, `loss.item()`, checks over the parameters), in a process whose hipGraph launch path is chosen by the
environment (DEBUG_CLR_GRAPH_PACKET_CAPTURE, read when the HIP runtime starts).  Prints one JSON line.

    DEBUG_CLR_GRAPH_PACKET_CAPTURE=1|0 python tests/_graph_replay_worker.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mcquic_amd import Compressor, parallel  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(3407)
    ks = [8192, 2048, 512]
    model = Compressor(128, 2, ks).to(dev).train()
    n, hw = 8, 256
    x = (torch.rand((n, 3, hw, hw), generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
    g = torch.Generator().manual_seed(5)
    us = []
    for lv, k in enumerate(ks):
        s = hw // 16 // (2 ** lv)
        us.append((torch.rand((n, 2, s, s, k), generator=g).to(dev), torch.rand((n, 2, s, s, k), generator=g).to(dev)))
    out = {"packet_capture_env": os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"), "memset_nodes_ok": parallel.memset_nodes_replay_correctly(dev)}

    # (1) the main graph alone over fixed inputs: every replay must reproduce the first one (up to the order of the two atomic
    #     addends of the soft assignment's input gradient)
    step = parallel.GraphedTrainStep(model, torch.optim.SGD(model.parameters(), lr=0.0), x, forward_kwargs={"uniforms": us}, capture_post=False)
    first, worst = None, 0.0
    for i in range(5):
        step.flat.zero_()
        step.graphs[0].replay()
        torch.cuda.synchronize()
        if first is None:
            first = step.flat.clone()
            out["first_replay_finite"] = bool(torch.isfinite(first).all())
            continue
        off = 0
        for p in step.live:
            a, b = step.flat[off: off + p.numel()], first[off: off + p.numel()]
            err = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
            worst = max(worst, err) if err == err else float("inf")
            off += p.numel()
    out["replay_vs_first_worst_relative"] = worst
    step.close()

    # (2) whole steps (update captured too) with eager work between them
    step = parallel.GraphedTrainStep(model, torch.optim.SGD(model.parameters(), lr=1e-4), x, forward_kwargs={"uniforms": us})
    losses, bad = [], []
    for i in range(6):
        loss = step(x * (1.0 - 0.1 * i))                                   # (six visibly different losses: a stale one shows)
        bad.append(sum(int(not torch.isfinite(p).all()) for p in model.parameters())
                   + sum(int(not torch.isfinite(p.grad).all()) for p in step.live))
        losses.append(float(loss))
    out["losses"] = losses
    out["non_finite_tensors_per_step"] = bad
    step.close()

    # (3) the inference graphs (enableGraphs: what batch-1 serving replays) with the same eager work between replays
    model.eval()
    xs = [(torch.rand((1, 3, 256, 384), generator=torch.Generator().manual_seed(40 + i)) * 2 - 1).to(dev) for i in range(4)]
    want = []
    for xi in xs:
        codes = model.encode(xi)
        want.append(([c.clone() for c in codes], model.decode(codes).clone()))
    model.enableGraphs(True)
    same = []
    for rnd in range(2):
        for xi, (wc, wr) in zip(xs, want):
            codes = model.encode(xi)
            rec = model.decode(codes)
            sum(int(not torch.isfinite(p).all()) for p in model.parameters())
            same.append(all(torch.equal(a, b) for a, b in zip(codes, wc)) and torch.equal(rec, wr))
    out["inference_graph_replays_equal_eager"] = same
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
