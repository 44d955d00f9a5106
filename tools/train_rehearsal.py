#!/usr/bin/env python3
"""A training loop, not a step: `parallel.GraphedTrainStep` driven the way the reference's trainer drives its model
(mcquic/train/trainer.py:263-296: forward, loss, backward, clip_grad_norm(4.0), optimizer step; hooks every few hundred steps:
validation, codebook re-assignment, mcquic/train/hooks.py:100-121) over synthetic images a model can learn (smooth random fields
+ texture, a fresh batch every step), with everything a loop does BETWEEN replays -- `loss.item()`, a validation pass through the
eager encode / decode, `reAssignCodebook`, finiteness checks -- the work that exposed the memset-node defect (docs/experiments.md
section 9.9).  Prints one JSON line: loss trace, validation PSNR trace, gradient norms, step time.

    python tools/train_rehearsal.py [--steps 600] [--channel 128] [--batch 8] [--crop 256] [--out profiles/r04_train_rehearsal.json]
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


def images(n, size, gen, dev):
    """Smooth colour fields (bicubic from 8x8 and 32x32 noise) plus a little pixel noise, in [-1, 1]."""
    lo = torch.nn.functional.interpolate(torch.randn((n, 3, 8, 8), generator=gen, device=dev), size=size, mode="bicubic", align_corners=False)
    mid = torch.nn.functional.interpolate(torch.randn((n, 3, 32, 32), generator=gen, device=dev), size=size, mode="bicubic", align_corners=False)
    x = 0.45 * lo + 0.2 * mid + 0.02 * torch.randn((n, 3, size, size), generator=gen, device=dev)
    return x.clamp_(-1.0, 1.0).contiguous()


def psnr(model, val):
    from mcquic_amd import ops
    with torch.no_grad():
        rec = model.decode(model.encode(val))
    h, w = val.shape[-2:]                                          # (decode returns the padded size: crop like `decompress` does)
    top, left = (rec.shape[-2] - h) // 2, (rec.shape[-1] - w) // 2
    rec = rec[..., top: top + h, left: left + w].contiguous()
    a, b = ops.detransform(val), ops.detransform(rec)
    mse = (a.double() - b.double()).pow(2).mean(dim=(1, 2, 3))
    return float((10 * torch.log10(255.0 ** 2 / (mse + 1e-4))).mean())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--channel", type=int, default=128)
    ap.add_argument("--m", type=int, default=2, help="codebooks per level (12 with --channel 192: the reference's model No. 12)")
    ap.add_argument("--ks", default="8192,2048,512")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--crop", type=int, default=256)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--every", type=int, default=100, help="validation + finiteness check + codebook re-assignment period")
    ap.add_argument("--warmup", type=int, default=0, help="linear learning-rate warm-up over this many steps (the reference: 2000, configs/a800_8.yaml); "
                                                          "the rate is a device tensor the captured update reads, refilled by the host every step")
    ap.add_argument("--fused", action="store_true", help="torch.optim.Adam(fused=True): torch's multi-tensor kernel (19 launches for this model)")
    ap.add_argument("--own-adam", action="store_true", help="mcquic_amd.optim.Adam: the whole model in one launch")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from mcquic_amd import Compressor, parallel
    dev = torch.device("cuda:0")
    torch.manual_seed(3407)
    ks = [int(k) for k in args.ks.split(",")]
    model = Compressor(args.channel, args.m, ks).to(dev).train()
    gen = torch.Generator(device=dev).manual_seed(1)
    val = images(4, args.crop, torch.Generator(device=dev).manual_seed(99), dev)
    lr = torch.tensor(args.lr, device=dev)
    if args.own_adam:
        from mcquic_amd import optim
        opt = optim.Adam(model.parameters(), lr=lr)
    else:
        opt = torch.optim.Adam(model.parameters(), lr=lr, capturable=True, **({"fused": True} if args.fused else {}))
    x = images(args.batch, args.crop, gen, dev)
    step = parallel.GraphedTrainStep(model, opt, x, max_grad_norm=4.0)
    trace = {"loss": [], "grad_norm": [], "psnr": [], "reassigned": [], "non_finite": 0}
    model.eval()
    step.invalidate()
    trace["psnr"].append((0, round(psnr(model, val), 3)))
    model.train()
    torch.cuda.synchronize()
    t_steps = 0.0
    for i in range(1, args.steps + 1):
        x = images(args.batch, args.crop, gen, dev)
        if args.warmup:
            lr.fill_(args.lr * min(1.0, i / args.warmup))         # (a scheduler's job; the captured update reads the tensor)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = step(x)
        torch.cuda.synchronize()
        t_steps += time.perf_counter() - t0
        lv = loss.item()                                           # (what a logger does every step)
        if i % max(10, args.steps // 200) == 0 or i == 1 or os.environ.get("REHEARSAL_TRACE_ALL"):
            trace["loss"].append((i, round(lv, 6)))
            trace["grad_norm"].append((i, round(float(step.grad_norm), 5)))
        if lv != lv:
            trace["non_finite"] += 1
        if i == args.every:
            trace["_mem_first"] = round(torch.cuda.memory_allocated(dev) / 2 ** 20, 1)
        if i % args.every == 0:
            trace["non_finite"] += sum(int(not torch.isfinite(p).all()) for p in model.parameters())
            step.invalidate()
            model.eval()
            trace["psnr"].append((i, round(psnr(model, val), 3)))
            model.train()
            if i % (2 * args.every) == 0 and i < args.steps:
                trace["reassigned"].append((i, round(float(model.reAssignCodebook()), 4)))
    mem = {"allocated_mb_after_first_period": None, "allocated_mb_at_end": round(torch.cuda.memory_allocated(dev) / 2 ** 20, 1),
           "peak_allocated_mb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 20, 1), "reserved_mb": round(torch.cuda.memory_reserved(dev) / 2 ** 20, 1)}
    mem["allocated_mb_after_first_period"] = trace.pop("_mem_first", None)
    step.close()
    first = sum(v for _, v in trace["loss"][:3]) / 3
    last = sum(v for _, v in trace["loss"][-3:]) / 3
    out = {"what": "GraphedTrainStep(Adam, lr in a device tensor, max_grad_norm=4.0) on fresh synthetic batches; loss.item() every step; every "
                   f"{args.every} steps: finiteness of all parameters, eager encode/decode PSNR on 4 held-out images, codebook re-assignment every {2 * args.every}",
           "model": f"Compressor({args.channel}, {args.m}, {ks})", "lr": args.lr, "lr_warmup_steps": args.warmup, "optimizer": "mcquic_amd.optim.Adam" if args.own_adam else ("torch Adam fused" if args.fused else "torch Adam foreach"), "post_captured": step.post is not None, "batch": args.batch, "crop": args.crop, "steps": args.steps,
           "ms_per_step": round(t_steps / args.steps * 1e3, 3), "loss_first": round(first, 6), "loss_last": round(last, 6),
           "psnr_first": trace["psnr"][0][1], "psnr_last": trace["psnr"][-1][1], "memset_nodes_ok": parallel.memset_nodes_replay_correctly(dev), "memory": mem, **trace}
    line = json.dumps(out)
    print(line)
    if args.out:
        with open(args.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
