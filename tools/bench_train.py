#!/usr/bin/env python3
"""Training-step timing (BASELINE config #5): forward + Gumbel straight-through backward of Compressor(128, 2,
[8192, 2048, 512]) on 256x256 crops, 8 images per GPU, gradients all-reduced by torch DDP over RCCL when launched
with torch.distributed.run.  Not the headline metric (bench.py is); prints one JSON line on rank 0.

    python tools/bench_train.py [--steps K --warmup W --batch 8]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_train.py
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks (one per GPU); > 1 re-executes under torch.distributed.run")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--crop", type=int, default=256)
    ap.add_argument("--graph", action="store_true", help="capture forward+backward in one hipGraph and replay it (single GPU)")
    ap.add_argument("--graph-streams", action="store_true", help="with --graph: keep the branch streams on during capture (experiment)")
    ap.add_argument("--split", action="store_true", help="probe (tools/probes/split_capture.py): the step as several single-stream hipGraphs, the weight "
                    "gradients' batches on a side stream beside the input gradients' chain")
    ap.add_argument("--graphed", action="store_true",
                    help="parallel.GraphedTrainStep: main hipGraph (forward + backward, flat gradient buffer) + one gradient all-reduce "
                         "over the ranks + post hipGraph (SGD update, frequency EMA) -- the data-parallel step without DDP's hooks")
    ap.add_argument("--optimizer-step", action="store_true",
                    help="SGD step inside the timed region: the weights change every step, so every conv re-packs its forward and "
                         "input-gradient operand streams each step, as in a real training loop")
    ap.add_argument("--tile-rule", default="", help="tile experiments: comma-separated nprob:H:Cout:tile[:flagmask] entries, e.g. 1:128:128:0x121 -- "
                                                    "3x3 stride-1 launches with that many problems on H x H maps get that forced tile")
    args = ap.parse_args()
    if args.tile_rule:
        from mcquic_amd import ops
        rules = {}
        for ent in args.tile_rule.split(","):
            f = ent.split(":")
            rules[(int(f[0]), int(f[1]), int(f[2]))] = int(f[3], 16)
        orig_conv, orig_multi = ops.conv2d, ops.conv2d_multi

        def conv(x, w, stride=1, **kw):
            t = rules.get((1, x.shape[2], w.cout)) if (w.ksize == 3 and stride == 1 and "tile" not in kw) else None
            return orig_conv(x, w, stride, **(dict(kw, tile=t) if t else kw))

        def multi(xs, ws, stride=1, per_problem=None, **shared):
            t = rules.get((len(xs), xs[0].shape[2], ws[0].cout)) if (ws[0].ksize == 3 and stride == 1 and "tile" not in shared) else None
            if t:
                shared = dict(shared, tile=t)
            ops.conv2d = orig_conv
            try:
                return orig_multi(xs, ws, stride, per_problem=per_problem, **shared)
            finally:
                ops.conv2d = conv
        ops.conv2d, ops.conv2d_multi = conv, multi
    if (args.graph or args.split) and not args.graph_streams:
        # capturing the nested stream forks of the training graph crashes hipGraph capture (ROCm 7.2): one stream there
        os.environ["MCQUIC_AMD_BRANCH_STREAMS"] = "0"
    from mcquic_amd import launch
    rank, local, world, launched = launch.ensure_world(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    launch.pin_rank_cores(local, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    use_dist = world > 1 or launched                     # under torch.distributed.run the DDP / RCCL path runs even at N = 1
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        if launch.rccl_check(dist, dev) != world:
            raise SystemExit("bench_train.py: RCCL does not span the requested ranks")
    from mcquic_amd import Compressor
    from mcquic_amd.autograd import backward, mse_loss
    torch.manual_seed(3407)
    model = Compressor(128, 2, [8192, 2048, 512]).to(dev).train()
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local]) if use_dist and not args.graphed else model
    x = (torch.rand((args.batch, 3, args.crop, args.crop), generator=torch.Generator().manual_seed(rank)) * 2 - 1).to(dev)

    opt = torch.optim.SGD(model.parameters(), lr=1e-6) if args.optimizer_step else None

    def step():
        for p in model.parameters():
            p.grad = None
        xHat, yHat, codes, logits = net(x)
        loss = mse_loss(xHat, x)                            # plain MSE through this library's reduction (no memset node in a capture)
        backward(loss, defer_reduce=net is model)          # (under torch DDP the bucket hooks read gradients while the pass runs)
        if opt is not None:
            opt.step()
        return loss

    if args.graphed:
        from mcquic_amd import parallel
        args.optimizer_step = True
        gstep = parallel.GraphedTrainStep(model, torch.optim.SGD(model.parameters(), lr=1e-6), x)

        def step():                                        # noqa: F811
            return gstep(x)
    for _ in range(args.warmup):
        step()
    if args.split and not use_dist and not args.graphed:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes"))
        os.environ.setdefault("MCQUIC_AMD_WGRAD_SIDE", "1")
        from split_capture import SplitCapture            # (probe: measured slower than one graph, see its header)
        cap = SplitCapture(dev)
        with torch.cuda.stream(cap.main):
            step()
            step()
        torch.cuda.synchronize()
        for p in model.parameters():
            p.grad = None
        with cap:
            xHat, yHat, codes, logits = net(x)
            static_loss = mse_loss(xHat, x)
            backward(static_loss)
            if opt is not None:
                opt.step()
        print("split capture:", [w for w, _ in cap.program], file=sys.stderr)

        def step():                                        # noqa: F811
            cap.replay()
            return static_loss
        step()
    elif args.graph and not use_dist and not args.graphed:
        # whole-step capture: ~5000 kernel launches per step become one graph launch
        torch.cuda.synchronize()
        for p in model.parameters():
            p.grad = None
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            xHat, yHat, codes, logits = net(x)
            static_loss = mse_loss(xHat, x)
            backward(static_loss)
            if opt is not None:
                opt.step()                                 # (the update itself is part of the captured step; the forward above
                                                           #  starts with the grouped re-pack of every stale operand stream)

        def step():                                        # noqa: F811
            graph.replay()
            return static_loss
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)))
        print(json.dumps({"metric": "training step (forward + backward), 256x256 crops, qp=2 model", "n_gpus": world,
                          "images_per_gpu": args.batch, "ms_per_step": round(dt / args.steps * 1e3, 2),
                          "images_per_s": round(world * args.batch * args.steps / dt, 2), "loss": float(loss), "grad_norm": gn,
                          "dtype": "f32", "graph": bool(((args.graph or args.split) and not use_dist) or args.graphed), "split": bool(args.split), "ddp": bool(use_dist and not args.graphed),
                          "graphed_data_parallel": bool(args.graphed),
                          "optimizer_step": "SGD inside the timed region (weights re-packed every step)" if args.optimizer_step else "not included"}))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
