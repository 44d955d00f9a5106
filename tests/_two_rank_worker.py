"""Worker of tests/test_gpu_two_ranks.py: one of TWO processes that share cuda:0 and talk over gloo (RCCL refuses two ranks
on one device).  Each rank drives the HIP kernels on its shard; rank 0 checks the combined result against the same work done
by one process and writes a JSON verdict.

    RANK=r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/_two_rank_worker.py validate|train out.json
"""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def validate_mode(rank, world, dev):
    """validator.py:40-58 sharded: every rank runs `validate` (compress -> rANS -> decompress -> PSNR / MS-SSIM / bpp) on its
    parallel.shard_range slice, rows gathered, code histograms all-reduced (IdealBPP)."""
    from mcquic_amd import Compressor, parallel, validate
    from oracle import mcquic_ref as R
    ks = [8192, 2048, 512]
    sd = R.make_state_dict(128, 2, ks, seed=0)
    model = Compressor(128, 2, ks).eval()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    n = 7                                                               # ragged: 4 + 3
    x = R.make_images(n, 768, 512, seed=3407).to(dev)
    lo, hi = parallel.shard_range(n, rank, world)
    rows = validate.validate(model, x[lo:hi])                           # [n, 3] on every rank (all_gather over gloo)
    codes = model.encode(x[lo:hi])
    hist = parallel.code_histograms(codes, ks)                          # summed over both ranks
    out = {"rows_shape": list(rows.shape)}
    if rank == 0:
        # the same shards, one after the other, in THIS process (no collectives): must be bit-equal
        spans = [parallel.shard_range(n, r, world) for r in range(world)]
        alone = torch.cat([validate.validate(model, x[a:b], gather=False) for a, b in spans], 0)
        out["rows_bit_equal"] = bool(torch.equal(rows, alone))
        out["rows_max_abs_diff"] = float((rows - alone).abs().nan_to_num().max())
        want_hist = [torch.zeros(2, k, dtype=torch.int64, device=dev) for k in ks]
        for a, b in spans:
            for lv, c in enumerate(model.encode(x[a:b])):
                for g in range(2):
                    want_hist[lv][g] += torch.bincount(c[:, g].reshape(-1), minlength=ks[lv])
        out["hist_equal"] = all(bool(torch.equal(h, w)) for h, w in zip(hist, want_hist))
        out["hist_total"] = int(sum(int(h.sum()) for h in hist))
        out["hist_expected_total"] = n * 2 * (48 * 32 + 24 * 16 + 12 * 8)
        out["ideal_bpp"] = validate.ideal_bpp(hist, n * 768 * 512)
        out["psnr_min_db"] = float(rows[:, 0].min())
    return out


def train_mode(rank, world, dev):
    """config #5 in miniature: DDP over the two ranks, 2 crops of 256x256 each, against ONE process stepping on all 4 crops
    (same weights, same uniform draws): DDP's gradient averaging over equal shards = the gradient of the global mean loss."""
    from mcquic_amd import Compressor, parallel
    ks = [8192, 2048, 512]
    per, hw = 2, 256
    n = per * world
    torch.manual_seed(3407)
    model = Compressor(128, 2, ks).to(dev).train()
    g = torch.Generator().manual_seed(11)
    x = (torch.rand((n, 3, hw, hw), generator=g) * 2 - 1).to(dev)
    us = []
    for lv, k in enumerate(ks):
        s = hw // 16 // (2 ** lv)
        us.append((torch.rand((n, 2, s, s, k), generator=g).to(dev), torch.rand((n, 2, s, s, k), generator=g).to(dev)))
    ema0 = [f.detach().clone() for f in model._quantizer._entropyCoder._freqEMA]
    solo = dist.new_group([0])                                           # a world of one for rank 0's reference pass
    out, want, codes1, ema_solo = {}, {}, None, None
    if rank == 0:
        # ONE process, all four crops, before DDP hooks exist.  No cross-rank collective may run in here (rank 1 waits at the
        # barrier below): the frequency-EMA update's all-reduce is pointed at the single-rank group
        import mcquic_amd.parallel as P
        orig = P.all_reduce_
        P.all_reduce_ = lambda buf, group=None: orig(buf, solo)
        try:
            xHat1, _, codes1, _ = model(x, uniforms=us)
        finally:
            P.all_reduce_ = orig
        torch.nn.functional.mse_loss(xHat1, x).backward()
        torch.cuda.synchronize()
        want = {name: p.grad.detach().clone() for name, p in model.named_parameters() if p.grad is not None}
        ema_solo = [f.detach().clone() for f in model._quantizer._entropyCoder._freqEMA]
        for p in model.parameters():
            p.grad = None
        with torch.no_grad():
            for f, f0 in zip(model._quantizer._entropyCoder._freqEMA, ema0):
                f.copy_(f0)
    dist.barrier()
    lo, hi = rank * per, (rank + 1) * per
    net = parallel.data_parallel(model, dev)
    xHat, yHat, codes, logits = net(x[lo:hi], uniforms=[(a[lo:hi], b[lo:hi]) for a, b in us])
    loss = torch.nn.functional.mse_loss(xHat, x[lo:hi])
    loss.backward()
    torch.cuda.synchronize()
    codes_all = [torch.cat(parallel_gather(c, world, dev), 0) for c in codes]
    if rank == 0:
        worst, worst_name, n_grads = 0.0, "", 0
        for name, p in model.named_parameters():
            if name not in want:
                continue
            n_grads += 1
            got, ref = p.grad, want[name]
            rel = float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-12)
            if rel > worst:
                worst, worst_name = rel, name
        out["n_grads"] = n_grads
        out["worst_grad_rel_err"] = worst
        out["worst_grad_name"] = worst_name
        out["codes_equal"] = all(bool(torch.equal(a, b)) for a, b in zip(codes_all, codes1))
        out["ema_max_abs_diff"] = max(float((a - b).abs().max()) for a, b in zip(ema_solo, model._quantizer._entropyCoder._freqEMA))
        out["loss_ddp_rank0"] = float(loss)
    return out


def graphed_mode(rank, world, dev, clip=None):
    """parallel.GraphedTrainStep over the two ranks (2 crops each; main graph, flat gradient all-reduce + count all-reduce staged
    through the host under gloo, post graph) against ONE process making the same SGD update on all 4 crops.  `clip`: both sides
    clip by that global norm first (the graphed step: the norm of the AVERAGED gradient, inside its captured update)."""
    import copy
    from mcquic_amd import Compressor, parallel
    ks = [8192, 2048, 512]
    per, hw, lr = 2, 256, 1e-2
    n = per * world
    torch.manual_seed(3407)
    model = Compressor(128, 2, ks).to(dev).train()
    g = torch.Generator().manual_seed(11)
    x = (torch.rand((n, 3, hw, hw), generator=g) * 2 - 1).to(dev)
    us = []
    for lv, k in enumerate(ks):
        s = hw // 16 // (2 ** lv)
        us.append((torch.rand((n, 2, s, s, k), generator=g).to(dev), torch.rand((n, 2, s, s, k), generator=g).to(dev)))
    out, ref, ref_ema = {}, None, None
    if rank == 0:
        solo_model = copy.deepcopy(model)
        solo = dist.new_group([0])
        import mcquic_amd.parallel as P
        orig = P.all_reduce_
        P.all_reduce_ = lambda buf, group=None: orig(buf, solo)
        try:
            opt = torch.optim.SGD(solo_model.parameters(), lr=lr)
            xHat = solo_model(x, uniforms=us)[0]
            torch.nn.functional.mse_loss(xHat, x).backward()
            if clip is not None:
                out["solo_grad_norm"] = float(torch.nn.utils.clip_grad_norm_(solo_model.parameters(), clip))
            opt.step()
        finally:
            P.all_reduce_ = orig
        torch.cuda.synchronize()
        ref = [p.detach().clone() for p in solo_model.parameters()]
        ref_ema = [f.detach().clone() for f in solo_model._quantizer._entropyCoder._freqEMA]
    else:
        dist.new_group([0])                                             # (new_group is collective over the default group)
    dist.barrier()
    lo, hi = rank * per, (rank + 1) * per
    init = [p.detach().clone() for p in model.parameters()]
    if rank != 0:                                                       # a replica that starts elsewhere: the constructor's broadcast
        with torch.no_grad():                                           # from rank 0 must bring it back, or the update below is wrong
            for p in model.parameters():
                p.add_(0.01)
    step = parallel.GraphedTrainStep(model, torch.optim.SGD(model.parameters(), lr=lr), x[lo:hi],
                                     forward_kwargs={"uniforms": [(a[lo:hi], b[lo:hi]) for a, b in us]}, max_grad_norm=clip)
    loss = step(x[lo:hi])
    if clip is not None and rank == 0:
        out["grad_norm"] = float(step.grad_norm)
    step.close()
    torch.cuda.synchronize()
    if rank == 0:
        worst, worst_name, moved = 0.0, "", 0.0
        for (name, p), r, p0 in zip(model.named_parameters(), ref, init):
            if not p.requires_grad:
                continue
            upd = float((r - p0).abs().max())
            if upd == 0.0:
                continue
            # error against 1e-4 of the update itself plus the rounding of the parameter's own magnitude (an update of a few
            # ulps cannot agree better than to an ulp)
            tol = 1e-4 * upd + 4 * 1.1920929e-07 * float(r.abs().max())
            rel = float((p.detach() - r).abs().max()) / tol
            moved = max(moved, upd)
            if rel > worst:
                worst, worst_name = rel, name
        out["worst_update_rel_err"] = worst
        out["worst_update_name"] = worst_name
        out["largest_update"] = moved
        out["post_captured"] = step.post is not None
        out["segments"] = step.segments
        out["slice_mb"] = [round(sl.numel() * 4 / 1e6, 1) for sl in step.slices]
        out["ema_max_abs_diff"] = max(float((a - b).abs().max()) for a, b in zip(ref_ema, model._quantizer._entropyCoder._freqEMA))
        out["loss_rank0"] = float(loss)
    return out


def parallel_gather(t, world, dev):
    host = t.cpu()
    out = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(out, host)
    return [o.to(dev) for o in out]


def main():
    mode, path = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)                                       # BOTH ranks on the one GPU
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {"validate": validate_mode, "train": train_mode, "graphed": graphed_mode,
           "graphed_clip": lambda r, w, d: graphed_mode(r, w, d, clip=5e-3)}[mode](rank, world, dev)
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0:
        from mcquic_amd import _lib
        out["lib"] = _lib.LIB_PATH
        json.dump(out, open(path, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
