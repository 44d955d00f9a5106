"""CPU, world_size 2 over gloo: the N>1 path = contiguous image shards + statistics collectives only."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mcquic_amd import parallel
    ok = True
    for n in (5, 1):                                          # ragged: 3 + 2; then 1 + 0 -- rank 1's shard is EMPTY (fewer images than ranks)
        lo, hi = parallel.shard_range(n, rank, world)
        g = torch.Generator().manual_seed(0)
        stats_all = torch.rand((n, 3), generator=g)
        codes_all = [torch.randint(0, k, (n, 2, s, s), generator=g) for k, s in ((32, 4), (16, 2), (8, 1))]
        stats = parallel.gather_image_stats(stats_all[lo:hi])
        hist = parallel.code_histograms([c[lo:hi] for c in codes_all], [32, 16, 8])
        ok = ok and torch.equal(stats, stats_all)
        for h, c, k in zip(hist, codes_all, (32, 16, 8)):
            want = torch.stack([torch.bincount(c[:, m].reshape(-1), minlength=k) for m in range(2)])
            ok = ok and torch.equal(h, want)
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_shard_range_covers_batch():
    from mcquic_amd import parallel
    for n in (0, 1, 7, 32, 256):
        for world in (1, 2, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_stats_collectives_world2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world))


def _ema_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mcquic_amd.modules.entropyCoder import EntropyCoder
    from oracle import mcquic_ref as R
    m, ks = 2, [32, 16, 8]
    coder = EntropyCoder(m, ks)
    g = torch.Generator().manual_seed(1)
    codes_all = [torch.randint(0, k, (6, m, s, s), generator=g) for k, s in zip(ks, (4, 2, 1))]
    before = [f.detach().clone() for f in coder._freqEMA]
    lo, hi = (0, 3) if rank == 0 else (3, 6)                      # each rank sees its shard of the batch
    coder([c[lo:hi] for c in codes_all])
    ok = True
    for lv, (c, k) in enumerate(zip(codes_all, ks)):
        counts = torch.stack([torch.bincount(c[:, g_].reshape(-1), minlength=k) for g_ in range(m)]).float()
        want = R.freq_ema_update(before[lv], counts)              # the reference's update on the WHOLE batch
        ok = ok and torch.allclose(coder._freqEMA[lv], want, atol=1e-7)
    # codebook sync: rank 1 ends up with rank 0's codebook
    from mcquic_amd import Compressor
    torch.manual_seed(rank)
    model = Compressor(8, 2, [32, 16, 8])
    model.syncCodebook()
    ref = [cb.detach().clone() for cb in model.Codebooks]
    for cb in ref:
        dist.broadcast(cb, 0)
    ok = ok and all(torch.equal(a, b) for a, b in zip(model.Codebooks, ref))
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_freq_ema_allreduce_and_codebook_sync_world2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_ema_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world))


def _ddp_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mcquic_amd import parallel
    torch.manual_seed(rank)                                    # different initial weights: rank 0's must win
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    net = parallel.data_parallel(model, torch.device("cpu"), stage_through_host=True)     # the two-ranks-on-one-GPU code path
    g = torch.Generator().manual_seed(5)
    x, y = torch.rand((8, 6), generator=g), torch.rand((8, 3), generator=g)
    lo, hi = parallel.shard_range(8, rank, world)
    torch.nn.functional.mse_loss(net(x[lo:hi]), y[lo:hi]).backward()
    got = [p.grad.clone() for p in model.parameters()]
    torch.manual_seed(0)
    solo = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    torch.nn.functional.mse_loss(solo(x), y).backward()
    ret[rank] = all(torch.allclose(a, p.grad, rtol=1e-5, atol=1e-7) for a, p in zip(got, solo.parameters())) and \
        all(torch.equal(p, q) for p, q in zip(model.parameters(), solo.parameters()))
    dist.destroy_process_group()


def test_data_parallel_host_staged_buckets_world2_gloo():
    """parallel.data_parallel with the host-staged state broadcast + gradient buckets (what two ranks sharing one GPU under gloo
    use): both ranks end with rank 0's weights and the gradient of the global mean loss."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world))


def _deferred_worker(rank, world, port, ret):
    """The exchange part of parallel.GraphedTrainStep on CPU tensors: every rank's EntropyCoder leaves its LOCAL counts in a
    buffer (deferred mode), one all-reduce makes them global, applyCounts gives the frequency EMA the in-forward all-reduce
    gives; a flat float buffer averaged the same way."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mcquic_amd import parallel
    from mcquic_amd.modules.entropyCoder import EntropyCoder
    ks = [32, 16, 8]
    g = torch.Generator().manual_seed(1)
    codes_all = [torch.randint(0, k, (6, 2, s, s), generator=g) for k, s in zip(ks, (4, 2, 1))]
    lo, hi = parallel.shard_range(6, rank, world)
    mine = [c[lo:hi] for c in codes_all]
    direct, deferred = EntropyCoder(2, ks), EntropyCoder(2, ks)
    direct(mine)                                             # all-reduce inside forward (the reference's order)
    deferred.deferCounts(True)
    deferred(mine)
    assert all(torch.equal(a, torch.ones_like(a) / a.shape[-1]) for a in deferred._freqEMA)     # nothing applied yet
    counts = deferred.countSink().clone()
    assert torch.equal(counts, parallel.local_code_counts(mine, ks))
    parallel.all_reduce_(counts)
    deferred.applyCounts(counts)
    deferred.deferCounts(False)
    ok = all(torch.equal(a, b) for a, b in zip(direct._freqEMA, deferred._freqEMA))
    ok = ok and int(counts.sum()) == 6 * 2 * (16 + 4 + 1)
    flat = torch.full((5,), float(rank + 1))
    parallel.all_reduce_(flat)
    ok = ok and torch.equal(flat, torch.full((5,), 3.0))
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_deferred_counts_and_flat_allreduce_world2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_deferred_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world))


def test_prefetch_is_the_identity_on_a_cpu_device():
    from mcquic_amd import parallel
    batches = [torch.full((2, 3), float(i)) for i in range(4)]
    out = list(parallel.prefetch(iter(batches), torch.device("cpu")))
    assert len(out) == 4 and all(a is b for a, b in zip(out, batches))
    assert list(parallel.prefetch([], "cpu")) == []
