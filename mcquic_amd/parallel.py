"""Multi-GPU layer of the Compressor path: one process per GPU, images sharded, no data-path collective.

Every image is independent in encode / decode, so a batch is cut into contiguous per-rank
slices and each rank runs the whole model on its slice.  Collectives (RCCL over xGMI with backend "nccl", gloo
in the CPU tests) appear only where validation statistics are combined:
  * all_gather of per-image statistics rows (e.g. [psnr, ms_ssim, bits]) -- what the reference's validator
    accumulates on rank 0 only (mcquic/validate/validator.py:40-58);
  * all_reduce(sum) of per-level code histograms [m, k_l] -- what `IdealBPP` counts
    (mcquic/validate/handlers.py:110-187) and what EntropyCoder.forward all-reduces in training
    (mcquic/modules/entropyCoder.py:28-44).  The three levels travel in ONE flat buffer (86 KB for qp=2): the
    message is latency-bound, so one collective instead of three.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) slice of `n` images for `rank`; the first n % world ranks get one extra."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def prefetch(batches, device: torch.device):
    """Host-resident batches -> device tensors, the copy of batch i + 1 in flight while the caller computes on batch i (what the
    reference's validator gets from a pinned DataLoader + `.to(device, non_blocking=True)`, mcquic/validate/validator.py:40-58).
    The copies run on a side stream (SDMA engines, no CU is taken from the kernels); the caller's stream waits on each batch's
    event and the tensor is tied to it with `record_stream`, so the allocator cannot hand its memory to the next copy early.
    Give PINNED host tensors -- a pageable source makes the copy synchronous.  On a CPU device: the batches as they are."""
    device = torch.device(device)
    if device.type != "cuda":
        yield from batches
        return
    side = torch.cuda.Stream(device)

    def load(host):
        with torch.cuda.stream(side):
            t = host.to(device, non_blocking=True)
            done = torch.cuda.Event()
            done.record(side)
        return t, done
    it = iter(batches)
    try:
        ahead = load(next(it))
    except StopIteration:
        return
    while ahead is not None:
        t, done = ahead
        try:
            ahead = load(next(it))                            # issued BEFORE the caller launches anything on batch i
        except StopIteration:
            ahead = None
        here = torch.cuda.current_stream(device)
        here.wait_event(done)
        t.record_stream(here)
        yield t


def _staged(t: torch.Tensor, group=None) -> torch.Tensor:
    """The tensor a collective of `group` should run on: device tensors as they are under RCCL ("nccl"); under gloo (the CPU
    tests, and the two-processes-on-one-GPU test: RCCL refuses two ranks on one device) a host copy."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        return t.cpu()
    return t


def gather_image_stats(local: torch.Tensor, group=None) -> torch.Tensor:
    """all_gather of per-image rows [n_local, f] -> [n_total, f] in rank order (ragged shards allowed)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    home = local.device
    local = _staged(local, group)
    counts = torch.zeros(world, dtype=torch.int64, device=local.device)
    counts[dist.get_rank(group)] = local.shape[0]
    dist.all_reduce(counts, group=group)
    cap = int(counts.max())
    padded = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    return torch.cat([o[: int(c)] for o, c in zip(out, counts)], 0).to(home)


def local_code_counts(codes: Sequence[torch.Tensor], ks: Sequence[int]) -> torch.Tensor:
    """This rank's per-level code counts, all levels in ONE flat int64 buffer (level l: m_l * k_l entries, [m, k] row-major)."""
    flat = []
    for code, k in zip(codes, ks):
        n, m = code.shape[0], code.shape[1]
        idx = code.permute(1, 0, 2, 3).reshape(m, -1) + (torch.arange(m, device=code.device) * k)[:, None]
        # scatter_add instead of torch.bincount: bincount reads its maximum back to the host (a sync, and illegal
        # inside a captured hipGraph)
        idx = idx.reshape(-1)
        flat.append(torch.zeros(m * k, dtype=torch.int64, device=code.device).scatter_add_(0, idx, torch.ones_like(idx)))
    return torch.cat(flat)


def split_code_counts(buf: torch.Tensor, ms: Sequence[int], ks: Sequence[int]) -> List[torch.Tensor]:
    """The [m_l, k_l] views of a flat count buffer (local_code_counts' layout)."""
    out, off = [], 0
    for m, k in zip(ms, ks):
        out.append(buf[off: off + m * k].reshape(m, k))
        off += m * k
    return out


def all_reduce_(buf: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum of `buf` over the ranks of `group` (RCCL on device tensors; staged through the host under gloo)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        staged = _staged(buf, group)
        dist.all_reduce(staged, group=group)
        if staged is not buf:
            buf.copy_(staged)
    return buf


def code_histograms(codes: Sequence[torch.Tensor], ks: Sequence[int], group=None) -> List[torch.Tensor]:
    """Per-level code counts [m, k_l] (int64) summed over all images of all ranks with ONE all_reduce."""
    buf = all_reduce_(local_code_counts(codes, ks), group)
    return split_code_counts(buf, [code.shape[1] for code in codes], ks)


def data_parallel(model: torch.nn.Module, device: torch.device, group=None, **kwargs):
    """torch DistributedDataParallel around the training-mode Compressor: gradients produced by the HIP kernels
    (mcquic_amd.autograd) are all-reduced bucket by bucket -- over RCCL / xGMI with backend "nccl" (the reference's
    `torchrun` + DDP set-up, mcquic/train/ddp.py:79-95).  Under gloo with device parameters (tests: two ranks sharing one GPU)
    nothing relies on gloo's device support: the initial state broadcast and the gradient buckets are staged through the host
    (a communication hook), buffers are constants of the layers and are not re-broadcast."""
    from torch.nn.parallel import DistributedDataParallel
    staged = kwargs.pop("stage_through_host", None)
    if staged is None:
        staged = device.type == "cuda" and dist.get_backend(group) == "gloo"
    if staged:
        with torch.no_grad():                                  # rank 0's parameters and buffers, like DDP's own init sync
            for t in list(model.parameters()) + list(model.buffers()):
                host = t.detach().cpu()
                dist.broadcast(host, dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                if dist.get_rank(group) != 0:
                    t.copy_(host)
        kwargs.setdefault("init_sync", False)
        kwargs.setdefault("broadcast_buffers", False)
    ddp = DistributedDataParallel(model, device_ids=[device.index] if device.type == "cuda" else None, process_group=group, **kwargs)
    if staged:
        world = dist.get_world_size(group)

        def staged_allreduce(state, bucket):
            buf = bucket.buffer()
            host = buf.detach().cpu()
            dist.all_reduce(host, group=group)
            buf.copy_(host.div_(world))
            fut = torch.futures.Future()
            fut.set_result(buf)
            return fut
        ddp.register_comm_hook(None, staged_allreduce)
    return ddp


def _backward(loss):
    from .autograd import backward
    backward(loss)


def _default_loss(out, x):
    """The distortion term alone, mean((xHat - x)^2) (mcquic/loss/__init__.py:62), through this library's own reduction."""
    from .autograd import mse_loss
    return mse_loss(out[0], x)


_MEMSET_NODES_OK = {}


def memset_nodes_replay_correctly(device=None, refresh: bool = False) -> bool:
    """Does THIS process replay the memset nodes of a captured hipGraph correctly?  (ROCm 7.2 does not once eager blit work has
    run between replays, unless DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 was in the environment when the HIP runtime started -- see
    mcquic_amd/__init__.py.)  The check is the defect's own reproducer at small scale: a captured ATen reduction large enough for
    its two-stage path (whose semaphores are zeroed by a memset node), replayed over a changing input with 700 small eager
    reductions + device-to-host copies in between; ~0.1 s, once per process and device.  GraphedTrainStep consults it for a caller-supplied
    loss function -- the default loss and every kernel of this package are free of memset nodes either way."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index in _MEMSET_NODES_OK and not refresh:
        return _MEMSET_NODES_OK[dev.index]
    with torch.no_grad():
        big = torch.rand(1 << 21, device=dev)
        small = [torch.ones(1000 + 37 * i, device=dev) for i in range(700)]
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            big.sum()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        mode = dict(capture_error_mode="thread_local") if (dist.is_available() and dist.is_initialized()) else {}
        with torch.cuda.graph(g, **mode):
            out = big.sum()
        ok = True
        for _ in range(3):
            big.add_(1.0)
            g.replay()
            sum(int(torch.isfinite(t).all()) for t in small)
            torch.cuda.synchronize(dev)
            want = float(big.sum())
            ok = ok and abs(float(out) - want) <= 1e-4 * abs(want)
    _MEMSET_NODES_OK[dev.index] = ok
    return ok


class GraphedTrainStep:
    """One rank's data-parallel training step (BASELINE configs[4]: `torchrun` + DDP in the reference, mcquic/train/ddp.py:79-95;
    its gradient exchange is overlapped with the backward pass bucket by bucket, mcquic/train/trainer.py:94,105) with the host out
    of the loop.  DDP's eager step is bound by the host here -- ~1 000 launches of ~20 us each per step, with two host cores per
    rank on an 8-GPU node -- and a step captured WITH DDP's hooks replays slower than eager (the bucket all-reduces live on RCCL's
    stream: a multi-stream graph, docs/experiments.md).  So the step is cut where the exchange is:

        segment graphs  forward + backward of the local shard on a static input, captured as `segments` hipGraphs in BACKWARD
                        order -- [forward, loss, decoder backward] -> [quantizer backward] -> [encoder backward] -- each ending
                        with the copy of its parameters' gradients into ITS slice of ONE flat float32 buffer; this rank's code
                        counts of the frequency-EMA update go into one int64 buffer (known after the forward)
        exchange        outside any graph: as soon as a segment's replay is enqueued, the all-reduce of its slice is started
                        (async on RCCL's stream, which waits for exactly that segment) while the next segment replays --
                        the slices are 21.5 / 166.3 / 14.4 MB for the qp=2 model (decoder / quantizer / encoder; 202.2 MB =
                        50 558 738 float32 gradients in all), so only the encoder's 14 MB are exchanged after the last kernel;
                        one message per segment is what a point-to-point xGMI ring wants, not 25 MB buckets
        post graph      gradients / world size, the optimizer's update, the frequency EMA from the GLOBAL counts
                        (mcquic/modules/entropyCoder.py:28-44 all-reduces them inside forward; here that collective is deferred)

    `segments=1` is the whole step as one graph followed by one all-reduce (what a single rank runs: nothing to overlap with);
    the default is 3 under a process group of more than one rank, when the model has the Compressor's three stages and the
    loss is the default one (a custom `loss_fn` sees (xHat, yHat, codes, logits) with yHat / logits as LEAVES of the decoder's
    segment: its gradients on them are carried into the quantizer's segment).
    Link time of the exchange at 8 ranks: a ring all-reduce moves 2 x 7/8 x 202 MB = 354 MB through every GPU; ~1.2 ms if RCCL
    spreads it over the seven xGMI links of a GPU (~300 GB/s bus bandwidth), ~4.6 ms on one link direction --
    5-20 % of a 22 ms step when serial, which is why it is overlapped.

    The forward of the next replay starts with the grouped re-pack of every operand stream the update made stale
    (`Conv2d.repack_stale`, in place): all parameters are marked changed before the capture so that those launches are recorded.
    The step keeps every operand stream, count sink and static buffer its graphs have addresses of alive for its own lifetime,
    so using the model eagerly in between (`invalidate()`, then evaluation / `compress`, then more steps) is safe.

        step = GraphedTrainStep(model, optimizer, example_x)        # after torch.distributed is initialised (or not at all)
        loss = step(x)                                              # x: this rank's shard, shape of example_x
        step.invalidate(); model.eval(); ...; model.train()         # evaluating in between: eager code re-packs its streams
        step.close()                                                # done: the step cannot be replayed any more

    Under more than one rank the constructor broadcasts rank 0's parameters and buffers (as DDP does), so the replicas start
    equal.  An optimizer that already holds state (resumed from a checkpoint) keeps it: its tensors are restored after the
    throw-away update that brings missing state into existence.
    A captured update bakes in whatever the optimizer read on the host at capture time: give a scheduled learning rate to the
    optimizer as a TENSOR (torch.optim reads it on the device then; Adam / AdamW additionally need `capturable=True`), or pass
    `capture_post=False` and the update (with the EMA) runs eagerly after the exchange -- ~15 launches, still no per-layer host work.
    `max_grad_norm` clips the averaged gradient by its global norm in front of the update (the reference's step does,
    mcquic/train/trainer.py:280; torch.nn.utils.clip_grad_norm_'s arithmetic over the flat buffer, on the device); the norm before
    clipping is `step.grad_norm` (a 0-dim device tensor, valid after each call).
    """

    def __init__(self, model: torch.nn.Module, optimizer, example_x: torch.Tensor, loss_fn=None, group=None,
                 forward_kwargs: dict | None = None, warmup: int = 2, capture_post: bool = True, segments: int | None = None,
                 broadcast: bool = True, max_grad_norm: float | None = None):
        if not example_x.is_cuda:
            raise RuntimeError("GraphedTrainStep needs a HIP device (hipGraph capture)")
        self.model, self.optimizer, self.group = model, optimizer, group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.x = example_x.detach().clone()
        self.kwargs = dict(forward_kwargs or {})
        staged = all(hasattr(model, a) for a in ("_encoder", "_quantizer", "_decoder", "_repackStale")) and set(self.kwargs) <= {"uniforms"}
        if segments is None:
            segments = 3 if (self.world > 1 and staged) else 1
        if segments not in (1, 3):
            raise ValueError("segments must be 1 (one graph, one all-reduce) or 3 (decoder / quantizer / encoder backward)")
        if segments == 3 and not staged:
            raise ValueError("segments=3 needs a model with _encoder / _quantizer / _decoder stages and no forward_kwargs but `uniforms`")
        self.segments = segments
        if max_grad_norm is not None and not max_grad_norm > 0:
            raise ValueError("max_grad_norm must be positive (or None: no clipping)")
        self.max_grad_norm = max_grad_norm
        self.grad_norm = None                                 # 0-dim device tensor: the global gradient norm of the last step, before clipping
        if loss_fn is not None and not memset_nodes_replay_correctly(example_x.device):
            import warnings
            warnings.warn("this process replays memset nodes of captured hipGraphs wrongly (ROCm 7.2; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 was not "
                          "in the environment when the HIP runtime started): a library reduction inside `loss_fn` (x.mean(), x.sum() over "
                          "~1e5+ elements) may return stale values from a replay.  Import mcquic_amd before the first device call, or set the "
                          "variable yourself.", RuntimeWarning, stacklevel=2)
        self.loss_fn = loss_fn or _default_loss
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.coders = [m for m in model.modules() if hasattr(m, "deferCounts")]
        self.closed = False
        if self.world > 1 and broadcast:
            self._broadcast_state()
        model.train()
        for c in self.coders:
            c.deferCounts(True)
        from . import ops
        from .nn import blocks
        streams = blocks._BRANCH_STREAMS
        blocks._BRANCH_STREAMS = False                       # nested stream forks crash hipGraph capture (ROCm 7.2): one stream
        # (under a process group RCCL's watchdog thread may query events while we capture: thread-local mode keeps a capture from
        #  being invalidated by what OTHER threads do; the autograd thread's launches are captured either way -- capture follows
        #  the stream, not the thread)
        mode = dict(capture_error_mode="thread_local") if (dist.is_available() and dist.is_initialized()) else {}
        # warm-up and capture share ONE side stream: autograd's gradient accumulators remember the stream they were created on, and
        # a capture on another stream would pull that one (the legacy default stream, if the warm-up ran there) into the graph
        self._stream = torch.cuda.Stream(self.x.device)
        self._stream.wait_stream(torch.cuda.current_stream(self.x.device))
        try:
            ops.section_trace(True)                          # which copy of its operand stream every conv launch of the step reads
            try:
                with torch.cuda.stream(self._stream):
                    for _ in range(max(1, warmup)):          # caches, workspaces, the coders' count sinks
                        for k in range(self.segments):
                            self._segment(k)
                torch.cuda.current_stream(self.x.device).wait_stream(self._stream)
            finally:
                ops.section_trace(False)
            masks, pinned = {}, []
            for m in model.modules():
                for pk in (m.__dict__.get("_packed"), getattr(m.__dict__.get("_dgradCache"), "packed", None)):
                    if pk is not None and hasattr(pk, "wp"):
                        pinned.append(pk.wp)                 # the graphs re-pack into / read from these addresses for as long as they live
                        used = ops.sections_used(pk)
                        if used:
                            masks[id(pk)] = used
            model.__dict__["_packMasks"] = masks             # the captured re-pack refreshes those copies only (replays repeat the launches)
            # flat layout = segment order (backward order), parameters without a gradient left out
            groups = self._param_groups()
            self.live_groups = [[p for p in g if p.grad is not None] for g in groups]
            self.live = [p for g in self.live_groups for p in g]
            self.flat = torch.empty(sum(p.numel() for p in self.live), dtype=torch.float32, device=self.x.device)
            self.slices, off = [], 0
            for g in self.live_groups:
                n = sum(p.numel() for p in g)
                self.slices.append(self.flat[off: off + n])
                off += n
            self._gather_tables()
            self._init_optimizer_state()
            torch.cuda.synchronize()
            with torch.no_grad():
                torch._foreach_add_(self.params, 0.0)        # every parameter "changed": the capture records all re-packs
            for p in self.params:
                p.grad = None
            self.graphs = []
            pool = torch.cuda.graph_pool_handle() if self.segments > 1 else None
            self.counts = None
            for k in range(self.segments):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, stream=self._stream, **mode):
                    self._segment(k)
                    if self.live_groups[k]:
                        self._gather(k)
                    if k == 0:
                        sinks = [c.countSink() for c in self.coders]
                        # (one coder per model: its count buffer itself -- the sampling kernels add into it, nothing is copied)
                        self.counts = (sinks[0] if len(sinks) == 1 else torch.cat(sinks)) if sinks else None
                self.graphs.append(g)
            self.graph = self.graphs[0]
            self._carry = None                               # (the autograd graph between segments: only needed while capturing)
            self._pinned = pinned + [c.countSink() for c in self.coders]
        except BaseException:
            for c in self.coders:
                c.deferCounts(False)
            raise
        finally:
            model.__dict__.pop("_packMasks", None)
            blocks._BRANCH_STREAMS = streams
        off = 0
        for p in self.live:                                  # the optimizer reads the reduced gradients
            p.grad = self.flat[off: off + p.numel()].view_as(p)
            off += p.numel()
        self.post = None
        if hasattr(self.optimizer, "prepare"):                # (mcquic_amd.optim.Adam: device tables for the flat gradient views, built
            self.optimizer.prepare()                          #  outside the capture)
        if capture_post:
            before = self._snapshot_optimizer_state()
            try:
                post = torch.cuda.CUDAGraph()
                with torch.cuda.graph(post, stream=self._stream, **mode):
                    self._post()
                self.post = post
            except Exception:                                # an optimizer that cannot be captured: its update runs eagerly
                torch.cuda.synchronize()
                self.post = None
                self._restore_optimizer_state(before)        # (host-side counters may have moved before the capture gave up)

    def _gather(self, k: int):
        """Segment k's gradients into its slice of the flat buffer: ONE launch over device tables (mcq_gather_flat_f32) instead of
        torch.cat's six.  Called inside the capture: the gradients' addresses (allocations of the captured backward pass, the same
        on every replay) reach the device table through a copy node from a pinned host tensor this object keeps alive."""
        from . import _lib
        from .ops import check, _guard, _stream
        params = self.live_groups[k]
        lib = _lib.load()
        dev = self.flat.device
        tb = self._gatherTables[k]
        grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in params]
        tb["grads"] = grads                                   # (alive as long as the tables point at them)
        tb["host"].copy_(torch.tensor([g.data_ptr() for g in grads], dtype=torch.int64))
        tb["ptrs"].copy_(tb["host"], non_blocking=True)       # (a copy NODE inside the capture: re-read from the pinned tensor on every replay)
        with _guard(dev):
            check(lib.mcq_gather_flat_f32(tb["ptrs"].data_ptr(), self.slices[k].data_ptr(), tb["offs"].data_ptr(), tb["numel"].data_ptr(),
                                          tb["blk_t"].data_ptr(), tb["blk_f"].data_ptr(), tb["nblocks"], _stream()), "mcq_gather_flat_f32")

    def _gather_tables(self):
        """The static side of `_gather` (sizes, offsets, chunk tables, the pinned pointer buffer), built OUTSIDE the captures: host-to-device
        copies from pageable memory invalidate a capture."""
        from . import _lib
        chunk = _lib.load().mcq_adam_chunk()
        dev = self.flat.device
        self._gatherTables = {}
        for k, params in enumerate(self.live_groups):
            if not params:
                continue
            sizes = [p.numel() for p in params]
            offs, at = [], 0
            for n in sizes:
                offs.append(at)
                at += n
            blk_t, blk_f = [], []
            for i, n in enumerate(sizes):
                for first in range(0, n, chunk):
                    blk_t.append(i)
                    blk_f.append(first)
            self._gatherTables[k] = dict(
                host=torch.empty(len(params), dtype=torch.int64, pin_memory=True), ptrs=torch.empty(len(params), dtype=torch.int64, device=dev),
                offs=torch.tensor(offs, dtype=torch.int64).to(dev), numel=torch.tensor(sizes, dtype=torch.int64).to(dev),
                blk_t=torch.tensor(blk_t, dtype=torch.int32).to(dev), blk_f=torch.tensor(blk_f, dtype=torch.int64).to(dev), nblocks=len(blk_t))
        torch.cuda.synchronize()

    # ---- replica state ---------------------------------------------------------------------------------------------------
    def _broadcast_state(self):
        """Rank 0's parameters and buffers to every rank (what DDP's constructor does): replicas that start from different
        weights would average gradients of diverging models without any error."""
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        seen = set()
        with torch.no_grad():
            for t in list(self.model.parameters()) + list(self.model.buffers()):
                if id(t) in seen:
                    continue
                seen.add(id(t))
                buf = _staged(t.detach(), self.group)
                buf = buf if buf.is_contiguous() else buf.contiguous()
                dist.broadcast(buf, src, group=self.group)
                if buf.data_ptr() != t.data_ptr():
                    t.copy_(buf)

    # ---- optimizer state ---------------------------------------------------------------------------------------------------
    def _snapshot_optimizer_state(self):
        return {(id(p), k): v.detach().clone() for p, st in self.optimizer.state.items() for k, v in st.items() if torch.is_tensor(v)}

    def _restore_optimizer_state(self, snap):
        """State tensors that existed at the snapshot get their values back; the ones created since are zeroed -- which is the
        state a fresh optimizer starts from (SGD with momentum, Adam / AdamW; an optimizer whose initial state is not zeros
        needs its own warm-up)."""
        with torch.no_grad():
            for p, st in self.optimizer.state.items():
                for k, v in st.items():
                    if torch.is_tensor(v):
                        old = snap.get((id(p), k))
                        if old is None:
                            v.zero_()
                        else:
                            v.copy_(old)

    def _init_optimizer_state(self):
        """An optimizer creates its state (momentum buffers, Adam's moments and step counter) inside its first `step()`; created
        inside the capture they would be re-created by every replay.  One throw-away update with the warm-up gradients brings
        them into existence; then the parameters and every state tensor that existed BEFORE (an optimizer resumed from a
        checkpoint: momentum, moments, step counters) are restored and only the newly created ones are zeroed."""
        saved = [p.detach().clone() for p in self.params]
        snap = self._snapshot_optimizer_state()
        self.optimizer.step()
        with torch.no_grad():
            for p, v in zip(self.params, saved):
                p.copy_(v)
        self._restore_optimizer_state(snap)

    # ---- the step's segments ---------------------------------------------------------------------------------------------
    def _param_groups(self):
        """Parameters by segment, in the order the backward pass finishes them: decoder, quantizer, encoder (+ anything else)."""
        if self.segments == 1:
            return [list(self.params)]
        taken, groups = set(), []
        for stage in (self.model._decoder, self.model._quantizer):
            g = [p for p in stage.parameters() if p.requires_grad and id(p) not in taken]
            taken.update(id(p) for p in g)
            groups.append(g)
        groups.append([p for p in self.params if id(p) not in taken])
        return groups

    def _segment(self, k: int):
        if self.segments == 1:
            self.loss = self._forward_backward()
            return
        from . import ops
        m = self.model
        if k == 0:
            for p in self.params:
                p.grad = None
            m._repackStale()
            y = m._trainEncode(self.x) if hasattr(m, "_trainEncode") else m._encoder(self.x)     # (no padding in the training forward, compressor.py:39)
            yl = y.detach().requires_grad_()
            tw = ops.silu_twin(y)
            if tw is not None:
                ops.set_silu_twin(yl, tw)
            yHat, codes, logits = m._quantizer(yl, self.kwargs.get("uniforms"))
            yh = yHat.detach().requires_grad_()
            tw = ops.silu_twin(yHat)
            if tw is not None:
                ops.set_silu_twin(yh, tw)
            lgl = [lg.detach().requires_grad_() if (torch.is_tensor(lg) and lg.requires_grad) else lg for lg in logits]
            xHat = m._decoder(yh)
            loss = self.loss_fn((xHat, yh, codes, lgl), self.x)
            _backward(loss)                                   # decoder parameters, d yHat (, d logits)
            self.loss = loss.detach()
            self._carry = (y, yl, yHat, yh, logits, lgl)
        elif k == 1:
            y, yl, yHat, yh, logits, lgl = self._carry
            outs, grads = [yHat], [yh.grad]
            for lg, leaf in zip(logits, lgl):
                if torch.is_tensor(leaf) and leaf is not lg and leaf.grad is not None:
                    outs.append(lg)
                    grads.append(leaf.grad)
            from .autograd import run_backward
            run_backward(outs, grads)                         # quantizer parameters, d y
        else:
            y, yl = self._carry[0], self._carry[1]
            from .autograd import run_backward
            run_backward([y], [yl.grad])                      # encoder parameters
            self._carry = None

    def _forward_backward(self):
        for p in self.params:
            p.grad = None
        out = self.model(self.x, **self.kwargs)
        loss = self.loss_fn(out, self.x)
        _backward(loss)
        return loss.detach()

    def _post(self):
        if self.world > 1:
            self.flat.mul_(1.0 / self.world)
        if self.max_grad_norm is not None:                    # trainer.py:280 `clip_grad_norm(4.0)`: global norm of the AVERAGED gradient,
            from . import ops                                 # two launches over the flat buffer, nothing read by the host
            self.grad_norm = ops.clip_by_norm_(self.flat, self.max_grad_norm)
        self.optimizer.step()
        off = 0
        for c in self.coders:
            n = c.countSink().numel()
            c.applyCounts(self.counts[off: off + n])
            off += n

    def _start_all_reduce(self, buf: torch.Tensor):
        """Sum `buf` over the ranks without waiting for it: under RCCL an async all-reduce (its stream waits for what the current
        stream has enqueued so far -- the segment that just filled `buf` -- and the returned work is joined before the post
        graph); under gloo (tests) the blocking host-staged form."""
        if buf.numel() == 0:
            return None
        if buf.is_cuda and dist.get_backend(self.group) == "gloo":
            all_reduce_(buf, self.group)
            return None
        return dist.all_reduce(buf, group=self.group, async_op=True)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if self.closed:
            raise RuntimeError("GraphedTrainStep: the step was closed (its count sinks are gone); build a new one")
        if tuple(x.shape) != tuple(self.x.shape):             # (`copy_` would BROADCAST a smaller batch into the static input silently)
            raise RuntimeError(f"GraphedTrainStep was captured for shards of shape {tuple(self.x.shape)}, got {tuple(x.shape)}")
        self.x.copy_(x, non_blocking=True)
        works = []
        for k, g in enumerate(self.graphs):
            g.replay()
            if self.world > 1:
                works.append(self._start_all_reduce(self.slices[k]))       # ... while the next segment replays
                if k == 0 and self.counts is not None:
                    works.append(self._start_all_reduce(self.counts))
        for w in works:
            if w is not None:
                w.wait()                                      # (the current stream waits; the host does not)
        if self.post is not None:
            self.post.replay()
        else:
            self._post()
        for c in self.coders:
            c.resetFreqAndCDF()
        return self.loss

    def invalidate(self):
        """Replays change the parameters without the host seeing it (no version counter moves): mark every parameter changed so
        that the next EAGER use of the model (evaluation, a checkpoint's `compress`) re-packs its operand streams -- they are one
        update behind after a replay.  Call before using the model outside this step; `close()` does.  The step itself stays
        usable: its graphs re-pack the copies they read at the start of every replay, into streams this object keeps alive."""
        with torch.no_grad():
            torch._foreach_add_(self.params, 0.0)

    def close(self):
        """Back to the eager step for good: the coders update their EMA inside forward again, `.grad` is whatever the last step
        left, and this object refuses further replays (its count sinks are released)."""
        if self.closed:
            return
        self.closed = True
        for c in self.coders:
            c.deferCounts(False)
        self.invalidate()
        self._pinned = []
