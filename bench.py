#!/usr/bin/env python3
"""Throughput of the Compressor encode+decode hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = `encode` (images -> codes) followed by `decode` (codes -> pixels) of one synthetic batch of
32 images of 768x512 per GPU through the qp=2 model `Compressor(128, 2, [8192, 2048, 512])`, tensor path only
(the protocol of the reference's `Validator.speed`, mcquic/validate/validator.py:60-97, minus the dead entropy
coder).  Inputs are resident in HBM before the timed region.  Image batches shard across ranks with no
data-path collective ("weak" scaling: 32 images per GPU); RCCL is used only for the barrier / max-time reduce.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel `conv_mfma_kernel` (fp32 MFMA bound):
algorithmic FLOPs of the conv launches in the timed steps / the time during which a conv kernel was in flight
(union of the per-launch [start, end] intervals, HIP events on the launch streams), both measured live in the timed
region -- on every conv launch of its first 8 steps (`roofline.event_steps`; the default run has 5): keeping ~130 k HIP
events alive over a 200-step run slowed the launches themselves by 3 %.  `cpu_baseline` times the CPU oracle (oracle/mcquic_ref.py, a plain PyTorch
restatement of the reference proven bit-equal to it) on this host's cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# hardware queues for the branch streams next to RCCL's own streams (see mcquic_amd/__init__.py); must precede HIP init
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# graph replays through the ordinary command path: ROCm 7.2's recorded launch packets replay memset nodes wrongly (same file)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MATRIX_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
HBM_PEAK_GBS = 8000.0
BATCH_PER_GPU = 32
H, W = 768, 512
MODEL = dict(channel=128, m=2, k=[8192, 2048, 512])


class ConvProfiler:
    """Brackets every conv launch with HIP events on the stream it is launched on (no host sync in the region).

    Branch streams let two conv kernels overlap, so per-launch durations are not additive: the busy time reported
    is the UNION of the [start, end] intervals (time during which at least one conv kernel was in flight), from
    event timestamps relative to one base event."""

    MAX_STEPS = 8       # timed steps that carry events (132 k live HIP events over a 200-step run slowed the launches 3 %)

    def __init__(self):
        self.records = []           # (start, end, algorithmic flops, bytes, executed MFMA flops)
        self.base = None
        self._descs = []            # flags of the descriptors built since the current launch wrapper started
        self.steps = 0              # timed steps bracketed so far
        self.active = True

    def next_step(self):
        """Call at the start of every timed step: the first MAX_STEPS of them are bracketed, the rest run bare."""
        self.active = self.steps < self.MAX_STEPS
        if self.active:
            self.steps += 1

    def install(self):
        from mcquic_amd import ops
        self._orig = ops.conv2d
        prof = self
        self.base = torch.cuda.Event(enable_timing=True)
        self.base.record()
        self._orig_desc = ops._conv_desc

        def wrapped_desc(*a, **kw):
            out = prof._orig_desc(*a, **kw)
            prof._descs.append(int(out[0].flags))
            return out
        ops._conv_desc = wrapped_desc

        def executed(flops, flags):
            """MFMA work actually issued: the Winograd forms run 16 of 36 (F(2x2, 3x3)) / 12 of 18 (F(2, 3)) k-steps."""
            if flags & (ops.CONV_WINOGRAD2D | ops.CONV_WINOGRAD2D16):
                return flops * 16.0 / 36.0
            if flags & ops.CONV_WINOGRAD:
                return flops * 12.0 / 18.0
            return flops

        def wrapped(x, w, stride=1, **kw):
            if not prof.active:
                return prof._orig(x, w, stride, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            del prof._descs[:]
            s.record()
            y = prof._orig(x, w, stride, **kw)
            e.record()
            n, cin, h, wd = x.shape
            pad = w.ksize // 2
            ho = (h + 2 * pad - w.ksize) // stride + 1
            wo = (wd + 2 * pad - w.ksize) // stride + 1
            flops = 2.0 * n * ho * wo * w.cout * cin * w.ksize * w.ksize
            if any(kw.get(key) is not None for key in ("post_gdn", "post_igdn", "post_gate")):
                flops += 2.0 * y.numel() * y.shape[1]       # the [128, 128] 1x1 layer that runs inside this launch (MCQ_CONV_POST_*)
            # algorithmic HBM bytes of the fused op: input and output once, plus every output-shaped side tensor its
            # contract needs (residual / multiplier / gate reads, the SiLU twin write)
            sides = sum(1 for key in ("res", "gdn_mul", "igdn_mul", "gate_mul", "gate_id", "mul") if kw.get(key) is not None)
            sides += 1 if kw.get("dual_silu") else 0
            nbytes = 4.0 * (x.numel() + y.numel() * (1 + sides))
            prof.records.append((s, e, flops, nbytes, executed(flops, prof._descs[-1] if prof._descs else 0)))
            return y

        self._orig_multi = ops.conv2d_multi

        def wrapped_multi(xs, ws, stride=1, per_problem=None, **shared):
            if not prof.active:
                return prof._orig_multi(xs, ws, stride, per_problem=per_problem, **shared)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ops.conv2d = prof._orig                     # (the multi launch is ONE launch: no per-problem brackets inside)
            del prof._descs[:]
            s.record()
            ys = prof._orig_multi(xs, ws, stride, per_problem=per_problem, **shared)
            e.record()
            ops.conv2d = wrapped
            flops = nbytes = done = 0.0
            for i, (x, w, y) in enumerate(zip(xs, ws, ys)):
                n, cin, h, wd = x.shape
                fl = 2.0 * y.shape[0] * y.shape[2] * y.shape[3] * w.cout * cin * w.ksize * w.ksize
                flops += fl
                done += executed(fl, prof._descs[i] if i < len(prof._descs) else 0)
                kw = dict(shared, **((per_problem or [{}] * len(xs))[i]))
                sides = sum(1 for key in ("res", "gdn_mul", "igdn_mul", "gate_mul", "gate_id", "mul") if kw.get(key) is not None)
                sides += 1 if kw.get("dual_silu") else 0
                nbytes += 4.0 * (x.numel() + y.numel() * (1 + sides))
            prof.records.append((s, e, flops, nbytes, done))
            return ys

        ops.conv2d = wrapped
        ops.conv2d_multi = wrapped_multi
        return self

    def remove(self):
        from mcquic_amd import ops
        ops.conv2d = self._orig
        ops.conv2d_multi = self._orig_multi
        ops._conv_desc = self._orig_desc

    def summary(self):
        iv = sorted((self.base.elapsed_time(r[0]), self.base.elapsed_time(r[1])) for r in self.records)
        busy, cur_s, cur_e = 0.0, None, None
        for a, b in iv:
            if cur_e is None or a > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = a, b
            else:
                cur_e = max(cur_e, b)
        if cur_e is not None:
            busy += cur_e - cur_s
        fl = sum(r[2] for r in self.records)
        by = sum(r[3] for r in self.records)
        return dict(launches=len(self.records), ms=busy, flops=fl, bytes=by, steps=max(self.steps, 1),
                    mfma_flops=sum(r[4] for r in self.records), sum_ms=sum(r[0].elapsed_time(r[1]) for r in self.records))


def pmc_traffic():
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summary (separate
    FETCH_SIZE / WRITE_SIZE passes of this same command; profiles/rNN_pmc.json).  None if absent."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc.json")), reverse=True):     # the latest round's passes
        try:
            d = json.load(open(path))
            d["_file"] = os.path.basename(path)
            from mcquic_amd.build import csrc_sha
            d["_stale"] = d.get("csrc_sha") != csrc_sha()       # collected on other kernel sources than the ones running now
            return d
        except (OSError, ValueError):
            continue
    return None


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask and cgroup CPU quota, not the host's core count
    (oversubscribing a quota-limited container with one thread per host core is pathologically slow)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_baseline(state_dict_cpu, x_cpu, min_seconds=12.0, max_iters=12):
    """The CPU oracle on this host: batch len(x_cpu) of 768x512, 1 warm-up pass, then timed encode+decode passes
    until `min_seconds` of CPU work have been measured (bounded sample: about 10-30 s in all)."""
    from oracle import mcquic_ref as R
    cores = usable_cores()
    torch.set_num_threads(cores)
    codes = R.encode(state_dict_cpu, x_cpu)          # warm-up (also the parity sample)
    pixels = R.decode(state_dict_cpu, codes)
    t0 = time.perf_counter()
    iters = 0
    while iters < 2 or (time.perf_counter() - t0 < min_seconds and iters < max_iters):
        c = R.encode(state_dict_cpu, x_cpu)
        R.decode(state_dict_cpu, c)
        iters += 1
    dt = time.perf_counter() - t0
    base = {"value": round(len(x_cpu) * iters / dt, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle/mcquic_ref.py (PyTorch-CPU restatement, bit-equal to the reference in the build container): "
                      f"batch {len(x_cpu)}x3x{H}x{W}, 1 warm-up + {iters} timed encode+decode passes, {dt:.1f} s"}
    return base, codes, pixels


def parity_report(gpu_codes, gpu_pixels, cpu_codes, cpu_pixels):
    from oracle import mcquic_ref as R
    mism = sum(int((a.cpu() != b).sum()) for a, b in zip(gpu_codes, cpu_codes))
    total = sum(b.numel() for b in cpu_codes)
    err = float((gpu_pixels.cpu() - cpu_pixels).abs().max())
    psnr = float(R.psnr(R.detransform(gpu_pixels.cpu()), R.detransform(cpu_pixels)).min())
    return {"code_mismatches": mism, "codes": total, "decode_max_abs_err": err,
            "psnr_gpu_vs_cpu_u8_min_db": round(psnr, 2), "note": "decode compared from the CPU oracle's codes"}


def _timed(fn, steps, warmup=1):
    """ms per call: `warmup` untimed calls, then `steps` calls between two events on the current stream."""
    for _ in range(warmup):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def neon_figures(dev, dense: bool = True, infer_batch: int = 8, train_batch: int = 4, side: int = 512):
    """`secondary.neon`: encode + decode rate and one training step (forward + Gumbel straight-through backward, eager) of
    Neon(32, 4096, [16, 8, 4, 2, 2], denseNorm) on `side` x `side` images, with the conv kernels' share of fp32-MFMA peak and the
    training step's peak device memory (what the reference's `checkpoint_wrapper` on encoder / decoder is there to bound,
    mcquic/modules/compressor.py:230-231)."""
    from mcquic_amd import Neon
    from mcquic_amd.autograd import backward, mse_loss
    torch.manual_seed(3407)
    cfg = (32, 4096, [16, 8, 4, 2, 2])
    model = Neon(*cfg, dense).to(dev).eval()
    g = torch.Generator().manual_seed(1)
    x = (torch.rand((infer_batch, 3, side, side), generator=g) * 2 - 1).to(dev)
    prof = ConvProfiler().install()
    prof.MAX_STEPS = 2
    prof.active = False
    model.decode(model.encode(x))
    prof.active = True
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        prof.next_step()
        model.decode(model.encode(x))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 2 * 1e3
    prof.remove()
    c = prof.summary()
    out = {"model": f"Neon{cfg} denseNorm={dense}, random-init weights", "side": side, "infer_batch": infer_batch,
           "images_s": round(infer_batch / ms * 1e3, 2), "ms_per_step": round(ms, 3),
           "frac_of_peak": round(c["mfma_flops"] / (c["ms"] * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4),
           "conv_gflop_per_image": round(c["flops"] / c["steps"] / infer_batch / 1e9, 2),
           "whole_step_frac": round(c["flops"] / c["steps"] / (ms * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4)}
    # training step, the reference's per-GPU batch
    model.train()
    xt = x[:train_batch].contiguous()
    prof = ConvProfiler().install()                             # conv FLOPs of one training-mode forward (under no_grad: same launches' shapes)
    prof.MAX_STEPS = 1
    prof.next_step()
    with torch.no_grad():
        model(xt)
    torch.cuda.synchronize()
    prof.remove()
    fwd_flops = prof.summary()["flops"]

    def step():
        for p in model.parameters():
            p.grad = None
        xHat = model(xt)[0]
        loss = mse_loss(xHat, xt)
        backward(loss)
        return loss.detach()                                    # (no handle on the autograd graph outlives the step: a later capture needs that)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats(dev)
    t0 = time.perf_counter()
    for _ in range(3):
        loss = step()
    torch.cuda.synchronize()
    tms = (time.perf_counter() - t0) / 3 * 1e3
    peak_bytes = torch.cuda.max_memory_allocated(dev)
    # ... and the same step as one hipGraph (what parallel.GraphedTrainStep replays; the eager figure is bound by the host)
    gms = None
    from mcquic_amd.nn import blocks as _blocks
    streams = _blocks._BRANCH_STREAMS
    try:
        _blocks._BRANCH_STREAMS = False                        # (nested stream forks crash hipGraph capture on ROCm 7.2)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        for p in model.parameters():
            p.grad = None
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        gms = _timed(graph.replay, 5, warmup=1)
        del graph
    except Exception as exc:                                    # noqa: BLE001
        out["train_graph_error"] = repr(exc)[:200]
    finally:
        _blocks._BRANCH_STREAMS = streams
    if gms is not None:
        out["train_step_graph_ms"] = round(gms, 3)
        out["train_graph_frac_of_peak"] = round(3.0 * fwd_flops / (gms * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4)
    out.update({"train_batch": train_batch, "train_step_ms": round(tms, 3), "train_loss": round(float(loss), 6),
                "train_frac_of_peak": round(3.0 * fwd_flops / (tms * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4),
                "train_tflop_per_step": round(3.0 * fwd_flops / 1e12, 3),
                "train_peak_memory_gb": round(peak_bytes / 2 ** 30, 2),
                "device_memory_gb": round(torch.cuda.get_device_properties(dev).total_memory / 2 ** 30, 1),
                "checkpoint_wrapper": "not applied: the step's peak memory without recompute is the figure above"})
    del model, x, xt
    torch.cuda.empty_cache()
    return out


def secondary(model, x, dev, cpu_codes, cpu_pix, nb):
    """Round results the headline does not show, measured in the same run (about 40 s; every entry is independent and
    carries its own `error` if it fails -- the headline line never depends on them):
      winograd2d   the OPT-IN F(2x2, 3x3) mode on the same batch: images/s, the fraction of the fp32-MFMA peak in REAL issued
                   work (never the direct form's FLOPs), and the code / pixel parity of the same 4 images against the oracle
      train_step   BASELINE configs[4]'s per-GPU work: forward + Gumbel straight-through backward of the qp=2 model on
                   8 x 256x256 crops as ONE captured hipGraph, 10 replays
      vq_config4   BASELINE configs[3]: M=4, K=4096, D=256 distance + argmin on 49 152 vectors per codebook
      batch1       one 768x512 image, encode+decode as hipGraph replays (latency)
      host_buffers the headline workload with images and reconstructions in pinned host memory (the PCIe-inclusive rate)
      speed_protocol the reference's own Mpps protocol (`Validator.speed`: batch 10, compress / decompress with the rANS byte streams)"""
    from mcquic_amd import Compressor, ops
    from mcquic_amd.nn import blocks
    sec = {}
    # ---- the headline workload with the images in (pinned) HOST memory on both sides -----------------------------------------
    try:
        xh = torch.empty(x.shape, dtype=x.dtype, pin_memory=True).copy_(x)
        yh = torch.empty(x.shape, dtype=x.dtype, pin_memory=True)

        def host_step():
            yh.copy_(model.decode(model.encode(xh.to(dev, non_blocking=True))), non_blocking=True)
        host_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            host_step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        mb = x.numel() * x.element_size() / 1e6
        sec["host_buffers"] = {"images_s": round(x.shape[0] / ms * 1e3, 2), "ms_per_step": round(ms, 3),
                               "h2d_mb_per_step": round(mb, 1), "d2h_mb_per_step": round(mb, 1),
                               "note": "PCIe-inclusive: float32 images copied from pinned host memory, encode + decode, reconstructions "
                                       "copied back to pinned host memory, nothing overlapped; never the headline `value` (inputs resident in HBM)"}
        # ... and as a caller with a stream of batches would run it: parallel.prefetch copies batch i + 1 while batch i is in the
        # kernels; the reconstructions leave on the COMPUTE stream (a copy-out on a second side stream measures slower than the
        # serial copy on this stack, tools/probes/prefetch_overlap.py: +8 ms per batch against +3.6)
        from mcquic_amd import parallel

        def piped(n_batches):
            for xd in parallel.prefetch([xh] * n_batches, dev):
                yh.copy_(model.decode(model.encode(xd)), non_blocking=True)
        piped(3)                                              # (the side streams' allocator pools fill up here, not in the timed loop)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        piped(6)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 6 * 1e3
        sec["host_buffers"]["images_s_prefetched"] = round(x.shape[0] / ms * 1e3, 2)
        sec["host_buffers"]["ms_per_step_prefetched"] = round(ms, 3)
        sec["host_buffers"]["prefetched"] = ("6 batches through parallel.prefetch: the copy of batch i + 1 overlaps the kernels of batch i "
                                             "(side stream); the copy-out stays on the compute stream")
        del xh, yh
    except Exception as exc:                                  # noqa: BLE001
        sec["host_buffers"] = {"error": repr(exc)[:300]}
    # ---- the reference's own throughput protocol (byte streams included) ---------------------------------------------------
    try:
        from mcquic_amd import validate
        enc, dec = validate.speed(model, iters=20)
        # the same batch through encode() / decode() alone (no byte streams): what the coder costs at THIS batch size
        x10 = torch.rand(10, 3, 768, 512).to(dev)
        c10 = model.encode(x10)
        t_enc = _timed(lambda: model.encode(x10), 20, warmup=1)
        t_dec = _timed(lambda: model.decode(c10), 20, warmup=1)
        del x10, c10
        sec["speed_protocol"] = {"encode_mpps": round(enc, 2), "decode_mpps": round(dec, 2),
                                 "tensor_only_batch10": {"encode_mpps": round(10 * 0.393216 / t_enc * 1e3, 2), "decode_mpps": round(10 * 0.393216 / t_dec * 1e3, 2),
                                                         "note": "encode() / decode() of the same 10-image batch, no entropy coder: the coder's cost is the gap to THESE "
                                                                 "(the 32-image headline's per-direction rates are another batch size)"},
                                 "coder_overlap": os.environ.get("MCQUIC_AMD_CODER_OVERLAP", "1") != "0",
                                 "images_s_encode_decode_768x512": round(1.0 / (0.393216 / enc + 0.393216 / dec), 2),
                                 "protocol": "mcquic/validate/validator.py:60-97 with 20 instead of 50 iterations: torch.rand(10, 3, 768, 512), one "
                                             "warm-up, `compress` x 20 then `decompress` x 20 between events, host rANS coding INCLUDED",
                                 "reference_published": {"encode_mpps": 25.45, "decode_mpps": 22.03, "hardware": "1x RTX 3090 (TF32), README.md:304"}}
    except Exception as exc:                                  # noqa: BLE001
        sec["speed_protocol"] = {"error": repr(exc)[:300]}
    # ---- opt-in Winograd F(2x2, 3x3) ------------------------------------------------------------------------------------
    try:
        ops.set_winograd(2)
        prof = ConvProfiler().install()
        prof.MAX_STEPS = 3

        def step():
            prof.next_step()
            return model.decode(model.encode(x))
        prof.active = False
        model.decode(model.encode(x))                         # re-pack + warm-up, not bracketed
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        prof.remove()
        c = prof.summary()
        w = {"arithmetic": "OPT-IN, not the reference's: F(2x2,3x3) on the 3x3 stride-1 layers with Cout % 128 == 0 and >= 20 k pixels, F(2,3) along x elsewhere above 128 k pixels",
             "images_s": round(x.shape[0] / ms * 1e3, 2), "ms_per_step": round(ms, 3),
             "mfma_work_frac": round(c["mfma_flops"] / (c["ms"] * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4),
             "mfma_gflop_per_step": round(c["mfma_flops"] / c["steps"] / 1e9, 1),
             "direct_form_gflop_per_step": round(c["flops"] / c["steps"] / 1e9, 1)}
        if cpu_codes is not None:
            codes = model.encode(x[:nb])
            pix = model.decode([t.to(dev) for t in cpu_codes])
            w["parity"] = parity_report(codes, pix, cpu_codes, cpu_pix)
        sec["winograd2d"] = w
    except Exception as exc:                                  # noqa: BLE001 -- a secondary figure must not take the headline down
        sec["winograd2d"] = {"error": repr(exc)[:300]}
    finally:
        ops.set_winograd(0)
    # ---- the reference's second listed model, No. 12 (README.md:306) --------------------------------------------------------
    try:
        torch.manual_seed(3412)
        m12 = Compressor(192, 12, [8192, 2048, 512]).eval().to(dev)
        nb12 = min(16, x.shape[0])
        x12 = x[:nb12].contiguous()
        prof = ConvProfiler().install()
        prof.MAX_STEPS = 2
        prof.active = False
        m12.decode(m12.encode(x12))                            # pack + warm-up, not bracketed
        prof.active = True
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            prof.next_step()
            m12.decode(m12.encode(x12))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 2 * 1e3
        prof.remove()
        c = prof.summary()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record(); c12 = m12.encode(x12); e[1].record(); m12.decode(c12); e[2].record()
        torch.cuda.synchronize()
        enc12, dec12 = validate.speed(m12, iters=10)
        sec["model12"] = {"model": "Compressor(192, 12, [8192, 2048, 512]) (the reference's model No. 12), random-init weights",
                          "batch": nb12, "images_s": round(nb12 / ms * 1e3, 2), "ms_per_step": round(ms, 3),
                          "encode_mpps": round(nb12 * H * W / 1e3 / e[0].elapsed_time(e[1]), 2),
                          "decode_mpps": round(nb12 * H * W / 1e3 / e[1].elapsed_time(e[2]), 2),
                          "frac_of_peak": round(c["mfma_flops"] / (c["ms"] * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4),
                          "conv_gflop_per_image": round(c["flops"] / c["steps"] / nb12 / 1e9, 1),
                          "whole_step_frac": round(c["flops"] / c["steps"] / (ms * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4),
                          "speed_protocol": {"encode_mpps": round(enc12, 2), "decode_mpps": round(dec12, 2),
                                             "protocol": "Validator.speed, batch 10, rANS byte streams included, 10 iterations"},
                          "reference_published": {"encode_mpps": 11.07, "decode_mpps": 10.21, "hardware": "1x RTX 3090 (TF32), README.md:306"}}
        del m12, x12, c12
    except Exception as exc:                                  # noqa: BLE001
        sec["model12"] = {"error": repr(exc)[:300]}
    # ---- Neon at the shapes the snapshot's trainer builds (mcquic/train/ddp.py:79-83, configs/neon.yaml: channel 32, k = 4096, five
    #      levels, denseNorm; 512 x 512 crops, 4 per GPU): stride-1 stem, AttentionBlocks at full resolution, widths 32 / 64 ----------
    try:
        sec["neon"] = neon_figures(dev)
    except Exception as exc:                                  # noqa: BLE001
        sec["neon"] = {"error": repr(exc)[:300]}
    # ---- batch-1 latency, hipGraph replay ---------------------------------------------------------------------------------
    try:
        model.enableGraphs(True)
        x1 = x[:1].contiguous()
        ms = _timed(lambda: model.decode(model.encode(x1)), 30, warmup=3)
        sec["batch1"] = {"ms_per_image_encode_decode": round(ms, 3), "graphs": True,
                         "frac_of_peak": round(536.63e9 / (ms * 1e-3) / (FP32_MATRIX_PEAK_TFLOPS * 1e12), 4)}
    except Exception as exc:                                  # noqa: BLE001
        sec["batch1"] = {"error": repr(exc)[:300]}
    finally:
        model.enableGraphs(False)
    # ---- VQ distance + argmin, BASELINE configs[3] --------------------------------------------------------------------
    try:
        from mcquic_amd.utils import synthetic as SY
        lat_c, cb_c = SY.vq_case("config4")                   # (seed 0: latents, then the codebook -- the tensors fixture F2b was captured on)
        lat = lat_c.to(dev)
        cb = ops.PackedCodebook(cb_c.to(dev))
        ms = _timed(lambda: ops.vq_assign(lat, cb), 10, warmup=2)
        flops = 2.0 * 4 * 32 * 48 * 32 * 4096 * 256
        sec["vq_config4"] = {"ms": round(ms, 4), "tflops": round(flops / (ms * 1e-3) / 1e12, 2),
                             "frac_of_peak": round(flops / (ms * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4),
                             "workload": "M=4 K=4096 D=256, 49152 vectors per codebook (412.3 GFLOP)"}
        # the codes of the timed launch against the REAL reference's (fixture F2b: per-image hashes captured from
        # mcquic/modules/quantizer.py:144-179 on these very tensors)
        f2b = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "f2b_vq_fullsize.npz")
        if os.path.exists(f2b):
            import numpy as _np
            want = _np.load(f2b)["config4_code_hash"]
            codes = ops.vq_assign(lat, cb).cpu()
            same = sum(int(SY.code_hash(codes[i]) == want[i].tobytes()) for i in range(codes.shape[0]))
            sec["vq_config4"]["images_bit_equal_to_reference"] = f"{same}/{codes.shape[0]}"
        del lat, cb
    except Exception as exc:                                  # noqa: BLE001
        sec["vq_config4"] = {"error": repr(exc)[:300]}
    # ---- training step, BASELINE configs[4] per-GPU work ------------------------------------------------------------------
    streams = blocks._BRANCH_STREAMS
    try:
        blocks._BRANCH_STREAMS = False                        # nested stream forks crash hipGraph capture (ROCm 7.2): one stream
        torch.manual_seed(3407)
        tm = Compressor(**MODEL).to(dev).train()
        xt = (torch.rand((8, 3, 256, 256), generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)

        from mcquic_amd.autograd import backward, mse_loss

        def train_step():
            for p in tm.parameters():
                p.grad = None
            xHat, _, _, _ = tm(xt)
            loss = mse_loss(xHat, xt)                          # (this library's reduction: no memset node inside the captured step)
            backward(loss)                                     # (loss.backward() with a cached root gradient: no fill launch)
            return loss
        for _ in range(2):
            train_step()
        torch.cuda.synchronize()
        for p in tm.parameters():
            p.grad = None
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            xHat, _, _, _ = tm(xt)
            static_loss = mse_loss(xHat, xt)
            backward(static_loss)
        ms = _timed(graph.replay, 10, warmup=1)
        flops = 3.0 * 536.63e9 * 8 * (256 * 256) / (768 * 512)      # forward + input gradients + weight gradients
        sec["train_step"] = {"ms": round(ms, 3), "graph": True, "images_per_step": 8, "crop": 256,
                             "frac_of_peak": round(flops / (ms * 1e-3) / (FP32_MATRIX_PEAK_TFLOPS * 1e12), 4),
                             "tflop_per_step": round(flops / 1e12, 3), "loss": round(float(static_loss.detach()), 6),
                             "workload": "forward + Gumbel straight-through backward, Compressor(128, 2, [8192, 2048, 512]), 8 x 3 x 256 x 256, one hipGraph"}
        del graph
        # ... and with the optimizer inside the captured step: the SGD update plus the re-pack of every convolution's two operand
        # streams that it makes necessary (the forward starts with it)
        try:
            opt = torch.optim.SGD(tm.parameters(), lr=1e-6)
            for _ in range(2):
                train_step()
                opt.step()
            torch.cuda.synchronize()
            for p in tm.parameters():
                p.grad = None
            graph2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph2):
                xHat, _, _, _ = tm(xt)
                backward(mse_loss(xHat, xt))
                opt.step()
            ms2 = _timed(graph2.replay, 10, warmup=1)
            sec["train_step"]["ms_with_sgd"] = round(ms2, 3)
            sec["train_step"]["frac_of_peak_with_sgd"] = round(flops / (ms2 * 1e-3) / (FP32_MATRIX_PEAK_TFLOPS * 1e12), 4)
            del graph2, opt
        except Exception as exc:                              # noqa: BLE001
            sec["train_step"]["ms_with_sgd"] = None
            sec["train_step"]["with_sgd_error"] = repr(exc)[:200]
        # ... and as the data-parallel step is meant to run (parallel.GraphedTrainStep: main graph, flat gradient buffer, [the
        # all-reduce over the ranks: none at N = 1], post graph with the SGD update and the frequency EMA; the in-graph re-pack
        # refreshes only the operand-stream copies its launches read)
        try:
            from mcquic_amd import parallel
            gstep = parallel.GraphedTrainStep(tm, torch.optim.SGD(tm.parameters(), lr=1e-6), xt)
            ms3 = _timed(lambda: gstep(xt), 10, warmup=2)
            gstep.close()
            sec["train_step"]["ms_graphed_data_parallel"] = round(ms3, 3)
            del gstep
        except Exception as exc:                              # noqa: BLE001
            sec["train_step"]["ms_graphed_data_parallel"] = None
            sec["train_step"]["graphed_error"] = repr(exc)[:200]
        # ... and the reference's own step (mcquic/train/trainer.py:270-283, configs/a800_8.yaml): Adam, gradient clipping by
        # global norm 4.0, the rate in a device tensor -- mcquic_amd.optim.Adam updates the whole model in one launch
        try:
            from mcquic_amd import optim, parallel
            gstep = parallel.GraphedTrainStep(tm, optim.Adam(tm.parameters(), lr=torch.tensor(1e-6, device=dev)), xt, max_grad_norm=4.0)
            ms4 = _timed(lambda: gstep(xt), 10, warmup=2)
            sec["train_step"]["ms_graphed_adam_clip"] = round(ms4, 3)
            sec["train_step"]["grad_norm"] = round(float(gstep.grad_norm), 6)
            gstep.close()
            del gstep
        except Exception as exc:                              # noqa: BLE001
            sec["train_step"]["ms_graphed_adam_clip"] = None
            sec["train_step"]["adam_error"] = repr(exc)[:200]
        del tm
    except Exception as exc:                                  # noqa: BLE001
        sec["train_step"] = {"error": repr(exc)[:300]}
    finally:
        blocks._BRANCH_STREAMS = streams
    return sec


def two_core_child(args):
    """`bench.py --two-core-child`: what ONE rank of an 8-GPU node gets from the host -- two cores (a 16-core box / 8 ranks) --
    for the three host-driven loops of the path: the eager B32 encode+decode step (~330 Python-side launches), the same step as
    hipGraph replays, the reference's speed protocol (host rANS coder in the loop) and the graphed data-parallel training step.
    The affinity is set before torch starts its thread pools (the equivalent of `taskset -c a,b python bench.py ...`)."""
    cores = sorted(os.sched_getaffinity(0))[:2]
    os.sched_setaffinity(0, set(cores))
    os.environ["OMP_NUM_THREADS"] = "2"
    torch.set_num_threads(2)
    from mcquic_amd import Compressor, parallel, validate
    from mcquic_amd.nn import blocks
    from mcquic_amd.utils import synthetic
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    out = {"cores": cores}
    model = synthetic.bench_model().to(dev)
    x = synthetic.bench_images(0, args.batch, H, W).to(dev)

    def wall(fn, steps, warmup=1):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    try:
        ms = wall(lambda: model.decode(model.encode(x)), 4, warmup=2)
        out["b32_eager"] = {"images_s": round(args.batch / ms * 1e3, 2), "ms_per_step": round(ms, 3)}
        model.enableGraphs(True)
        ms = wall(lambda: model.decode(model.encode(x)), 4, warmup=2)
        out["b32_graphs"] = {"images_s": round(args.batch / ms * 1e3, 2), "ms_per_step": round(ms, 3)}
        model.enableGraphs(False)
    except Exception as exc:                                  # noqa: BLE001
        out["b32_error"] = repr(exc)[:300]
    try:
        enc, dec = validate.speed(model, iters=10)
        out["speed_protocol"] = {"encode_mpps": round(enc, 2), "decode_mpps": round(dec, 2)}
    except Exception as exc:                                  # noqa: BLE001
        out["speed_protocol"] = {"error": repr(exc)[:300]}
    del model, x
    streams = blocks._BRANCH_STREAMS
    try:
        blocks._BRANCH_STREAMS = False
        torch.manual_seed(3407)
        tm = Compressor(**MODEL).to(dev).train()
        xt = (torch.rand((8, 3, 256, 256), generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
        for seg in (1, 3):
            gstep = parallel.GraphedTrainStep(tm, torch.optim.SGD(tm.parameters(), lr=1e-6), xt, segments=seg)
            out[f"train_graphed_ms_segments{seg}"] = round(wall(lambda: gstep(xt), 10, warmup=2), 3)
            gstep.close()
            del gstep
    except Exception as exc:                                  # noqa: BLE001
        out["train_error"] = repr(exc)[:300]
    finally:
        blocks._BRANCH_STREAMS = streams
    print("TWOCORE " + json.dumps(out), flush=True)
    return 0


def child_json(argv, tag, timeout):
    """Run `bench.py argv...` as a child process and return the JSON object on its `tag ` line (or an {"error": ...} record)."""
    import subprocess
    try:
        run = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv, capture_output=True, text=True, timeout=timeout)
        for line in reversed(run.stdout.splitlines()):
            if line.startswith(tag + " "):
                return json.loads(line[len(tag) + 1:])
        return {"error": f"child exited with code {run.returncode} and no result: {run.stderr[-300:]}"}
    except Exception as exc:                                  # noqa: BLE001
        return {"error": repr(exc)[:300]}


def secondary_child(args):
    """`bench.py --secondary-child FILE`: the secondary measurements in a process of their own -- a crash or hang in an opt-in
    mode or in the training-step capture must not be able to take the headline line down with it."""
    from mcquic_amd.utils import synthetic
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    model = synthetic.bench_model().to(dev)
    x = synthetic.bench_images(0, args.batch, H, W).to(dev)
    cpu_codes = cpu_pix = None
    if args.secondary_child and os.path.exists(args.secondary_child):
        blob = torch.load(args.secondary_child)
        cpu_codes, cpu_pix = blob["codes"], blob["pixels"]
    sec = secondary(model, x, dev, cpu_codes, cpu_pix, min(args.cpu_batch, args.batch))
    print("SECONDARY " + json.dumps(sec), flush=True)
    return 0


def secondary_in_child(args, cpu_codes, cpu_pix, timeout=420):
    import subprocess
    import tempfile
    handle = ""
    try:
        if cpu_codes is not None:
            fd, handle = tempfile.mkstemp(suffix=".pt", prefix="mcq_bench_")
            os.close(fd)
            torch.save({"codes": cpu_codes, "pixels": cpu_pix}, handle)
        cmd = [sys.executable, os.path.abspath(__file__), "--secondary-child", handle, "--batch", str(args.batch), "--cpu-batch", str(args.cpu_batch)]
        run = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
        for line in reversed(run.stdout.splitlines()):
            if line.startswith("SECONDARY "):
                return json.loads(line[len("SECONDARY "):])
        return {"error": f"child exited with code {run.returncode} and no result: {run.stderr[-300:]}"}
    except Exception as exc:                                  # noqa: BLE001 -- timeout, spawn failure: the headline stands on its own
        return {"error": repr(exc)[:300]}
    finally:
        if handle and os.path.exists(handle):
            os.remove(handle)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="images per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=4, help="images in the CPU-oracle sample (SURVEY 8(d): batch 4)")
    ap.add_argument("--graphs", action="store_true", help="replay encode/decode as captured hipGraphs (small-batch latency)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` measurements (opt-in Winograd mode, training step, VQ config #4, batch-1 latency; ~40 s)")
    ap.add_argument("--secondary-child", default=None, metavar="FILE",
                    help="internal: run ONLY the `secondary` measurements (FILE = optional torch file with the oracle's codes / pixels "
                         "for the parity leg) and print their JSON object; bench.py starts this as a child process")
    ap.add_argument("--two-core-child", action="store_true",
                    help="internal: the host-bound loops of the path with this process confined to two host cores (one rank's share "
                         "of an 8-GPU node); prints one TWOCORE line")
    ap.add_argument("--winograd", type=int, nargs="?", const=1, default=0, choices=(0, 1, 2),
                    help="OPT-IN experiment, never the headline: large 3x3 stride-1 layers in a Winograd form (not the reference's arithmetic); "
                         "1 = F(2, 3) along x, 2 = F(2x2, 3x3) where the layer allows it")
    args = ap.parse_args()
    if args.two_core_child:
        return two_core_child(args)
    if args.secondary_child is not None:
        return secondary_child(args)

    from mcquic_amd import launch
    # `--gpus N` IS the world size: without a launcher's environment the script re-executes itself under
    # torch.distributed.run with N ranks (one per GPU); a launcher whose WORLD_SIZE disagrees is an error (exit 2)
    rank, local_rank, world, use_dist = launch.ensure_world(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product path has no CPU fallback)")
    if world > torch.cuda.device_count():
        raise SystemExit(f"bench.py: --gpus {world} but this node has {torch.cuda.device_count()} HIP devices")
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    torch.cuda.set_device(local_rank)                        # (first: the runtime's primary context belongs on this rank's GPU)
    dev = torch.device("cuda", local_rank)
    cores = launch.pin_rank_cores(local_rank, local_world)  # disjoint host cores per rank, from the GPU's NUMA node
    rccl_world = None
    if use_dist:                                             # launched by torch.distributed.run: the RCCL path even at N=1
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)      # "nccl" is RCCL on ROCm
        rccl_world = launch.rccl_check(dist, dev)           # an actual all-reduce: the ranks RCCL really connected
        if rccl_world != world or dist.get_world_size() != world:
            raise SystemExit(f"bench.py: RCCL spans {rccl_world} ranks, expected {world}")

    from mcquic_amd import Compressor, ops
    if args.winograd:
        ops.set_winograd(args.winograd)
    from mcquic_amd.utils import synthetic
    model = synthetic.bench_model().to(dev)                   # torch.manual_seed(3407): the same random-init weights on every rank
    if args.graphs:
        model.enableGraphs(True)
    x = synthetic.bench_images(rank, args.batch, H, W).to(dev)         # seed 3407 + rank: configs[2] = ranks 0..7 of this

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        codes = model.encode(x)
        return codes, model.decode(codes)

    for _ in range(args.warmup):
        step()
    prof = ConvProfiler().install()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        prof.next_step()
        step()
    torch.cuda.synchronize()
    own = time.perf_counter() - t0                            # this rank's K steps, before it waits for the others
    barrier()
    dt = time.perf_counter() - t0
    prof.remove()
    rank_ms = None
    if use_dist:
        t = torch.tensor([dt, own], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        dt = max(float(v[0].item()) for v in every)           # the contract's figure: barrier to barrier, MAX over ranks
        rank_ms = [round(float(v[1].item()) / args.steps * 1e3, 3) for v in every]

    # secondary: each direction on its own (Mpps as in the reference's README), untimed region
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record(); codes = model.encode(x); e[1].record(); pix = model.decode(codes); e[2].record()
    torch.cuda.synchronize()
    enc_ms, dec_ms = e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])

    if rank == 0:
        conv = prof.summary()
        images = world * args.batch * args.steps
        value = images / dt
        # the profiler sees Python-side launches only: under --graphs the step is one graph replay and there is nothing to
        # bracket -- the per-kernel fields are then null (never 0.0); whole_step_frac still holds
        seen = conv["launches"] > 0 and conv["ms"] > 0
        # roofline.achieved counts the MFMA work actually ISSUED (= the algorithmic FLOPs in the default direct form; in the
        # opt-in Winograd modes 16/36 resp. 12/18 of them on the layers that run transformed), so frac is a true pipe
        # utilisation and cannot exceed 1; the direct-form-equivalent rate is reported beside it
        achieved_tf = conv["mfma_flops"] / (conv["ms"] * 1e-3) / 1e12 if seen else None
        equivalent_tf = conv["flops"] / (conv["ms"] * 1e-3) / 1e12 if seen else None
        pmc = pmc_traffic()
        step_s = dt / args.steps
        out = {
            "metric": "images/sec encode+decode, 768x512 Kodak-shape batch, qp=2",
            "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_s * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (uniform [-1,1) images, random-init weights)",
            "arithmetic": (("OPT-IN winograd F(2x2,3x3) (4/9 of the multiplications; F(2,3) along x where a layer does not qualify)" if args.winograd == 2 else
                            "OPT-IN winograd F(2,3) along x (2/3 of the multiplications)") +
                           " for the large 3x3 stride-1 layers, float32; NOT the reference's arithmetic -- roofline.achieved counts the MFMA work "
                           "issued, roofline.equivalent_direct the direct form's FLOPs")
                          if args.winograd else "direct form, exact fp32 MFMA (the reference's arithmetic)",
            "rccl_world": rccl_world, "rank0_cores": None if cores is None else len(cores),
            # each rank's own ms per step BEFORE it waits at the closing barrier (ms_per_step is barrier to barrier, MAX over
            # ranks): tells a slow GPU / a starved host apart from a scaling loss; null without a launcher
            "rank_ms_per_step": rank_ms,
            "config": {"workload": f"qp=2 reference model Compressor(128, 2, [8192, 2048, 512]), batch={args.batch} "
                                   f"768x512 random images per GPU, encode+decode tensor path (BASELINE configs[1])",
                       "images_per_gpu": args.batch, "parallelism": f"dp{world} (independent image shards, no data-path collective)"},
            "encode_ms": round(enc_ms, 3), "decode_ms": round(dec_ms, 3),
            "encode_mpps": round(args.batch * H * W / 1e3 / enc_ms, 3), "decode_mpps": round(args.batch * H * W / 1e3 / dec_ms, 3),
            "roofline": {
                "bound": "mfma", "kernel": "conv_mfma_kernel (fp32 v_mfma_f32_32x32x2_f32 implicit GEMM, all tile variants; + the 16-row conv_head16_kernel of the image head)",
                "achieved": None if achieved_tf is None else round(achieved_tf, 2), "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": None if achieved_tf is None else round(achieved_tf / FP32_MATRIX_PEAK_TFLOPS, 4),
                "equivalent_direct": None if equivalent_tf is None else round(equivalent_tf, 2),
                "source": "HIP events around every conv launch of the first event_steps timed steps" if seen else
                          "none: the step is a hipGraph replay (--graphs), no Python-side launch to bracket; see whole_step_frac",
                "traffic": None if pmc is None else round(pmc["all_conv_launches"]["fetch_bytes_per_launch_x2_corrected"] + pmc["all_conv_launches"]["write_size_bytes_per_launch"]),
                "traffic_stale": None if pmc is None else bool(pmc["_stale"]),
                "traffic_unit": "HBM-side bytes per conv kernel launch, averaged over all conv launches of a step like algorithmic_bytes_per_launch (PMC FETCH_SIZE x2-corrected + WRITE_SIZE, separate passes, profiles/" + ((pmc or {}).get("_file") or "rNN_pmc.json") + "; traffic_stale = those passes ran on other kernel sources than this run)",
                "algorithmic_bytes_per_launch": round(conv["bytes"] / conv["launches"]) if seen else None,
                "launches_per_step": conv["launches"] // conv["steps"] if seen else None,
                "event_steps": conv["steps"] if seen else 0,
                "avg_launch_ms": round(conv["sum_ms"] / conv["launches"], 4) if seen else None,
                "conv_busy_ms_per_step": round(conv["ms"] / conv["steps"], 3) if seen else None,
                "algorithmic_gflop_per_step": round(conv["flops"] / conv["steps"] / 1e9, 2) if seen else round(536.63 * args.batch, 2),
                "mfma_gflop_per_step": round(conv["mfma_flops"] / conv["steps"] / 1e9, 2) if seen else None,
                "hbm_algorithmic_gbs": round(conv["bytes"] / (conv["ms"] * 1e-3) / 1e9, 1) if seen else None,
                "whole_step_frac": round((conv["mfma_flops"] / conv["steps"] + 3.435e9 * args.batch if seen and args.winograd else 536.63e9 * args.batch)
                                         / step_s / (FP32_MATRIX_PEAK_TFLOPS * 1e12), 4),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            nb = min(args.cpu_batch, args.batch)
            sd_cpu = {k: v.detach().cpu() for k, v in model.state_dict().items()}
            base, cpu_codes, cpu_pix = cpu_baseline(sd_cpu, x[:nb].cpu())
            gpu_pix = model.decode([c.to(dev) for c in cpu_codes])
            out["cpu_baseline"] = base
            out["parity"] = parity_report([c[:nb] for c in codes], gpu_pix, cpu_codes, cpu_pix)
        else:
            sd_cpu = cpu_codes = cpu_pix = None
        # (not under a launcher: the secondary's hipGraph captures do not mix with RCCL's watchdog thread -- a capture in global
        #  mode is invalidated by its event queries; the driver's N = 1 run is the plain command)
        if world == 1 and not use_dist and not args.no_secondary and not args.winograd and not args.graphs:
            del model, codes, pix                 # (the child builds its own model on the same GPU)
            torch.cuda.empty_cache()
            out["secondary"] = secondary_in_child(args, cpu_codes, cpu_pix)
            # one rank's share of the host on an 8-GPU node (16 cores / 8 ranks): the host-driven loops again, confined to two cores
            two = child_json(["--two-core-child", "--batch", str(args.batch)], "TWOCORE", 420)
            sec = out["secondary"] if isinstance(out["secondary"], dict) else {}
            try:
                # (the graph-replay loop has no all-core measurement in this run to compare against: only its absolute figure is
                #  reported, two["b32_graphs"]; a percentage against the EAGER headline would mix two effects -- ADVICE r4)
                ref = {"b32_eager": value,
                       "speed_encode": sec.get("speed_protocol", {}).get("encode_mpps"), "speed_decode": sec.get("speed_protocol", {}).get("decode_mpps"),
                       "train": sec.get("train_step", {}).get("ms_graphed_data_parallel")}
                drop = {}
                if "b32_eager" in two:
                    drop["b32_eager_pct"] = round(100 * (1 - two["b32_eager"]["images_s"] / ref["b32_eager"]), 2)
                if ref["speed_encode"] and "encode_mpps" in two.get("speed_protocol", {}):
                    drop["speed_encode_pct"] = round(100 * (1 - two["speed_protocol"]["encode_mpps"] / ref["speed_encode"]), 2)
                    drop["speed_decode_pct"] = round(100 * (1 - two["speed_protocol"]["decode_mpps"] / ref["speed_decode"]), 2)
                if ref["train"] and two.get("train_graphed_ms_segments1"):
                    drop["train_graphed_pct"] = round(100 * (two["train_graphed_ms_segments1"] / ref["train"] - 1), 2)
                two["loss_vs_all_cores"] = drop
                two["note"] = ("the same loops with the process confined to two host cores (`taskset -c a,b`): what a rank of an 8-GPU node "
                               "has; loss_vs_all_cores in percent against this run's headline / secondary figures (negative = faster)")
            except Exception as exc:                          # noqa: BLE001
                two["compare_error"] = repr(exc)[:200]
            if isinstance(out["secondary"], dict):
                out["secondary"]["two_core_host"] = two
    # RCCL writes its version banner through C stdio, which a pipe only sees at exit: every rank flushes it out before the
    # last barrier so that rank 0's ONE line below is the last thing on stdout
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    sys.exit(main() or 0)
