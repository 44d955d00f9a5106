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
        self.records = []           # (start, end, flops, bytes)
        self.base = None
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

        def wrapped(x, w, stride=1, **kw):
            if not prof.active:
                return prof._orig(x, w, stride, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = prof._orig(x, w, stride, **kw)
            e.record()
            n, cin, h, wd = x.shape
            pad = w.ksize // 2
            ho = (h + 2 * pad - w.ksize) // stride + 1
            wo = (wd + 2 * pad - w.ksize) // stride + 1
            flops = 2.0 * n * ho * wo * w.cout * cin * w.ksize * w.ksize
            # algorithmic HBM bytes of the fused op: input and output once, plus every output-shaped side tensor its
            # contract needs (residual / multiplier / gate reads, the SiLU twin write)
            sides = sum(1 for key in ("res", "gdn_mul", "igdn_mul", "gate_mul", "gate_id", "mul") if kw.get(key) is not None)
            sides += 1 if kw.get("dual_silu") else 0
            nbytes = 4.0 * (x.numel() + y.numel() * (1 + sides))
            prof.records.append((s, e, flops, nbytes))
            return y

        self._orig_multi = ops.conv2d_multi

        def wrapped_multi(xs, ws, stride=1, per_problem=None, **shared):
            if not prof.active:
                return prof._orig_multi(xs, ws, stride, per_problem=per_problem, **shared)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ops.conv2d = prof._orig                     # (the multi launch is ONE launch: no per-problem brackets inside)
            s.record()
            ys = prof._orig_multi(xs, ws, stride, per_problem=per_problem, **shared)
            e.record()
            ops.conv2d = wrapped
            flops = nbytes = 0.0
            for i, (x, w, y) in enumerate(zip(xs, ws, ys)):
                n, cin, h, wd = x.shape
                flops += 2.0 * y.shape[0] * y.shape[2] * y.shape[3] * w.cout * cin * w.ksize * w.ksize
                kw = dict(shared, **((per_problem or [{}] * len(xs))[i]))
                sides = sum(1 for key in ("res", "gdn_mul", "igdn_mul", "gate_mul", "gate_id", "mul") if kw.get(key) is not None)
                sides += 1 if kw.get("dual_silu") else 0
                nbytes += 4.0 * (x.numel() + y.numel() * (1 + sides))
            prof.records.append((s, e, flops, nbytes))
            return ys

        ops.conv2d = wrapped
        ops.conv2d_multi = wrapped_multi
        return self

    def remove(self):
        from mcquic_amd import ops
        ops.conv2d = self._orig
        ops.conv2d_multi = self._orig_multi

    def summary(self):
        iv = sorted((self.base.elapsed_time(s), self.base.elapsed_time(e)) for s, e, _, _ in self.records)
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
                    sum_ms=sum(s.elapsed_time(e) for s, e, _, _ in self.records))


def pmc_traffic():
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summary (separate
    FETCH_SIZE / WRITE_SIZE passes of this same command; profiles/rNN_pmc.json).  None if absent."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc.json")), reverse=True):     # the latest round's passes
        try:
            d = json.load(open(path))
            d["_file"] = os.path.basename(path)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="images per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=4, help="images in the CPU-oracle sample (SURVEY 8(d): batch 4)")
    ap.add_argument("--graphs", action="store_true", help="replay encode/decode as captured hipGraphs (small-batch latency)")
    ap.add_argument("--winograd", type=int, nargs="?", const=1, default=0, choices=(0, 1, 2),
                    help="OPT-IN experiment, never the headline: large 3x3 stride-1 layers in a Winograd form (not the reference's arithmetic); "
                         "1 = F(2, 3) along x, 2 = F(2x2, 3x3) where the layer allows it")
    args = ap.parse_args()

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
    torch.manual_seed(3407)                                   # same random-init weights on every rank
    model = Compressor(**MODEL).eval().to(dev)
    if args.graphs:
        model.enableGraphs(True)
    g = torch.Generator(device="cpu").manual_seed(3407 + rank)
    x = (torch.rand((args.batch, 3, H, W), generator=g) * 2 - 1).to(dev)

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
    barrier()
    dt = time.perf_counter() - t0
    prof.remove()
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # secondary: each direction on its own (Mpps as in the reference's README), untimed region
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record(); codes = model.encode(x); e[1].record(); pix = model.decode(codes); e[2].record()
    torch.cuda.synchronize()
    enc_ms, dec_ms = e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])

    if rank == 0:
        conv = prof.summary()
        images = world * args.batch * args.steps
        value = images / dt
        achieved_tf = conv["flops"] / (conv["ms"] * 1e-3) / 1e12 if conv["ms"] > 0 else 0.0
        out = {
            "metric": "images/sec encode+decode, 768x512 Kodak-shape batch, qp=2",
            "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (uniform [-1,1) images, random-init weights)",
            "arithmetic": (("OPT-IN winograd F(2x2,3x3) (4/9 of the multiplications; F(2,3) along x where a layer does not qualify)" if args.winograd == 2 else
                            "OPT-IN winograd F(2,3) along x (2/3 of the multiplications)") +
                           " for the 3x3 stride-1 layers of >= 128 k pixels, float32; NOT the reference's arithmetic -- roofline.achieved counts "
                           "the direct form's FLOPs and is an equivalent rate here")
                          if args.winograd else "direct form, exact fp32 MFMA (the reference's arithmetic)",
            "rccl_world": rccl_world, "rank0_cores": None if cores is None else len(cores),
            "config": {"workload": f"qp=2 reference model Compressor(128, 2, [8192, 2048, 512]), batch={args.batch} "
                                   f"768x512 random images per GPU, encode+decode tensor path (BASELINE configs[1])",
                       "images_per_gpu": args.batch, "parallelism": f"dp{world} (independent image shards, no data-path collective)"},
            "encode_ms": round(enc_ms, 3), "decode_ms": round(dec_ms, 3),
            "encode_mpps": round(args.batch * H * W / 1e3 / enc_ms, 3), "decode_mpps": round(args.batch * H * W / 1e3 / dec_ms, 3),
            "roofline": {
                "bound": "mfma", "kernel": "conv_mfma_kernel (fp32 v_mfma_f32_32x32x2_f32 implicit GEMM, all tile variants; + the 16-row conv_head16_kernel of the image head)",
                "achieved": round(achieved_tf, 2), "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved_tf / FP32_MATRIX_PEAK_TFLOPS, 4),
                "traffic": (lambda t: None if t is None else round(t["all_conv_launches"]["fetch_bytes_per_launch_x2_corrected"] + t["all_conv_launches"]["write_size_bytes_per_launch"]))(pmc_traffic()),
                "traffic_unit": "HBM-side bytes per conv kernel launch, averaged over all conv launches of a step like algorithmic_bytes_per_launch (PMC FETCH_SIZE x2-corrected + WRITE_SIZE, separate passes, profiles/" + ((pmc_traffic() or {}).get("_file") or "rNN_pmc.json") + ")",
                "algorithmic_bytes_per_launch": round(conv["bytes"] / max(conv["launches"], 1)),
                "launches_per_step": conv["launches"] // conv["steps"],
                "event_steps": conv["steps"],
                "avg_launch_ms": round(conv["sum_ms"] / max(conv["launches"], 1), 4),
                "conv_busy_ms_per_step": round(conv["ms"] / conv["steps"], 3),
                "algorithmic_gflop_per_step": round(conv["flops"] / conv["steps"] / 1e9, 2),
                "hbm_algorithmic_gbs": round(conv["bytes"] / (conv["ms"] * 1e-3) / 1e9, 1) if conv["ms"] > 0 else None,
                "whole_step_frac": round(536.63e9 * args.batch / (dt / args.steps) / (FP32_MATRIX_PEAK_TFLOPS * 1e12), 4),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            nb = min(args.cpu_batch, args.batch)
            sd_cpu = {k: v.detach().cpu() for k, v in model.state_dict().items()}
            base, cpu_codes, cpu_pix = cpu_baseline(sd_cpu, x[:nb].cpu())
            gpu_pix = model.decode([c.to(dev) for c in cpu_codes])
            out["cpu_baseline"] = base
            out["parity"] = parity_report([c[:nb] for c in codes], gpu_pix, cpu_codes, cpu_pix)
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
    main()
