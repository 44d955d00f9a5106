#!/usr/bin/env python3
"""Capture golden vectors from the REAL reference (xiaosu-zhu/McQuic, imported unmodified from /root/reference
through oracle/ref_harness.py).  Runs only in the build container; the outputs (small .npz files of inputs /
expected outputs -- data, no reference source) are committed under tests/golden/.

    python tests/golden/make_golden.py

Inputs and weights come from the repo's own seeded generators (oracle.mcquic_ref.make_state_dict / make_images /
plain torch.Generator), so fixtures store seeds + expected outputs and stay small.
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mcquic_ref as R          # noqa: E402  (generators only; expected values come from the reference)
from oracle import ref_harness              # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()


F7_CASES = [(1, 2, 200, 176), (2, 3, 193, 211), (3, 1, 768, 512), (4, 2, 161, 400)]   # (seed, n, h, w)


def want(tag: str) -> bool:
    """`python make_golden.py f9 f10` regenerates only the named fixture sets (default: all of them)."""
    sel = [a.lower() for a in sys.argv[1:]]
    return not sel or tag in sel


F5C_NEAR = 2e-5       # top-2 gaps below this are recorded as near-ties


from mcquic_amd.utils.synthetic import bench_images, bench_state_dict      # noqa: E402  (bench.py's own workload generators)
from mcquic_amd.utils.synthetic import code_hash as _code_hash              # noqa: E402


def code_hash(code_img: torch.Tensor) -> np.ndarray:
    return np.frombuffer(_code_hash(code_img), dtype=np.uint8)


def capture_f5c(RQ, shards: int = 8, per_shard: int = 32, chunk: int = 8):
    import copy
    import time
    sd = bench_state_dict()
    model = ref_harness.reference_compressor(128, 2, [8192, 2048, 512], sd)
    model64 = copy.deepcopy(model).double()
    orig_distance = RQ._multiCodebookQuantization._distance
    state = {}

    def rec32(self, x):
        dist = orig_distance(self, x)
        top2 = torch.topk(dist, 2, dim=-1, largest=False)
        state["top2"].append((top2.values.clone(), top2.indices.clone()))
        state["dist32"].append(dist)
        return dist

    def rec64(self, x):
        dist = orig_distance(self, x)
        lv = len(state["d64"])
        c32 = state["codes32"][lv]
        state["d64"].append((dist.gather(-1, c32.unsqueeze(-1)).squeeze(-1) - dist.min(-1).values).clone())
        return dist

    hashes = np.zeros((shards * per_shard, 3, 8), dtype=np.uint8)
    near, near_gap, flips, flip_g32, flip_g64 = [], [], [], [], []
    downstream = 0
    recs, xsha = [], []
    thread_diffs = None
    totals = [0, 0, 0]
    t0 = time.time()
    for r in range(shards):
        x = bench_images(r, per_shard)
        xsha.append(np.frombuffer(bytes.fromhex(sha(x)), dtype=np.uint8))
        shard_codes = [[], [], []]
        for lo in range(0, per_shard, chunk):
            xs = x[lo:lo + chunk]
            state.update(top2=[], dist32=[], d64=[])
            RQ._multiCodebookQuantization._distance = rec32
            try:
                with torch.inference_mode():
                    c32 = model.encode(xs)
            finally:
                RQ._multiCodebookQuantization._distance = orig_distance
            state["codes32"] = c32
            RQ._multiCodebookQuantization._distance = rec64
            try:
                with torch.inference_mode():
                    c64 = model64.encode(xs.double())
            finally:
                RQ._multiCodebookQuantization._distance = orig_distance
            alive = torch.ones(len(xs), dtype=torch.bool)
            for lv in range(3):
                vals, idx = state["top2"][lv]
                gap = vals[..., 1] - vals[..., 0]
                second = torch.where(idx[..., 0] == c32[lv], idx[..., 1], idx[..., 0])      # (exact ties: topk's order is not argmin's)
                totals[lv] += c32[lv].numel()
                for i in range(len(xs)):
                    hashes[r * per_shard + lo + i, lv] = code_hash(c32[lv][i])
                    shard_codes[lv].append(c32[lv][i])
                for (i, g, yy, xx) in (gap < F5C_NEAR).nonzero().tolist():
                    near.append((r, lo + i, lv, g, yy, xx, int(c32[lv][i, g, yy, xx]), int(second[i, g, yy, xx])))
                    near_gap.append(float(gap[i, g, yy, xx]))
                diff = c32[lv] != c64[lv]
                downstream += int((diff & ~alive[:, None, None, None]).sum())
                first = diff & alive[:, None, None, None]
                for (i, g, yy, xx) in first.nonzero().tolist():
                    a, b = int(c32[lv][i, g, yy, xx]), int(c64[lv][i, g, yy, xx])
                    d32 = state["dist32"][lv][i, g, yy, xx]
                    flips.append((r, lo + i, lv, g, yy, xx, a, b))
                    flip_g32.append(float(d32[b] - d32[a]))
                    flip_g64.append(float(state["d64"][lv][i, g, yy, xx]))
                alive &= ~first.flatten(1).any(1)
            if lo == 0:
                with torch.inference_mode():
                    recs.append(model.decode([c[:1] for c in c32])[..., ::16, ::16].numpy())
            print(f"f5c: shard {r} images {lo}..{lo + len(xs) - 1}  near-ties {len(near)}  self-flips {len(flips)}  "
                  f"{time.time() - t0:.0f} s", flush=True)
        if r == 0:                               # the reference against itself with 1 thread instead of 8
            torch.set_num_threads(1)
            with torch.inference_mode():
                c1 = model.encode(x[:8])
            torch.set_num_threads(8)
            thread_diffs = sum(int((a != torch.stack(shard_codes[lv][:8])).sum()) for lv, a in enumerate(c1))
    sdsha = hashlib.sha256(b"".join(sd[k].contiguous().numpy().tobytes() for k in sorted(sd))).hexdigest()
    out = {"shape": np.array([shards, per_shard, 768, 512]), "near_threshold": np.array([F5C_NEAR]),
           "codes_per_level": np.array(totals), "code_hash": hashes,
           "near": np.array(near, dtype=np.int32).reshape(-1, 8), "near_gap": np.array(near_gap, dtype=np.float32),
           "selfflip": np.array(flips, dtype=np.int32).reshape(-1, 8), "selfflip_gap32": np.array(flip_g32, dtype=np.float32),
           "selfflip_gap64": np.array(flip_g64, dtype=np.float64), "selfflip_downstream": np.array([downstream]),
           "threads_1_vs_8_code_diffs_first8": np.array([thread_diffs]),
           "rec_strided": np.concatenate(recs, 0), "x_sha": np.stack(xsha),
           "state_dict_sha": np.frombuffer(bytes.fromhex(sdsha), dtype=np.uint8)}
    np.savez_compressed(os.path.join(OUT, "f5c_config2_census.npz"), **out)
    print(f"f5c: {len(near)} near-ties, {len(flips)} reference self-flips (float32 vs float64), gaps32 {flip_g32}")


def capture_f5c_backend(RQ, chunk: int = 8):
    """Second sensitivity probe, merged into f5c_config2_census.npz: the reference in float32 with oneDNN switched off
    (`torch.backends.mkldnn.flags(enabled=False)`: ATen's native convolution = another float32 summation order, the analogue
    of what a different kernel does) against its own default run -- images whose code hashes differ are re-run with
    recording to find the first flips and the default run's gaps there."""
    import time
    path = os.path.join(OUT, "f5c_config2_census.npz")
    d = dict(np.load(path))
    shards, per = int(d["shape"][0]), int(d["shape"][1])
    sd = bench_state_dict()
    model = ref_harness.reference_compressor(128, 2, [8192, 2048, 512], sd)
    orig_distance = RQ._multiCodebookQuantization._distance
    flips, gaps, downstream, t0 = [], [], 0, time.time()
    for r in range(shards):
        x = bench_images(r, per)
        for lo in range(0, per, chunk):
            xs = x[lo:lo + chunk]
            with torch.backends.mkldnn.flags(enabled=False), torch.inference_mode():
                alt = model.encode(xs)
            differs = [i for i in range(len(xs))
                       if any(code_hash(alt[lv][i]).tobytes() != d["code_hash"][r * per + lo + i, lv].tobytes() for lv in range(3))]
            for i in differs:
                dists = []

                def rec(self, xx):
                    dist = orig_distance(self, xx)
                    dists.append(dist)
                    return dist
                RQ._multiCodebookQuantization._distance = rec
                try:
                    with torch.inference_mode():
                        c32 = model.encode(xs[i:i + 1])
                finally:
                    RQ._multiCodebookQuantization._distance = orig_distance
                alive = True
                for lv in range(3):
                    bad = (c32[lv][0] != alt[lv][i]).nonzero().tolist()
                    if not alive:
                        downstream += len(bad)
                        continue
                    for g, yy, xx in bad:
                        a, b = int(c32[lv][0, g, yy, xx]), int(alt[lv][i, g, yy, xx])
                        flips.append((r, lo + i, lv, g, yy, xx, a, b))
                        gaps.append(float(dists[lv][0, g, yy, xx, b] - dists[lv][0, g, yy, xx, a]))
                    alive = alive and not bad
            print(f"f5c backend: shard {r} images {lo}..{lo + len(xs) - 1}  self-flips {len(flips)}  {time.time() - t0:.0f} s", flush=True)
    d["selfflip_backend"] = np.array(flips, dtype=np.int32).reshape(-1, 8)
    d["selfflip_backend_gap32"] = np.array(gaps, dtype=np.float32)
    d["selfflip_backend_downstream"] = np.array([downstream])
    np.savez_compressed(path, **d)
    print(f"f5c backend: {len(flips)} reference self-flips (oneDNN vs native convolution), gaps32 {gaps}")


NEON_K4096 = (32, 4096, [16, 8, 4, 2, 2])      # F13 / F14: the shapes SURVEY 8(f) row 4 names (k = 4096, d = 8, m = 1; configs/neon.yaml: five levels)


def capture_neon(C, RQ, dense: bool, fname: str, cfg=(32, 256, [8, 4, 2, 2]), hw: int = 128, stride: int = 4, logit_stride: int = 8):
    from oracle import neon_ref as NR            # generators only
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("gloo", rank=0, world_size=1)
    ch, k, size = cfg
    assert hw // 16 == size[0], "the level maps are `size` itself when the image side is 16 x size[0]"
    sd = NR.make_state_dict(ch, k, size, seed=3, denseNorm=dense)
    model = C.Neon(ch, k, size, dense).eval()
    model.load_state_dict(sd, strict=True)
    xi = R.make_images(2, hw, hw, seed=5)
    f10 = {"config": np.array([ch, k] + size), "n_state_dict_entries": np.array([len(model.state_dict())])}
    gaps = []
    orig_distance = RQ._multiCodebookQuantization._distance

    def recording_distance(self, x):
        dist_ = orig_distance(self, x)
        top2 = torch.topk(dist_, 2, dim=-1, largest=False).values
        gaps.append((top2[..., 1] - top2[..., 0]).clone())
        return dist_
    RQ._multiCodebookQuantization._distance = recording_distance
    try:
        with torch.inference_mode():
            codes = model.encode(xi)
    finally:
        RQ._multiCodebookQuantization._distance = orig_distance
    with torch.inference_mode():
        rec = model.decode(codes)
        rb = model.residual_backward(codes[1], 2)
        rf0 = model.residual_forward(codes[0], None, 0)
        rf1 = model.residual_forward(codes[1], rf0, 1)
    for lv, cd in enumerate(codes):
        f10[f"code{lv}"] = cd.numpy().astype(np.int16)
        f10[f"gap{lv}"] = gaps[lv].numpy()
    f10["rec_strided"] = rec[..., ::stride, ::stride].numpy()
    f10["rec_sha"] = np.frombuffer(bytes.fromhex(sha(rec)), dtype=np.uint8)
    f10["residual_backward_1_2"] = rb.numpy()
    f10["residual_forward_1"] = rf1.numpy()
    model.train()
    g = torch.Generator().manual_seed(9)
    shapes = [(2, 1, sz, sz, k) for sz in reversed(size)]         # (quantizations run from the smallest level up)
    us = [(torch.rand(sh, generator=g), torch.rand(sh, generator=g)) for sh in shapes]
    it = iter([u for pair in us for u in pair])
    orig = torch.rand_like
    torch.rand_like = lambda t, **kw: next(it).clone()
    try:
        xHat, yHat, codesT, logitsT = model(xi.clone())
    finally:
        torch.rand_like = orig
    f10["train_xHat_strided"] = xHat.detach()[..., ::stride, ::stride].numpy()
    f10["train_yHat"] = yHat.detach().numpy()
    for lv in range(len(size)):
        f10[f"train_code{lv}"] = codesT[lv].numpy().astype(np.int16)
        f10[f"train_logit{lv}_strided"] = logitsT[lv].detach()[..., ::logit_stride].numpy()
        f10[f"train_ema{lv}"] = model._quantizer._entropyCoder._freqEMA[lv].detach().numpy()
    np.savez_compressed(os.path.join(OUT, fname), **f10)


def main():
    C = ref_harness.load()
    import mcquic.nn as RN                      # the reference's layers
    import mcquic.modules.quantizer as RQ
    from mcquic.data.transforms import AlignedPadding
    torch.manual_seed(0)
    torch.set_num_threads(8)

    # ---- F1: per-block I/O at C = 8 (odd sizes) -------------------------------------------------------
    if want("f1"):
        blocks = {}
        c = 8
        x = rand((2, c, 13, 18), 101)
        blocks["x"] = x.numpy()
        for name, ctor, mk in [("ResidualBlock", lambda: RN.ResidualBlock(c, c), R._rb),
                               ("ResidualBlockWithStride", lambda: RN.ResidualBlockWithStride(c, c), R._rb_stride),
                               ("ResidualBlockShuffle", lambda: RN.ResidualBlockShuffle(c, c), R._rb_shuffle),
                               ("AttentionBlock", lambda: RN.blocks.AttentionBlock(c), R._attn)]:
            sd = {}
            mk(sd, "", c, 11)
            mod = ctor().eval()
            mod.load_state_dict(sd, strict=True)
            with torch.inference_mode():
                blocks[name] = mod(x.clone()).numpy()
        for name, cls in [("GenDivNorm", RN.GenDivNorm), ("InvGenDivNorm", RN.InvGenDivNorm)]:
            sd = {}
            R._gdn_params(sd, "", c, 12)
            mod = cls(c).eval()
            mod.load_state_dict(sd, strict=True)
            with torch.inference_mode():
                blocks[name] = mod(x.clone() * 2).numpy()
        np.savez_compressed(os.path.join(OUT, "f1_blocks_c8.npz"), **blocks)

    # ---- F2 / F3: VQ distance + argmin, gather (the reference's _multiCodebookQuantization) -------------
    if want("f2"):
        vq = {}
        for tag, (m, k, d, n, h, w) in {"qp2_l2": (2, 512, 64, 2, 6, 8), "qp2_l1": (2, 2048, 64, 1, 6, 8),
                                        "small": (2, 32, 4, 2, 8, 8)}.items():
            g = torch.Generator().manual_seed(7)
            cb = torch.randn((m, k, d), generator=g) * np.sqrt(2 / (5 * d))
            xx = torch.randn((n, m * d, h, w), generator=g) * 0.1
            q = RQ._multiCodebookQuantization(torch.nn.Parameter(cb), 0.0)
            dq = RQ._multiCodebookDeQuantization(torch.nn.Parameter(cb))
            with torch.inference_mode():
                dist = q._distance(xx)
                code = q.encode(xx)
                deq = dq.decode(code)
            top2 = torch.topk(dist, 2, dim=-1, largest=False).values
            vq[tag + "_shape"] = np.array([m, k, d, n, h, w])
            vq[tag + "_code"] = code.numpy().astype(np.int16)
            vq[tag + "_gap"] = (top2[..., 1] - top2[..., 0]).numpy()
            vq[tag + "_mindist"] = top2[..., 0].numpy()
            vq[tag + "_deq_sha"] = np.frombuffer(bytes.fromhex(sha(deq)), dtype=np.uint8)
        np.savez_compressed(os.path.join(OUT, "f2_vq.npz"), **vq)

    # ---- F2b: the VQ kernel's own configuration at FULL size (BASELINE configs[3]) --------------------------
    # the reference's _multiCodebookQuantization.encode (mcquic/modules/quantizer.py:144-179) on bench.py's `vq_config4` tensors
    # and on the qp=2 model's level-0 shape at batch 32: per-image code hashes + every near-tie (top-2 gap < F5C_NEAR in the
    # reference's own float32 distances) with both candidates -- the F5c format, a few KB.  Two images per call (the [n, m, h, w, k]
    # distance tensor of the whole batch would be 3.2 GB).
    if want("f2b"):
        from mcquic_amd.utils.synthetic import VQ_CASES, vq_case
        f2b = {"near_threshold": np.array([F5C_NEAR])}
        for tag in VQ_CASES:
            lat, cb = vq_case(tag)
            q = RQ._multiCodebookQuantization(torch.nn.Parameter(cb), 0.0)
            hashes, near, near_gap = [], [], []
            smallest = np.inf
            for lo in range(0, lat.shape[0], 2):
                xs = lat[lo:lo + 2]
                with torch.inference_mode():
                    dist = q._distance(xs)
                    code = q.encode(xs)
                top2 = torch.topk(dist, 2, dim=-1, largest=False)
                gap = top2.values[..., 1] - top2.values[..., 0]
                second = torch.where(top2.indices[..., 0] == code, top2.indices[..., 1], top2.indices[..., 0])
                smallest = min(smallest, float(gap.min()))
                for i in range(len(xs)):
                    hashes.append(code_hash(code[i]))
                for (i, g, yy, xx) in (gap < F5C_NEAR).nonzero().tolist():
                    near.append((lo + i, g, yy, xx, int(code[i, g, yy, xx]), int(second[i, g, yy, xx])))
                    near_gap.append(float(gap[i, g, yy, xx]))
                print(f"f2b {tag}: images {lo}..{lo + len(xs) - 1}  near-ties {len(near)}", flush=True)
            c = VQ_CASES[tag]
            f2b[tag + "_shape"] = np.array([c["m"], c["k"], c["d"], c["n"], c["h"], c["w"], c["seed"]])
            f2b[tag + "_code_hash"] = np.stack(hashes)
            f2b[tag + "_near"] = np.array(near, dtype=np.int32).reshape(-1, 6)
            f2b[tag + "_near_gap"] = np.array(near_gap, dtype=np.float32)
            f2b[tag + "_smallest_gap"] = np.array([smallest], dtype=np.float32)
            f2b[tag + "_input_sha"] = np.frombuffer(bytes.fromhex(sha(lat)) + bytes.fromhex(sha(cb)), dtype=np.uint8)
        np.savez_compressed(os.path.join(OUT, "f2b_vq_fullsize.npz"), **f2b)

    # ---- F15: NeonQuantizer (mcquic/modules/quantizer.py:469-573), the quantizer class no model of the snapshot builds --------------
    # encode / decode of the REAL class on a seeded 32-channel tensor: codes, the reference's own top-2 gaps, the restored tensor
    if want("f15"):
        from oracle import neon_ref as NR
        m15, k15 = [2, 4, 1], [64, 32, 16]
        sd15 = NR.make_neon_quantizer_state_dict(m15, k15, seed=5)
        nq = RQ.NeonQuantizer(m15, k15).eval()
        nq.load_state_dict(sd15, strict=True)
        x15 = rand((2, 32, 48, 80), 15)
        gaps = []
        orig_distance = RQ._multiCodebookQuantization._distance

        def rec(self, x):
            dist = orig_distance(self, x)
            top2 = torch.topk(dist, 2, dim=-1, largest=False).values
            gaps.append((top2[..., 1] - top2[..., 0]).clone())
            return dist
        RQ._multiCodebookQuantization._distance = rec
        try:
            with torch.inference_mode():
                codes15 = nq.encode(x15)
        finally:
            RQ._multiCodebookQuantization._distance = orig_distance
        with torch.inference_mode():
            rec15 = nq.decode(codes15)
        f15 = {"m": np.array(m15), "k": np.array(k15), "x_shape": np.array(x15.shape), "rec_strided": rec15[..., ::4, ::4].numpy(), "rec_mean_abs": np.array([float(rec15.abs().mean())]),
               "keys": np.array(sorted(nq.state_dict().keys()))}
        for lv, c in enumerate(codes15):
            f15[f"code{lv}"] = c.numpy().astype(np.int16)
            f15[f"gap{lv}"] = gaps[lv].numpy()
        np.savez_compressed(os.path.join(OUT, "f15_neon_quantizer.npz"), **f15)

    # ---- F4: the full small model Compressor(8, 2, [32, 16, 8]) ----------------------------------------
    if want("f4"):
        small = {}
        sd = R.make_state_dict(8, 2, [32, 16, 8], seed=1)
        model = ref_harness.reference_compressor(8, 2, [32, 16, 8], sd)
        for tag, (n, h, w) in {"pad": (2, 200, 136), "aligned": (1, 128, 256)}.items():
            xi = R.make_images(n, h, w)
            with torch.inference_mode():
                codes = model.encode(xi)
                rec = model.decode(codes)
            small[tag + "_shape"] = np.array([n, h, w])
            for lv, cd in enumerate(codes):
                small[f"{tag}_code{lv}"] = cd.numpy().astype(np.int16)
            small[tag + "_rec_strided"] = rec[..., ::4, ::4].numpy()
            small[tag + "_rec_crop"] = rec[..., 32:96, 32:96].numpy()
            small[tag + "_rec_sha"] = np.frombuffer(bytes.fromhex(sha(rec)), dtype=np.uint8)
            small[tag + "_padded_sha"] = np.frombuffer(bytes.fromhex(sha(AlignedPadding()(xi))), dtype=np.uint8)
        np.savez_compressed(os.path.join(OUT, "f4_small_model.npz"), **small)

    # ---- F5: the qp=2 model Compressor(128, 2, [8192, 2048, 512]) on one 256x384 image -----------------
    if want("f5"):
        qp2 = {}
        sd = R.make_state_dict(128, 2, [8192, 2048, 512], seed=0)
        model = ref_harness.reference_compressor(128, 2, [8192, 2048, 512], sd)
        xi = R.make_images(1, 256, 384)
        with torch.inference_mode():
            codes = model.encode(xi)
            rec = model.decode(codes)
        qp2["shape"] = np.array([1, 256, 384])
        qp2["n_state_dict_entries"] = np.array([len(model.state_dict())])
        for lv, cd in enumerate(codes):
            qp2[f"code{lv}"] = cd.numpy().astype(np.int16)
        qp2["rec_crop"] = rec[:, :, 96:160, 160:224].numpy()
        qp2["rec_mean_abs"] = np.array([rec.abs().mean().item()])
        qp2["rec_sha"] = np.frombuffer(bytes.fromhex(sha(rec)), dtype=np.uint8)
        np.savez_compressed(os.path.join(OUT, "f5_qp2_model.npz"), **qp2)
    # ---- F6: training-mode forward of the small model (reference with its one broken attribute repaired:
    if want("f6"):
        #          _multiCodebookQuantization._freqEMA = the level's entropy-coder EMA, see oracle/mcquic_ref.py) -----
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            dist.init_process_group("gloo", rank=0, world_size=1)
        ch, m, ks = 8, 2, [32, 16, 8]
        sd = R.make_state_dict(ch, m, ks, seed=2)
        g = torch.Generator().manual_seed(3)
        for lv, k in enumerate(ks):
            f = torch.rand((m, k), generator=g) ** 3 + 1e-3
            sd[f"_quantizer._entropyCoder._freqEMA.{lv}"] = f / f.sum(-1, keepdim=True)
        model = ref_harness.reference_compressor(ch, m, ks, sd)
        for lv, enc in enumerate(model._quantizer._encoders):
            enc._quantizer._freqEMA = model._quantizer._entropyCoder._freqEMA[lv]
        model.train()
        xi = R.make_images(2, 128, 128, seed=4)
        shapes = [(2, m, 8, 8, 32), (2, m, 4, 4, 16), (2, m, 2, 2, 8)]
        us = [(torch.rand(sh, generator=g), torch.rand(sh, generator=g)) for sh in shapes]
        it = iter([u for pair in us for u in pair])
        orig = torch.rand_like
        torch.rand_like = lambda t, **kw: next(it).clone()
        try:
            xHat, yHat, codes, logits = model(xi.clone())
        finally:
            torch.rand_like = orig
        f6 = {"xHat_strided": xHat.detach()[..., ::2, ::2].numpy(), "yHat": yHat.detach().numpy()}
        for lv in range(3):
            f6[f"code{lv}"] = codes[lv].numpy().astype(np.int16)
            f6[f"logit{lv}"] = logits[lv].detach().numpy()
            f6[f"ema{lv}"] = model._quantizer._entropyCoder._freqEMA[lv].detach().numpy()
        np.savez_compressed(os.path.join(OUT, "f6_train_forward.npz"), **f6)

    # ---- F8: rANS byte streams + quantized CDFs from the reference's native coder (oracle/_ref, built from the
    if want("f8"):
        #          reference's own sources by oracle/Makefile) ---------------------------------------------------------
        import glob
        import importlib.util
        so = glob.glob(os.path.join(ROOT, "oracle", "_ref", "rans*.so"))
        if so:
            spec = importlib.util.spec_from_file_location("rans", so[0])
            RA = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(RA)
            rng = np.random.default_rng(0)
            f8 = {}
            for tag, k, m, h, w in (("k8", 8, 2, 5, 7), ("k512", 512, 2, 12, 8), ("k8192", 8192, 2, 48, 32)):
                pmfs = []
                for g in range(m):
                    pm = rng.random(k).astype(np.float32) ** (4 if g == 0 else 1)
                    pmfs.append((pm / pm.sum()).astype(np.float32))
                cdfs = [RA.pmfToQuantizedCDF(pm.tolist(), 16) for pm in pmfs]
                sym = rng.integers(0, k, m * h * w).astype(np.int32)
                idx = np.repeat(np.arange(m, dtype=np.int32), h * w)
                by = RA.RansEncoder().encodeWithIndexes(sym.tolist(), idx.tolist(), cdfs, [k + 2] * m, [0] * m)
                f8[tag + "_pmf"] = np.stack(pmfs)
                f8[tag + "_cdf"] = np.asarray(cdfs, dtype=np.uint32)
                f8[tag + "_sym"] = sym
                f8[tag + "_shape"] = np.array([m, h, w, k])
                f8[tag + "_bytes"] = np.frombuffer(by, dtype=np.uint8)
            # bypass path: CompressAI's own convention cdfSizes = len(cdf) (sentinel = last slot), symbols beyond it / negative
            k = 16
            pm = (np.ones(k) / k).astype(np.float32)
            cdf = RA.pmfToQuantizedCDF(pm.tolist(), 16)
            sym = np.array([0, 3, 14, 15, 16, 40, 1000, -1, -7, 5], dtype=np.int32)
            by = RA.RansEncoder().encodeWithIndexes(sym.tolist(), [0] * len(sym), [cdf], [k + 1], [0])
            f8["bypass_cdf"] = np.asarray(cdf, dtype=np.uint32)
            f8["bypass_sym"] = sym
            f8["bypass_bytes"] = np.frombuffer(by, dtype=np.uint8)
            np.savez_compressed(os.path.join(OUT, "f8_rans.npz"), **f8)
    # ---- F7: validation metrics (MS-SSIM, PSNR, IdealBPP) from the reference's own validate/ code ---------
    if want("f7"):
        import importlib.util
        from oracle import metrics_ref as MR        # generators only
        spec = importlib.util.spec_from_file_location("ref_metrics", os.path.join(ref_harness.REF, "mcquic/validate/metrics.py"))
        RM = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(RM)
        RH = ref_harness.load_validate_handlers()
        f7 = {"cases": np.array(F7_CASES, dtype=np.int64)}
        msssim_mod, psnr_mod, decibel = RM.MsSSIM(sizeAverage=False), RM.PSNR(sizeAverage=False), RH["Decibel"](1.0)
        for i, (seed, n, h, w) in enumerate(F7_CASES):
            x, y = MR.make_u8_pair(seed, n, h, w)
            out = msssim_mod(x.float(), y.float())                    # handlers.py:26 (1 - ms_ssim)
            f7[f"msssim_{i}"] = (1.0 - out).numpy()
            f7[f"msssim_db_{i}"] = decibel(out).numpy()
            f7[f"psnr_{i}"] = psnr_mod(x.float(), y.float()).numpy()
            f7[f"sha_{i}"] = np.frombuffer(bytes.fromhex(sha(x) + sha(y)), dtype=np.uint8)
        ks, ms = list(MR.CODE_BATCH_KS), [2, 2, 2]
        handler = RH["IdealBPP"](ms, ks)
        for codes in MR.make_code_batches():                            # accumulated over two batches like a validation run
            handler(codes=codes, images=torch.zeros(3, 3, 768, 512, dtype=torch.uint8))
        f7["ideal_bpp"] = np.array([handler.Result], dtype=np.float64)
        np.savez_compressed(os.path.join(OUT, "f7_metrics.npz"), **f7)

    # ---- F1b: per-block I/O at C = 128, the network's width ------------------------------
    if want("f1b"):
        blocks = {}
        c = 128
        x = rand((1, c, 9, 10), 131)
        blocks["x"] = x.numpy()
        for name, ctor, mk in [("ResidualBlock", lambda: RN.ResidualBlock(c, c), R._rb),
                               ("ResidualBlockWithStride", lambda: RN.ResidualBlockWithStride(c, c), R._rb_stride),
                               ("ResidualBlockShuffle", lambda: RN.ResidualBlockShuffle(c, c), R._rb_shuffle),
                               ("AttentionBlock", lambda: RN.blocks.AttentionBlock(c), R._attn)]:
            sd = {}
            mk(sd, "", c, 21)
            mod = ctor().eval()
            mod.load_state_dict(sd, strict=True)
            with torch.inference_mode():
                blocks[name] = mod(x.clone()).numpy()
        for name, cls in [("GenDivNorm", RN.GenDivNorm), ("InvGenDivNorm", RN.InvGenDivNorm)]:
            sd = {}
            R._gdn_params(sd, "", c, 22)
            mod = cls(c).eval()
            mod.load_state_dict(sd, strict=True)
            with torch.inference_mode():
                blocks[name] = mod(x.clone() * 2).numpy()
        np.savez_compressed(os.path.join(OUT, "f1b_blocks_c128.npz"), **blocks)

    # ---- F5b: the qp=2 model at BASELINE's sizes: 2 x 3 x 768 x 512 and 1 x 3 x 1152 x 2048 (assets/sample.png's
    #           geometry, no padding): every code index, the reconstruction's SHA, a strided view and a crop -------
    if want("f5b"):
        big = {}
        sd = R.make_state_dict(128, 2, [8192, 2048, 512], seed=0)
        model = ref_harness.reference_compressor(128, 2, [8192, 2048, 512], sd)
        for tag, (n, h, w, seed) in {"kodak": (2, 768, 512, 3407), "sample": (1, 1152, 2048, 11)}.items():
            xi = R.make_images(n, h, w, seed=seed)
            gaps = []                           # the reference's own top-2 distance gap per vector (near-tie audit)
            orig_distance = RQ._multiCodebookQuantization._distance

            def recording_distance(self, x):
                dist = orig_distance(self, x)
                top2 = torch.topk(dist, 2, dim=-1, largest=False).values
                gaps.append((top2[..., 1] - top2[..., 0]).clone())
                return dist
            RQ._multiCodebookQuantization._distance = recording_distance
            try:
                with torch.inference_mode():
                    codes = model.encode(xi)
            finally:
                RQ._multiCodebookQuantization._distance = orig_distance
            with torch.inference_mode():
                rec = model.decode(codes)
            big[tag + "_shape"] = np.array([n, h, w, seed])
            for lv, cd in enumerate(codes):
                big[f"{tag}_code{lv}"] = cd.numpy().astype(np.int16)
                big[f"{tag}_gap{lv}"] = gaps[lv].numpy()
            big[tag + "_rec_strided"] = rec[..., ::16, ::16].numpy()
            big[tag + "_rec_crop"] = rec[:, :, h // 2 - 32:h // 2 + 32, w // 2 - 32:w // 2 + 32].numpy()
            big[tag + "_rec_mean_abs"] = np.array([rec.abs().mean().item()])
            big[tag + "_rec_sha"] = np.frombuffer(bytes.fromhex(sha(rec)), dtype=np.uint8)
        np.savez_compressed(os.path.join(OUT, "f5b_qp2_fullsize.npz"), **big)

    # ---- F9: reAssignCodebook (mcquic/modules/quantizer.py:111-136) with torch.randperm made to return a recorded
    #          permutation per crowded group ---------------------------------------------------------------------
    if want("f9"):
        f9 = {"cases": np.array([(m, k, d, seed) for m, k, d, _, seed in R.REASSIGN_CASES], dtype=np.int64)}
        for ci, (m, k, d, dead_frac, seed) in enumerate(R.REASSIGN_CASES):
            cb, f, perms = R.reassign_case(m, k, d, dead_frac, seed)
            q = RQ._multiCodebookQuantization(torch.nn.Parameter(cb.clone()), 0.0)
            crowded = [int((f[gi] < 1e-6).sum()) > k // 2 for gi in range(m)]
            it = iter([perms[gi] for gi in range(m) if crowded[gi]])
            orig = torch.randperm
            torch.randperm = lambda n, **kw: next(it).clone()
            try:
                changed = q.reAssignCodebook(f.clone())
            finally:
                torch.randperm = orig
            f9[f"freq_{ci}"] = f.numpy()
            for gi in range(m):
                f9[f"perm_{ci}_{gi}"] = perms[gi].numpy().astype(np.int32)
            f9[f"new_codebook_{ci}"] = q._codebook.detach().numpy()
            f9[f"changed_{ci}"] = changed.numpy()
        np.savez_compressed(os.path.join(OUT, "f9_reassign.npz"), **f9)
    # ---- F10 / F11: the Neon model family (mcquic/modules/compressor.py:181-241, ResidualBackwardQuantizer quantizer.py:577-765):
    #           encode / decode / residual_backward / residual_forward and the training-mode forward; F11 = the same with
    #           denseNorm=True (nn.GroupNorm in place of the ResidualBlocks' second activation, nn/blocks.py:179-200) -------
    if want("f10"):
        capture_neon(C, RQ, False, "f10_neon.npz")
    if want("f11"):
        capture_neon(C, RQ, True, "f11_neon_dense_norm.npz")
    # ---- F13 / F14: Neon at the shapes the snapshot's trainer builds (mcquic/train/ddp.py:79-83, configs/neon.yaml: channel 32,
    #           k = 4096, five levels) on 2 x 256x256: stride-1 stem, AttentionBlocks at FULL resolution, widths 32 / 64 -------
    if want("f13"):
        capture_neon(C, RQ, False, "f13_neon_k4096.npz", NEON_K4096, hw=256, stride=8, logit_stride=128)
    if want("f14"):
        capture_neon(C, RQ, True, "f14_neon_k4096_dense_norm.npz", NEON_K4096, hw=256, stride=8, logit_stride=128)

    # ---- F12: the reference's second listed model, No. 12 = Compressor(192, 12, [8192, 2048, 512]) (README.md:306):
    #           channel 192, twelve codebooks of 16-dimensional codewords.  1 x 768x512 and 2 x 200x136 (sizes that are no multiple of 128):
    #           every code, the reference's top-2 gap per vector, strided pixels + crop of the reconstruction --------
    if want("f12"):
        m12 = {}
        sd = R.make_state_dict(192, 12, [8192, 2048, 512], seed=12)
        model = ref_harness.reference_compressor(192, 12, [8192, 2048, 512], sd)
        for tag, (n, h, w, seed) in {"kodak": (1, 768, 512, 3412), "ragged": (2, 200, 136, 13)}.items():
            xi = R.make_images(n, h, w, seed=seed)
            gaps = []
            orig_distance = RQ._multiCodebookQuantization._distance

            def recording_distance12(self, x):
                dist = orig_distance(self, x)
                top2 = torch.topk(dist, 2, dim=-1, largest=False).values
                gaps.append((top2[..., 1] - top2[..., 0]).clone())
                return dist
            RQ._multiCodebookQuantization._distance = recording_distance12
            try:
                with torch.inference_mode():
                    codes = model.encode(xi)
            finally:
                RQ._multiCodebookQuantization._distance = orig_distance
            with torch.inference_mode():
                rec = model.decode(codes)
            m12[tag + "_shape"] = np.array([n, h, w, seed])
            for lv, cd in enumerate(codes):
                m12[f"{tag}_code{lv}"] = cd.numpy().astype(np.int16)
                m12[f"{tag}_gap{lv}"] = gaps[lv].numpy()
            m12[tag + "_rec_strided"] = rec[..., ::16, ::16].numpy()
            m12[tag + "_rec_crop"] = rec[:, :, h // 2 - 32:h // 2 + 32, w // 2 - 32:w // 2 + 32].numpy()
            m12[tag + "_rec_mean_abs"] = np.array([rec.abs().mean().item()])
        np.savez_compressed(os.path.join(OUT, "f12_model12.npz"), **m12)

    # ---- F5c: BASELINE configs[2] -- the 256 images `bench.py --gpus 8` generates (8 rank-seeded shards of 32 x 768x512,
    #           bench.py's own random-init qp=2 weights) through the REFERENCE in float32, and the reference's own
    #           sensitivity on them: the same model in float64 (and shard 0 with 1 instead of 8 threads).  Stored: a hash
    #           of every image's codes per level, every near-tie vector (top-2 gap < 2e-5 in the reference's own
    #           distances) with both candidates, and the vectors where the reference DISAGREES WITH ITSELF (float32 vs
    #           float64) with their gaps -- the data behind the near-tie protocol of DESIGN section 6.  ~25 min of CPU:
    #           only when named (`python make_golden.py f5c`) -------------------------------------------------------
    if "f5c" in [a.lower() for a in sys.argv[1:]]:
        capture_f5c(RQ)
    if "f5c" in [a.lower() for a in sys.argv[1:]] or "f5c_backend" in [a.lower() for a in sys.argv[1:]]:
        capture_f5c_backend(RQ)

    print("golden vectors written to", OUT)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f"  {f}: {os.path.getsize(os.path.join(OUT, f)) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
