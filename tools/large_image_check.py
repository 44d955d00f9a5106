#!/usr/bin/env python3
"""An image beyond the kernels' 2 GiB-per-image addressing, end to end (VERDICT r3 missing #4): 3584 x 3840 = 13.8 MP, whose
1792 x 1920 maps are 128 x 1792 x 1920 x 4 B = 1.76 GB -- with the prefetch rings' over-read allowance (32 channels) past the 2 GiB
limit, so the convolutions on them (first ResidualBlock of the encoder, last of the decoder, the image head) run in row bands
(ops._conv2d_banded) at the REAL limit, not a lowered one.  Encode on the GPU against the CPU oracle (near-tie audit), decode of
the oracle's codes against the oracle's reconstruction.  ~2 minutes, most of it the CPU oracle.  Prints one JSON line.

    python tools/large_image_check.py [--height 3584 --width 3840]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=3584)
    ap.add_argument("--width", type=int, default=3840)
    a = ap.parse_args()
    from mcquic_amd import Compressor, ops
    from oracle import mcquic_ref as R
    dev = torch.device("cuda:0")
    sd = R.make_state_dict(128, 2, [8192, 2048, 512], seed=0)
    model = Compressor(128, 2, [8192, 2048, 512]).eval()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    x = R.make_images(1, a.height, a.width, seed=41)
    banded = []
    orig = ops._conv2d_banded

    def spy(xx, w, stride, rows, fused):
        banded.append((tuple(xx.shape), w.cout, stride, rows))
        return orig(xx, w, stride, rows, fused)
    ops._conv2d_banded = spy
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    codes = model.encode(x.to(dev))
    torch.cuda.synchronize()
    t_enc = time.perf_counter() - t0
    torch.set_num_threads(os.cpu_count() or 1)
    t0 = time.perf_counter()
    collect = {}
    want = R.quantizer_encode(sd, R.encoder(sd, R.aligned_padding(x)), collect)
    t_cpu = time.perf_counter() - t0
    flips, worst_gap, alive = 0, 0.0, True
    per_level = []
    for lv, (g, w_) in enumerate(zip(codes, want)):
        bad = (g.cpu() != w_)
        per_level.append(int(bad.sum()))
        if alive and bad.any():
            dist = R.vq_distance(collect["q"][lv], sd[f"_quantizer._encoders.{lv}._quantizer._codebook"]).double()
            dg = torch.gather(dist, -1, g.cpu().unsqueeze(-1)).squeeze(-1)
            dw = torch.gather(dist, -1, w_.unsqueeze(-1)).squeeze(-1)
            worst_gap = max(worst_gap, float((dg - dw).abs()[bad].max()))
            flips += int(bad.sum())
            alive = False                        # deeper levels quantize another residual from here on
    n_enc_bands = len(banded)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec = model.decode([c.to(dev) for c in want])
    torch.cuda.synchronize()
    t_dec = time.perf_counter() - t0
    ref = R.decode(sd, want)
    err = float((rec.cpu() - ref).abs().max())
    out = {"image": [a.height, a.width], "megapixels": round(a.height * a.width / 1e6, 2),
           "stem_output_slab_gb": round(128 * (a.height // 2) * (a.width // 2) * 4 / 1e9, 3),
           "banded_launches_encode": n_enc_bands, "banded_launches_decode": len(banded) - n_enc_bands,
           "largest_banded_layer": max(banded, key=lambda b: b[0][1] * b[0][2] * b[0][3])[0] if banded else None,
           "codes": [int(c.numel()) for c in want], "code_mismatches_per_level": per_level,
           "first_flips": flips, "worst_oracle_gap_at_a_first_flip": worst_gap,
           "decode_max_abs_err_vs_oracle": err, "gpu_encode_s": round(t_enc, 2), "gpu_decode_s": round(t_dec, 2), "cpu_oracle_encode_s": round(t_cpu, 1)}
    print(json.dumps(out))
    assert n_enc_bands > 0, "the image was meant to exceed the single-launch limit"
    assert worst_gap < 2e-6 and flips <= 4, out
    assert err <= 1e-4, out


if __name__ == "__main__":
    main()
