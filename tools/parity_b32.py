#!/usr/bin/env python3
"""Index / pixel parity of the HIP path against the CPU oracle at BASELINE's full workloads, with a first-flip census.

    python tools/parity_b32.py                       # configs[1]: 32 x 768x512, oracle weights (the round-2 audit)
    python tools/parity_b32.py --shards 8            # configs[2]: the 256 images `bench.py --gpus 8` generates (seed 3407 + rank,
                                                     #   bench.py's own random-init weights), shard after shard on one GPU
    python tools/parity_b32.py --shards 8 --winograd 2

Per shard the GPU encodes ONE batch of 32 images; the oracle runs in chunks of 4 on the host cores (~0.7 s per image and
direction on 16 cores: ~12 min for 256 images).  A *first flip* is a code that differs although every code upstream of it
(coarser... i.e. earlier levels of the same image) agrees: a near-tie decided the other way.  Deeper levels of that image
then quantize a different residual ("downstream differences": conditioned on other inputs, not errors).  For each first
flip the record holds the oracle's own distance gap between the two candidates.  With --shards the run is also checked
against the committed reference fixture tests/golden/f5c_config2_census.npz (captured from the REAL reference in the build
container): per-image code hashes (bit-equality with the reference itself, not only with its restatement), every first
flip must be one of the reference's recorded near-ties decided for the runner-up, and the reference's own float32-vs-float64
self-flips give the scale against which the HIP path's flips are judged.  Decoded pixels are compared from the ORACLE's
codes, so a flip cannot leak into the pixel bar."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# the bars are the ones tests/test_gpu_config2_census.py ENFORCES on the same census (one definition, so a profile written by this
# tool can never carry looser bars than the test that guards the workload)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_config2_census import GAP_BAR, MAX_FLIPS  # noqa: E402

CENSUS_CODES = 1032192   # codes of configs[2]'s 256 images (all levels): what MAX_FLIPS is counted over


def max_first_flips(total_codes: int) -> int:
    """MAX_FLIPS scaled to the number of codes audited (never below one)."""
    return max(1, -(-MAX_FLIPS * int(total_codes) // CENSUS_CODES))


def census(model, sd, x, dev, R, chunk, shard, fixture, rec):
    """Encode x [n, 3, H, W] on the GPU in one batch, compare with the oracle chunk by chunk; appends to `rec`."""
    n = x.shape[0]
    codes = [c.cpu() for c in model.encode(x.to(dev))]
    from mcquic_amd.utils.synthetic import code_hash
    for lo in range(0, n, chunk):
        xs = x[lo:lo + chunk]
        collect = {}
        want = R.quantizer_encode(sd, R.encoder(sd, R.aligned_padding(xs)), collect)
        alive = torch.ones(len(xs), dtype=torch.bool)
        for lv, wc in enumerate(want):
            g = codes[lv][lo:lo + chunk]
            rec["downstream"][lv] += int(((g != wc) & ~alive[:, None, None, None]).sum())
            bad = (g != wc) & alive[:, None, None, None]
            rec["total"][lv] += wc.numel()
            if bad.any():
                dist = R.vq_distance(collect["q"][lv], sd[f"_quantizer._encoders.{lv}._quantizer._codebook"]).double()
                for (i, gi, yy, xx) in bad.nonzero().tolist():
                    a, b = int(wc[i, gi, yy, xx]), int(g[i, gi, yy, xx])
                    gap = float(dist[i, gi, yy, xx, b] - dist[i, gi, yy, xx, a])
                    flip = {"shard": shard, "image": lo + i, "level": lv, "group": gi, "y": yy, "x": xx, "oracle": a, "hip": b,
                            "oracle_gap": gap}
                    if fixture is not None:
                        key = (shard, lo + i, lv, gi, yy, xx)
                        near = fixture["near"].get(key)
                        flip["in_reference_near_ties"] = near is not None
                        if near is not None:
                            flip["reference_best_second_gap"] = [near[0], near[1], near[2]]
                            flip["hip_took_the_reference_runner_up"] = (near[0] == a and near[1] == b)
                    rec["flips"].append(flip)
                rec["first"][lv] += int(bad.sum())
            alive &= ~bad.flatten(1).any(1)
            if fixture is not None:                      # the oracle against the reference's recorded hashes: always equal
                for i in range(len(xs)):
                    if code_hash(wc[i]) != fixture["hash"][shard * 32 + lo + i, lv].tobytes():
                        rec["oracle_vs_reference_hash_mismatches"] += 1
        if fixture is not None:
            for i in range(len(xs)):
                same = all(code_hash(codes[lv][lo + i]) == fixture["hash"][shard * 32 + lo + i, lv].tobytes() for lv in range(len(want)))
                rec["images_bit_equal_to_reference"] += int(same)
        rec_cpu = R.decode(sd, want)
        rec_gpu = model.decode([c.to(dev) for c in want]).cpu()          # pixels from the ORACLE's codes
        rec["pix_err"] = max(rec["pix_err"], float((rec_gpu - rec_cpu).abs().max()))
        rec["psnr_min"] = min(rec["psnr_min"], float(R.psnr(R.detransform(rec_gpu), R.detransform(rec_cpu)).min()))
        if fixture is not None and lo == 0:                              # ... and against the reference's own reconstruction
            ref = torch.from_numpy(fixture["rec_strided"][shard])
            rec["pix_err_vs_reference"] = max(rec["pix_err_vs_reference"], float((rec_gpu[0, :, ::16, ::16] - ref).abs().max()))
    rec["images"] += n


def load_fixture():
    path = os.path.join(ROOT, "tests", "golden", "f5c_config2_census.npz")
    if not os.path.exists(path):
        return None
    d = np.load(path)
    near = {tuple(int(v) for v in row[:6]): (int(row[6]), int(row[7]), float(gap)) for row, gap in zip(d["near"], d["near_gap"])}
    return {"hash": d["code_hash"], "near": near, "rec_strided": d["rec_strided"], "state_dict_sha": bytes(d["state_dict_sha"]).hex(),
            "x_sha": d["x_sha"], "selfflip_gap32": d["selfflip_gap32"].tolist(), "selfflip": d["selfflip"].tolist(),
            "selfflip_backend_gap32": d["selfflip_backend_gap32"].tolist() if "selfflip_backend_gap32" in d.files else [],
            "selfflip_backend": d["selfflip_backend"].tolist() if "selfflip_backend" in d.files else [],
            "codes_per_level": d["codes_per_level"].tolist()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=32, help="images per shard")
    ap.add_argument("--shards", type=int, default=0, help="N > 0: bench.py's own workload, ranks 0..N-1 (configs[2] = 8)")
    ap.add_argument("--chunk", type=int, default=4)
    ap.add_argument("--out", default=None)
    ap.add_argument("--winograd", type=int, nargs="?", const=1, default=0, choices=(0, 1, 2),
                    help="audit an OPT-IN Winograd path instead of the default direct form: 1 = F(2, 3) along x, 2 = F(2x2, 3x3)")
    a = ap.parse_args()
    from mcquic_amd import Compressor, ops
    from mcquic_amd.utils import synthetic as S
    if a.winograd:
        ops.set_winograd(a.winograd)
    from oracle import mcquic_ref as R
    dev = torch.device("cuda:0")
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 64)))
    fixture = None
    if a.shards:
        model = S.bench_model()
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        fixture = load_fixture() if a.images == 32 else None
        if fixture is not None and S.state_dict_sha(sd) != fixture["state_dict_sha"]:
            raise SystemExit("the random-init weights on this box are not the ones the reference fixture was captured with")
        model = model.to(dev)
    else:
        ks = [8192, 2048, 512]
        sd = R.make_state_dict(128, 2, ks, seed=0)
        model = Compressor(128, 2, ks).eval()
        model.load_state_dict(sd, strict=True)
        model = model.to(dev)
    rec = {"total": [0, 0, 0], "first": [0, 0, 0], "downstream": [0, 0, 0], "flips": [], "pix_err": 0.0, "psnr_min": float("inf"),
           "images": 0, "images_bit_equal_to_reference": 0, "oracle_vs_reference_hash_mismatches": 0, "pix_err_vs_reference": 0.0}
    t0 = time.time()
    for shard in range(max(a.shards, 1)):
        x = S.bench_images(shard, a.images) if a.shards else R.make_images(a.images, 768, 512, seed=3407)
        census(model, sd, x, dev, R, a.chunk, shard, fixture, rec)
        print(f"shard {shard}: first flips so far {rec['first']}, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    gaps = [f["oracle_gap"] for f in rec["flips"]]
    out = {"arithmetic": ("OPT-IN winograd F(2x2,3x3) on the large 3x3 stride-1 layers" if a.winograd == 2 else
                          "OPT-IN winograd F(2,3) on the large 3x3 stride-1 layers") if a.winograd else "direct form (default)",
           "workload": (f"qp=2 model with bench.py's random-init weights, {a.shards} shards x {a.images} x 3 x 768 x 512, image seeds 3407 + rank "
                        f"(BASELINE configs[2] = the batches `bench.py --gpus {a.shards}` generates)") if a.shards else
                       f"qp=2 model, {a.images} x 3 x 768 x 512, seed 3407 (BASELINE configs[1])",
           "images": rec["images"], "codes_per_level": rec["total"], "first_flips_per_level": rec["first"], "first_flips": sum(rec["first"]),
           "first_flip_rate_per_code": sum(rec["first"]) / max(sum(rec["total"]), 1),
           "worst_oracle_gap_at_a_first_flip": max(gaps) if gaps else 0.0, "first_flip_list": rec["flips"],
           "downstream_differences_per_level": rec["downstream"],
           "bars": {"oracle_gap_below": GAP_BAR, "first_flips_at_most": max_first_flips(sum(rec["total"])), "pixels_max_abs": 1e-4,
                    "source": "tests/test_gpu_config2_census.py (GAP_BAR, MAX_FLIPS per 1 032 192 codes)"},
           "note": "a first flip = a code that differs although all codes upstream of it agree; excused only if the oracle's own "
                   "distance gap between the two candidates is below the bar; deeper levels of that image then quantize a different residual",
           "decode_max_abs_err": rec["pix_err"], "psnr_gpu_vs_cpu_u8_min_db": round(rec["psnr_min"], 2),
           "oracle_seconds": round(time.time() - t0, 1), "device": torch.cuda.get_device_name(0)}
    if fixture is not None:
        ref_gaps = fixture["selfflip_gap32"]
        out["reference_fixture"] = {
            "file": "tests/golden/f5c_config2_census.npz (the REAL reference on the same 256 images, captured in the build container)",
            "images_bit_equal_to_the_reference_all_levels": rec["images_bit_equal_to_reference"],
            "oracle_vs_reference_code_hash_mismatches": rec["oracle_vs_reference_hash_mismatches"],
            "first_flips_inside_the_reference_near_tie_set": sum(1 for f in rec["flips"] if f.get("in_reference_near_ties")),
            "first_flips_that_took_the_reference_runner_up": sum(1 for f in rec["flips"] if f.get("hip_took_the_reference_runner_up")),
            "reference_self_flips_float32_vs_float64": len(ref_gaps), "reference_self_flip_gaps_float32": ref_gaps,
            "reference_self_flips_onednn_vs_native_conv": len(fixture["selfflip_backend_gap32"]),
            "reference_self_flip_gaps_onednn_vs_native_conv": fixture["selfflip_backend_gap32"],
            "reference_self_flip_sites": [f[:6] for f in fixture["selfflip"] + fixture["selfflip_backend"]],
            "reference_near_ties_below_2e-5": len(fixture["near"]),
            "decode_max_abs_err_vs_reference_strided_first_image_of_each_shard": rec["pix_err_vs_reference"]}
    path = a.out or os.path.join(ROOT, "gpurun_out", "parity_b256.json" if a.shards else "parity_b32.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, "w"), indent=1)
    short = dict(out)
    short.pop("first_flip_list")
    print(json.dumps(short))
    ok = (not gaps or max(gaps) < GAP_BAR) and sum(rec["first"]) <= max_first_flips(sum(rec["total"])) and rec["pix_err"] <= 1e-4
    if fixture is not None:
        ok = ok and rec["oracle_vs_reference_hash_mismatches"] == 0 and all(f.get("hip_took_the_reference_runner_up") for f in rec["flips"])
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
