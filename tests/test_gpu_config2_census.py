"""BASELINE configs[2] at its own size on ONE GPU: the 256 images `bench.py --gpus 8` generates (8 rank-seeded shards of
32 x 768x512, bench.py's random-init qp=2 weights) against the REAL reference, through the fixture
tests/golden/f5c_config2_census.npz captured from it in the build container (tests/golden/make_golden.py f5c):

  * per image and level a hash of the reference's codes  -> which images are bit-equal to the reference on every level;
  * every near-tie vector of the reference (top-2 distance gap < 2e-5 in ITS float32 arithmetic) with both candidates;
  * the vectors where the reference disagrees with ITSELF -- run in float64, and run in float32 with oneDNN switched off
    (ATen's native convolution: another summation order) -- its own sensitivity: on these 256 images BOTH probes flip one
    and the same level-0 code, at a gap of 2.4e-7 between its two best codewords.

A *first flip* (a code that differs although everything upstream of it agrees) is only legitimate at one of the recorded
near-ties, taking exactly the reference's runner-up there; the census bounds their number and their gaps with data from the
reference instead of an argument about float32 noise (missing #2 / #3).  Images that do differ are located with
the CPU oracle (bit-equal to the reference: its hashes are checked against the fixture here too)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "f5c_config2_census.npz")

# Bars set from what was MEASURED (profiles/r03_parity_b256.json, r04_parity_b256.json): the direct form has 2 first flips in the
# 1 032 192 codes, at gaps 1.2e-7 and 3.6e-7 (1 and 3 float32 ulps of distances of order 1); the opt-in F(2x2, 3x3) mode 3; the
# reference flips 1 code against ITSELF (float32 vs float64, oneDNN vs native convolution) at 2.4e-7.
GAP_BAR = 2e-6            # a first flip is excused only below this gap in the reference's own distances (~8x the widest seen)
MAX_FLIPS = 8             # ... and at most this many in the 1.03 M codes (4x the direct form's count, the order of the reference's own)


def _census(dev, winograd=0):
    from mcquic_amd import ops
    from mcquic_amd.utils import synthetic as S
    from oracle import mcquic_ref as R
    d = np.load(FIXTURE)
    shards, per = int(d["shape"][0]), int(d["shape"][1])
    near = {tuple(int(v) for v in row[:6]): (int(row[6]), int(row[7]), float(gap)) for row, gap in zip(d["near"], d["near_gap"])}
    model = S.bench_model()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    assert S.state_dict_sha(sd) == bytes(d["state_dict_sha"]).hex(), "random-init weights differ from the capture's"
    model = model.to(dev)
    ops.set_winograd(winograd)
    try:
        flips, equal_images, total, pix_err = [], 0, 0, 0.0
        for r in range(shards):
            x = S.bench_images(r, per)
            codes = [c.cpu() for c in model.encode(x.to(dev))]
            total += sum(c.numel() for c in codes)
            first_codes = None
            for i in range(per):
                same = [S.code_hash(codes[lv][i]) == d["code_hash"][r * per + i, lv].tobytes() for lv in range(3)]
                if all(same):
                    equal_images += 1
                    if i == 0:
                        first_codes = [c[:1] for c in codes]
                    continue
                # locate the first flip with the oracle (one image: ~1 s); the oracle itself must hash to the reference
                collect = {}
                want = R.quantizer_encode(sd, R.encoder(sd, R.aligned_padding(x[i:i + 1])), collect)
                assert all(S.code_hash(want[lv][0]) == d["code_hash"][r * per + i, lv].tobytes() for lv in range(3)), \
                    "the CPU oracle's codes do not hash to the reference's"
                if i == 0:
                    first_codes = want
                lv = same.index(False)
                bad = (codes[lv][i] != want[lv][0]).nonzero().tolist()
                dist = R.vq_distance(collect["q"][lv], sd[f"_quantizer._encoders.{lv}._quantizer._codebook"]).double()[0]
                for g, yy, xx in bad:
                    a, b = int(want[lv][0, g, yy, xx]), int(codes[lv][i, g, yy, xx])
                    flips.append({"key": (r, i, lv, g, yy, xx), "ref": a, "hip": b, "gap": float(dist[g, yy, xx, b] - dist[g, yy, xx, a])})
            # pixels: decode of the reference's own codes for the shard's first image against the reference's reconstruction
            rec = model.decode([c.to(dev) for c in first_codes]).cpu()
            pix_err = max(pix_err, float((rec[0, :, ::16, ::16] - torch.from_numpy(d["rec_strided"][r])).abs().max()))
    finally:
        ops.set_winograd(0)
    return d, near, flips, equal_images, total, pix_err


def _judge(d, near, flips, equal_images, total, pix_err, tag="direct"):
    n_images = int(d["shape"][0] * d["shape"][1])
    ref_self = np.concatenate([d["selfflip_gap32"], d["selfflip_backend_gap32"]]) if "selfflip_backend_gap32" in d.files else d["selfflip_gap32"]
    print(f"config[2] census: {equal_images}/{n_images} images bit-equal to the reference on all levels; {len(flips)} first flips in "
          f"{total} codes, gaps {[round(f['gap'], 9) for f in flips]}; reference self-flips (float32 vs float64, oneDNN vs native conv) {len(ref_self)} with "
          f"gaps {ref_self.tolist()}; pixels vs reference {pix_err:.2e}")
    assert pix_err <= 1e-4
    assert len(flips) <= MAX_FLIPS, f"{len(flips)} first flips in {total} codes (bar {MAX_FLIPS}: 2 were measured, the reference flips 1 against itself)"
    from _record import record
    record(f"config2_census[{tag}]", first_flips=len(flips), codes=total,
           widest_gap=max([f["gap"] for f in flips], default=0.0), images_bit_equal=equal_images, pixels_vs_reference=pix_err, bar_flips=MAX_FLIPS, bar_gap=GAP_BAR)
    assert equal_images >= n_images - len(flips)
    for f in flips:
        assert f["key"] in near, f"a code differs away from every recorded near-tie of the reference: {f}"
        best, second, gap = near[f["key"]]
        assert (f["ref"], f["hip"]) == (best, second), f"the HIP path chose neither of the reference's two candidates: {f}"
        assert 0 <= f["gap"] < GAP_BAR and abs(f["gap"] - gap) <= 1e-6
    # data, not an argument: the reference's own float32-vs-float64 flips on these images sit at the same gap scale; the HIP
    # path's flips must not reach further than a small multiple of the widest gap at which the reference itself flips
    if len(ref_self) and flips:
        assert max(f["gap"] for f in flips) <= max(8.0 * float(ref_self.max()), 2e-6)


@pytest.mark.skipif(not os.path.exists(FIXTURE), reason="fixture not captured")
def test_config2_256_images_against_the_reference(dev):
    _judge(*_census(dev))


@pytest.mark.skipif(not os.path.exists(FIXTURE), reason="fixture not captured")
def test_config2_256_images_winograd_fast_mode(dev):
    """The same census for the OPT-IN F(2x2, 3x3) mode (not the reference's arithmetic: documented as a fast mode only if its
    flips stay inside the reference's near-tie set like the direct form's)."""
    _judge(*_census(dev, winograd=2), tag="winograd2d")
