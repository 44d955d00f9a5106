"""End-to-end parity of mcquic_amd.Compressor (HIP) against the CPU oracle on seeded synthetic weights."""
import pytest
import torch

from oracle import mcquic_ref as R

pytestmark = pytest.mark.gpu


def _compare(dev, channel, m, k, n, h, w, seed, pix_tol):
    from mcquic_amd import Compressor
    sd = R.make_state_dict(channel, m, k, seed=seed)
    model = Compressor(channel, m, k).eval()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    x = R.make_images(n, h, w)
    collect = {}
    want_codes = R.quantizer_encode(sd, R.encoder(sd, R.aligned_padding(x)), collect)
    got_codes = model.encode(x.to(dev))
    # Near-tie protocol: a code may differ from the oracle's only where the oracle's own distance gap
    # between the two candidates is below 1e-5 (fp32 summation-order noise is ~1e-6); such a flip changes the residual
    # the deeper levels see, so an image is compared down to its first excused flip only.
    mism = 0
    alive = torch.ones(n, dtype=torch.bool)
    for lv, (g, wc) in enumerate(zip(got_codes, want_codes)):
        assert g.dtype == torch.int64 and g.shape == wc.shape
        bad = (g.cpu() != wc) & alive[:, None, None, None]
        if bad.any():
            cb = sd[f"_quantizer._encoders.{lv}._quantizer._codebook"]
            rows = torch.nonzero(bad.flatten(1).any(1)).flatten()
            dist = R.vq_distance(collect["q"][lv][rows], cb).double()
            dg = torch.gather(dist, -1, g.cpu()[rows].unsqueeze(-1)).squeeze(-1)
            dw = torch.gather(dist, -1, wc[rows].unsqueeze(-1)).squeeze(-1)
            gap = (dg - dw).abs()[bad[rows]].max().item()
            assert gap < 1e-5, f"level {lv}: {int(bad.sum())} code mismatches, worst oracle gap {gap:.3e}"
            mism += int(bad.sum())
            alive &= ~bad.flatten(1).any(1)
    # decode parity is checked from the ORACLE's codes so that an audited near-tie does not leak into pixels
    want = R.decode(sd, want_codes)
    got = model.decode([c.to(dev) for c in want_codes]).cpu()
    assert got.shape == want.shape
    err = (got - want).abs().max().item()
    assert err <= pix_tol, f"decode max abs err {err:.3e}"
    u_got, u_want = R.detransform(got), R.detransform(want)
    psnr = R.psnr(u_got, u_want)
    assert float(psnr.min()) > 60.0       # identical up to a few rounding-boundary pixels
    return mism, err


def test_small_model_padded_input(dev):
    _compare(dev, 8, 2, [32, 16, 8], n=2, h=200, w=136, seed=1, pix_tol=1e-4)


def test_small_model_aligned_input(dev):
    _compare(dev, 8, 2, [32, 16, 8], n=3, h=128, w=256, seed=2, pix_tol=1e-4)


def test_qp2_model_one_image(dev):
    """Compressor(128, 2, [8192, 2048, 512]) on one 256x384 image (the CPU oracle takes a few seconds)."""
    _compare(dev, 128, 2, [8192, 2048, 512], n=1, h=256, w=384, seed=0, pix_tol=1e-4)


def test_qp2_model_kodak_shape_batch(dev):
    """The BASELINE geometry itself: three 768x512 images through the qp=2 model against the CPU oracle (~20 s of CPU) --
    every tile variant of the three latent levels and the image-head kernel at the size the benchmark runs them."""
    mism, err = _compare(dev, 128, 2, [8192, 2048, 512], n=3, h=768, w=512, seed=5, pix_tol=1e-4)
    assert mism == 0, f"{mism} audited near-tie code mismatches (none has been observed so far)"


def test_qp2_model_eight_kodak_images_exact(dev):
    """Eight 768x512 images (24 k level-0 vectors x 2 codebooks) against the CPU oracle: zero code mismatches, no
    near-tie excuse taken (widen the index-parity sample)."""
    mism, err = _compare(dev, 128, 2, [8192, 2048, 512], n=8, h=768, w=512, seed=0, pix_tol=1e-4)
    assert mism == 0, f"{mism} code mismatches"


@pytest.mark.parametrize("level", [1, 2])
def test_qp2_model_winograd_opt_in(dev, level):
    """The OPT-IN Winograd paths (ops.set_winograd; never the default; 1 = F(2, 3) along x, 2 = F(2x2, 3x3)): three 768x512 images
    through the qp=2 model under the same near-tie protocol and the same 1e-4 pixel bar as the direct form -- every 3x3
    stride-1 layer with 18 k pixels or more takes a Winograd kernel (96x64 ... 384x256 maps), the rest the direct one."""
    from mcquic_amd import ops
    ops.set_winograd(level, min_pixels=3 * 96 * 64)
    try:
        launches = {"n": 0}
        real = ops._lib.load().mcq_conv2d_f32

        def counting(desc, stream):
            launches["n"] += bool(desc._obj.flags & ((ops.CONV_WINOGRAD2D | ops.CONV_WINOGRAD2D16) if level == 2 else ops.CONV_WINOGRAD))
            return real(desc, stream)
        lib = ops._lib.load()
        lib.mcq_conv2d_f32 = counting
        try:
            mism, err = _compare(dev, 128, 2, [8192, 2048, 512], n=3, h=768, w=512, seed=5, pix_tol=1e-4)
        finally:
            lib.mcq_conv2d_f32 = real
        assert launches["n"] >= 30, f"only {launches['n']} convolutions took the Winograd kernel"
        assert mism <= 1
    finally:
        ops.set_winograd(False, min_pixels=128 * 1024)


def test_encode_is_batch_invariant(dev):
    from mcquic_amd import Compressor
    sd = R.make_state_dict(128, 2, [8192, 2048, 512], seed=0)
    model = Compressor(128, 2, [8192, 2048, 512]).eval()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    x = R.make_images(4, 128, 128).to(dev)
    all_codes = model.encode(x)
    one = model.encode(x[2:3])
    for a, b in zip(all_codes, one):
        assert torch.equal(a[2:3], b)


def test_odd_batch_matches_oracle(dev):
    _compare(dev, 8, 2, [32, 16, 8], n=9, h=128, w=128, seed=4, pix_tol=1e-4)


def test_repeatable_under_allocator_reuse(dev):
    """Branch-stream hand-offs: repeated encode/decode calls (allocator reuse across streams) stay identical."""
    from mcquic_amd import Compressor
    sd = R.make_state_dict(8, 2, [32, 16, 8], seed=5)
    model = Compressor(8, 2, [32, 16, 8]).eval()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    x = R.make_images(16, 256, 256).to(dev)
    codes0 = model.encode(x)
    rec0 = model.decode(codes0)
    for _ in range(5):
        codes = model.encode(x)
        rec = model.decode(codes)
        for a, b in zip(codes, codes0):
            assert torch.equal(a, b)
        assert torch.equal(rec, rec0)


def test_compress_decompress_round_trip(dev):
    """compress -> bytes -> decompress: codes survive the rANS streams bit-exactly, the restored image equals
    decode(codes) cropped back to the header's size (reference: compressor.py:67-77,90-112)."""
    from mcquic_amd import Compressor
    sd = R.make_state_dict(8, 2, [32, 16, 8], seed=6)
    model = Compressor(8, 2, [32, 16, 8]).eval()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    model.QuantizationParameter = "2"
    x = R.make_images(3, 200, 136).to(dev)
    codes, binaries, headers = model.compress(x)
    assert len(binaries) == 3 and all(len(b) == 3 and all(isinstance(s, bytes) for s in b) for b in binaries)
    h = headers[0]
    assert h.QuantizationParameter == "2" and h.ImageSize.height == 200 and h.ImageSize.width == 136 and h.ImageSize.channel == 3
    assert h.CodeSize.heights == [16, 8, 4] and h.CodeSize.widths == [16, 8, 4] and h.CodeSize.k == [32, 16, 8]
    want_codes = model.encode(x)
    for a, b in zip(codes, want_codes):
        assert torch.equal(a, b)
    restored = model.decompress(binaries, headers)
    assert tuple(restored.shape) == (3, 3, 200, 136)
    full = model.decode(codes)
    assert torch.equal(restored, R.aligned_crop_back(full, 200, 136))
    bits = sum(len(s) for b in binaries for s in b) * 8
    assert 0 < bits / (3 * 200 * 136) < 1.0                      # bpp of the uniform-prior streams is sane


def test_cli_compress_then_restore(dev, tmp_path):
    """`python -m mcquic_amd img.png out.mcq` then `python -m mcquic_amd out.mcq out.png` (the reference's
    `mcquic -qp 2 sample.png ./ ; mcquic sample.mcq ./` smoke, .github/workflows/test-all.yml:37-45), with a
    ragged image size so that padding and the crop-back are exercised.  Restoring twice gives identical pixels."""
    import numpy as np
    from PIL import Image
    from mcquic_amd.demo import main
    from mcquic_amd.utils import File
    rng = np.random.default_rng(0)
    img = (rng.random((150, 200, 3)) * 255).astype(np.uint8)
    src = tmp_path / "sample.png"
    Image.fromarray(img).save(src)
    assert main(["-q", str(src), str(tmp_path)]) == 0
    mcq = tmp_path / "sample.mcq"
    doc = File.deserialize(mcq.read_bytes())
    assert doc.FileHeader.ImageSize.height == 150 and doc.FileHeader.ImageSize.width == 200
    assert doc.FileHeader.CodeSize.heights == [16, 8, 4] and len(doc.Content) == 3
    out1, out2 = tmp_path / "a.png", tmp_path / "b.png"
    assert main(["-q", str(mcq), str(out1)]) == 0 and main(["-q", str(mcq), str(out2)]) == 0
    a, b = np.asarray(Image.open(out1)), np.asarray(Image.open(out2))
    assert a.shape == (150, 200, 3) and np.array_equal(a, b)


def test_sample_png_geometry_against_oracle(dev):
    """BASELINE configs[0]: one 2048x1152 image (the geometry of the reference's assets/sample.png: no padding,
    latents 128x72 / 64x36 / 32x18) through the qp=2 model; codes and pixels against the CPU oracle."""
    _compare(dev, 128, 2, [8192, 2048, 512], n=1, h=1152, w=2048, seed=0, pix_tol=1e-4)


def test_inference_mode_like_the_reference_harness(dev):
    """The reference wraps its callers in torch.inference_mode() (cli.py:60, validator.py:40,60): same results."""
    from mcquic_amd import Compressor
    sd = R.make_state_dict(8, 2, [32, 16, 8], seed=7)
    model = Compressor(8, 2, [32, 16, 8]).eval()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    x = R.make_images(2, 128, 128).to(dev)
    codes = model.encode(x)
    rec = model.decode(codes)
    with torch.inference_mode():
        codes_i = model.encode(x)
        rec_i = model.decode(codes_i)
        model2 = Compressor(8, 2, [32, 16, 8]).eval()          # parameters born as inference tensors
        model2.load_state_dict(sd, strict=True)
        model2 = model2.to(dev)
        codes_2 = model2.encode(x)
    for a, b, c in zip(codes, codes_i, codes_2):
        assert torch.equal(a, b) and torch.equal(a, c)
    assert torch.equal(rec, rec_i)


def test_graph_replay_matches_eager(dev):
    """enableGraphs(): encode / decode replayed from captured hipGraphs give the eager results, also for new data."""
    from mcquic_amd import Compressor
    sd = R.make_state_dict(8, 2, [32, 16, 8], seed=8)
    model = Compressor(8, 2, [32, 16, 8]).eval()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    xs = [R.make_images(2, 128, 256, seed=s).to(dev) for s in (1, 2, 3)]
    eager = [(model.encode(x), None) for x in xs]
    eager = [(c, model.decode(c)) for c, _ in eager]
    model.enableGraphs(True)
    for x, (codes, rec) in zip(xs, eager):
        got_codes = model.encode(x)
        got = model.decode(got_codes)
        for a, b in zip(codes, got_codes):
            assert torch.equal(a, b)
        assert torch.equal(rec, got)
    assert len(model._graphs) == 2
    model.enableGraphs(False)
    assert torch.equal(model.decode(eager[0][0]), eager[0][1])


def test_validate_helpers(dev):
    from mcquic_amd import Compressor, validate
    sd = R.make_state_dict(8, 2, [32, 16, 8], seed=9)
    model = Compressor(8, 2, [32, 16, 8]).eval()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    from oracle import metrics_ref as M
    x = R.make_images(3, 256, 256).to(dev)
    rows = validate.validate(model, x)
    assert tuple(rows.shape) == (3, 3) and torch.isfinite(rows).all() and float(rows[:, 2].min()) > 0
    a, b = R.detransform(x.cpu()), R.detransform(model.decode(model.encode(x)).cpu())
    assert torch.allclose(rows[:, 0].cpu(), M.psnr_u8(a, b), rtol=1e-13, atol=0)   # exact integer error sums
    assert torch.allclose(rows[:, 1].cpu(), M.ms_ssim_db(M.ms_ssim(a, b)).double(), atol=2e-3)
    small = validate.validate(model, R.make_images(2, 128, 128).to(dev), msssim=False)
    assert torch.isnan(small[:, 1]).all() and torch.isfinite(small[:, [0, 2]]).all()
    with pytest.raises(ValueError):
        validate.validate(model, R.make_images(1, 128, 128).to(dev))
    enc, dec = validate.speed(model, iters=2, batch=2, height=128, width=128)
    assert enc > 0 and dec > 0


@pytest.mark.parametrize("family", ["Compressor", "Neon"])
def test_empty_shard_is_empty_tensors_of_the_right_geometry(dev, family):
    """A rank that `parallel.shard_range` leaves without images (fewer images than ranks): like the reference's PyTorch
    layers, the API hands back EMPTY tensors with the geometry a non-empty batch would have -- no launch, no error -- and
    `validate` still joins the statistics gather with zero rows."""
    import mcquic_amd
    from mcquic_amd import validate
    model = (mcquic_amd.Compressor(32, 2, [64, 32, 16]) if family == "Compressor" else mcquic_amd.Neon(32, 256, [8, 4, 2, 2])).eval().to(dev)
    full = model.encode(torch.zeros((1, 3, 128, 128), device=dev))
    x = torch.empty((0, 3, 128, 128), device=dev)
    codes = model.encode(x)
    assert len(codes) == len(full)
    for c, f in zip(codes, full):
        assert c.dtype == torch.int64 and c.is_cuda and tuple(c.shape) == (0,) + tuple(f.shape[1:])
    pix = model.decode(codes)
    assert pix.is_cuda and tuple(pix.shape) == (0,) + tuple(model.decode(full).shape[1:])
    c2, binaries, headers = model.compress(x)
    assert binaries == [] and headers == [] and all(tuple(a.shape) == tuple(b.shape) for a, b in zip(c2, codes))
    rows = validate.validate(model, x, msssim=False)
    assert tuple(rows.shape) == (0, 3) and rows.dtype == torch.float64


def test_prefetch_feeds_the_same_batches(dev):
    """parallel.prefetch: pinned host batches arrive on the device in order and intact while the previous batch is in the
    kernels; codes of the prefetched batches = codes of the same batches copied up front."""
    import mcquic_amd
    from mcquic_amd import parallel
    model = mcquic_amd.Compressor(32, 2, [64, 32, 16]).eval().to(dev)
    host = [(torch.rand((2, 3, 128, 128), generator=torch.Generator().manual_seed(i)) * 2 - 1).pin_memory() for i in range(5)]
    want = [[c.cpu() for c in model.encode(h.to(dev))] for h in host]
    got = []
    for xd in parallel.prefetch(host, dev):
        assert xd.is_cuda
        got.append([c.cpu() for c in model.encode(xd)])
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert all(torch.equal(a, b) for a, b in zip(g, w))
    assert list(parallel.prefetch([], dev)) == []


def test_row_band_fallback_equals_the_single_launch(dev):
    """Images beyond ~16 MP (VERDICT r3 missing #4): `encode` / `decode` must not refuse an image whose activations exceed the
    kernels' 2 GiB-per-image addressing (the reference has no limit but memory, mcquic/modules/compressor.py:67-117).  With the
    limit lowered to 20 MB the 384x256 maps of a 768x512 image (50 MB per layer) run in row bands while the smaller maps run
    whole: the same codes, and pixels to float32 reassociation (a band is a smaller launch and may split its k-steps)."""
    from mcquic_amd import Compressor, ops
    sd = R.make_state_dict(128, 2, [8192, 2048, 512], seed=0)
    model = Compressor(128, 2, [8192, 2048, 512]).eval()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    x = R.make_images(1, 768, 512, seed=3407).to(dev)
    codes = model.encode(x)
    rec = model.decode(codes)
    prev = ops.set_slab_limit(20 << 20)
    try:
        banded_codes = model.encode(x)
        banded_rec = model.decode(codes)
    finally:
        ops.set_slab_limit(prev)
    for lv, (a, b) in enumerate(zip(codes, banded_codes)):
        assert torch.equal(a, b), f"level {lv}: {(a != b).sum().item()} codes differ between the banded and the single-launch run"
    assert float((rec - banded_rec).abs().max()) <= 2e-6


@pytest.mark.parametrize("which", [pytest.param("sample", id="sample"), pytest.param("all", id="all", marks=pytest.mark.sweep)])
def test_random_geometries_against_the_oracle(dev, which):
    """Twenty-four seeded random (batch, height, width) triples -- odd sizes, the smallest sizes reflect padding admits (a side of s
    pixels is padded to the next multiple of 128, which needs both pads < s: s >= 44), sizes around the 128-pixel padding steps,
    tall and wide strips -- through two small models against the CPU oracle: codes under the near-tie protocol, pixels within
    1e-4.  The fixed shapes above pin the tiles the benchmark runs; this sweeps the tails (partial pixel blocks, padding splits,
    levels smaller than one tile).  `-m gpu` runs a sample of six of them (three edge sizes, three free ones: the CPU oracle is
    100 s of the full sweep), `-m "gpu and sweep"` all twenty-four."""
    import random
    rng = random.Random(20260929)
    edge = [44, 45, 63, 64, 65, 127, 128, 129, 130, 200, 255, 256, 257, 300, 383, 384, 385, 511, 512, 513]
    take = set(range(24)) if which == "all" else {0, 1, 2, 10, 11, 12}
    total = 0
    for it in range(24):
        if it < 10:
            h, w = rng.choice(edge), rng.choice(edge)
        else:
            h, w = rng.randint(44, 520), rng.randint(44, 520)
        n = rng.choice([1, 1, 2, 3, 5])
        if it not in take:
            continue
        if it % 2 == 0:
            mism, _ = _compare(dev, 8, 2, [32, 16, 8], n=n, h=h, w=w, seed=100 + it, pix_tol=1e-4)
        else:
            mism, _ = _compare(dev, 16, 4, [64, 16, 8], n=n, h=h, w=w, seed=100 + it, pix_tol=1e-4)
        total += mism
    assert total <= 8, f"{total} audited near-tie mismatches over {len(take)} geometries"


def test_sizes_reflect_padding_cannot_take_are_refused_like_the_reference(dev):
    """A 300 x 1 image would need 63 + 64 reflected columns from one: F.pad raises in the reference (transforms.py:86-99); same here."""
    from mcquic_amd import Compressor
    model = Compressor(8, 2, [32, 16, 8]).eval().to(dev)
    x = R.make_images(2, 300, 1)
    with pytest.raises(RuntimeError):
        R.aligned_padding(x)
    with pytest.raises(RuntimeError):
        model.encode(x.to(dev))


def test_row_bands_at_random_limits_and_sizes(dev):
    """The row-band fallback at 16 seeded random (image size, slab limit) pairs through a 32-channel model: limits from 'only the
    largest maps are banded' down to 'a band is a handful of rows' (halo rows of 3x3 layers, stride-2 layers whose bands must start
    on even rows, pixel-shuffle layers that double them, 1x1 layers without halo): the same codes as the single-launch run and
    pixels to float32 reassociation; then `compress` -> `decompress` under a limit equals the unbanded round trip."""
    import random
    from mcquic_amd import Compressor, ops
    rng = random.Random(23)
    ks = [64, 32, 16]
    sd = R.make_state_dict(32, 2, ks, seed=4)
    model = Compressor(32, 2, ks).eval()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    banded_calls, orig = [0], ops._conv2d_banded

    def counting(*a, **k):
        banded_calls[0] += 1
        return orig(*a, **k)
    ops._conv2d_banded = counting
    for it in range(16):
        n, h, w = rng.randint(1, 3), rng.randint(44, 700), rng.randint(44, 500)
        x = R.make_images(n, h, w, seed=300 + it).to(dev)
        codes = model.encode(x)
        rec = model.decode(codes)
        # the largest activation of this image: 64 channels' worth (cin + 32) x the padded half-resolution map
        ph, pw = -(-h // 128) * 128, -(-w // 128) * 128
        largest = 64 * (ph // 2) * (pw // 2) * 4
        limit = max(int(largest * rng.choice([0.6, 0.3, 0.1, 0.03])), 40 * 64 * pw * 4)      # (never below ~40 rows of the widest map:
        # a limit under one row's worth is refused loudly, ops._band_rows)
        prev = ops.set_slab_limit(limit)
        try:
            banded_codes = model.encode(x)
            banded_rec = model.decode(codes)
            if it % 4 == 0:
                _, binaries, headers = model.compress(x)
                back = model.decompress(binaries, headers)
        finally:
            ops.set_slab_limit(prev)
        what = f"#{it} n{n} {h}x{w} limit {limit}"
        for lv, (a, b) in enumerate(zip(codes, banded_codes)):
            assert torch.equal(a, b), f"{what} level {lv}: {(a != b).sum().item()} codes differ"
        assert float((rec - banded_rec).abs().max()) <= 2e-6, what
        if it % 4 == 0:
            top, left = (rec.shape[-2] - h) // 2, (rec.shape[-1] - w) // 2
            assert float((back - rec[..., top: top + h, left: left + w]).abs().max()) <= 2e-6, what
    ops._conv2d_banded = orig
    assert banded_calls[0] >= 100, f"only {banded_calls[0]} banded launches: the limits did not bite"
