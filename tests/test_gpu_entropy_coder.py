"""The host-side rANS coder's golden-vector tests again, on the GPU box (`-m gpu` run): the library the driver loads
there is the one whose byte streams must equal the reference's (tests/golden/f8_rans.npz), and `Compressor.compress`
feeds it from device memory."""
import pytest
import torch

import test_entropy_coder as T

pytestmark = pytest.mark.gpu


def test_cdf_and_bytes_match_reference_vectors_on_the_gpu_box(dev):
    T.test_cdf_and_bytes_match_reference_vectors()
    T.test_bypass_symbols_match_reference_vectors()


def test_batched_coder_on_the_gpu_box(dev):
    T.test_batched_coder_equals_one_stream_at_a_time()
    T.test_out_of_range_symbols_are_errors_not_overreads()
    T.test_decompress_rejects_hostile_headers()


def test_compress_from_device_codes_equals_host_codes(dev):
    """The batched path (pinned D2H per level + one C call) gives the byte streams of coding each image alone."""
    from mcquic_amd.modules import entropyCoder as E
    m, ks = 2, [8192, 2048, 512]
    coder = E.EntropyCoder(m, ks).to(dev)
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for f in coder._freqEMA:
            f.copy_((torch.rand(f.shape, generator=g) ** 2 + 1e-4).to(dev))
    codes = [torch.randint(0, k, (5, m, 48 >> lv, 32 >> lv), generator=g) for lv, k in enumerate(ks)]
    binaries, sizes = coder.compress([c.to(dev) for c in codes])
    for i in range(5):
        alone, _ = coder.compress([c[i:i + 1] for c in codes])
        assert alone[0] == binaries[i]
    for a, b in zip(codes, coder.decompress(binaries, sizes)):
        assert b.is_cuda and torch.equal(a, b.cpu())


def test_level_wise_coder_jobs_give_the_all_at_once_bytes(dev):
    """Round 6: the coder taken level by level (EntropyCoder.beginCompress / beginDecompress: side-stream copies, one host thread,
    smallest level decoded first) -- same byte streams, same CodeSize, same codes as the all-at-once calls, for several jobs in
    flight one after the other and with the levels submitted while the device is busy."""
    from mcquic_amd.modules import entropyCoder as E
    m, ks = 2, [8192, 2048, 512]
    coder = E.EntropyCoder(m, ks).to(dev)
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for f in coder._freqEMA:
            f.copy_((torch.rand(f.shape, generator=g) ** 2 + 1e-4).to(dev))
    for n in (1, 5, 10):
        codes = [torch.randint(0, k, (n, m, 48 >> lv, 32 >> lv), generator=g).to(dev) for lv, k in enumerate(ks)]
        want, sizes = coder.compress(codes)
        busy = torch.randn(2048, 2048, device=dev)
        job = coder.beginCompress(len(codes))
        for lv, c in enumerate(codes):
            busy = busy @ busy * 1e-3                               # (the device has work queued in front of every copy)
            job.submit(lv, c)
        got, got_sizes = job.finish()
        assert got == want
        assert [(s.m, s.heights, s.widths, s.k) for s in got_sizes] == [(s.m, s.heights, s.widths, s.k) for s in sizes]
        dj = coder.beginDecompress(got, got_sizes)
        for lv in reversed(range(len(codes))):
            back = dj.level(lv)
            assert back.is_cuda and back.dtype == torch.int64 and torch.equal(back, codes[lv])
    with pytest.raises(RuntimeError):
        coder.beginCompress(2)
    job = coder.beginCompress(3)
    job.submit(0, torch.zeros((1, m, 4, 4), dtype=torch.int64, device=dev))
    with pytest.raises(RuntimeError):
        job.finish()                                                # (levels missing)
    bad = coder.beginCompress(3)
    for lv, k in enumerate(ks):
        bad.submit(lv, torch.full((1, m, 4, 4), k, dtype=torch.int64, device=dev))    # a code index outside [0, k): the host thread's error arrives at finish()
    with pytest.raises(RuntimeError):
        bad.finish()


def test_compress_decompress_overlap_equals_the_sequential_path(dev, monkeypatch):
    """Compressor.compress / decompress with the level-wise coder against MCQUIC_AMD_CODER_OVERLAP=0: codes, bytes, headers and the
    restored batch are identical."""
    from mcquic_amd import Compressor
    from mcquic_amd.modules import entropyCoder as E
    from oracle import mcquic_ref as R
    sd = R.make_state_dict(8, 2, [32, 16, 8], seed=1)
    model = Compressor(8, 2, [32, 16, 8]).eval()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    x = R.make_images(3, 200, 136).to(dev)
    assert E.CODER_OVERLAP
    codes, binaries, headers = model.compress(x)
    out = model.decompress(binaries, headers)
    monkeypatch.setattr(E, "CODER_OVERLAP", False)
    codes0, binaries0, headers0 = model.compress(x)
    out0 = model.decompress(binaries0, headers0)
    assert binaries == binaries0 and all(torch.equal(a, b) for a, b in zip(codes, codes0))
    assert [h.CodeSize.heights for h in headers] == [h.CodeSize.heights for h in headers0]
    assert torch.equal(out, out0) and tuple(out.shape) == (3, 3, 200, 136)
