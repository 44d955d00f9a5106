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
