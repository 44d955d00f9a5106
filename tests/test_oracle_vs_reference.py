"""CPU, build container only: the oracle against the REAL reference imported from /root/reference (skipped on
machines where the reference tree does not exist, e.g. the GPU box)."""
import pytest
import torch

from oracle import mcquic_ref as R
from oracle import ref_harness

pytestmark = pytest.mark.skipif(not ref_harness.available(), reason="reference tree not present")


def test_state_dict_layout_is_the_references():
    sd = R.make_state_dict(8, 2, [32, 16, 8], seed=3)
    ref = ref_harness.reference_compressor(8, 2, [32, 16, 8]).state_dict()
    assert set(sd) == set(ref) and len(ref) == 718
    for k in ref:
        assert tuple(sd[k].shape) == tuple(ref[k].shape), k


def test_encode_decode_bit_equal_to_reference():
    sd = R.make_state_dict(8, 2, [32, 16, 8], seed=3)
    model = ref_harness.reference_compressor(8, 2, [32, 16, 8], sd)
    x = R.make_images(2, 130, 250, seed=9)
    with torch.inference_mode():
        rc = model.encode(x)
        rd = model.decode(rc)
    oc = R.encode(sd, x)
    for a, b in zip(rc, oc):
        assert torch.equal(a, b)
    assert torch.equal(rd, R.decode(sd, oc))


def test_hip_module_tree_matches_reference_keys():
    from mcquic_amd import Compressor
    ref = ref_harness.reference_compressor(8, 2, [32, 16, 8]).state_dict()
    mine = Compressor(8, 2, [32, 16, 8]).state_dict()
    assert list(mine.keys()) == list(ref.keys())
    for k in ref:
        assert tuple(mine[k].shape) == tuple(ref[k].shape), k


def test_metrics_oracle_against_reference_handlers():
    """oracle/metrics_ref.py against the reference's MsSSIM / PSNR / IdealBPP handlers, live (fresh seeds)."""
    from oracle import metrics_ref as M
    H = ref_harness.load_validate_handlers()
    x, y = M.make_u8_pair(901, 2, 211, 180)
    msssim, psnr = H["MsSSIM"](), H["PSNR"]()
    want_db = torch.tensor(msssim.handle(images=x, restored=y))
    assert torch.allclose(M.ms_ssim_db(M.ms_ssim(x, y)), want_db, atol=2e-3)
    assert torch.allclose(M.psnr_u8(x, y), torch.tensor(psnr.handle(images=x, restored=y), dtype=torch.float64),
                          rtol=1e-14, atol=0)
    ks = list(M.CODE_BATCH_KS)
    handler = H["IdealBPP"]([2, 2, 2], ks)
    hist = [torch.zeros(2, k) for k in ks]
    count = [torch.zeros(2) for _ in ks]
    for codes in M.make_code_batches(seed=5, batches=3):
        handler(codes=codes, images=torch.zeros(3, 3, 768, 512, dtype=torch.uint8))
        for lv, (c, k) in enumerate(zip(codes, ks)):
            for g in range(2):
                hist[lv][g] += torch.bincount(c[:, g].flatten(), minlength=k)
                count[lv][g] += c[:, g].numel()
    assert abs(M.ideal_bpp(hist, count, 3 * 3 * 768 * 512) - handler.Result) <= 1e-6 * handler.Result


@pytest.mark.parametrize("dense", [False, True], ids=["plain", "denseNorm"])
def test_neon_oracle_bit_equal_to_reference(dense):
    """oracle/neon_ref.py against the reference's Neon / ResidualBackwardQuantizer, live: state_dict layout, encode,
    decode, residual_backward, residual_forward; with denseNorm=True (nn.GroupNorm(groups, C) in place of the ResidualBlocks'
    second activation, mcquic/nn/blocks.py:179-200; 32 groups need a width that is a multiple of 32)."""
    from oracle import neon_ref as N
    C = ref_harness.load()
    ch, k, size = (32 if dense else 16), 64, [4, 2, 2]
    sd = N.make_state_dict(ch, k, size, seed=11, denseNorm=dense)
    model = C.Neon(ch, k, size, dense).eval()
    ref = model.state_dict()
    assert set(sd) == set(ref)
    for key in ref:
        assert tuple(sd[key].shape) == tuple(ref[key].shape), key
    model.load_state_dict(sd, strict=True)
    x = R.make_images(2, 64, 64, seed=12)
    with torch.inference_mode():
        rc = model.encode(x)
        rd = model.decode(rc)
        rb = model.residual_backward(rc[1], 2)
        rf = model.residual_forward(rc[1], model.residual_forward(rc[0], None, 0), 1)
    oc = N.encode(sd, x)
    assert all(torch.equal(a, b) for a, b in zip(rc, oc))
    assert torch.equal(rd, N.decode(sd, oc))
    assert torch.equal(rb, N.residual_backward(sd, oc[1], 2))
    assert torch.equal(rf, N.residual_forward(sd, oc[1], N.residual_forward(sd, oc[0], None, 0), 1))


def test_hip_neon_dense_norm_module_tree_matches_reference_keys():
    from mcquic_amd import Neon
    C = ref_harness.load()
    ref = C.Neon(32, 256, [8, 4, 2, 2], True).state_dict()
    mine = Neon(32, 256, [8, 4, 2, 2], True).state_dict()
    assert list(mine) == list(ref)
    assert all(tuple(mine[k].shape) == tuple(ref[k].shape) for k in ref)


def test_hip_neon_module_tree_matches_reference_keys():
    from mcquic_amd import Neon
    C = ref_harness.load()
    ref = C.Neon(32, 256, [8, 4, 2, 2]).state_dict()
    mine = Neon(32, 256, [8, 4, 2, 2]).state_dict()
    assert list(mine.keys()) == list(ref.keys()) and len(ref) == 819
    for key in ref:
        assert tuple(mine[key].shape) == tuple(ref[key].shape), key
