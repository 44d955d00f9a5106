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
