"""Codebook maintenance: `reAssignCodebook` against vectors captured from the reference
(mcquic/modules/quantizer.py:111-136 with torch.randperm returning a recorded permutation, tests/golden/make_golden.py
F9), and the packed-operand cache noticing the in-place codebook update."""
import os

import numpy as np
import pytest
import torch

from oracle import mcquic_ref as R

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _check_case(z, ci, device):
    from mcquic_amd.modules import quantizer as MQ
    m, k, d, dead_frac, seed = R.REASSIGN_CASES[ci]
    assert tuple(int(v) for v in z["cases"][ci]) == (m, k, d, seed)
    cb, freq, perms = R.reassign_case(m, k, d, dead_frac, seed)
    np.testing.assert_array_equal(freq.numpy(), z[f"freq_{ci}"])
    for g in range(m):
        np.testing.assert_array_equal(perms[g].numpy(), z[f"perm_{ci}_{g}"])
    prio = R.reassign_priority(freq, perms)
    q = MQ._multiCodebookQuantization(torch.nn.Parameter(cb.clone().to(device)), MQ._CodebookCache())
    version = q._codebook._version
    changed = q.reAssignCodebook(freq.clone(), priority=prio.to(device)).cpu().view(m, k)
    assert q._codebook._version > version                     # the packed-operand cache keys on it
    got = q._codebook.detach().cpu()
    want, want_changed = torch.from_numpy(z[f"new_codebook_{ci}"]), torch.from_numpy(z[f"changed_{ci}"]).view(m, k)
    defined = R.reassign_defined_mask(freq, prio)
    assert torch.equal(got[defined], want[defined])
    assert torch.equal(changed[defined], want_changed[defined])
    # slots the reference leaves implementation-defined (unstable argsort over tied zeros): a refilled dead codeword, here too
    for g, s in torch.nonzero(~defined).tolist():
        dead = torch.nonzero(freq[g] < 1e-6).flatten()
        assert any(torch.equal(got[g, s], cb[g, t]) for t in dead.tolist())
    return int(defined.sum()), int((~defined).sum())


@pytest.mark.parametrize("ci", range(len(R.REASSIGN_CASES)))
def test_reassign_matches_reference_cpu(ci):
    _check_case(np.load(os.path.join(G, "f9_reassign.npz")), ci, torch.device("cpu"))


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(len(R.REASSIGN_CASES)))
def test_reassign_matches_reference_gpu(dev, ci):
    _check_case(np.load(os.path.join(G, "f9_reassign.npz")), ci, dev)


def test_reassign_draws_its_own_priorities():
    from mcquic_amd.modules import quantizer as MQ
    cb, freq, _ = R.reassign_case(2, 32, 4, [0.9, 0.1], 7)
    q = MQ._multiCodebookQuantization(torch.nn.Parameter(cb.clone()), MQ._CodebookCache())
    changed = q.reAssignCodebook(freq).view(2, 32)
    dead = freq < 1e-6
    assert not changed[~dead].any()                             # live codewords never move
    assert int((changed & dead)[0].sum()) <= 16                 # a crowded group refills at most k // 2
    assert torch.equal(q._codebook.detach()[~dead], cb[~dead])


@pytest.mark.gpu
def test_encode_follows_a_reassigned_codebook(dev):
    """after reAssignCodebook the packed codebook must be rebuilt -- encode / decode on the GPU equal
    the oracle on the NEW codebooks, and differ from the results before the update."""
    from mcquic_amd import Compressor
    ch, m, ks = 8, 2, [32, 16, 8]
    sd = R.make_state_dict(ch, m, ks, seed=1)
    g = torch.Generator().manual_seed(5)
    for lv, k in enumerate(ks):                                # frequencies with dead codewords on every level
        f = torch.rand((m, k), generator=g) * (torch.rand((m, k), generator=g) > 0.4)
        sd[f"_quantizer._entropyCoder._freqEMA.{lv}"] = f / f.sum(-1, keepdim=True)
    model = Compressor(ch, m, ks).eval()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    x = R.make_images(2, 128, 128)
    before = [c.cpu() for c in model.encode(x.to(dev))]
    rec_before = model.decode([c.to(dev) for c in before]).cpu()
    share = float(model.reAssignCodebook())
    assert 0.0 < share < 1.0
    sd2 = {key: v.detach().cpu() for key, v in model.state_dict().items()}
    assert any(not torch.equal(sd2[key], sd[key]) for key in sd if key.endswith("_quantizer._codebook"))
    want = R.encode(sd2, x)
    got = [c.cpu() for c in model.encode(x.to(dev))]
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    rec = model.decode([c.to(dev) for c in before]).cpu()       # same codes, new codebooks
    assert float((rec - R.decode(sd2, before)).abs().max()) <= 1e-4
    assert float((rec - rec_before).abs().max()) > 1e-3
