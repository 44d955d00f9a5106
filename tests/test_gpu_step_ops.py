"""The per-step bookkeeping kernels of the training quantizer (csrc/step_ops.hip) and the small fused forms that replaced
engine- / ATen-issued launches in the captured training step, each against the tensor ops the reference spells them with
(mcquic/modules/quantizer.py:194-200, mcquic/modules/entropyCoder.py:28-44, mcquic/nn/base.py:17-29)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _freqs(ms, ks, seed, dead=0.3):
    g = torch.Generator().manual_seed(seed)
    out = []
    for m, k in zip(ms, ks):
        f = torch.rand((m, k), generator=g)
        f = torch.where(torch.rand((m, k), generator=g) < dead, torch.zeros_like(f), f)      # unused codewords: frequency 0
        out.append(f / f.sum(-1, keepdim=True))
    return out


@pytest.mark.parametrize("ms,ks", [([2, 2, 2], [8192, 2048, 512]), ([1, 1, 1, 1], [4096] * 4), ([3], [1000]), ([12, 12], [8192, 37])])
def test_step_prologue_matches_the_reference_ops(dev, ms, ks):
    """Exponent of `_randomDrop` bit for bit (quantizer.py:196-198 evaluated with torch on the CPU), generator snapshots =
    {seed, offset + l} with the state moved past them, count buffer zeroed."""
    from mcquic_amd import ops
    freqs = _freqs(ms, ks, 7)
    ops.seed_rng(99, dev)
    first = ops.rng_snapshot(dev)                                   # (moves the offset to 1)
    counts = torch.full((sum(m * k for m, k in zip(ms, ks)),), 7, dtype=torch.int64, device=dev)
    st = ops.vq_step_prologue([f.to(dev) for f in freqs], 1e-6, True, counts)
    for lv, (f, k) in enumerate(zip(freqs, ks)):
        bits = math.log2(k)
        usage = (f > 1e-6).float().mean().clamp(0., 1.)
        want = -(bits - 1) * (usage ** 2) + bits
        e, rng, cnt = st.level(lv)
        assert float(e.cpu()) == float(want), (lv, float(e.cpu()), float(want))
        assert int(rng[0]) == 99 and int(rng[1]) == int(first[1]) + 1 + lv
        assert cnt.numel() == ms[lv] * k
    assert int(st.counts.abs().sum()) == 0
    nxt = ops.rng_snapshot(dev)
    assert int(nxt[1]) == int(first[1]) + 1 + len(ks)
    # without a generator / a count buffer
    st2 = ops.vq_step_prologue([f.to(dev) for f in freqs], 1e-6, False)
    assert st2.snaps is None and st2.counts is None and torch.equal(st2.exponents, st.exponents)


@pytest.mark.parametrize("m,k,hw", [(2, 8192, 16), (2, 2048, 8), (2, 512, 4), (1, 20000, 3), (3, 40, 5)])
def test_sampling_kernel_counts_its_codes(dev, m, k, hw):
    """`counts` of mcq_vq_gumbel_sample_f32 = the histogram of the codes it returns (entropyCoder.py:33-35), every kernel variant."""
    from mcquic_amd import ops, parallel
    g = torch.Generator().manual_seed(k)
    n = 3
    logits = torch.randn((n, m, hw, hw, k), generator=g).to(dev)
    freq = _freqs([m], [k], 3)[0].to(dev)
    ops.seed_rng(5, dev)
    st = ops.vq_step_prologue([freq], 1e-6, True, torch.empty(m * k, dtype=torch.int64, device=dev))
    e, rng, cnt = st.level(0)
    codes, index, hot = ops.vq_gumbel_sample(logits, None, None, freq, e, rng, cnt)
    want = parallel.local_code_counts([codes], [k])
    assert torch.equal(cnt, want) and int(cnt.sum()) == n * m * hw * hw


@pytest.mark.parametrize("ms,ks", [([2, 2, 2], [8192, 2048, 512]), ([1, 1], [4096, 4096]), ([3, 5], [100, 7])])
def test_freq_ema_update_is_the_reference_update(dev, ms, ks):
    """(1 - ema) * count / total + ema * freq, bit for bit the tensor-op form of entropyCoder.py:37-43 (float32, CPU)."""
    from mcquic_amd import ops
    g = torch.Generator().manual_seed(11)
    freqs = _freqs(ms, ks, 13, dead=0.0)
    counts = [torch.randint(0, 50, (m, k), generator=g) for m, k in zip(ms, ks)]
    for ema in (0.9, 0.998):
        want = []
        for f, c in zip(freqs, counts):
            total = c.to(torch.float32)
            normalized = total / total.sum(-1, keepdim=True)
            want.append((1 - ema) * normalized + ema * f)
        got = [f.clone().to(dev) for f in freqs]
        ops.freq_ema_update_(got, torch.cat([c.reshape(-1) for c in counts]).to(dev), ema)
        for a, b in zip(got, want):
            assert torch.equal(a.cpu(), b)


def test_entropy_coder_forward_from_counted_codes(dev):
    """EntropyCoder.forward fed by the step's count buffer = the same call counting from the codes (and = the CPU module)."""
    from mcquic_amd import ops
    from mcquic_amd.modules.entropyCoder import EntropyCoder
    ks = [64, 32, 16]
    g = torch.Generator().manual_seed(2)
    codes = [torch.randint(0, k, (5, 2, s, s), generator=g) for k, s in zip(ks, (8, 4, 2))]
    cpu, a, b = EntropyCoder(2, ks), EntropyCoder(2, ks).to(dev), EntropyCoder(2, ks).to(dev)
    cpu(codes)
    a([c.to(dev) for c in codes])
    buf = b.countBuffer(dev)
    from mcquic_amd import parallel
    buf.copy_(parallel.local_code_counts([c.to(dev) for c in codes], ks))
    b([c.to(dev) for c in codes], counts=buf)
    for x, y, z in zip(cpu._freqEMA, a._freqEMA, b._freqEMA):
        assert torch.equal(x, y.cpu()) and torch.equal(x, z.cpu())
    assert b.countBuffer(dev).data_ptr() == buf.data_ptr()          # allocated once: a captured step keeps its address


def test_temperature_grad_applies_lower_bound_rule(dev):
    from mcquic_amd import ops
    g = torch.Generator().manual_seed(4)
    n, m, h, w = 8, 4, 16, 16
    dtrow = torch.randn((n, m, h, w), generator=g)
    dtrow[:, 2] = dtrow[:, 2].abs()                                # group 2: positive sum, temperature below the bound -> blocked
    dtrow[:, 3] = -dtrow[:, 3].abs()                               # group 3: negative sum, temperature below the bound -> passes
    temp = torch.tensor([1.0, 0.5, 1e-9, 1e-9]).reshape(m, 1, 1, 1)
    bound = 1e-6
    got = ops.vq_temperature_grad(dtrow.to(dev), temp.to(dev), bound).cpu()
    s = dtrow.double().sum((0, 2, 3))
    mask = (temp.reshape(-1) >= bound) | (s < 0)
    want = (mask.double() * s).reshape(m, 1, 1, 1)
    assert got.shape == temp.shape
    assert float(got[2]) == 0.0 and float(got[3]) < 0
    assert float((got.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())


def test_small_fused_forms(dev):
    """nonneg_reparam_bwd2 = two nonneg_reparam_bwd; silu_bwd(x, dy, other) = silu_bwd(x, dy) + other; axpby / soft dequantisation
    with their SiLU twins = silu of the result -- bit for bit against the separate launches."""
    from mcquic_amd import ops
    g = torch.Generator().manual_seed(8)
    p0, d0 = torch.randn(128, generator=g).to(dev), torch.randn(128, generator=g).to(dev)
    p1, d1 = torch.randn((128, 128), generator=g).to(dev), torch.randn((128, 128), generator=g).to(dev)
    a, b = ops.nonneg_reparam_bwd2(p0, d0, 0.1, p1, d1, 0.2)
    assert torch.equal(a, ops.nonneg_reparam_bwd(p0, d0, 0.1)) and torch.equal(b, ops.nonneg_reparam_bwd(p1, d1, 0.2))
    x, dy, other = [torch.randn((2, 16, 12, 12), generator=g).to(dev) for _ in range(3)]
    assert torch.equal(ops.silu_bwd(x, dy, other), ops.silu_bwd(x, dy) + other)
    r = ops.axpby(x, dy, 1.0, -1.0, dual_silu=True)
    assert torch.equal(r, x - dy) and torch.equal(ops.silu_twin(r), ops.silu(r))
    cb = ops.PackedCodebook((torch.randn((2, 64, 8), generator=g) * 0.3).to(dev))
    index = torch.randint(0, 64, (2, 2, 12, 12), generator=g).to(dev)
    hot = torch.rand((2, 2, 12, 12), generator=g).to(dev)
    q = ops.vq_dequant_soft(index, hot, cb, dual_silu=True)
    assert torch.equal(q, ops.vq_dequant_soft(index, hot, cb)) and torch.equal(ops.silu_twin(q), ops.silu(q))


# ATen operators that launch nothing: allocation, views, metadata
_NO_LAUNCH = ("empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "view", "_unsafe_view", "reshape", "_reshape_alias", "detach", "alias", "as_strided",
              "expand", "select", "slice", "t", "transpose", "permute", "unsqueeze", "squeeze", "view_as", "is_same_size", "sym_size",
              "sym_stride", "sym_numel", "sym_storage_offset", "stride", "size", "numel", "dim", "is_contiguous", "lift_fresh", "_to_copy_meta",
              "unbind", "split", "chunk", "narrow", "contiguous", "result_type", "is_pinned", "set_", "record_stream", "is_nonzero_meta")


@pytest.mark.parametrize("kind", ["compressor", "neon", "neon_dense_norm"])
def test_training_step_issues_no_aten_kernels(dev, kind):
    """One training step (forward + backward) of a Compressor dispatches NO ATen operator that launches a kernel: every launch is
    this library's (VERDICT r4 counted 98 `at::native::*` launches per replay of the captured qp=2 step: the soft assignment's
    bookkeeping, the frequency EMA, LowerBound's rule, the autograd engine's own gradient sums and the root gradient's fill).
    Read at the dispatcher (TorchDispatchMode: every ATen call of the step, the autograd thread's included), which cannot miss a
    launch the way a tracer can; a profiler trace of a replay of the captured step backs it where the tracer delivers one."""
    from torch.utils._python_dispatch import TorchDispatchMode
    from mcquic_amd import Compressor
    from mcquic_amd.autograd import backward, mse_loss
    from mcquic_amd.nn import blocks

    class Log(TorchDispatchMode):
        def __init__(self):
            super().__init__()
            self.ops = []

        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            self.ops.append(func.overloadpacket.__name__ if hasattr(func, "overloadpacket") else str(func))
            return func(*args, **(kwargs or {}))

    streams = blocks._BRANCH_STREAMS
    blocks._BRANCH_STREAMS = False
    try:
        torch.manual_seed(3407)
        if kind == "compressor":
            model = Compressor(16, 2, [64, 32, 16]).to(dev).train()
        else:
            from mcquic_amd import Neon
            model = Neon(32, 256, [8, 4, 2, 2], kind == "neon_dense_norm").to(dev).train()
        x = (torch.rand((2, 3, 128, 128), generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)

        def step():
            for p in model.parameters():
                p.grad = None
            xHat = model(x)[0]
            loss = mse_loss(xHat, x)
            backward(loss)
            return loss
        for _ in range(2):
            step()                                                  # (packs, caches, the cached root gradient)
        torch.cuda.synchronize()
        with Log() as log:
            step()
        torch.cuda.synchronize()
    finally:
        blocks._BRANCH_STREAMS = streams
    launching = sorted({o for o in log.ops if o not in _NO_LAUNCH})
    assert len(log.ops) > 200, len(log.ops)                         # (the mode does see the step: ~1 allocation per launch)
    assert not launching, f"ATen operators with kernels of their own inside the training step: {launching}"


def test_training_step_replay_trace_holds_only_our_kernels(dev):
    """The same statement from a profiler trace of one replay of the captured step, taken in a fresh process
    (tests/_replay_names_worker.py); skipped when the tracer records (almost) nothing -- it does that now and then on this stack."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    run = subprocess.run([sys.executable, os.path.join(here, "_replay_names_worker.py")], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    names = json.loads(run.stdout.strip().splitlines()[-1])
    if len(names) < 100:
        pytest.skip(f"the tracer delivered {len(names)} device activities for a replay of ~600 kernels")
    foreign = [n for n in names if "at::native" in n or "emset" in n or "fillBuffer" in n or "elementwise_kernel" in n or "copyBuffer" in n]
    assert not foreign, (len(foreign), sorted(set(foreign))[:8])
    ours = [n for n in names if "conv_mfma" in n or "conv_t16" in n or "vq_" in n]
    assert len(ours) > 50


@pytest.mark.parametrize("shape", [(8, 128, 16, 16), (2, 128, 64, 64), (3, 32, 5, 7), (1, 192, 4, 4)])
def test_gate_backward_from_the_recomputed_1x1_launch(dev, shape):
    """MCQ_CONV_GATE_BWD: (d a, d s) from the epilogue of the recomputed s = conv1x1(b) against the stand-alone kernels on a stored s
    (mcq_gate_bwd_f32 after mcq_conv2d_f32), and the forward with the gate as the 1x1 launch's epilogue against conv + mcq_gate_f32."""
    from mcquic_amd import ops
    n, c, h, w = shape
    g = torch.Generator().manual_seed(c + h)
    b, a, x, dout = [torch.randn(shape, generator=g).to(dev) for _ in range(4)]
    wt = (torch.randn((c, c, 1, 1), generator=g) / c ** 0.5).to(dev)
    bias = torch.randn(c, generator=g).to(dev)
    pk = ops.PackedConv(wt, bias)
    s = ops.conv2d(b, pk)
    want_da, want_ds = ops.gate_bwd(a, s, dout)
    (da, ds), = ops.conv2d_gate_bwd([b], [pk], [a], [dout])
    for got, want, name in ((da, want_da, "d a"), (ds, want_ds, "d s")):
        err = float((got - want).abs().max()) / max(float(want.abs().max()), 1e-12)
        assert err <= 2e-6, (name, err)
    out = ops.conv2d(b, pk, gate_mul=a, gate_id=x, dual_silu=True)
    ref = ops.gate(a, s, x)
    assert float((out - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    assert torch.equal(ops.silu_twin(out), ops.silu(out))
    # two gates in one launch = the two single launches
    b2, a2, d2 = [torch.randn(shape, generator=g).to(dev) for _ in range(3)]
    both = ops.conv2d_gate_bwd([b, b2], [pk, pk], [a, a2], [dout, d2])
    (da2, ds2), = ops.conv2d_gate_bwd([b2], [pk], [a2], [d2])
    assert torch.equal(both[0][0], da) and torch.equal(both[0][1], ds) and torch.equal(both[1][0], da2) and torch.equal(both[1][1], ds2)


@pytest.mark.parametrize("shape", [(8, 128, 32, 32), (2, 128, 64, 64), (3, 32, 9, 7), (1, 8, 4, 4), (2, 192, 16, 12)])
def test_pixel_shuffle_store_with_side_tensors(dev, shape):
    """MCQ_CONV_SHUFFLE2 | DSILU_MUL | RESIDUAL (the input-gradient launch of a strided block's first convolution: * silu'(x) + d_skip
    through the shuffle) against the shuffle store followed by the stand-alone kernels; each side operation alone as well."""
    from mcquic_amd import ops
    n, c, h, w = shape
    g = torch.Generator().manual_seed(c * h)
    x = torch.randn((n, c, h, w), generator=g).to(dev)
    wt = (torch.randn((4 * c, c, 3, 3), generator=g) / (3 * c ** 0.5)).to(dev)
    pk = ops.PackedConv(wt, None)
    m, r = [torch.randn((n, c, 2 * h, 2 * w), generator=g).to(dev) for _ in range(2)]
    plain = ops.conv2d(x, pk, shuffle2=True)
    for kw, want in ((dict(dsilu_mul=m, res=r), ops.silu_bwd(m, plain, r)), (dict(dsilu_mul=m), ops.silu_bwd(m, plain)), (dict(res=r), plain + r)):
        got = ops.conv2d(x, pk, shuffle2=True, **kw)
        err = float((got - want).abs().max()) / max(float(want.abs().max()), 1e-12)
        assert err <= 2e-6, (sorted(kw), err)


@pytest.mark.parametrize("up", [False, True])
@pytest.mark.parametrize("shape", [(2, 32, 16, 16), (1, 128, 8, 8), (2, 8, 7, 9)])
def test_scale_block_node_equals_op_by_op_graph(dev, up, shape):
    """ScaleBlockFn (strided / shuffle block as one autograd node, gradients meeting in a conv epilogue) against the op-by-op
    autograd graph of the same block: outputs bit-equal, every gradient within 4e-6 of the largest entry."""
    from mcquic_amd.nn import blocks
    n, c, h, w = shape
    torch.manual_seed(c + h)
    blk = (blocks.ResidualBlockShuffle(c, c) if up else blocks.ResidualBlockWithStride(c, c)).to(dev).train()
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn((n, c, h, w), generator=g).to(dev)
    res = {}
    for mode in (True, False):
        blocks._BLOCK_NODES = mode
        try:
            for p in blk.parameters():
                p.grad = None
            x = x0.clone().requires_grad_()
            y = blk(x * 1.0)
            gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5)).to(dev)
            (y * gy).sum().backward()
            res[mode] = (y.detach(), x.grad.clone(), {k: p.grad.clone() for k, p in blk.named_parameters() if p.grad is not None})
        finally:
            blocks._BLOCK_NODES = True
    assert torch.equal(res[True][0], res[False][0])
    pairs = [("dx", res[True][1], res[False][1])] + [(k, res[True][2][k], res[False][2][k]) for k in res[False][2]]
    assert set(res[True][2]) == set(res[False][2])
    for name, a, b in pairs:
        err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)
        assert err <= 4e-6, (name, err)


@pytest.mark.parametrize("shape,groups", [((2, 32, 192, 192), 32), ((1, 8, 128, 128), 1), ((2, 4, 97, 101), 2), ((3, 32, 512, 64), 32),
                                          ((2, 32, 16, 16), 32), ((2, 8, 8, 8), 1), ((4, 32, 64, 64), 32), ((2, 32, 17, 15), 4), ((1, 8, 32, 32), 1),
                                          ((8, 32, 512, 512), 32), ((1, 8, 512, 640), 1), ((2, 16, 300, 211), 4)])
def test_group_norm_large_runs(dev, shape, groups):
    """nn.GroupNorm forward (+ SiLU twin) and backward on runs long enough for the chunked kernels (many workgroups per
    (image, group): Neon's GroupNorm(32, 32) on 512 x 512 maps) and, for comparison, on the short runs the one-workgroup kernels
    keep -- against torch's CPU GroupNorm in float64."""
    from mcquic_amd import ops
    n, c, h, w = shape
    g = torch.Generator().manual_seed(c * h + groups)
    x = (torch.randn(shape, generator=g) * 1.5 + 0.3)
    gamma, beta = torch.randn(c, generator=g), torch.randn(c, generator=g)
    dy = torch.randn(shape, generator=g)
    xr = x.double().requires_grad_()
    gr, br = gamma.double().requires_grad_(), beta.double().requires_grad_()
    ref = torch.nn.functional.group_norm(xr, groups, gr, br, 1e-5)
    ref.backward(dy.double())
    y, mean, rstd = ops.group_norm(x.to(dev), gamma.to(dev), beta.to(dev), groups, 1e-5, dual_silu=True, want_stats=True)
    assert float((y.cpu().double() - ref.detach()).abs().max()) <= 5e-6 * max(1.0, float(ref.detach().abs().max()))
    assert torch.equal(ops.silu_twin(y), ops.silu(y))
    dx, dw, db = ops.group_norm_bwd(x.to(dev), dy.to(dev), gamma.to(dev), mean, rstd, groups)
    for got, want, name in ((dx, xr.grad, "dx"), (dw, gr.grad, "dgamma"), (db, br.grad, "dbeta")):
        err = float((got.cpu().double() - want).abs().max()) / max(float(want.abs().max()), 1e-12)
        assert err <= 2e-5, (name, err)
    # deterministic: a second run gives the same bits
    y2 = ops.group_norm(x.to(dev), gamma.to(dev), beta.to(dev), groups, 1e-5)
    assert torch.equal(y, y2)


def test_group_norm_one_launch_equals_two_launches(dev):
    """Round 6: statistics and normalisation of the chunked GroupNorm in ONE launch per direction (workgroups of a run meet on a
    device-scope counter and finish from registers) -- bit for bit the two-launch form's outputs (MCQUIC_AMD_GN_FUSED=0, read once
    per process: the other form runs in a child), forward with statistics and SiLU twin, backward with parameter gradients.  The
    last shape's runs are longer than the one-launch form takes (320 workgroups): both processes run the same kernels there."""
    import hashlib
    import os
    import subprocess
    import sys
    code = """
import hashlib, sys, torch
from mcquic_amd import ops
dev = torch.device('cuda:0')
for shape, groups in (((8, 32, 512, 512), 32), ((2, 32, 192, 192), 32), ((1, 8, 128, 128), 1), ((2, 16, 300, 211), 4), ((1, 8, 512, 640), 1)):
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(shape, generator=g) * 1.5 + 0.3).to(dev)
    c = shape[1]
    gamma, beta, dy = torch.randn(c, generator=g).to(dev), torch.randn(c, generator=g).to(dev), torch.randn(shape, generator=g).to(dev)
    h = hashlib.sha256()
    for _ in range(3):                                             # (several launches in a row: the counters start from zero every time)
        y, mean, rstd = ops.group_norm(x, gamma, beta, groups, 1e-5, dual_silu=True, want_stats=True)
        dx, dw, db = ops.group_norm_bwd(x, dy, gamma, mean, rstd, groups)
        for t in (y, ops.silu_twin(y), mean, rstd, dx, dw, db):
            h.update(t.cpu().numpy().tobytes())
    print('GN', shape, h.hexdigest())
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for fused in ("1", "0"):
        env = dict(os.environ, MCQUIC_AMD_GN_FUSED=fused, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=root)
        assert run.returncode == 0, run.stderr[-2000:]
        outs[fused] = [ln for ln in run.stdout.splitlines() if ln.startswith("GN ")]
    assert len(outs["1"]) == 5 and outs["1"] == outs["0"]


def test_grouped_gdn_operand_refresh_equals_layer_by_layer(dev):
    """autograd.refresh_gdn_operands (one re-parametrisation launch + two grouped packs for all stale GDN layers) leaves the same
    folded parameters and operand streams as the per-layer path, in place (addresses kept) when the parameters change again."""
    from mcquic_amd import autograd as AG
    from mcquic_amd.nn.gdn import GenDivNorm, InvGenDivNorm
    torch.manual_seed(4)
    layers = [GenDivNorm(128).to(dev), InvGenDivNorm(128).to(dev), GenDivNorm(32).to(dev), GenDivNorm(128).to(dev)]
    with torch.no_grad():
        for m in layers:
            m.gamma.add_(torch.randn_like(m.gamma) * 0.01)
            m.beta.add_(torch.rand_like(m.beta))
    assert AG.refresh_gdn_operands(layers) == 4
    assert AG.refresh_gdn_operands(layers) == 0                     # nothing stale: nothing launched
    addr = [(m.__dict__["_trainOperands"][1].wp.data_ptr(), m.__dict__["_trainOperands"][2].wp.data_ptr()) for m in layers]
    for rnd in range(2):
        for m in layers:
            grouped = m.__dict__["_trainOperands"]
            m2 = type(m)(m.beta.numel()).to(dev)
            with torch.no_grad():
                m2.beta.copy_(m.beta)
                m2.gamma.copy_(m.gamma)
            f, b = AG._gdn_operands(m2, m2.beta, m2.gamma, AG._gdn_bounds(m2))
            assert torch.equal(grouped[1].wp, f.wp) and torch.equal(grouped[1].bias, f.bias) and torch.equal(grouped[2].wp, b.wp)
        with torch.no_grad():                                       # an "optimizer step": versions move, the refresh is in place
            for m in layers[:3]:
                m.gamma.mul_(1.01)
                m.beta.add_(0.001)
        assert AG.refresh_gdn_operands(layers) == 3
        assert addr == [(m.__dict__["_trainOperands"][1].wp.data_ptr(), m.__dict__["_trainOperands"][2].wp.data_ptr()) for m in layers]


@pytest.mark.parametrize("kind", ["compressor", "neon_dense_norm"])
def test_deferred_reduce_passes_give_the_same_gradients(dev, kind, monkeypatch):
    """autograd.backward (weight-gradient reduce passes recorded and run batched at the end of the pass) against a plain
    loss.backward() (every launch reduces right away): every parameter gradient bit for bit -- same partial tiles, same order.
    Third mode: the opt-in queue of round 6 (MCQUIC_AMD_WGRAD_SIDE=1: the weight-gradient launches themselves issued in batches
    on a side stream, docs/experiments.md 11.8) -- the same bits again (a gradient the engine cloned because it saw a second
    reference to a queued output would hold whatever the buffer held before the launch ran)."""
    from mcquic_amd import Compressor, Neon, ops
    from mcquic_amd.autograd import backward, mse_loss
    torch.manual_seed(3407)
    model = (Compressor(32, 2, [64, 32, 16]) if kind == "compressor" else Neon(32, 256, [8, 4, 2, 2], True)).to(dev).train()
    x = (torch.rand((4, 3, 128, 128), generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
    us = None
    if kind == "compressor":
        g = torch.Generator().manual_seed(1)
        us = [(torch.rand((4, 2, s, s, k), generator=g).to(dev), torch.rand((4, 2, s, s, k), generator=g).to(dev)) for s, k in ((8, 64), (4, 32), (2, 16))]
    else:
        g = torch.Generator().manual_seed(1)
        us = [(torch.rand((4, 1, s, s, 256), generator=g).to(dev), torch.rand((4, 1, s, s, 256), generator=g).to(dev)) for s in (2, 2, 4, 8)]
    grads = {}
    ema0 = [f.detach().clone() for f in model._quantizer._entropyCoder._freqEMA]
    for mode in ("plain", "deferred", "queued"):
        monkeypatch.setattr(ops, "_WGRAD_SIDE", mode == "queued")
        for p in model.parameters():
            p.grad = None
        with torch.no_grad():                                       # (the forward moves the frequency EMA, which the random drop reads)
            for f, f0 in zip(model._quantizer._entropyCoder._freqEMA, ema0):
                f.copy_(f0)
        loss = mse_loss(model(x, uniforms=us)[0], x)
        if mode == "plain":
            loss.backward()
        else:
            backward(loss)
        assert ops._lib.load().mcq_wgrad_pending() == 0
        assert not ops._defer["queue"] and not ops._defer["keep"] and ops._defer["side"] is None
        torch.cuda.synchronize()
        grads[mode] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    assert set(grads["plain"]) == set(grads["deferred"]) == set(grads["queued"]) and len(grads["plain"]) > 50
    for n in grads["plain"]:
        assert torch.equal(grads["plain"][n], grads["deferred"][n]), n
        assert torch.equal(grads["plain"][n], grads["queued"][n]), n


def test_deferral_steps_aside_for_accumulation_hooks_and_frozen_weights(dev):
    """the deferred reduce passes hand the engine dW tensors that are only written by the flush.  (1) a second
    backward() onto existing .grad (gradient accumulation) must add REDUCED values: autograd.backward then reduces launch by
    launch; (2) the same with a tensor hook on a parameter (the hook must see the real gradient); (3) a frozen conv weight's
    dW is dropped by the engine -- its memory must not be handed out before the flush writes it."""
    from mcquic_amd import Compressor
    from mcquic_amd.autograd import backward, mse_loss, _leaves_take_by_stealing
    torch.manual_seed(3407)
    model = Compressor(32, 2, [64, 32, 16]).to(dev).train()
    x = (torch.rand((4, 3, 128, 128), generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
    g = torch.Generator().manual_seed(1)
    us = [(torch.rand((4, 2, s, s, k), generator=g).to(dev), torch.rand((4, 2, s, s, k), generator=g).to(dev)) for s, k in ((8, 64), (4, 32), (2, 16))]
    ema0 = [f.detach().clone() for f in model._quantizer._entropyCoder._freqEMA]

    def loss_of():
        with torch.no_grad():
            for f, f0 in zip(model._quantizer._entropyCoder._freqEMA, ema0):
                f.copy_(f0)
        return mse_loss(model(x, uniforms=us)[0], x)

    def grads():
        return {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    for p in model.parameters():
        p.grad = None
    loss_of().backward()
    once = grads()
    # (1) accumulation: two passes through autograd.backward without clearing
    for p in model.parameters():
        p.grad = None
    l1 = loss_of()
    assert _leaves_take_by_stealing([l1])
    backward(l1)
    l2 = loss_of()
    assert not _leaves_take_by_stealing([l2])
    backward(l2)
    for n, v in grads().items():
        assert torch.equal(v, once[n] + once[n]), n
    # (2) a hook on one conv weight sees the reduced gradient
    for p in model.parameters():
        p.grad = None
    w = model._encoder[1]._branch[1].weight
    seen = []
    h = w.register_hook(lambda gr: seen.append(gr.clone()))
    backward(loss_of())
    h.remove()
    assert len(seen) == 1 and torch.equal(seen[0], once["_encoder.1._branch.1.weight"])
    # (3) frozen weights: every other gradient unchanged, nothing written over a live tensor
    for p in model.parameters():
        p.grad = None
    frozen = [model._encoder[1]._branch[1].weight, model._decoder[0]._branch[3].weight, model._encoder[2]._branch[2].gamma]
    for p in frozen:
        p.requires_grad_(False)
    backward(loss_of())
    torch.cuda.synchronize()
    got = grads()
    for p in frozen:
        p.requires_grad_(True)
    skipped = {"_encoder.1._branch.1.weight", "_decoder.0._branch.3.weight", "_encoder.2._branch.2.gamma"}
    assert skipped.isdisjoint(got)
    for n, v in got.items():
        assert torch.equal(v, once[n]), n


def test_generator_state_round_trip(dev):
    """ops.get_rng_state / set_rng_state: a run resumed from the saved state draws what the uninterrupted run drew."""
    from mcquic_amd import ops
    ops.seed_rng(77, dev)
    ops.rng_snapshot(dev)
    saved = ops.get_rng_state(dev)
    a = [ops.hash_uniform(ops.rng_snapshot(dev), 1, (1000,)) for _ in range(3)]
    ops.seed_rng(5, dev)                                            # (something else re-seeds in between)
    ops.set_rng_state(saved, dev)
    b = [ops.hash_uniform(ops.rng_snapshot(dev), 1, (1000,)) for _ in range(3)]
    assert saved.device.type == "cpu" and saved.tolist() == [77, 1]
    for x, y in zip(a, b):
        assert torch.equal(x, y)
