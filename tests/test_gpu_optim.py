"""mcquic_amd.optim.Adam / AdamW (one launch over the whole model, csrc/train_ops.hip: mcq_adam_step_f32) against torch.optim.Adam /
AdamW: same parameters after several updates, checkpoints exchanged in both directions, a device learning rate, and the update
captured inside parallel.GraphedTrainStep (the reference's step: mcquic/train/trainer.py:283 with `Adam`, configs/a800_8.yaml:20-25)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
SHAPES = [(1,), (2, 1, 1, 1), (3,), (128,), (4097,), (128, 128, 3, 3), (2, 8192, 64), (5, 7, 11), (12288,), (128, 3, 3, 3)]


def _params(dev, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in SHAPES]


def _grads(params, seed):
    g = torch.Generator().manual_seed(seed)
    for p in params:
        p.grad = (torch.randn(p.shape, generator=g) * 0.1).to(p.device)


def _close(a, b, tol, what):
    scale = max(float(b.abs().max()), 1e-12)
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"{what}: {err / scale:.3e} of the largest entry"


@pytest.mark.parametrize("kw", [dict(), dict(weight_decay=0.1), dict(betas=(0.8, 0.95), eps=1e-6), dict(maximize=True)])
@pytest.mark.parametrize("decoupled", [False, True])
def test_adam_matches_torch(dev, kw, decoupled):
    """Six updates next to torch.optim.Adam / AdamW on the device AND next to the same optimizer in float64 on the CPU: Adam divides
    by sqrt(v), so where a gradient nearly cancels (L2 decay: g + wd * p) one rounding of the sum moves the update by a visible
    fraction of lr -- torch's float32 run is that far from the float64 one too.  Bar: within 2e-6 of torch's float32 result, or no
    further from the float64 truth than 3x torch's own float32 distance."""
    from mcquic_amd import optim
    ours, theirs = _params(dev, 1), _params(dev, 1)
    truth = [torch.nn.Parameter(p.detach().double().cpu()) for p in ours]
    if decoupled:
        oo, ot, o64 = optim.AdamW(ours, lr=3e-3, **kw), torch.optim.AdamW(theirs, lr=3e-3, **kw), torch.optim.AdamW(truth, lr=3e-3, **kw)
    else:
        oo, ot, o64 = optim.Adam(ours, lr=3e-3, **kw), torch.optim.Adam(theirs, lr=3e-3, **kw), torch.optim.Adam(truth, lr=3e-3, **kw)
    for it in range(6):
        _grads(ours, 10 + it)
        _grads(theirs, 10 + it)
        for p, q in zip(truth, ours):
            p.grad = q.grad.double().cpu()
        oo.step()
        ot.step()
        o64.step()
    for i, (a, b, t) in enumerate(zip(ours, theirs, truth)):
        scale = max(float(b.detach().abs().max()), 1e-12)
        d_torch = float((a.detach() - b.detach()).abs().max())
        e_ours = float((a.detach().double().cpu() - t.detach()).abs().max())
        e_theirs = float((b.detach().double().cpu() - t.detach()).abs().max())
        assert d_torch <= 2e-6 * scale or e_ours <= 3 * e_theirs + 1e-7 * scale, (SHAPES[i], d_torch / scale, e_ours / scale, e_theirs / scale)
        loose = 1.0 if d_torch <= 2e-6 * scale else 20.0           # (moments inherit what the decayed parameters differ by)
        _close(oo.state[a]["exp_avg"], ot.state[b]["exp_avg"], 2e-6 * loose, "exp_avg")
        _close(oo.state[a]["exp_avg_sq"], ot.state[b]["exp_avg_sq"], 4e-6 * loose, "exp_avg_sq")
        assert float(oo.state[a]["step"]) == 6.0


def test_adam_checkpoints_travel_both_ways(dev):
    """torch.optim.Adam's state_dict loads into ours and the other way round; the runs continue as one."""
    from mcquic_amd import optim
    a, b = _params(dev, 2), _params(dev, 2)
    ot = torch.optim.Adam(a, lr=1e-2)
    for it in range(3):
        _grads(a, it)
        ot.step()
    with torch.no_grad():
        for x, y in zip(a, b):
            y.copy_(x)
    oo = optim.Adam(b, lr=1e-2)
    oo.load_state_dict(copy.deepcopy(ot.state_dict()))
    for it in range(3, 6):
        _grads(a, it)
        _grads(b, it)
        ot.step()
        oo.step()
    for x, y in zip(a, b):
        _close(y.detach(), x.detach(), 2e-6, "after loading torch's checkpoint")
    # ... and back: ours -> a fresh torch optimizer (capturable, so `step` stays a device tensor)
    c = _params(dev, 2)
    with torch.no_grad():
        for x, y in zip(b, c):
            y.copy_(x)
    o2 = torch.optim.Adam(c, lr=1e-2, capturable=True)
    sd = copy.deepcopy(oo.state_dict())
    for gpar in sd["param_groups"]:
        for k in ("decoupled",):
            gpar.pop(k, None)
        gpar.update(amsgrad=False, foreach=None, fused=None, differentiable=False, capturable=True, decoupled_weight_decay=False)
    o2.load_state_dict(sd)
    for it in range(6, 8):
        _grads(b, it)
        _grads(c, it)
        oo.step()
        o2.step()
    for x, y in zip(b, c):
        _close(y.detach(), x.detach(), 2e-6, "torch continuing from our checkpoint")


def test_adamw_checkpoints_keep_the_decoupled_decay(dev):
    """a torch.optim.AdamW checkpoint (group key `decoupled_weight_decay`, no `decoupled`) loaded into optim.AdamW must stay
    AdamW -- with weight_decay > 0 the two decays move the parameters visibly differently -- and our checkpoint loaded into
    torch.optim.AdamW likewise, without hand-patching the group keys.  The flat moment buffers keep their addresses across a load."""
    from mcquic_amd import optim
    a, b = _params(dev, 5), _params(dev, 5)
    ot = torch.optim.AdamW(a, lr=1e-2, weight_decay=0.3)
    for it in range(3):
        _grads(a, it)
        ot.step()
    with torch.no_grad():
        for x, y in zip(a, b):
            y.copy_(x)
    oo = optim.AdamW(b, lr=1e-2, weight_decay=0.3)
    _grads(b, 0)
    oo.prepare()
    addr = [oo._plans[0].flat_m.data_ptr(), oo._plans[0].flat_v.data_ptr(), oo._plans[0].step.data_ptr()]
    oo.load_state_dict(copy.deepcopy(ot.state_dict()))
    assert oo.param_groups[0]["decoupled"] is True
    for it in range(3, 6):
        _grads(a, it)
        _grads(b, it)
        ot.step()
        oo.step()
    assert addr == [oo._plans[0].flat_m.data_ptr(), oo._plans[0].flat_v.data_ptr(), oo._plans[0].step.data_ptr()]
    for x, y in zip(a, b):
        _close(y.detach(), x.detach(), 2e-6, "AdamW continuing from torch's AdamW checkpoint")
    # L2 decay instead would have ended elsewhere
    l2 = _params(dev, 5)
    with torch.no_grad():
        for x, y in zip(a, l2):
            y.copy_(x)
    assert any(float((x.detach() - y.detach()).abs().max()) > 1e-3 for x, y in zip(a, _params(dev, 5)))
    # ... and back into torch.optim.AdamW, group keys as they come
    c = _params(dev, 5)
    with torch.no_grad():
        for x, y in zip(b, c):
            y.copy_(x)
    o2 = torch.optim.AdamW(c, lr=1e-2, weight_decay=0.3, capturable=True)
    sd = copy.deepcopy(oo.state_dict())
    assert sd["param_groups"][0]["decoupled_weight_decay"] is True
    for gpar in sd["param_groups"]:
        gpar.update(amsgrad=False, foreach=None, fused=None, differentiable=False, capturable=True)
    o2.load_state_dict(sd)
    assert o2.param_groups[0]["decoupled_weight_decay"] is True
    for it in range(6, 8):
        _grads(b, it)
        _grads(c, it)
        oo.step()
        o2.step()
    for x, y in zip(b, c):
        _close(y.detach(), x.detach(), 2e-6, "torch AdamW continuing from our checkpoint")


def test_adam_device_learning_rate_and_capture(dev):
    """The learning rate as a device tensor: refilled between steps, read by the kernel (also from a captured graph)."""
    from mcquic_amd import optim
    ours, theirs = _params(dev, 3), _params(dev, 3)
    lr_o, lr_t = torch.tensor(0.0, device=dev), torch.tensor(0.0, device=dev)
    oo, ot = optim.Adam(ours, lr=lr_o), torch.optim.Adam(theirs, lr=lr_t, capturable=True)
    _grads(ours, 5)
    _grads(theirs, 5)
    oo.prepare()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        lr_o.fill_(1e-3)
        oo.step()                                              # warm-up outside the capture
    torch.cuda.current_stream().wait_stream(side)
    lr_t.fill_(1e-3)
    ot.step()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        oo.step()
    for it in range(4):
        rate = 1e-3 * (it + 2)
        lr_o.fill_(rate)
        lr_t.fill_(rate)
        g = torch.Generator().manual_seed(50 + it)
        for p, q in zip(ours, theirs):
            fresh = (torch.randn(p.shape, generator=g) * 0.1).to(dev)
            p.grad.copy_(fresh)                                # (same addresses: the graph reads them)
            q.grad.copy_(fresh)
        graph.replay()
        ot.step()
    for a, b in zip(ours, theirs):
        _close(a.detach(), b.detach(), 2e-6, "captured update with a scheduled rate")
    assert float(oo.state[ours[0]]["step"]) == 5.0
    # a changed gradient address inside a capture is refused, not silently captured as a host copy
    ours[0].grad = ours[0].grad.clone()
    g2 = torch.cuda.CUDAGraph()
    with pytest.raises(RuntimeError):
        with torch.cuda.graph(g2):
            oo.step()


def test_adam_refuses_cpu_tensors():
    from mcquic_amd import optim
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError):
        optim.Adam([p], lr=1e-3).step()


def test_graphed_step_with_own_adam_equals_eager_torch_adam(dev):
    from mcquic_amd import Compressor, optim, parallel
    from test_gpu_graphed_step import _uniforms
    ch, ks, hw, n, steps = 32, [64, 32, 16], 64, 2, 4
    torch.manual_seed(21)
    eager = Compressor(ch, 2, ks).to(dev).train()
    graphed = copy.deepcopy(eager)
    xs = [(torch.rand((n, 3, hw, hw), generator=torch.Generator().manual_seed(70 + i)) * 2 - 1).to(dev) for i in range(steps)]
    us = _uniforms(n, hw, ks, dev, 13)
    opt_e = torch.optim.Adam(eager.parameters(), lr=1e-3)
    losses_e = []
    for x in xs:
        opt_e.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(eager(x, uniforms=us)[0], x)
        loss.backward()
        opt_e.step()
        losses_e.append(float(loss.detach()))
    step = parallel.GraphedTrainStep(graphed, optim.Adam(graphed.parameters(), lr=1e-3), xs[0], forward_kwargs={"uniforms": us})
    assert step.post is not None, "the update should have been captured"
    losses_g = [float(step(x)) for x in xs]
    step.close()
    for a, b in zip(losses_e, losses_g):
        assert abs(a - b) <= 2e-6 * max(1.0, abs(a)), (losses_e, losses_g)
    for (name, pe), (_, pg) in zip(eager.named_parameters(), graphed.named_parameters()):
        scale = max(float(pe.detach().abs().max()), 1e-12)
        assert float((pe.detach() - pg.detach()).abs().max()) <= 2e-5 * scale, name      # (Adam divides by sqrt(v): rounding of tiny gradients shows)
