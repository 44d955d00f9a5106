"""parallel.GraphedTrainStep -- the data-parallel training step (BASELINE configs[4]; reference: torchrun + DDP,
mcquic/train/ddp.py:79-95, frequency EMA mcquic/modules/entropyCoder.py:28-44) as main hipGraph + flat all-reduce + post
hipGraph -- against the eager step it replaces: same losses, same parameters, same frequency EMA after several updates."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _uniforms(n, hw, ks, dev, seed):
    g = torch.Generator().manual_seed(seed)
    us = []
    for lv, k in enumerate(ks):
        s = hw // 16 // (2 ** lv)
        us.append((torch.rand((n, 2, s, s, k), generator=g).to(dev), torch.rand((n, 2, s, s, k), generator=g).to(dev)))
    return us


@pytest.mark.parametrize("segments", [1, 3])
@pytest.mark.parametrize("cfg", [(32, [64, 32, 16], 64), (128, [8192, 2048, 512], 128)])
def test_graphed_step_equals_eager_step(dev, cfg, segments):
    """segments = 1: the whole step as one hipGraph; 3: [forward + decoder backward] -> [quantizer backward] -> [encoder backward],
    the cut a multi-rank run overlaps its gradient all-reduces at (forced here at world size 1)."""
    from mcquic_amd import Compressor, parallel
    ch, ks, hw = cfg
    n, steps, lr = 2, 3, 1e-3
    torch.manual_seed(7)
    eager = Compressor(ch, 2, ks).to(dev).train()
    graphed = copy.deepcopy(eager)
    xs = [(torch.rand((n, 3, hw, hw), generator=torch.Generator().manual_seed(20 + i)) * 2 - 1).to(dev) for i in range(steps)]
    us = _uniforms(n, hw, ks, dev, 5)

    opt_e = torch.optim.SGD(eager.parameters(), lr=lr)
    losses_e = []
    for x in xs:
        opt_e.zero_grad(set_to_none=True)
        out = eager(x, uniforms=us)
        loss = torch.nn.functional.mse_loss(out[0], x)
        loss.backward()
        opt_e.step()
        losses_e.append(float(loss.detach()))

    opt_g = torch.optim.SGD(graphed.parameters(), lr=lr)
    step = parallel.GraphedTrainStep(graphed, opt_g, xs[0], forward_kwargs={"uniforms": us}, segments=segments)
    assert step.post is not None, "SGD's update should have been captured"
    assert len(step.graphs) == segments and len(step.slices) == segments
    # the gradient message is EVERY trainable parameter (202.2 MB for the qp=2 model: 50 558 738 float32), cut by stage
    trainable = sum(p.numel() for p in graphed.parameters() if p.requires_grad)
    assert step.flat.numel() == trainable == sum(sl.numel() for sl in step.slices)
    if ch == 128:
        assert trainable == 50558738
        if segments == 3:
            assert [sl.numel() for sl in step.slices] == [5376396, 41587206, 3595136]      # decoder, quantizer, encoder
    losses_g = [float(step(x)) for x in xs]
    step.close()
    torch.cuda.synchronize()

    for a, b in zip(losses_e, losses_g):
        assert abs(a - b) <= 1e-6 * max(1.0, abs(a)), (losses_e, losses_g)
    moved = 0.0
    init = dict(copy.deepcopy(eager).named_parameters())            # (only for the names)
    for (name, pe), (_, pg) in zip(eager.named_parameters(), graphed.named_parameters()):
        scale = max(float(pe.detach().abs().max()), 1e-12)
        assert float((pe.detach() - pg.detach()).abs().max()) <= 2e-6 * scale, name
        moved = max(moved, float(pe.detach().abs().max()))
    assert moved > 0 and len(init) > 0
    for fe, fg in zip(eager._quantizer._entropyCoder._freqEMA, graphed._quantizer._entropyCoder._freqEMA):
        assert torch.allclose(fe, fg, rtol=0, atol=1e-7)
    # leaving the graphed step: the model's eager paths see the updated weights (operand streams re-packed on demand)
    graphed.eval()
    eager.eval()
    with torch.no_grad():
        ce, cg = eager.encode(xs[0]), graphed.encode(xs[0])
    assert all(torch.equal(a, b) for a, b in zip(ce, cg))


def test_graphed_step_survives_eager_use_in_between(dev):
    """train -> invalidate -> evaluate (encode re-packs every conv's streams) -> train.  The step's graphs hold the
    addresses of the operand streams they were captured with; eager re-packs now happen in place and the step pins what it
    captured, so the interleaved run equals an eager loop doing the same, and a closed step refuses to replay."""
    from mcquic_amd import Compressor, parallel
    ks, hw, n, lr = [64, 32, 16], 64, 2, 1e-2
    torch.manual_seed(9)
    eager = Compressor(32, 2, ks).to(dev).train()
    graphed = copy.deepcopy(eager)
    xs = [(torch.rand((n, 3, hw, hw), generator=torch.Generator().manual_seed(40 + i)) * 2 - 1).to(dev) for i in range(4)]
    us = _uniforms(n, hw, ks, dev, 6)
    opt_e = torch.optim.SGD(eager.parameters(), lr=lr, momentum=0.9)
    codes_e = []
    for i, x in enumerate(xs):
        opt_e.zero_grad(set_to_none=True)
        torch.nn.functional.mse_loss(eager(x, uniforms=us)[0], x).backward()
        opt_e.step()
        if i == 1:
            eager.eval()
            with torch.no_grad():
                codes_e = eager.encode(xs[0])
            eager.train()
    step = parallel.GraphedTrainStep(graphed, torch.optim.SGD(graphed.parameters(), lr=lr, momentum=0.9), xs[0], forward_kwargs={"uniforms": us})
    streams = {id(m): m._packed.wp.data_ptr() for m in graphed.modules() if getattr(m, "_packed", None) is not None}
    for i, x in enumerate(xs):
        step(x)
        if i == 1:
            step.invalidate()
            graphed.eval()
            with torch.no_grad():
                codes_g = graphed.encode(xs[0])
            # churn the allocator: were the old streams freed, this would land on them
            junk = [torch.full((1 << 18,), float("nan"), device=dev) for _ in range(64)]
            del junk
            graphed.train()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(codes_e, codes_g))
    now = {id(m): m._packed.wp.data_ptr() for m in graphed.modules() if getattr(m, "_packed", None) is not None}
    assert len(streams) > 100 and all(now[k] == v for k, v in streams.items()), "operand streams moved"     # (eval packs a few layers more)
    for (name, pe), (_, pg) in zip(eager.named_parameters(), graphed.named_parameters()):
        scale = max(float(pe.detach().abs().max()), 1e-12)
        assert float((pe.detach() - pg.detach()).abs().max()) <= 4e-6 * scale, name
    step.close()
    with pytest.raises(RuntimeError, match="closed"):
        step(xs[0])


@pytest.mark.parametrize("own", [False, True])
def test_graphed_step_keeps_a_resumed_optimizer_state(dev, own):
    """(`own`: mcquic_amd.optim.Adam resumed from torch.optim.Adam's checkpoint.)  ADVICE r3: an optimizer restored from a checkpoint (momentum buffers, Adam moments, step counters) must come out of the
    constructor with that state -- the throw-away update that creates missing state restores what was there."""
    from mcquic_amd import Compressor, parallel
    ks, hw = [64, 32, 16], 64
    torch.manual_seed(5)
    model = Compressor(32, 2, ks).to(dev).train()
    x = (torch.rand((2, 3, hw, hw), generator=torch.Generator().manual_seed(3)) * 2 - 1).to(dev)
    us = _uniforms(2, hw, ks, dev, 8)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)
    for _ in range(3):                                        # "the checkpoint": three eager updates
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.mse_loss(model(x, uniforms=us)[0], x).backward()
        opt.step()
    want = {(i, k): v.detach().clone() for i, (p, st) in enumerate(opt.state.items()) for k, v in st.items() if torch.is_tensor(v)}
    assert want and all(int(st["step"]) == 3 for st in opt.state.values())
    params = [p.detach().clone() for p in model.parameters()]
    if own:
        from mcquic_amd import optim
        sd = opt.state_dict()
        opt = optim.Adam(model.parameters(), lr=1e-3)
        opt.load_state_dict(sd)
    step = parallel.GraphedTrainStep(model, opt, x, forward_kwargs={"uniforms": us})
    got = {(i, k): v for i, (p, st) in enumerate(opt.state.items()) for k, v in st.items() if torch.is_tensor(v)}
    assert got.keys() == want.keys()
    for key in want:
        assert torch.equal(got[key].detach().to(want[key].device), want[key]), key
    assert all(torch.equal(a, b.detach()) for a, b in zip(params, model.parameters()))
    step(x)
    assert all(int(st["step"]) == 4 for st in opt.state.values())
    step.close()


def test_graphed_step_really_updates(dev):
    """The captured update moves the parameters on every replay and the loss goes down on a fixed batch."""
    from mcquic_amd import Compressor, parallel
    torch.manual_seed(3)
    model = Compressor(32, 2, [64, 32, 16]).to(dev).train()
    x = (torch.rand((2, 3, 64, 64), generator=torch.Generator().manual_seed(1)) * 2 - 1).to(dev)
    before = [p.detach().clone() for p in model.parameters() if p.requires_grad]
    step = parallel.GraphedTrainStep(model, torch.optim.SGD(model.parameters(), lr=0.05), x,
                                     forward_kwargs={"uniforms": _uniforms(2, 64, [64, 32, 16], dev, 9)})
    losses = [float(step(x)) for _ in range(12)]
    with pytest.raises(RuntimeError, match="captured for shards of shape"):
        step(x[:1])                                           # (a smaller shard would broadcast into the static input)
    step.close()
    after = [p.detach() for p in model.parameters() if p.requires_grad]
    assert any(not torch.equal(a, b) for a, b in zip(before, after))
    assert losses[-1] < losses[0], losses


@pytest.mark.parametrize("capturable", [True, False])
def test_graphed_step_with_adam(dev, capturable):
    """Adam (what the reference's configs train with): captured in the post graph when built with capturable=True, run eagerly
    after the exchange otherwise -- either way the loss on a fixed batch goes down and the state advances every step."""
    from mcquic_amd import Compressor, parallel
    torch.manual_seed(4)
    model = Compressor(32, 2, [64, 32, 16]).to(dev).train()
    x = (torch.rand((2, 3, 64, 64), generator=torch.Generator().manual_seed(2)) * 2 - 1).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=2e-3, capturable=capturable)
    step = parallel.GraphedTrainStep(model, opt, x, forward_kwargs={"uniforms": _uniforms(2, 64, [64, 32, 16], dev, 9)})
    assert (step.post is not None) == capturable
    losses = [float(step(x)) for _ in range(10)]
    step.close()
    assert losses[-1] < losses[0], losses
    st = next(iter(opt.state.values()))
    assert int(st["step"]) == 10


def test_graphed_step_neon(dev):
    """The Neon family (ResidualBackwardQuantizer, VariousMCoder's frequency EMA, op-by-op autograd blocks) through the same
    graphed step: captures, replays, stays finite, its deferred frequency EMA is applied."""
    from mcquic_amd import Neon, parallel
    torch.manual_seed(11)
    model = Neon(32, 256, [8, 4, 2, 2], False).to(dev).train()             # (tests/test_neon.py's fixture configuration)
    x = (torch.rand((2, 3, 128, 128), generator=torch.Generator().manual_seed(6)) * 2 - 1).to(dev)
    before = [p.detach().clone() for p in model.parameters() if p.requires_grad]
    step = parallel.GraphedTrainStep(model, torch.optim.SGD(model.parameters(), lr=1e-2), x)
    losses = [float(step(x)) for _ in range(8)]
    step.close()
    assert all(l == l and l < 10 for l in losses), losses
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, [p for p in model.parameters() if p.requires_grad]))
    assert len(step.coders) == 1 and step.counts is not None
    assert any(not torch.allclose(f, torch.ones_like(f) / f.shape[-1]) for f in step.coders[0]._freqEMA)   # global counts applied


def test_no_autograd_graph_outlives_an_iteration(dev):
    """After backward() nothing may still carry a graph: a tensor that does keeps AccumulateGrad nodes alive across iterations
    (memory, and hipGraph capture of the next iteration turns into a nested fork -- hipStreamEndCapture segfaulted on the Neon
    family until SiluFn stopped returning the twin object its input carries)."""
    import gc
    from mcquic_amd import Compressor, Neon
    for model, hw in ((Compressor(32, 2, [64, 32, 16]), 64), (Neon(32, 256, [8, 4, 2, 2], False), 128), (Neon(32, 256, [8, 4, 2, 2], True), 128)):
        model = model.to(dev).train()
        x = (torch.rand((2, 3, hw, hw), generator=torch.Generator().manual_seed(1)) * 2 - 1).to(dev)
        gc.collect()
        earlier = {id(o) for o in gc.get_objects() if isinstance(o, torch.Tensor) and o.grad_fn is not None}   # (other tests' leftovers)
        for _ in range(2):
            for p in model.parameters():
                p.grad = None
            out = model(x)
            torch.nn.functional.mse_loss(out[0], x).backward()
            del out
        gc.collect()
        alive = [(tuple(o.shape), type(o.grad_fn).__name__) for o in gc.get_objects()
                 if isinstance(o, torch.Tensor) and o.grad_fn is not None and id(o) not in earlier]
        assert not alive, (type(model).__name__, alive[:8])


@pytest.mark.parametrize("cfg", [(32, [64, 32, 16], 64), (128, [8192, 2048, 512], 128)])
@pytest.mark.parametrize("clips", [True, False])
def test_graphed_step_clips_like_clip_grad_norm(dev, cfg, clips):
    """The reference's step clips the gradient by its global norm before the update (mcquic/train/trainer.py:280): the graphed
    step's `max_grad_norm` against torch.nn.utils.clip_grad_norm_ in an eager loop -- same norms, same parameters afterwards --
    with a bound that bites (a third of the first step's norm) and one that never does."""
    from mcquic_amd import Compressor, parallel
    ch, ks, hw = cfg
    n, steps, lr = 2, 3, 1e-2
    torch.manual_seed(11)
    eager = Compressor(ch, 2, ks).to(dev).train()
    graphed = copy.deepcopy(eager)
    xs = [(torch.rand((n, 3, hw, hw), generator=torch.Generator().manual_seed(60 + i)) * 2 - 1).to(dev) for i in range(steps)]
    us = _uniforms(n, hw, ks, dev, 12)
    # the first step's norm sets the bound
    torch.nn.functional.mse_loss(eager(xs[0], uniforms=us)[0], xs[0]).backward()
    first = float(torch.nn.utils.clip_grad_norm_(eager.parameters(), 1e9))
    eager = copy.deepcopy(graphed)
    bound = first / 3 if clips else first * 100
    opt_e = torch.optim.SGD(eager.parameters(), lr=lr)
    norms_e = []
    for x in xs:
        opt_e.zero_grad(set_to_none=True)
        torch.nn.functional.mse_loss(eager(x, uniforms=us)[0], x).backward()
        norms_e.append(float(torch.nn.utils.clip_grad_norm_(eager.parameters(), bound)))
        opt_e.step()
    step = parallel.GraphedTrainStep(graphed, torch.optim.SGD(graphed.parameters(), lr=lr), xs[0], forward_kwargs={"uniforms": us},
                                     max_grad_norm=bound)
    assert step.post is not None
    norms_g = []
    for x in xs:
        step(x)
        norms_g.append(float(step.grad_norm))
    step.close()
    assert abs(norms_e[0] - first) <= 1e-5 * first
    for a, b in zip(norms_e, norms_g):
        assert abs(a - b) <= 2e-5 * a, (norms_e, norms_g)
    for (name, pe), (_, pg) in zip(eager.named_parameters(), graphed.named_parameters()):
        scale = max(float(pe.detach().abs().max()), 1e-12)
        assert float((pe.detach() - pg.detach()).abs().max()) <= 4e-6 * scale, name
    with pytest.raises(ValueError):
        parallel.GraphedTrainStep(graphed, torch.optim.SGD(graphed.parameters(), lr=lr), xs[0], max_grad_norm=0.0)


def test_sumsq_and_clip_by_norm(dev):
    from mcquic_amd import ops
    for n in (1, 300, 65537, 50558738):
        x = torch.randn(n, generator=torch.Generator().manual_seed(n % 1000)).to(dev)
        want = float(x.double().pow(2).sum())
        assert abs(float(ops.sumsq(x)) - want) <= 3e-7 * want
        norm = want ** 0.5
        y = x.clone()
        got = ops.clip_by_norm_(y, norm / 2)
        assert abs(float(got) - norm) <= 3e-7 * norm
        ref = x * ((norm / 2) / (norm + 1e-6))
        assert float((y - ref).abs().max()) <= 2e-7 * float(ref.abs().max()) + 1e-30
        z = x.clone()
        ops.clip_by_norm_(z, norm * 2)
        assert torch.equal(z, x)                                              # a bound above the norm leaves the buffer alone
