"""torch.autograd glue for the training step (BASELINE config #5): each op below runs HIP kernels in both
directions; autograd only walks the graph and accumulates parameter gradients (so torch DDP / optimizers work on
the module tree unchanged).  The inference path (`eval()` / no_grad) never comes through here -- it uses the
fused launches of nn/blocks.py.

Backward identities (reference: what torch.autograd derives for mcquic/nn/*.py):
  conv  y = W * x + b          dx = W^T (flipped) * dy  -- the forward kernel on transformed weights;
                               stride 2: the sub-pixel identity dX = PixelShuffle2(conv3x3(dY, W2)), W2 holding for
                               each input phase (i, j) the taps that reach it; pixel-shuffle convs: unshuffle dY first
                               dW = mcq_conv2d_wgrad_f32,  db = channel sums of dy
  silu, gate, GDN / IGDN       element-wise kernels of csrc/train_ops.hip (+ 1x1 dgrad / wgrad for gamma)
"""
from __future__ import annotations

from typing import Optional

import torch

import os

from . import ops

_GDN_BWD_FUSED = os.environ.get("MCQUIC_AMD_GDN_BWD_FUSED", "1") != "0"

# ---- input-gradient operand streams (packed straight from the OIHW parameter, cached per weight version) ----------
class _DgradCache:
    def __init__(self):
        self.key, self.packed, self.winograd = None, None, False

    def get(self, weight: torch.Tensor, stride: int) -> ops.PackedConv:
        key = (ops.tensor_version(weight), weight.data_ptr(), stride)
        if key != self.key or self.winograd != ops.winograd_enabled():
            if self.packed is not None and self.key is not None and self.key[2] == stride and not self.winograd and not ops.winograd_enabled():
                # same layer, new weight version: into the existing stream (same address, no allocation)
                self.packed = ops.pack_convs([weight], dgrad=True, stride=stride, into=[self.packed])[0]
            else:
                self.packed = ops.PackedConv.dgrad(weight, stride)  # one pack launch (flip / transpose / sub-pixel scatter inside)
            self.key, self.winograd = key, ops.winograd_enabled()
        return self.packed


def _dgrad_packed(conv, weight: torch.Tensor) -> ops.PackedConv:
    cache = conv.__dict__.get("_dgradCache")
    if cache is None:
        cache = conv.__dict__["_dgradCache"] = _DgradCache()
    return cache.get(weight, conv.stride)


class ConvFn(torch.autograd.Function):
    """y = conv(x, W) + b (+ res); `shuffle2` stores through PixelShuffle(2)."""

    @staticmethod
    def forward(ctx, x, weight, bias, res, conv, shuffle2, dual_silu):
        y = ops.conv2d(x, conv.packed(), conv.stride, shuffle2=shuffle2, res=res, dual_silu=bool(dual_silu) and not shuffle2)
        sy = ops.silu_twin(y)                       # silu(y) from the same launch, for a consumer that starts with an activation
        ctx.save_for_backward(x, weight)
        ctx.conv, ctx.shuffle2, ctx.has_bias, ctx.has_res = conv, shuffle2, bias is not None, res is not None
        if sy is not None:
            ctx.mark_non_differentiable(sy)
        ctx.set_materialize_grads(False)
        return y, sy

    @staticmethod
    def backward(ctx, dy, _dsy=None):
        x, weight = ctx.saved_tensors
        if dy is None:
            return (None,) * 7
        conv = ctx.conv
        dy = dy.contiguous()
        dyc = ops.pixel_unshuffle2(dy) if ctx.shuffle2 else dy
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = _dgrad_packed(conv, weight)
            if conv.stride == 1:
                dx = ops.conv2d(dyc, wt)
            else:
                dx = ops.conv2d(dyc, wt, shuffle2=True)
                if dx.shape[-2] != x.shape[-2] or dx.shape[-1] != x.shape[-1]:
                    dx = dx[..., :x.shape[-2], :x.shape[-1]].contiguous()
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if want_db:
                dw, db = ops.conv2d_wgrad(x, dyc, conv.kernelSize, conv.stride, want_bias=True)   # one kernel for both
            else:
                dw = ops.conv2d_wgrad(x, dyc, conv.kernelSize, conv.stride)
        elif want_db:
            db = ops.channel_sum(dyc)
        return dx, dw, db, (dy if ctx.has_res else None), None, None, None


class SiluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        twin = ops.silu_twin(x)                     # the producer may have stored silu(x) beside x already
        # (a fresh tensor object over the twin's storage: returning the twin itself would hang THIS node on the object x carries
        #  as an attribute -- x -> twin -> grad_fn -> saved x, a cycle through C++ that outlives the iteration and keeps
        #  AccumulateGrad nodes alive, which breaks hipGraph capture of the next iteration)
        return twin.detach() if twin is not None else ops.silu(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.silu_bwd(x, dy.contiguous())


class SiluForkFn(torch.autograd.Function):
    """(silu(x), x) for a block whose branch starts with the activation while its skip convolution reads x itself
    (ResidualBlockWithStride / ResidualBlockShuffle, mcquic/nn/blocks.py:98-159): the two gradients meet in ONE launch,
    dx = d_silu * silu'(x) + d_skip, instead of a SiLU-backward launch followed by an add issued by the autograd engine."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        twin = ops.silu_twin(x)
        ctx.set_materialize_grads(False)
        return (twin.detach() if twin is not None else ops.silu(x)), x.view_as(x)

    @staticmethod
    def backward(ctx, dsx, dx2):
        (x,) = ctx.saved_tensors
        if dsx is None:
            return dx2
        return ops.silu_bwd(x, dsx.contiguous(), None if dx2 is None else dx2.contiguous())


class SubPassFn(torch.autograd.Function):
    """(a - b, b) where b is consumed once more downstream -- a level's residual z - dequant next to the decoder's use of the
    dequantised sample (mcquic/modules/quantizer.py:295-305): b's two gradients meet in one launch, d b = d_b2 - d_residual
    (a negation launch plus an engine-issued add otherwise).  The residual carries its SiLU twin: the next level starts with
    an activation."""

    @staticmethod
    def forward(ctx, a, b):
        r = ops.axpby(a, b, 1.0, -1.0, dual_silu=True)
        sr = ops.silu_twin(r)
        ctx.mark_non_differentiable(sr)
        ctx.set_materialize_grads(False)
        return r, b.view_as(b), sr

    @staticmethod
    def backward(ctx, dr, db2, _dsr):
        if dr is None:
            return None, db2
        dr = dr.contiguous()
        if db2 is None:
            return dr, ops.axpby(dr, dr, -1.0, 0.0)
        return dr, ops.axpby(db2.contiguous(), dr, 1.0, -1.0)


class ForkFn(torch.autograd.Function):
    """k aliases of x for k consumers: their gradients are summed by this library's add kernels (one launch for two or three
    consumers) instead of by adds the autograd engine issues itself."""

    @staticmethod
    def forward(ctx, x, k):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(k))

    @staticmethod
    def backward(ctx, *ds):
        ds = [d.contiguous() for d in ds if d is not None]
        if not ds:
            return None, None
        dx = ds[0]
        i = 1
        while i + 1 < len(ds):
            dx = ops.add3(dx, ds[i], ds[i + 1])
            i += 2
        if i < len(ds):
            dx = ops.add(dx, ds[i])
        return dx, None


def fork(x, k: int):
    """k aliases of x (ForkFn), each carrying x's SiLU twin."""
    tx = ops.silu_twin(x)
    outs = ForkFn.apply(x, k)
    if tx is not None:
        for o in outs:
            ops.set_silu_twin(o, tx)
    return outs


class MseFn(torch.autograd.Function):
    """F.mse_loss(a, b) (mcquic/loss/__init__.py:62) from this library's own reduction: no library memset inside a captured step."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        ctx.save_for_backward(a, b)
        return ops.mse(a, b)

    @staticmethod
    def backward(ctx, dloss):
        a, b = ctx.saved_tensors
        da, db = ops.mse_bwd(a, b, dloss.contiguous().float(), want_db=ctx.needs_input_grad[1])
        return (da if ctx.needs_input_grad[0] else None), db


_ONES = {}


def _leaves_take_by_stealing(roots) -> bool:
    """Deferring the weight gradients' reduce passes hands the autograd engine dW / db tensors whose values only exist after the
    flush.  That is sound only if nothing LOOKS at them before: every parameter the pass reaches must take its gradient by stealing
    the tensor -- `.grad is None` (a second backward() without zero_grad would run `p.grad += dW` on unreduced memory) and no tensor
    hook / post-accumulate hook on it (they would be called with it).  Walks the graph under `roots` once (a few hundred nodes).
    (Hooks put on the AccumulateGrad nodes themselves from C++ -- torch DDP's reducer -- are invisible from here: with DDP use
    `loss.backward()`, see `backward`.)"""
    seen, stack = set(), [t.grad_fn for t in roots if t is not None and t.grad_fn is not None]
    while stack:
        fn = stack.pop()
        if fn in seen:
            continue
        seen.add(fn)
        v = getattr(fn, "variable", None)
        if v is not None:                                       # an AccumulateGrad node
            if v.grad is not None or getattr(v, "_backward_hooks", None) or getattr(v, "_post_accumulate_grad_hooks", None):
                return False
            continue
        stack.extend(nf for nf, _ in fn.next_functions if nf is not None)
    return True


def backward(loss: torch.Tensor, defer_reduce: bool = True) -> None:
    """`loss.backward()` with the root gradient taken from a cached 1.0 on the loss's device (the engine otherwise allocates and
    FILLS one per call -- the one ATen launch left in a captured training step) and with the weight gradients' reduce passes of the
    whole pass run in a few launches at its end (ops.wgrad_deferral) -- when that is sound: every parameter reached has `.grad is
    None` and no hooks (`_leaves_take_by_stealing`; gradient accumulation over several backward passes, hooks: the reduce passes run
    one by one as the launches are made, same values).  NOT for a backward pass something else listens to from C++ (torch DDP
    reads gradients from bucket hooks while the pass is still running): there, call `loss.backward()` as usual."""
    key = (loss.device, loss.dtype)
    one = _ONES.get(key)
    if one is None:
        one = _ONES[key] = torch.ones((), dtype=loss.dtype, device=loss.device)
    if not defer_reduce or not _leaves_take_by_stealing([loss]):    # (something reads gradients DURING the pass)
        loss.backward(one if loss.dim() == 0 else None)
        return
    with ops.wgrad_deferral():                                  # (this pass is ours end to end: the weight gradients' reduce passes run batched)
        loss.backward(one if loss.dim() == 0 else None)


def run_backward(tensors, grad_tensors) -> None:
    """torch.autograd.backward(tensors, grad_tensors) with the weight gradients' reduce passes batched (see `backward`)."""
    if not _leaves_take_by_stealing(list(tensors)):
        torch.autograd.backward(tensors, grad_tensors)
        return
    with ops.wgrad_deferral():
        torch.autograd.backward(tensors, grad_tensors)


def mse_loss(a, b):
    """mean((a - b)^2): `MseFn` on a HIP device in float32, torch's own elsewhere."""
    if a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.shape == b.shape and a.numel():
        return MseFn.apply(a, b)
    return torch.nn.functional.mse_loss(a, b)


# ---- ResidualBlock / AttentionBlock as whole autograd nodes ---------------------------------------------------------
# `denseNorm` blocks (mcquic/nn/blocks.py:179-200: nn.GroupNorm where the second activation was; what configs/neon.yaml trains) take
# the same nodes: the middle of the block is  t1 -> GroupNorm -> u  instead of  t1 -> SiLU -> s1  (the SiLU rode in conv1's epilogue;
# the normalisation is launches of its own, csrc/norm.hip), everything around it -- multi-problem launches, the input gradient's
# `* silu'(x) + dy` epilogue, grouped weight gradients -- is shared.  A normalised block's tape entry carries six tensors
# (x, sx, t1, u, mean, rstd) where a plain one carries four (x, sx, t1, s1); its parameters are (w1, b1, w2, b2, gamma, beta).
def _rb_norm(block):
    """The block's GroupNorm module, or None for a plain block."""
    return block._branch[2] if getattr(block, "denseNorm", False) else None


def _rb_nsaved(block) -> int:
    return 6 if _rb_norm(block) is not None else 4


def _rb_params(block):
    c1, c2 = block._branch[1], block._branch[3]
    ps = [c1.weight, c1.bias, c2.weight, c2.bias]
    gn = _rb_norm(block)
    if gn is not None:
        ps += [gn.weight, gn.bias]
    return ps


def _rb_forward(block, x, sx):
    """y = conv2(act2(conv1(silu(x)))) + x, act2 = SiLU (two fused launches) or GroupNorm; returns (y, silu(y), what backward needs)."""
    ys, sys_, saved = _rb_forward_multi([block], [x], [sx])
    return ys[0], sys_[0], saved[0]


def _rb_backward(block, saved, dy, pairs, need_dx: bool = True, norm_grads=None):
    """Input gradient of a ResidualBlock; the two weight-gradient operand pairs are appended to `pairs` for a grouped launch:
    d_t1 = conv(dy, W2^T) * silu'(t1);  dx = conv(d_t1, W1^T) * silu'(x) + dy   (normalised: d_t1 = GroupNorm'(conv(dy, W2^T)))."""
    return _rb_backward_multi([block], [saved], [dy], pairs, norm_grads=norm_grads, need_dx=need_dx)[0]


def _wgrads(pairs):
    """[(dW, db), ...] of the 3x3 stride-1 convs whose (input, output-gradient) pairs are given: one grouped launch."""
    return ops.conv2d_wgrad_group([a for a, _ in pairs], [b for _, b in pairs], want_bias=True)


def _rb_forward_multi(blocks, xs, sxs):
    """Several ResidualBlocks of one shape side by side: each of the two layers is ONE launch for all of them."""
    norm = _rb_norm(blocks[0]) is not None
    t1s = ops.conv2d_multi(sxs, [blk._branch[1].packed() for blk in blocks], dual_silu=not norm)
    if norm:
        mids, stats = [], []
        for blk, t1 in zip(blocks, t1s):
            gn = blk._branch[2]
            u, mean, rstd = ops.group_norm(t1, gn.weight, gn.bias, gn.num_groups, gn.eps, want_stats=True)
            mids.append(u)
            stats.append((mean, rstd))
    else:
        mids = [ops.silu_twin(t) for t in t1s]
    ys = ops.conv2d_multi(mids, [blk._branch[3].packed() for blk in blocks], per_problem=[dict(res=x) for x in xs], dual_silu=True)
    if norm:
        saved = [(x, sx, t1, u, st[0], st[1]) for x, sx, t1, u, st in zip(xs, sxs, t1s, mids, stats)]
    else:
        saved = [(x, sx, t1, s1) for x, sx, t1, s1 in zip(xs, sxs, t1s, mids)]
    return ys, [ops.silu_twin(y) for y in ys], saved


def _rb_backward_multi(blocks, saveds, dys, pairs, norm_grads=None, need_dx: bool = True):
    """Input gradients of several ResidualBlocks of one shape, two launches in all (+ the normalisations' own); operand pairs
    appended per block as (conv1 pair, conv2 pair); a normalised block's (d gamma, d beta) go into `norm_grads` (one entry per block)."""
    c1s, c2s = [blk._branch[1] for blk in blocks], [blk._branch[3] for blk in blocks]
    norm = _rb_norm(blocks[0]) is not None
    if norm:
        d_us = ops.conv2d_multi(dys, [_dgrad_packed(c, c.weight) for c in c2s])
        d_t1s = []
        for blk, sv, d_u in zip(blocks, saveds, d_us):
            gn = blk._branch[2]
            d_t1, dgw, dgb = ops.group_norm_bwd(sv[2], d_u, gn.weight, sv[4], sv[5], gn.num_groups, want_params=True)
            d_t1s.append(d_t1)
            if norm_grads is not None:
                norm_grads.append((dgw, dgb))
    else:
        d_t1s = ops.conv2d_multi(dys, [_dgrad_packed(c, c.weight) for c in c2s], per_problem=[dict(dsilu_mul=sv[2]) for sv in saveds])
    dxs = [None] * len(blocks)
    if need_dx:
        dxs = ops.conv2d_multi(d_t1s, [_dgrad_packed(c, c.weight) for c in c1s],
                               per_problem=[dict(dsilu_mul=sv[0], res=dy) for sv, dy in zip(saveds, dys)])
    for sv, d_t1, dy in zip(saveds, d_t1s, dys):
        pairs.append((sv[1], d_t1))
        pairs.append((sv[3], dy))
    return dxs


class ResidualBlockFn(torch.autograd.Function):
    """y = conv2(act2(conv1(silu(x)))) + x  (mcquic/nn/blocks.py:179-200) as two fused launches each way:
         forward   t1, silu(t1) = conv1(silu(x))                      (SiLU twin stored by the producing launch)
                   y,  silu(y)  = conv2(silu(t1)) + x
         backward  d_t1 = conv(dy, W2^T) * silu'(t1)                   (MCQ_CONV_DSILU_MUL epilogue)
                   dx   = conv(d_t1, W1^T) * silu'(x) + dy             (... + MCQ_CONV_RESIDUAL: the skip path's gradient)
                   dW1, db1, dW2, db2: ONE grouped weight-gradient launch over (silu(x), d_t1) and (silu(t1), dy)
       No stand-alone SiLU / SiLU-backward / add kernels, no channel-major copies.  `sx` = silu(x) is an input so that a
       producer's twin is reused; the second output silu(y) is the next block's `sx` (no gradient flows through it: every
       consumer differentiates through y itself).  `denseNorm`: act2 = GroupNorm (see above), parameters + (gamma, beta)."""

    @staticmethod
    def forward(ctx, x, sx, *rest):
        block = rest[-1]
        y, sy, saved = _rb_forward(block, x, sx)
        ctx.save_for_backward(*saved)
        ctx.block = block
        ctx.mark_non_differentiable(sy)
        ctx.set_materialize_grads(False)        # (or autograd fills a zero tensor of sy's size for `_dsy` on every backward)
        return y, sy

    @staticmethod
    def backward(ctx, dy, _dsy):
        pairs, ng = [], []
        dx = _rb_backward(ctx.block, ctx.saved_tensors, dy.contiguous(), pairs, ctx.needs_input_grad[0], norm_grads=ng)
        (dw1, db1), (dw2, db2) = _wgrads(pairs)
        extra = list(ng[0]) if ng else []
        return (dx, None, dw1, db1, dw2, db2, *extra, None)


class AttentionBlockFn(torch.autograd.Function):
    """out = a * sigmoid(b) + x,  a = RB^3(x),  b = conv1x1(RB^3(x))   (mcquic/nn/blocks.py:245-288) as ONE autograd node.
    The two stacks apply the same layer shapes to different tensors: layer by layer they share a launch (mcq_conv2d_multi_f32),
    in both directions -- on the 16x16 ... 4x4 maps of a training crop a launch is latency, not work --, every ResidualBlock
    is two fused launches each way, and the twelve 3x3 weight gradients of the block leave in ONE grouped launch.
    Parameter order: main RB 0..2 then side RB 0..2, each (w1, b1, w2, b2[, gamma, beta]), then the 1x1 conv's (w, b)."""

    @staticmethod
    def forward(ctx, x, sx, *rest):
        block = rest[-1]
        saved = []
        a, sa, b, sb = x, sx, x, sx
        for i in range(3):
            (a, b), (sa, sb), keep = _rb_forward_multi([block._mainBranch[i], block._sideBranch[i]], [a, b], [sa, sb])
            saved.extend(keep[0])
            saved.extend(keep[1])
        # a * sigmoid(conv1x1(b)) + x as the 1x1 launch's epilogue (like inference), silu(out) for the block that follows from the same
        # launch; the 1x1 output itself is not kept: backward recomputes it in the launch that needs it (MCQ_CONV_GATE_BWD)
        out = ops.conv2d(b, block._sideBranch[3].packed(), gate_mul=a, gate_id=x, dual_silu=True)
        sout = ops.silu_twin(out)
        ctx.save_for_backward(a, b, *saved)
        ctx.block = block
        ctx.mark_non_differentiable(sout)
        ctx.set_materialize_grads(False)
        return out, sout

    @staticmethod
    def backward(ctx, dout, _dsout):
        block = ctx.block
        a, b = ctx.saved_tensors[:2]
        saved = ctx.saved_tensors[2:]
        ns = _rb_nsaved(block._mainBranch[0])
        main = [saved[2 * ns * i: 2 * ns * i + ns] for i in range(3)]
        side = [saved[2 * ns * i + ns: 2 * ns * i + 2 * ns] for i in range(3)]
        dout = dout.contiguous()
        c11 = block._sideBranch[3]
        (h, dbb), = ops.conv2d_gate_bwd([b], [c11.packed()], [a], [dout])     # d a, d (conv1x1 output)
        g = ops.conv2d(dbb, _dgrad_packed(c11, c11.weight))
        dw11, db11 = ops.conv2d_wgrad(b, dbb, 1, 1, want_bias=True)
        pairs = []                                               # appended RB 2, 1, 0; per RB: main (conv1, conv2), side (conv1, conv2)
        norms = {}                                               # RB index -> [(d gamma, d beta) of main, of side]
        for i in (2, 1, 0):
            ng = []
            h, g = _rb_backward_multi([block._mainBranch[i], block._sideBranch[i]], [main[i], side[i]], [h, g], pairs, norm_grads=ng)
            norms[i] = ng
        dx = ops.add3(h, g, dout)
        grads = _wgrads(pairs)
        by_rb = {i: grads[4 * k: 4 * k + 4] for k, i in enumerate((2, 1, 0))}
        flat = []
        for stack in (0, 1):                                     # main RB 0..2, then side RB 0..2
            for i in range(3):
                (dw1, db1), (dw2, db2) = by_rb[i][2 * stack], by_rb[i][2 * stack + 1]
                flat.extend([dw1, db1, dw2, db2])
                if norms[i]:
                    flat.extend(norms[i][stack])
        return (dx, None, *flat, dw11, db11, None)


# ---- same-structure stacks in lockstep ------------------------------------------------------------------------------------
# `latentHead` / `quantizationHead` (RB, AttentionBlock, conv3x3; mcquic/modules/compressor.py:148-160) consume the same z, and
# `dequantizationHead` / `sideHead` (AttentionBlock, conv3x3, RB; :166-175) are independent until their sum: layer by layer the
# two stacks are the same shapes on different tensors, so every layer is ONE multi-problem launch for both (four problems inside
# their AttentionBlocks).  15 + 15 convolutions become 10 launches each way.
def _kind(m) -> str:
    name = type(m).__name__
    if name == "ResidualBlock" and m._skip is None and not getattr(m, "denseNorm", False):
        return "rb"
    if name == "AttentionBlock" and not getattr(m, "denseNorm", False):
        return "attn"
    if name == "Conv2d" and m.kernelSize == 3 and m.stride == 1:
        return "conv"
    raise NotImplementedError(f"lockstep: {name} is not a layer of the paired heads")


def lockstep_compatible(stacks) -> bool:
    try:
        kinds = [[_kind(m) for m in st] for st in stacks]
    except NotImplementedError:
        return False
    return all(k == kinds[0] for k in kinds[1:])


def _lockstep_forward(stacks, xs, keep: bool):
    """Run the k stacks layer by layer; with `keep` returns the tape backward needs."""
    k = len(stacks)
    xs = list(xs)
    tape = []
    nlayers = len(stacks[0])
    for li, layer in enumerate(zip(*stacks)):
        kind = _kind(layer[0])
        if kind == "rb":
            ys, sys_, saved = _rb_forward_multi(list(layer), xs, [_silu_of(x) for x in xs])
            for y, sy in zip(ys, sys_):
                ops.set_silu_twin(y, sy)
            tape.append((kind, layer, saved))
            xs = ys
        elif kind == "attn":
            a, b = list(xs), list(xs)
            sa = [_silu_of(x) for x in xs]
            sb = list(sa)
            saved = []
            for i in range(3):
                blocks = [m._mainBranch[i] for m in layer] + [m._sideBranch[i] for m in layer]
                ys, sys_, sv = _rb_forward_multi(blocks, a + b, sa + sb)
                a, b, sa, sb = ys[:k], ys[k:], sys_[:k], sys_[k:]
                saved.append(sv)
            outs = ops.conv2d_multi(b, [m._sideBranch[3].packed() for m in layer],
                                    per_problem=[dict(gate_mul=ai, gate_id=xi) for ai, xi in zip(a, xs)], dual_silu=True)
            tape.append((kind, layer, (a, b, saved)))
            xs = outs
        else:
            # (a conv3x3 in the middle of a head feeds a ResidualBlock: its launch writes the SiLU twin that block starts from)
            ys = ops.conv2d_multi(xs, [m.packed() for m in layer], dual_silu=li + 1 < nlayers)
            tape.append((kind, layer, xs))
            xs = ys
    return xs, (tape if keep else None)


def _lockstep_backward(tape, dys, grads):
    """Input gradients of the k stacks; parameter gradients go into `grads` (id(parameter) -> tensor)."""
    dys = [d.contiguous() for d in dys]
    pairs, owners = [], []                                     # 3x3 weight-gradient operand pairs and the convs they belong to
    for kind, layer, saved in reversed(tape):
        k = len(layer)
        if kind == "rb":
            before = len(pairs)
            dys = _rb_backward_multi(list(layer), saved, dys, pairs)
            for m in layer:
                owners.extend([m._branch[1], m._branch[3]])
            assert len(pairs) - before == 2 * k
        elif kind == "attn":
            a, b, rbsaved = saved
            c11s = [m._sideBranch[3] for m in layer]
            pairs11 = ops.conv2d_gate_bwd(b, [c.packed() for c in c11s], a, dys)        # all stacks' gates in one launch
            hs = [h for h, _ in pairs11]
            gs = ops.conv2d_multi([dbb for _, dbb in pairs11], [_dgrad_packed(c, c.weight) for c in c11s])
            for c11, bi, (_, dbb) in zip(c11s, b, pairs11):
                dw11, db11 = ops.conv2d_wgrad(bi, dbb, 1, 1, want_bias=True)
                grads[id(c11.weight)], grads[id(c11.bias)] = dw11, db11
            for i in (2, 1, 0):
                blocks = [m._mainBranch[i] for m in layer] + [m._sideBranch[i] for m in layer]
                out = _rb_backward_multi(blocks, rbsaved[i], hs + gs, pairs)
                hs, gs = out[:k], out[k:]
                for blk in blocks:
                    owners.extend([blk._branch[1], blk._branch[3]])
            dys = [ops.add3(h, g, dout) for h, g, dout in zip(hs, gs, dys)]
        else:
            xs = saved
            for m, x, dy in zip(layer, xs, dys):
                pairs.append((x, dy))
                owners.append(m)
            dys = ops.conv2d_multi(dys, [_dgrad_packed(m, m.weight) for m in layer])
    for conv, (dw, db) in zip(owners, _wgrads(pairs)):
        grads[id(conv.weight)] = dw
        if conv.bias is not None:
            grads[id(conv.bias)] = db
    return dys


def _tape_flatten(obj, tensors):
    """Tape -> the same nesting with every tensor replaced by its index into `tensors` (modules and strings stay)."""
    if torch.is_tensor(obj):
        tensors.append(obj)
        return ("#t", len(tensors) - 1)
    if isinstance(obj, (list, tuple)) and not (len(obj) == 2 and obj[0] == "#t"):
        return type(obj)(_tape_flatten(o, tensors) for o in obj)
    return obj


def _tape_restore(spec, tensors):
    if isinstance(spec, tuple) and len(spec) == 2 and spec[0] == "#t":
        return tensors[spec[1]]
    if isinstance(spec, (list, tuple)):
        return type(spec)(_tape_restore(o, tensors) for o in spec)
    return spec


class LockstepFn(torch.autograd.Function):
    """k same-structure stacks (nn.Sequential of ResidualBlock / AttentionBlock / conv3x3) on k inputs as ONE autograd node.
    apply(x_1 .. x_k, *parameters (all stacks', in `stack.parameters()` order), stacks) -> (y_1 .. y_k).
    The taped activations go through ctx.save_for_backward (version-checked, released with the graph, readable by a second
    backward under retain_graph=True); ctx keeps only the tape's structure."""

    @staticmethod
    def forward(ctx, *args):
        stacks, shared = args[-1]
        k = len(stacks)
        ctx.shared = shared
        # (`shared`: ONE input feeds all k stacks -- latentHead / quantizationHead on the same z: its k gradients are summed here,
        #  by this library's add, instead of by the autograd engine)
        ys, tape = _lockstep_forward(stacks, args[:1] * k if shared else args[:k], keep=True)
        tensors = []
        ctx.tape_spec = _tape_flatten([(kind, saved) for kind, _, saved in tape], tensors)
        ctx.layers = [layer for _, layer, _ in tape]
        ctx.save_for_backward(*tensors)
        ctx.stacks, ctx.k = stacks, k
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        grads = {}
        entries = _tape_restore(ctx.tape_spec, ctx.saved_tensors)
        tape = [(kind, layer, saved) for (kind, saved), layer in zip(entries, ctx.layers)]
        dxs = _lockstep_backward(tape, list(dys), grads)
        params = [p for st in ctx.stacks for p in st.parameters()]
        if ctx.shared:
            dx = dxs[0]
            for i in range(1, len(dxs) - 1, 2):
                dx = ops.add3(dx, dxs[i], dxs[i + 1])
            if len(dxs) % 2 == 0:
                dx = ops.add(dx, dxs[-1])
            dxs = [dx]
        return (*dxs, *[grads.get(id(p)) for p in params], None)


def lockstep(stacks, xs):
    """Training graph: the stacks in lockstep (LockstepFn); falls back to one after the other when their structures differ."""
    if not (ops._MULTI and lockstep_compatible(stacks)):
        return [st(x) for st, x in zip(stacks, xs)]
    params = [p for st in stacks for p in st.parameters()]
    shared = len(xs) > 1 and all(x is xs[0] for x in xs[1:])
    return list(LockstepFn.apply(*(xs[:1] if shared else xs), *params, (tuple(stacks), shared)))


class GateFn(torch.autograd.Function):
    """out = a * sigmoid(b) + x."""

    @staticmethod
    def forward(ctx, a, b, x):
        ctx.save_for_backward(a, b)
        out = ops.gate(a, b, x, dual_silu=True)                  # (+ silu(out): the block that follows starts with an activation)
        sout = ops.silu_twin(out)
        ctx.mark_non_differentiable(sout)
        ctx.set_materialize_grads(False)
        return out, sout

    @staticmethod
    def backward(ctx, dout, _dsout=None):
        a, b = ctx.saved_tensors
        dout = dout.contiguous()
        da, db = ops.gate_bwd(a, b, dout)
        return da, db, dout


class AxpbyFn(torch.autograd.Function):
    """out = alpha * a + beta * b (alpha, beta in {+1, -1})."""

    @staticmethod
    def forward(ctx, a, b, alpha, beta, dual_silu=False):
        ctx.alpha, ctx.beta = alpha, beta
        out = ops.axpby(a, b, alpha, beta, dual_silu=bool(dual_silu))
        sout = ops.silu_twin(out)
        if sout is not None:
            ctx.mark_non_differentiable(sout)
        ctx.set_materialize_grads(False)
        return out, sout

    @staticmethod
    def backward(ctx, dout, _dsout=None):
        if dout is None:
            return None, None, None, None, None
        dout = dout.contiguous()
        da = dout if ctx.alpha == 1.0 else ops.axpby(dout, dout, ctx.alpha, 0.0)
        db = dout if ctx.beta == 1.0 else ops.axpby(dout, dout, ctx.beta, 0.0)
        return da, db, None, None, None


class LowerBoundFn(torch.autograd.Function):
    """max(x, bound) with the reference's gradient rule (mcquic/nn/base.py:17-29): the gradient passes where
    x >= bound or where it pushes x up.  Parameter-sized tensors (C, C x C): plain torch ops."""

    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.max(x, bound)

    @staticmethod
    def backward(ctx, g):
        x, bound = ctx.saved_tensors
        return ((x >= bound) | (g < 0)).to(g.dtype) * g, None


def _gdn_bounds(module):
    """(beta bound, beta pedestal, gamma bound, gamma pedestal) as floats, read from the module's buffers ONCE (a float() of a
    device buffer is a host sync, and illegal inside a captured hipGraph); the buffers are constants of the layer."""
    cached = module.__dict__.get("_boundsCache")
    key = (module.beta_reparam.lowerBound.bound.data_ptr(), module.gamma_reparam.lowerBound.bound.data_ptr())
    if cached is None or cached[0] != key:
        vals = (float(module.beta_reparam.lowerBound.bound), float(module.beta_reparam.eps),
                float(module.gamma_reparam.lowerBound.bound), float(module.gamma_reparam.eps))
        cached = module.__dict__["_boundsCache"] = (key, vals)
    return cached[1]


def _gdn_operands(module, beta_p, gamma_p, bounds):
    """(forward operand stream of beta + gamma @ x^2, operand stream of 2 gamma^T for the input gradient) of a GDN layer in
    the training graph, folded and packed once per parameter version (4 launches) like a convolution's streams."""
    key = (ops.tensor_version(beta_p), beta_p.data_ptr(), ops.tensor_version(gamma_p), gamma_p.data_ptr())
    cached = module.__dict__.get("_trainOperands")
    if cached is None or cached[0] != key:
        bb, be, gb, ge = bounds
        beta = ops.nonneg_reparam(beta_p, bb, be)
        gamma = ops.nonneg_reparam(gamma_p, gb, ge)
        packed = ops.PackedConv(gamma[..., None, None], beta, copy_bias=False)
        back = ops.PackedConv.dgrad(gamma[..., None, None], 1, scale=2.0)      # 2 gamma^T, packed in one launch
        cached = module.__dict__["_trainOperands"] = (key, packed, back)
    return cached[1], cached[2]


def refresh_gdn_operands(modules) -> int:
    """Bring the training operands of all GDN / IGDN `modules` whose parameters changed up to date in grouped launches: ONE launch
    for every stale layer's two re-parametrisations, one grouped pack for the forward streams, one for the input-gradient streams
    (2 gamma^T) -- what `_gdn_operands` does layer by layer with 4 launches each (40 per step of the qp = 2 model after an optimizer
    update).  Everything is written in place: the folded tensors and both operand streams keep their addresses (a captured step holds
    them).  Returns the number of layers refreshed.  Layers of one channel count share the launches."""
    stale = []
    for m in modules:
        key = (ops.tensor_version(m.beta), m.beta.data_ptr(), ops.tensor_version(m.gamma), m.gamma.data_ptr())
        cached = m.__dict__.get("_trainOperands")
        if cached is None or cached[0] != key:
            stale.append((m, key))
    if not stale:
        return 0
    by_shape = {}
    for m, key in stale:
        by_shape.setdefault((tuple(m.gamma.shape), m.gamma.device), []).append((m, key))
    for group in by_shape.values():
        ps, outs, bounds, peds = [], [], [], []
        for m, _ in group:
            fold = m.__dict__.get("_trainFold")
            if fold is None or fold[0].device != m.beta.device:
                fold = m.__dict__["_trainFold"] = (torch.empty_like(m.beta.detach()), torch.empty_like(m.gamma.detach()))
            bb, be, gb, ge = _gdn_bounds(m)
            ps += [m.beta, m.gamma]
            outs += [fold[0], fold[1]]
            bounds += [bb, gb]
            peds += [be, ge]
        ops.nonneg_reparam_multi_(ps, outs, bounds, peds)
        gammas = [m.__dict__["_trainFold"][1][..., None, None] for m, _ in group]
        betas = [m.__dict__["_trainFold"][0] for m, _ in group]
        old = [m.__dict__.get("_trainOperands") for m, _ in group]
        fwd = ops.pack_convs(gammas, betas, into=[None if o is None else o[1] for o in old])
        back = ops.pack_convs(gammas, dgrad=True, stride=1, scale=2.0, into=[None if o is None else o[2] for o in old])
        for (m, key), f, b in zip(group, fwd, back):
            m.__dict__["_trainOperands"] = (key, f, b)
    return len(stale)


def _gdn_forward(module, x, beta_p, gamma_p, inverse: bool):
    """(y, forward operand stream, input-gradient operand stream, (beta bound, gamma bound)) of a GDN / IGDN layer."""
    bb, be, gb, ge = _gdn_bounds(module)
    packed, back = _gdn_operands(module, beta_p, gamma_p, (bb, be, gb, ge))
    y = ops.conv2d(x, packed, square_in=True, igdn_mul=x) if inverse else ops.conv2d(x, packed, square_in=True, gdn_mul=x)
    return y, packed, back, (bb, gb)


def _gdn_backward(x, packed, back, bounds, beta_p, gamma_p, dy, inverse: bool):
    """(dx, d beta_p, d gamma_p): 5 launches -- the recomputed s-launch with the element-wise part in its epilogue, the 1x1
    input-gradient launch, the 1x1 weight gradient on x^2 (+ its reduce) and both re-parametrisation gradients in one."""
    if _GDN_BWD_FUSED:
        dxd, ds = ops.conv2d_gdn_bwd(x, packed, dy, inverse)      # s = beta + gamma @ x^2 recomputed, dy f(s) and dy x f'(s) from its epilogue
    else:                                                          # (A/B switch: the round-3 form, s stored and read back)
        dxd, ds = ops.gdn_bwd_prep(x, ops.conv2d(x, packed, square_in=True), dy, inverse)
    dx = ops.conv2d(ds, back, mul=x, res=dxd)                      # dy f(s) + 2 x (gamma^T ds)
    with ops.wgrad_now():                                          # (read right below: no deferred reduce for this one)
        dgamma, dbeta = ops.conv2d_wgrad(x, ds, 1, 1, square_x=True, want_bias=True)
    dbeta_p, dgamma_p = ops.nonneg_reparam_bwd2(beta_p, dbeta, bounds[0], gamma_p, dgamma[:, :, 0, 0], bounds[1])
    return dx, dbeta_p, dgamma_p


class GdnFn(torch.autograd.Function):
    """y = x * f(beta + gamma @ x^2) with the non-negative re-parametrisation of beta [C], gamma [C, C] INSIDE the node
    (mcquic/nn/gdn.py:67-91, mcquic/nn/base.py:17-29,81-84): folding, the 1x1 launch, and in backward the two input-gradient
    pieces, gamma's and beta's gradients (one 1x1 weight-gradient launch on x^2) and the re-parametrisation's own gradient
    rule are HIP launches -- no ATen max / pow / compare / mask glue (it was ~1 ms of a 28 ms training step)."""

    @staticmethod
    def forward(ctx, x, beta_p, gamma_p, module, inverse):
        y, packed, back, bounds = _gdn_forward(module, x, beta_p, gamma_p, inverse)
        ctx.save_for_backward(x, beta_p, gamma_p)
        ctx.packed, ctx.back, ctx.inverse, ctx.bounds = packed, back, inverse, bounds
        return y

    @staticmethod
    def backward(ctx, dy):
        x, beta_p, gamma_p = ctx.saved_tensors
        dx, dbeta_p, dgamma_p = _gdn_backward(x, ctx.packed, ctx.back, ctx.bounds, beta_p, gamma_p, dy.contiguous(), ctx.inverse)
        return dx, dbeta_p, dgamma_p, None, None


class ScaleBlockFn(torch.autograd.Function):
    """ResidualBlockWithStride (`up` False: SiLU, conv3 s2, GDN, conv3, + conv3 s2 skip; mcquic/nn/blocks.py:98-122) and
    ResidualBlockShuffle (`up` True: SiLU, pixelShuffle3x3, IGDN, conv3, + pixelShuffle3x3 skip; :141-159) as ONE autograd node:
    forward = the four launches of the op-by-op graph; backward keeps those launches too, except that the gradient reaching x
    along the branch (through the first convolution and the SiLU) and the one along the skip convolution MEET IN THE EPILOGUE of
    the branch's input-gradient launch (* silu'(x) + d_skip; through the PixelShuffle store for the strided block) instead of in a
    SiLU-backward launch plus an add -- one launch and three tensor passes less per block.
    Parameter order: conv1 (w, b), GDN (beta, gamma), conv2 (w, b), skip (w, b)."""

    @staticmethod
    def forward(ctx, x, sx, w1, b1, beta_p, gamma_p, w2, b2, ws, bs, block, up):
        c1, gdn_m, c2, cs = _scale_block_layers(block, up)
        stride = 1 if up else 2
        t = ops.conv2d(sx, c1.packed(), stride, shuffle2=up)
        u, packed, back, bounds = _gdn_forward(gdn_m, t, beta_p, gamma_p, up)
        skip = ops.conv2d(x, cs.packed(), stride, shuffle2=up)
        y = ops.conv2d(u, c2.packed(), res=skip, dual_silu=True)       # (+ silu(y) for the block that follows)
        sy = ops.silu_twin(y)
        ctx.save_for_backward(x, sx, t, u, beta_p, gamma_p, w1, w2, ws)
        ctx.block, ctx.up, ctx.packed, ctx.back, ctx.bounds = block, up, packed, back, bounds
        ctx.mark_non_differentiable(sy)
        ctx.set_materialize_grads(False)
        return y, sy

    @staticmethod
    def backward(ctx, dy, _dsy):
        x, sx, t, u, beta_p, gamma_p, w1, w2, ws = ctx.saved_tensors
        up = ctx.up
        c1, gdn_m, c2, cs = _scale_block_layers(ctx.block, up)
        stride = 1 if up else 2
        dy = dy.contiguous()
        need_dx = ctx.needs_input_grad[0]
        d_u = ops.conv2d(dy, _dgrad_packed(c2, w2))
        dw2, db2 = ops.conv2d_wgrad(u, dy, 3, 1, want_bias=True)
        dyc = ops.pixel_unshuffle2(dy) if up else dy
        dx_skip = None
        if need_dx:
            dx_skip = ops.conv2d(dyc, _dgrad_packed(cs, ws)) if up else _crop_like(ops.conv2d(dyc, _dgrad_packed(cs, ws), shuffle2=True), x)
        dws, dbs = ops.conv2d_wgrad(x, dyc, 3, stride, want_bias=True)
        d_t, dbeta_p, dgamma_p = _gdn_backward(t, ctx.packed, ctx.back, ctx.bounds, beta_p, gamma_p, d_u, up)
        d_tc = ops.pixel_unshuffle2(d_t) if up else d_t
        dx = None
        if need_dx:
            wt1 = _dgrad_packed(c1, w1)
            if up:
                dx = ops.conv2d(d_tc, wt1, dsilu_mul=x, res=dx_skip)
            elif x.shape[-2] == 2 * d_tc.shape[-2] and x.shape[-1] == 2 * d_tc.shape[-1]:
                dx = ops.conv2d(d_tc, wt1, shuffle2=True, dsilu_mul=x, res=dx_skip)
            else:                                                  # (odd maps: the sub-pixel form over-covers by a row / column)
                dx = ops.silu_bwd(x, _crop_like(ops.conv2d(d_tc, wt1, shuffle2=True), x), dx_skip)
        dw1, db1 = ops.conv2d_wgrad(sx, d_tc, 3, stride, want_bias=True)
        return dx, None, dw1, db1, dbeta_p, dgamma_p, dw2, db2, dws, dbs, None, None


def _scale_block_layers(block, up: bool):
    """(first conv, GDN / IGDN, closing conv, skip conv) of a strided / shuffle block (the shuffle convs sit inside their Sequential)."""
    c1, cs = block._branch[1], block._skip
    if up:
        c1, cs = c1[0], cs[0]
    return c1, block._branch[2], block._branch[3], cs


def _crop_like(t: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    if t.shape[-2] != x.shape[-2] or t.shape[-1] != x.shape[-1]:
        t = t[..., :x.shape[-2], :x.shape[-1]].contiguous()
    return t


def scale_block(x, block, up: bool):
    """ResidualBlockWithStride / ResidualBlockShuffle in the training graph (ScaleBlockFn); the result carries its SiLU twin."""
    c1, gdn_m, c2, cs = _scale_block_layers(block, up)
    y, sy = ScaleBlockFn.apply(x, _silu_of(x), c1.weight, c1.bias, gdn_m.beta, gdn_m.gamma, c2.weight, c2.bias, cs.weight, cs.bias, block, up)
    ops.set_silu_twin(y, sy)
    return y


class GroupNormFn(torch.autograd.Function):
    """nn.GroupNorm(groups, C) with affine parameters (`denseNorm=True`, mcquic/nn/blocks.py:179-200), HIP both ways."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps):
        y, mean, rstd = ops.group_norm(x, weight, bias, groups, eps, want_stats=True)
        ctx.save_for_backward(x, weight, mean, rstd)
        ctx.groups = groups
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, mean, rstd = ctx.saved_tensors
        dx, dw, db = ops.group_norm_bwd(x, dy.contiguous(), weight, mean, rstd, ctx.groups,
                                        want_params=ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        return dx, dw, db, None, None


def group_norm(x, module):
    return GroupNormFn.apply(x, module.weight, module.bias, module.num_groups, module.eps)


def conv(x, module, res: Optional[torch.Tensor] = None, shuffle2: bool = False, dual_silu: bool = False):
    """`dual_silu`: the launch also stores silu(y), which the result then carries as its twin (the strided / shuffle blocks'
    closing convolutions feed blocks that start with an activation: one stand-alone SiLU launch less per block)."""
    y, sy = ConvFn.apply(x, module.weight, module.bias, res, module, shuffle2, dual_silu)
    if sy is not None:
        ops.set_silu_twin(y, sy)
    return y


def silu(x):
    return SiluFn.apply(x)


def _silu_of(x: torch.Tensor) -> torch.Tensor:
    """silu(x) as a plain tensor: the producer's twin if there is one, else one SiLU launch (remembered on x)."""
    sx = ops.silu_twin(x)
    if sx is None:
        with torch.no_grad():
            sx = ops.silu(x.detach())
        ops.set_silu_twin(x, sx)
    return sx


def attention_block(x, block):
    """AttentionBlock in the training graph (AttentionBlockFn)."""
    params = []
    for stack in (block._mainBranch, block._sideBranch):
        for i in range(3):
            params.extend(_rb_params(stack[i]))
    c11 = block._sideBranch[3]
    out, sout = AttentionBlockFn.apply(x, _silu_of(x), *params, c11.weight, c11.bias, block)
    ops.set_silu_twin(out, sout)
    return out


def residual_block(x, block):
    """ResidualBlock in the training graph (ResidualBlockFn); the result carries silu(result) as its twin."""
    y, sy = ResidualBlockFn.apply(x, _silu_of(x), *_rb_params(block), block)
    ops.set_silu_twin(y, sy)
    return y


def gate(a, b, x):
    return _with_twin(*GateFn.apply(a, b, x))


def _with_twin(out, sout):
    if sout is not None:
        ops.set_silu_twin(out, sout)
    return out


def add(a, b, dual_silu: bool = False):
    """a + b; `dual_silu`: the result carries silu(result) as its twin (its consumer starts with an activation)."""
    return _with_twin(*AxpbyFn.apply(a, b, 1.0, 1.0, dual_silu))


def sub(a, b):
    return _with_twin(*AxpbyFn.apply(a, b, 1.0, -1.0, False))


def sub_pass(a, b):
    """(a - b, b'): b' is b for its second consumer (SubPassFn); both results keep / carry their SiLU twins."""
    tb = ops.silu_twin(b)
    r, b2, sr = SubPassFn.apply(a, b)
    ops.set_silu_twin(r, sr)
    if tb is not None:
        ops.set_silu_twin(b2, tb)
    return r, b2


def silu_fork(x):
    """(silu(x), x') for a block whose skip path reads x itself (SiluForkFn)."""
    tx = ops.silu_twin(x)
    sx, x2 = SiluForkFn.apply(x)
    if tx is not None:
        ops.set_silu_twin(x2, tx)
    return sx, x2


def gdn(x, module, inverse: bool):
    """GenDivNorm / InvGenDivNorm with the reference's re-parametrisation inside the graph (nn/base.py:81-84)."""
    return GdnFn.apply(x, module.beta, module.gamma, module, inverse)


class SoftQuantizeFn(torch.autograd.Function):
    """One level of the training quantizer (reference: _multiCodebookQuantization.forward + the two
    _multiCodebookDeQuantization.forward calls on its sample, quantizer.py:181-239,262-274):
        logit = (-dist / sqrt(k)) * max(T, eps) ; random drop ; sample = gumbelSoftmax(logit, hard=True)
        deq   = sample @ codebook                       (differentiable output; used by the residual AND the decoder)
        logits                                          (differentiable output, as in the reference: a regulariser on them
                                                         reaches the latents, codebook and temperature)
        code  = argmax(logit)                           (non-differentiable)
    Backward: straight-through -- the gradient reaches `sample` through y_soft only -- then through the logits to the
    latent, the codebook (distance terms + the sample @ codebook product) and the temperature; a gradient on the returned
    logits joins in front of `_logit` (the shipped losses, mcquic/loss/__init__.py:47-62, never produce one: then nothing
    extra is launched)."""

    @staticmethod
    def forward(ctx, x, codebook, temperature, freq_ema, u_drop, u_gumbel, drop_exponent, packed, bound, rng=None, counts=None):
        logits = ops.vq_logits(x, packed, temperature, bound)
        code, index, hot = ops.vq_gumbel_sample(logits, u_drop, u_gumbel, freq_ema, drop_exponent, rng, counts)
        deq = ops.vq_dequant_soft(index, hot, packed, dual_silu=True)          # (+ silu(deq): the decoder side starts with an activation)
        sdeq = ops.silu_twin(deq)
        ctx.save_for_backward(x, logits, u_gumbel, index, hot, temperature, rng)     # (u_gumbel None: remade from `rng` in backward)
        ctx.packed, ctx.bound = packed, bound
        ctx.mark_non_differentiable(code, sdeq)
        ctx.set_materialize_grads(False)        # (an unused `dlogits` would be a zero fill of the [n, m, h, w, k] logits: 134 MB at level 0)
        return deq, code, logits, sdeq

    @staticmethod
    def backward(ctx, ddeq, _dcode, dlogits, _dsdeq=None):
        x, logits, u_gumbel, index, hot, temperature, rng = ctx.saved_tensors
        packed = ctx.packed
        if ddeq is None and dlogits is None:
            return (None,) * 11
        ddeq = torch.zeros_like(x) if ddeq is None else ddeq.contiguous()
        ds = ops.vq_inner(ddeq, packed)                                        # dSample = dDeq . C^T
        raw = None
        if dlogits is not None:                                                # the logits before the random drop, recomputed
            dlogits = dlogits.contiguous()
            raw = ops.vq_logits(x, packed, temperature, ctx.bound)
        rowsum, dtrow = ops.vq_softmax_bwd(logits, u_gumbel, ds, temperature, ctx.bound, dlogits, raw, rng)   # ds now holds d dist
        dx, dcb = ops.vq_soft_bwd(ds, rowsum, x, ddeq, index, hot, packed)
        dt = ops.vq_temperature_grad(dtrow, temperature, ctx.bound)            # [m, 1, 1, 1]: the rows' terms summed, LowerBound's rule
        return dx, dcb, dt, None, None, None, None, None, None, None, None
