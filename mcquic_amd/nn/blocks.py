"""Residual / attention blocks of the Compressor path, fused onto the HIP conv kernel.

Same classes, constructor arguments and state_dict keys as the reference (mcquic/nn/blocks.py):
`ResidualBlock` (:162-200), `ResidualBlockWithStride` (:81-122), `ResidualBlockShuffle` (:124-159),
`AttentionBlock` (:245-288).  What the reference runs as separate torch kernels (SiLU, `out += identity`,
GDN's multiply, `a * sigmoid(b) + x`) rides in the prologue / epilogue of the conv launches:

Every block's closing launch also stores silu(out) beside out (`dual_silu`): the next block's act1 is then
computed once per element in an epilogue instead of once per tap inside the consumer's k-loop.

    ResidualBlock            2 launches   conv(silu_in, silu_out) ; conv(+ x)
    ResidualBlockWithStride  4 launches   conv s2(silu_in) ; GDN 1x1 ; skip conv s2 ; conv(+ skip)
    ResidualBlockShuffle     4 launches   conv+shuffle(silu_in) ; IGDN 1x1 ; skip conv+shuffle ; conv(+ skip)
    AttentionBlock          13 launches   6 ResidualBlocks ; 1x1 conv with the gate a*sigmoid(b)+x as its epilogue
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
from torch import nn

from .. import autograd as AG
from .. import ops
from .convs import conv1x1, conv3x3, pixelShuffle3x3
from .gdn import GenDivNorm, InvGenDivNorm

__all__ = ["ResidualBlockWithStride", "ResidualBlockShuffle", "ResidualBlock", "AttentionBlock"]

# Independent branches (AttentionBlock's main / side stacks, a strided block's skip conv) are enqueued on a second
# HIP stream: the tail of one branch's kernel (a launch is only ~6 rounds of waves at 192x128) is filled by the other
# branch's workgroups, and the launch-latency-bound small levels run two kernels at once.  MCQUIC_AMD_BRANCH_STREAMS=0
# turns it off (single stream, same results).
_BRANCH_STREAMS = os.environ.get("MCQUIC_AMD_BRANCH_STREAMS", "1") != "0"
_NORM_NODES = os.environ.get("MCQUIC_AMD_NORM_NODES", "1") != "0"        # A/B switch: 0 = `denseNorm` blocks op by op in the training graph
_BLOCK_NODES = os.environ.get("MCQUIC_AMD_BLOCK_NODES", "1") != "0"      # A/B switch: 0 = strided / shuffle blocks op by op in the training graph
_side_streams: Dict[tuple, "torch.cuda.Stream"] = {}
_MULTI_MAX_PIXELS = int(os.environ.get("MCQUIC_AMD_MULTI_MAX_PIXELS", str(64 * 1024)))   # N * H * W up to which AttentionBlock stacks share launches
# ... and below this many pixels nothing is forked at all: a launch that does not fill the chip for several rounds has no tail worth
# filling, and forks inside a captured hipGraph cost more than they give.  Round 4 (tools/probes/batch1_order.py): one 768x512 image
# as graph replays, 5.64-5.67 ms without forks in every order of events; with them 5.87 ms when captured in a fresh process and
# 6.6-6.7 ms when captured after eager 32-image steps had created their side streams (the bench line's 6.9 vs the stand-alone 5.9
# of round 3), eager anywhere between 5.9 and 12.7.  At 32 images the forked maps (96x64 and up: 196 k pixels) are unaffected.
_FORK_MIN_PIXELS = int(os.environ.get("MCQUIC_AMD_FORK_MIN_PIXELS", str(128 * 1024)))


def _side_stream(main: "torch.cuda.Stream") -> "torch.cuda.Stream":
    """The side stream paired with `main` (one per main stream, so pipelined sub-batches do not share one)."""
    key = (main.device.index, main.cuda_stream)
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=main.device)
    return st


class _fork:
    """`with _fork(x) as f:` runs the body on the side stream after everything enqueued so far; `f.join(t)` makes the
    main stream wait for it and hands tensor `t` (allocated on the side stream) over to the main stream."""

    def __init__(self, x: torch.Tensor):
        self.on = _BRANCH_STREAMS and x.is_cuda and x.shape[0] * x.shape[-2] * x.shape[-1] >= _FORK_MIN_PIXELS
        if self.on:
            self.main = torch.cuda.current_stream(x.device)
            self.side = _side_stream(self.main)
        self.ctx = None

    def __enter__(self):
        if self.on:
            self.side.wait_stream(self.main)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False

    def join(self, t: torch.Tensor) -> torch.Tensor:
        if self.on:
            self.main.wait_stream(self.side)
            t.record_stream(self.main)
        return t


class _residulBlock(nn.Module):
    """Container with the reference's layout: `_branch` = Sequential(act1, conv1, act2, conv2), `_skip`
    (reference: mcquic/nn/blocks.py:62-78)."""

    def __init__(self, act1: nn.Module, conv1: nn.Module, act2: nn.Module, conv2: nn.Module, skip: Optional[nn.Module]):
        super().__init__()
        self._branch = nn.Sequential(act1, conv1, act2, conv2)
        self._skip = skip


class GroupNorm(nn.GroupNorm):
    """nn.GroupNorm(groups, C) (affine, eps 1e-5: same parameters `weight`, `bias` and state_dict keys) on the HIP kernel of
    csrc/norm.hip.  In this snapshot of the reference the blocks' `groups` argument only sets the group count of this layer,
    which `denseNorm=True` puts in place of a ResidualBlock's second activation (mcquic/nn/blocks.py:179-200: conv3x3 is called
    without `groups`): every convolution stays dense."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training and torch.is_grad_enabled():
            return AG.group_norm(x, self)
        return ops.group_norm(x, self.weight, self.bias, self.num_groups, self.eps)


class ResidualBlock(_residulBlock):
    """SiLU, conv3, SiLU, conv3, + x; with different widths the skip is a 1x1 conv (mcquic/nn/blocks.py:179-182).
    `denseNorm=True`: GroupNorm(groups, outChannels) stands where the second SiLU was (:196)."""

    def __init__(self, inChannels: int, outChannels: int, groups: int = 1, denseNorm: bool = False):
        super().__init__(nn.SiLU(), conv3x3(inChannels, outChannels), GroupNorm(groups, outChannels) if denseNorm else nn.SiLU(),
                         conv3x3(outChannels, outChannels), conv1x1(inChannels, outChannels) if inChannels != outChannels else None)
        self.denseNorm = bool(denseNorm)

    def forward(self, x: torch.Tensor, res2: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.training and torch.is_grad_enabled():
            if self._skip is None and (_NORM_NODES or not self.denseNorm):      # training graph: the block as one autograd node
                return AG.residual_block(x, self)
            sx, x = AG.silu_fork(x)                               # (width-changing / normalised blocks: op-by-op autograd; x's two
            t = self._branch[1](sx)                               #  gradients -- through the activation and along the skip -- meet in one launch)
            t = self._branch[2](t) if self.denseNorm else AG.silu(t)
            return self._branch[3](t, res=x if self._skip is None else self._skip(x), dual_silu=True)
        if self.denseNorm:
            t = self._branch[2](self._branch[1](x, silu_in=True))     # GroupNorm(conv1(silu(x))): no activation in front of conv2
        else:
            t = self._branch[1](x, silu_in=True, silu_out=True)  # silu(conv1(silu(x)))
        identity = x if self._skip is None else self._skip(x)
        return self._branch[3](t, res=identity, dual_silu=True)   # conv2(.) + identity


class ResidualBlockWithStride(_residulBlock):
    """SiLU, conv3 s2, GDN, conv3, + conv3 s2 skip."""

    def __init__(self, inChannels: int, outChannels: int, stride: int = 2, groups: int = 1, denseNorm: bool = False):
        # (`groups` / `denseNorm` are accepted and unused, exactly like the reference's strided / shuffle blocks, :98-159)
        if stride != 2:
            raise NotImplementedError("only stride-2 ResidualBlockWithStride is on the path")
        super().__init__(nn.SiLU(), conv3x3(inChannels, outChannels, stride=stride), GenDivNorm(outChannels),
                         conv3x3(outChannels, outChannels), conv3x3(inChannels, outChannels, stride=stride))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training and torch.is_grad_enabled():
            if _BLOCK_NODES and self._branch[1].bias is not None:
                return AG.scale_block(x, self, up=False)                       # the whole block as one autograd node
            sx, x = AG.silu_fork(x)                                           # (the branch's and the skip's gradients meet in one launch)
            t = self._branch[2](self._branch[1](sx))
            return self._branch[3](t, res=self._skip(x), dual_silu=True)      # (+ silu(out) for the block that follows)
        with _fork(x) as f:
            identity = self._skip(x)
        c1, gdn = self._branch[1], self._branch[2]
        fuse = ops.post_ok(x, c1.packed(), c1.stride, "gdn")
        if fuse:
            t = c1(x, silu_in=True, post_gdn=gdn.packed_post(), tile=0 if fuse == 1 else fuse)      # the normalisation inside the strided convolution's launch
        else:
            t = gdn(c1(x, silu_in=True))
        return self._branch[3](t, res=f.join(identity), dual_silu=True)


class ResidualBlockShuffle(_residulBlock):
    """SiLU, pixelShuffle3x3 (x2), IGDN, conv3, + pixelShuffle3x3 skip."""

    def __init__(self, inChannels: int, outChannels: int, upsample: int = 2, groups: int = 1, denseNorm: bool = False):
        if upsample != 2:
            raise NotImplementedError("only 2x ResidualBlockShuffle is on the path")
        super().__init__(nn.SiLU(), pixelShuffle3x3(inChannels, outChannels, upsample), InvGenDivNorm(outChannels),
                         conv3x3(outChannels, outChannels), pixelShuffle3x3(inChannels, outChannels, upsample))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training and torch.is_grad_enabled():
            if _BLOCK_NODES and self._branch[1][0].bias is not None:
                return AG.scale_block(x, self, up=True)
            sx, x = AG.silu_fork(x)
            t = self._branch[2](self._branch[1](sx))
            return self._branch[3](t, res=self._skip(x), dual_silu=True)
        with _fork(x) as f:
            identity = self._skip(x)
        up, igdn = self._branch[1][0], self._branch[2]
        fuse = ops.post_ok(x, up.packed(), 1, "igdn", shuffle2=True)
        if fuse:
            # the inverse normalisation inside the up-sampling convolution's launch: row tiles in sub-pixel-major order, so that a wave
            # holds the 128 channels the normalisation mixes; the PixelShuffle is still only the store pattern
            t = ops.conv2d(x, up.packed_subpixel(), 1, silu_in=True, shuffle2=True, post_igdn=igdn.packed_post(), tile=0 if fuse == 1 else fuse)
        else:
            t = igdn(self._branch[1](x, silu_in=True))
        return self._branch[3](t, res=f.join(identity), dual_silu=True)


class AttentionBlock(nn.Module):
    """a = RB^3(x); b = conv1x1(RB^3(x)); out = a * sigmoid(b) + x (reference: blocks.py:245-288)."""

    def __init__(self, channel: int, groups: int = 1, denseNorm: bool = False):
        super().__init__()
        self._mainBranch = nn.Sequential(*[ResidualBlock(channel, channel, groups, denseNorm) for _ in range(3)])
        self._sideBranch = nn.Sequential(*[ResidualBlock(channel, channel, groups, denseNorm) for _ in range(3)],
                                         conv1x1(channel, channel))
        self.denseNorm = bool(denseNorm)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training and torch.is_grad_enabled():
            if self.denseNorm and not _NORM_NODES:                # (A/B: normalised blocks op by op)
                xa, xb, xc = AG.fork(x, 3)                        # (three consumers: their gradients summed by one launch of ours)
                return AG.gate(self._mainBranch(xa), self._sideBranch(xb), xc)
            return AG.attention_block(x, self)
        if ops._MULTI and not self.denseNorm and x.shape[0] * x.shape[2] * x.shape[3] <= _MULTI_MAX_PIXELS:
            # the two stacks apply the same layer shapes to different tensors: layer by layer they share a launch
            # (mcq_conv2d_multi_f32) -- twice the workgroups per launch, half the launches.  Measured on one MI355X: batch 1
            # with hipGraphs 7.40 -> 7.09 ms per encode+decode; at batch 32 the big maps do better with the stacks on two
            # streams (257.1 vs 255.7 images/s), so maps above _MULTI_MAX_PIXELS keep the fork below
            a = b = x
            for i in range(3):
                m, sd = self._mainBranch[i]._branch, self._sideBranch[i]._branch
                ta, tb = ops.conv2d_multi([a, b], [m[1].packed(), sd[1].packed()], silu_in=True, silu_out=True)
                a, b = ops.conv2d_multi([ta, tb], [m[3].packed(), sd[3].packed()], per_problem=[dict(res=a), dict(res=b)], dual_silu=True)
            return self._sideBranch[3](b, gate_mul=a, gate_id=x, dual_silu=True)
        last, gate = self._sideBranch[2], self._sideBranch[3]
        fuse = 0 if (self.denseNorm or last._skip is not None) else ops.post_ok(x, last._branch[3].packed(), 1, "gate")
        if fuse:
            # conv1x1 + gate inside the launch of the side stack's LAST convolution (the tile it has just finished is the 1x1 layer's
            # input: `b` is never stored); that launch needs `a`, so it runs on the main stream behind the main stack
            with _fork(x) as f:
                b = x
                for i in range(2):
                    b = self._sideBranch[i](b)
                t = last._branch[1](b, silu_in=True, silu_out=True)
            a = self._mainBranch(x)
            f.join(b)
            return ops.conv2d(f.join(t), last._branch[3].packed(), 1, res=b, post_gate=gate.packed_post(), gate_mul=a, gate_id=x, dual_silu=True, tile=0 if fuse == 1 else fuse)
        with _fork(x) as f:
            b = x
            for i in range(3):
                b = self._sideBranch[i](b)
        a = self._mainBranch(x)
        return self._sideBranch[3](f.join(b), gate_mul=a, gate_id=x, dual_silu=True)


def run_stack(stack, x: torch.Tensor) -> torch.Tensor:
    """`stack(x)` for an nn.Sequential of this package's layers in the TRAINING graph, with one look-ahead: a plain convolution
    that feeds a block starting with an activation also stores silu(.) (its launch's twin output) -- what lockstep does for the
    paired heads; a stack run on its own (the last level's dequantizationHead: AttentionBlock, conv3x3, ResidualBlock) otherwise
    pays a stand-alone SiLU launch."""
    mods = list(stack)
    for i, m in enumerate(mods):
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        if type(m).__name__ == "Conv2d" and isinstance(nxt, (_residulBlock, AttentionBlock)) and m.training and torch.is_grad_enabled():
            x = m(x, dual_silu=True)
        else:
            x = m(x)
    return x


# ---- same-structure stacks in lockstep (inference) -----------------------------------------------------------------------------
# `latentHead` / `quantizationHead` (ResidualBlock, AttentionBlock, conv3x3 on the same z; mcquic/modules/compressor.py:148-160)
# and `dequantizationHead` / `sideHead` (AttentionBlock, conv3x3, ResidualBlock; :166-175) apply the same layer shapes to
# different tensors: layer by layer the stacks share ONE launch (ops.conv2d_multi; four problems inside their AttentionBlocks).
# On the 48x32 ... 12x8 latent maps a launch is one round of waves or less -- prologue, LDS reduction and epilogue exposed --
# so two or four problems per launch fill the chip where one does not (the training graph does the same: autograd.LockstepFn).
def _infer_kind(m) -> Optional[str]:
    if isinstance(m, ResidualBlock) and m._skip is None and not m.denseNorm:
        return "rb"
    if isinstance(m, AttentionBlock) and not m.denseNorm:
        return "attn"
    if type(m).__name__ == "Conv2d" and m.kernelSize == 3 and m.stride == 1:
        return "conv"
    return None


def lockstep_ok(stacks, xs) -> bool:
    """Can `lockstep_infer` take these stacks?  Same layer kinds, same input shapes, maps small enough for shared launches."""
    if not (ops._MULTI and len(stacks) >= 2):
        return False
    kinds = [[_infer_kind(m) for m in st] for st in stacks]
    if any(None in k for k in kinds) or any(k != kinds[0] for k in kinds[1:]):
        return False
    if any(x.shape != xs[0].shape for x in xs[1:]):
        return False
    return xs[0].shape[0] * xs[0].shape[2] * xs[0].shape[3] <= _MULTI_MAX_PIXELS


def lockstep_infer(stacks, xs, layers: Optional[int] = None):
    """Run the first `layers` layers (default: all) of k same-structure stacks on k inputs, one multi-problem launch per
    convolution layer.  Returns the k outputs; every output carries its SiLU twin (the next consumer's act1)."""
    k = len(stacks)
    xs = list(xs)
    for li, layer in enumerate(zip(*stacks)):
        if layers is not None and li >= layers:
            break
        kind = _infer_kind(layer[0])
        if kind != "conv":
            # one flag set per launch: either every input brings its SiLU twin (then no problem evaluates SiLU in its k-loop)
            # or none does
            twins = [ops.silu_twin(x) is not None for x in xs]
            if any(twins) and not all(twins):
                for x, has in zip(xs, twins):
                    if not has:
                        ops.set_silu_twin(x, ops.silu(x))
        if kind == "rb":
            ts = ops.conv2d_multi(xs, [m._branch[1].packed() for m in layer], silu_in=True, silu_out=True)
            xs = ops.conv2d_multi(ts, [m._branch[3].packed() for m in layer], per_problem=[dict(res=x) for x in xs], dual_silu=True)
        elif kind == "attn":
            a, b = list(xs), list(xs)
            for i in range(3):
                blocks = [m._mainBranch[i] for m in layer] + [m._sideBranch[i] for m in layer]
                ts = ops.conv2d_multi(a + b, [blk._branch[1].packed() for blk in blocks], silu_in=True, silu_out=True)
                ys = ops.conv2d_multi(ts, [blk._branch[3].packed() for blk in blocks], per_problem=[dict(res=t) for t in a + b], dual_silu=True)
                a, b = ys[:k], ys[k:]
            xs = ops.conv2d_multi(b, [m._sideBranch[3].packed() for m in layer],
                                  per_problem=[dict(gate_mul=ai, gate_id=xi) for ai, xi in zip(a, xs)], dual_silu=True)
        else:
            xs = ops.conv2d_multi(xs, [m.packed() for m in layer], dual_silu=True)
    return xs
