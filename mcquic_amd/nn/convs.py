"""Convolution layers of the Compressor path on HIP kernels.

Mirrors the factories of the reference (mcquic/nn/convs.py): `conv3x3` (:77-100), `conv1x1` (:257-276)
and `pixelShuffle3x3` (:221-255), with the same parameter names / state_dict keys (`weight`, `bias`;
`0.weight`, `0.bias` for the pixel-shuffle pair).  Only the configurations `Compressor` uses exist
(groups=1, zeros padding, r=2 up-sampling).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn

from .. import ops

__all__ = ["Conv2d", "conv3x3", "conv1x1", "pixelShuffle3x3", "PixelShuffle3x3"]


class Conv2d(nn.Module):
    """Parameter holder + launcher for one dense conv (kernel 1 or 3, stride 1 or 2, padding k//2).

    Initialisation follows nn.Conv2d's default law (kaiming_uniform(a=sqrt(5)) => U(+-1/sqrt(fan_in))),
    which is what the reference gets from `nn.Conv2d(...)`.
    """

    def __init__(self, inChannels: int, outChannels: int, kernelSize: int, stride: int = 1, bias: bool = True):
        super().__init__()
        self.inChannels, self.outChannels, self.kernelSize, self.stride = inChannels, outChannels, kernelSize, stride
        bound = 1.0 / math.sqrt(inChannels * kernelSize * kernelSize)
        self.weight = nn.Parameter(torch.empty(outChannels, inChannels, kernelSize, kernelSize).uniform_(-bound, bound))
        self.bias = nn.Parameter(torch.empty(outChannels).uniform_(-bound, bound)) if bias else None
        self._packed: Optional[ops.PackedConv] = None
        self._packedKey = None

    def _key(self):
        return (ops.tensor_version(self.weight), self.weight.data_ptr(), None if self.bias is None else (ops.tensor_version(self.bias), self.bias.data_ptr()),
                ops.winograd_enabled())

    def packed(self) -> ops.PackedConv:
        """Weights in MFMA operand order; re-packed whenever the parameters change (version / storage / device)."""
        key = self._key()
        if self._packed is None or key != self._packedKey:
            # in place where the shapes allow it: the streams keep their addresses (a captured training step may hold them)
            with torch.no_grad():
                if self._packed is None or not self._packed.repack_(self.weight, self.bias):
                    self._packed = ops.PackedConv(self.weight, self.bias)
            self._packedKey = key
        return self._packed

    def packed_subpixel(self) -> ops.PackedConv:
        """(inference) The operand streams of a pixelShuffle3x3's convolution with its rows in SUB-PIXEL-MAJOR order -- row 128 T + c =
        channel 4 c + T, so that the 128 rows of tile T are the 128 shuffled channels of sub-pixel T = 2 dy + dx -- for the launch that
        runs the InvGenDivNorm behind it in its own epilogue (MCQ_CONV_POST_IGDN | MCQ_CONV_SHUFFLE2; Cout = 4 x 128)."""
        key = self._key()
        pk = self.__dict__.get("_packedSub")
        if pk is None or self.__dict__.get("_packedSubKey") != key:
            with torch.no_grad():
                c = self.outChannels // 4
                w = self.weight.detach().reshape(c, 4, self.inChannels, self.kernelSize, self.kernelSize).transpose(0, 1).reshape(self.weight.shape).contiguous()
                b = None if self.bias is None else self.bias.detach().reshape(c, 4).t().reshape(-1).contiguous()
                pk = ops.PackedConv(w, b, copy_bias=False, winograd=False)
            self.__dict__["_packedSub"], self.__dict__["_packedSubKey"] = pk, key
        return pk

    def packed_post(self) -> "ops.PackedPost":
        """(inference) A [128, 128] 1x1 layer in the operand order of the MCQ_CONV_POST_GATE epilogue (the AttentionBlock's conv1x1)."""
        key = self._key()
        pk = self.__dict__.get("_packedPost")
        if pk is None or self.__dict__.get("_packedPostKey") != key:
            with torch.no_grad():
                pk = ops.PackedPost(self.weight, self.bias)
            self.__dict__["_packedPost"], self.__dict__["_packedPostKey"] = pk, key
        return pk

    @staticmethod
    def repack_stale(convs, masks=None) -> int:
        """Bring the operand streams of all `convs` up to date in grouped launches (ops.pack_convs: up to 16 weights of one
        shape per launch) -- what a training loop does once per step after its optimizer update instead of 2 launches + a bias
        copy per convolution.  Forward streams for every stale conv; input-gradient streams for the stale ones that have
        been asked for one before.  Returns the number of streams re-packed.  `masks` ({id(PackedConv): section mask}, from
        ops.sections_used): copies to refresh when a stream is re-packed in place -- only for a re-pack recorded inside a captured
        step whose launches are known (parallel.GraphedTrainStep); everything else packs whole streams."""
        if ops.winograd_enabled():
            return 0            # (the opt-in Winograd streams are packed per convolution: grouped packs carry none)
        fwd, bwd = {}, {}
        for c in convs:
            params = c._parameters                   # (nn.Module.__getattr__ costs more than the whole check)
            w, b = params["weight"], params.get("bias")
            wv, wptr = ops.tensor_version(w), w.data_ptr()
            key = (wv, wptr, None if b is None else (ops.tensor_version(b), b.data_ptr()), False)
            if c._packed is None or key != c._packedKey:
                fwd.setdefault((w.shape, w.device), []).append((c, key, w, b))
            cache = c.__dict__.get("_dgradCache")
            if cache is not None and cache.key is not None and cache.key != (wv, wptr, c.stride):
                bwd.setdefault((w.shape, w.device, c.stride), []).append((c, cache, w, (wv, wptr, c.stride)))
        done = 0
        for group in fwd.values():
            packs = ops.pack_convs([w for _, _, w, _ in group], [b for _, _, _, b in group], into=[c._packed for c, _, _, _ in group],
                                   masks=None if masks is None else [masks.get(id(c._packed), 0) for c, _, _, _ in group])
            for (c, key, _, _), pk in zip(group, packs):
                c._packed, c._packedKey = pk, key
            done += len(group)
        for (_, _, stride), group in bwd.items():
            packs = ops.pack_convs([w for _, _, w, _ in group], dgrad=True, stride=stride, into=[cache.packed for _, cache, _, _ in group],
                                   masks=None if masks is None else [masks.get(id(cache.packed), 0) for _, cache, _, _ in group])
            for (c, cache, _, key), pk in zip(group, packs):
                cache.packed, cache.key = pk, key
                cache.winograd = False                    # (grouped packs carry no Winograd stream: those launches stay direct)
            done += len(group)
        return done

    def forward(self, x: torch.Tensor, **fused) -> torch.Tensor:
        if self.training and torch.is_grad_enabled():
            # training graph: plain conv (+ residual / pixel-shuffle store) with HIP backward kernels
            from .. import autograd as AG
            extra = set(fused) - {"res", "shuffle2", "dual_silu"}
            if extra:
                raise NotImplementedError(f"fused conv options {sorted(extra)} are inference-only")
            return AG.conv(x, self, res=fused.get("res"), shuffle2=bool(fused.get("shuffle2", False)), dual_silu=bool(fused.get("dual_silu", False)))
        return ops.conv2d(x, self.packed(), self.stride, **fused)

    def extra_repr(self) -> str:
        return f"{self.inChannels}, {self.outChannels}, kernel_size={self.kernelSize}, stride={self.stride}"


def conv3x3(inChannels: int, outChannels: int, stride: int = 1, bias: bool = True, groups: int = 1) -> Conv2d:
    """3x3 conv with padding 1 (reference: mcquic/nn/convs.py:77-100)."""
    if groups != 1:
        raise NotImplementedError("grouped convolutions are not on the Compressor path")
    return Conv2d(inChannels, outChannels, 3, stride, bias)


def conv1x1(inChannels: int, outChannels: int, stride: int = 1, bias: bool = True, groups: int = 1) -> Conv2d:
    """1x1 conv (reference: mcquic/nn/convs.py:257-276)."""
    if groups != 1 or stride != 1:
        raise NotImplementedError("only dense stride-1 1x1 convolutions are on the Compressor path")
    return Conv2d(inChannels, outChannels, 1, 1, bias)


class PixelShuffle3x3(nn.Sequential):
    """Sequential(Conv2d(C, C_out * r^2, 3, padding=1), PixelShuffle(r)) as ONE kernel: the shuffle is the
    conv's store pattern (reference: mcquic/nn/convs.py:250-255).  Keys: `0.weight`, `0.bias`."""

    def __init__(self, inChannels: int, outChannels: int, r: int = 2):
        if r != 2:
            raise NotImplementedError("only 2x up-sampling is on the Compressor path")
        super().__init__(Conv2d(inChannels, outChannels * r * r, 3), nn.PixelShuffle(r))

    def forward(self, x: torch.Tensor, **fused) -> torch.Tensor:
        return self[0](x, shuffle2=True, **fused)


def pixelShuffle3x3(inChannels: int, outChannels: int, r: float = 1, groups: int = 1) -> PixelShuffle3x3:
    if groups != 1:
        raise NotImplementedError("grouped convolutions are not on the Compressor path")
    return PixelShuffle3x3(inChannels, outChannels, int(r))
