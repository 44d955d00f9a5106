"""`Compressor` with the reference's API surface on HIP kernels (reference: mcquic/modules/compressor.py).

    Compressor(channel, m, k, permutationRate=0.0)         # compressor.py:121
    .encode(x) -> List[LongTensor [n, m, h_l, w_l]]        # :79-88
    .decode(codes) -> Tensor [n, 3, H, W]                  # :114-117
    .compress(x) / .decompress(binaries, headers)          # :67-77 / :90-112 (rANS streams, csrc/rans.cpp)
    .Codebooks / .NormalizedFreq / .CDFs / .CodeUsage / .QuantizationParameter

The module tree and every state_dict key are the reference's (718 entries for the qp=2 shape), so a
reference checkpoint's `model` dict loads with `load_state_dict(strict=True)`.
"""
from __future__ import annotations

import operator
from typing import List, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from ..nn import AttentionBlock, ResidualBlock, ResidualBlockShuffle, ResidualBlockWithStride, conv3x3, pixelShuffle3x3
from ..utils.specification import FileHeader, ImageSize
from .quantizer import BaseQuantizer, ResidualBackwardQuantizer, UMGMQuantizer

from ..utils.specification import VERSION as __version__   # the reference snapshot's mcquic.__version__ (FileHeader)


_VERSION_OF = operator.attrgetter("_version")
_DATA_PTR_OF = operator.methodcaller("data_ptr")


def _forget_captures_hook(module, incompatible_keys):
    """load_state_dict post-hook of BaseCompressor: new weights, so captured hipGraphs (which replay packed copies) are stale."""
    module._forgetCaptures()


class _GraphedCall:
    """One captured hipGraph of `fn` for fixed input shapes: replay costs one launch instead of ~165 Python-side
    kernel launches (at batch 1 the eager path is host-bound: ~25 us of Python per launch against ~10 us kernels).
    Inputs are copied into the graph's static buffers, outputs are cloned out of them."""

    def __init__(self, fn, inputs):
        self.static_in = [t.clone() for t in inputs]
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):                      # warm-up off the capture: weight packing, allocator pools
            for _ in range(2):
                fn(*self.static_in)
        cur.wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = fn(*self.static_in)

    def __call__(self, inputs):
        for dst, src in zip(self.static_in, inputs):
            dst.copy_(src)
        self.graph.replay()
        out = self.static_out
        return [t.clone() for t in out] if isinstance(out, (list, tuple)) else out.clone()


class AlignedPadding(nn.Module):
    """Reflect-pad H, W up to multiples of `base` (reference: mcquic/data/transforms.py:81-99).
    A no-op for 768x512.  Data movement only (torch's reflect pad on the device)."""

    def __init__(self, base: int = 128):
        super().__init__()
        self._base = base

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        h, w = x.shape[-2], x.shape[-1]
        wPadding = ((w // self._base + 1) * self._base - w) % self._base
        hPadding = ((h // self._base + 1) * self._base - h) % self._base
        if wPadding == 0 and hPadding == 0:
            return x
        padLeft = wPadding // 2
        padTop = hPadding // 2
        return F.pad(x, (padLeft, wPadding - padLeft, padTop, hPadding - padTop), "reflect")


class BaseCompressor(nn.Module):
    def __init__(self, encoder: nn.Module, quantizer: BaseQuantizer, decoder: nn.Module):
        super().__init__()
        self._encoder = encoder
        self._decoder = decoder
        self._quantizer = quantizer
        self._qp = "-1"
        self._padding = AlignedPadding()
        self._graphs = None          # {(kind, shapes, device): _GraphedCall} once enableGraphs(True)
        self._graphStamp = None      # fingerprint of the weights the captures were taken under
        self.register_load_state_dict_post_hook(_forget_captures_hook)      # (module-level function: a lambda here made the model unpicklable)

    def enableGraphs(self, enabled: bool = True):
        """Replay `encode` / `decode` as captured hipGraphs (one per input shape).  For latency-bound small batches.
        A capture bakes in the addresses of the packed weights / codebooks it ran on, so every captured call first
        compares a fingerprint of all parameters and buffers (version counters + storage addresses, ~0.15 ms) with the one
        the captures were taken under: an optimizer step, `load_state_dict`, `reAssignCodebook`, `.to(...)`, `p.data = ...`
        or any other in-place update drops all captures before anything is replayed.  Two things the fingerprint cannot
        see: `p.data.copy_(...)` (no version bump, same address) and a Parameter OBJECT swapped for another one
        (`module.weight = nn.Parameter(...)`): call `enableGraphs()` again after either."""
        self._graphs = {} if enabled else None
        self._graphStamp = None
        return self

    def _weightStamp(self):
        from .. import ops
        tensors = self.__dict__.get("_stampTensors")
        if tensors is None:                      # (walking the module tree costs ~3 ms: done once, redone after _apply)
            tensors = self.__dict__["_stampTensors"] = list(self.parameters()) + list(self.buffers())
        ptrs = tuple(map(_DATA_PTR_OF, tensors))   # `p.data = other` / set_() move the storage without touching the counter
        try:
            return tuple(map(_VERSION_OF, tensors)), ptrs
        except RuntimeError:                     # tensors created under torch.inference_mode() carry no counter
            return tuple(ops.tensor_version(t) for t in tensors), ptrs

    def _forgetCaptures(self):
        self.__dict__.pop("_stampTensors", None)
        if self._graphs is not None:
            self._graphs.clear()

    def _apply(self, fn, *args, **kwargs):
        """`.to()` / `.cuda()` / `.float()` replace storages (and buffer objects): forget the captures and the tensor list."""
        self._forgetCaptures()
        return super()._apply(fn, *args, **kwargs)

    def _graphed(self, kind: str, fn, inputs):
        from .. import ops
        stamp = (self._weightStamp(), ops.winograd_enabled())     # (the opt-in switch re-packs every convolution: as good as new weights)
        if stamp != self._graphStamp:            # weights changed since the captures: their packed operands are stale
            self._graphs.clear()
            self._graphStamp = stamp
        key = (kind, tuple(tuple(t.shape) for t in inputs), inputs[0].device)
        g = self._graphs.get(key)
        if g is None:
            g = self._graphs[key] = _GraphedCall(fn, inputs)
        return g(inputs)

    @property
    def QuantizationParameter(self) -> str:
        return self._qp

    @QuantizationParameter.setter
    def QuantizationParameter(self, qp: str):
        self._qp = qp

    def forward(self, x: torch.Tensor, uniforms=None):
        """Training-mode forward (compressor.py:35-43): (xHat, yHat, codes, logits); None in eval mode like the
        reference.  With grad enabled the step runs through mcquic_amd.autograd (HIP kernels in both directions:
        xHat.backward(...) fills every parameter's .grad); under torch.no_grad() the fused inference kernels are used.
        `uniforms`: optional per-level (u_drop, u_gumbel) draws replacing the two `torch.rand_like(logit)` calls.
        The returned `logits` carry their graph like the reference's (autograd.SoftQuantizeFn: a gradient on them reaches
        latents, codebooks and temperatures; the shipped losses, mcquic/loss/__init__.py:47-62, never produce one)."""
        if not self.training:
            return None
        self._check(x)
        if torch.is_grad_enabled():
            self._repackStale()
            y = self._trainEncode(x)
            yHat, codes, logits = self._quantizer(y, uniforms)
            return self._decoder(yHat), yHat, codes, logits
        y = self._encode_latent(x, pad=False)
        yHat, codes, logits = self._quantizer(y, uniforms)
        xHat = self._decoder(yHat)
        return xHat, yHat, codes, logits

    def _trainEncode(self, x: torch.Tensor) -> torch.Tensor:
        """`self._encoder(x)` in the training graph -- no padding there (compressor.py:39); the stem conv also stores silu(.)
        for the block that follows (every encoder here continues with an activation)."""
        y = self._encoder[0](x, dual_silu=True)
        for i in range(1, len(self._encoder)):
            y = self._encoder[i](y)
        return y

    def _repackStale(self):
        """After an optimizer step every convolution's operand streams are stale: refresh them in grouped launches
        (nn.convs.Conv2d.repack_stale) before the step instead of one by one inside it."""
        convs = self.__dict__.get("_convList")
        if convs is None:
            from ..nn.convs import Conv2d
            convs = self.__dict__["_convList"] = [m for m in self.modules() if isinstance(m, Conv2d)]
        from ..nn.convs import Conv2d
        Conv2d.repack_stale(convs, self.__dict__.get("_packMasks"))     # (_packMasks: only while parallel.GraphedTrainStep captures)
        gdns = self.__dict__.get("_gdnList")
        if gdns is None:
            from ..nn.gdn import GenDivNorm
            gdns = self.__dict__["_gdnList"] = [m for m in self.modules() if isinstance(m, GenDivNorm)]
        if gdns:
            from .. import autograd as AG
            AG.refresh_gdn_operands(gdns)                               # the GDN layers' folded parameters and operand streams, grouped too

    def reAssignCodebook(self) -> torch.Tensor:
        return self._quantizer.reAssignCodebook()

    def syncCodebook(self):
        return self._quantizer.syncCodebook()

    @property
    def Codebooks(self):
        return self._quantizer.Codebooks

    @property
    def CDFs(self):
        return self._quantizer.CDFs

    @property
    def NormalizedFreq(self):
        return self._quantizer.NormalizedFreq

    @property
    def CodeUsage(self):
        return torch.cat(list((freq > 1e-6).flatten() for freq in self._quantizer.NormalizedFreq)).float().mean()

    def _check(self, x: torch.Tensor):
        if x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError(f"expected an image batch [n, 3, h, w], got {tuple(x.shape)}")

    def _encode_latent(self, x: torch.Tensor, pad: bool = True) -> torch.Tensor:
        """`self._encoder(self._padding(x))`; the stem conv also emits silu(.) for the first ResidualBlock."""
        y = self._encoder[0](self._padding(x) if pad else x, dual_silu=True)
        for i in range(1, len(self._encoder)):
            y = self._encoder[i](y)
        return y

    def encode(self, x: torch.Tensor) -> List[torch.Tensor]:
        self._check(x)
        if x.shape[0] == 0:
            # an empty shard (fewer images than ranks, parallel.shard_range): PyTorch's layers hand back empty tensors of the
            # right geometry (the reference's behaviour); the kernels refuse empty launches, so the geometry comes from one blank image
            return [c[:0] for c in self.encode(x.new_zeros((1,) + tuple(x.shape[1:])))]
        with torch.no_grad():
            if self._graphs is not None and x.is_cuda and not torch.cuda.is_current_stream_capturing():
                return self._graphed("encode", lambda t: self._quantizer.encode(self._encode_latent(t)), [x.contiguous()])
            return self._quantizer.encode(self._encode_latent(x))

    def decode(self, codes: List[torch.Tensor]) -> torch.Tensor:
        if codes[0].shape[0] == 0:                            # (empty shard, see encode)
            return self.decode([c.new_zeros((1,) + tuple(c.shape[1:])) for c in codes])[:0]
        with torch.no_grad():
            if self._graphs is not None and codes[0].is_cuda and not torch.cuda.is_current_stream_capturing():
                return self._graphed("decode", lambda *c: self._decoder(self._quantizer.decode(list(c))), [c.contiguous() for c in codes])
            return self._decoder(self._quantizer.decode(codes))

    def compress(self, x: torch.Tensor) -> Tuple[List[torch.Tensor], List[List[bytes]], List[FileHeader]]:
        self._check(x)
        n, c, h, w = x.shape
        if n == 0:                                            # (empty shard: no streams, no headers)
            return self.encode(x), [], []
        with torch.no_grad():
            codes, binaries, codeSizes = self._quantizer.compress(self._encode_latent(x))
        header = [FileHeader(__version__, self._qp, codeSize, ImageSize(height=h, width=w, channel=c)) for codeSize in codeSizes]
        return codes, binaries, header

    def _checkHeaderGeometry(self, headers: List[FileHeader]):
        """Untrusted `.mcq` headers: the code sizes must be the ones THIS image size produces (level 0 = the 128-aligned
        image at 1/16 resolution), before the coder allocates anything from them."""
        for header in headers:
            size, code = header.ImageSize, header.CodeSize
            if not (0 < int(size.height) <= (1 << 20) and 0 < int(size.width) <= (1 << 20)):
                raise RuntimeError("The header's image size is out of range.")
            base = self._padding._base
            ph, pw = -(-int(size.height) // base) * base, -(-int(size.width) // base) * base
            if len(code.heights) < 1 or int(code.heights[0]) * 16 != ph or int(code.widths[0]) * 16 != pw:
                raise RuntimeError(f"The header's code size {list(code.heights)} x {list(code.widths)} does not belong to a "
                                   f"{size.height} x {size.width} image.")

    def decompress(self, binaries: List[List[bytes]], headers: List[FileHeader]) -> torch.Tensor:
        if type(self)._encode_latent is BaseCompressor._encode_latent:       # (Compressor's 16x stem; Neon's geometry differs)
            self._checkHeaderGeometry(headers)
        with torch.no_grad():
            restored = self._decoder(self._quantizer.decompress(binaries, [header.CodeSize for header in headers]))
        imageSize = headers[0].ImageSize
        H, W = restored.shape[-2], restored.shape[-1]
        h, w = imageSize.height, imageSize.width
        cropTop, cropLeft = (H - h) // 2, (W - w) // 2
        return restored[..., cropTop:cropTop + h, cropLeft:cropLeft + w]


class Compressor(BaseCompressor):
    def __init__(self, channel: int, m: int, k: List[int], permutationRate: float = 0.0):
        encoder = nn.Sequential(
            conv3x3(3, channel, 2),
            ResidualBlock(channel, channel),
            ResidualBlockWithStride(channel, channel),
            AttentionBlock(channel),
            ResidualBlock(channel, channel),
            ResidualBlockWithStride(channel, channel),
            ResidualBlock(channel, channel))
        decoder = nn.Sequential(
            ResidualBlock(channel, channel),
            ResidualBlockShuffle(channel, channel),
            AttentionBlock(channel),
            ResidualBlock(channel, channel),
            ResidualBlockShuffle(channel, channel),
            ResidualBlock(channel, channel),
            pixelShuffle3x3(channel, 3, 2))
        quantizer = UMGMQuantizer(channel, m, k, permutationRate, {
            "latentStageEncoder": lambda: nn.Sequential(
                ResidualBlockWithStride(channel, channel), ResidualBlock(channel, channel), AttentionBlock(channel)),
            "quantizationHead": lambda: nn.Sequential(
                ResidualBlock(channel, channel), AttentionBlock(channel), conv3x3(channel, channel)),
            "latentHead": lambda: nn.Sequential(
                ResidualBlock(channel, channel), AttentionBlock(channel), conv3x3(channel, channel)),
            "restoreHead": lambda: nn.Sequential(
                AttentionBlock(channel), ResidualBlock(channel, channel), ResidualBlockShuffle(channel, channel)),
            "dequantizationHead": lambda: nn.Sequential(
                AttentionBlock(channel), conv3x3(channel, channel), ResidualBlock(channel, channel)),
            "sideHead": lambda: nn.Sequential(
                AttentionBlock(channel), conv3x3(channel, channel), ResidualBlock(channel, channel)),
        })
        super().__init__(encoder, quantizer, decoder)



class Neon(BaseCompressor):
    """The stage-1 model of the reference's generative branch (mcquic/modules/compressor.py:181-241; what the snapshot's
    trainer builds, mcquic/train/ddp.py:79-83): a 3x-strided encoder / decoder around a `ResidualBackwardQuantizer`.
    Same constructor, module tree and state_dict keys; `checkpoint_wrapper` (fairscale activation checkpointing, a
    training-memory measure) is not applied.  `groups` / `denseNorm`: see nn/blocks.py (GroupNorm on csrc/norm.hip)."""

    def __init__(self, channel: int, k: int, size: List[int], denseNorm: bool = False, *_, **__):
        quantizer = ResidualBackwardQuantizer(k, size, denseNorm)
        q = quantizer.channel
        encoder = nn.Sequential(
            conv3x3(3, channel),
            AttentionBlock(channel, 32, denseNorm),
            ResidualBlock(channel, channel, 32, denseNorm),
            ResidualBlock(channel, channel, 32, denseNorm),
            ResidualBlockWithStride(channel, channel, 2, 32, denseNorm),
            ResidualBlock(channel, channel, 32, denseNorm),
            ResidualBlockWithStride(channel, channel, 2, 32, denseNorm),
            ResidualBlock(channel, channel, 32, denseNorm),
            ResidualBlockWithStride(channel, channel, 2, 32, denseNorm),
            AttentionBlock(channel, 32, denseNorm),
            ResidualBlock(channel, 2 * channel, 32, denseNorm),
            ResidualBlock(2 * channel, 2 * channel, 32, denseNorm),
            ResidualBlock(2 * channel, 2 * channel, 32, denseNorm),
            ResidualBlock(2 * channel, 2 * channel, 32, denseNorm),
            ResidualBlock(2 * channel, q, 1, denseNorm),
            AttentionBlock(q, 1, denseNorm))
        decoder = nn.Sequential(
            AttentionBlock(q, 1, denseNorm),
            ResidualBlock(q, 2 * channel, 1, denseNorm),
            ResidualBlock(2 * channel, 2 * channel, 32, denseNorm),
            ResidualBlock(2 * channel, 2 * channel, 32, denseNorm),
            ResidualBlock(2 * channel, 2 * channel, 32, denseNorm),
            ResidualBlock(2 * channel, channel, 32, denseNorm),
            AttentionBlock(channel, 32, denseNorm),
            ResidualBlock(channel, channel, 32, denseNorm),
            ResidualBlockShuffle(channel, channel, 2, 32, denseNorm),
            ResidualBlock(channel, channel, 32, denseNorm),
            ResidualBlockShuffle(channel, channel, 2, 32, denseNorm),
            ResidualBlock(channel, channel, 32, denseNorm),
            ResidualBlockShuffle(channel, channel, 2, 32, denseNorm),
            ResidualBlock(channel, channel, 32, denseNorm),
            ResidualBlock(channel, channel, 32, denseNorm),
            AttentionBlock(channel, 32, denseNorm),
            conv3x3(channel, 3))
        super().__init__(encoder, quantizer, decoder)

    def _encode_latent(self, x: torch.Tensor, pad: bool = True) -> torch.Tensor:
        return self._encoder(self._padding(x) if pad else x)

    def residual_backward(self, code: torch.Tensor, level: int) -> torch.Tensor:
        """[n, c, 2h, 2w] <- ([n, m, h, w], level)  (compressor.py:233-235)."""
        with torch.no_grad():
            return self._quantizer.residual_backward(code, level)

    def residual_forward(self, code: torch.Tensor, formerLevel, level: int) -> torch.Tensor:
        """compressor.py:237-239."""
        with torch.no_grad():
            return self._quantizer.residual_forward(code, formerLevel, level)
