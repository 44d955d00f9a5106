"""Cascaded multi-codebook quantizer on HIP kernels (reference: mcquic/modules/quantizer.py).

`UMGMQuantizer.encode / decode` keep the reference's level cascade (:411-428) and module tree
(`_encoders.{l}.{_latentStageEncoder,_quantizationHead,_latentHead,_quantizer,_dequantizer}`,
`_decoders.{l}.{_dequantizationHead,_sideHead,_restoreHead,_dequantizer}`; the codebook is one
Parameter visible under three names, as in the reference).  The distance + argmin never materialises
the [n, m, h, w, k] tensor (ops.vq_assign), and the residual `z - dequant(code)` of the next level is
the epilogue of latentHead's last conv.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Union

import torch
from torch import nn

from .. import ops
from .entropyCoder import EntropyCoder

EPS = 1e-6


class _CodebookCache:
    """Packed (MFMA operand order) view of a codebook Parameter, rebuilt when the Parameter changes."""

    def __init__(self):
        self._packed: Optional[ops.PackedCodebook] = None
        self._key = None

    def get(self, codebook: torch.Tensor) -> ops.PackedCodebook:
        key = (ops.tensor_version(codebook), codebook.data_ptr())
        if self._packed is None or key != self._key:
            self._packed = ops.PackedCodebook(codebook)
            self._key = key
        return self._packed


class LowerBound(nn.Module):
    def __init__(self, bound: float):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))


class _multiCodebookQuantization(nn.Module):
    """reference: quantizer.py:99-239.  Parameters: `_codebook` [m, k, d] (shared), `_temperature` [m, 1, 1, 1],
    buffer `_bound.bound`."""

    def __init__(self, codebook: nn.Parameter, cache: _CodebookCache):
        super().__init__()
        self._m, self._k, self._d = codebook.shape
        self._codebook = codebook
        self._temperature = nn.Parameter(torch.ones((self._m, 1, 1, 1)))
        self._bound = LowerBound(EPS)
        self._cache = [cache]          # list: keep the cache out of nn.Module's attribute registration

    @torch.no_grad()
    def reAssignCodebook(self, freq: torch.Tensor) -> torch.Tensor:
        """Replace never-assigned codewords by copies of the most used ones (reference: quantizer.py:111-136).
        Parameter-sized host logic (torch ops on [k, d]); returns the per-codeword "changed" mask, flattened."""
        codebook = self._codebook.detach().clone()
        freq = freq.to(self._codebook.device).detach().clone()
        for m, (codebookGroup, freqGroup) in enumerate(zip(self._codebook, freq)):
            neverAssignedLoc = freqGroup < EPS
            totalNeverAssigned = int(neverAssignedLoc.sum())
            if totalNeverAssigned > self._k // 2:          # more than half never assigned: drop a random part of them
                mask = torch.zeros((totalNeverAssigned,), device=self._codebook.device)
                maskIdx = torch.randperm(len(mask), device=mask.device)[self._k // 2:]
                mask[maskIdx] = -1.
                freqGroup[neverAssignedLoc] = mask
                neverAssignedLoc = (freqGroup < EPS) * (freqGroup > (-EPS))
                totalNeverAssigned = int(neverAssignedLoc.sum())
            argIdx = torch.argsort(freqGroup, descending=True)
            mostAssigned = codebookGroup[argIdx]
            codebook.data[m, neverAssignedLoc] = mostAssigned[:totalNeverAssigned]
        diff = ((codebook - self._codebook) ** 2).sum(-1) > 1e-4
        self._codebook.data.copy_(codebook)
        return diff.flatten()

    @torch.no_grad()
    def syncCodebook(self):
        """Broadcast rank 0's codebook (reference: quantizer.py:138-142)."""
        import torch.distributed as dist
        codebook = self._codebook.detach().clone()
        dist.broadcast(codebook, 0)
        self._codebook.data.copy_(codebook)

    def encode(self, x: torch.Tensor) -> torch.Tensor:
        """[n, m*d, h, w] -> int64 [n, m, h, w] = argmin_k ((x2 + c2) - 2 x.c), first index on ties (:144-179)."""
        return ops.vq_assign(x, self._cache[0].get(self._codebook))

    def forward(self, x: torch.Tensor, freqEMA: torch.Tensor, uniforms=None):
        """Training-mode forward (:181-239), forward values only (no autograd graph yet).

        logit = (-dist / sqrt(k)) * max(temperature, eps); random drop against the level's frequency EMA;
        gumbelSoftmax(hard=True); code = argmax(logit).  The straight-through sample y_hard - y_soft + y_soft is
        zero except at its arg-max, so it is carried as (index, hot value) instead of a dense [n, m, h, w, k]
        tensor.  `uniforms` = (u_drop, u_gumbel), the reference's two `torch.rand_like(logit)` draws; drawn here
        with torch.rand when None.  Returns ((index, hot), code, logit)."""
        cb = self._cache[0].get(self._codebook)
        n, _, h, w = x.shape
        shape = (n, self._m, h, w, self._k)
        if uniforms is None:
            uniforms = (torch.rand(shape, device=x.device), torch.rand(shape, device=x.device))
        bits = math.log2(self._k)
        # exponent of _randomDrop (:196-198), kept on the device: no host sync in the step
        with torch.no_grad():
            usage = (freqEMA > EPS).float().mean().clamp(0., 1.)
            exponent = -(bits - 1) * (usage ** 2) + bits
        if torch.is_grad_enabled():
            from ..autograd import SoftQuantizeFn
            deq, code, logit = SoftQuantizeFn.apply(x, self._codebook, self._temperature, freqEMA.detach(), uniforms[0], uniforms[1],
                                                    exponent, cb, float(EPS))
            return deq, code, logit           # the sample is represented by its (differentiable) dequantisation
        logit = ops.vq_logits(x, cb, self._temperature, float(EPS))
        code, index, hot = ops.vq_gumbel_sample(logit, uniforms[0], uniforms[1], freqEMA, exponent)
        return (index, hot), code, logit


class _multiCodebookDeQuantization(nn.Module):
    """reference: quantizer.py:242-274."""

    def __init__(self, codebook: nn.Parameter, cache: _CodebookCache):
        super().__init__()
        self._m, self._k, self._d = codebook.shape
        self._codebook = codebook
        self._cache = [cache]

    def forward(self, sample) -> torch.Tensor:
        """bmm(sample, codebook) (:262-274) for the (index, hot value) form of the straight-through sample."""
        if torch.is_tensor(sample):           # training graph: SoftQuantizeFn already produced sample @ codebook
            return sample
        index, hot = sample
        return ops.vq_dequant_soft(index, hot, self._cache[0].get(self._codebook))

    def decode(self, code: torch.Tensor, dual_silu: bool = False) -> torch.Tensor:
        """int64 [n, m, h, w] -> [n, m*d, h, w] (:249-259)."""
        return ops.vq_gather(code, self._cache[0].get(self._codebook), dual_silu=dual_silu)


class _quantizerEncoder(nn.Module):
    """reference: quantizer.py:277-328."""

    def __init__(self, quantizer, dequantizer, latentStageEncoder, quantizationHead, latentHead):
        super().__init__()
        self._quantizer = quantizer
        self._dequantizer = dequantizer
        self._latentStageEncoder = latentStageEncoder
        self._quantizationHead = quantizationHead
        self._latentHead = latentHead

    @property
    def Codebook(self):
        return self._quantizer._codebook

    def syncCodebook(self):
        self._quantizer.syncCodebook()

    def reAssignCodebook(self, freq: torch.Tensor) -> torch.Tensor:
        return self._quantizer.reAssignCodebook(freq)

    def encode(self, x: torch.Tensor):
        z = self._latentStageEncoder(x)
        code = self._quantizer.encode(self._quantizationHead(z))
        if self._latentHead is None:
            return None, code
        deq = self._dequantizer.decode(code)
        # z' - dequant(code): the subtraction is the epilogue of latentHead's closing conv3x3 (:318)
        head = self._latentHead
        t = z
        for i in range(len(head) - 1):
            t = head[i](t)
        return head[len(head) - 1](t, res=deq, res_scale=-1.0, dual_silu=True), code


    def _forward(self, x: torch.Tensor, freqEMA: torch.Tensor, uniforms=None):
        """Training-mode level (:295-305): returns (sample, residual for the next level, code, logit)."""
        z = self._latentStageEncoder(x)
        q, code, logit = self._quantizer(self._quantizationHead(z), freqEMA, uniforms)
        if self._latentHead is None:
            return q, None, code, logit
        deq = self._dequantizer(q)
        head = self._latentHead
        if torch.is_grad_enabled():
            from .. import autograd as AG
            return q, AG.sub(head(z), deq), code, logit
        t = z
        for i in range(len(head) - 1):
            t = head[i](t)
        return q, head[len(head) - 1](t, res=deq, res_scale=-1.0, dual_silu=True), code, logit

    def forward(self, x: torch.Tensor, freqEMA: torch.Tensor, uniforms=None):
        return self._forward(x, freqEMA, uniforms)


class _quantizerDecoder(nn.Module):
    """reference: quantizer.py:330-365."""

    def __init__(self, dequantizer, dequantizationHead, sideHead, restoreHead):
        super().__init__()
        self._dequantizer = dequantizer
        self._dequantizationHead = dequantizationHead
        self._sideHead = sideHead
        self._restoreHead = restoreHead

    def decode(self, code: torch.Tensor, formerLevel: Optional[torch.Tensor]):
        q = self._dequantizationHead(self._dequantizer.decode(code, dual_silu=True))
        if self._sideHead is not None:
            xHat = ops.add(q, self._sideHead(formerLevel), dual_silu=True)     # q + sideHead(formerLevel) (:354)
        else:
            xHat = q
        return self._restoreHead(xHat)


    def forward(self, q, formerLevel: Optional[torch.Tensor]):
        """Training-mode level (:359-365): like decode, from the straight-through sample."""
        x = self._dequantizationHead(self._dequantizer(q))
        if self._sideHead is not None:
            if torch.is_grad_enabled():
                from .. import autograd as AG
                x = AG.add(x, self._sideHead(formerLevel))
            else:
                x = ops.add(x, self._sideHead(formerLevel), dual_silu=True)
        return self._restoreHead(x)


class BaseQuantizer(nn.Module):
    """reference: quantizer.py:32-78."""

    def __init__(self, m: int, k: List[int]):
        super().__init__()
        self._entropyCoder = EntropyCoder(m, k)
        self._m = m
        self._k = k

    @property
    def CDFs(self):
        return self._entropyCoder.CDFs

    @property
    def NormalizedFreq(self):
        return self._entropyCoder.NormalizedFreq

    def compress(self, x: torch.Tensor):
        codes = self.encode(x)
        binaries, codeSize = self._entropyCoder.compress(codes)
        return codes, binaries, codeSize

    def decompress(self, binaries, codeSize) -> torch.Tensor:
        return self.decode(self._entropyCoder.decompress(binaries, codeSize))


class UMGMQuantizer(BaseQuantizer):
    """reference: quantizer.py:368-467."""
    _components = ["latentStageEncoder", "quantizationHead", "latentHead", "dequantizationHead", "sideHead", "restoreHead"]

    def __init__(self, channel: int, m: int, k: Union[int, List[int]], permutationRate: float,
                 components: Dict[str, Callable[[], nn.Module]]):
        if isinstance(k, int):
            k = [k]
        super().__init__(m, k)
        lse, qh, lh, dqh, sh, rh = [components[key] for key in self._components]
        encoders, decoders = [], []
        for i, ki in enumerate(k):
            last = i == len(k) - 1
            # SmallInit, N(0, 2 / (5 d)) (quantizer.py:394-398)
            codebook = nn.Parameter(nn.init.normal_(torch.empty(m, ki, channel // m), std=math.sqrt(2 / (5 * channel / m))))
            cache = _CodebookCache()
            quantizer = _multiCodebookQuantization(codebook, cache)
            dequantizer = _multiCodebookDeQuantization(codebook, cache)
            encoders.append(_quantizerEncoder(quantizer, dequantizer, lse(), qh(), None if last else lh()))
            decoders.append(_quantizerDecoder(dequantizer, dqh(), None if last else sh(), rh()))
        self._encoders = nn.ModuleList(encoders)
        self._decoders = nn.ModuleList(decoders)

    @property
    def Codebooks(self):
        return list(encoder.Codebook for encoder in self._encoders)

    def encode(self, x: torch.Tensor) -> List[torch.Tensor]:
        codes = []
        for encoder in self._encoders:
            x, code = encoder.encode(x)
            codes.append(code)
        return codes

    def decode(self, codes: List[torch.Tensor]) -> Optional[torch.Tensor]:
        if len(codes) != len(self._decoders):
            raise RuntimeError(f"expected {len(self._decoders)} code levels, got {len(codes)}")
        self._entropyCoder._checkShape(codes)
        formerLevel = None
        for decoder, code in zip(self._decoders[::-1], codes[::-1]):
            formerLevel = decoder.decode(code, formerLevel)
        return formerLevel

    def forward(self, x: torch.Tensor, uniforms=None):
        """Training-mode forward (:443-467), forward values only: per level soft-assign with the level's frequency EMA
        (the reference snapshot passes `permutationRate` where this tensor belongs, :399 -- repaired here), decoders in
        reverse, then the code-frequency EMA update with its all-reduce (entropyCoder.py:28-44).
        `uniforms`: optional list of (u_drop, u_gumbel) per level.  Returns (yHat, codes, logits)."""
        quantizeds, codes, logits = [], [], []
        for lv, encoder in enumerate(self._encoders):
            quantized, x, code, logit = encoder(x, self._entropyCoder._freqEMA[lv], None if uniforms is None else uniforms[lv])
            quantizeds.append(quantized)
            codes.append(code)
            logits.append(logit)
        formerLevel = None
        for decoder, quantized in zip(self._decoders[::-1], quantizeds[::-1]):
            formerLevel = decoder(quantized, formerLevel)
        self._entropyCoder(codes)
        return formerLevel, codes, logits

    def reAssignCodebook(self) -> torch.Tensor:
        """reference: quantizer.py:430-436."""
        freqs = self.NormalizedFreq
        reassigned = [encoder.reAssignCodebook(freq) for encoder, freq in zip(self._encoders, freqs)]
        return torch.cat(reassigned).float().mean()

    def syncCodebook(self):
        """reference: quantizer.py:438-441."""
        import torch.distributed as dist
        dist.barrier()
        for encoder in self._encoders:
            encoder.syncCodebook()
