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
from typing import Callable, Dict, List, Optional, Tuple, Union

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
        key = (codebook._version, codebook.data_ptr())
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

    def encode(self, x: torch.Tensor) -> torch.Tensor:
        """[n, m*d, h, w] -> int64 [n, m, h, w] = argmin_k ((x2 + c2) - 2 x.c), first index on ties (:144-179)."""
        return ops.vq_assign(x, self._cache[0].get(self._codebook))


class _multiCodebookDeQuantization(nn.Module):
    """reference: quantizer.py:242-274."""

    def __init__(self, codebook: nn.Parameter, cache: _CodebookCache):
        super().__init__()
        self._m, self._k, self._d = codebook.shape
        self._codebook = codebook
        self._cache = [cache]

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

    def reAssignCodebook(self):
        raise NotImplementedError("training-side codebook maintenance is a 'next' row (SURVEY.md §8(f) #4)")

    def syncCodebook(self):
        raise NotImplementedError("training-side codebook maintenance is a 'next' row (SURVEY.md §8(f) #4)")
