"""Entropy-coder state that sits beside the tensor path (reference: mcquic/modules/entropyCoder.py:15-154).

Only what the Compressor API needs at this stage is present: the per-level code-frequency EMA
(`_freqEMA.{l}` in the state_dict) and the `NormalizedFreq` view.  `compress` / `decompress` (rANS byte
streams) raise NotImplementedError exactly as the reference snapshot does (entropyCoder.py:107,140:
both bodies start with `raise NotImplementedError`); the rANS coder is the first "next" row of
SURVEY.md §8(f).
"""
from __future__ import annotations

from typing import List

import torch
from torch import nn


class EntropyCoder(nn.Module):
    def __init__(self, m: int, k: List[int], ema: float = 0.9):
        super().__init__()
        # initial value is uniform (entropyCoder.py:22)
        self._freqEMA = nn.ParameterList(nn.Parameter(torch.ones(m, ki) / ki, requires_grad=False) for ki in k)
        self._m, self._k, self._ema = m, k, ema

    @property
    def NormalizedFreq(self) -> List[torch.Tensor]:
        """List of [m, k_l] probabilities (entropyCoder.py:50-55,73-79)."""
        return [(f / f.sum(-1, keepdim=True)).detach().clone() for f in self._freqEMA]

    @property
    def CDFs(self):
        raise NotImplementedError("rANS CDF tables: next row (SURVEY.md §8(f) #1); dead in the reference snapshot too")

    def _checkShape(self, codes: List[torch.Tensor]):
        """Same checks and RuntimeErrors as entropyCoder.py:79-93."""
        info = ("Please give codes with correct shape, for example, [[1, 2, 24, 24], [1, 2, 12, 12], ...], which is a "
                "`level` length list. each code has shape [n, m, h, w]. ")
        if len(codes) < 1:
            raise RuntimeError("Length of codes is 0.")
        n, m = codes[0].shape[0], codes[0].shape[1]
        for code in codes:
            newN, newM = code.shape[0], code.shape[1]
            if n < 1:
                raise RuntimeError(info + "Now `n` = 0.")
            if m != newM:
                raise RuntimeError(info + "Now `m` is inconsisitent.")
            if n != newN:
                raise RuntimeError(info + "Now `n` is inconsisitent.")
        return n, m

    def compress(self, codes):
        raise NotImplementedError("EntropyCoder.compress: rANS byte streams are a 'next' row (SURVEY.md §8(f) #1); "
                                  "the reference snapshot raises here as well (entropyCoder.py:107)")

    def decompress(self, binaries, codeSizes):
        raise NotImplementedError("EntropyCoder.decompress: rANS byte streams are a 'next' row (SURVEY.md §8(f) #1); "
                                  "the reference snapshot raises here as well (entropyCoder.py:140)")
