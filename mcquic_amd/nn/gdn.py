"""GDN / inverse GDN on HIP kernels (reference: mcquic/nn/gdn.py:28-91, mcquic/nn/base.py:17-84).

y = x * rsqrt(beta + gamma @ x^2)  (GDN)   /   y = x * sqrt(beta + gamma @ x^2)  (IGDN)

One kernel: a 1x1 MFMA conv that squares its input on load (CONV_SQUARE_IN), adds beta as the bias and
multiplies the input by 1/sqrt(.) or sqrt(.) in the epilogue.  The non-negative re-parametrisation of
beta / gamma (max(p, bound)^2 - eps^2) is folded once per weight version instead of on every call.
State_dict keys match the reference: beta, gamma, {beta,gamma}_reparam.eps,
{beta,gamma}_reparam.lowerBound.bound.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from .. import ops

__all__ = ["GenDivNorm", "InvGenDivNorm", "NonNegativeParametrizer", "LowerBound"]

EPS = 1e-6  # mcquic/consts.py:24


class LowerBound(nn.Module):
    """Holds the `bound` buffer (reference: mcquic/nn/base.py:31-54)."""

    def __init__(self, bound: float):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))


class NonNegativeParametrizer(nn.Module):
    """Buffers of the reference's re-parametrisation (mcquic/nn/base.py:57-84): eps = Eps^2,
    lowerBound.bound = sqrt(minimum + Eps^2); init(x) = sqrt(max(x + eps, eps))."""

    def __init__(self, minimum: float = 0.0, eps: float = EPS):
        super().__init__()
        self.register_buffer("eps", torch.Tensor([float(eps) ** 2]))
        self.lowerBound = LowerBound((float(minimum) + float(eps) ** 2) ** 0.5)

    def init(self, x: torch.Tensor) -> torch.Tensor:
        return torch.sqrt(torch.max(x + self.eps, self.eps))


class GenDivNorm(nn.Module):
    _inverse = False

    def __init__(self, inChannels: int, groups: int = 1, biasBound: float = 1e-4, weightInit: float = 0.1):
        super().__init__()
        if groups != 1:
            raise NotImplementedError("grouped GDN is not on the Compressor path")
        self.beta_reparam = NonNegativeParametrizer(minimum=float(biasBound))
        self.beta = nn.Parameter(self.beta_reparam.init(torch.ones(inChannels)))
        self.gamma_reparam = NonNegativeParametrizer()
        self.gamma = nn.Parameter(self.gamma_reparam.init(float(weightInit) * torch.eye(inChannels)))
        self._packed: Optional[ops.PackedConv] = None
        self._packedKey = None
        self._post: Optional[ops.PackedPost] = None
        self._postKey = None

    def packed_post(self) -> "ops.PackedPost":
        """The folded (gamma, beta) in the operand order of the MCQ_CONV_POST_GDN / _IGDN epilogue: the normalisation then runs inside
        the launch of the convolution in front of it (inference; channel 128 -- ops.post_ok says when)."""
        pk = self.packed()
        if self._post is None or self._postKey is not self._packedKey:
            gb, gp = self.gamma_reparam.lowerBound.bound, self.gamma_reparam.eps
            gamma = ops.nonneg_reparam(self.gamma, float(gb), float(gp))
            self._post = ops.PackedPost(gamma, pk.bias)
            self._postKey = self._packedKey
        return self._post

    def packed(self) -> ops.PackedConv:
        bb, gb = self.beta_reparam.lowerBound.bound, self.gamma_reparam.lowerBound.bound
        # keyed on versions / storages only: reading a buffer's value here would force a device sync per call
        key = (ops.tensor_version(self.beta), self.beta.data_ptr(), ops.tensor_version(self.gamma), self.gamma.data_ptr(),
               ops.tensor_version(bb), bb.data_ptr(), ops.tensor_version(gb), gb.data_ptr())
        if self._packed is None or key != self._packedKey:
            beta = ops.nonneg_reparam(self.beta, float(self.beta_reparam.lowerBound.bound), float(self.beta_reparam.eps))
            gamma = ops.nonneg_reparam(self.gamma, float(self.gamma_reparam.lowerBound.bound), float(self.gamma_reparam.eps))
            self._packed = ops.PackedConv(gamma[..., None, None], beta)
            self._packedKey = key
        return self._packed

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training and torch.is_grad_enabled():
            from .. import autograd as AG
            return AG.gdn(x, self, self._inverse)
        if self._inverse:
            return ops.conv2d(x, self.packed(), square_in=True, igdn_mul=x)
        return ops.conv2d(x, self.packed(), square_in=True, gdn_mul=x)


class InvGenDivNorm(GenDivNorm):
    _inverse = True
