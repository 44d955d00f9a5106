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

from . import ops


# ---- weight transforms for input gradients (parameter-sized tensors; cached per weight version) -----------------
def _dgrad_weight(weight: torch.Tensor, stride: int) -> torch.Tensor:
    cout, cin, k, _ = weight.shape
    if stride == 1:
        return weight.flip(2, 3).transpose(0, 1).contiguous()                 # [cin, cout, k, k]
    if k != 3 or stride != 2:
        raise NotImplementedError("dgrad: only stride-2 3x3 convolutions are on the path")
    # dX[2a + i][2b + j] = sum_{ty,tx} dY[a + ty][b + tx] * W[ky(i, ty)][kx(j, tx)]:  i = 0 -> (ty 0, ky 1);
    # i = 1 -> (ty 0, ky 2), (ty +1, ky 0).  Laid out as a 3x3 conv with 4 * cin outputs followed by PixelShuffle(2).
    w2 = torch.zeros((cin, 2, 2, cout, 3, 3), dtype=weight.dtype, device=weight.device)
    kmap = {(0, 0): 1, (1, 0): 2, (1, 1): 0}                                  # (phase, tap offset) -> kernel index
    wt = weight.permute(1, 0, 2, 3)                                           # [cin, cout, ky, kx]
    for (i, ty), ky in kmap.items():
        for (j, tx), kx in kmap.items():
            w2[:, i, j, :, ty + 1, tx + 1] = wt[:, :, ky, kx]
    return w2.reshape(cin * 4, cout, 3, 3).contiguous()


class _DgradCache:
    def __init__(self):
        self.key, self.packed = None, None

    def get(self, weight: torch.Tensor, stride: int) -> ops.PackedConv:
        key = (ops.tensor_version(weight), weight.data_ptr(), stride)
        if key != self.key:
            self.packed = ops.PackedConv(_dgrad_weight(weight.detach(), stride), None)
            self.key = key
        return self.packed


class ConvFn(torch.autograd.Function):
    """y = conv(x, W) + b (+ res); `shuffle2` stores through PixelShuffle(2)."""

    @staticmethod
    def forward(ctx, x, weight, bias, res, conv, shuffle2):
        y = ops.conv2d(x, conv.packed(), conv.stride, shuffle2=shuffle2, res=res)
        ctx.save_for_backward(x, weight)
        ctx.conv, ctx.shuffle2, ctx.has_bias, ctx.has_res = conv, shuffle2, bias is not None, res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        conv = ctx.conv
        dy = dy.contiguous()
        dyc = ops.pixel_unshuffle2(dy) if ctx.shuffle2 else dy
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if not hasattr(conv, "_dgradCache"):
                conv._dgradCache = _DgradCache()
            wt = conv._dgradCache.get(weight, conv.stride)
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
        return dx, dw, db, (dy if ctx.has_res else None), None, None


class SiluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.silu(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.silu_bwd(x, dy.contiguous())


class GateFn(torch.autograd.Function):
    """out = a * sigmoid(b) + x."""

    @staticmethod
    def forward(ctx, a, b, x):
        ctx.save_for_backward(a, b)
        return ops.gate(a, b, x)

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        dout = dout.contiguous()
        da, db = ops.gate_bwd(a, b, dout)
        return da, db, dout


class AxpbyFn(torch.autograd.Function):
    """out = alpha * a + beta * b (alpha, beta in {+1, -1})."""

    @staticmethod
    def forward(ctx, a, b, alpha, beta):
        ctx.alpha, ctx.beta = alpha, beta
        return ops.axpby(a, b, alpha, beta)

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        da = dout if ctx.alpha == 1.0 else ops.axpby(dout, dout, ctx.alpha, 0.0)
        db = dout if ctx.beta == 1.0 else ops.axpby(dout, dout, ctx.beta, 0.0)
        return da, db, None, None


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


class GdnFn(torch.autograd.Function):
    """y = x * f(beta + gamma @ x^2) on the FOLDED (non-negative) beta [C], gamma [C, C]."""

    @staticmethod
    def forward(ctx, x, beta, gamma, inverse):
        packed = ops.PackedConv(gamma.detach()[..., None, None], beta.detach())
        y = ops.conv2d(x, packed, square_in=True, igdn_mul=x) if inverse else ops.conv2d(x, packed, square_in=True, gdn_mul=x)
        ctx.save_for_backward(x, gamma)
        ctx.packed, ctx.inverse = packed, inverse
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        dy = dy.contiguous()
        s = ops.conv2d(x, ctx.packed, square_in=True)                          # beta + gamma @ x^2, recomputed
        dxd, ds = ops.gdn_bwd_prep(x, s, dy, ctx.inverse)
        back = ops.PackedConv((2.0 * gamma.detach().t().contiguous())[..., None, None], None)
        dx = ops.conv2d(ds, back, mul=x, res=dxd)                              # dy f(s) + 2 x (gamma^T ds)
        dgamma, dbeta = ops.conv2d_wgrad(x, ds, 1, 1, square_x=True, want_bias=True)
        dgamma = dgamma[:, :, 0, 0]
        return dx, dbeta, dgamma, None


def conv(x, module, res: Optional[torch.Tensor] = None, shuffle2: bool = False):
    return ConvFn.apply(x, module.weight, module.bias, res, module, shuffle2)


def silu(x):
    return SiluFn.apply(x)


def gate(a, b, x):
    return GateFn.apply(a, b, x)


def add(a, b):
    return AxpbyFn.apply(a, b, 1.0, 1.0)


def sub(a, b):
    return AxpbyFn.apply(a, b, 1.0, -1.0)


def gdn(x, module, inverse: bool):
    """GenDivNorm / InvGenDivNorm with the reference's re-parametrisation inside the graph (nn/base.py:81-84)."""
    beta = LowerBoundFn.apply(module.beta, module.beta_reparam.lowerBound.bound) ** 2 - module.beta_reparam.eps
    gamma = LowerBoundFn.apply(module.gamma, module.gamma_reparam.lowerBound.bound) ** 2 - module.gamma_reparam.eps
    return GdnFn.apply(x, beta, gamma, inverse)


class SoftQuantizeFn(torch.autograd.Function):
    """One level of the training quantizer (reference: _multiCodebookQuantization.forward + the two
    _multiCodebookDeQuantization.forward calls on its sample, quantizer.py:181-239,262-274):
        logit = (-dist / sqrt(k)) * max(T, eps) ; random drop ; sample = gumbelSoftmax(logit, hard=True)
        deq   = sample @ codebook                       (differentiable output; used by the residual AND the decoder)
        code  = argmax(logit), logits                   (non-differentiable outputs)
    Backward: straight-through -- the gradient reaches `sample` through y_soft only -- then through the logits to the
    latent, the codebook (distance terms + the sample @ codebook product) and the temperature."""

    @staticmethod
    def forward(ctx, x, codebook, temperature, freq_ema, u_drop, u_gumbel, drop_exponent, packed, bound):
        logits = ops.vq_logits(x, packed, temperature, bound)
        code, index, hot = ops.vq_gumbel_sample(logits, u_drop, u_gumbel, freq_ema, drop_exponent)
        deq = ops.vq_dequant_soft(index, hot, packed)
        ctx.save_for_backward(x, logits, u_gumbel, index, hot, temperature)
        ctx.packed, ctx.bound = packed, bound
        ctx.mark_non_differentiable(code, logits)
        return deq, code, logits

    @staticmethod
    def backward(ctx, ddeq, _dcode, _dlogits):
        x, logits, u_gumbel, index, hot, temperature = ctx.saved_tensors
        packed = ctx.packed
        ddeq = ddeq.contiguous()
        ds = ops.vq_inner(ddeq, packed)                                        # dSample = dDeq . C^T
        rowsum, dtrow = ops.vq_softmax_bwd(logits, u_gumbel, ds, temperature, ctx.bound)   # ds now holds d dist
        dx, dcb = ops.vq_soft_bwd(ds, rowsum, x, ddeq, index, hot, packed)
        dtb = ops.channel_sum(dtrow)                                           # [m]: d max(T, bound)
        t = temperature.detach().reshape(-1)
        dt = (((t >= ctx.bound) | (dtb < 0)).to(dtb.dtype) * dtb).reshape(temperature.shape)   # LowerBound's rule
        return dx, dcb, dt, None, None, None, None, None, None
