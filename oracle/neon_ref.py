"""CPU oracle for the `Neon` model family: a plain-PyTorch (fp32, CPU) restatement of
`mcquic.modules.compressor.Neon` (compressor.py:181-241) and `ResidualBackwardQuantizer` (quantizer.py:577-765).

TEST INFRASTRUCTURE ONLY (see oracle/mcquic_ref.py for the rules): used by tests/ as the checker, pinned against the
real reference imported from /root/reference (tests/test_oracle_vs_reference.py, golden F10 from tests/golden/make_golden.py).

Functional over the reference's `state_dict` layout:
  _encoder.{0..15}, _decoder.{0..16}                                       compressor.py:184-225
  _quantizer._encoders.{i}.{0: RB(8->32), 1: Attn(32), 2: RBStride | RB(32->32), 3: conv1x1(32->8, no bias)}
  _quantizer._backwards.{i} / _decoders.{i}.{0: conv1x1(8->32, no bias), 1: RBShuffle | RB, 2: Attn, 3: RB(32->8)}
  _quantizer._quantizers.{i}.{_codebook [1,k,8] (ONE tensor shared by all levels), _temperature, _freqEMA, _bound.bound}
In this snapshot the `groups` argument of the blocks only parametrises GroupNorm under `denseNorm=True`
(mcquic/nn/blocks.py:179-200: `conv3x3(inChannels, outChannels)` is called without it), so every convolution is dense.
`denseNorm=True` (a state_dict that holds `..._branch.2.weight`: nn.GroupNorm's affine parameters where the second SiLU
was) is followed as well; the group counts are the constructor's (compressor.py:184-225: 32 for the `channel`-wide blocks,
1 for the blocks that touch the 8-channel latent; quantizer.py:600-651: 1) and are passed down as arguments.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.nn.functional as F

from . import mcquic_ref as R

StateDict = R.StateDict


def residual_block(sd: StateDict, pre: str, x: torch.Tensor, groups: int = 1) -> torch.Tensor:
    """mcquic/nn/blocks.py:162-200 + :70-78: SiLU, conv3, SiLU | GroupNorm(groups, C) (`denseNorm`, :196), conv3,
    `out += identity`; identity = conv1x1(x) when the widths differ (:179-182)."""
    out = R.conv3x3(sd, pre + "_branch.1.", F.silu(x))
    if pre + "_branch.2.weight" in sd:                     # denseNorm=True: nn.GroupNorm in place of the second activation
        out = F.group_norm(out, groups, sd[pre + "_branch.2.weight"], sd[pre + "_branch.2.bias"], 1e-5)
    else:
        out = F.silu(out)
    out = R.conv3x3(sd, pre + "_branch.3.", out)
    identity = R.conv1x1(sd, pre + "_skip.", x) if pre + "_skip.weight" in sd else x
    out += identity
    return out


def attention_block(sd: StateDict, pre: str, x: torch.Tensor, groups: int = 1) -> torch.Tensor:
    """mcquic/nn/blocks.py:245-288 over this file's residual_block (GroupNorm-aware)."""
    a = x
    for i in range(3):
        a = residual_block(sd, f"{pre}_mainBranch.{i}.", a, groups)
    b = x
    for i in range(3):
        b = residual_block(sd, f"{pre}_sideBranch.{i}.", b, groups)
    b = R.conv1x1(sd, pre + "_sideBranch.3.", b)
    out = a * torch.sigmoid(b)
    out += x
    return out


def conv1x1_nobias(sd: StateDict, pre: str, x: torch.Tensor) -> torch.Tensor:
    """conv1x1(..., bias=False) (mcquic/nn/convs.py:257-276)."""
    return F.conv2d(x, sd[pre + "weight"])


def encoder(sd: StateDict, x: torch.Tensor, pre: str = "_encoder.") -> torch.Tensor:
    """compressor.py:184-205."""
    y = R.conv3x3(sd, pre + "0.", x)
    y = attention_block(sd, pre + "1.", y, 32)
    y = residual_block(sd, pre + "2.", y, 32)
    y = residual_block(sd, pre + "3.", y, 32)
    y = R.residual_block_with_stride(sd, pre + "4.", y)
    y = residual_block(sd, pre + "5.", y, 32)
    y = R.residual_block_with_stride(sd, pre + "6.", y)
    y = residual_block(sd, pre + "7.", y, 32)
    y = R.residual_block_with_stride(sd, pre + "8.", y)
    y = attention_block(sd, pre + "9.", y, 32)
    for i in range(10, 14):
        y = residual_block(sd, f"{pre}{i}.", y, 32)
    y = residual_block(sd, pre + "14.", y, 1)              # ResidualBlock(2 * channel, 8, 1, denseNorm) (:203)
    return attention_block(sd, pre + "15.", y, 1)


def decoder(sd: StateDict, y: torch.Tensor, pre: str = "_decoder.") -> torch.Tensor:
    """compressor.py:206-225."""
    x = attention_block(sd, pre + "0.", y, 1)
    x = residual_block(sd, pre + "1.", x, 1)               # ResidualBlock(8, 2 * channel, 1, denseNorm) (:207)
    for i in range(2, 6):
        x = residual_block(sd, f"{pre}{i}.", x, 32)
    x = attention_block(sd, pre + "6.", x, 32)
    x = residual_block(sd, pre + "7.", x, 32)
    x = R.residual_block_shuffle(sd, pre + "8.", x)
    x = residual_block(sd, pre + "9.", x, 32)
    x = R.residual_block_shuffle(sd, pre + "10.", x)
    x = residual_block(sd, pre + "11.", x, 32)
    x = R.residual_block_shuffle(sd, pre + "12.", x)
    x = residual_block(sd, pre + "13.", x, 32)
    x = residual_block(sd, pre + "14.", x, 32)
    x = attention_block(sd, pre + "15.", x, 32)
    return R.conv3x3(sd, pre + "16.", x)


def num_levels(sd: StateDict) -> int:
    lv = 0
    while f"_quantizer._quantizers.{lv}._codebook" in sd:
        lv += 1
    return lv


def _strided(sd: StateDict, pre: str) -> bool:
    return pre + "_branch.2.beta" in sd            # a GDN sits only in the strided / shuffle variants


def latent_stage_encoder(sd: StateDict, pre: str, x: torch.Tensor) -> torch.Tensor:
    """quantizer.py:600-605 / :628-633: RB(8->32), Attn(32), RBStride(32, 32) | RB(32, 32), conv1x1(32->8, no bias)."""
    x = residual_block(sd, pre + "0.", x)
    x = attention_block(sd, pre + "1.", x)
    x = R.residual_block_with_stride(sd, pre + "2.", x) if _strided(sd, pre + "2.") else residual_block(sd, pre + "2.", x)
    return conv1x1_nobias(sd, pre + "3.", x)


def restore_stack(sd: StateDict, pre: str, x: torch.Tensor) -> torch.Tensor:
    """`backward` / `restoreHead` (quantizer.py:611-623, :639-651): conv1x1(8->32, no bias), RBShuffle | RB, Attn, RB(32->8)."""
    x = conv1x1_nobias(sd, pre + "0.", x)
    x = R.residual_block_shuffle(sd, pre + "1.", x) if _strided(sd, pre + "1.") else residual_block(sd, pre + "1.", x)
    x = attention_block(sd, pre + "2.", x)
    return residual_block(sd, pre + "3.", x)


def _backward(sd: StateDict, i: int, x: torch.Tensor) -> torch.Tensor:
    pre = f"_quantizer._backwards.{i}."
    return restore_stack(sd, pre, x) if pre + "0.weight" in sd else x      # nn.Identity() on the last level (:617, :645)


def quantizer_encode(sd: StateDict, y: torch.Tensor, collect: Optional[dict] = None) -> List[torch.Tensor]:
    """ResidualBackwardQuantizer.encode (quantizer.py:676-694): all latents first, then codes from the smallest level up,
    each level quantizing what the coarser levels' `backward` stacks did not explain.  Codes come out small -> large."""
    levels = num_levels(sd)
    latents = []
    x = y
    for i in range(levels):
        x = latent_stage_encoder(sd, f"_quantizer._encoders.{i}.", x)
        latents.append(x)
    codes = []
    current = torch.zeros_like(latents[-1])
    for i in reversed(range(levels)):
        cb = sd[f"_quantizer._quantizers.{i}._codebook"]
        residual = latents[i] - current
        if collect is not None:
            collect.setdefault("q", []).append(residual)
        code = R.vq_encode(residual, cb)
        codes.append(code)
        current = _backward(sd, i, R.vq_decode(code, cb))
    return codes


def quantizer_decode(sd: StateDict, codes: List[torch.Tensor]) -> torch.Tensor:
    """ResidualBackwardQuantizer.decode (quantizer.py:696-704)."""
    levels = num_levels(sd)
    former = None
    for j, code in enumerate(codes):
        i = levels - 1 - j
        q = R.vq_decode(code, sd[f"_quantizer._dequantizers.{i}._codebook"])
        former = restore_stack(sd, f"_quantizer._decoders.{i}.", q if former is None else q + former)
    return former


def residual_backward(sd: StateDict, code: torch.Tensor, level: int) -> torch.Tensor:
    """quantizer.py:671-674: `self._dequantizers[-level], self._backwards[-level]`."""
    levels = num_levels(sd)
    i = (-level) % levels
    return _backward(sd, i, R.vq_decode(code, sd[f"_quantizer._dequantizers.{i}._codebook"]))


def residual_forward(sd: StateDict, code: torch.Tensor, former: Optional[torch.Tensor], level: int) -> torch.Tensor:
    """quantizer.py:706-713: `self._decoders[-(level+1)]`."""
    levels = num_levels(sd)
    i = levels - 1 - level
    q = R.vq_decode(code, sd[f"_quantizer._dequantizers.{i}._codebook"])
    return restore_stack(sd, f"_quantizer._decoders.{i}.", q + former if former is not None else q)


def encode(sd: StateDict, x: torch.Tensor) -> List[torch.Tensor]:
    """BaseCompressor.encode (compressor.py:79-88)."""
    return quantizer_encode(sd, encoder(sd, R.aligned_padding(x)))


def decode(sd: StateDict, codes: List[torch.Tensor]) -> torch.Tensor:
    """BaseCompressor.decode (compressor.py:114-117)."""
    return decoder(sd, quantizer_decode(sd, codes))


def forward_train(sd: StateDict, x: torch.Tensor, uniforms):
    """BaseCompressor.forward in training mode (compressor.py:35-43) with ResidualBackwardQuantizer.forward
    (quantizer.py:727-765).  uniforms[j] = (u_drop, u_gumbel) for the j-th quantization (smallest level first).
    Returns (xHat, yHat, codes, logits, oneHotCounts)."""
    y = encoder(sd, x)
    levels = num_levels(sd)
    latents = []
    cur = y
    for i in range(levels):
        cur = latent_stage_encoder(sd, f"_quantizer._encoders.{i}.", cur)
        latents.append(cur)
    quantizeds, codes, logits, counts = [], [], [], []
    current = torch.zeros_like(latents[-1])
    for j, i in enumerate(reversed(range(levels))):
        pre = f"_quantizer._quantizers.{i}."
        cb = sd[pre + "_codebook"]
        residual = latents[i] - current
        logit = R.vq_logit(residual, cb, sd[pre + "_temperature"], sd[pre + "_bound.bound"])
        logit = R.random_drop(logit, sd[pre + "_freqEMA"], uniforms[j][0])
        sample, _, _ = R.gumbel_softmax_hard(logit, uniforms[j][1], 1.0)
        code = logit.argmax(-1, keepdim=True)
        one_hot = torch.zeros_like(logit).scatter_(-1, code, 1)
        quantized = R.dequant_soft(sample, cb)
        quantizeds.append(quantized)
        codes.append(code[..., 0].contiguous())
        logits.append(logit)
        counts.append(one_hot.sum((0, 2, 3)))
        current = _backward(sd, i, quantized)
    former = torch.zeros_like(quantizeds[0])
    for j, quantized in enumerate(quantizeds):
        i = levels - 1 - j
        former = restore_stack(sd, f"_quantizer._decoders.{i}.", former + quantized)
    return decoder(sd, former), former, codes, logits, counts


# ----------------------------------------------------------------------------------------------
# seeded synthetic weights in the reference's layout (the same generator family as mcquic_ref.make_state_dict)
# ----------------------------------------------------------------------------------------------
_DENSE_NORM = [False]        # set by make_state_dict(..., denseNorm=True) while it builds


def _rb(sd, pre, cin, cout, seed):
    R._conv_params(sd, pre + "_branch.1.", cout, cin, 3, seed)
    R._conv_params(sd, pre + "_branch.3.", cout, cout, 3, seed + 1)
    if cin != cout:
        R._conv_params(sd, pre + "_skip.", cout, cin, 1, seed + 2)
    _group_norm_params(sd, pre, cout, seed)


def _group_norm_params(sd, pre, c, seed):
    if _DENSE_NORM[0]:                                     # nn.GroupNorm's affine pair, away from its (1, 0) initial values
        import numpy as np
        sd[pre + "_branch.2.weight"] = torch.from_numpy(R._rng(pre + "_branch.2.weight", seed).uniform(0.5, 1.5, c).astype(np.float32))
        sd[pre + "_branch.2.bias"] = torch.from_numpy(R._rng(pre + "_branch.2.bias", seed).uniform(-0.1, 0.1, c).astype(np.float32))


def _attn(sd, pre, c, seed):
    R._attn(sd, pre, c, seed)                              # (the convolutions of F10's denseNorm=False state_dict, unchanged)
    for i in range(3):
        _group_norm_params(sd, f"{pre}_mainBranch.{i}.", c, seed)
        _group_norm_params(sd, f"{pre}_sideBranch.{i}.", c, seed)


def _conv_nobias(sd, pre, cout, cin, seed):
    tmp = {}
    R._conv_params(tmp, "t.", cout, cin, 1, seed)
    sd[pre + "weight"] = tmp["t.weight"]


def make_state_dict(channel: int, k: int, size: List[int], seed: int = 0, denseNorm: bool = False) -> StateDict:
    """Every tensor of `Neon(channel, k, size, denseNorm)` with seeded synthetic values, keys and shapes as the reference's
    `state_dict()`."""
    _DENSE_NORM[0] = bool(denseNorm)
    try:
        return _make_state_dict(channel, k, size, seed)
    finally:
        _DENSE_NORM[0] = False


def _make_state_dict(channel: int, k: int, size: List[int], seed: int) -> StateDict:
    sd: StateDict = {}
    c, c2, qc = channel, 2 * channel, 8
    s = seed * 100000
    # encoder (compressor.py:184-205)
    R._conv_params(sd, "_encoder.0.", c, 3, 3, s + 1)
    _attn(sd, "_encoder.1.", c, s + 10)
    _rb(sd, "_encoder.2.", c, c, s + 40)
    _rb(sd, "_encoder.3.", c, c, s + 50)
    R._rb_stride(sd, "_encoder.4.", c, s + 60)
    _rb(sd, "_encoder.5.", c, c, s + 70)
    R._rb_stride(sd, "_encoder.6.", c, s + 80)
    _rb(sd, "_encoder.7.", c, c, s + 90)
    R._rb_stride(sd, "_encoder.8.", c, s + 100)
    _attn(sd, "_encoder.9.", c, s + 110)
    _rb(sd, "_encoder.10.", c, c2, s + 140)
    _rb(sd, "_encoder.11.", c2, c2, s + 150)
    _rb(sd, "_encoder.12.", c2, c2, s + 160)
    _rb(sd, "_encoder.13.", c2, c2, s + 170)
    _rb(sd, "_encoder.14.", c2, qc, s + 180)
    _attn(sd, "_encoder.15.", qc, s + 190)
    # decoder (compressor.py:206-225)
    _attn(sd, "_decoder.0.", qc, s + 300)
    _rb(sd, "_decoder.1.", qc, c2, s + 330)
    _rb(sd, "_decoder.2.", c2, c2, s + 340)
    _rb(sd, "_decoder.3.", c2, c2, s + 350)
    _rb(sd, "_decoder.4.", c2, c2, s + 360)
    _rb(sd, "_decoder.5.", c2, c, s + 370)
    _attn(sd, "_decoder.6.", c, s + 380)
    _rb(sd, "_decoder.7.", c, c, s + 410)
    R._rb_shuffle(sd, "_decoder.8.", c, s + 420)
    _rb(sd, "_decoder.9.", c, c, s + 430)
    R._rb_shuffle(sd, "_decoder.10.", c, s + 440)
    _rb(sd, "_decoder.11.", c, c, s + 450)
    R._rb_shuffle(sd, "_decoder.12.", c, s + 460)
    _rb(sd, "_decoder.13.", c, c, s + 470)
    _rb(sd, "_decoder.14.", c, c, s + 480)
    _attn(sd, "_decoder.15.", c, s + 490)
    R._conv_params(sd, "_decoder.16.", 3, c, 3, s + 520)
    # quantizer (quantizer.py:577-668): one codebook for all levels
    g = torch.Generator().manual_seed(s + 7)
    codebook = torch.randn((1, k, qc), generator=g) * math.sqrt(2 / (5 * qc))
    levels = len(size)
    last = size[0] * 2
    for i, this in enumerate(size):
        strided = this == last // 2
        if not strided and this != last:
            raise ValueError("The given size sequence does not half or equal to from left to right.")
        last = this
        base = s + 1000 + 200 * i
        pre = f"_quantizer._encoders.{i}."
        _rb(sd, pre + "0.", qc, 4 * qc, base)
        _attn(sd, pre + "1.", 4 * qc, base + 10)
        if strided:
            R._rb_stride(sd, pre + "2.", 4 * qc, base + 40)
        else:
            _rb(sd, pre + "2.", 4 * qc, 4 * qc, base + 40)
        _conv_nobias(sd, pre + "3.", qc, 4 * qc, base + 50)
        for name, off in (("_backwards", 60), ("_decoders", 120)):
            if name == "_backwards" and i == levels - 1:
                continue                                   # nn.Identity()
            pre = f"_quantizer.{name}.{i}."
            _conv_nobias(sd, pre + "0.", 4 * qc, qc, base + off)
            if strided:
                R._rb_shuffle(sd, pre + "1.", 4 * qc, base + off + 1)
            else:
                _rb(sd, pre + "1.", 4 * qc, 4 * qc, base + off + 1)
            _attn(sd, pre + "2.", 4 * qc, base + off + 10)
            _rb(sd, pre + "3.", 4 * qc, qc, base + off + 40)
        sd[f"_quantizer._quantizers.{i}._codebook"] = codebook
        sd[f"_quantizer._quantizers.{i}._temperature"] = torch.ones((1, 1, 1, 1))
        sd[f"_quantizer._quantizers.{i}._freqEMA"] = torch.ones((1, k)) / k
        sd[f"_quantizer._quantizers.{i}._bound.bound"] = torch.tensor([R.EPS])
        sd[f"_quantizer._dequantizers.{i}._codebook"] = codebook
    for j in range(levels):                                # quantizer i reads _entropyCoder._freqEMA[-(i + 1)] (:607)
        sd[f"_quantizer._entropyCoder._freqEMA.{j}"] = sd[f"_quantizer._quantizers.{levels - 1 - j}._freqEMA"]
    return sd


# ----------------------------------------------------------------------------------------------
# NeonQuantizer (mcquic/modules/quantizer.py:469-573): a 32-channel cascade with a group count per level and identity heads.
# No model of the snapshot builds it; its encode / decode run in the reference (its training forward raises), so these two are
# what tests/golden/f15_neon_quantizer.npz pins.  Keys as in `NeonQuantizer(m, k).state_dict()` (no `_quantizer.` prefix).
# ----------------------------------------------------------------------------------------------
def neon_quantizer_levels(sd: StateDict) -> int:
    lv = 0
    while f"_encoders.{lv}._quantizer._codebook" in sd:
        lv += 1
    return lv


def neon_quantizer_encode(sd: StateDict, x: torch.Tensor, collect: Optional[dict] = None) -> List[torch.Tensor]:
    """NeonQuantizer.encode (:519-527) over _quantizerEncoder.encode (:310-318) with identity heads: every level returns z - dequant."""
    codes = []
    for i in range(neon_quantizer_levels(sd)):
        pre = f"_encoders.{i}."
        z = latent_stage_encoder(sd, pre + "_latentStageEncoder.", x)
        cb = sd[pre + "_quantizer._codebook"]
        if collect is not None:
            collect.setdefault("q", []).append(z)
        code = R.vq_encode(z, cb)
        codes.append(code)
        x = z - R.vq_decode(code, cb)
    return codes


def neon_quantizer_decode(sd: StateDict, codes: List[torch.Tensor]) -> torch.Tensor:
    """NeonQuantizer.decode (:529-533) over _quantizerDecoder.decode (:351-357): the last level has no side head."""
    levels = neon_quantizer_levels(sd)
    former = None
    for i in reversed(range(levels)):
        q = R.vq_decode(codes[i], sd[f"_decoders.{i}._dequantizer._codebook"])
        former = restore_stack(sd, f"_decoders.{i}._restoreHead.", q if i == levels - 1 else q + former)
    return former


def make_neon_quantizer_state_dict(m: List[int], k: List[int], seed: int = 0) -> StateDict:
    sd: StateDict = {}
    s = seed * 100000
    for i, (mi, ki) in enumerate(zip(m, k)):
        base = s + 300 * i
        pre = f"_encoders.{i}._latentStageEncoder."
        _rb(sd, pre + "0.", 32, 32, base)
        _attn(sd, pre + "1.", 32, base + 10)
        R._rb_stride(sd, pre + "2.", 32, base + 40)
        _conv_nobias(sd, pre + "3.", 32, 32, base + 50)
        g = torch.Generator().manual_seed(base + 7)
        codebook = torch.randn((mi, ki, 32 // mi), generator=g) * math.sqrt(2 / (5 * 32 / float(mi)))
        sd[f"_encoders.{i}._quantizer._codebook"] = codebook
        sd[f"_encoders.{i}._quantizer._temperature"] = torch.ones((mi, 1, 1, 1))
        sd[f"_encoders.{i}._quantizer._bound.bound"] = torch.tensor([R.EPS])
        sd[f"_encoders.{i}._dequantizer._codebook"] = codebook
        sd[f"_decoders.{i}._dequantizer._codebook"] = codebook
        pre = f"_decoders.{i}._restoreHead."
        _conv_nobias(sd, pre + "0.", 32, 32, base + 60)
        R._rb_shuffle(sd, pre + "1.", 32, base + 61)
        _attn(sd, pre + "2.", 32, base + 70)
        _rb(sd, pre + "3.", 32, 32, base + 100)
        sd[f"_entropyCoder._freqEMA.{i}"] = torch.ones((mi, ki)) / ki
    return sd
