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
from .entropyCoder import EntropyCoder, VariousMCoder

EPS = 1e-6
_TORCH_RAND = __import__("os").environ.get("MCQUIC_AMD_TORCH_RAND", "0") == "1"      # A/B switch: draw the uniforms with torch.rand


class _CodebookCache:
    """Packed (MFMA operand order) view of a codebook Parameter, rebuilt when the Parameter changes."""

    def __init__(self):
        self._packed: Optional[ops.PackedCodebook] = None
        self._key = None

    def invalidate(self):
        self._packed, self._key = None, None

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

    def __init__(self, codebook: nn.Parameter, cache: _CodebookCache, freqEMA: Optional[nn.Parameter] = None):
        super().__init__()
        self._m, self._k, self._d = codebook.shape
        self._codebook = codebook
        self._temperature = nn.Parameter(torch.ones((self._m, 1, 1, 1)))
        if freqEMA is not None:        # ResidualBackwardQuantizer hands the level's EMA over (quantizer.py:607): a shared
            self._freqEMA = freqEMA    # Parameter, visible in the state_dict under this module as well (`_freqEMA`)
        self._bound = LowerBound(EPS)
        self._cache = [cache]          # list: keep the cache out of nn.Module's attribute registration

    @torch.no_grad()
    def reAssignCodebook(self, freq: torch.Tensor, priority: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Dead codewords (normalised frequency < 1e-6) are overwritten by the most used ones: the i-th dead codeword in
        index order receives the i-th most frequent codeword of its group; when more than k // 2 of a group are dead only
        a random k // 2 of them are refilled (reference behaviour: mcquic/modules/quantizer.py:111-136, driven by the
        CodebookReassign hook, mcquic/train/hooks.py:100-121).  Returns the flattened [m * k] mask of codewords that moved
        by more than 1e-4 in squared distance.

        One batched device-side pass over all m groups -- two sorts, a cumulative count and a gather -- with no host
        read-back (the reference loops over groups and syncs twice per group).  `priority` [m, k]: the dead codewords with
        the smallest priorities are the ones refilled when there are too many (default: fresh uniform draws; the parity
        test passes the ranks of the reference's `torch.randperm` draw)."""
        old = self._codebook.detach()
        dev = old.device
        freq = freq.detach().to(dev, torch.float32)
        m, k, d = old.shape
        half = k // 2
        dead = freq < EPS                                                       # [m, k]
        if priority is None:
            priority = torch.rand((m, k), device=dev)
        # rank of every codeword when the dead ones are lined up by priority (live ones pushed behind them)
        lineup = torch.where(dead, priority.to(dev, torch.float32), torch.full_like(freq, float("inf"))).argsort(dim=-1, stable=True)
        place = torch.empty_like(lineup).scatter_(-1, lineup, torch.arange(k, device=dev).expand(m, k))
        refill = dead & (place < half)                                          # at most k // 2 per group
        # popularity order: live codewords by frequency, then the refilled dead ones (0), then the skipped dead ones (-1);
        # a group with few dead codewords keeps its measured frequencies untouched, like the reference
        crowded = dead.sum(-1, keepdim=True) > half
        score = torch.where(crowded & dead, torch.where(refill, torch.zeros_like(freq), -torch.ones_like(freq)), freq)
        donors = score.argsort(dim=-1, descending=True, stable=True)                # [m, k] codeword indices, most used first
        slot = (refill.cumsum(-1) - 1).clamp_(min=0)                            # i for the i-th refilled codeword of a group
        source = donors.gather(-1, slot)                                        # donor index per codeword position
        moved = old.gather(1, source[..., None].expand(m, k, d))
        fresh = torch.where(refill[..., None], moved, old)
        changed = ((fresh - old) ** 2).sum(-1) > 1e-4
        self._store(fresh)
        return changed.flatten()

    @torch.no_grad()
    def syncCodebook(self):
        """Every rank takes rank 0's codebook (reference: quantizer.py:138-142): one broadcast of a detached copy, written
        back in place so the Parameter object (and the optimizer state keyed on it) stays the same."""
        import torch.distributed as dist
        from ..parallel import _staged
        buf = _staged(self._codebook.detach().clone())
        dist.broadcast(buf, 0)
        self._store(buf.to(self._codebook.device))

    def _store(self, value: torch.Tensor):
        """In-place update of the shared codebook Parameter that the packed-operand cache notices: `copy_` on the Parameter
        itself bumps its version counter (`.data.copy_` does not), and the cache is dropped explicitly as well, for
        parameters created under torch.inference_mode(), which carry no version counter."""
        self._codebook.copy_(value)
        self._cache[0].invalidate()

    def encode(self, x: torch.Tensor) -> torch.Tensor:
        """[n, m*d, h, w] -> int64 [n, m, h, w] = argmin_k ((x2 + c2) - 2 x.c), first index on ties (:144-179)."""
        return ops.vq_assign(x, self._cache[0].get(self._codebook))

    def forward(self, x: torch.Tensor, freqEMA: torch.Tensor, uniforms=None, step=None, codebook=None):
        """Training-mode forward (:181-239), forward values only (no autograd graph yet).
        `codebook`: an alias of `self._codebook` to differentiate through (a quantizer whose codebook is shared by several levels
        hands every level its own alias so that their gradients are summed by one node, autograd.fork).

        logit = (-dist / sqrt(k)) * max(temperature, eps); random drop against the level's frequency EMA;
        gumbelSoftmax(hard=True); code = argmax(logit).  The straight-through sample y_hard - y_soft + y_soft is
        zero except at its arg-max, so it is carried as (index, hot value) instead of a dense [n, m, h, w, k]
        tensor.  `uniforms` = (u_drop, u_gumbel), the reference's two `torch.rand_like(logit)` draws; without them the
        draws are made inside the kernels that use them, from a generator snapshot (2 x 134 MB per training step that are
        never written or read; MCQUIC_AMD_TORCH_RAND=1 draws the tensors with torch.rand instead, the round-3 form).
        `step` = this level's share of the cascade's prologue launch (ops.VqStep.level: drop exponent, generator snapshot,
        code-count buffer); a stand-alone call makes its own.  Returns ((index, hot), code, logit)."""
        cb = self._cache[0].get(self._codebook)
        n, _, h, w = x.shape
        if uniforms is None and (_TORCH_RAND or not x.is_cuda):
            shape = (n, self._m, h, w, self._k)
            uniforms = (torch.rand(shape, device=x.device), torch.rand(shape, device=x.device))
        if step is None:
            # exponent of _randomDrop (:196-198) and the generator snapshot, made on the device (no host sync in the step)
            step = ops.vq_step_prologue([freqEMA], EPS, want_rng=uniforms is None).level(0)
        exponent, rng, counts = step
        if uniforms is None:
            uniforms = (None, None)
        else:
            rng = None
        if torch.is_grad_enabled():
            from ..autograd import SoftQuantizeFn
            deq, code, logit, sdeq = SoftQuantizeFn.apply(x, self._codebook if codebook is None else codebook, self._temperature, freqEMA.detach(), uniforms[0], uniforms[1],
                                                          exponent, cb, float(EPS), rng, counts)
            ops.set_silu_twin(deq, sdeq)
            return deq, code, logit           # the sample is represented by its (differentiable) dequantisation
        logit = ops.vq_logits(x, cb, self._temperature, float(EPS))
        code, index, hot = ops.vq_gumbel_sample(logit, uniforms[0], uniforms[1], freqEMA, exponent, rng, counts)
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
        from ..nn import blocks
        z = self._latentStageEncoder(x)
        head = self._latentHead
        if head is None:
            return None, self._quantizer.encode(self._quantizationHead(z))
        if blocks.lockstep_ok([self._quantizationHead, head], [z, z]):
            # latentHead(z) does not depend on the codes until its closing conv: everything before that runs in lockstep with
            # quantizationHead (same layer shapes on the same z: one multi-problem launch per layer, four problems in the
            # AttentionBlocks)
            last = len(head) - 1
            q, t = blocks.lockstep_infer([self._quantizationHead, head], [z, z], layers=last)
            code = self._quantizer.encode(self._quantizationHead[last](q))
        else:
            code = self._quantizer.encode(self._quantizationHead(z))
            t = z
            for i in range(len(head) - 1):
                t = head[i](t)
        deq = self._dequantizer.decode(code)
        # z' - dequant(code): the subtraction is the epilogue of latentHead's closing conv3x3 (:318)
        return head[len(head) - 1](t, res=deq, res_scale=-1.0, dual_silu=True), code


    def _forward(self, x: torch.Tensor, freqEMA: torch.Tensor, uniforms=None, step=None):
        """Training-mode level (:295-305): returns (sample, residual for the next level, code, logit)."""
        z = self._latentStageEncoder(x)
        if self._latentHead is None:
            q, code, logit = self._quantizer(self._quantizationHead(z), freqEMA, uniforms, step)
            return q, None, code, logit
        head = self._latentHead
        if torch.is_grad_enabled():
            # latentHead(z) does not depend on the codes: it runs in lockstep with quantizationHead -- same layer shapes,
            # one multi-problem launch per layer for both (autograd.LockstepFn; 15 + 15 convolutions, launch-bound on the
            # 16x16 ... 4x4 maps of a training crop, become 10 launches each way)
            from .. import autograd as AG
            hz, qin = AG.lockstep([head, self._quantizationHead], [z, z])
            q, code, logit = self._quantizer(qin, freqEMA, uniforms, step)
            # z' - dequant (:303); the sample's second use (the decoder) goes through the same node, so that its two gradients
            # meet in one launch of ours
            residual, q = AG.sub_pass(hz, self._dequantizer(q))
            return q, residual, code, logit
        q, code, logit = self._quantizer(self._quantizationHead(z), freqEMA, uniforms, step)
        deq = self._dequantizer(q)
        t = z
        for i in range(len(head) - 1):
            t = head[i](t)
        return q, head[len(head) - 1](t, res=deq, res_scale=-1.0, dual_silu=True), code, logit

    def forward(self, x: torch.Tensor, freqEMA: torch.Tensor, uniforms=None, step=None):
        return self._forward(x, freqEMA, uniforms, step)


class _quantizerDecoder(nn.Module):
    """reference: quantizer.py:330-365."""

    def __init__(self, dequantizer, dequantizationHead, sideHead, restoreHead):
        super().__init__()
        self._dequantizer = dequantizer
        self._dequantizationHead = dequantizationHead
        self._sideHead = sideHead
        self._restoreHead = restoreHead

    def decode(self, code: torch.Tensor, formerLevel: Optional[torch.Tensor]):
        from ..nn import blocks
        deq = self._dequantizer.decode(code, dual_silu=True)
        if self._sideHead is None:
            return self._restoreHead(self._dequantizationHead(deq))
        if blocks.lockstep_ok([self._dequantizationHead, self._sideHead], [deq, formerLevel]):
            q, side = blocks.lockstep_infer([self._dequantizationHead, self._sideHead], [deq, formerLevel])    # independent until their sum
        else:
            q, side = self._dequantizationHead(deq), self._sideHead(formerLevel)
        return self._restoreHead(ops.add(q, side, dual_silu=True))                # q + sideHead(formerLevel) (:354)


    def forward(self, q, formerLevel: Optional[torch.Tensor]):
        """Training-mode level (:359-365): like decode, from the straight-through sample."""
        if self._sideHead is not None and torch.is_grad_enabled():
            from .. import autograd as AG                      # sideHead and dequantizationHead in lockstep (see _quantizerEncoder._forward)
            x, side = AG.lockstep([self._dequantizationHead, self._sideHead], [self._dequantizer(q), formerLevel])
            return self._restoreHead(AG.add(x, side, dual_silu=True))
        from ..nn import blocks
        x = blocks.run_stack(self._dequantizationHead, self._dequantizer(q))
        if self._sideHead is not None:
            if torch.is_grad_enabled():
                from .. import autograd as AG
                x = AG.add(x, self._sideHead(formerLevel), dual_silu=True)
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

    def _stepPrologue(self, x: torch.Tensor, uniforms):
        """The per-step bookkeeping of all levels in ONE launch (ops.vq_step_prologue): every level's drop exponent and generator
        snapshot, and the zeroed code-count buffer of the entropy coder that the sampling kernels add into.  HIP devices only:
        the training forward has no CPU / PyTorch path (a level run on its own makes its own prologue launch, equally on the device)."""
        if not x.is_cuda:
            raise NotImplementedError(f"mcquic_amd: the training forward of {type(self).__name__} runs on a HIP device only (input on "
                                      f"{x.device}); there is no CPU / PyTorch fallback -- move the model and the batch to cuda")
        coder = self._entropyCoder
        want_rng = uniforms is None and not _TORCH_RAND
        return ops.vq_step_prologue(list(coder._freqEMA), EPS, want_rng, coder.countBuffer(x.device))

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

    def compress(self, x: torch.Tensor):
        """`encode` + the entropy coder (mcquic/modules/entropyCoder.py:108-126) with the coder taken level by level: a level's codes
        leave for the host (a copy in stream order, then a host thread) as soon as they are enqueued, while the GPU computes the levels below it --
        level 0 holds three quarters of the symbols and is ready first.  Same codes, same bytes as encode() + coder.compress()."""
        from .entropyCoder import CODER_OVERLAP
        if not (CODER_OVERLAP and x.is_cuda):
            return super().compress(x)
        job = self._entropyCoder.beginCompress(len(self._encoders))
        codes = []
        for lv, encoder in enumerate(self._encoders):
            x, code = encoder.encode(x)
            codes.append(code)
            job.submit(lv, code)
        binaries, codeSize = job.finish()
        return codes, binaries, codeSize

    def decompress(self, binaries, codeSize) -> torch.Tensor:
        """The coder's streams -> `decode` (entropyCoder.py:141-154) with the levels decoded on a host thread in the order the decoder
        cascade consumes them -- the last (smallest) level first: its kernels are enqueued while the larger levels are still being
        decoded."""
        from .entropyCoder import CODER_OVERLAP
        if not (CODER_OVERLAP and self._entropyCoder._freqEMA[0].is_cuda):
            return super().decompress(binaries, codeSize)
        job = self._entropyCoder.beginDecompress(binaries, codeSize)
        formerLevel = None
        for lv in reversed(range(len(self._decoders))):
            formerLevel = self._decoders[lv].decode(job.level(lv), formerLevel)
        return formerLevel

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
        st = self._stepPrologue(x, uniforms)
        for lv, encoder in enumerate(self._encoders):
            quantized, x, code, logit = encoder(x, self._entropyCoder._freqEMA[lv], None if uniforms is None else uniforms[lv],
                                                None if st is None else st.level(lv))
            quantizeds.append(quantized)
            codes.append(code)
            logits.append(logit)
        formerLevel = None
        for decoder, quantized in zip(self._decoders[::-1], quantizeds[::-1]):
            formerLevel = decoder(quantized, formerLevel)
        self._entropyCoder(codes, counts=None if st is None else st.counts)
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



class VariousMQuantizer(BaseQuantizer):
    """reference: quantizer.py:88-93 (group count per level, statistics in a VariousMCoder)."""

    def __init__(self, m: List[int], k: List[int]):
        nn.Module.__init__(self)
        self._entropyCoder = VariousMCoder(m, k)
        self._m = m
        self._k = k


class _plainHeadEncoder(_quantizerEncoder):
    """`_quantizerEncoder` whose `quantizationHead` / `latentHead` are nn.Identity() (NeonQuantizer, quantizer.py:490-491): the codes
    come straight from the latent-stage encoder's output, and EVERY level -- the last one too -- returns z - dequant(code)
    (quantizer.py:310-318 with `latentHead` never None)."""

    def encode(self, x: torch.Tensor):
        z = self._latentStageEncoder(x)
        code = self._quantizer.encode(z)
        return ops.axpby(z, self._dequantizer.decode(code), 1.0, -1.0), code

    def forward(self, *a, **kw):
        raise NotImplementedError("NeonQuantizer's training forward raises in the reference snapshot too (it hands the float 0.5 to "
                                  "_multiCodebookQuantization where the frequency EMA tensor belongs, quantizer.py:493,194-200)")


class _plainHeadDecoder(_quantizerDecoder):
    """`_quantizerDecoder` with nn.Identity() `dequantizationHead` / `sideHead` (quantizer.py:497-498): q + formerLevel, restoreHead."""

    def decode(self, code: torch.Tensor, formerLevel: Optional[torch.Tensor]):
        q = self._dequantizer.decode(code)
        return self._restoreHead(q if self._sideHead is None else ops.add(q, formerLevel))

    def forward(self, *a, **kw):
        raise NotImplementedError("NeonQuantizer's training forward raises in the reference snapshot too (quantizer.py:493)")


class NeonQuantizer(VariousMQuantizer):
    """reference: quantizer.py:469-573 -- the one quantizer class of that file no model or config of the snapshot instantiates: a
    32-channel cascade like UMGMQuantizer's with a group count PER LEVEL (`m[i]` codebooks of 32 / m[i] dimensions, k[i] codewords),
    identity heads, and a latent-stage encoder / restore head per level (ResidualBlock, AttentionBlock, strided / shuffle block, a
    bias-free 1x1 convolution).  Same module tree and state_dict keys; inference only (encode / decode / compress / decompress,
    Codebooks, reAssignCodebook): the reference's own training forward raises (see `_plainHeadEncoder.forward`)."""

    def __init__(self, m: List[int], k: List[int]):
        if not isinstance(k, list):
            raise AttributeError
        from ..nn import AttentionBlock, ResidualBlock, ResidualBlockShuffle, ResidualBlockWithStride, conv1x1
        super().__init__(m, k)
        encoders, decoders = [], []
        for i, (ki, mi) in enumerate(zip(k, m)):
            codebook = nn.Parameter(nn.init.normal_(torch.empty(mi, ki, 32 // mi), std=math.sqrt(2 / (5 * 32 / float(mi)))))
            cache = _CodebookCache()
            quantizer = _multiCodebookQuantization(codebook, cache)
            dequantizer = _multiCodebookDeQuantization(codebook, cache)
            lse = nn.Sequential(ResidualBlock(32, 32), AttentionBlock(32), ResidualBlockWithStride(32, 32), conv1x1(32, 32, bias=False))
            restore = nn.Sequential(conv1x1(32, 32, bias=False), ResidualBlockShuffle(32, 32), AttentionBlock(32), ResidualBlock(32, 32))
            encoders.append(_plainHeadEncoder(quantizer, dequantizer, lse, nn.Identity(), nn.Identity()))
            decoders.append(_plainHeadDecoder(dequantizer, nn.Identity(), nn.Identity() if i < len(k) - 1 else None, restore))
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
        formerLevel = None
        for decoder, code in zip(self._decoders[::-1], codes[::-1]):
            formerLevel = decoder.decode(code, formerLevel)
        return formerLevel

    def reAssignCodebook(self) -> torch.Tensor:
        reassigned = [encoder.reAssignCodebook(freq) for encoder, freq in zip(self._encoders, self.NormalizedFreq)]
        return torch.cat(reassigned).float().mean()

    def syncCodebook(self):
        import torch.distributed as dist
        dist.barrier()
        for encoder in self._encoders:
            encoder.syncCodebook()

    def forward(self, x: torch.Tensor):
        raise NotImplementedError("NeonQuantizer's training forward raises in the reference snapshot too (it hands the float 0.5 to "
                                  "_multiCodebookQuantization where the frequency EMA tensor belongs, quantizer.py:493,194-200)")


class ResidualBackwardQuantizer(VariousMQuantizer):
    """The quantizer of `Neon` (reference: quantizer.py:577-765): ONE codebook [1, k, 8] shared by all levels; every level has
    a latent-stage encoder (down), a `backward` stack and a `restoreHead` (both up); codes are produced from the SMALLEST
    level up, each level quantizing what the coarser levels' `backward` stacks did not explain, and come out small -> large.
    Same module tree / state_dict keys as the reference (`_encoders`, `_decoders`, `_backwards`, `_quantizers`,
    `_dequantizers`); the layers are the HIP-backed ones of mcquic_amd.nn."""

    def __init__(self, k: int, size: List[int], denseNorm: bool = False):
        from ..nn import AttentionBlock, ResidualBlock, ResidualBlockShuffle, ResidualBlockWithStride, conv1x1
        channel = 8
        self.channel = channel
        super().__init__([1] * len(size), [k] * len(size))
        codebook = nn.Parameter(nn.init.trunc_normal_(torch.empty(1, k, channel), std=math.sqrt(2 / (5 * channel))))
        cache = _CodebookCache()
        encoders, backwards, decoders, quantizers, dequantizers = [], [], [], [], []
        lastSize = size[0] * 2
        for i, thisSize in enumerate(size):
            if thisSize == lastSize // 2:
                down = lambda: ResidualBlockWithStride(channel * 4, channel * 4, 2, 1, denseNorm)       # noqa: E731
                up = lambda: ResidualBlockShuffle(channel * 4, channel * 4, 2, 1, denseNorm)            # noqa: E731
            elif thisSize == lastSize:
                down = up = lambda: ResidualBlock(channel * 4, channel * 4, 1, denseNorm)               # noqa: E731
            else:
                raise ValueError("The given size sequence does not half or equal to from left to right.")
            lastSize = thisSize

            def upStack():
                return nn.Sequential(conv1x1(channel, channel * 4, bias=False), up(), AttentionBlock(channel * 4, 1, denseNorm),
                                     ResidualBlock(channel * 4, channel, 1, denseNorm))
            encoders.append(nn.Sequential(ResidualBlock(channel, channel * 4, 1, denseNorm), AttentionBlock(channel * 4, 1, denseNorm),
                                          down(), conv1x1(channel * 4, channel, bias=False)))
            backwards.append(upStack() if i < len(size) - 1 else nn.Identity())
            decoders.append(upStack())
            # NOTE (reference): quantizers run large -> small, `_freqEMA` is stored small -> large
            quantizers.append(_multiCodebookQuantization(codebook, cache, self._entropyCoder._freqEMA[-(i + 1)]))
            dequantizers.append(_multiCodebookDeQuantization(codebook, cache))
        self._encoders = nn.ModuleList(encoders)
        self._decoders = nn.ModuleList(decoders)
        self._backwards = nn.ModuleList(backwards)
        self._quantizers = nn.ModuleList(quantizers)
        self._dequantizers = nn.ModuleList(dequantizers)

    @property
    def Codebooks(self):
        return list(quantizer._codebook for quantizer in self._quantizers)

    def _latents(self, x: torch.Tensor) -> List[torch.Tensor]:
        latents = []
        grad = self.training and torch.is_grad_enabled()
        for i, encoder in enumerate(self._encoders):
            x = encoder(x)
            if grad and i + 1 < len(self._encoders):
                from .. import autograd as AG
                x, keep = AG.fork(x, 2)            # (a level's latent feeds the next level AND its own residual: one node sums the two gradients)
                latents.append(keep)
            else:
                latents.append(x)
        return latents

    def encode(self, x: torch.Tensor) -> List[torch.Tensor]:
        """quantizer.py:676-694."""
        latents = self._latents(x)
        codes, current = [], None
        for quantizer, dequantizer, backward, latent in zip(self._quantizers[::-1], self._dequantizers[::-1], self._backwards[::-1],
                                                            latents[::-1]):
            residual = latent if current is None else ops.axpby(latent, current, 1.0, -1.0)      # (latent - 0 is latent, bit for bit)
            code = quantizer.encode(residual)
            codes.append(code)
            current = backward(dequantizer.decode(code))
        return codes

    def decode(self, codes: List[torch.Tensor]) -> Optional[torch.Tensor]:
        """quantizer.py:696-704."""
        if len(codes) != len(self._decoders):
            raise RuntimeError(f"expected {len(self._decoders)} code levels, got {len(codes)}")
        self._entropyCoder._checkShape(codes)
        formerLevel = None
        for decoder, dequantizer, code in zip(self._decoders[::-1], self._dequantizers[::-1], codes):
            quantized = dequantizer.decode(code)
            formerLevel = decoder(quantized if formerLevel is None else ops.add(quantized, formerLevel))
        return formerLevel

    def residual_backward(self, code: torch.Tensor, level: int) -> torch.Tensor:
        """quantizer.py:671-674."""
        return self._backwards[-level](self._dequantizers[-level].decode(code))

    def residual_forward(self, code: torch.Tensor, formerLevel: Optional[torch.Tensor], level: int) -> torch.Tensor:
        """quantizer.py:706-713."""
        if formerLevel is None and level > 0:
            raise RuntimeError("For reconstruction after level-0, you should provide not None formerLevel as input.")
        if formerLevel is not None and level == 0:
            raise RuntimeError("For reconstruction at level-0, you should provide None formerLevel as input.")
        decoder, dequantizer = self._decoders[-(level + 1)], self._dequantizers[-(level + 1)]
        quantized = dequantizer.decode(code)
        return decoder(ops.add(quantized, formerLevel)) if formerLevel is not None else decoder(quantized)

    def reAssignCodebook(self) -> torch.Tensor:
        """quantizer.py:715-721."""
        reassigned = [quantizer.reAssignCodebook(freq) for quantizer, freq in zip(self._quantizers, self.NormalizedFreq)]
        return torch.cat(reassigned).float().mean()

    def syncCodebook(self):
        """quantizer.py:723-726."""
        import torch.distributed as dist
        dist.barrier()
        for quantizer in self._quantizers:
            quantizer.syncCodebook()

    def forward(self, x: torch.Tensor, uniforms=None):
        """Training-mode forward (quantizer.py:727-765): (yHat, codes, logits), codes / logits small -> large;
        `uniforms[j]` = (u_drop, u_gumbel) of the j-th quantization (smallest level first)."""
        from .. import autograd as AG
        grad = torch.is_grad_enabled()
        latents = self._latents(x)
        quantizeds, codes, logits = [], [], []
        current = None
        st = self._stepPrologue(x, uniforms)          # (the j-th quantization reads `_entropyCoder._freqEMA[j]`, see __init__)
        # ONE codebook under every level: each level differentiates through its own alias and one node sums the gradients
        cbs = AG.fork(self._quantizers[0]._codebook, len(self._quantizers)) if grad else [None] * len(self._quantizers)
        for j, (quantizer, dequantizer, backward, latent) in enumerate(zip(self._quantizers[::-1], self._dequantizers[::-1],
                                                                           self._backwards[::-1], latents[::-1])):
            if current is None:
                residual = latent
            else:
                residual = AG.sub(latent, current) if grad else ops.axpby(latent, current, 1.0, -1.0)
            sample, code, logit = quantizer(residual, quantizer._freqEMA, None if uniforms is None else uniforms[j],
                                            None if st is None else st.level(j), cbs[j])
            quantized = dequantizer(sample)
            # (two consumers: the decoder side and the `backward` stack -- or, behind the last level's identity, the next residual)
            quantized, forBackward = AG.fork(quantized, 2) if grad else (quantized, quantized)
            quantizeds.append(quantized)
            codes.append(code)
            logits.append(logit)
            current = backward(forBackward)
        formerLevel = None
        for decoder, quantized in zip(self._decoders[::-1], quantizeds):
            if formerLevel is None:
                formerLevel = decoder(quantized)                                  # (0 + quantized is quantized, bit for bit)
            else:
                formerLevel = decoder(AG.add(formerLevel, quantized) if grad else ops.add(formerLevel, quantized))
        self._entropyCoder(codes, counts=None if st is None else st.counts)
        return formerLevel, codes, logits
