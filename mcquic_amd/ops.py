"""Tensor-level wrappers over the C-ABI of libmcquic_hip.so.

PyTorch is used here only for device memory (torch.empty) and the current HIP stream; every op below
is a hand-written gfx950 kernel reached through ctypes.  CPU tensors are rejected -- there is no
fallback implementation.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import (CONV_DSILU_MUL, CONV_DUAL_SILU, CONV_MUL, CONV_GATE, CONV_GDN, CONV_IGDN, CONV_RESIDUAL, CONV_SHUFFLE2, CONV_SILU_IN,
                   CONV_SILU_OUT, CONV_SQUARE_IN, CONV_WINOGRAD, CONV_WINOGRAD2D, CONV_WINOGRAD2D16, CONV_GDN_BWD, CONV_IGDN_BWD, CONV_GATE_BWD, CONV_TAPS_LR, CONV_POST_GDN, CONV_POST_IGDN, CONV_POST_GATE, ConvDesc, check)

# OPT-IN fast path, never the default and never the headline bench: large 3x3 stride-1 layers in the Winograd F(2, 3) form
# along x (mcq_pack_conv_weight_winograd_f32 + MCQ_CONV_WINOGRAD): 2/3 of the multiplications, float32 throughout, but not
# the reference's arithmetic -- results differ from the direct form in the last bits, so near-tie code indices may move.
_WINOGRAD = os.environ.get("MCQUIC_AMD_WINOGRAD", "0") in ("1", "2")
# ... and where Cout % 128 == 0, in both directions: F(2x2, 3x3), 4/9 of the multiplications (MCQUIC_AMD_WINOGRAD=2)
_WINOGRAD_2D = os.environ.get("MCQUIC_AMD_WINOGRAD", "0") == "2"
_WINOGRAD_MIN_PIXELS = int(os.environ.get("MCQUIC_AMD_WINOGRAD_MIN_PIXELS", str(128 * 1024)))   # N*H*W below this: direct form
# (the 2-D form still wins far below that: 32 x 48x32 maps 112 -> 86 us per launch, one 192x128 map at batch 1 -- 24 k pixels --
#  7.3 -> 6.0 ms per encode+decode with the maps above it; at 12 k pixels, 32 x 24x16, it loses, 41 -> 44 us)
_WINOGRAD_MIN_PIXELS_2D = int(os.environ.get("MCQUIC_AMD_WINOGRAD_MIN_PIXELS_2D", str(20 * 1024)))
# which instance runs the F(2x2, 3x3) layers: 16 = v_mfma_f32_16x16x4_f32, two waves per SIMD (csrc/conv_wino16.hip; layers with
# Cin % 16 == 0 and a SiLU / residual / twin / PixelShuffle epilogue), 32 = the one-wave-per-SIMD 32x32x2 instance of conv_mfma.hip
_W2D_KERNEL = int(os.environ.get("MCQUIC_AMD_W2D_KERNEL", "16"))
_W16_EPILOGUES = CONV_SILU_OUT | CONV_RESIDUAL | CONV_DUAL_SILU


def winograd_enabled() -> int:
    """0 = off (the default), 1 = F(2, 3) along x, 2 = F(2x2, 3x3) where the layer allows it."""
    return (2 if _WINOGRAD_2D else 1) if _WINOGRAD else 0


def set_winograd(enabled, min_pixels: Optional[int] = None) -> None:
    """Switch the opt-in Winograd path on / off for convolutions packed from now on (nn.Conv2d re-packs on the change):
    False / 0 = off, True / 1 = F(2, 3) along x, 2 = F(2x2, 3x3) for layers with Cout % 128 == 0 (the 1-D form elsewhere).
    `min_pixels`: layers with fewer than N*H*W input pixels keep the direct form (default 128 k)."""
    global _WINOGRAD, _WINOGRAD_2D, _WINOGRAD_MIN_PIXELS, _WINOGRAD_MIN_PIXELS_2D
    _WINOGRAD = bool(enabled)
    _WINOGRAD_2D = enabled is not True and int(enabled) >= 2
    if min_pixels is not None:
        _WINOGRAD_MIN_PIXELS = int(min_pixels)
        _WINOGRAD_MIN_PIXELS_2D = min(_WINOGRAD_MIN_PIXELS_2D, int(min_pixels)) if int(min_pixels) < 128 * 1024 else 20 * 1024

# A producer asked for `dual_silu` hangs silu(y) on its result under this attribute; a consumer asked for
# `silu_in` uses the twin instead of re-evaluating SiLU inside its k-loop (9 taps x 2 half-waves times per
# element).  The twin is stored with y's version counter and storage address: an in-place update of y by a caller
# (y.add_(...), y.copy_(...), y.set_(...)) invalidates it -- the consumer then evaluates SiLU itself -- and views /
# clones never carry it.  (Tensors made under torch.inference_mode() have no version counter and cannot be checked;
# nothing in this package updates activations in place.)
_TWIN = "_mcq_silu_twin"


def silu_twin(t: torch.Tensor) -> Optional[torch.Tensor]:
    rec = getattr(t, _TWIN, None)
    if rec is None:
        return None
    twin, version, ptr = rec
    if version != tensor_version(t) or ptr != t.data_ptr():
        delattr(t, _TWIN)                       # y changed since the producer wrote silu(y): stale
        return None
    return twin


def set_silu_twin(t: torch.Tensor, twin: torch.Tensor) -> None:
    setattr(t, _TWIN, (twin, tensor_version(t), t.data_ptr()))



def tensor_version(t: torch.Tensor) -> int:
    """`t._version` for cache keys; tensors created under torch.inference_mode() (the reference CLI builds its model
    there, mcquic/cli.py:60) do not track one -- they cannot be updated by an optimizer either, so the storage
    pointer alone identifies their contents."""
    try:
        return t._version
    except RuntimeError:
        return -1


# Host-side cost matters where launches are ~10 us of GPU work (batch 1, the training step's ~4700 launches): the raw
# stream handle and a device guard that does nothing when the tensor already lives on the current device replace
# torch.cuda.current_stream() / torch.cuda.device(), which cost ~4 us and ~3 us per call.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """hipStream_t of the current device's current stream (call inside `_guard`)."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


class _guard:
    """`with _guard(t.device):` makes t's device current for the launch, like torch.cuda.device, but free when it already is."""
    __slots__ = ("idx", "prev")

    def __init__(self, device: torch.device):
        self.idx = device.index
        self.prev = -1

    def __enter__(self):
        if self.idx is not None and _cur_device is not None:
            cur = _cur_device()
            if cur != self.idx:
                torch.cuda.set_device(self.idx)
                self.prev = cur
        elif self.idx is not None:
            self.prev = torch.cuda.current_device()
            torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev >= 0:
            torch.cuda.set_device(self.prev)
        return False


def _dev(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"mcquic_amd: `{name}` must live on a HIP device (got {t.device}); "
                           "the HIP kernels have no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"mcquic_amd: `{name}` must be {dtype} (got {t.dtype})")
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    """Device address for a `void*` parameter / struct field (ctypes converts the int; None = NULL)."""
    return None if t is None else t.data_ptr()


class PackedConv:
    """A conv weight re-laid for the MFMA operand stream (+ its bias), see mcq_pack_conv_weight_f32."""

    __slots__ = ("wp", "bias", "cout", "cin", "ksize", "wino", "wino2d", "wino16", "lr_taps")
    # lr_taps: the packed 3x3 filter is zero outside its lower-right 2 x 2 taps (the input-gradient stream of a stride-2 layer): launches
    # say so (MCQ_CONV_TAPS_LR) and walk 4 of the 9 taps

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], copy_bias: bool = True, winograd: Optional[bool] = None):
        weight = _dev(weight.detach(), "weight")
        cout, cin, kh, kw = weight.shape
        if kh != kw or kh not in (1, 3):
            raise ValueError(f"unsupported kernel size {kh}x{kw}")
        lib = _lib.load()
        n = lib.mcq_packed_conv_weight_floats(cout, cin, kh)
        self.wp = torch.empty(n, dtype=torch.float32, device=weight.device)
        with _guard(weight.device):
            check(lib.mcq_pack_conv_weight_f32(_ptr(weight), cout, cin, kh, _ptr(self.wp), _stream()), "mcq_pack_conv_weight_f32")
        # (the copy decouples the pack from later in-place updates of the parameter; a caller that hands over a fresh tensor skips it)
        self.bias = None if bias is None else (_dev(bias.detach(), "bias").clone() if copy_bias else _dev(bias.detach(), "bias"))
        self.cout, self.cin, self.ksize = cout, cin, kh
        self.wino = self.wino2d = self.wino16 = None
        self.lr_taps = False
        level = (2 if _WINOGRAD_2D else 1) if _WINOGRAD else 0
        if winograd is not None:
            level = 1 if winograd is True else int(winograd)
        if level >= 1 and kh == 3 and cout % 64 == 0:
            self.wino = torch.empty(lib.mcq_packed_conv_winograd_floats(cout, cin), dtype=torch.float32, device=weight.device)
            with _guard(weight.device):
                check(lib.mcq_pack_conv_weight_winograd_f32(_ptr(weight), cout, cin, _ptr(self.wino), _stream()),
                      "mcq_pack_conv_weight_winograd_f32")
        if level >= 2 and kh == 3 and cout % 128 == 0 and cin % 8 == 0:
            self.wino2d = torch.empty(lib.mcq_packed_conv_winograd2d_floats(cout, cin), dtype=torch.float32, device=weight.device)
            with _guard(weight.device):
                check(lib.mcq_pack_conv_weight_winograd2d_f32(_ptr(weight), cout, cin, _ptr(self.wino2d), _stream()),
                      "mcq_pack_conv_weight_winograd2d_f32")
            if _W2D_KERNEL == 16 and cin % 16 == 0:
                self.wino16 = torch.empty(lib.mcq_packed_conv_winograd16_floats(cout, cin), dtype=torch.float32, device=weight.device)
                with _guard(weight.device):
                    check(lib.mcq_pack_conv_weight_winograd16_f32(_ptr(weight), cout, cin, _ptr(self.wino16), _stream()),
                          "mcq_pack_conv_weight_winograd16_f32")

    def repack_(self, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> bool:
        """Pack `weight` (+ `bias`) into THIS object's streams, in place: same addresses, no allocation -- what a captured
        hipGraph that reads them (parallel.GraphedTrainStep) and the allocator both prefer to a new object per weight version.
        False (nothing written) when the shapes differ or Winograd streams are involved: the caller then builds a new pack."""
        if self.wino is not None or self.wino2d is not None or self.wino16 is not None or _WINOGRAD:
            return False
        weight = _dev(weight.detach(), "weight")
        cout, cin, kh, kw = weight.shape
        lib = _lib.load()
        if kh != kw or (cout, cin, kh) != (self.cout, self.cin, self.ksize) or weight.device != self.wp.device or \
                self.wp.numel() != lib.mcq_packed_conv_weight_floats(cout, cin, kh) or (bias is None) != (self.bias is None):
            return False
        if bias is not None:
            bias = _dev(bias.detach(), "bias")
            if bias.data_ptr() != self.bias.data_ptr():       # (a pack that references the live parameter has nothing to copy)
                if bias.shape != self.bias.shape:
                    return False
                try:
                    self.bias.copy_(bias)
                except RuntimeError:                          # (a copy made under torch.inference_mode() cannot be updated outside it)
                    return False
        with _guard(weight.device):
            check(lib.mcq_pack_conv_weight_f32(_ptr(weight), cout, cin, kh, _ptr(self.wp), _stream()), "mcq_pack_conv_weight_f32")
        return True

    @classmethod
    def dgrad(cls, weight: torch.Tensor, stride: int, scale: float = 1.0, winograd: Optional[bool] = None) -> "PackedConv":
        """Operand stream of the layer's input-gradient convolution, packed straight from its OIHW weight in one launch
        (mcq_pack_conv_dgrad_weight_f32): stride 1 -> a [cin, cout, k, k] conv; stride 2 -> a [4 cin, cout, 3, 3] conv whose
        result goes through the PixelShuffle(2) store."""
        weight = _dev(weight.detach(), "weight")
        cout, cin, kh, kw = weight.shape
        lib = _lib.load()
        co_d, ci_d = ctypes.c_int32(0), ctypes.c_int32(0)
        if kh != kw or lib.mcq_dgrad_weight_shape(cout, cin, kh, stride, ctypes.byref(co_d), ctypes.byref(ci_d)) != 0:
            raise NotImplementedError(f"input gradient of a {kh}x{kw} stride-{stride} convolution is not on the path")
        self = cls.__new__(cls)
        self.wp = torch.empty(lib.mcq_packed_conv_weight_floats(co_d.value, ci_d.value, kh), dtype=torch.float32, device=weight.device)
        with _guard(weight.device):
            check(lib.mcq_pack_conv_dgrad_weight_f32(_ptr(weight), cout, cin, kh, stride, float(scale), _ptr(self.wp), _stream()),
                  "mcq_pack_conv_dgrad_weight_f32")
        self.bias = None
        self.cout, self.cin, self.ksize = co_d.value, ci_d.value, kh
        self.wino = self.wino2d = self.wino16 = None
        self.lr_taps = stride == 2 and kh == 3
        if (winograd if winograd is not None else _WINOGRAD) and kh == 3 and stride == 1 and scale == 1.0 and cin % 64 == 0:
            self.wino = torch.empty(lib.mcq_packed_conv_winograd_floats(cin, cout), dtype=torch.float32, device=weight.device)
            with _guard(weight.device):
                check(lib.mcq_pack_conv_dgrad_weight_winograd_f32(_ptr(weight), cout, cin, _ptr(self.wino), _stream()),
                      "mcq_pack_conv_dgrad_weight_winograd_f32")
        return self


# (round 6) the 1x1 layer BEHIND a 3x3 convolution inside that convolution's launch (MCQ_CONV_POST_*: GDN / IGDN after a strided /
# shuffle convolution, the AttentionBlock's gate after its side stack): A/B switch, 0 = always its own launch
# ("1" = all, "0" = none, or a comma list of gdn / igdn / gate)
# Round 6: 3x3 stride-1 launches that store no SiLU twin run the PIXEL-PAIR form of the unsplit 128 x 64 tile (tile bit 0x400, round 5:
# a third fewer activation loads, 64 / 128-bit epilogue accesses, the same bits -- tests/test_gpu_ops.py::test_conv_pixel_pair_tile_is_bit_equal)
# from 160 k output pixels up, i.e. where the launcher takes that tile for several rounds of the chip: -0.8 % on such launches, the
# 32-image step +0.17 % (258.76 -> 259.19 images/s over four alternating runs).  With a twin the form loses 1 %, on one-round launches
# the replay hides the gain, and a forced tile bit on small maps would take the 16 x 16 tile kernel away from them (measured: one
# image 5.67 -> 5.78 ms) -- hence the conditions.  MCQUIC_AMD_PAIR_RULE=0: the standing tile everywhere (A/B).
_PAIR_RULE = os.environ.get("MCQUIC_AMD_PAIR_RULE", "1") != "0"
_PAIR_MIN_PIXELS = int(os.environ.get("MCQUIC_AMD_PAIR_MIN_PIXELS", str(160 * 1024)))
_fp = os.environ.get("MCQUIC_AMD_FUSE_POST", "1")
_FUSE_POST = {"gdn", "igdn", "gate"} if _fp == "1" else set() if _fp == "0" else set(_fp.split(","))
# wave tiles (128 channels x 32 pixels) from which a launch is fused: the library's own rule is 2048 (two per SIMD); one image's large
# maps gain from 768 up (tools/bench_post.py --batch1), forced through tile 0x41
_POST_MIN_WAVES = int(os.environ.get("MCQUIC_AMD_POST_MIN_WAVES", "768"))


class PackedPost:
    """A [128, 128] 1x1 layer (+ bias) in the operand order of the MCQ_CONV_POST_* epilogues (mcq_pack_post1x1_weight_f32): its
    contraction runs over the producing wave's accumulator registers, in their order."""

    __slots__ = ("wp", "bias")

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor]):
        weight = _dev(weight.detach(), "weight")
        if tuple(weight.shape[:2]) != (128, 128) or weight.numel() != 128 * 128:
            raise ValueError("PackedPost: a [128, 128] 1x1 layer")
        lib = _lib.load()
        self.wp = torch.empty(lib.mcq_packed_post1x1_floats(), dtype=torch.float32, device=weight.device)
        with _guard(weight.device):
            check(lib.mcq_pack_post1x1_weight_f32(_ptr(weight), _ptr(self.wp), _stream()), "mcq_pack_post1x1_weight_f32")
        self.bias = None if bias is None else _dev(bias.detach(), "bias").clone()


def post_ok(x: torch.Tensor, w: "PackedConv", stride: int, kind: str, shuffle2: bool = False) -> int:
    """Does the 1x1 layer `kind` ("gdn" / "igdn" / "gate") behind the 3x3 convolution `w` of `x` run inside that convolution's launch?
    0 = no (the caller launches it on its own), else the `tile` to pass with the post_* option: 1 = the library's own choice (maps
    that fill the chip with unsplit 128-row tiles), 0x41 = forced (fewer tiles than that, still a measured gain)."""
    if kind not in _FUSE_POST or not x.is_cuda or w.ksize != 3 or _WINOGRAD or _band_rows(x, w, stride):
        return 0
    flag = {"gdn": CONV_POST_GDN, "igdn": CONV_POST_IGDN, "gate": CONV_POST_GATE}[kind] | (CONV_SHUFFLE2 if shuffle2 else 0)
    n, cin, h, wd = x.shape
    waves = int(_lib.load().mcq_conv2d_post_ok(n, cin, h, wd, w.cout, w.ksize, stride, flag))
    if waves < max(_POST_MIN_WAVES, 1):
        return 0
    return 1 if waves >= 2048 else 0x41


def pack_convs(weights: Sequence[torch.Tensor], biases: Optional[Sequence[Optional[torch.Tensor]]] = None, *, dgrad: bool = False,
               stride: int = 1, scale: float = 1.0, into: Optional[Sequence[Optional[PackedConv]]] = None,
               masks: Optional[Sequence[int]] = None) -> List[PackedConv]:
    """PackedConv (or PackedConv.dgrad) of several weights of ONE shape in ceil(n / 16) launches
    (mcq_pack_conv_weight_multi_f32).  The biases are referenced, not copied: this is the re-pack after an optimizer step,
    whose caller re-packs again whenever a parameter changes."""
    lib = _lib.load()
    ws = [_dev(w.detach(), "weight") for w in weights]
    if not ws:
        return []
    cout, cin, kh, kw = ws[0].shape
    if any(w.shape != ws[0].shape or w.device != ws[0].device for w in ws):
        raise ValueError("pack_convs: all weights must have one shape and one device")
    co, ci = cout, cin
    if dgrad:
        co_d, ci_d = ctypes.c_int32(0), ctypes.c_int32(0)
        if kh != kw or lib.mcq_dgrad_weight_shape(cout, cin, kh, stride, ctypes.byref(co_d), ctypes.byref(ci_d)) != 0:
            raise NotImplementedError(f"input gradient of a {kh}x{kw} stride-{stride} convolution is not on the path")
        co, ci = co_d.value, ci_d.value
    if kh != kw or kh not in (1, 3):
        raise ValueError(f"unsupported kernel size {kh}x{kw}")
    if co <= 16 and kh == 3:                                   # narrow layers carry a second copy: one by one
        if dgrad:
            return [PackedConv.dgrad(w, stride, scale) for w in ws]
        return [PackedConv(w, None if biases is None else biases[i], copy_bias=False) for i, w in enumerate(ws)]
    floats = lib.mcq_packed_conv_weight_floats(co, ci, kh)
    # `into`: the streams these weights were packed into before -- the re-pack after an optimizer step writes them in place
    # (same stream order as the launches that read them; no allocation, no new objects, addresses a captured graph can keep)
    reuse = into is not None and len(into) == len(ws) and all(
        pk is not None and pk.wp.numel() == floats and pk.wp.device == ws[0].device and pk.wino is None and pk.wino2d is None and pk.wino16 is None
        and (pk.cout, pk.cin, pk.ksize) == (co, ci, kh) for pk in into)
    slab = None if reuse else torch.empty((len(ws), floats), dtype=torch.float32, device=ws[0].device)
    dsts = [pk.wp for pk in into] if reuse else [slab[i] for i in range(len(ws))]
    cap = lib.mcq_pack_conv_weight_max_multi()
    with _guard(ws[0].device):
        for at in range(0, len(ws), cap):
            n = min(cap, len(ws) - at)
            src = (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws[at:at + n]])
            dst = (ctypes.c_void_p * n)(*[d.data_ptr() for d in dsts[at:at + n]])
            if reuse and masks is not None:
                # (only ever for an in-place re-pack: the copies the mask leaves out keep what an earlier full pack wrote)
                mk = (ctypes.c_uint8 * n)(*[int(m) & 15 for m in masks[at:at + n]])
                check(lib.mcq_pack_conv_weight_multi_masked_f32(src, dst, mk, n, cout, cin, kh, 1 if dgrad else 0, stride, float(scale), _stream()),
                      "mcq_pack_conv_weight_multi_masked_f32")
            else:
                check(lib.mcq_pack_conv_weight_multi_f32(src, dst, n, cout, cin, kh, 1 if dgrad else 0, stride, float(scale), _stream()),
                      "mcq_pack_conv_weight_multi_f32")
    if reuse:
        for i, pk in enumerate(into):
            pk.bias = None if dgrad or biases is None or biases[i] is None else _dev(biases[i].detach(), "bias")
        return list(into)
    out = []
    for i in range(len(ws)):
        pk = PackedConv.__new__(PackedConv)
        pk.wp = slab[i]
        pk.bias = None if dgrad or biases is None or biases[i] is None else _dev(biases[i].detach(), "bias")
        pk.cout, pk.cin, pk.ksize = co, ci, kh
        pk.wino = pk.wino2d = pk.wino16 = None
        pk.lr_taps = bool(dgrad) and stride == 2 and kh == 3
        out.append(pk)
    return out


def section_trace(on: bool) -> None:
    """Start (clearing earlier records) / stop recording which copy of its operand stream every conv launch reads
    (mcq_conv_section_trace); `sections_used(pack)` reads a stream's record back.  parallel.GraphedTrainStep uses it so that the
    re-pack inside its captured step refreshes only the copies the captured launches read."""
    _lib.load().mcq_conv_section_trace(1 if on else 0)


def sections_used(pack: Optional[PackedConv]) -> int:
    """Mask of the copies of `pack.wp` read by conv launches since section_trace(True) (0 = none seen = treat as all)."""
    return 0 if pack is None else int(_lib.load().mcq_conv_sections_used(pack.wp.data_ptr()))


def _conv_desc(x: torch.Tensor, w: PackedConv, stride: int = 1, *, silu_in: bool = False, square_in: bool = False,
               silu_out: bool = False, res: Optional[torch.Tensor] = None, res_scale: float = 1.0,
               gdn_mul: Optional[torch.Tensor] = None, igdn_mul: Optional[torch.Tensor] = None,
               gate_mul: Optional[torch.Tensor] = None, gate_id: Optional[torch.Tensor] = None,
               mul: Optional[torch.Tensor] = None, dsilu_mul: Optional[torch.Tensor] = None, shuffle2: bool = False,
               dual_silu: bool = False, tile: int = 0, winograd: Optional[bool] = None,
               post_gdn: Optional["PackedPost"] = None, post_igdn: Optional["PackedPost"] = None, post_gate: Optional["PackedPost"] = None):
    """(mcq_conv_desc, y, y_silu or None, tensors the descriptor points at) for one fused conv launch."""
    if silu_in:
        twin = silu_twin(x)
        if twin is not None:
            x, silu_in = twin, False
    x = _dev(x, "x")
    n, cin, h, wd = x.shape
    if cin != w.cin:
        raise ValueError(f"channel mismatch: x has {cin}, weight expects {w.cin}")
    pad = w.ksize // 2
    ho = (h + 2 * pad - w.ksize) // stride + 1
    wo = (wd + 2 * pad - w.ksize) // stride + 1
    flags = 0
    if silu_in:
        flags |= CONV_SILU_IN
    if square_in:
        flags |= CONV_SQUARE_IN
    if silu_out:
        flags |= CONV_SILU_OUT
    if shuffle2:
        flags |= CONV_SHUFFLE2
        y = torch.empty((n, w.cout // 4, 2 * ho, 2 * wo), dtype=torch.float32, device=x.device)
    else:
        y = torch.empty((n, w.cout, ho, wo), dtype=torch.float32, device=x.device)
    y2 = None
    if dual_silu:
        flags |= CONV_DUAL_SILU
        y2 = torch.empty_like(y)
    mul_in = mul
    mul = None
    if res is not None:
        flags |= CONV_RESIDUAL
        res = _dev(res, "res")
        if res.shape != y.shape:
            raise ValueError(f"residual shape {tuple(res.shape)} != output shape {tuple(y.shape)}")
    for flag, t in ((CONV_GDN, gdn_mul), (CONV_IGDN, igdn_mul), (CONV_GATE, gate_mul), (CONV_MUL, mul_in), (CONV_DSILU_MUL, dsilu_mul)):
        if t is not None:
            if not (flag == CONV_GATE and post_gate is not None):     # (post_gate: the gate closes the FUSED 1x1 layer, MCQ_CONV_POST_GATE)
                flags |= flag
            mul = _dev(t, "mul")
            if mul.shape != y.shape:
                raise ValueError(f"mul shape {tuple(mul.shape)} != output shape {tuple(y.shape)}")
    if gate_id is not None:
        gate_id = _dev(gate_id, "gate_id")
        if gate_id.shape != y.shape:
            raise ValueError("gate identity shape mismatch")
    post = None
    for flag, pk in ((CONV_POST_GDN, post_gdn), (CONV_POST_IGDN, post_igdn), (CONV_POST_GATE, post_gate)):
        if pk is not None:
            if post is not None:
                raise ValueError("one fused 1x1 layer per launch")
            flags |= flag
            post = pk
    if post is not None:
        winograd = False                             # (the fused 1x1 layer lives in the direct kernel's epilogue)
    wp = w.wp
    auto_2d = winograd is None and _WINOGRAD and _WINOGRAD_2D and w.wino2d is not None and n * h * wd >= _WINOGRAD_MIN_PIXELS_2D
    if winograd or auto_2d or (winograd is None and _WINOGRAD and n * h * wd >= _WINOGRAD_MIN_PIXELS):
        ok = w.wino is not None and stride == 1 and not (flags & (CONV_SILU_IN | CONV_SQUARE_IN))
        two_d = w.wino2d is not None and (_WINOGRAD_2D if winograd is None or winograd is True else int(winograd) >= 2)
        if winograd is not None and winograd is not True and int(winograd) >= 2 and w.wino2d is None:
            raise ValueError("winograd=2 needs a 3x3 layer with Cout % 128 == 0 and Cin % 8 == 0, packed with winograd=2")
        if ok and two_d and _W2D_KERNEL == 16 and w.wino16 is not None and (
                (flags & ~_W16_EPILOGUES) == 0 or flags == CONV_SHUFFLE2):
            flags |= CONV_WINOGRAD2D16               # the two-waves-per-SIMD instance (epilogues it does not carry stay below)
            wp = w.wino16
        elif ok and two_d:
            flags |= CONV_WINOGRAD2D
            wp = w.wino2d
        elif ok:
            flags |= CONV_WINOGRAD
            wp = w.wino
        elif winograd:
            raise ValueError("winograd=True needs a 3x3 stride-1 layer with Cout % 64 == 0, packed with winograd=True, and no input prologue")
    if _TAPS_LR and getattr(w, "lr_taps", False) and stride == 1 and wp is w.wp and not (flags & (CONV_SILU_IN | CONV_SQUARE_IN)):
        flags |= CONV_TAPS_LR
    if (_PAIR_RULE and tile == 0 and w.ksize == 3 and stride == 1 and not dual_silu and dsilu_mul is None and post is None
            and n * ho * wo >= _PAIR_MIN_PIXELS):
        tile = 0x400                                 # (the pixel-pair form wherever the launcher takes the unsplit 128 x 64 tile: see _PAIR_RULE)
    d = ConvDesc(_ptr(x), _ptr(wp), _ptr(w.bias), _ptr(y), _ptr(y2), _ptr(res), _ptr(mul), _ptr(gate_id),
                 n, cin, h, wd, w.cout, w.ksize, stride, flags, float(res_scale), tile,
                 None if post is None else _ptr(post.wp), None if post is None else _ptr(post.bias))
    return d, y, y2, (x, res, mul, gate_id, post)


# ---- images whose activations do not fit one launch --------------------------------------------------------------------------
# The kernels address one image's [C, H, W] slab through 32-bit buffer offsets whose top bit is the out-of-range marker
# (MCQ_ETOOLARGE at 2 GiB): a 6000 x 4000 photo's stem output is 3000 x 2000 x 128 x 4 B = 3.07 GB.  The reference has no limit
# but memory (mcquic/modules/compressor.py:67-117), so such layers run band by band: rows [o0, o1) of the output from the input
# rows under them (+ the 3x3 halo), each band a contiguous copy that fits.  Every output pixel still sees its own taps in the
# kernel's (channel pair, tap) order, so a band's rows are the unsplit launch's rows bit for bit wherever both take the same
# tile / split (tests/test_gpu_model.py::test_row_band_fallback_equals_the_single_launch at a lowered limit).
_SLAB_LIMIT = int(os.environ.get("MCQUIC_AMD_SLAB_LIMIT", str(1 << 31)))
_TENSOR_OPTS = ("res", "gdn_mul", "igdn_mul", "gate_mul", "gate_id", "mul", "dsilu_mul")


def set_slab_limit(nbytes: Optional[int]) -> int:
    """Per-image activation bytes above which a convolution runs in row bands (default 2 GiB = the kernels' own limit); tests
    lower it.  Returns the previous value; None restores the default."""
    global _SLAB_LIMIT
    prev = _SLAB_LIMIT
    _SLAB_LIMIT = (1 << 31) if nbytes is None else int(nbytes)
    return prev


def _slab_bytes(cin: int, h: int, wd: int, cout: int, ho: int, wo: int) -> int:
    """What conv_validate / conv_launch bound (csrc/conv_mfma.hip): the input slab plus the prefetch rings' over-read, the output
    slab with its last 128-row tile complete."""
    return max((cin + 32) * h * wd * 4, -(-cout // 128) * 128 * ho * wo * 4)


def _band_rows(x: torch.Tensor, w: PackedConv, stride: int) -> int:
    """0 when the layer fits one launch, else the output rows per band."""
    n, cin, h, wd = x.shape
    pad = w.ksize // 2
    ho, wo = (h + 2 * pad - w.ksize) // stride + 1, (wd + 2 * pad - w.ksize) // stride + 1
    if _slab_bytes(cin, h, wd, w.cout, ho, wo) < _SLAB_LIMIT:
        return 0
    per_row = max((cin + 32) * wd * 4 * stride, -(-w.cout // 128) * 128 * wo * 4)      # bytes one more output row costs
    rows = (_SLAB_LIMIT - 1) // per_row - 4                                            # (halo rows on both sides)
    if rows < 1:
        raise RuntimeError(f"mcquic_amd: one output row of a {cin}->{w.cout} layer on a {wd}-pixel-wide map exceeds the slab limit")
    return int(rows)


def _band_plan(h: int, ksize: int, stride: int, rows: int):
    """[(o0, o1, b0, b1, g0)] covering the output rows of a conv (kernel `ksize`, padding ksize // 2, `stride`) on an `h`-row map in
    bands of at most `rows` output rows: output rows [o0, o1) come from input rows [b0, b1) -- every tap of a kept row is inside
    the band or outside the image, where the conv pads -- and the band's local output row 0 is global output row g0 (b0 is a
    multiple of the stride, so the band's own conv computes rows g0, g0 + 1, ... and rows o0 - g0 .. o1 - g0 of it are kept)."""
    pad = ksize // 2
    ho = (h + 2 * pad - ksize) // stride + 1
    plan = []
    for o0 in range(0, ho, rows):
        o1 = min(ho, o0 + rows)
        b0 = max(0, stride * o0 - stride) if pad else stride * o0
        b1 = min(h, stride * (o1 - 1) + ksize - pad)
        plan.append((o0, o1, b0, b1, b0 // stride))
    return plan


def _conv2d_banded(x: torch.Tensor, w: PackedConv, stride: int, rows: int, fused: dict) -> torch.Tensor:
    if fused.get("silu_in"):
        twin = silu_twin(x)
        if twin is not None:
            x, fused = twin, dict(fused, silu_in=False)
    x = _dev(x, "x")
    n, cin, h, wd = x.shape
    k, pad = w.ksize, w.ksize // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (wd + 2 * pad - k) // stride + 1
    shuffle2, dual = bool(fused.get("shuffle2")), bool(fused.get("dual_silu"))
    up = 2 if shuffle2 else 1
    y = torch.empty((n, w.cout // 4, 2 * ho, 2 * wo) if shuffle2 else (n, w.cout, ho, wo), dtype=torch.float32, device=x.device)
    y2 = torch.empty_like(y) if dual else None
    sides = {kk: _dev(fused[kk], kk) for kk in _TENSOR_OPTS if fused.get(kk) is not None}
    if shuffle2 and sides:
        # (side inputs have the SHUFFLED output's 2 ho rows; the bands below are cut in pre-shuffle rows.  The kernels take no side
        #  tensor with the PixelShuffle store either -- conv_validate -- so nothing in the model asks for this)
        raise NotImplementedError("row bands: a PixelShuffle store together with output-shaped side tensors is not supported")
    plain = {kk: v for kk, v in fused.items() if kk not in _TENSOR_OPTS}
    for o0, o1, b0, b1, g0 in _band_plan(h, k, stride, rows):
        xb = x[:, :, b0:b1].contiguous()
        hl = (b1 - b0 + 2 * pad - k) // stride + 1
        local = {kk: v[:, :, g0:g0 + hl].contiguous() for kk, v in sides.items()}
        for kk, v in local.items():
            if v.shape[2] != hl:                                    # (the band computes rows past the map's end: pad the side input)
                local[kk] = torch.nn.functional.pad(v, (0, 0, 0, hl - v.shape[2]))
        yb = conv2d(xb, w, stride, **plain, **local)
        lo, hi = up * (o0 - g0), up * (o1 - g0)
        y[:, :, up * o0:up * o1] = yb[:, :, lo:hi]
        if dual:
            y2[:, :, up * o0:up * o1] = silu_twin(yb)[:, :, lo:hi]
    if dual:
        set_silu_twin(y, y2)
    return y


def conv2d(x: torch.Tensor, w: PackedConv, stride: int = 1, **fused) -> torch.Tensor:
    """y = epilogue(conv(prologue(x)) + bias); one kernel launch (mcq_conv2d_f32).  Options: silu_in, square_in, silu_out,
    res (+ res_scale), gdn_mul / igdn_mul / gate_mul (+ gate_id) / mul / dsilu_mul, shuffle2, dual_silu, tile.
    A layer whose per-image slab exceeds the kernels' 2 GiB addressing runs in row bands (same results, see above)."""
    rows = _band_rows(x, w, stride)
    if rows:
        return _conv2d_banded(x, w, stride, rows, fused)
    d, y, y2, keep = _conv_desc(x, w, stride, **fused)
    with _guard(y.device):
        check(_lib.load().mcq_conv2d_f32(ctypes.byref(d), _stream()), "mcq_conv2d_f32")
    if y2 is not None:
        set_silu_twin(y, y2)
    return y


def conv2d_gdn_bwd(x: torch.Tensor, w: PackedConv, dy: torch.Tensor, inverse: bool):
    """(dy f(s), dy x f'(s)) with s = beta + gamma @ x^2 recomputed by THIS launch (`w` = the layer's folded 1x1 operand stream):
    the element-wise part of the GDN / IGDN backward as the epilogue of the s-convolution (MCQ_CONV_GDN_BWD / _IGDN_BWD) instead of
    a launch of its own behind a stored s -- one launch and 2 x the tensor less HBM traffic per layer."""
    x, dy = _dev(x, "x"), _dev(dy, "dy")
    if w.ksize != 1 or x.shape != dy.shape or x.shape[1] != w.cin or w.cout != w.cin:
        raise ValueError("conv2d_gdn_bwd: a square 1x1 layer and dy of x's shape")
    n, c, h, wd = x.shape
    dxd, ds = torch.empty_like(x), torch.empty_like(x)
    d = ConvDesc(_ptr(x), _ptr(w.wp), _ptr(w.bias), _ptr(dxd), _ptr(ds), _ptr(dy), _ptr(x), None,
                 n, c, h, wd, w.cout, 1, 1, CONV_SQUARE_IN | (CONV_IGDN_BWD if inverse else CONV_GDN_BWD), 1.0, 0)
    with _guard(x.device):
        check(_lib.load().mcq_conv2d_f32(ctypes.byref(d), _stream()), "mcq_conv2d_f32")
    return dxd, ds


def conv2d_gate_bwd(bs, ws, as_, douts):
    """[(d a_i, d s_i)] of out_i = a_i * sigmoid(s_i) + x_i with s_i = conv1x1(b_i) RECOMPUTED by this launch (`ws[i]` = the gate's
    1x1 operand stream): d a = dout sigmoid(s), d s = dout a sigmoid(s) (1 - sigmoid(s)) leave from the epilogue
    (MCQ_CONV_GATE_BWD), so the forward stores no s and runs the gate as its 1x1 launch's epilogue (mcquic/nn/blocks.py:281-288).
    Several gates of one geometry (the paired heads' AttentionBlocks) share the launch."""
    built = []
    for b, w, a, dout in zip(bs, ws, as_, douts):
        b, a, dout = _dev(b, "b"), _dev(a, "a"), _dev(dout, "dout")
        if w.ksize != 1 or b.shape[1] != w.cin or a.shape != dout.shape or a.shape[1] != w.cout or a.shape[0] != b.shape[0] or a.shape[2:] != b.shape[2:]:
            raise ValueError("conv2d_gate_bwd: a 1x1 layer, `a` and `dout` of its output's shape")
        n, c, h, wd = b.shape
        da, ds = torch.empty_like(a), torch.empty_like(a)
        d = ConvDesc(_ptr(b), _ptr(w.wp), _ptr(w.bias), _ptr(da), _ptr(ds), _ptr(dout), _ptr(a), None,
                     n, c, h, wd, w.cout, 1, 1, CONV_GATE_BWD, 1.0, 0)
        built.append((d, da, ds, (b, a, dout)))
    lib = _lib.load()
    cap = lib.mcq_conv2d_max_multi() if _MULTI else 1
    with _guard(built[0][1].device):
        for lo in range(0, len(built), cap):
            part = built[lo:lo + cap]
            if len(part) == 1:
                check(lib.mcq_conv2d_f32(ctypes.byref(part[0][0]), _stream()), "mcq_conv2d_f32")
            else:
                check(lib.mcq_conv2d_multi_f32((ConvDesc * len(part))(*[p[0] for p in part]), len(part), _stream()), "mcq_conv2d_multi_f32")
    return [(da, ds) for _, da, ds, _ in built]


_MULTI = os.environ.get("MCQUIC_AMD_MULTI_CONV", "1") != "0"      # A/B switch: 0 = one launch per convolution
_TAPS_LR = os.environ.get("MCQUIC_AMD_TAPS_LR", "1") != "0"        # A/B switch: 0 = stride-2 input-gradient launches walk all nine taps (5 of them zeros)


def conv2d_multi(xs, ws, stride: int = 1, per_problem=None, **shared):
    """Independent convolutions of ONE geometry and flag set in one launch (mcq_conv2d_multi_f32): xs[i] through ws[i] with
    the options of `shared` plus per_problem[i] (tensor-valued options such as res= / dsilu_mul=).  Returns [y_i]."""
    n = len(xs)
    per_problem = per_problem or [{}] * n
    lib = _lib.load()
    if n == 1 or not _MULTI or _band_rows(xs[0], ws[0], stride):
        return [conv2d(x, w, stride, **shared, **pp) for x, w, pp in zip(xs, ws, per_problem)]
    cap = lib.mcq_conv2d_max_multi()
    out = []
    for lo in range(0, n, cap):
        built = [_conv_desc(x, w, stride, **shared, **pp) for x, w, pp in zip(xs[lo:lo + cap], ws[lo:lo + cap], per_problem[lo:lo + cap])]
        k = len(built)
        table = (ConvDesc * k)(*[b[0] for b in built])
        with _guard(built[0][1].device):
            check(lib.mcq_conv2d_multi_f32(table, k, _stream()), "mcq_conv2d_multi_f32")
        for _, y, y2, _keep in built:
            if y2 is not None:
                set_silu_twin(y, y2)
            out.append(y)
    return out


def nonneg_reparam(p: torch.Tensor, bound: float, pedestal: float) -> torch.Tensor:
    """max(p, bound)^2 - pedestal (mcquic/nn/base.py:81-84), folded once per weight version."""
    p = _dev(p.detach(), "p")
    out = torch.empty_like(p)
    with _guard(p.device):
        check(_lib.load().mcq_nonneg_reparam_f32(_ptr(p), float(bound), float(pedestal), _ptr(out), p.numel(), _stream()),
              "mcq_nonneg_reparam_f32")
    return out


def nonneg_reparam_multi_(ps, outs, bounds, pedestals) -> None:
    """outs[i] = max(ps[i], bounds[i])^2 - pedestals[i] for many parameters in ceil(n / 64) launches (mcq_nonneg_reparam_multi_f32);
    `outs` are written in place (their addresses are what packed operand streams and captured graphs hold)."""
    lib = _lib.load()
    ps = [_dev(p.detach(), "p") for p in ps]
    for o, p in zip(outs, ps):
        if o.shape != p.shape or o.dtype != torch.float32 or not o.is_contiguous() or o.device != p.device:
            raise ValueError("nonneg_reparam_multi_: every output must be a contiguous float32 tensor of its parameter's shape")
    cap = lib.mcq_nonneg_reparam_max_multi()
    with _guard(ps[0].device):
        for lo in range(0, len(ps), cap):
            k = min(cap, len(ps) - lo)
            src = (ctypes.c_void_p * k)(*[p.data_ptr() for p in ps[lo:lo + k]])
            dst = (ctypes.c_void_p * k)(*[o.data_ptr() for o in outs[lo:lo + k]])
            ns = (ctypes.c_int64 * k)(*[p.numel() for p in ps[lo:lo + k]])
            bd = (ctypes.c_float * k)(*[float(b) for b in bounds[lo:lo + k]])
            pd = (ctypes.c_float * k)(*[float(b) for b in pedestals[lo:lo + k]])
            check(lib.mcq_nonneg_reparam_multi_f32(src, dst, ns, bd, pd, k, _stream()), "mcq_nonneg_reparam_multi_f32")


def nonneg_reparam_bwd(p: torch.Tensor, dfolded: torch.Tensor, bound: float) -> torch.Tensor:
    """Gradient of max(p, bound)^2 - pedestal w.r.t. p under LowerBound's rule (mcq_nonneg_reparam_bwd_f32)."""
    p, dfolded = _dev(p.detach(), "p"), _dev(dfolded, "dfolded")
    if p.shape != dfolded.shape:
        raise ValueError("nonneg_reparam_bwd: shape mismatch")
    out = torch.empty_like(p)
    with _guard(p.device):
        check(_lib.load().mcq_nonneg_reparam_bwd_f32(_ptr(p), _ptr(dfolded), float(bound), _ptr(out), p.numel(), _stream()),
              "mcq_nonneg_reparam_bwd_f32")
    return out


def nonneg_reparam_bwd2(p0: torch.Tensor, d0: torch.Tensor, bound0: float, p1: torch.Tensor, d1: torch.Tensor, bound1: float):
    """nonneg_reparam_bwd for two parameters (a GDN layer's beta and gamma) in one launch (mcq_nonneg_reparam_bwd2_f32)."""
    p0, d0, p1, d1 = _dev(p0.detach(), "p0"), _dev(d0, "d0"), _dev(p1.detach(), "p1"), _dev(d1, "d1")
    if p0.numel() != d0.numel() or p1.numel() != d1.numel():
        raise ValueError("nonneg_reparam_bwd2: shape mismatch")
    o0, o1 = torch.empty_like(p0), torch.empty_like(p1)
    with _guard(p0.device):
        check(_lib.load().mcq_nonneg_reparam_bwd2_f32(_ptr(p0), _ptr(d0), float(bound0), _ptr(o0), p0.numel(), _ptr(p1), _ptr(d1), float(bound1),
                                                      _ptr(o1), p1.numel(), _stream()), "mcq_nonneg_reparam_bwd2_f32")
    return o0, o1


class PackedCodebook:
    """Codebook [m, k, d] in MFMA operand order + codeword norms (mcq_vq_pack_codebook_f32)."""

    __slots__ = ("packed", "codebook", "m", "k", "d")

    def __init__(self, codebook: torch.Tensor):
        cb = _dev(codebook.detach(), "codebook").clone()
        m, k, d = cb.shape
        lib = _lib.load()
        n = lib.mcq_packed_codebook_floats(m, k, d)
        self.packed = torch.empty(n, dtype=torch.float32, device=cb.device)
        with _guard(cb.device):
            check(lib.mcq_vq_pack_codebook_f32(_ptr(cb), m, k, d, _ptr(self.packed), _stream()), "mcq_vq_pack_codebook_f32")
        self.codebook, self.m, self.k, self.d = cb, m, k, d


def vq_assign(x: torch.Tensor, cb: PackedCodebook) -> torch.Tensor:
    """int64 codes [n, m, h, w] = argmin_k distance (mcq_vq_assign_f32)."""
    x = _dev(x, "x")
    n, c, h, w = x.shape
    if c != cb.m * cb.d:
        raise ValueError(f"latent has {c} channels, codebook expects {cb.m}*{cb.d}")
    codes = torch.empty((n, cb.m, h, w), dtype=torch.int64, device=x.device)
    lib = _lib.load()
    nbytes = lib.mcq_vq_assign_workspace_bytes(n, cb.m, cb.d, h, w, cb.k)       # > 0: a launch too small to fill the GPU ranges the codewords
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
    with _guard(x.device):
        check(lib.mcq_vq_assign_ws_f32(_ptr(x), _ptr(cb.packed), _ptr(codes), n, cb.m, cb.d, h, w, cb.k, _ptr(ws), _stream()),
              "mcq_vq_assign_ws_f32")
    return codes


def vq_gather(codes: torch.Tensor, cb: PackedCodebook, dual_silu: bool = False) -> torch.Tensor:
    """fp32 [n, m*d, h, w] = codebook[g, codes] (mcq_vq_gather_f32)."""
    codes = _dev(codes, "codes", torch.int64)
    n, m, h, w = codes.shape
    if m != cb.m:
        raise RuntimeError(f"codes carry m={m}, codebook has m={cb.m}")
    out = torch.empty((n, m * cb.d, h, w), dtype=torch.float32, device=codes.device)
    out2 = torch.empty_like(out) if dual_silu else None
    with _guard(codes.device):
        check(_lib.load().mcq_vq_gather_f32(_ptr(codes), _ptr(cb.codebook), _ptr(out), _ptr(out2), n, m, cb.d, h, w, cb.k,
                                            _stream()), "mcq_vq_gather_f32")
    if out2 is not None:
        set_silu_twin(out, out2)
    return out


def vq_logits(x: torch.Tensor, cb: PackedCodebook, temperature: torch.Tensor, bound: float) -> torch.Tensor:
    """Training logits [n, m, h, w, k] = (-dist / sqrt(k)) * max(temperature, bound) (mcq_vq_logits_f32)."""
    x = _dev(x, "x")
    n, c, h, w = x.shape
    if c != cb.m * cb.d:
        raise ValueError(f"latent has {c} channels, codebook expects {cb.m}*{cb.d}")
    t = _dev(temperature.detach().reshape(-1), "temperature")
    logits = torch.empty((n, cb.m, h, w, cb.k), dtype=torch.float32, device=x.device)
    with _guard(x.device):
        check(_lib.load().mcq_vq_logits_f32(_ptr(x), _ptr(cb.packed), _ptr(t), float(bound), _ptr(logits), n, cb.m, cb.d, h, w,
                                            cb.k, _stream()), "mcq_vq_logits_f32")
    return logits


# ---- the generator behind the soft assignment's draws (csrc/vq_train.hip: rng_uniform) ------------------------------------------
_rng_states = {}


def seed_rng(seed: int, device=None) -> None:
    """(Re)seed the in-kernel generator of `device` (default: the current one) and rewind its offset."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    st = _rng_states.get(dev.index)
    val = torch.tensor([int(seed) & 0x7fffffffffffffff, 0], dtype=torch.int64)
    if st is None:
        _rng_states[dev.index] = val.to(dev)
    else:
        st.copy_(val)


def get_rng_state(device=None) -> torch.Tensor:
    """{seed, offset} of the in-kernel generator of `device` as a CPU int64[2] tensor -- what a checkpoint stores beside
    torch.cuda.get_rng_state() (the soft assignment's draws are NOT torch's stream, so torch's state does not cover them)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    st = _rng_states.get(dev.index)
    if st is None:
        seed_rng(torch.initial_seed(), dev)
        st = _rng_states[dev.index]
    return st.detach().cpu().clone()


def set_rng_state(state: torch.Tensor, device=None) -> None:
    """Restore a state returned by get_rng_state (in place: a captured step keeps reading the same device tensor)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    state = torch.as_tensor(state, dtype=torch.int64).reshape(2)
    st = _rng_states.get(dev.index)
    if st is None:
        _rng_states[dev.index] = state.to(dev)
    else:
        st.copy_(state)


def rng_snapshot(device) -> torch.Tensor:
    """{seed, offset} for ONE pair of draws (a fresh int64[2] device tensor), after which the generator's offset moves on -- both
    on the device and in stream order, so the sequence is captured by a hipGraph and advances on every replay.  Seeded from
    torch's own seed at first use (`torch.manual_seed` before it, or `seed_rng` at any time)."""
    dev = torch.device(device)
    st = _rng_states.get(dev.index)
    if st is None:
        seed_rng(torch.initial_seed(), dev)
        st = _rng_states[dev.index]
    snap = st.clone()
    st[1:].add_(1)
    return snap


class VqStep:
    """What ONE mcq_vq_step_prologue_f32 launch prepared for a training forward of `levels` quantizations: the random drop's
    exponents (float32 [levels]), one generator snapshot per level (int64 [levels, 2]; None when the caller brings the draws)
    and the zeroed code-count buffer the sampling kernels add into (int64, level l at `offsets[l]`; None: no counting)."""

    __slots__ = ("exponents", "snaps", "counts", "offsets", "sizes")

    def level(self, l: int):
        """(drop exponent [1], generator snapshot [2] or None, this level's [m * k] counts or None)."""
        cnt = None if self.counts is None else self.counts[self.offsets[l]: self.offsets[l] + self.sizes[l]]
        return self.exponents[l: l + 1], (None if self.snaps is None else self.snaps[l]), cnt


def vq_step_prologue(freqs: Sequence[torch.Tensor], eps: float, want_rng: bool, counts: Optional[torch.Tensor] = None) -> VqStep:
    """The bookkeeping in front of a training forward's level cascade as ONE launch (mcq_vq_step_prologue_f32): per level the
    exponent of `_randomDrop` from its frequency EMA `freqs[l]` [m_l, k_l] (mcquic/modules/quantizer.py:194-198), a generator
    snapshot per level with the generator advanced past them (`want_rng`), and `counts` (int64, sum of m_l * k_l entries) zeroed."""
    fs = [_dev(f.detach(), "freq_ema") for f in freqs]
    levels = len(fs)
    lib = _lib.load()
    dev = fs[0].device
    ms = (ctypes.c_int32 * levels)(*[int(f.shape[0]) for f in fs])
    ks = (ctypes.c_int32 * levels)(*[int(f.shape[1]) for f in fs])
    st = VqStep()
    st.sizes = [int(f.shape[0]) * int(f.shape[1]) for f in fs]
    st.offsets = [sum(st.sizes[:l]) for l in range(levels)]
    st.exponents = torch.empty(levels, dtype=torch.float32, device=dev)
    st.snaps, state = None, None
    if want_rng:
        state = _rng_states.get(dev.index)
        if state is None:
            seed_rng(torch.initial_seed(), dev)
            state = _rng_states[dev.index]
        st.snaps = torch.empty((levels, 2), dtype=torch.int64, device=dev)
    st.counts = None
    if counts is not None:
        st.counts = _dev(counts, "counts", torch.int64)
        if st.counts.numel() != sum(st.sizes):
            raise ValueError("vq_step_prologue: the count buffer must hold sum(m_l * k_l) entries")
    # (the library's tables hold mcq_vq_max_levels() levels per launch -- 32; the reference's generator configs run 17 -- and
    #  anything beyond goes chunk by chunk: the generator's offset advances by each chunk's level count on the device, so the
    #  snapshots are the ones a single launch would hand out; the count buffer is zeroed whole by the first chunk)
    cap = int(lib.mcq_vq_max_levels())
    with _guard(dev):
        for l0 in range(0, levels, cap):
            nl = min(cap, levels - l0)
            ptrs = (ctypes.c_void_p * nl)(*[f.data_ptr() for f in fs[l0: l0 + nl]])
            ms_c = (ctypes.c_int32 * nl)(*ms[l0: l0 + nl])
            ks_c = (ctypes.c_int32 * nl)(*ks[l0: l0 + nl])
            first = l0 == 0 and st.counts is not None
            check(lib.mcq_vq_step_prologue_f32(ptrs, ms_c, ks_c, nl, float(eps), _ptr(st.exponents[l0: l0 + nl]), _ptr(state),
                                               None if st.snaps is None else _ptr(st.snaps[l0: l0 + nl]),
                                               _ptr(st.counts) if first else None, st.counts.numel() if first else 0, _stream()),
                  "mcq_vq_step_prologue_f32")
    return st


def vq_temperature_grad(dtrow: torch.Tensor, temperature: torch.Tensor, bound: float) -> torch.Tensor:
    """d temperature (the Parameter's shape) from the soft-max backward's per-row terms [n, m, h, w]: their sum over images and
    pixels per group, then LowerBound's gradient rule (mcquic/nn/base.py:24-29) -- one launch (mcq_vq_temperature_grad_f32)."""
    dtrow = _dev(dtrow, "dtrow")
    n, m, h, w = dtrow.shape
    t = _dev(temperature.detach().reshape(-1), "temperature")
    out = torch.empty(temperature.shape, dtype=torch.float32, device=dtrow.device)
    with _guard(dtrow.device):
        check(_lib.load().mcq_vq_temperature_grad_f32(_ptr(dtrow), _ptr(t), float(bound), _ptr(out), n, m, h * w, _stream()),
              "mcq_vq_temperature_grad_f32")
    return out


def freq_ema_update_(freqs: Sequence[torch.Tensor], counts: torch.Tensor, ema: float) -> None:
    """In place on every level's frequency EMA [m_l, k_l]: (1 - ema) * counts_l / sum_k counts_l + ema * freq_l
    (mcquic/modules/entropyCoder.py:37-43) for all levels in ONE launch; `counts` = the levels' histograms back to back (int64)."""
    fs = [_dev(f.detach(), "freq_ema") for f in freqs]
    for f, g in zip(fs, freqs):
        if f.data_ptr() != g.data_ptr():
            raise ValueError("freq_ema_update_: the frequency tensors must be contiguous (they are updated in place)")
    levels = len(fs)
    counts = _dev(counts, "counts", torch.int64)
    if counts.numel() != sum(int(f.numel()) for f in fs):
        raise ValueError("freq_ema_update_: `counts` must hold sum(m_l * k_l) entries")
    ms = (ctypes.c_int32 * levels)(*[int(f.shape[0]) for f in fs])
    ks = (ctypes.c_int32 * levels)(*[int(f.shape[1]) for f in fs])
    lib = _lib.load()
    cap = int(lib.mcq_vq_max_levels())
    with _guard(counts.device):
        off = 0
        for l0 in range(0, levels, cap):                     # (a launch's tables hold `cap` levels; see vq_step_prologue)
            nl = min(cap, levels - l0)
            ptrs = (ctypes.c_void_p * nl)(*[f.data_ptr() for f in fs[l0: l0 + nl]])
            ms_c = (ctypes.c_int32 * nl)(*ms[l0: l0 + nl])
            ks_c = (ctypes.c_int32 * nl)(*ks[l0: l0 + nl])
            size = sum(int(f.numel()) for f in fs[l0: l0 + nl])
            check(lib.mcq_freq_ema_update_f32(ptrs, ms_c, ks_c, nl, _ptr(counts[off: off + size]), float(ema), _stream()),
                  "mcq_freq_ema_update_f32")
            off += size


def hash_uniform(rng: torch.Tensor, stream_id: int, shape) -> torch.Tensor:
    """The generator's draws for a tensor of `shape` (stream 0 = random drop, 1 = Gumbel noise) under the snapshot `rng`:
    exactly what the kernels use in place of a missing u_drop / u_gumbel (mcq_hash_uniform_f32)."""
    rng = _dev(rng, "rng", torch.int64)
    out = torch.empty(tuple(shape), dtype=torch.float32, device=rng.device)
    with _guard(rng.device):
        check(_lib.load().mcq_hash_uniform_f32(_ptr(rng), int(stream_id), _ptr(out), out.numel(), _stream()), "mcq_hash_uniform_f32")
    return out


def vq_gumbel_sample(logits: torch.Tensor, u_drop: Optional[torch.Tensor], u_gumbel: Optional[torch.Tensor], freq_ema: torch.Tensor,
                     drop_exponent: torch.Tensor, rng: Optional[torch.Tensor] = None, counts: Optional[torch.Tensor] = None):
    """In place on `logits`: random drop; returns (codes, sample_index, sample_hot), each [n, m, h, w].  A draw that is None is
    made inside the kernel from the generator snapshot `rng` (rng_snapshot).  `counts` (int64 [m * k], zeroed by the caller once
    per step): every code made here is counted into it -- the level's histogram for the frequency EMA."""
    logits = _dev(logits, "logits")
    n, m, h, w, k = logits.shape
    if (u_drop is None or u_gumbel is None) and rng is None:
        raise ValueError("vq_gumbel_sample: give both uniform draws or a generator snapshot")
    u_drop = None if u_drop is None else _dev(u_drop, "u_drop")
    u_gumbel = None if u_gumbel is None else _dev(u_gumbel, "u_gumbel")
    rng = None if rng is None else _dev(rng, "rng", torch.int64)
    if any(u is not None and u.shape != logits.shape for u in (u_drop, u_gumbel)):
        raise ValueError("uniform draws must have the logits' shape")
    freq = _dev(freq_ema.detach(), "freq_ema")
    expo = _dev(drop_exponent.detach().reshape(1), "drop_exponent")
    codes = torch.empty((n, m, h, w), dtype=torch.int64, device=logits.device)
    index = torch.empty_like(codes)
    hot = torch.empty((n, m, h, w), dtype=torch.float32, device=logits.device)
    if counts is not None:
        counts = _dev(counts, "counts", torch.int64)
        if counts.numel() != m * k:
            raise ValueError("vq_gumbel_sample: `counts` must hold m * k entries")
    with _guard(logits.device):
        check(_lib.load().mcq_vq_gumbel_sample_f32(_ptr(logits), _ptr(u_drop), _ptr(u_gumbel), _ptr(rng), _ptr(freq), _ptr(expo), _ptr(codes),
                                                   _ptr(index), _ptr(hot), _ptr(counts), n, m, h, w, k, _stream()), "mcq_vq_gumbel_sample_f32")
    return codes, index, hot


def vq_dequant_soft(index: torch.Tensor, hot: torch.Tensor, cb: PackedCodebook, dual_silu: bool = False) -> torch.Tensor:
    """sample @ codebook for the one-hot-valued straight-through sample -> [n, m*d, h, w] (`dual_silu`: + its SiLU twin)."""
    index = _dev(index, "sample_index", torch.int64)
    hot = _dev(hot, "sample_hot")
    n, m, h, w = index.shape
    out = torch.empty((n, m * cb.d, h, w), dtype=torch.float32, device=index.device)
    out2 = torch.empty_like(out) if dual_silu else None
    with _guard(index.device):
        check(_lib.load().mcq_vq_dequant_soft_f32(_ptr(index), _ptr(hot), _ptr(cb.codebook), _ptr(out), _ptr(out2), n, m, cb.d, h, w, cb.k,
                                                  _stream()), "mcq_vq_dequant_soft_f32")
    if out2 is not None:
        set_silu_twin(out, out2)
    return out


def add(a: torch.Tensor, b: torch.Tensor, dual_silu: bool = False) -> torch.Tensor:
    a, b = _dev(a, "a"), _dev(b, "b")
    if a.shape != b.shape:
        raise ValueError("add: shape mismatch")
    out = torch.empty_like(a)
    out2 = torch.empty_like(a) if dual_silu else None
    with _guard(a.device):
        check(_lib.load().mcq_add_f32(_ptr(a), _ptr(b), _ptr(out), _ptr(out2), a.numel(), _stream()), "mcq_add_f32")
    if out2 is not None:
        set_silu_twin(out, out2)
    return out


def add3(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
    """(a + b) + c in one launch (mcq_add3_f32)."""
    a, b, c = _dev(a, "a"), _dev(b, "b"), _dev(c, "c")
    if a.shape != b.shape or a.shape != c.shape:
        raise ValueError("add3: shape mismatch")
    out = torch.empty_like(a)
    with _guard(a.device):
        check(_lib.load().mcq_add3_f32(_ptr(a), _ptr(b), _ptr(c), _ptr(out), a.numel(), _stream()), "mcq_add3_f32")
    return out


def mse(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """mean((a - b)^2) as a 0-dim tensor (mcq_mse_f32: two launches, fixed summation order, no memset -- see csrc/train_ops.hip)."""
    a, b = _dev(a, "a"), _dev(b, "b")
    if a.shape != b.shape or a.numel() == 0:
        raise ValueError("mse: shape mismatch or empty input")
    lib = _lib.load()
    out = torch.empty((), dtype=torch.float32, device=a.device)
    ws = torch.empty(lib.mcq_mse_workspace_bytes(a.numel()) // 8, dtype=torch.float64, device=a.device)
    with _guard(a.device):
        check(lib.mcq_mse_f32(_ptr(a), _ptr(b), _ptr(out), _ptr(ws), a.numel(), _stream()), "mcq_mse_f32")
    return out


def mse_bwd(a: torch.Tensor, b: torch.Tensor, dloss: torch.Tensor, want_db: bool = False):
    """(da, db or None) of `mse`: da = 2 (a - b) / n * dloss (mcq_mse_bwd_f32); `dloss` stays on the device."""
    a, b, dloss = _dev(a, "a"), _dev(b, "b"), _dev(dloss, "dloss")
    da = torch.empty_like(a)
    db = torch.empty_like(a) if want_db else None
    with _guard(a.device):
        check(_lib.load().mcq_mse_bwd_f32(_ptr(a), _ptr(b), _ptr(dloss), _ptr(da), _ptr(db) if want_db else None, a.numel(), _stream()),
              "mcq_mse_bwd_f32")
    return da, db


def sumsq(x: torch.Tensor) -> torch.Tensor:
    """sum(x^2) as a 0-dim float32 tensor (mcq_sumsq_f32: double partials, fixed order, no memset)."""
    x = _dev(x, "x")
    lib = _lib.load()
    out = torch.empty((), dtype=torch.float32, device=x.device)
    ws = torch.empty(lib.mcq_mse_workspace_bytes(x.numel()) // 8, dtype=torch.float64, device=x.device)
    with _guard(x.device):
        check(lib.mcq_sumsq_f32(_ptr(x), _ptr(out), _ptr(ws), x.numel(), _stream()), "mcq_sumsq_f32")
    return out


def clip_by_norm_(x: torch.Tensor, max_norm: float, eps: float = 1e-6) -> torch.Tensor:
    """In place: x *= max_norm / (||x|| + eps) where that is < 1 (torch.nn.utils.clip_grad_norm_ over one flat buffer; the reference's
    trainer.py:280).  Returns ||x|| BEFORE clipping as a 0-dim device tensor; no host read, so it can be captured."""
    x = _dev(x, "x")
    if not x.is_contiguous():
        raise ValueError("clip_by_norm_: needs a contiguous buffer (it is scaled in place)")
    sq = sumsq(x)
    norm = torch.empty((), dtype=torch.float32, device=x.device)
    with _guard(x.device):
        check(_lib.load().mcq_clip_by_norm_f32(_ptr(x), _ptr(sq), float(max_norm), float(eps), _ptr(norm), x.numel(), _stream()),
              "mcq_clip_by_norm_f32")
    return norm


def group_norm(x: torch.Tensor, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor], groups: int, eps: float = 1e-5,
               dual_silu: bool = False, want_stats: bool = False):
    """nn.GroupNorm(groups, C) on [n, C, h, w] (mcq_group_norm_f32; `denseNorm=True`, mcquic/nn/blocks.py:179-200).
    With `want_stats` returns (y, mean, rstd), the per-(image, group) statistics the backward pass needs."""
    x = _dev(x, "x")
    n, c, h, w = x.shape
    if c % groups != 0:
        raise ValueError(f"GroupNorm: {c} channels do not divide into {groups} groups")
    weight = None if weight is None else _dev(weight.detach(), "weight")
    bias = None if bias is None else _dev(bias.detach(), "bias")
    y = torch.empty_like(x)
    y2 = torch.empty_like(x) if dual_silu else None
    mean = torch.empty(n * groups, dtype=torch.float32, device=x.device) if want_stats else None
    rstd = torch.empty_like(mean) if want_stats else None
    lib = _lib.load()
    nws = lib.mcq_group_norm_workspace_floats(n, c, h * w, groups)       # > 0: large runs, many workgroups per (image, group)
    ws = torch.empty(nws, dtype=torch.float32, device=x.device) if nws else None
    with _guard(x.device):
        check(lib.mcq_group_norm_f32(_ptr(x), _ptr(weight), _ptr(bias), _ptr(y), _ptr(y2), _ptr(mean), _ptr(rstd), _ptr(ws), n, c, h * w,
                                     groups, float(eps), _stream()), "mcq_group_norm_f32")
    if y2 is not None:
        set_silu_twin(y, y2)
    return (y, mean, rstd) if want_stats else y


def group_norm_bwd(x: torch.Tensor, dy: torch.Tensor, weight: Optional[torch.Tensor], mean: torch.Tensor, rstd: torch.Tensor,
                   groups: int, want_params: bool = True):
    """(dx, dweight, dbias) of group_norm (mcq_group_norm_bwd_f32)."""
    x, dy = _dev(x, "x"), _dev(dy, "dy")
    n, c, h, w = x.shape
    weight = None if weight is None else _dev(weight.detach(), "weight")
    lib = _lib.load()
    ws = torch.empty(lib.mcq_group_norm_bwd_workspace_floats(n, c, h * w, groups), dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    dw = torch.empty(c, dtype=torch.float32, device=x.device) if want_params else None
    db = torch.empty(c, dtype=torch.float32, device=x.device) if want_params else None
    with _guard(x.device):
        check(lib.mcq_group_norm_bwd_f32(_ptr(x), _ptr(dy), _ptr(weight), _ptr(mean), _ptr(rstd), _ptr(dx), _ptr(dw), _ptr(db), _ptr(ws),
                                         n, c, h * w, groups, _stream()), "mcq_group_norm_bwd_f32")
    return dx, dw, db


def detransform(x: torch.Tensor) -> torch.Tensor:
    """[-1, 1] fp32 -> uint8 (mcquic/utils/vision.py:143-146)."""
    x = _dev(x, "x")
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    with _guard(x.device):
        check(_lib.load().mcq_detransform_u8(_ptr(x), _ptr(out), x.numel(), _stream()), "mcq_detransform_u8")
    return out


# ---- validation metrics (mcquic/validate/handlers.py) ----------------------------------------------------------
def _u8_pair(x: torch.Tensor, y: torch.Tensor):
    x, y = _dev(x, "x", torch.uint8), _dev(y, "y", torch.uint8)
    if x.shape != y.shape or x.dim() != 4:
        raise ValueError(f"expected two uint8 [N, C, H, W] batches of one shape, got {tuple(x.shape)} and {tuple(y.shape)}")
    return x, y


def ms_ssim(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """MS-SSIM value in [0, 1] per image of two uint8 batches (mcquic/validate/metrics.py:142-193 as called by
    handlers.py:14-27).  Sides must exceed 160 pixels (metrics.py:163-166)."""
    x, y = _u8_pair(x, y)
    n, c, h, w = x.shape
    lib = _lib.load()
    nbytes = lib.mcq_ms_ssim_workspace_bytes(n, c, h, w)
    if nbytes == 0:
        raise ValueError(f"MS-SSIM needs image sides larger than 160 pixels, got {h}x{w}")
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=x.device)
    out = torch.empty(n, dtype=torch.float32, device=x.device)
    with _guard(x.device):
        check(lib.mcq_ms_ssim_u8(_ptr(x), _ptr(y), _ptr(out), _ptr(ws), n, c, h, w, _stream()), "mcq_ms_ssim_u8")
    return out


def sqdiff_sum(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """Exact per-image sum of squared differences of two uint8 batches, int64 [N]."""
    x, y = _u8_pair(x, y)
    n = x.shape[0]
    out = torch.empty(n, dtype=torch.int64, device=x.device)
    with _guard(x.device):
        check(_lib.load().mcq_sqdiff_sum_u8(_ptr(x), _ptr(y), _ptr(out), x[0].numel(), n, _stream()), "mcq_sqdiff_sum_u8")
    return out


# ---- backward-pass kernels (training step) ---------------------------------------------------------------------
def nchw_to_nhwc(x: torch.Tensor, square: bool = False) -> torch.Tensor:
    x = _dev(x, "x")
    n, c, h, w = x.shape
    out = torch.empty((n, h, w, c), dtype=torch.float32, device=x.device)
    with _guard(x.device):
        check(_lib.load().mcq_nchw_to_nhwc_f32(_ptr(x), _ptr(out), n, c, h * w, int(square), _stream()), "mcq_nchw_to_nhwc_f32")
    return out


_WGRAD_ROWS = os.environ.get("MCQUIC_AMD_WGRAD_ROWS", "1") != "0"      # A/B switch: 0 = always the NHWC weight-gradient kernel

# ---- the weight gradients' reduce passes, batched over a backward pass (mcq_wgrad_defer / mcq_wgrad_flush) -------------------------------
_WGRAD_DEFER = os.environ.get("MCQUIC_AMD_WGRAD_DEFER", "1") != "0"    # A/B switch: 0 = every weight-gradient launch reduces right away
_defer = {"on": False, "keep": [], "queue": [], "small": None, "main": None, "side": None, "forked": False, "split": None}
# Round 6, measured and left OFF (MCQUIC_AMD_WGRAD_SIDE=1 turns it on): inside a deferral nothing reads a weight gradient before the
# flush, so the weight-gradient LAUNCHES need not stand in the chain of input gradients either.  They are queued (outputs allocated at
# once, operands held) and issued in batches on ONE side stream: when the backward pass moves from chip-filling maps to the few-pixel
# ones (or back) everything queued so far goes to the side stream behind one fork edge; one join at the flush.  Eager step on one
# MI355X 22.09 -> 21.56 ms (the host is the bound there and the side stream hides some of it); a captured step LOSES: as a second
# branch inside one hipGraph 36.8 ms against 21.2 (~85 edges: 35.5), as hipGraphs of their own beside the main stream's
# (tools/probes/split_capture.py) 22.6 -- the row walks hold every wave slot and each launch of the latency-bound chain over the
# 16x16 ... 4x4 maps waits for one.  Inside a plain capture the queue is therefore issued in line.  docs/experiments.md section 11.8.
_WGRAD_SIDE = os.environ.get("MCQUIC_AMD_WGRAD_SIDE", "0") == "1"
_WGRAD_SMALL_PIXELS = 4096                                          # images x output pixels up to which a launch counts as few-pixel
_wgrad_streams = {}


def _queue_wgrad(small: bool, operands, launch) -> None:
    """One weight-gradient call of a deferred pass: `launch()` runs later, on the side stream or at the flush."""
    dev = operands[0].device
    main = torch.cuda.current_stream(dev)
    if _defer["main"] is None:
        key = (dev.index, main.cuda_stream)
        side = _wgrad_streams.get(key)
        if side is None:
            side = _wgrad_streams[key] = torch.cuda.Stream(device=dev)
        _defer["main"], _defer["side"], _defer["forked"] = main, side, False
    elif main != _defer["main"]:                                    # (a launch from another stream than the pass began on: in line)
        launch()
        return
    if _defer["small"] is not None and small != _defer["small"] and _defer["queue"]:
        _issue_wgrads(side=True)
    _defer["small"] = small
    _defer["queue"].append(launch)
    _defer["keep"].extend(operands)                                 # (the side stream reads them: not the allocator's to hand out before the join)


def _issue_wgrads(side: bool) -> None:
    queue, _defer["queue"] = _defer["queue"], []
    if not queue:
        return
    main = _defer["main"]
    if side and _defer["split"] is not None:
        # a capture that knows about the fork (tools/probes/split_capture.py): the batch becomes a hipGraph of its own, replayed on
        # the side stream beside the main stream's next graph
        _defer["split"].fork(queue)
        _defer["forked"] = True
    elif side and not torch.cuda.is_current_stream_capturing():
        _defer["side"].wait_stream(main)
        _defer["forked"] = True
        with torch.cuda.stream(_defer["side"]):
            for launch in queue:
                launch()
    else:                                                           # (in line: the flush's rest, or a plain capture)
        with torch.cuda.stream(main):
            for launch in queue:
                launch()


def _join_wgrad_side() -> None:
    main, side, forked = _defer["main"], _defer["side"], _defer["forked"]
    _defer["main"] = _defer["side"] = _defer["small"] = None
    _defer["forked"] = False
    del _defer["queue"][:]
    if forked:
        if _defer["split"] is not None:
            _defer["split"].join()
        else:
            main.wait_stream(side)


class wgrad_deferral:
    """`with ops.wgrad_deferral():` around a backward pass this library owns end to end (autograd.backward, parallel.GraphedTrainStep):
    inside, weight-gradient launches leave their partial tiles in their workspaces and record the reduce pass; on exit ALL recorded
    passes run in a few launches (a captured training step: 56 reduce launches -> 3).  Nothing may read a weight gradient before the
    exit -- torch DDP's bucket hooks do, which is why a plain `loss.backward()` never defers.  Workspaces are kept alive until then."""

    def __enter__(self):
        self.active = _WGRAD_DEFER and not _defer["on"]
        if self.active:
            _defer["on"] = True
            _lib.load().mcq_wgrad_defer(1)
        return self

    def __exit__(self, exc_type, exc, tb):
        if not self.active:
            return False
        lib = _lib.load()
        try:
            if exc_type is None:
                _issue_wgrads(side=False)                           # (what is still queued: in line, beside whatever the side stream still holds)
        finally:
            lib.mcq_wgrad_defer(0)
            _defer["on"] = False
        try:
            _join_wgrad_side()                                      # (the reduce passes read what the side stream's launches left)
            if exc_type is None and lib.mcq_wgrad_pending():
                dev = _defer["keep"][0].device if _defer["keep"] else torch.device("cuda", torch.cuda.current_device())
                with _guard(dev):
                    check(lib.mcq_wgrad_flush(0, _stream()), "mcq_wgrad_flush")
            else:
                lib.mcq_wgrad_flush(1, None)
        finally:
            del _defer["keep"][:]
        return False


class wgrad_now:
    """Inside a `wgrad_deferral`: the weight gradients of this block are reduced right away (their values are read next)."""

    def __enter__(self):
        self.was = _defer["on"]
        if self.was:
            _lib.load().mcq_wgrad_defer(0)
            _defer["on"] = False
        return self

    def __exit__(self, *exc):
        if self.was:
            _lib.load().mcq_wgrad_defer(1)
            _defer["on"] = True
        return False


def _keep(*tensors) -> None:
    """Workspace AND outputs of a deferred weight-gradient launch stay alive until the flush has written them: a dW the autograd
    engine drops (a frozen weight: requires_grad False) would otherwise be handed to another tensor and the flush would write
    over that one.  Outputs are held through their STORAGE, never the tensor itself (nor a view of it: a view made without grad
    mode still references its base tensor): AccumulateGrad only takes a gradient as it is while nothing else references that very
    tensor -- with a second reference it clones it, during the pass, i.e. before the flush has written the values."""
    if _defer["on"]:
        _defer["keep"].append(tensors[0])
        _defer["keep"].extend(t.untyped_storage() for t in tensors[1:] if t is not None)


class _Later:
    """An output of a queued launch, held WITHOUT a reference to the tensor the caller got (AccumulateGrad takes a gradient as it is
    only while nothing else references that very tensor; with a second reference it clones it -- before the launch has run)."""

    def __init__(self, t: Optional[torch.Tensor]):
        self.storage, self.shape = (t.untyped_storage(), tuple(t.shape)) if t is not None else (None, None)

    def tensor(self) -> Optional[torch.Tensor]:
        if self.storage is None:
            return None
        return torch.empty(0, dtype=torch.float32, device=self.storage.device).set_(self.storage, 0, self.shape)


def _wgrad_outputs(x, dy, ksize: int, want_bias: bool):
    dw = torch.empty((dy.shape[1], x.shape[1], ksize, ksize), dtype=torch.float32, device=x.device)
    return dw, (torch.empty((dy.shape[1],), dtype=torch.float32, device=x.device) if want_bias else None)


def conv2d_wgrad_group(xs, dys, want_bias: bool = True):
    if not (_defer["on"] and _WGRAD_SIDE):
        return _conv2d_wgrad_group(xs, dys, want_bias)
    xs, dys = [_dev(t, "x") for t in xs], [_dev(t, "dy") for t in dys]
    outs = [_wgrad_outputs(x, dy, 3, want_bias) for x, dy in zip(xs, dys)]
    small = xs[0].shape[0] * dys[0].shape[2] * dys[0].shape[3] <= _WGRAD_SMALL_PIXELS
    later = [(_Later(dw), _Later(db)) for dw, db in outs]
    _queue_wgrad(small, xs + dys, lambda: _conv2d_wgrad_group(xs, dys, want_bias, [(a.tensor(), b.tensor()) for a, b in later]))
    return outs


def conv2d_wgrad(x: torch.Tensor, dy: torch.Tensor, ksize: int, stride: int, square_x: bool = False, want_bias: bool = False):
    if not (_defer["on"] and _WGRAD_SIDE):
        return _conv2d_wgrad(x, dy, ksize, stride, square_x, want_bias)
    x, dy = _dev(x, "x"), _dev(dy, "dy")
    out = _wgrad_outputs(x, dy, ksize, want_bias)
    small = x.shape[0] * dy.shape[2] * dy.shape[3] <= _WGRAD_SMALL_PIXELS
    later = (_Later(out[0]), _Later(out[1]))
    _queue_wgrad(small, [x, dy], lambda: _conv2d_wgrad(x, dy, ksize, stride, square_x, want_bias, (later[0].tensor(), later[1].tensor())))
    return out if want_bias else out[0]


def _conv2d_wgrad_group(xs, dys, want_bias: bool = True, outs=None):
    """Weight (and bias) gradients of several 3x3 stride-1 convolutions of ONE shape in one launch pair
    (mcq_conv2d_wgrad_nchw_group_f32).  Returns [(dW, db or None), ...]; shapes the row-walk kernel does not take, or a
    single pair, go through conv2d_wgrad one by one."""
    lib = _lib.load()
    n, cin, h, w = xs[0].shape
    cout = dys[0].shape[1]
    same = all(x.shape == xs[0].shape for x in xs) and all(d.shape == dys[0].shape for d in dys)
    nws = lib.mcq_conv2d_wgrad_nchw_workspace_floats(n, cin, h, w, cout) if (_WGRAD_ROWS and same and tuple(dys[0].shape[2:]) == (h, w)) else 0
    if not nws or len(xs) == 1:
        out = []
        for i, (x, dy) in enumerate(zip(xs, dys)):
            r = _conv2d_wgrad(x, dy, 3, 1, want_bias=want_bias, out=outs[i] if outs else None)
            out.append(r if want_bias else (r, None))
        return out
    out = []
    cap = lib.mcq_conv2d_wgrad_nchw_max_group()
    # (round 3: capping a group by its operand footprint -- twelve 8 x 128 x 64 x 64 problems as three launches of four --
    #  changed nothing, 984 vs 1038 us: the 82 us per convolution are the kernel's own, not cache eviction between problems)
    for lo in range(0, len(xs), cap):
        gx = [_dev(t, "x") for t in xs[lo:lo + cap]]
        gd = [_dev(t, "dy") for t in dys[lo:lo + cap]]
        k = len(gx)
        ws = torch.empty(nws * k, dtype=torch.float32, device=gx[0].device)
        if outs:
            dws, dbs = [o[0] for o in outs[lo:lo + k]], ([o[1] for o in outs[lo:lo + k]] if want_bias else None)
        else:
            dws = [torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=gx[0].device) for _ in range(k)]
            dbs = [torch.empty((cout,), dtype=torch.float32, device=gx[0].device) for _ in range(k)] if want_bias else None
        table = ctypes.c_void_p * k
        with _guard(gx[0].device):
            check(lib.mcq_conv2d_wgrad_nchw_group_f32(table(*[t.data_ptr() for t in gx]), table(*[t.data_ptr() for t in gd]),
                                                      table(*[t.data_ptr() for t in dws]),
                                                      table(*[t.data_ptr() for t in dbs]) if want_bias else None, k, _ptr(ws),
                                                      n, cin, h, w, cout, _stream()), "mcq_conv2d_wgrad_nchw_group_f32")
        _keep(ws, *dws, *(dbs or []))
        out.extend(zip(dws, dbs if want_bias else [None] * k))
    return out


def _conv2d_wgrad(x: torch.Tensor, dy: torch.Tensor, ksize: int, stride: int, square_x: bool = False, want_bias: bool = False,
                  out=None):
    """dW [Cout, Cin, k, k] of y = conv(x, W) + b from NCHW x and dy (channel-major copies are made here, one launch for
    the pair); with `want_bias` returns (dW, db) where db[co] = sum of dy over images and pixels, from the same kernel."""
    x, dy = _dev(x, "x"), _dev(dy, "dy")
    n, cin, h, w = x.shape
    cout, ho, wo = dy.shape[1], dy.shape[2], dy.shape[3]
    lib = _lib.load()
    if ksize == 3 and stride == 1 and not square_x and _WGRAD_ROWS:
        # straight from the NCHW tensors (csrc/wgrad_rows.hip); 0 = a shape that kernel does not take
        nws = lib.mcq_conv2d_wgrad_nchw_workspace_floats(n, cin, h, w, cout)
        if nws:
            ws = torch.empty(nws, dtype=torch.float32, device=x.device)
            dw, db = out if out is not None else _wgrad_outputs(x, dy, 3, want_bias)
            with _guard(x.device):
                check(lib.mcq_conv2d_wgrad_nchw_f32(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), _ptr(ws), n, cin, h, w, cout, _stream()),
                      "mcq_conv2d_wgrad_nchw_f32")
            _keep(ws, dw, db)
            return (dw, db) if want_bias else dw
    if ksize == 3 and stride == 2 and not square_x and _WGRAD_ROWS and (ho, wo) == (h // 2, w // 2):
        nws = lib.mcq_conv2d_wgrad_s2_nchw_workspace_floats(n, cin, h, w, cout)
        if nws:
            ws = torch.empty(nws, dtype=torch.float32, device=x.device)
            dw, db = out if out is not None else _wgrad_outputs(x, dy, 3, want_bias)
            with _guard(x.device):
                check(lib.mcq_conv2d_wgrad_s2_nchw_f32(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), _ptr(ws), n, cin, h, w, cout, _stream()),
                      "mcq_conv2d_wgrad_s2_nchw_f32")
            _keep(ws, dw, db)
            return (dw, db) if want_bias else dw
    if ksize == 1 and stride == 1 and _WGRAD_ROWS:
        nws = lib.mcq_conv2d_wgrad1x1_nchw_workspace_floats(n, cin, h, w, cout)
        if nws:
            ws = torch.empty(nws, dtype=torch.float32, device=x.device)
            dw, db = out if out is not None else _wgrad_outputs(x, dy, 1, want_bias)
            with _guard(x.device):
                check(lib.mcq_conv2d_wgrad1x1_nchw_f32(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), _ptr(ws), n, cin, h, w, cout,
                                                       int(square_x), _stream()), "mcq_conv2d_wgrad1x1_nchw_f32")
            _keep(ws, dw, db)
            return (dw, db) if want_bias else dw
    xt = torch.empty((n, h, w, cin), dtype=torch.float32, device=x.device)
    dyt = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=x.device)
    ws = torch.empty(lib.mcq_conv2d_wgrad_workspace_floats(n, cin, h, w, cout, ksize, stride), dtype=torch.float32, device=x.device)
    dw, db = out if out is not None else _wgrad_outputs(x, dy, ksize, want_bias)
    with _guard(x.device):
        check(lib.mcq_nchw_to_nhwc_pair_f32(_ptr(x), _ptr(xt), cin, h * w, int(square_x), _ptr(dy), _ptr(dyt), cout, ho * wo, n,
                                            _stream()), "mcq_nchw_to_nhwc_pair_f32")
        check(lib.mcq_conv2d_wgrad_f32(_ptr(xt), _ptr(dyt), _ptr(dw), _ptr(db), _ptr(ws), n, cin, h, w, cout, ksize, stride,
                                       _stream()), "mcq_conv2d_wgrad_f32")
    return (dw, db) if want_bias else dw


def channel_sum(x: torch.Tensor) -> torch.Tensor:
    x = _dev(x, "x")
    n, c, h, w = x.shape
    out = torch.empty((c,), dtype=torch.float32, device=x.device)
    ws = torch.empty((min(n, 16) * c,), dtype=torch.float32, device=x.device) if n > 1 else None
    with _guard(x.device):
        check(_lib.load().mcq_channel_sum_f32(_ptr(x), _ptr(out), _ptr(ws), n, c, h * w, _stream()), "mcq_channel_sum_f32")
    return out


def silu_bwd(x: torch.Tensor, dy: torch.Tensor, other: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dy * silu'(x) (+ other, a gradient that reaches x along a second path) in one launch."""
    x, dy = _dev(x, "x"), _dev(dy, "dy")
    other = None if other is None else _dev(other, "other")
    if dy.shape != x.shape or (other is not None and other.shape != x.shape):
        raise ValueError("silu_bwd: shape mismatch")
    dx = torch.empty_like(x)
    with _guard(x.device):
        check(_lib.load().mcq_silu_bwd_f32(_ptr(x), _ptr(dy), _ptr(other), _ptr(dx), x.numel(), _stream()), "mcq_silu_bwd_f32")
    return dx


def gate_bwd(a: torch.Tensor, b: torch.Tensor, dout: torch.Tensor):
    a, b, dout = _dev(a, "a"), _dev(b, "b"), _dev(dout, "dout")
    da, db = torch.empty_like(a), torch.empty_like(a)
    with _guard(a.device):
        check(_lib.load().mcq_gate_bwd_f32(_ptr(a), _ptr(b), _ptr(dout), _ptr(da), _ptr(db), a.numel(), _stream()), "mcq_gate_bwd_f32")
    return da, db


def gdn_bwd_prep(x: torch.Tensor, s: torch.Tensor, dy: torch.Tensor, inverse: bool):
    x, s, dy = _dev(x, "x"), _dev(s, "s"), _dev(dy, "dy")
    dxd, ds = torch.empty_like(x), torch.empty_like(x)
    with _guard(x.device):
        check(_lib.load().mcq_gdn_bwd_prep_f32(_ptr(x), _ptr(s), _ptr(dy), int(inverse), _ptr(dxd), _ptr(ds), x.numel(), _stream()),
              "mcq_gdn_bwd_prep_f32")
    return dxd, ds


def pixel_unshuffle2(x: torch.Tensor) -> torch.Tensor:
    x = _dev(x, "x")
    n, c, h2, w2 = x.shape
    out = torch.empty((n, c * 4, h2 // 2, w2 // 2), dtype=torch.float32, device=x.device)
    with _guard(x.device):
        check(_lib.load().mcq_pixel_unshuffle2_f32(_ptr(x), _ptr(out), n, c, h2 // 2, w2 // 2, _stream()), "mcq_pixel_unshuffle2_f32")
    return out


def silu(x: torch.Tensor) -> torch.Tensor:
    x = _dev(x, "x")
    y = torch.empty_like(x)
    with _guard(x.device):
        check(_lib.load().mcq_silu_f32(_ptr(x), _ptr(y), x.numel(), _stream()), "mcq_silu_f32")
    return y


def gate(a: torch.Tensor, b: torch.Tensor, x: torch.Tensor, dual_silu: bool = False) -> torch.Tensor:
    """a * sigmoid(b) + x; with `dual_silu` the result carries silu(result) as its twin (same launch)."""
    a, b, x = _dev(a, "a"), _dev(b, "b"), _dev(x, "x")
    out = torch.empty_like(a)
    out2 = torch.empty_like(a) if dual_silu else None
    with _guard(a.device):
        check(_lib.load().mcq_gate_f32(_ptr(a), _ptr(b), _ptr(x), _ptr(out), _ptr(out2), a.numel(), _stream()), "mcq_gate_f32")
    if out2 is not None:
        set_silu_twin(out, out2)
    return out


def axpby(a: torch.Tensor, b: torch.Tensor, alpha: float, beta: float, dual_silu: bool = False) -> torch.Tensor:
    a, b = _dev(a, "a"), _dev(b, "b")
    if a.shape != b.shape:
        raise ValueError("axpby: shape mismatch")
    out = torch.empty_like(a)
    out2 = torch.empty_like(a) if dual_silu else None
    with _guard(a.device):
        check(_lib.load().mcq_axpby_f32(_ptr(a), _ptr(b), float(alpha), float(beta), _ptr(out), _ptr(out2), a.numel(), _stream()), "mcq_axpby_f32")
    if out2 is not None:
        set_silu_twin(out, out2)
    return out


def vq_inner(x: torch.Tensor, cb: PackedCodebook) -> torch.Tensor:
    """[n, m, h, w, k] inner products <x_v, c_k> (mcq_vq_inner_f32)."""
    x = _dev(x, "x")
    n, c, h, w = x.shape
    out = torch.empty((n, cb.m, h, w, cb.k), dtype=torch.float32, device=x.device)
    with _guard(x.device):
        check(_lib.load().mcq_vq_inner_f32(_ptr(x), _ptr(cb.packed), _ptr(out), n, cb.m, cb.d, h, w, cb.k, _stream()), "mcq_vq_inner_f32")
    return out


def vq_softmax_bwd(logits: torch.Tensor, u_gumbel: Optional[torch.Tensor], ds: torch.Tensor, temperature: torch.Tensor, bound: float,
                   dlogits: Optional[torch.Tensor] = None, raw_logits: Optional[torch.Tensor] = None, rng: Optional[torch.Tensor] = None):
    """In place on `ds` (dSample -> d dist); returns (rowsum, dtrow), each [n, m, h, w].  `dlogits`: a gradient on the returned
    logits themselves, with `raw_logits` = the logits before the random drop.  `u_gumbel` None: the forward's Gumbel draw is
    remade from its generator snapshot `rng`."""
    logits, ds = _dev(logits, "logits"), _dev(ds, "ds")
    if u_gumbel is None and rng is None:
        raise ValueError("vq_softmax_bwd: give the Gumbel draw or the forward's generator snapshot")
    u_gumbel = None if u_gumbel is None else _dev(u_gumbel, "u_gumbel")
    rng = None if rng is None else _dev(rng, "rng", torch.int64)
    if dlogits is not None:
        dlogits, raw_logits = _dev(dlogits, "dlogits"), _dev(raw_logits, "raw_logits")
        if dlogits.shape != logits.shape or raw_logits.shape != logits.shape:
            raise ValueError("dlogits / raw_logits must have the logits' shape")
    n, m, h, w, k = logits.shape
    t = _dev(temperature.detach().reshape(-1), "temperature")
    rowsum = torch.empty((n, m, h, w), dtype=torch.float32, device=logits.device)
    dtrow = torch.empty_like(rowsum)
    with _guard(logits.device):
        check(_lib.load().mcq_vq_softmax_bwd_f32(_ptr(logits), _ptr(u_gumbel), _ptr(rng), _ptr(ds), _ptr(t), float(bound), _ptr(rowsum), _ptr(dtrow),
                                                 _ptr(dlogits), _ptr(raw_logits), n, m, h, w, k, _stream()), "mcq_vq_softmax_bwd_f32")
    return rowsum, dtrow


def vq_soft_bwd(ddist: torch.Tensor, rowsum: torch.Tensor, x: torch.Tensor, ddeq: torch.Tensor, index: torch.Tensor,
                hot: torch.Tensor, cb: PackedCodebook):
    """(dx [n, m*d, h, w], dcodebook [m, k, d]) of the soft assignment + soft dequantisation."""
    x, ddeq = _dev(x, "x"), _dev(ddeq, "ddeq")
    n, m, h, w, k = ddist.shape
    c, hw = x.shape[1], h * w
    xt = torch.empty((n, h, w, c), dtype=torch.float32, device=x.device)
    dqt = torch.empty_like(xt)
    with _guard(x.device):                                     # both channel-major copies from one launch
        check(_lib.load().mcq_nchw_to_nhwc_pair_f32(_ptr(x), _ptr(xt), c, hw, 0, _ptr(ddeq), _ptr(dqt), c, hw, n, _stream()),
              "mcq_nchw_to_nhwc_pair_f32")
    dx = torch.empty_like(x)
    dcb = torch.empty_like(cb.codebook)
    with _guard(x.device):
        check(_lib.load().mcq_vq_soft_bwd_f32(_ptr(ddist), _ptr(rowsum), _ptr(x), _ptr(xt), _ptr(dqt), _ptr(index), _ptr(hot),
                                              _ptr(cb.codebook), _ptr(dx), _ptr(dcb), n, m, cb.d, h, w, k, _stream()), "mcq_vq_soft_bwd_f32")
    return dx, dcb
