"""ctypes binding of libmcquic_hip.so (the C-ABI declared in include/mcquic_hip.h).

There is no fallback: if the shared library is missing or a symbol is absent, importing the op layer
raises.  The product path never routes through a CPU or PyTorch implementation of these ops.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int32, c_int64, c_size_t, c_uint32, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCQUIC_AMD_LIB") or os.path.join(HERE, "libmcquic_hip.so")   # override: kernel A/B experiments

MCQ_OK, MCQ_EINVAL, MCQ_ELAUNCH, MCQ_ETOOLARGE = 0, -1, -2, -3
_ERR = {MCQ_EINVAL: "MCQ_EINVAL (invalid argument)", MCQ_ELAUNCH: "MCQ_ELAUNCH (kernel launch failed)",
        MCQ_ETOOLARGE: "MCQ_ETOOLARGE (tensor exceeds the addressing window)"}

CONV_SILU_IN, CONV_SQUARE_IN, CONV_SILU_OUT, CONV_RESIDUAL = 0x1, 0x2, 0x4, 0x8
CONV_GDN, CONV_IGDN, CONV_GATE, CONV_SHUFFLE2, CONV_DUAL_SILU, CONV_MUL, CONV_DSILU_MUL = 0x10, 0x20, 0x40, 0x80, 0x100, 0x200, 0x400
CONV_WINOGRAD = 0x800
CONV_WINOGRAD2D = 0x1000
CONV_WINOGRAD2D16 = 0x2000
CONV_GDN_BWD, CONV_IGDN_BWD = 0x4000, 0x8000
CONV_GATE_BWD = 0x10000
CONV_TAPS_LR = 0x20000
CONV_POST_GDN, CONV_POST_IGDN, CONV_POST_GATE = 0x40000, 0x80000, 0x100000     # (round 6) the following 1x1 layer inside the 3x3 launch


class ConvDesc(Structure):
    """struct mcq_conv_desc (include/mcquic_hip.h)."""
    _fields_ = [("x", c_void_p), ("w_packed", c_void_p), ("bias", c_void_p), ("y", c_void_p), ("y_silu", c_void_p), ("res", c_void_p),
                ("mul", c_void_p), ("gate_id", c_void_p),
                ("N", c_int32), ("Cin", c_int32), ("H", c_int32), ("W", c_int32), ("Cout", c_int32),
                ("ksize", c_int32), ("stride", c_int32), ("flags", c_uint32), ("res_scale", c_float),
                ("tile", c_int32), ("post_w", c_void_p), ("post_bias", c_void_p)]


# every symbol include/mcquic_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "mcq_packed_conv_weight_floats": (c_size_t, [c_int32, c_int32, c_int32]),
    "mcq_pack_conv_weight_f32": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "mcq_dgrad_weight_shape": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "mcq_pack_conv_dgrad_weight_f32": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p]),
    "mcq_nonneg_reparam_bwd_f32": (c_int32, [c_void_p, c_void_p, c_float, c_void_p, c_int64, c_void_p]),
    "mcq_conv2d_f32": (c_int32, [POINTER(ConvDesc), c_void_p]),
    "mcq_packed_post1x1_floats": (c_size_t, []),
    "mcq_pack_post1x1_weight_f32": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "mcq_conv2d_post_ok": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_uint32]),
    "mcq_packed_conv_winograd_floats": (c_size_t, [c_int32, c_int32]),
    "mcq_pack_conv_weight_winograd_f32": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "mcq_pack_conv_dgrad_weight_winograd_f32": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "mcq_packed_conv_winograd2d_floats": (c_size_t, [c_int32, c_int32]),
    "mcq_pack_conv_weight_winograd2d_f32": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "mcq_packed_conv_winograd16_floats": (c_size_t, [c_int32, c_int32]),
    "mcq_pack_conv_weight_winograd16_f32": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "mcq_conv2d_winograd_ok": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_uint32]),
    "mcq_conv2d_small_launch": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_uint32, c_int32]),
    "mcq_pack_conv_weight_max_multi": (c_int32, []),
    "mcq_pack_conv_weight_multi_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p]),
    "mcq_pack_conv_weight_multi_masked_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p]),
    "mcq_conv_section_trace": (None, [c_int32]),
    "mcq_conv_sections_used": (c_uint32, [c_void_p]),
    "mcq_conv2d_max_multi": (c_int32, []),
    "mcq_conv2d_multi_f32": (c_int32, [POINTER(ConvDesc), c_int32, c_void_p]),
    "mcq_nonneg_reparam_f32": (c_int32, [c_void_p, c_float, c_float, c_void_p, c_int64, c_void_p]),
    "mcq_nonneg_reparam_max_multi": (c_int32, []),
    "mcq_nonneg_reparam_multi_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "mcq_packed_codebook_floats": (c_size_t, [c_int32, c_int32, c_int32]),
    "mcq_vq_pack_codebook_f32": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "mcq_vq_assign_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                    c_int32, c_void_p]),
    "mcq_vq_assign_workspace_bytes": (c_size_t, [c_int32] * 6),
    "mcq_vq_assign_ws_f32": (c_int32, [c_void_p, c_void_p, c_void_p] + [c_int32] * 6 + [c_void_p, c_void_p]),
    "mcq_vq_gather_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                    c_int32, c_void_p]),
    "mcq_vq_logits_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                    c_int32, c_void_p]),
    "mcq_vq_gumbel_sample_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "mcq_vq_max_levels": (c_int32, []),
    "mcq_vq_step_prologue_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mcq_vq_temperature_grad_f32": (c_int32, [c_void_p, c_void_p, c_float, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "mcq_freq_ema_update_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_double, c_void_p]),
    "mcq_nonneg_reparam_bwd2_f32": (c_int32, [c_void_p, c_void_p, c_float, c_void_p, c_int64, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_void_p]),
    "mcq_hash_uniform_f32": (c_int32, [c_void_p, c_uint32, c_void_p, c_int64, c_void_p]),
    "mcq_vq_dequant_soft_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                          c_int32, c_void_p]),
    "mcq_vq_inner_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "mcq_vq_softmax_bwd_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "mcq_vq_soft_bwd_f32": (c_int32, [c_void_p] * 10 + [c_int32] * 6 + [c_void_p]),
    "mcq_nchw_to_nhwc_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "mcq_conv2d_wgrad_workspace_floats": (c_size_t, [c_int32] * 7),
    "mcq_conv2d_wgrad_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                       c_int32, c_int32, c_int32, c_void_p]),
    "mcq_conv2d_wgrad_nchw_workspace_floats": (c_size_t, [c_int32] * 5),
    "mcq_conv2d_wgrad_nchw_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                            c_int32, c_void_p]),
    "mcq_conv2d_wgrad_nchw_max_group": (c_int32, []),
    "mcq_conv2d_wgrad_nchw_group_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32,
                                                  c_int32, c_int32, c_void_p]),
    "mcq_conv2d_wgrad_s2_nchw_workspace_floats": (c_size_t, [c_int32] * 5),
    "mcq_conv2d_wgrad_s2_nchw_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                               c_int32, c_void_p]),
    "mcq_conv2d_wgrad1x1_nchw_workspace_floats": (c_size_t, [c_int32] * 5),
    "mcq_conv2d_wgrad1x1_nchw_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                               c_int32, c_int32, c_void_p]),
    "mcq_wgrad_defer": (None, [c_int32]),
    "mcq_wgrad_pending": (c_int32, []),
    "mcq_wgrad_flush": (c_int32, [c_int32, c_void_p]),
    "mcq_nchw_to_nhwc_pair_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_int32,
                                            c_int32, c_void_p]),
    "mcq_channel_sum_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "mcq_silu_f32": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p]),
    "mcq_gate_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mcq_axpby_f32": (c_int32, [c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_int64, c_void_p]),
    "mcq_silu_bwd_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mcq_gate_bwd_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mcq_gdn_bwd_prep_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_void_p]),
    "mcq_pixel_unshuffle2_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "mcq_add_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mcq_add3_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mcq_mse_workspace_bytes": (c_size_t, [c_int64]),
    "mcq_mse_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mcq_mse_bwd_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mcq_gather_flat_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "mcq_adam_chunk": (c_int32, []),
    "mcq_adam_step_f32": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_double, c_double, c_double, c_double,
                                    c_double, c_int32, c_int32, c_void_p, c_void_p]),
    "mcq_sumsq_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mcq_clip_by_norm_f32": (c_int32, [c_void_p, c_void_p, c_float, c_float, c_void_p, c_int64, c_void_p]),
    "mcq_detransform_u8": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p]),
    "mcq_pmf_to_quantized_cdf": (c_int32, [c_void_p, c_int32, c_int32, c_void_p]),
    "mcq_rans_encode_with_indexes": (c_int64, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                               c_void_p, c_int64]),
    "mcq_rans_decode_with_indexes": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_int32, c_void_p]),
    "mcq_rans_encode_batch_with_indexes": (c_int32, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                     c_void_p, c_int32, c_void_p, c_int64, c_void_p, c_int32]),
    "mcq_rans_decode_batch_with_indexes": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                                     c_void_p, c_void_p, c_int32, c_void_p, c_int32]),
    "mcq_ms_ssim_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32, c_int32]),
    "mcq_ms_ssim_u8": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "mcq_ms_ssim_window": (None, [c_void_p]),
    "mcq_sqdiff_sum_u8": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "mcq_group_norm_workspace_floats": (c_size_t, [c_int32, c_int32, c_int32, c_int32]),
    "mcq_group_norm_f32": (c_int32, [c_void_p] * 8 + [c_int32, c_int32, c_int32, c_int32, c_float, c_void_p]),
    "mcq_group_norm_bwd_workspace_floats": (c_size_t, [c_int32, c_int32, c_int32, c_int32]),
    "mcq_group_norm_bwd_f32": (c_int32, [c_void_p] * 9 + [c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "mcq_selftest_launch_failure": (c_int32, [c_void_p]),
    "mcq_version": (c_char_p, []),
    "mcq_abi_version": (c_int32, []),
}

ABI_VERSION = 9          # MCQ_ABI_VERSION of include/mcquic_hip.h these prototypes were written against

_lib = None


def load() -> ctypes.CDLL:
    """dlopen the in-tree library and type every entry point; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: run `python -m mcquic_amd.build` (or __graft_entry__.build()). "
            "mcquic_amd has no CPU / PyTorch fallback for its HIP kernels.")
    lib = ctypes.CDLL(LIB_PATH)
    try:
        lib.mcq_abi_version.restype = c_int32
        built = int(lib.mcq_abi_version())
    except AttributeError:
        built = None
    if built != ABI_VERSION:        # a stale .so under these prototypes would misalign arguments silently
        raise ImportError(f"{LIB_PATH} was built for ABI version {built}, this binding needs {ABI_VERSION}: "
                          "rebuild with `python -m mcquic_amd.build --force`")
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != MCQ_OK:
        raise RuntimeError(f"{what} failed: {_ERR.get(status, status)}")
