"""ctypes binding of libmvlpt_hip.so (C ABI in include/mvlpt_hip.h).

There is NO CPU fallback: if the shared library is missing (build it with ``make`` or
``python -c 'import __graft_entry__ as g; g.build()'``) importing this module raises, and every
compute call needs a HIP device.
"""
from __future__ import annotations

import ctypes as C
import os

# PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64; it must be loaded FIRST so that libmvlpt_hip.so binds to
# the same HIP runtime instance (two runtimes in one process do not see each other's devices, streams or pointers).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MVLPT_HIP_LIB") or os.path.join(_HERE, "libmvlpt_hip.so")   # override: A/B builds of the same ABI

DT_F32, DT_F16, DT_BF16 = 0, 1, 2
LABEL_INT64, LABEL_PROB_F32 = 0, 1
EPI_STORE16, EPI_GELU, EPI_RESID32, EPI_GELUBWD, EPI_STORE32, EPI_GELU_SPLIT, EPI_GELUBWD_SPLIT, EPI_STORE_SPLIT = 0, 1, 2, 3, 4, 5, 6, 7
PREC_FAST, PREC_SPLIT_GRAD, PREC_SPLIT_ALL = 0, 1, 2


class MvlptArch(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "image_resolution", "patch_size", "vision_width", "vision_layers", "vision_heads",
        "context_length", "text_width", "text_layers", "text_heads", "embed_dim", "compute_dtype")]


class MvlptKernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_int64), ("ms", C.c_double),
                ("flops", C.c_double), ("bytes", C.c_double), ("busy_ms", C.c_double), ("flops_executed", C.c_double)]


class MvlptImageDesc(C.Structure):
    _fields_ = [("offset", C.c_int64)] + [(n, C.c_int32) for n in (
        "height", "width", "crop_top", "crop_left", "crop_height", "crop_width", "resize_height", "resize_width",
        "out_top", "out_left", "flip", "reserved")]


_vp, _i, _f = C.c_void_p, C.c_int, C.c_float

# name -> (restype, argtypes): every symbol include/mvlpt_hip.h declares
SIGNATURES = {
    "mvlpt_create": (_i, [C.POINTER(MvlptArch), C.POINTER(_vp)]),
    "mvlpt_destroy": (_i, [_vp]),
    "mvlpt_set_precision": (_i, [_vp, _i]),
    "mvlpt_trim": (_i, [_vp]),
    "mvlpt_debug_checksums": (_i, [_vp, _i, C.POINTER(C.c_uint64), _i]),
    "mvlpt_set_ln_fold": (_i, [_vp, _i, _i]),
    "mvlpt_set_resid_packed": (_i, [_vp, _i]),
    "mvlpt_set_vpt_dropout": (_i, [_vp, _vp, _i, _i, _i, _i]),
    "mvlpt_last_error": (C.c_char_p, [_vp]),
    "mvlpt_version": (C.c_char_p, []),
    "mvlpt_stream_create_cus": (_i, [_i, _i, C.POINTER(_vp)]),
    "mvlpt_stream_destroy": (_i, [_vp]),
    "mvlpt_stream_cus": (_i, [_vp]),
    "mvlpt_load_frozen": (_i, [_vp, C.c_char_p, _vp, _i, C.POINTER(C.c_int64), _i, _vp]),
    "mvlpt_frozen_ready": (_i, [_vp]),
    "mvlpt_image_fwd": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _i, _vp]),
    "mvlpt_image_bwd": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "mvlpt_text_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _i, _vp]),
    "mvlpt_text_bwd": (_i, [_vp, _vp, _vp, _vp]),
    "mvlpt_logits_fwd": (_i, [_vp, _vp, _vp, _f, _vp, _vp, _i, _i, _vp, _vp]),
    "mvlpt_logits_bwd": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "mvlpt_cross_entropy": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "mvlpt_op_gemm": (_i, [_i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mvlpt_op_gemm_split": (_i, [_i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mvlpt_op_layernorm_fwd_split": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "mvlpt_op_layernorm_bwd_split": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "mvlpt_op_attention32_fwd": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mvlpt_op_attention32_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mvlpt_op_pack_weight_mixed": (_i, [_i, _vp, _i, _i, _i, _vp, C.POINTER(C.c_int), _vp]),
    "mvlpt_op_gemm_mixed": (_i, [_i, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mvlpt_op_cast_mixed": (_i, [_i, _vp, _vp, C.c_int64, _i, _vp]),
    "mvlpt_op_fold_vectors": (_i, [_i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "mvlpt_op_gemm_ln_producer": (_i, [_i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, C.POINTER(C.c_int), _vp]),
    "mvlpt_op_gemm_folded": (_i, [_i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "mvlpt_op_fold_weight": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _i, _vp]),
    "mvlpt_op_respk_pack": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "mvlpt_op_assemble_packed": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mvlpt_op_respk_unpack": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp]),
    "mvlpt_op_gemm_residp": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, C.POINTER(C.c_int), _vp]),
    "mvlpt_op_layernorm_fwd_mixed": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "mvlpt_op_layernorm_bwd_mixed": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "mvlpt_op_attention32_fwd_mixed": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mvlpt_op_attention32_bwd_mixed": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mvlpt_op_layernorm_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "mvlpt_op_layernorm_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "mvlpt_op_attention_fwd": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mvlpt_op_attention_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mvlpt_op_cast": (_i, [_i, _vp, _vp, C.c_int64, _vp]),
    "mvlpt_preprocess": (_i, [_vp, _vp, C.c_int64, C.POINTER(MvlptImageDesc), _i, _i, _i, C.POINTER(_f), C.POINTER(_f), _vp, _i, _vp, _vp]),
    "mvlpt_profile_begin": (_i, [_vp, _i]),
    "mvlpt_profile_pause": (_i, [_vp, _i]),
    "mvlpt_profile_end": (_i, [_vp, C.POINTER(MvlptKernelStat), _i]),
}


def load_library(path: str = LIB_PATH) -> C.CDLL:
    if not os.path.isfile(path):
        raise ImportError(
            f"{path} not found: build the HIP extension first (`make` at the repo root, or "
            "`__graft_entry__.build()`); mvlpt_amd has no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    return lib


lib = load_library()


def last_error(handle=None) -> str:
    msg = lib.mvlpt_last_error(handle)
    return msg.decode() if msg else ""


def check(rc: int, handle=None, what: str = "") -> None:
    if rc != 0:
        raise RuntimeError(f"libmvlpt_hip {what} failed (code {rc}): {last_error(handle) or last_error(None)}")
