"""ctypes binding of libvitk.so (include/vitk.h).  No torch types cross this boundary.

The product path has NO fallback: if the library is missing this module raises at
import of the first kernel call site, with the command that builds it.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VITK_LIB") or os.path.join(HERE, "libvitk.so")     # VITK_LIB: A/B against another build of the library (tools)
# same ABI; its 16-bit type is IEEE half (model.half()).  With VITK_LIB set: the sibling of that file (libvitk*.so -> libvitk_f16*.so)
LIB_PATH_F16 = (os.path.join(os.path.dirname(LIB_PATH), os.path.basename(LIB_PATH).replace("libvitk", "libvitk_f16", 1)) if os.environ.get("VITK_LIB")
                else os.path.join(HERE, "libvitk_f16.so"))

F32, BF16 = 0, 1          # dtype tags; 1 = "the library's 16-bit type" (bfloat16 in libvitk, half in libvitk_f16)
HALF_TYPE_F16 = 2
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_RESID, EPI_GELU_BWD, EPI_RESID16, EPI_BIAS_GELU_DG, EPI_MUL_AUX = 0, 1, 2, 3, 4, 5, 6, 7
EPI_BIAS_GELU_DG8, EPI_MUL_AUX8 = 8, 9       # the gelu' factor as 8-bit codes (include/vitk.h)
VITK_VERSION = 138


class RowMap(C.Structure):
    _fields_ = [("group", C.c_int64), ("gstride", C.c_int64), ("offset", C.c_int64)]


class Mat(C.Structure):
    _fields_ = [("p", C.c_void_p), ("dt", C.c_int), ("s_b1", C.c_int64), ("s_b2", C.c_int64),
                ("s_row", C.c_int64), ("s_col", C.c_int64)]


class HND(C.Structure):
    _fields_ = [("p", C.c_void_p), ("s_h", C.c_int64), ("s_n", C.c_int64)]


class BHND(C.Structure):
    _fields_ = [("p", C.c_void_p), ("s_b", C.c_int64), ("s_h", C.c_int64), ("s_n", C.c_int64)]


IDENT = RowMap(0, 0, 0)

_vp, _i, _i64, _f, _u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64

# name -> (restype, argtypes); mirrors include/vitk.h one to one (tests/test_abi.py checks the
# header against this table and against the exported symbols).
SIGNATURES = {
    "vitk_version": (_i, []),
    "vitk_last_error": (C.c_char_p, []),
    "vitk_layernorm_fwd": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _i64, _i64, _f, RowMap, RowMap, _vp, _i64, _i64, _vp]),
    "vitk_layernorm_bwd_blocks": (_i64, [_i64, _i64]),
    "vitk_layernorm_bwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i64, _i64, RowMap, RowMap, RowMap, _vp]),
    "vitk_layernorm_bwd_finalize": (_i, [_vp, _i64, _i64, _vp, _vp, _i, _vp, _vp]),
    "vitk_layernorm_bwd_finalize_ex": (_i, [_vp, _i64, _i64, _vp, _vp, _i, _vp, _i, _vp]),
    "vitk_colsum_partials": (_i, [_vp, _i64, _i64, _i64, _vp, _i, _i, _vp]),
    "vitk_colsum_ws_floats": (_i64, [_i64, _i64]),
    "vitk_colsum": (_i, [_vp, _i, _i64, _i64, _i64, _vp, _i, _i, _vp, _vp]),
    "vitk_gemm_nt_bf16": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i, _vp, _vp, _vp, _vp]),
    "vitk_half_type": (_i, []),
    "vitk_adam_step": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _i, _i64, _f, _vp]),
    "vitk_gemm_nt_bf16_drop": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i, _vp, _vp, _vp, _vp, _f, C.c_uint32, _vp]),
    "vitk_layernorm_bwd_drop": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i64, _i64, RowMap, RowMap, RowMap, _f, C.c_uint32, _vp]),
    "vitk_layernorm_bwd_s16": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i64, _i64, RowMap, RowMap, RowMap, _vp]),
    "vitk_gemm_nt_fp8": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i, _vp, _vp, _vp, _f, _vp]),
    "vitk_fp8_amax_scale": (_i, [_vp, _i, _i64, _vp, _vp]),
    "vitk_quantize_fp8": (_i, [_vp, _i, _vp, _i64, _vp, _f, _vp]),
    "vitk_layernorm_fwd_fp8": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _i64, _i64, _f, RowMap, RowMap, _vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "vitk_gemm_nt_fp8_ex": (_i, [_vp, _i64, _i, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vitk_fp8_update_scales": (_i, [_vp, _vp, _i64, _vp]),
    "vitk_gemm_nt_fp8_v2": (_i, [_vp, _i64, _i, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "vitk_gemm_nt_fp8_colsum_rows": (_i64, [_i64, _i64, _i64, _i64]),
    "vitk_quantize_fp8_delayed": (_i, [_vp, _i, _vp, _i64, _vp, _vp, _i, _vp]),
    "vitk_fp8_update_scales_fmt": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "vitk_gemm_tn_fp8_splits": (_i64, [_i64, _i64, _i64, _i]),
    "vitk_gemm_tn_fp8": (_i, [_vp, _i64, _vp, _i64, _vp, _i, _i64, _i, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _i, _vp]),
    "vitk_gemm_nt_colsum_rows": (_i64, [_i64, _i64, _i64, _i64]),
    "vitk_gemm_nt_plan": (_i, [_i64, _i64, _i64, _i64, _vp]),
    "vitk_comm_unique_id": (_i, [_vp]),
    "vitk_comm_init": (_i, [_vp, _i, _i, _vp]),
    "vitk_comm_allreduce": (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    "vitk_comm_destroy": (_i, [_vp]),
    "vitk_pack_w_nt_bytes": (_i64, [_i64, _i64]),
    "vitk_pack_w_nt": (_i, [_vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "vitk_pack_w_nt_many": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "vitk_gemm_nt_bf16_gelu_bwd_colsum": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp]),
    "vitk_gemm_nt_bf16_mul_aux_colsum": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp]),
    "vitk_gemm_nt_bf16_mul_aux8_colsum": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp]),
    "vitk_gemm_tn_splits": (_i64, [_i64, _i64, _i64]),
    "vitk_set_cu_reserve": (_i, [_i]),
    "vitk_get_cu_reserve": (_i, []),
    "vitk_test_occupy_cus": (_i, [_i, _f, _vp]),
    "vitk_gemm_tn_bf16": (_i, [_vp, _i64, _vp, _i64, _vp, _i, _i64, _i, _i64, _i64, _i64, _vp, _i64, _vp]),
    "vitk_gemm_tn_pair_splits": (_i64, [_i64, _i64, _i64, _i64, _i64]),
    "vitk_gemm_tn_bf16_pair": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i, _i64, _i64, _i, _i64, _vp, _i64, _vp]),
    "vitk_gemm_generic": (_i, [Mat, Mat, Mat, _vp, _i, _i64, _i64, _i64, _i64, _i64, _f, _f, _vp]),
    "vitk_attn_fwd_bf16": (_i, [BHND, BHND, BHND, BHND, _vp, _i64, _i64, _i64, _i64, _f, _vp]),
    "vitk_attn_bwd_bf16": (_i, [BHND, BHND, BHND, BHND, BHND, _vp, _vp, BHND, BHND, BHND, _i64, _i64, _i64, _i64, _f, _vp]),
    "vitk_attn_fwd_bf16_drop": (_i, [BHND, BHND, BHND, BHND, _vp, _i64, _i64, _i64, _i64, _f, _f, C.c_uint32, _vp]),
    "vitk_attn_bwd_bf16_drop": (_i, [BHND, BHND, BHND, BHND, BHND, _vp, _vp, BHND, BHND, BHND, _i64, _i64, _i64, _i64, _f, _f, C.c_uint32, _vp]),
    "vitk_split2": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "vitk_attn_fwd_x2": (_i, [BHND] * 7 + [_vp, _i64, _i64, _i64, _i64, _f, _vp]),
    "vitk_attn_bwd_x2": (_i, [BHND] * 9 + [_vp, _vp, BHND, BHND, BHND, _i64, _i64, _i64, _i64, _f, _vp]),
    "vitk_dropout_keep": (_i, [_vp, _i64, _i64, _f, C.c_uint32, _vp]),
    "vitk_attn_varlen_fwd_bf16": (_i, [HND, HND, HND, HND, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _f, _vp]),
    "vitk_attn_varlen_bwd_bf16": (_i, [HND, HND, HND, HND, HND, _vp, _vp, HND, HND, HND, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _f, _vp]),
    "vitk_attn_varlen_fwd_bf16_drop": (_i, [HND, HND, HND, HND, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _f, _f, C.c_uint32, _vp]),
    "vitk_attn_varlen_bwd_bf16_drop": (_i, [HND, HND, HND, HND, HND, _vp, _vp, HND, HND, HND, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _f, _f, C.c_uint32, _vp]),
    "vitk_rmsnorm_heads_rows": (_i64, [_i64, _i64]),
    "vitk_rmsnorm_heads_fwd": (_i, [_vp, _i64, _vp, _vp, _i64, _vp, _i, _i64, _i64, _i64, _vp]),
    "vitk_rmsnorm_heads_bwd": (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _i, _i64, _i64, _i64, _vp]),
    "vitk_softmax_fwd": (_i, [_vp, _vp, _i, _i64, _i64, _f, _vp]),
    "vitk_softmax_bwd": (_i, [_vp, _vp, _vp, _i, _i64, _i64, _f, _vp]),
    "vitk_patchify": (_i, [_vp, _vp, _i, _i64, _i64, _i64, _i64, _i64, _i64, _vp]),
    "vitk_unpatchify": (_i, [_vp, _vp, _i, _i64, _i64, _i64, _i64, _i64, _i64, _vp]),
    "vitk_patchify_cpp": (_i, [_vp, _vp, _i, _i64, _i64, _i64, _i64, _i64, _i64, _vp]),
    "vitk_unpatchify_cpp": (_i, [_vp, _vp, _i, _i64, _i64, _i64, _i64, _i64, _i64, _vp]),
    "vitk_patch_ln_serves": (_i, [_i, _i64, _i64, _i64, _i64, _i64]),
    "vitk_patch_ln_bwd_blocks": (_i64, [_i64]),
    "vitk_patch_ln_fwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _f, _vp]),
    "vitk_patch_ln_bwd_params": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp]),
    "vitk_gather_add2": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i64, _vp]),
    "vitk_csr_rowsum": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i64, _i64, _vp]),
    "vitk_gelu_fwd": (_i, [_vp, _vp, _i, _i64, _vp]),
    "vitk_gelu_bwd": (_i, [_vp, _vp, _vp, _i, _i64, _vp]),
    "vitk_add_rows": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i64, _i64, _vp]),
    "vitk_cast": (_i, [_vp, _i, _vp, _i, _i64, _vp]),
    "vitk_cast_many": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp]),
    "vitk_fold_many": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "vitk_write_cls_rows": (_i, [_vp, _i, _vp, _vp, _i, _i64, _i64, _i64, _i64, _vp]),
    "vitk_mean_pool_fwd": (_i, [_vp, _i, _vp, _i, _i64, _i64, _i64, _vp]),
    "vitk_mean_pool_bwd": (_i, [_vp, _i, _vp, _i, _i64, _i64, _i64, _vp]),
    "vitk_dropout_fwd": (_i, [_vp, _vp, _vp, _i, _i64, _f, _u64, _u64, _vp]),
    "vitk_dropout_bwd": (_i, [_vp, _vp, _vp, _i, _i64, _f, _vp]),
    "vitk_copy_cols": (_i, [_vp, _i64, _vp, _i64, _i, _i64, _i64, _i64, _vp]),
    "vitk_split_bf16x3": (_i, [_vp, _i64, _vp, _i64, _i64, _i64, _i64, _i, _vp]),
    "vitk_concat_tokens": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i64, _i64, _i64, _vp]),
    "vitk_gather_tokens": (_i, [_vp, _vp, _vp, _i, _i64, _i64, _i64, _i64, _i, _vp]),
    "vitk_transpose": (_i, [_vp, _vp, _i, _i64, _i64, _vp]),
}

_lib = None
_lib_f16 = None


class VitkError(RuntimeError):
    pass


def _open(path: str, half_type: int):
    if not os.path.exists(path):
        raise VitkError(
            f"{path} not found: the HIP kernel library is not built. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `python -m vit_pytorch_amd._build`). There is no CPU/eager fallback.")
    import torch  # noqa: F401  -- torch must load ITS HIP runtime first; a second runtime copy cannot see the device
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    v = lib.vitk_version()
    if v != VITK_VERSION:
        raise VitkError(f"{os.path.basename(path)} version {v} != binding version {VITK_VERSION}; rebuild")
    if lib.vitk_half_type() != half_type:
        raise VitkError(f"{os.path.basename(path)} was built for the wrong 16-bit type; rebuild")
    return lib


def load():
    """Load libvitk.so (16-bit type: bfloat16) once.  Raises VitkError (never falls back) if it is absent or stale."""
    global _lib
    if _lib is None:
        _lib = _open(LIB_PATH, BF16)
    return _lib


def load_f16():
    """Load libvitk_f16.so (the same kernels instantiated for IEEE half) once."""
    global _lib_f16
    if _lib_f16 is None:
        _lib_f16 = _open(LIB_PATH_F16, HALF_TYPE_F16)
    return _lib_f16


_tls = threading.local()


def note_last_lib(lib):
    _tls.lib = lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = (getattr(_tls, "lib", None) or load()).vitk_last_error().decode("utf-8", "replace")
        raise VitkError(f"{what}: vitk error {rc}: {msg}")
