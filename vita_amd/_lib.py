"""ctypes binding of libvita_hip.so (the C ABI declared in include/vita_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent this
module raises, loudly.  Build with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C vita_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VITA_AMD_LIB") or os.path.join(_HERE, "libvita_hip.so")   # env override: kernel experiments

i64, i32, f32, vp = C.c_int64, C.c_int32, C.c_float, C.c_void_p


class GemmDesc(C.Structure):
    _fields_ = [
        ("a", vp), ("b", vp), ("c", vp), ("bias", vp), ("aux_in", vp), ("aux_out", vp),
        ("m", i64), ("n", i64), ("k", i64),
        ("lda", i64), ("ldb", i64), ("ldc", i64), ("ld_aux", i64),
        ("batch_outer", i64), ("batch_inner", i64),
        ("a_stride_outer", i64), ("a_stride_inner", i64),
        ("b_stride_outer", i64), ("b_stride_inner", i64),
        ("c_stride_outer", i64), ("c_stride_inner", i64),
        ("a_layout", i32), ("b_layout", i32), ("c_dtype", i32), ("epilogue", i32), ("split_k", i32),
        ("row_seg_phys", i32), ("row_seg_log", i32), ("col_seg_phys", i32), ("col_seg_log", i32),
        ("alpha", f32),
    ]


# name -> argtypes (restype int unless noted).  Mirrors include/vita_hip.h one to one.
SIGNATURES = {
    "vita_version": [],
    "vita_spin_us": [i64, vp],
    "vita_gemm_bf16": [C.POINTER(GemmDesc), vp],
    "vita_gemm_bf16_ex": [C.POINTER(GemmDesc), i32, vp],
    "vita_pack_weight_bf16": [vp, i64, i64, vp, i64, i64, i32, i32, i32, i32, vp],
    "vita_pack_vec_f32": [vp, i64, vp, i64, i32, i32, vp],
    "vita_unpack_vec_f32": [vp, i64, vp, i64, i32, i32, vp],
    "vita_layernorm_fwd": [vp, i64, vp, vp, vp, i64, i32, vp, vp, i64, i64, f32, vp],
    "vita_layernorm_bwd_nblk": [i64],
    "vita_layernorm_bwd": [vp, i64, i32, vp, i64, vp, vp, vp, vp, vp, i64, vp, i64, vp, vp, vp, i64, i64, vp],
    "vita_layernorm_bwd_acc": [vp, i64, i32, vp, i64, vp, vp, vp, vp, vp, i64, vp, i64, vp, vp, vp, i64, i64, vp],
    "vita_layernorm_bwd_noalloc": [vp, i64, i32, vp, i64, vp, vp, vp, vp, vp, i64, vp, i64, vp, vp, i64, i64, vp],
    "vita_argsort_mask": [vp, i64, i64, i64, vp, vp, vp, vp],
    "vita_row_gather_f32": [vp, i64, i64, vp, i64, vp, vp, vp, i64, i64, i64, i64, vp],
    "vita_row_scatter_f32": [vp, i64, i64, vp, i64, vp, i64, i64, vp, i64, i64, vp],
    "vita_im2col_bf16": [vp, i64, i64, i64, i64, i64, i64, i64, i64, vp, i64, vp, i64, i64, i64, vp],
    "vita_tokens_assemble": [vp, i64, vp, vp, vp, i64, i64, i64, vp, i64, i64, vp],
    "vita_cast_f32_bf16": [vp, i64, vp, i64, i64, i64, vp],
    "vita_colsum": [vp, i32, i64, i64, i64, vp, vp],
    "vita_colsum_acc": [vp, i32, i64, i64, i64, vp, vp],
    "vita_softmax_rows": [vp, i64, vp, i64, i64, i64, f32, vp],
    "vita_softmax_bwd_rows": [vp, i64, vp, i64, vp, i64, i64, i64, f32, vp],
    "vita_l2_normalize_fwd": [vp, i64, i64, f32, vp, vp, vp],
    "vita_l2_normalize_bwd": [vp, vp, vp, i64, i64, vp, vp],
    "vita_sgemm_f32": [vp, vp, vp, i64, i64, i64, i64, i64, i64, i32, i32, f32, f32, vp],
    "vita_ce_rows": [vp, i64, i64, i64, i64, f32, vp, vp, vp, i64, vp],
    "vita_sigmoid_mse_patch": [vp, i64, vp, i64, i64, i64, i64, i64, i64, i64, i64, vp, f32, vp, vp, vp, i64, vp],
    "vita_attn_supported": [i64, i64],
    "vita_attn_fwd_ws_bytes": [i64],
    "vita_attn_fwd": [vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, f32, vp],
    "vita_attn_bwd_ws_bytes": [i64, i64, i64, i64],
    "vita_attn_bwd": [vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, f32, vp],
    "vita_tabular_embed_fwd": [vp, i64, i64, vp, vp, vp, vp, vp, vp, i64, i64, vp],
    "vita_tabular_embed_bwd": [vp, i64, vp, i64, i64, vp, vp, vp, vp, i64, vp],
    "vita_token_mean_fwd": [vp, i64, i64, i64, i64, vp, vp],
    "vita_token_mean_bwd": [vp, i64, i64, i64, vp, i64, vp],
    "vita_batchnorm_fwd": [vp, i64, i64, vp, vp, f32, i32, vp, vp, vp, vp, vp, f32, vp],
    "vita_batchnorm_eval": [vp, i64, i64, vp, vp, f32, i32, vp, vp, vp, vp],
    "vita_batchnorm_bwd": [vp, vp, vp, i64, i64, vp, vp, vp, i32, vp, vp, vp, vp],
    "vita_add_bias_rows": [vp, i64, i64, vp, vp],
    "vita_colsum_small_f32": [vp, i64, i64, vp, vp],
    "vita_adam_step": [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, f32, i64, vp],
    "vita_cine_partial_floats": [i64],
    "vita_cine_prepare": [vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, vp, i64, i32, vp, vp, vp, vp],
}
_I64_RET = {"vita_layernorm_bwd_nblk", "vita_cine_partial_floats", "vita_attn_bwd_ws_bytes", "vita_attn_fwd_ws_bytes"}
OPTIONAL = set()

_lib = None


def lib():
    """Load (once) and return the shared library; raises RuntimeError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"vita_amd: {LIB_PATH} is missing - the HIP extension is not built "
            "(run __graft_entry__.build()).  There is no CPU fallback.")
    l = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(l, name)
        except AttributeError:
            if name in OPTIONAL:
                continue
            raise RuntimeError(f"vita_amd: symbol {name} missing from {LIB_PATH}")
        fn.argtypes = argtypes
        fn.restype = i64 if name in _I64_RET else C.c_int
    l.vita_last_error.restype = C.c_char_p
    l.vita_last_error.argtypes = []
    # Diagnostic builds (timing-only dissection switches ATT_DBG / G2_DBG, phase stamps ATT_STAMP) export these markers: such a
    # library computes WRONG results by design (or carries stamp overhead) and must never be mistaken for the product build
    for marker in ("vita_debug_build_attn", "vita_debug_build_gemm"):
        if hasattr(l, marker):
            flags = int(getattr(l, marker)())
            if flags:
                import warnings
                warnings.warn(f"vita_amd: {LIB_PATH} is a DIAGNOSTIC build ({marker} = {flags}): timing only, results may be INVALID",
                              RuntimeWarning, stacklevel=2)
    _lib = l
    return l


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().vita_last_error().decode()
        raise RuntimeError(f"vita_amd: {what} failed with code {rc}: {msg}")
