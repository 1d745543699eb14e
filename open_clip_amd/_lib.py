"""ctypes binding of ``libopenclip_hip.so`` (C ABI: include/openclip_hip.h).

The product path has NO fallback: if the shared library is missing ``load()`` raises, and every op in
``open_clip_amd.ops`` goes through ``call()``.  ``SIGNATURES`` mirrors the header one-to-one
(tests/test_cabi.py checks header <-> table <-> exported symbols).
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OCN_LIB_PATH") or os.path.join(_HERE, "libopenclip_hip.so")  # OCN_LIB_PATH: developer builds (A/B of compile-time experiments)

_p, _i, _f, _l = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64

# name -> argtypes (every function returns int status unless listed in _SPECIAL)
SIGNATURES = {
    "ocn_gemm_nt": [_i, _p, _i, _p, _i, _p, _i, _i, _i, _i, _p, _p, _p, _f, _p],
    "ocn_gemm_tn_accum": [_p, _i, _p, _i, _p, _i, _i, _i, _i, _p, _f, _p],
    "ocn_gemm_tn_accum_det": [_p, _i, _p, _i, _p, _i, _i, _i, _i, _p, _f, _p, _l, _p],
    "ocn_gemm_tn_accum2": [_p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _f, _p],
    "ocn_gemm_nt_splitk": [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _p, _p, _p, _i, _f, _p, _p],
    "ocn_scale_rows_bf16": [_p, _i, _p, _p, _i, _i, _i, _p],
    "ocn_sub_scaled_rows": [_p, _i, _p, _i, _p, _f, _i, _i, _p],
    "ocn_cast_f32_bf16": [_p, _p, _l, _p],
    "ocn_cast_f32_bf16_scaled": [_p, _p, _l, _p, _p],
    "ocn_cast_transpose_f32_bf16": [_p, _p, _i, _i, _p],
    "ocn_layernorm_fwd": [_p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p],
    "ocn_layernorm_bwd": [_p, _i, _p, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _p],
    "ocn_colsum_f32": [_p, _p, _i, _i, _i, _p],
    "ocn_attn_fwd": [_p, _p, _p, _i, _i, _i, _i, _f, _p],
    "ocn_attn_bwd": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "ocn_attn_fwd_hd": [_p, _p, _p, _i, _i, _i, _i, _i, _f, _p],
    "ocn_attn_bwd_hd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p],
    "ocn_attn_fwd_varlen": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "ocn_attn_bwd_varlen": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "ocn_attn_pooled_fwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "ocn_attn_pooled_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "ocn_patchify": [_p, _i, _p, _i, _i, _i, _i, _i, _p],
    "ocn_patchify_u8": [_p, _i, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), _p, _i, _i, _i, _i, _i, _p],
    "ocn_embed_assemble_fwd": [_p, _p, _p, _p, _i, _i, _i, _p],
    "ocn_embed_assemble_bwd": [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    "ocn_token_embed_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    "ocn_token_embed_bwd": [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    "ocn_token_embed_bwd_sorted": [_p, _p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p],
    "ocn_seq_pack_plan": [_p, _p, _p, _p, _i, _i, _p],
    "ocn_seq_pack_rows": [_p, _p, _p, _p, _i, _i, _p],
    "ocn_token_range_check": [_p, _l, _i, _p, _p],
    "ocn_seq_bucket_plan": [_p, _p, _p, _i, _i, _p],
    "ocn_token_embed_fwd_rows": [_p, _p, _p, _p, _p, _l, _i, _i, _p],
    "ocn_token_embed_bwd_sorted_varlen": [_p, _p, _p, _i, _p, _p, _p, _i, _i, _l, _i, _i, _i, _p],
    "ocn_argmax_rows": [_p, _p, _i, _i, _p],
    "ocn_gather_rows": [_p, _i, _p, _p, _i, _i, _i, _p],
    "ocn_gather_rows_bf16": [_p, _p, _p, _i, _i, _i, _p],
    "ocn_scatter_rows": [_p, _p, _p, _p, _i, _i, _i, _p],
    "ocn_scatter_add_rows": [_p, _p, _p, _p, _i, _i, _i, _p],
    "ocn_l2norm_fwd": [_p, _p, _p, _p, _i, _i, _f, _p],
    "ocn_l2norm_bwd": [_p, _p, _p, _p, _i, _i, _p],
    "ocn_softmax_ce_rows": [_p, _i, _p, _i, _i, _i, _i, _f, _f, _f, _p, _p, _p, _p],
    "ocn_fused_logits_ce": [_p, _i, _p, _i, _i, _i, _i, _i, _f, _f, _p, _i, _p, _p, _p, _p, _p],
    "ocn_siglip_rows": [_p, _i, _p, _i, _i, _i, _i, _i, _f, _p, _f, _f, _f, _p, _p, _p, _p, _p],
    "ocn_sumsq_accum": [_p, _l, _p, _p],
    "ocn_adamw_step": [_p, _p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _i, _p, _p],
    "ocn_adamw_multi": [_p, _p, _i, _f, _f, _f, _p, _f, _p],
    "ocn_sumsq_multi": [_p, _p, _i, _p, _p, _p],
    "ocn_comm_unique_id": [_p],
    "ocn_comm_init": [_p, _i, _i, _p],
    "ocn_comm_destroy": [_p],
    "ocn_comm_allgather": [_p, _p, _p, _l, _i, _p],
    "ocn_comm_reduce_scatter_sum": [_p, _p, _p, _l, _i, _p],
    "ocn_comm_allreduce_sum": [_p, _p, _l, _i, _p],
    "ocn_comm_allreduce_avg": [_p, _p, _l, _i, _p],
    "ocn_comm_broadcast": [_p, _p, _l, _i, _i, _p],
    "ocn_comm_count": [_p, _p, _p],
    "ocn_set_tile_rescue": [_i],
    "ocn_comm_sendrecv": [_p, _p, _i, _p, _i, _l, _i, _p],
    "ocn_probe_mfma32": [_p, _p, _p, _p],
    "ocn_probe_tr16": [_p, _p, _p],
}
# developer entry points (include/openclip_hip_debug.h): kernel selection / ablation knobs for tools/ and a few tests; never called by
# the product modules
DEBUG_SIGNATURES = {
    "ocn_set_gemm_variant": [_i],
    "ocn_set_tuning": [_i, _i],
    "ocn_debug_occupy": [_i, _i, _p, _p],
    "ocn_debug_nt5_trace": [_p],
    "ocn_debug_stream_with_cu_mask": [_p, _i, _p],
}
_SPECIAL = {"ocn_last_error": ([], ctypes.c_char_p), "ocn_version": ([], _i), "ocn_gemm_tn_det_workspace_bytes": ([_i, _i, _i], _l),
            "ocn_fused_logits_ce_workspace_floats": ([_i, _i], _l), "ocn_gemm_nt_splitk_plan": ([_i, _i, _i], _i), "ocn_layernorm_bwd_det_workspace_floats": ([_i, _i], _l), "ocn_get_tile_rescue": ([], _i)}

ABI_VERSION = 104  # == OCN_ABI_VERSION of include/openclip_hip.h (tests/test_cabi.py compares the two): load() refuses any other library

_lib = None
_lock = threading.Lock()


def load():
    """Loads the shared library (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is required (no fallback path). "
                "Build it with `python -m open_clip_amd.build` (needs hipcc, cross-compiles for gfx950)."
            )
        import torch  # noqa: F401  -- BEFORE the CDLL: the library must bind to the HIP runtime torch has loaded (its bundled
        #                 libamdhip64), not pull a second copy from /opt/rocm that knows no device ("no ROCm-capable device")
        lib = ctypes.CDLL(LIB_PATH)
        lib.ocn_version.argtypes, lib.ocn_version.restype = [], _i
        if lib.ocn_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} reports C-ABI version {lib.ocn_version()}, this package binds version {ABI_VERSION} "
                               "(include/openclip_hip.h): rebuild it with `python -m open_clip_amd.build` -- argument lists differ between versions")
        for name, argtypes in list(SIGNATURES.items()) + list(DEBUG_SIGNATURES.items()):
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = _i
        for name, (argtypes, restype) in _SPECIAL.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = restype
        _lib = lib
    return _lib


def call(name, *args):
    """Calls ``name`` and raises RuntimeError(ocn_last_error()) on a non-zero status
    (the reference's convention is Python exceptions / asserts, e.g. factory.py:416, loss.py:37)."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib.ocn_last_error().decode()}")
