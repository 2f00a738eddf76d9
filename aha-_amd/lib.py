"""ctypes binding of libaha_amd.so (the C ABI in include/aha_amd.h).

There is NO fallback: if the shared library is missing or a symbol is absent, importing this
module raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C aha-_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AHA_AMD_LIB") or os.path.join(_HERE, "libaha_amd.so")     # AHA_AMD_LIB: another build of the same ABI (A/B timing of kernel variants)


class ModelDesc(C.Structure):
    _fields_ = [
        ("image_size", C.c_int32), ("patch_size", C.c_int32), ("v_hidden", C.c_int32), ("v_layers", C.c_int32),
        ("v_heads", C.c_int32), ("v_inter", C.c_int32), ("v_ln_eps", C.c_float),
        ("hidden", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32), ("kv_heads", C.c_int32),
        ("head_dim", C.c_int32), ("inter", C.c_int32), ("vocab", C.c_int32),
        ("rope_theta", C.c_float), ("rms_eps", C.c_float),
        ("max_positions", C.c_int32), ("pool_stride", C.c_int32), ("pool_mode", C.c_int32),
        ("max_step_tokens", C.c_int32), ("max_vit_frames", C.c_int32), ("v_kind", C.c_int32),
    ]


class TensorView(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("shape", C.c_int64 * 4), ("ndim", C.c_int32),
                ("reserved", C.c_int32)]


CACHE_NONE, CACHE_SINK, CACHE_SLIDING, CACHE_STATIC = 0, 1, 2, 3
ATTN_TRAILING, ATTN_HF449_SDPA, ATTN_FA2 = 0, 1, 2

# every symbol include/aha_amd.h declares: (name, restype, argtypes)
_P, _I, _F = C.c_void_p, C.c_int, C.c_float
SYMBOLS = [
    ("aha_ctx_create", _I, [C.POINTER(ModelDesc), _I, C.POINTER(_P)]),
    ("aha_ctx_load_weights", _I, [_P, C.POINTER(TensorView), C.c_size_t, _P]),
    ("aha_ctx_set_rope_table", _I, [_P, _P, _P, _I, _P]),
    ("aha_ctx_set_rerotation_table", _I, [_P, _I, _I, _I, _P, _P, _P]),
    ("aha_ctx_has_rerotation_table", _I, [_P, _I, _I, _I]),
    ("aha_ctx_set_tuning", _I, [_P, C.c_char_p, _I]),
    ("aha_ctx_destroy", None, [_P]),
    ("aha_last_error", C.c_char_p, [_P]),
    ("aha_vit_encode", _I, [_P, _P, _I, _P, _P]),
    ("aha_vit_encode_pooled_first", _I, [_P, _P, _I, _I, _P, _P]),
    ("aha_vit_encode_live", _I, [_P, _P, _I, _I, _I, _P, _P]),
    ("aha_vit_last_tower_output", _I, [_P, _I, _P, _P]),
    ("aha_frame_ingest", _I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    ("aha_embed_tokens", _I, [_P, _P, _I, _P, _P]),
    ("aha_stream_open", _I, [_P, _I, _I, _I, _I, C.POINTER(_P)]),
    ("aha_stream_reset", _I, [_P]),
    ("aha_stream_seq_length", _I, [_P]),
    ("aha_stream_seen_tokens", _I, [_P]),
    ("aha_stream_set_attn_semantics", _I, [_P, _I]),
    ("aha_stream_set_position_offset", _I, [_P, _I]),
    ("aha_stream_export_kv", _I, [_P, _P, _I, _I, _P, _P]),
    ("aha_stream_destroy", None, [_P]),
    ("aha_lm_step", _I, [_P, C.POINTER(_P), _I, _P, _I, _P, _P, _P, _P]),
    ("aha_lm_heads_all", _I, [_P, _P, _P]),
    ("aha_lm_last_hidden_all", _I, [_P, _P, _P]),
    ("aha_lm_logits_last", _I, [_P, _P, _P, _P]),
    ("aha_lm_last_step_work", _I, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("aha_lm_last_gemm_time", _I, [_P, _I, C.POINTER(_F), C.POINTER(_I), C.POINTER(C.c_double)]),
    ("aha_version", C.c_char_p, []),
    ("aha_lm_logits_all", _I, [_P, _P, _P]),
    ("aha_generate_greedy", _I, [_P, _P, _P, _I, _I, C.c_int64, _F, _P, _I, C.POINTER(_I), C.POINTER(C.c_int64), C.POINTER(_I), _P]),
    ("aha_generate_greedy_cb", _I, [_P, _P, _P, _I, _I, C.c_int64, _F, _P, _I, C.POINTER(_I), C.POINTER(C.c_int64), C.POINTER(_I), _P, _P, _P]),
    ("aha_linear_create", _I, [_P, _P, _P, _I, _I, C.POINTER(_P), _P]),
    ("aha_linear_destroy", None, [_P]),
    ("aha_linear_split_k", _I, [_P, _P, _I]),
    ("aha_linear_forward", _I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _I, _P]),
    ("aha_linear_tile_forward", _I, [_P, _P, _I, _I, _P, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P]),
    ("aha_rmsnorm_forward", _I, [_P, _P, _I, _P, _P, _I, _I, _I, _F, _P]),
    ("aha_resid_rmsnorm_forward", _I, [_P, _P, _I, _P, _P, _P, _I, _I, _F, _P]),
    ("aha_heads_forward", _I, [_P, _P, _I, _I, _P, _P, _P]),
    ("aha_cache_update", _I, [_P, _P, _I, _P, _P, _I, _P, _P, _P]),
    ("aha_attention_forward", _I, [_P, C.POINTER(_P), _I, _P, _I, _I, C.POINTER(_I), _I, _P, _P]),
    ("aha_vit_attention_forward", _I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    ("aha_vit_layers_forward", _I, [_P, _P, _I, _I, _I, _P, _P]),
    ("aha_layernorm_forward", _I, [_P, _P, _I, _P, _P, _P, _I, _I, _I, _F, _P]),
    ("aha_vit_patchify_forward", _I, [_P, _P, _I, _P, C.POINTER(_I), _P]),
    ("aha_pool_forward", _I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    ("aha_pool_gather_rows_forward", _I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    ("aha_lm_debug_tap", _I, [_P, _I, _P, _P]),
    ("aha_lm_engine_stamps", _I, [_P, _P]),
    ("aha_comm_unique_id", _I, [_P, C.c_size_t]),
    ("aha_comm_init_rank", _I, [_P, C.c_size_t, _I, _I, _I, C.POINTER(_P)]),
    ("aha_comm_size", _I, [_P]),
    ("aha_comm_rank", _I, [_P]),
    ("aha_allgather_scores", _I, [_P, _P, _I, _P, _P]),
    ("aha_comm_destroy", None, [_P]),
    ("aha_comm_last_error", C.c_char_p, []),
]
EPI_SPLITK_F32, EPI_BF16, EPI_SWIGLU, EPI_F32 = 0, 1, 2, 3
ACT_NONE, ACT_GELU_TANH, ACT_GELU_ERF, ACT_QUICK_GELU = 0, 1, 2, 3
COMM_ID_BYTES = 128
TOKEN_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int)      # aha_token_cb


def load(path: str = LIB_PATH) -> C.CDLL:
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: the HIP extension is not built. There is no CPU fallback; run "
            f"`make -C {os.path.join(_HERE, 'csrc')}` (or __graft_entry__.build()).")
    lib = C.CDLL(path)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)            # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None


def get() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = load()
    return _lib
