"""ctypes binding of the C-ABI declared in include/nr_engine.h (libnr_engine.so).

The library is built in-tree by ``__graft_entry__.build()`` /
``news_recommendation_amd/csrc/build.sh`` (hipcc, gfx950).  There is NO CPU
fallback: if the shared object is missing or fails to load, importing callers
get a RuntimeError.
"""
import ctypes
import os
from ctypes import c_void_p, c_int, c_int64, c_uint64, c_float, c_double, c_char_p

_P = c_void_p
SIGNATURES = {
    'nr_version': ([], c_int),
    'nr_last_error': ([], c_char_p),
    'nr_supported_seq_len': ([c_int], c_int),
    'nr_gather_rows_f32': ([_P, _P, _P, c_int64, c_int, c_int64, _P], c_int),
    'nr_pack_qkv': ([_P, _P, _P, _P, _P, _P, _P, _P, _P], c_int),
    'nr_pack_additive': ([_P, _P, _P, c_int, _P, _P, _P, _P], c_int),
    'nr_mhsa_fwd': ([_P, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int, c_float, c_uint64, _P], c_int),
    'nr_mhsa_fwd_ex': ([_P, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int, c_float, c_uint64, _P], c_int),
    'nr_mhsa_fwd_len': ([_P, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int, c_float, c_uint64, _P], c_int),
    'nr_attn_bwd_len': ([_P, _P, _P, _P, c_int, _P, _P, _P, _P, c_int64, c_int, c_float, c_uint64, _P], c_int),
    'nr_pack_qkv32': ([_P, _P, _P, _P, _P, _P, _P, _P, _P], c_int),
    'nr_pack_encoder': ([_P] * 9 + [c_int] + [_P] * 9 + [_P], c_int),
    'nr_qkv_proj_fwd': ([_P, _P, c_int64, _P, _P, _P, _P, c_int64, c_int, c_float, c_uint64, _P], c_int),
    'nr_pack_qkv_dx': ([_P, _P, _P, _P, _P], c_int),
    'nr_dx_gemm': ([_P, _P, _P, c_int64, _P], c_int),
    'nr_tn_gemm_parts': ([c_int, c_int64], c_int),
    'nr_tn_gemm': ([_P, c_int, c_int, _P, _P, _P, c_int64, c_int, _P], c_int),
    'nr_attn_fwd': ([_P, _P, _P, c_int64, c_int, c_float, c_uint64, _P], c_int),
    'nr_attn_pool_fwd': ([_P, _P, _P, _P, _P, _P, _P, c_int64, _P, c_int64, c_int, c_int, c_float, c_uint64, _P], c_int),
    'nr_attn_bwd_hm': ([_P, _P, c_int, _P, _P, _P, _P, c_int64, c_int, c_float, c_uint64, _P], c_int),
    'nr_additive_fwd_v': ([_P, _P, _P, _P, _P, c_int64, _P, c_int64, _P, c_int64, c_int, c_int, _P], c_int),
    'nr_additive_fwd_flat': ([_P, _P, _P, _P, _P, c_int64, _P, c_int64, _P, c_int64, c_int, c_int, c_int, _P], c_int),
    'nr_attn_bwd': ([_P, _P, _P, _P, c_int, _P, _P, _P, c_int64, c_int, c_float, c_uint64, _P], c_int),
    'nr_additive_bwd_grid': ([c_int64, c_int], c_int64),
    'nr_additive_bwd': ([_P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int, _P], c_int),
    'nr_pack_additive_t': ([_P, c_int, _P, _P], c_int),
    'nr_wgrad_unpack': ([_P, c_int, _P, c_int, _P, c_int64, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P], c_int),
    'nr_additive_bwd_act': ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_float, c_int64, c_int, _P], c_int),
    'nr_additive_bwd_flat_grid': ([c_int64], c_int64),
    'nr_debug_pool3_stamps': ([_P], c_int),
    'nr_debug_attnb_stamps': ([_P], c_int),
    'nr_debug_xcd_probe': ([_P, _P, _P, c_int, _P], c_int),
    'nr_gru_persist_status': ([_P, _P], c_int),
    'nr_set_fault_words': ([_P], c_int),
    'nr_fault_state': ([_P], c_int),
    'nr_fault_clear': ([], c_int),
    'nr_debug_gru_fault': ([c_int, c_int], c_int),
    'nr_gru_persist_enabled': ([c_int, c_int, c_int], c_int),
    'nr_debug_gru_stamps': ([_P], c_int),
    'nr_debug_gru_stamps_bwd': ([_P], c_int),
    'nr_additive_bwd_flat': ([_P, _P, _P, _P, _P, _P, _P, c_int64, _P, _P, _P, _P, _P, c_float, c_int64, c_int, c_int, _P], c_int),
    'nr_additive_bwd_flat_gs': ([_P, _P, _P, _P, _P, _P, c_int64, _P, c_int64, _P, _P, _P, _P, _P, c_float, c_int64, c_int, c_int, _P], c_int),
    'nr_additive_bwd_ex': ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int, _P], c_int),
    'nr_gather_bf16': ([_P, _P, c_int64, _P, _P, c_int64, c_float, c_uint64, _P], c_int),
    'nr_embed_scatter_add': ([_P, _P, c_int, _P, c_int64, c_int64, c_float, c_uint64, _P], c_int),
    'nr_embed_scatter_sorted': ([_P, _P, _P, c_int, _P, c_int64, c_int64, c_float, c_uint64, _P], c_int),
    'nr_score_dot_bwd': ([_P, _P, _P, _P, _P, c_int64, c_int, c_int, _P], c_int),
    'nr_additive_fwd': ([_P, _P, _P, _P, _P, _P, c_int64, c_int, _P], c_int),
    'nr_score_dot': ([_P, _P, _P, c_int64, c_int, c_int, _P], c_int),
    'nr_score_ce_fwd': ([_P, _P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, _P], c_int),
    'nr_score_ce_bwd': ([_P, _P, _P, _P, _P, c_int64, _P, c_int64, c_int64, c_int, c_int, _P], c_int),
    'nr_accum_many': ([_P, c_int, _P], c_int),
    'nr_rows_to_f32': ([_P, c_int64, c_int, _P, c_int64, c_int64, _P], c_int),
    'nr_score_csr': ([_P, _P, _P, _P, _P, _P, c_int64, c_int64, c_int, _P], c_int),
    'nr_supported_pool_len': ([c_int], c_int),
    'nr_supported_conv_len': ([c_int], c_int),
    'nr_pack_conv': ([_P, _P, c_int, c_int, _P, _P, _P, _P], c_int),
    'nr_conv3_fwd': ([_P, _P, c_int64, _P, _P, _P, _P, c_int64, c_int, c_float, c_uint64, c_int64, _P], c_int),
    'nr_conv3_fwd_v': ([_P, _P, c_int64, _P, _P, _P, _P, c_int64, c_int, c_int, c_float, c_uint64, c_int64, _P], c_int),
    'nr_conv3_dgrad': ([_P, _P, _P, c_int64, c_int, _P], c_int),
    'nr_conv_act_bwd': ([_P, _P, c_int, _P, _P, c_int64, _P, c_int64, c_int, c_float, _P], c_int),
    'nr_additive_fwd_ex': ([_P, _P, _P, _P, _P, c_int64, _P, c_int64, _P, c_int64, c_int, _P], c_int),
    'nr_additive_dx': ([_P, c_int, _P, _P, _P, c_int64, c_int, c_int, _P], c_int),
    'nr_element_table_fwd': ([_P, c_int, c_int, _P, _P, _P, _P, _P, _P], c_int),
    'nr_element_table_bwd': ([_P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P], c_int),
    'nr_views_fill': ([_P, _P, _P, c_int, _P, c_int64, _P], c_int),
    'nr_scatter_sorted_f32': ([_P, _P, _P, c_int64, _P, c_int64, c_int64, c_int, _P], c_int),
    'nr_rows_scatter_add': ([_P, _P, c_int64, _P, _P, c_int64, c_int, c_int64, c_int, _P], c_int),
    'nr_gather_rows_strided': ([_P, _P, c_int64, c_int, _P, _P, c_int64, c_int64, _P], c_int),
    'nr_gru_dims': ([c_int, _P, _P, _P], c_int),
    'nr_pack_gru': ([_P, c_int, c_int, c_int, _P, _P, c_int, _P], c_int),
    'nr_tile_rows_bf16': ([_P, c_int, c_int, _P, _P], c_int),
    'nr_rows_to_bf16': ([_P, c_int64, c_int, _P, c_int, c_int64, _P], c_int),
    'nr_gru_fwd_step': ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P], c_int),
    'nr_gru_fwd_seq': ([_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P], c_int),
    'nr_gru_bwd_seq': ([_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P], c_int),
    'nr_gru_seq_buffers': ([c_int, c_int, c_int], c_int),
    'nr_gru_fwd_seq_n': ([_P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, _P], c_int),
    'nr_gru_fwd_seq_rows': ([_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P], c_int),
    'nr_gru_gate_rows': ([_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P], c_int),
    'nr_gru_bwd_seq_n': ([_P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, _P], c_int),
    'nr_gru_bwd_step': ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P], c_int),
    'nr_impression_metrics': ([_P, _P, _P, _P, c_int64, _P], c_int),
    'nr_adam_flat': ([_P, _P, _P, _P, c_int64, _P, c_int64, c_double, c_double, c_double, c_float, c_int, _P], c_int),
    'nr_row_adam_catchup': ([_P, c_int64, _P, _P, _P, _P, c_int64, c_int, _P, c_int64, c_double, c_double, c_double, _P], c_int),
    'nr_row_adam_catchup_ex': ([_P, c_int64, _P, _P, _P, _P, c_int64, c_int, _P, c_int64, c_int, c_double, c_double, c_double, _P], c_int),
    'nr_row_adam_flush': ([_P, _P, _P, _P, c_int64, c_int, _P, c_int64, c_double, c_double, c_double, _P], c_int),
    'nr_row_adam_step': ([_P, _P, c_int64, _P, c_int64, _P, _P, _P, _P, c_int64, c_int, _P, c_int64, c_double, c_double, c_double,
                          c_float, c_int, _P], c_int),
    'nr_pack_conv_dgrad': ([_P, c_int, c_int, _P, _P], c_int),
    'nr_pack_conv_fwd2': ([_P, c_int, c_int, _P, _P], c_int),
    'nr_conv3_fwd_gemm': ([_P, _P, c_int64, _P, _P, _P, _P, c_int64, c_int, c_int, c_float, c_uint64, c_int64, _P], c_int),
    'nr_conv3_dgrad_gemm': ([_P, _P, _P, c_int64, c_int, _P], c_int),
    'nr_gemm_nt': ([_P, c_int64, _P, c_int64, _P, c_int64, c_int64, c_int, c_int, _P], c_int),
    'nr_gemm_tn_parts': ([c_int, c_int, c_int64], c_int),
    'nr_gemm_tn': ([_P, c_int64, c_int, _P, c_int64, c_int, c_int, _P, _P, c_int64, c_int64, c_int, _P], c_int),
    'nr_transpose_bf16': ([_P, c_int, c_int, c_int64, _P, c_int64, _P], c_int),
    'nr_sum_parts': ([_P, c_int, c_int64, _P, c_int, _P], c_int),
    'nr_sort_ids_workspace': ([c_int64, c_int64], c_int64),
    'nr_sort_ids': ([_P, c_int64, c_int64, _P, _P, _P, c_int64, _P], c_int),
    'nr_dropout_mask': ([_P, c_int64, c_float, c_uint64, c_int, _P], c_int),
    'nr_set_step_counter': ([_P], c_int),
    'nr_step_counter_add': ([_P, ctypes.c_uint32, _P], c_int),
    'nr_g_dropout': ([_P, _P, c_int64, c_int64, c_float, c_uint64, c_int, _P], c_int),
    'nr_g_attn_fwd': ([_P, c_int64, _P, _P, c_int64, c_int, c_int, c_int, _P], c_int),
    'nr_g_attn_bwd': ([_P, c_int64, _P, _P, _P, c_int64, c_int, c_int, c_int, _P], c_int),
    'nr_g_additive_fwd': ([_P, c_int64, c_int, _P, c_int64, c_int, _P, _P, c_int64, _P, c_int64, c_int, c_int, _P], c_int),
    'nr_g_additive_bwd': ([_P, c_int64, c_int, _P, c_int64, c_int, _P, _P, _P, c_int64, _P, c_int64, _P, c_int64, c_int, _P], c_int),
    'nr_g_rows_axpy': ([_P, c_int64, _P, _P, c_int64, c_int, c_int, c_int64, c_int, _P], c_int),
    'nr_g_rows_to_seqpad': ([_P, c_int64, c_int, _P, c_int, c_int, c_int, c_int64, c_int, _P], c_int),
    'nr_g_relu_drop': ([_P, c_int64, _P, c_int, c_int, c_int, c_int64, c_float, c_uint64, c_int64, _P], c_int),
    'nr_g_relu_drop_bwd': ([_P, _P, _P, c_int, c_int, c_int, c_int, c_int64, c_float, _P], c_int),
    'nr_g_unpad_rows': ([_P, c_int64, _P, c_int, c_int, c_int, c_int64, _P], c_int),
    'nr_g_relu': ([_P, _P, _P, c_int64, c_float, _P], c_int),
    'nr_g_rows_split_bf16': ([_P, c_int64, c_int, _P, c_int, c_int64, _P], c_int),
    'nr_gemm_nt_rows': ([_P, c_int64, _P, c_int64, _P, c_int64, c_int64, c_int, c_int, _P], c_int),
    'nr_probe_mfma': ([_P, _P, _P, _P], c_int),
    'nr_probe_tr16': ([_P, _P, _P], c_int),
}

# layout constants mirrored from include/nr_engine.h
NR_D, NR_KP, NR_HEADS, NR_DK, NR_NP, NR_QP = 300, 320, 15, 20, 320, 208
NR_LDG = 3 * NR_KP
NR_QKV_HM_SEQ = NR_HEADS * 3 * 20 * NR_DK      # elements per sequence of the head-major Q | K | V^T saves
NR_K16 = 19

LIB_NAME = 'libnr_engine.so'
_lib = None


def bind(lib):
    """Attach argtypes/restype for every exported symbol (raises AttributeError if one is missing)."""
    for name, (argtypes, restype) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    return lib


def lib_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)


def load():
    """Load (once) the in-tree HIP library; fail loudly when it is absent."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        try:
            _lib = bind(ctypes.CDLL(path))
        except OSError as e:  # pragma: no cover
            raise RuntimeError(f"cannot load {path}: {e}. There is no CPU fallback.") from e
    return _lib


class EngineError(RuntimeError):
    pass


def check(lib, rc):
    if rc != 0:
        msg = lib.nr_last_error()
        raise EngineError(f"nr_engine error {rc}: {msg.decode() if msg else '?'}")
