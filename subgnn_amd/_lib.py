"""ctypes loader for libsubgnn_hip.so -- the only compute backend of this package.

There is deliberately NO fallback: if the HIP library is missing or a kernel reports an
error, the call raises.  (The CPU oracle under ``oracle/`` is test infrastructure and is
never imported from here.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libsubgnn_hip.so')

c_i64 = ctypes.c_int64
c_u64 = ctypes.c_uint64
c_int = ctypes.c_int
c_dbl = ctypes.c_double
c_ptr = ctypes.c_void_p


class MpnArgs(ctypes.Structure):
    """struct sgnn_mpn_args (include/subgnn_hip.h)."""
    _fields_ = [('src', ctypes.c_int32), ('sims_per_edge', ctypes.c_int32),
                ('R', c_i64), ('A', c_i64), ('D', c_i64),
                ('x', c_ptr), ('ids', c_ptr), ('id_div', c_i64), ('edge_mask', c_ptr), ('row_mask', c_ptr),
                ('sims', c_ptr), ('sims_ld', c_i64), ('sim_col', c_ptr), ('wp', c_ptr), ('bp', c_ptr),
                ('x_f16', ctypes.c_int32), ('z_act', c_ptr), ('flags', ctypes.c_int32)]


# name -> (restype, argtypes); mirrors include/subgnn_hip.h line by line
SIGNATURES = {
    'sgnn_abi_version': (c_int, []),
    'sgnn_warm_up': (c_int, [c_ptr]),
    'sgnn_last_error': (ctypes.c_char_p, []),
    'sgnn_degree_sequence': (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr,
                                     c_ptr, c_ptr]),
    'sgnn_degree_sequence_sorted_rows': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int,
                                                 c_ptr, c_ptr, c_ptr, c_ptr]),
    'sgnn_degree_sequence_hub_bitmaps': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_int,
                                                 c_ptr, c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sgnn_degree_sequence_search_threshold': (c_i64, []),
    'sgnn_cc_embed_fwd_f16': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_i64, c_ptr, c_ptr, c_ptr]),
    'sgnn_cc_labels': (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    'sgnn_cc_compact_stats': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr]),
    'sgnn_cc_compact': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_ptr]),
    'sgnn_cc_huge_workspace_bytes': (c_i64, [c_i64]),
    'sgnn_degree_sequence_huge_workspace_bytes': (c_i64, [c_i64]),
    'sgnn_degree_sequence_huge': (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_ptr]),
    'sgnn_cc_labels_huge': (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_i64, c_ptr]),
    'sgnn_cc_compact_huge': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr]),
    'sgnn_patch_in_border_huge_workspace_bytes': (c_i64, [c_i64]),
    'sgnn_patch_in_border_huge': (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_i64, c_ptr]),
    'sgnn_sort_sets': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr]),
    'sgnn_khop_border_workspace_bytes': (c_i64, [c_i64, c_i64, c_int]),
    'sgnn_khop_border_bitmap_fits_lds': (c_int, [c_i64]),
    'sgnn_khop_border': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr, c_ptr,
                                 c_ptr, c_ptr, c_i64, c_int, c_ptr]),
    'sgnn_khop_border_arena': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr,
                                       c_i64, c_int, c_ptr]),
    'sgnn_khop_border_sample_workspace_bytes': (c_i64, [c_i64, c_i64, c_int, c_int, c_int]),
    'sgnn_khop_border_sample': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_i64, c_int, c_i64, c_u64, c_u64, c_i64,
                                        c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_ptr]),
    'sgnn_sample_anchors_padded': (c_int, [c_ptr, c_i64, c_i64, c_i64, c_u64, c_u64, c_ptr, c_ptr]),
    'sgnn_sample_anchors_ragged': (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_u64, c_u64, c_i64, c_ptr, c_ptr]),
    'sgnn_choice_ragged': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_u64, c_u64, c_i64, c_ptr, c_ptr]),
    'sgnn_triangular_walks_both': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_dbl, c_u64,
                                           c_u64, c_u64, c_i64, c_i64, c_ptr, c_ptr]),
    'sgnn_triangular_walks': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_int,
                                      c_i64, c_i64, c_i64, c_dbl, c_u64, c_u64, c_i64, c_i64, c_int, c_ptr, c_ptr]),
    'sgnn_patch_in_border': (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    'sgnn_sp_similarity_dense': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    'sgnn_bfs_hops_workspace_bytes': (c_i64, [c_i64, c_i64, c_int]),
    'sgnn_bfs_min_hops_workspace_bytes': (c_i64, [c_i64, c_i64, c_int, c_i64]),
    'sgnn_bfs_min_hops_to_sets': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_ptr, c_i64, c_ptr, c_ptr,
                                          c_ptr, c_i64, c_ptr]),
    'sgnn_bfs_hops': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_ptr, c_i64, c_ptr]),
    'sgnn_min_hops_to_sets': (c_int, [c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    'sgnn_dtw_workspace_bytes': (c_i64, [c_i64, c_i64, c_i64, c_i64]),
    'sgnn_dtw_similarity': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_ptr, c_ptr,
                                    c_i64, c_ptr]),
    'sgnn_dtw_order_keys': (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    'sgnn_dtw_similarity_live': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_ptr, c_ptr,
                                         c_ptr, c_i64, c_ptr]),
    'sgnn_cc_embed_fwd': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_i64, c_ptr, c_ptr, c_ptr]),
    'sgnn_cc_embed_bwd': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_ptr]),
    'sgnn_mpn_fwd_chunks': (c_int, [ctypes.POINTER(MpnArgs)]),
    'sgnn_mpn_fwd': (c_int, [ctypes.POINTER(MpnArgs), c_ptr, c_ptr, c_ptr]),
    'sgnn_mpn_fwd_many_max_bodies': (c_i64, []),
    'sgnn_mpn_fwd_many': (c_int, [c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sgnn_mpn_bwd': (c_int, [ctypes.POINTER(MpnArgs), c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sgnn_attn_scores_epilogue': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr]),
    'sgnn_attn_scores_f16_workspace_bytes': (c_i64, [c_i64]),
    'sgnn_attn_scores_fwd_f16': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_i64, c_ptr]),
    'sgnn_lstm_supported': (c_int, [c_i64]),
    'sgnn_lstm_fwd': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sgnn_lstm_bwd': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr]),
    'sgnn_masked_sum_fwd': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr]),
    'sgnn_masked_sum_bwd': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr]),
    'sgnn_masked_sum_slot_fwd': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_i64, c_ptr]),
    'sgnn_masked_sum_slot_bwd': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr]),
    'sgnn_masked_sum_slots_fwd': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_i64, c_ptr]),
    'sgnn_masked_sum_slots_bwd': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_ptr]),
    'sgnn_readout_sum_fwd': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_i64, c_ptr]),
    'sgnn_grad_sumsq_partials': (c_i64, []),
    'sgnn_grad_sumsq': (c_int, [c_ptr, c_i64, c_ptr, c_ptr]),
    'sgnn_clip_coefficient': (c_int, [c_ptr, c_i64, c_ptr, c_i64, ctypes.c_float, c_ptr, c_ptr, c_ptr]),
    'sgnn_first_occurrence_mask': (c_int, [c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    'sgnn_filter_sets': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sgnn_pack_rows_count': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    'sgnn_pack_rows_write': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr]),
    'sgnn_pack_fused_max_rows': (c_i64, []),
    'sgnn_pack_fused_max_entries': (c_i64, []),
    'sgnn_pack_rows_fused': (c_int, [c_ptr, c_ptr, c_int, c_i64, c_i64, c_ptr, c_ptr, c_ptr]),
    'sgnn_filter_sets_fused': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr]),
    'sgnn_khop_sample_finish': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    'sgnn_cross_entropy_workspace_bytes': (c_i64, [c_i64]),
    'sgnn_cross_entropy_fwd': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr]),
    'sgnn_cross_entropy_bwd': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    'sgnn_column_sum_workspace_bytes': (c_i64, [c_i64, c_i64]),
    'sgnn_column_sum': (c_int, [c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_i64, c_ptr]),
    'sgnn_readout_sum_bwd_workspace_bytes': (c_i64, [c_i64, c_i64, c_i64]),
    'sgnn_readout_sum_bwd': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr,
                                     c_ptr, c_i64, c_ptr]),
    'sgnn_gather_rows_many_max': (c_i64, []),
    'sgnn_rows_gemm': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr]),
    'sgnn_rows_gemm_nt': (c_int, [c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    'sgnn_lstm_tail_fwd': (c_int, [c_ptr, c_i64, c_i64, c_i64, c_i64, c_int, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr]),
    'sgnn_lstm_tail_bwd': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_int, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sgnn_readout_many_max': (c_i64, []),
    'sgnn_readout_many_fwd': (c_int, [c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64,
                                      c_i64, c_ptr, c_i64, c_ptr]),
    'sgnn_readout_many_bwd_workspace_bytes': (c_i64, [c_i64, c_ptr, c_i64, c_i64]),
    'sgnn_readout_many_bwd': (c_int, [c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                      c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    'sgnn_gather_rows_many': (c_int, [c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    'sgnn_head_supported': (c_int, [c_i64, c_i64, c_i64]),
    'sgnn_head_blocks': (c_i64, [c_i64]),
    'sgnn_head_partial_floats': (c_i64, [c_i64, c_i64, c_i64]),
    'sgnn_head_fwd_workspace_bytes': (c_i64, [c_i64]),
    'sgnn_head_fwd': (c_int, [c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_float, c_ptr, c_ptr, c_ptr,
                              c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr]),
    'sgnn_head_bwd': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64,
                              ctypes.c_float, c_ptr, c_ptr, c_ptr]),
    'sgnn_contract_rows_max_jobs': (c_i64, []),
    'sgnn_contract_rows_blocks': (c_i64, [c_i64, c_i64, c_i64]),
    'sgnn_contract_rows_partial': (c_int, [c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sgnn_reduce_partials_max_jobs': (c_i64, []),
    'sgnn_reduce_partials': (c_int, [c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sgnn_probe_stream_copy': (c_int, [c_ptr, c_ptr, c_i64, c_int, c_ptr]),
    'sgnn_scatter_add_rows_workspace_bytes': (c_i64, [c_i64, c_i64]),
    'sgnn_mpn_bwd_shared_det_workspace_bytes': (c_i64, [c_i64, c_i64, c_i64]),
    'sgnn_mpn_bwd_shared_det': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr]),
    'sgnn_adam_step': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                               c_i64, c_ptr, c_int, c_ptr]),
    'sgnn_adam_step_counted': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                       ctypes.c_float, c_ptr, c_ptr, c_int, c_ptr]),
    'sgnn_optim_partials': (c_i64, [c_ptr, c_i64]),
    'sgnn_optim_sumsq': (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sgnn_optim_count': (c_int, [c_ptr, c_ptr, c_i64, c_ptr]),
    'sgnn_optim_adam': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                ctypes.c_float, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, ctypes.c_float, c_ptr, c_ptr]),
    'sgnn_update_fwd': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    'sgnn_update_fwd_chunks_max_rows': (c_i64, []),
    'sgnn_update_fwd_chunks': (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr]),
    'sgnn_update_many_max_bodies': (c_i64, []),
    'sgnn_update_fwd_many': (c_int, [c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr]),
    'sgnn_update_bwd_many': (c_int, [c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64,
                                     c_ptr]),
    'sgnn_update_bwd_workspace_bytes': (c_i64, [c_i64, c_i64]),
    'sgnn_update_bwd': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr]),
    'sgnn_sort_edges_by_key_workspace_bytes': (c_i64, [c_i64, c_i64]),
    'sgnn_sort_edges_by_key': (c_int, [c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_ptr]),
    'sgnn_scatter_add_rows_sorted': (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                             c_ptr, c_i64, c_ptr]),
    'sgnn_scatter_add_rows_multi_workspace_bytes': (c_i64, [c_i64, c_i64, c_i64]),
    'sgnn_scatter_add_rows_multi': (c_int, [c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr,
                                            c_i64, c_ptr]),
    'sgnn_mpn_bwd_edges': (c_int, [ctypes.POINTER(MpnArgs), c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sgnn_mpn_bwd_edges_many': (c_int, [c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sgnn_mpn_bwd_wp_partial': (c_int, [ctypes.POINTER(MpnArgs), c_ptr, c_ptr, c_i64, c_ptr]),
}

ERRORS = {-1: 'SGNN_ERR_BAD_ARG', -2: 'SGNN_ERR_SET_TOO_LARGE', -3: 'SGNN_ERR_NNZ_TOO_LARGE',
          -4: 'SGNN_ERR_LAUNCH', -5: 'SGNN_ERR_UNSUPPORTED_D'}

_lib = None


class SubgnnHipError(RuntimeError):
    pass


def load():
    """Load the shared library (raises if it has not been built: no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must bring in ITS HIP runtime first: loading libsubgnn_hip.so before torch would bind
    # it to a second copy of libamdhip64 that never sees torch's device context
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise SubgnnHipError('libsubgnn_hip.so is missing (%s): build it with `python -m subgnn_amd.build`; '
                             'this package has no CPU fallback' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    if lib.sgnn_abi_version() != 11:
        raise SubgnnHipError('ABI version mismatch')
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        lib = load()
        detail = lib.sgnn_last_error().decode() if rc == -4 else ''
        raise SubgnnHipError('%s failed: %s %s' % (what, ERRORS.get(rc, rc), detail))
