"""ctypes binding of libtmpnn.so (the C ABI declared in include/tmpnn.h).

The library is the product: there is NO fallback.  If the shared object is missing or a call
fails, a RuntimeError is raised -- nothing silently reroutes through torch ops or the oracle.
"""
from __future__ import annotations

import ctypes as C

import torch
import os
import re
from typing import Dict, List, Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# TMPNN_LIB_PATH: load another build of the SAME library (kernel experiments of tools/build_variant.sh); never a fallback
LIB_PATH = os.environ.get('TMPNN_LIB_PATH') or os.path.join(_HERE, 'lib', 'libtmpnn.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'tmpnn.h')

ABI_VERSION = 4

c_int = C.c_int
c_void_p = C.c_void_p
c_size_t = C.c_size_t
c_float = C.c_float


class CSegPlan(C.Structure):
    """struct tmpnn_seg_plan (include/tmpnn.h): the single-read segment sum of a dense graph."""
    _fields_ = [('T', C.c_int32), ('I', C.c_int32), ('nsplit', C.c_int32), ('t_row', c_void_p), ('items', c_void_p),
                ('rowptr2', c_void_p), ('inc2', c_void_p), ('ws', c_void_p), ('ws_floats', c_size_t)]


class CWinPlan(C.Structure):
    """struct tmpnn_win_plan (include/tmpnn.h): the single-read segment sum of a batch of small windows."""
    _fields_ = [('W', C.c_int32), ('nbig', C.c_int32), ('wrec', c_void_p), ('erow', c_void_p), ('recs', c_void_p),
                ('det', c_void_p), ('drow', c_void_p), ('big_order', c_void_p)]


class CGraph(C.Structure):
    """struct tmpnn_graph (include/tmpnn.h)."""
    _fields_ = [('N', C.c_int32), ('E', C.c_int32), ('Dn', C.c_int32),
                ('src', c_void_p), ('dst', c_void_p), ('edge_row', c_void_p), ('det_row', c_void_p),
                ('rowptr', c_void_p), ('inc', c_void_p), ('det_order', c_void_p), ('seg_plan', c_void_p),
                ('win_plan', c_void_p)]


_GP = C.POINTER(CGraph)


class CEdgeTiles(C.Structure):
    """struct tmpnn_edge_tiles (include/tmpnn.h)."""
    _fields_ = [('T', C.c_int32), ('rows_per_tile', C.c_int32), ('t_row', c_void_p), ('t_loc', c_void_p),
                ('t_dptr', c_void_p), ('t_dets', c_void_p)]


_TP = C.POINTER(CEdgeTiles)


class CDGraph(C.Structure):
    """struct tmpnn_dgraph (include/tmpnn.h): index-form graph whose sizes live on the device."""
    _fields_ = [('N', C.c_int32), ('cap', C.c_int32), ('meta', c_void_p), ('is_edge', c_void_p), ('pos', c_void_p),
                ('src', c_void_p), ('dst', c_void_p), ('src_pos', c_void_p), ('dst_pos', c_void_p),
                ('edge_row', c_void_p), ('det_row', c_void_p), ('rowptr', c_void_p), ('inc', c_void_p)]


_P3 = c_void_p * 3


class CMpParams(C.Structure):
    """struct tmpnn_mp_params (include/tmpnn.h): every parameter (or its gradient buffer) as a device pointer."""
    _fields_ = [('G', C.c_int32), ('H', C.c_int32), ('IN_e', C.c_int32), ('F_total', C.c_int32), ('F', C.c_int32 * 3),
                ('w1', _P3), ('b1', _P3), ('gamma', _P3), ('beta', _P3), ('w2', _P3), ('b2', _P3),
                ('run_mean', _P3), ('run_var', _P3), ('num_batches_tracked', _P3),
                ('e_wih', _P3), ('e_whh', _P3), ('e_bih', _P3), ('e_bhh', _P3),
                ('n_wih', _P3), ('n_whh', _P3), ('n_bih', _P3), ('n_bhh', _P3),
                ('w_node', c_void_p), ('b_node', c_void_p), ('w_edge', c_void_p), ('b_edge', c_void_p)]


class CTrackRows(C.Structure):
    """struct tmpnn_track_rows (include/tmpnn.h): the row form of the rolling tracking graph."""
    _fields_ = [('ts', c_void_p), ('det_id', c_void_p), ('assoc', c_void_p), ('is_edge', c_void_p), ('src', c_void_p),
                ('dst', c_void_p), ('labels', c_void_p)]


_DGP = C.POINTER(CDGraph)
_MPP = C.POINTER(CMpParams)
_TRP = C.POINTER(CTrackRows)

# name -> (restype, argtypes); must mirror include/tmpnn.h (tests/test_abi.py cross-checks the names)
_SIGNATURES = {
    'tmpnn_abi_version': (c_int, []),
    'tmpnn_last_error': (C.c_char_p, []),
    'tmpnn_gather_diff_fwd': (c_int, [_GP, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    'tmpnn_gather_concat_fwd': (c_int, [_GP, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    'tmpnn_gather_diff_bwd': (c_int, [_GP, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    'tmpnn_gather_concat_bwd': (c_int, [_GP, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    'tmpnn_segsum_fwd': (c_int, [_GP, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'tmpnn_segsum_fwd_live': (c_int, [_GP, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'tmpnn_segsum_bwd': (c_int, [_GP, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    'tmpnn_att_fwd': (c_int, [_GP, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'tmpnn_att_bwd_ws': (c_size_t, [c_int, c_int, c_int, c_int]),
    'tmpnn_att_index': (c_int, [_GP, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'tmpnn_att_bwd': (c_int, [_GP, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                              c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t,
                              c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'tmpnn_att_bwd_heads': (c_int, [_GP, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                    c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                    c_size_t, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'tmpnn_gru_fwd': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                              c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_size_t, c_void_p]),
    'tmpnn_gru_fwd_head_parts': (c_int, [c_int, c_int, c_int]),
    'tmpnn_gru_fwd_tiles': (c_int, [_TP, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_size_t, c_void_p]),
    'tmpnn_heads_finish': (c_int, [c_void_p, c_size_t, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'tmpnn_gru_bwd_data': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                   c_void_p, c_size_t, c_void_p, c_int, c_void_p, c_void_p,
                                   c_void_p, c_int, c_void_p, c_int,
                                   c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'tmpnn_gru_bwd_weights_ws': (c_size_t, [c_int, c_int, c_int]),
    'tmpnn_gru_bwd_weights': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                      c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_int, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'tmpnn_gru_bwd_fused_available': (c_int, [c_int, c_int, c_int]),
    'tmpnn_gru_bwd_fused_ws': (c_size_t, [c_int, c_int, c_int]),
    'tmpnn_gru_bwd_fused': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                    c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_int,
                                    c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                    c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_size_t, c_void_p]),
    'tmpnn_rows_linear': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    'tmpnn_transpose': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'tmpnn_input_bn_fwd': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'tmpnn_input_bn_bwd_ws': (c_size_t, [c_int, c_int, c_int, c_int]),
    'tmpnn_input_bn_bwd': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                   c_void_p, c_int, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_size_t, c_void_p]),
    'tmpnn_input_tf_supported': (c_int, [c_int, c_int, c_int]),
    'tmpnn_input_tf_fwd': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'tmpnn_input_tf_bwd_ws': (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    'tmpnn_input_tf_bwd': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                   c_void_p, c_int, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_size_t, c_void_p]),
    'tmpnn_heads_fwd': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p]),
    'tmpnn_heads_bwd_ws': (c_size_t, [c_int, c_int]),
    'tmpnn_heads_bwd': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_size_t, c_void_p]),
    'tmpnn_targets': (c_int, [_GP, c_void_p, c_void_p, c_void_p]),
    'tmpnn_ce_loss_ws': (c_size_t, [c_int]),
    'tmpnn_ce_loss_fwd': (c_int, [_GP, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'tmpnn_ce_loss_bwd': (c_int, [_GP, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'tmpnn_focal_loss_ws': (c_size_t, [c_int]),
    'tmpnn_focal_loss_fwd': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_float, c_int, c_float, c_float, c_void_p,
                                     c_void_p, c_size_t, c_void_p]),
    'tmpnn_focal_loss_bwd': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_float, c_int, c_float, c_float, c_void_p,
                                     c_float, c_void_p, c_void_p]),
    'tmpnn_bce_logits_ws': (c_size_t, [C.c_long]),
    'tmpnn_bce_logits_sum_fwd': (c_int, [c_void_p, c_void_p, C.c_long, c_void_p, c_void_p, c_size_t, c_void_p]),
    'tmpnn_bce_logits_sum_bwd': (c_int, [c_void_p, c_void_p, C.c_long, c_void_p, c_void_p, c_void_p]),
    'tmpnn_train_losses_supported': (c_int, [c_int, c_int]),
    'tmpnn_train_losses_ws': (c_size_t, [c_int, c_int]),
    'tmpnn_train_losses_fwd': (c_int, [_GP, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_size_t, c_void_p]),
    'tmpnn_train_losses_bwd': (c_int, [_GP, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_int, c_void_p, c_void_p, c_void_p]),
    'tmpnn_dgraph_ints': (c_size_t, [c_int]),
    'tmpnn_dgraph_bind': (c_int, [c_void_p, c_int, c_int, _DGP]),
    'tmpnn_graph_from_coo': (c_int, [c_int, c_void_p, c_void_p, C.c_int64, c_void_p, c_void_p, C.c_int64, _DGP, c_void_p]),
    'tmpnn_graph_from_coo_arena': (c_int, [c_int, c_void_p, c_void_p, C.c_int64, c_void_p, c_void_p, C.c_int64, c_void_p, c_int, c_void_p]),
    'tmpnn_graph_from_coo_arena_ws': (c_int, [c_int, c_void_p, c_void_p, C.c_int64, c_void_p, c_void_p, C.c_int64, c_void_p, c_int,
                                              c_void_p, c_size_t, c_void_p]),
    'tmpnn_graph_from_rows_ws': (c_int, [c_int, c_void_p, c_void_p, c_void_p, _DGP, c_void_p, c_size_t, c_void_p]),
    'tmpnn_track_finalize_ws': (c_size_t, [c_int]),
    'tmpnn_track_load': (c_int, [c_int, c_int, c_void_p, _TRP, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, _DGP, c_void_p,
                                 c_size_t, c_void_p]),
    'tmpnn_track_select': (c_int, [_DGP, _TRP, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'tmpnn_track_select_ws': (c_int, [_DGP, _TRP, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'tmpnn_track_hungarian_max_dets': (c_int, []),
    'tmpnn_track_extend': (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, _TRP, c_void_p, c_int, c_int,
                                   c_void_p, c_int, _DGP, c_void_p, c_size_t, c_void_p]),
    'tmpnn_track_extend_tf': (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, _TRP, c_void_p, c_int, c_void_p,
                                      c_void_p, c_void_p, c_size_t, _DGP, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'tmpnn_track_retire': (c_int, [_DGP, _TRP, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t,
                                   c_void_p, c_void_p, _TRP, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int,
                                   c_void_p, c_void_p, c_void_p]),
    'tmpnn_wide_supported': (c_int, [c_int, c_int]),
    'tmpnn_wide_prep_bytes': (c_size_t, [c_int, c_int]),
    'tmpnn_wide_prepare': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'tmpnn_wide_gru_fwd': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'tmpnn_wide_gru_fwd_tiled': (c_int, [c_void_p, c_void_p, c_int, _TP, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'tmpnn_wide_gru_bwd_diff_ws': (c_size_t, [c_int, c_int, c_int, c_int]),
    'tmpnn_wide_gru_bwd_diff': (c_int, [c_void_p, _GP, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_int,
                                        c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_size_t, c_void_p]),
    'tmpnn_wide_gru_bwd_diff_fused': (c_int, [c_void_p, _GP, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_int,
                                              c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_size_t, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'tmpnn_mp_iter_prep_floats': (c_size_t, [c_int, c_int, c_int]),
    'tmpnn_mp_iter_prepare': (c_int, [_MPP, c_void_p, c_void_p]),
    'tmpnn_mp_iter_save_floats': (c_size_t, [c_int, c_int, c_int, c_int]),
    'tmpnn_mp_iter_fwd': (c_int, [_MPP, c_void_p, _DGP, c_int, c_void_p, c_int, c_void_p, c_int,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'tmpnn_mp_iter_save_es_offset': (c_size_t, [c_int, c_int, c_int, c_int]),
    'tmpnn_mp_iter_fwd_parts': (c_int, [_MPP, c_void_p, _DGP, c_int, c_void_p, c_int, c_void_p, c_int,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    'tmpnn_mp_iter_bwd_parts': (c_int, [_MPP, c_void_p, _DGP, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, _MPP, c_void_p,
                                        c_size_t, c_int, c_void_p]),
    'tmpnn_mp_iter_bwd_ws': (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    'tmpnn_mp_iter_bwd': (c_int, [_MPP, c_void_p, _DGP, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, _MPP, c_void_p,
                                  c_size_t, c_void_p]),
}

_lib: Optional[C.CDLL] = None


# entry points only a comparison build (-DTMPNN_KEEP_VARIANTS, tools/build_variant.sh) exports: bound when the loaded library
# has them (TMPNN_LIB_PATH pointing at such a build), absent from the shipped one
_VARIANT_SIGNATURES = {
    'tmpnn_track_append': (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'tmpnn_gru_bwd_weights_variant': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                              c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_int, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    'tmpnn_gru_bwd_weights_choice': (c_int, []),
    'tmpnn_wide_gru_bwd_diff_aux': (c_int, [c_void_p, _GP, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_int,
                                            c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    'tmpnn_wide_gru_bwd_data_ws': (c_size_t, [c_int, c_int]),
    'tmpnn_wide_gru_bwd_data': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_int,
                                        c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'tmpnn_wide_gru_bwd_weights_ws': (c_size_t, [c_int, c_int]),
    'tmpnn_wide_gru_bwd_weights': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
}


def header_symbols(variants: bool = False) -> List[str]:
    """Every function name include/tmpnn.h declares for the shipped library (variants=True: also the comparison-build
    entry points of its `#ifdef TMPNN_KEEP_VARIANTS` section)."""
    with open(HEADER_PATH) as f:
        txt = f.read()
    if not variants:
        txt = re.sub(r'#ifdef TMPNN_KEEP_VARIANTS.*?#endif', '', txt, flags=re.S)
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(tmpnn_[a-z0-9_]+)\s*\(', txt)))


def load() -> C.CDLL:
    """Load libtmpnn.so (once).  Fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(hipcc --offload-arch=gfx950).  trackmpnn_amd has no CPU or torch fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f'libtmpnn.so does not export {name}; rebuild it') from e
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in _VARIANT_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    v = lib.tmpnn_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError(f'libtmpnn.so ABI version {v} != expected {ABI_VERSION}; rebuild it')
    _lib = lib
    return lib


def last_error() -> str:
    return load().tmpnn_last_error().decode('utf-8', 'replace')


def call(name: str, *args) -> None:
    """Invoke an int-returning entry point; raise RuntimeError(tmpnn_last_error()) on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f'{name} failed (code {rc}): {last_error()}')


def fn(name: str):
    """The bound ctypes function (for hot paths that call it many times; the caller checks the return code)."""
    return getattr(load(), name)


def ptr(t) -> Optional[int]:
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def raw_stream(device=None) -> int:
    """hipStream_t of torch's current stream on `device` (default: the current device) as an integer.  The public
    torch.cuda.current_stream() builds a Stream object per call (~7 us: twice per batch-1 forward call it was a seventh
    of the host time); the raw accessor is a plain C call."""
    idx = torch.cuda.current_device() if device is None or device.index is None else device.index
    return torch._C._cuda_getCurrentRawStream(idx)
