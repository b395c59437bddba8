"""ctypes binding of libagnn_hip.so (include/agnn.h).  No CPU fallback: missing library or
non-GPU tensors raise.  PyTorch is used only for device memory and the current HIP stream."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libagnn_hip.so")
MAX_SEG = 32

SPMM_MEAN, SPMM_SKIP_SELF, SPMM_ACCUM = 1, 2, 4
INT32_MAX = 2 ** 31 - 1


class AgnnError(RuntimeError):
    pass


class CooSeg(C.Structure):
    _fields_ = [("row", C.c_void_p), ("col", C.c_void_p), ("etype", C.c_void_p),
                ("etype_code", C.c_int64), ("n_edges", C.c_int64), ("n_rows", C.c_int64)]


class RowendItem(C.Structure):
    _fields_ = [("rowptr", C.c_void_p), ("perm", C.c_void_p), ("rowend", C.c_void_p), ("n_rows", C.c_int32),
                ("e_limit", C.c_int32)]


ROWEND_MAX_ITEMS = 64


class Sampler(C.Structure):
    _fields_ = [("n_rel", C.c_int32), ("rowptr", C.c_void_p * 8), ("col", C.c_void_p * 8), ("win_start", C.c_void_p),
                ("n_sub", C.c_int32), ("n_targets", C.c_int32), ("n_hops", C.c_int32), ("fan", C.c_int32 * 4), ("cap", C.c_int32 * 4),
                ("rng", C.c_void_p), ("node_gid", C.c_void_p), ("edges", C.c_void_p * 8), ("e_cap", C.c_int64), ("status", C.c_void_p), ("drops", C.c_void_p), ("kept", C.c_void_p)]


class ReltItem(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("y", C.c_void_p), ("ld_x", C.c_int64), ("ld_y", C.c_int64)]


RELT_MAX_ITEMS = 4


class Rel(C.Structure):
    _fields_ = [("src", C.c_void_p), ("rowptr", C.c_void_p), ("rowend", C.c_void_p), ("col", C.c_void_p),
                ("ew", C.c_void_p), ("colscale", C.c_void_p), ("ld_src", C.c_int64)]


class HgtSrcItem(C.Structure):
    _fields_ = [("rowptr", C.c_void_p), ("rowend", C.c_void_p), ("col", C.c_void_p), ("perm", C.c_void_p), ("alpha", C.c_void_p),
                ("gs", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p), ("ld_o", C.c_int64), ("n_src_rows", C.c_int32),
                ("col_limit", C.c_int32)]


class HgtDstItem(C.Structure):
    _fields_ = [("rels", C.c_void_p), ("n_rel", C.c_int32), ("q", C.c_void_p), ("ld_q", C.c_int64), ("n_rows", C.c_int64),
                ("out", C.c_void_p), ("ld_out", C.c_int64), ("m_out", C.c_void_p), ("linv_out", C.c_void_p)]


HGT_MAX_DST = 4


class HgtRel(C.Structure):
    _fields_ = [("k", C.c_void_p), ("v", C.c_void_p), ("rowptr", C.c_void_p), ("rowend", C.c_void_p),
                ("col", C.c_void_p), ("perm", C.c_void_p), ("pscale", C.c_void_p), ("ld", C.c_int64),
                ("alpha", C.c_void_p), ("gs", C.c_void_p), ("tdot", C.c_void_p), ("ld_tdot", C.c_int64)]


class Gated(C.Structure):
    _fields_ = [("rowptr", C.c_void_p), ("col", C.c_void_p), ("perm", C.c_void_p), ("a", C.c_void_p), ("b", C.c_void_p),
                ("h", C.c_void_p), ("c", C.c_void_p), ("ld", C.c_int64), ("ld_c", C.c_int64), ("n_rows", C.c_int64),
                ("H", C.c_int32)]


class WgradItem(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p), ("ld_dy", C.c_int64), ("ld_x", C.c_int64),
                ("ld_dw", C.c_int64), ("n", C.c_int64), ("out_f", C.c_int32), ("in_f", C.c_int32)]


class ColsumItem(C.Structure):
    _fields_ = [("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("n", C.c_int64), ("H", C.c_int32), ("dgamma", C.c_void_p),
                ("dbeta", C.c_void_p)]


WGRAD_BATCH_MAX = 16
PACK_MAX_SRC = 8


class PackItem(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p * PACK_MAX_SRC), ("ld_dst", C.c_int64), ("ld_src", C.c_int64),
                ("rows", C.c_int32), ("cols", C.c_int32), ("n_src", C.c_int32), ("vec_ok", C.c_int32)]


_lib: Optional[C.CDLL] = None

# every exported symbol of include/agnn.h: (name, restype, argtypes)
SIGNATURES = {
    "agnn_last_error": (C.c_char_p, []),
    "agnn_version": (C.c_int, []),
    "agnn_csr_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "agnn_csr_build": (C.c_int, [C.c_int, C.POINTER(CooSeg), C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "agnn_check_status": (C.c_int, [C.c_void_p, C.c_void_p]),
    "agnn_csr_rowend": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "agnn_csr_rowend_batch": (C.c_int, [C.c_int, C.POINTER(RowendItem), C.c_void_p]),
    "agnn_spmm_f32": (C.c_int, [C.c_int, C.POINTER(Rel), C.c_int64, C.c_int32, C.c_void_p, C.c_int64,
                                C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_uint32,
                                C.c_void_p]),
    "agnn_spmm_root_f32": (C.c_int, [C.c_int, C.POINTER(Rel), C.c_int64, C.c_int32, C.c_void_p, C.c_int64,
                                     C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_uint32,
                                     C.c_void_p]),
    "agnn_gru_fwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "agnn_gru_bwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                   C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "agnn_heads_fwd_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32),
                                     C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "agnn_gru_hprev_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "agnn_hgt_attn_fwd_multi_f32": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "agnn_hgt_attn_fwd_f32": (C.c_int, [C.c_int, C.POINTER(HgtRel), C.c_void_p, C.c_int64, C.c_int64, C.c_int32,
                                        C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "agnn_hgt_attn_bwd_dst_f32": (C.c_int, [C.c_int, C.POINTER(HgtRel), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                            C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                            C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "agnn_hgt_attn_bwd_src_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32,
                                            C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "agnn_hgt_attn_bwd_src_batch_f32": (C.c_int, [C.c_int32, C.POINTER(HgtSrcItem), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                                  C.c_int32, C.c_int32, C.c_void_p]),
    "agnn_sampler_num_nodes": (C.c_int64, [C.POINTER(Sampler)]),
    "agnn_sampler_edge_capacity": (C.c_int64, [C.POINTER(Sampler)]),
    "agnn_sample_hops": (C.c_int, [C.POINTER(Sampler), C.c_void_p]),
    "agnn_sample_members": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                      C.POINTER(C.c_int32), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "agnn_sample_compact_nodes": (C.c_int64, [C.POINTER(Sampler), C.POINTER(C.c_int32)]),
    "agnn_sample_compact": (C.c_int, [C.POINTER(Sampler), C.POINTER(C.c_int32), C.c_void_p, C.c_void_p, C.c_void_p]),
    "agnn_gather_rows_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "agnn_gather_i64": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "agnn_relt_fwd_f32": (C.c_int, [C.c_int, C.POINTER(ReltItem), C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]),
    "agnn_relt_bwd_f32": (C.c_int, [C.c_int, C.POINTER(ReltItem), C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]),
    "agnn_relt_dw_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int32, C.c_int32, C.c_int32, C.c_int64]),
    "agnn_relt_dw_f32": (C.c_int, [C.c_int, C.POINTER(ReltItem), C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_size_t,
                                  C.c_void_p]),
    "agnn_wgrad_batch_workspace_bytes": (C.c_size_t, [C.c_int32, C.POINTER(WgradItem)]),
    "agnn_wgrad_batch_f32": (C.c_int, [C.c_int32, C.POINTER(WgradItem), C.c_void_p, C.c_size_t, C.c_void_p]),
    "agnn_gemm_nt_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                                   C.c_int64, C.c_void_p]),
    "agnn_gemm_nn_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                                   C.c_int64, C.c_void_p]),
    "agnn_absdiff_fwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_void_p]),
    "agnn_absdiff_bwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "agnn_gated_fwd_f32": (C.c_int, [C.POINTER(Gated), C.c_void_p, C.c_int64, C.c_void_p]),
    "agnn_gated_bwd_dst_f32": (C.c_int, [C.POINTER(Gated), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "agnn_gated_bwd_src_f32": (C.c_int, [C.POINTER(Gated), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "agnn_norm_act_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "agnn_norm_act_fwd_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_float, C.c_float,
                                        C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "agnn_norm_act_bwd_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_float, C.c_float,
                                        C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "agnn_norm_act_colsum_batch_f32": (C.c_int, [C.c_int32, C.POINTER(ColsumItem), C.c_void_p]),
    "agnn_skip_act_workspace_bytes": (C.c_size_t, []),
    "agnn_skip_act_fwd_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_uint32,
                                        C.c_void_p, C.c_uint32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "agnn_skip_act_bwd_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_uint32,
                                        C.c_void_p, C.c_uint32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                        C.c_void_p, C.c_size_t, C.c_void_p]),
    "agnn_norm_act_colsum_f32": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "agnn_wgrad_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "agnn_wgrad_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                                 C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "agnn_pack_f32": (C.c_int, [C.c_int32, C.POINTER(PackItem), C.c_void_p]),
    "agnn_debug_stamp": (C.c_int, [C.c_void_p, C.c_void_p]),
    "agnn_gproj_fwd_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "agnn_gproj_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "agnn_gproj_bwd_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_size_t, C.c_void_p]),
    "agnn_spmm_self_grad_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_int32,
                                          C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "agnn_embed_cat_fwd_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "agnn_embed_workspace_bytes": (C.c_size_t, [C.c_int32, C.POINTER(C.c_int32), C.c_int32]),
    "agnn_embed_cat_bwd_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_int32), C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "agnn_train_loss_workspace_bytes": (C.c_size_t, []),
    "agnn_train_loss_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_float, C.c_int64,
                                      C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                      C.c_void_p]),
    "agnn_train_loss_final_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_float, C.c_int64,
                                            C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                            C.c_size_t, C.c_void_p]),
    "agnn_train_loss_bwd_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_int64,
                                          C.c_void_p]),
    "agnn_adamw_workspace_bytes": (C.c_size_t, []),
    "agnn_adamw_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float,
                                 C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t,
                                 C.c_void_p]),
    "agnn_multitask_ce_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_float,
                                        C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "agnn_multitask_ce_scale_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p,
                                              C.c_void_p, C.c_int64, C.c_void_p]),
}


def load() -> C.CDLL:
    """Load the HIP library or raise (the product path never degrades to a CPU implementation)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AgnnError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C analysisgnn_amd/csrc`). analysisgnn_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().agnn_last_error()
        raise AgnnError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def require_gpu(*tensors: Optional[torch.Tensor]) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise AgnnError("analysisgnn_amd ops need tensors on a HIP device (no CPU fallback); "
                            f"got a tensor on {t.device}")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise AgnnError(f"tensors on different devices: {dev} vs {t.device}")
    if dev is None:
        raise AgnnError("no tensor given")
    return dev


_STATUS: dict = {}


def status_word(device: torch.device) -> torch.Tensor:
    """The device-side status word handed to kernels that can detect an inconsistent index (agnn_csr_build): int32[1],
    zero-initialised once per device, only ever incremented."""
    key = str(device)
    if key not in _STATUS:
        if torch.cuda.is_current_stream_capturing():
            raise AgnnError("the device status word must exist before a hipGraph capture starts (it would be re-zeroed by every "
                            "replay): run the step once eagerly first, or call _lib.status_word(device)")
        _STATUS[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _STATUS[key]


def check_device_status(device: torch.device) -> None:
    """Synchronises and raises AgnnError when a kernel flagged an inconsistency since the word was created."""
    check(load().agnn_check_status(status_word(device).data_ptr(), stream_ptr(device)), "agnn_check_status")


def stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def f32c(t: torch.Tensor) -> torch.Tensor:
    """fp32 with unit inner stride and 16-byte aligned rows (copies only when necessary)."""
    if t.dtype != torch.float32:
        raise AgnnError(f"fp32 expected, got {t.dtype}")
    if t.dim() != 2:
        raise AgnnError(f"2-D feature matrix expected, got shape {tuple(t.shape)}")
    if t.stride(1) != 1 or t.stride(0) % 4 != 0 or t.data_ptr() % 16 != 0 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t


def make_rels(items: Sequence[dict]):
    arr = (Rel * len(items))()
    for i, it in enumerate(items):
        arr[i].src = it["src"]
        arr[i].rowptr = it["rowptr"]
        arr[i].rowend = it.get("rowend")
        arr[i].col = it["col"]
        arr[i].ew = it.get("ew")
        arr[i].colscale = it.get("colscale")
        arr[i].ld_src = it["ld_src"]
    return arr


# ---- measurement aid: time stamps captured into the step's hipGraph (scripts/step_stamps.py; AGNN_STAMPS=1) ----------------
STAMPS = {"on": bool(os.environ.get("AGNN_STAMPS")), "names": [], "buf": None,
          "only": [w for w in os.environ.get("AGNN_STAMPS", "").split(",") if w not in ("", "1", "all")]}


def stamp(name: str, device) -> None:
    """With AGNN_STAMPS set: a one-lane kernel on the current stream stores the 100 MHz device counter under `name` (a call
    made while a hipGraph is being captured becomes a node of it: every replay refreshes the slot).  Otherwise nothing."""
    if not STAMPS["on"] or (STAMPS["only"] and not any(w in name for w in STAMPS["only"])):
        return                          # AGNN_STAMPS=1: all of them; AGNN_STAMPS=a,b: names containing a or b (every stamp is a graph node)
    if STAMPS["buf"] is None:
        STAMPS["buf"] = torch.zeros(256, dtype=torch.int64, device=device)
    if name not in STAMPS["names"]:
        STAMPS["names"].append(name)
    k = STAMPS["names"].index(name)
    check(load().agnn_debug_stamp(STAMPS["buf"].data_ptr() + 8 * k, stream_ptr(torch.device(device))), "agnn_debug_stamp")
