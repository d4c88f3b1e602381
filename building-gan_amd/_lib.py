"""ctypes binding of csrc/libbgan_hip.so (declared in include/bgan.h).

The library is the only compute path: if it is missing or a call fails, an exception is raised;
nothing here falls back to PyTorch or to the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
def _lib_path() -> str:
    """The in-tree library.  BG_LIB_PATH may name another build of it for A/B timing (``make tuning``, a variant under
    csrc/variants/), but only inside this package's csrc/ directory: the loader never follows the variable elsewhere."""
    default = os.path.join(_HERE, "csrc", "libbgan_hip.so")
    alt = os.environ.get("BG_LIB_PATH")
    if not alt:
        return default
    root = os.path.realpath(os.path.join(_HERE, "csrc")) + os.sep
    real = os.path.realpath(alt)
    if not real.startswith(root):
        raise ImportError("BG_LIB_PATH=%r is outside %s: only builds of this package's own sources are loaded" % (alt, root))
    return real


LIB_PATH = _lib_path()

_p = C.c_void_p
_i64 = C.c_int64
_i32 = C.c_int32
_f = C.c_float
_int = C.c_int

# name -> argument types (all functions return int)
SIGNATURES = {
    "bg_zero_bytes": [_p, _i64, _p],
    "bg_copy2d_bytes": [_p, _i64, _p, _i64, _i64, _i64, _p],
    "bg_fill_copy_batch": [_i32, _p, _p, _p, _p, _p, _p, _p],
    "bg_plan_csr": [_p, _p, _i64, _i64, _p, _p, _p, _p, _p, _p],
    "bg_plan_segptr": [_p, _i64, _i64, _p, _p, _p],
    "bg_plan_floor_level": [_p, _p, _i64, _i64, _p, _p, _p],
    "bg_plan_counts": [_p, _i64, _p, _p],
    "bg_plan_chunks": [_p, _i64, _i32, _p, _p],
    "bg_spmm_f32": [_p, _i64, _p, _p, _p, _p, _p, _int, _p, _int, _p, _i64, _i64, _i64, _p],
    "bg_connectivity": [_p, _i64, _p, _p, _i64, _p, _i64, _i64, _p, _p, _i64, _p, _i64, _p, _i64, _p, _p],
    "bg_program_ratio_f32": [_p, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p],
    "bg_spmm_batch_f32": [_p, _i64, _i64, _p, _p, _p, _int, _p, _int, _p, _i64, _i64, _i64, _i64, _i64, _p],
    "bg_spmm_add_f32": [_p, _i64, _p, _p, _p, _int, _p, _int, _p, _i64, _p, _i64, _p, _i64, _p, _p, _int,
                        _p, _i64, _i64, _i64, _p],
    "bg_spmm_chain_f32": [_p, _i64, _p, _p, _p, _int, _p, _int, _p, _i64, _i64, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64,
                          _i64, _i64, _i32, _p],
    "bg_spmm_chain_grouped_f32": [_p, _i64, _p, _p, _p, _int, _p, _int, _p, _i64, _i64, _i64, _i64, _p, _i64, _i64, _p, _i64,
                                  _i64, _i64, _i64, _i32, _i64, _i64, _i64, _i64, _p],
    "bg_segreduce_sum_f32": [_p, _i64, _p, _p, _p, _p, _p, _int, _p, _i32, _i64, _p, _p, _i64, _i64, _i64, _p],
    "bg_segmax_f32": [_p, _i64, _p, _p, _p, _i32, _i64, _p, _p, _p, _i64, _p, _i64, _i64, _p],
    "bg_segmax_static_f32": [_p, _i64, _p, _p, _p, _i32, _i64, _p, _i64, _p, _i64, _i64, _p],
    "bg_gather_arg_f32": [_p, _i64, _p, _p, _i64, _i64, _i64, _p],
    "bg_scatter_arg_f32": [_p, _i64, _p, _p, _p, _i64, _i64, _i64, _i64, _p],
    "bg_segsum_acc_f32": [_p, _i64, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p],
    "bg_segmax_batch_f32": [_p, _i64, _i64, _p, _p, _p, _i64, _i64, _p, _i64, _i64, _i64, _i64, _p],
    "bg_gather_arg_pair_f32": [_p, _i64, _p, _p, _i64, _p, _i64, _p, _p, _i64, _i64, _i64, _p],
    "bg_scatter_arg_pair_f32": [_p, _i64, _p, _p, _p, _i64, _p, _i64, _p, _p, _p, _i64, _i64, _i64, _i64, _p],
    "bg_scatter_arg_batch_f32": [_p, _i64, _i64, _p, _i64, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _p, _i64, _p],
    "bg_sddmm_f32": [_p, _i64, _p, _i64, _p, _p, _p, _p, _i64, _i64, _p],
    "bg_cross_logits_f32": [_p, _i64, _p, _i64, _p, _p, _p, _p, _i64, _i64, _p],
    "bg_cross_logits_bwd_f32": [_p, _i64, _p, _i64, _p, _p, _p, _p, _p, _p, _i64, _p, _i32, _i64, _i64, _p],
    "bg_seg_gumbel_softmax_f32": [_p, _p, _p, _f, _p, _p, _p, _i64, _p],
    "bg_seg_softmax_bwd_f32": [_p, _p, _p, _f, _p, _i64, _p],
    "bg_seg_argmax_f32": [_p, _p, _p, _i64, _p],
    "bg_story_project_f32": [_p, _i64, _p, _p, _p, _p, _i64, _i64, _p, _i64, _i64, _i64, _p],
    "bg_gather_story_f32": [_p, _i64, _p, _p, _p, _int, _p, _i64, _i64, _p, _p, _p, _p, _i64, _i64, _p, _i64, _i64, _i64, _p],
    "bg_attention_fwd_f32": [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _p, _p, _p, _f,
                             _p, _p, _p, _p, _p, _i64, _i64, _i64, _p, _i64, _p, _p],
    "bg_mask_gumbel_f32": [_p, _i64, _p, _p, _p, _i64, _p],
    "bg_mask_gumbel_bwd_f32": [_p, _p, _p, _p, _p, _i64, _i64, _p],
    "bg_gp_sample_f32": [_p, _i64, _p, _i64, _p, _p, _i64, _i64, _i64, _p],
    "bg_pack_columns_f32": [_p, _p, _int, _i64, _i64, _p, _i64, _i64, _p],
    "bg_construct_output_f32": [_p, _p, _p, _p, _p, _p, _i64, _i64, _p, _p, _p, _i64, _p],
    "bg_story_bwd_f32": [_p, _i64, _p, _i64, _p, _i64, _p, _p, _p, _f, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _p],
    "bg_d_coef_init_f32": [_p, _p, _p, _p, _p, _p, _i64, _p, _i64, _i64, _i64, _i64, _i64, _i64, _p, _p, _p],
    "bg_d_coef_init_bwd_f32": [_p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _p, _p, _p, _p, _p, _i64, _p, _p],
    "bg_spmm_chain_pair_f32": [_p, _i64, _p, _i64, _i64, _i64, _p, _i64, _p, _i64, _i64, _i64, _p, _p, _p, _int, _p, _int,
                               _p, _i64, _i64, _i64, _i64, _i64, _i32, _p],
    "bg_cheb_shift_f32": [_p, _p, _i64, _i64, _i64, _i32, _p],
    "bg_tile_f32": [_p, _p, _i64, _i64, _p],
    "bg_d_coef_init_strided_f32": [_p, _p, _p, _p, _p, _p, _i64, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _p, _p, _p],
    "bg_d_coef_init_bwd_strided_f32": [_p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _p, _p, _p, _p, _p, _i64, _p, _i64,
                                       _i32, _p],
    "bg_d_compose_prep_f32": [_p, _i64, _p, _p, _i64, _p, _i64, _p, _p, _p],
    "bg_d_compose_bwd_finish_f32": [_p, _p, _i64, _p, _p, _p, _p, _p],
    "bg_d_table_forward_f32": [_p, _i64, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i32,
                               _p, _p, _p, _p, _p, _p],
    "bg_d_table_backward_f32": [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i32] + [_p] * 15,
    "bg_cheb_shift_slabs_f32": [_p, _p, _i64, _i64, _i64, _i64, _i64, _i32, _p],
    "bg_gemm_nbr_f32": [_i64, _i64, _int, _p, _p, _p, _p, _i64, _p, _i64, _p, _int, _f, _p, _i64, _p, _i64, _p, _i64, _p, _p, _p, _p,
                        _p, _i64, _p, _i64, _p],
    "bg_decoder_bwd_f32": [_i64, _i64, _i64, _i64, _p, _p, _i64, _p, _p, _i64, _p, _i64, _p, _p, _p, _i64, _p, _i64, _p, _p, _p,
                           _p, _i64, _p],
    "bg_decoder_fwd_pair_f32": [_i64, _i64, _i64, _p, _i64, _p, _i64, _p, _p, _p, _p, _p,
                                _i64, _p, _i64, _p, _i64, _p, _p, _p, _p, _p, _p],
    "bg_decoder_fwd_f32": [_i64, _i64, _i64, _p, _i64, _p, _i64, _p, _p, _p, _p, _p, _p],
    "bg_gemm_cheb_f32": [_int, _i64, _i64, _p, _i64, _i64, _i64, _i64, _int, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p],
    "bg_attention_bwd_head_f32": [_p, _i64, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _i64, _p, _p, _i64,
                                  _i64, _i64, _p],
    "bg_pos_terms_f32": [_p, _i64, _p, _p, _p, _i64, _f, _p],
    "bg_d_static_cols_f32": [_p, _i64, _i64, _p, _i64, _p, _p, _i64, _i64, _i64, _p],
    "bg_critic_loss_f32": [_p, _p, _i64, _p, _p, _i64, _p, _i64, _i64, _i64, _f, _p, _p, _p],
    "bg_critic_loss_bwd_f32": [_p, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _f, _p, _p, _p, _p],
    "bg_gemm_f32": [_int, _int, _i64, _i64, _int, _p, _p, _p, _p, _i64, _p, _i64, _p, _p, _int, _f, _p, _i64, _p, _i64,
                    _p, _p, _i64, _p, _i64, _p],
    "bg_gemm_tn_splitk_f32": [_i64, _i64, _p, _i64, _int, _p, _p, _p, _p, _i64, _p, _i32, _int, _p],
    "bg_colsum_f32": [_p, _i64, _i64, _i64, _p, _i32, _p, _int, _p],
    "bg_generator_forward_f32": [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p],
    "bg_d_coef_forward_f32": [_p, _p, _p, _p, _p, _p, _i64, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _p, _p, _p],
    "bg_d_coef_backward_f32": [_p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _p, _p, _p, _p, _p, _p, _p,
                               _p, _p, _p, _p],
    "bg_adam_f32": [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _i32, _p, _p],
    "bg_regroup_f32": [_p, _p, _i64, _i64, _i64, _i64, _int, _p],
    "bg_add_f32": [_p, _p, _p, _i64, _p],
    "bg_fill_f32": [_p, _i64, _f, _p],
    "bg_regroup_iota_f32": [_p, _p, _i64, _i64, _i64, _i64, _int, _p, _i64, _i64, _p],
    "bg_iota_mod_i32": [_p, _i64, _i64, _p],
    "bg_critic_head_forward_f32": [_p, _f, _p, _i64, _p, _p],
    "bg_critic_head_backward_f32": [_p, _f, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "bg_gloss_head_forward_f32": [_p, _p, _i64, _i64, _p, _i64, _p, _p],
    "bg_gloss_head_backward_f32": [_p, _i64, _p, _p, _i64, _p, _p],
    "bg_leaky_bwd_f32": [_p, _p, _f, _p, _i64, _p],
    "bg_rowscale_f32": [_p, _i64, _p, _p, _i64, _i64, _i64, _p],
    "bg_pstack_forward_f32": [_p, _p, _p, _p, _i64, _p, _p],
    "bg_pstack_backward_f32": [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _p],
    "bg_gstack_forward_f32": [_p, _p, _p, _p, _p, _p, _p, _i64, _p, _i64, _p, _p, _p, _p, _p],
    "bg_gstack_backward_f32": [_p, _p, _p, _p, _p, _p, _i64, _p, _i64] + [_p] * 6 + [_p] * 10 + [_p],
    "bg_plan_layout": [_p, _p, _p, _p],
    "bg_plan_build": [_p, _p, _p, _p],
}

_fp = C.c_void_p      # device pointers travel as integers
BG_OK, BG_ERR_ARG, BG_ERR_SHAPE, BG_ERR_ALIGN, BG_ERR_LAUNCH, BG_ERR_SCRATCH = 0, -1, -2, -3, -4, -5      # include/bgan.h


class GeneratorWeights(C.Structure):
    """include/bgan.h: bgGeneratorWeights (field order is the ABI)."""
    _fields_ = [(n, _i64) for n in ("H", "noise_dim", "program_layer", "voxel_layer", "n_class", "max_len")] + \
               [(n, _fp) for n in ("p_enc_w", "p_enc_b", "p_msg_w", "p_msg_b", "p_upd_w", "p_upd_b", "v_enc_w", "v_enc_b",
                                   "pe", "Wc")] + [("ldwc", _i64)] + \
               [(n, _fp) for n in ("bc", "Wsb", "Wcat", "bcat", "nnp_w", "nnp_b", "theta")]


class GeneratorPlan(C.Structure):
    """include/bgan.h: bgGeneratorPlan."""
    _fields_ = [(n, _i64) for n in ("Np", "Nv", "S", "Ec", "cx_max_len", "vox_all_has")] + \
               [(n, _fp) for n in ("pp_rowptr", "pp_col", "pp_scale", "prog_has", "vx_rowptr", "vx_col", "vx_count",
                                   "st_rowptr", "st_col", "st_count", "story_of_voxel", "sb_rowptr", "sb_col", "sb_count",
                                   "level", "cx_rowptr", "cx_col")]


class PlanIn(C.Structure):
    """include/bgan.h: bgPlanIn."""
    _fields_ = [(n, _i64) for n in ("Np", "Nv", "Ep", "Ev", "Ec", "B", "S", "n_slot")] + \
               [(n, _fp) for n in ("pe_src", "pe_dst", "ve_src", "ve_dst", "cep", "cev", "vfc", "vbi", "pcc", "slot_graph",
                                   "target_ratio")] + [("chunk_rows", _i32 * 4)]


class GeneratorBackPlan(C.Structure):
    """include/bgan.h: bgGeneratorBackPlan."""
    _fields_ = [(n, _fp) for n in ("vt_rowptr", "vt_col", "cp_rowptr", "cp_col", "cp_perm")] + [("cp_one_launch", _i64)]


class CriticHead(C.Structure):
    """include/bgan.h: bgCriticHead."""
    _fields_ = [(n, _i64) for n in ("N", "S", "B", "L", "ds", "dl", "H2", "Hd", "blk_n", "blk_max_rows", "blk_max_nnz",
                                    "blk_max_deg", "basis")] + \
               [(n, _fp) for n in ("Zs", "Zk", "W", "b_w1", "b_b1", "b_w2", "b_b2", "s_w1", "s_b1", "s_w2", "s_b2", "wb", "ws",
                                   "st_rowptr", "st_col", "story_of_voxel", "sg_rowptr", "sg_col", "graph_of_story",
                                   "graph_of_voxel", "a_rowptr", "a_col", "a_count", "at_rowptr", "at_col", "blk_ptr")]


# include/bgan.h: the BGP_* / BGS_* enums (order is the ABI; tests/test_cabi.py checks them against the header)
PLAN_I32 = ("PF_ROWPTR", "PF_COL", "PF_PERM", "PR_ROWPTR", "PR_COL", "PR_PERM", "SLOT_ROWPTR", "SLOT_COL", "SLOT_PERM",
            "SG_ROWPTR", "SG_COL", "SG_PERM", "VFW_ROWPTR", "VFW_COL", "VFW_PERM", "VRV_ROWPTR", "VRV_COL", "VRV_PERM",
            "ST_ROWPTR", "ST_COL", "ST_PERM", "ST_CHUNK", "GR_ROWPTR", "GR_COL", "GR_PERM", "GR_CHUNK",
            "SGR_ROWPTR", "SGR_COL", "SGR_PERM", "SGR_CHUNK", "CV_ROWPTR", "CV_COL", "CV_PERM", "CP_ROWPTR", "CP_COL",
            "CP_PERM", "CP_CHUNK", "ARANGE", "PCC32", "SLOTGRAPH32", "VFC32", "VBI32", "STORYGRAPH32", "CEV32", "CEP32",
            "FIRST", "LEVEL", "SS_ROWPTR", "SS_COL", "PP_ROWPTR", "PP_COL", "FLAGS", "STATS", "CURSOR", "STORYGRAPH64")
PLAN_F32 = ("PF_COUNT", "SLOT_COUNT", "VFW_COUNT", "ST_COUNT", "SGR_COUNT", "SS_COUNT", "PP_SCALE", "VOX_HAS", "PROG_HAS")
PLAN_STATS = ("MAX_LEVEL", "MIN_VDEG", "MIN_PDEG", "MIN_CROSS", "MIN_STORY", "MAXLEN_ST", "MAXLEN_GR", "MAXLEN_SGR",
              "MAXLEN_CP", "MAXLEN_CV", "NNZ_SS", "N_LEAVING", "BLK_NNZ", "VOX_MAX_DEG", "NNZ_SLOT")

_lib: Optional[C.CDLL] = None


class BganError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the shared library once; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BganError(f"{LIB_PATH} not found: build it with `make -C building-gan_amd/csrc` "
                        "(or __graft_entry__.build()); there is no fallback path")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.bg_version.restype = C.c_int
    lib.bg_generator_forward_workspace.restype = C.c_int64
    lib.bg_generator_forward_workspace.argtypes = [_i64] * 5
    lib.bg_d_coef_backward_workspace.restype = C.c_int64
    lib.bg_d_coef_backward_workspace.argtypes = [_i64] * 4
    lib.bg_pstack_workspace.restype = C.c_int64
    lib.bg_pstack_workspace.argtypes = [_i64] * 4
    lib.bg_gstack_workspace.restype = C.c_int64
    lib.bg_gstack_workspace.argtypes = [_i64] * 6
    lib.bg_gloss_head_workspace.restype = C.c_int64
    lib.bg_gloss_head_workspace.argtypes = [_p]
    lib.bg_critic_head_workspace.restype = C.c_int64
    lib.bg_critic_head_workspace.argtypes = [_p]
    lib.bg_abi_sizeof.restype = C.c_int64
    lib.bg_abi_sizeof.argtypes = [C.c_int]
    for which, st in enumerate((GeneratorWeights, GeneratorPlan, PlanIn, CriticHead, GeneratorBackPlan)):
        if lib.bg_abi_sizeof(which) != C.sizeof(st):
            raise BganError(f"{LIB_PATH} was built from another revision of include/bgan.h ({st.__name__}: "
                            f"{lib.bg_abi_sizeof(which)} bytes there, {C.sizeof(st)} here): rebuild it "
                            "(`make -C building-gan_amd/csrc`)")
    lib.bg_error_string.restype = C.c_char_p
    lib.bg_error_string.argtypes = [C.c_int]
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().bg_error_string(rc).decode()
        raise BganError(f"{what} failed with code {rc}: {msg}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


_PTR_ARRAYS = {}


def ptr_array_type(n: int):
    """``c_void_p * n``, made once per n (a ctypes array type is a class, and a class is a reference cycle of its own:
    one made per call is garbage that only the cyclic collector frees)."""
    t = _PTR_ARRAYS.get(n)
    if t is None:
        t = _PTR_ARRAYS[n] = C.c_void_p * n
    return t


def stream() -> int:
    """Raw handle of torch's current HIP stream on the current device.  Called once per kernel launch: the
    C-level getters cost ~0.3 us against ~4 us for torch.cuda.current_stream().cuda_stream (a Stream object per
    call), which was 1.2 ms of an eager training iteration."""
    if _raw_stream is not None and _get_device is not None:
        return _raw_stream(_get_device())
    return torch.cuda.current_stream().cuda_stream


def require_device(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise BganError(f"{what}: tensor must live on the GPU (got {t.device}); the HIP library is the only "
                        "compute path of building-gan_amd")
