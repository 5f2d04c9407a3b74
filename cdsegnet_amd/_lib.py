"""ctypes binding of libcdseg_hip.so (C ABI declared in include/cdseg.h).

The product path has NO fallback: if the HIP library is missing or a call fails, this
module raises.  (`python -m cdsegnet_amd.build` or `__graft_entry__.build()` builds it.)
"""
import ctypes
import os
import threading
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_long, c_size_t, c_uint64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcdseg_hip.so")          # 16-bit type = bfloat16
LIB_PATH_F16 = os.path.join(HERE, "libcdseg_hip_f16.so")  # 16-bit type = IEEE half (same sources, -DCDSEG_LP_F16)

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_SWISH = 0, 1, 2
ORDER_IDS = {"z": 0, "z-trans": 1, "hilbert": 2, "hilbert-trans": 3}
ERRORS = {-1: "CDSEG_ERR_ARG", -2: "CDSEG_ERR_LAUNCH", -3: "CDSEG_ERR_WORKSPACE", -4: "CDSEG_ERR_UNSUPPORTED"}


class CdsegError(RuntimeError):
    pass


class DuplicateVoxelsError(CdsegError):
    """Several input points share one (batch element, grid_coord) voxel.  `count` = the surplus points."""

    def __init__(self, msg, count):
        super().__init__(msg)
        self.count = int(count)


class GemmArgs(Structure):
    _fields_ = [
        ("A", c_void_p), ("W", c_void_p), ("bias", c_void_p), ("scale", c_void_p), ("shift", c_void_p),
        ("res", c_void_p), ("add_src", c_void_p), ("add_idx", c_void_p), ("nbr", c_void_p), ("out_idx", c_void_p),
        ("out", c_void_p), ("out2", c_void_p),
        ("M", c_long),
        ("N", c_int), ("K", c_int), ("kvol", c_int),
        ("lda", c_int), ("ldo", c_int), ("ldo2", c_int), ("ldres", c_int), ("ldadd", c_int),
        ("a_dtype", c_int), ("compute_dtype", c_int), ("out_dtype", c_int), ("out2_dtype", c_int),
        ("act", c_int), ("out2_pre_add", c_int),
        ("ws", c_void_p), ("ws_bytes", c_size_t),
        ("colbias", c_void_p), ("ln_pre_g", c_void_p), ("ln_pre_b", c_void_p), ("ln_post_g", c_void_p),
        ("ln_post_b", c_void_p), ("ln_out", c_void_p), ("ldln", c_int), ("ln_out_dtype", c_int), ("ln_eps", c_float),
        ("nbr_kmajor", c_int),
    ]


class BlockDesc(Structure):
    _fields_ = [
        ("dtype", c_int), ("channels", c_int), ("heads", c_int), ("hidden", c_int),
        ("attn_scale", c_float), ("ln_eps", c_float),
        ("cpe_conv_w", c_void_p), ("cpe_conv_b", c_void_p), ("cpe_lin_w", c_void_p), ("cpe_lin_b", c_void_p),
        ("cpe_ln_g", c_void_p), ("cpe_ln_b", c_void_p), ("norm1_g", c_void_p), ("norm1_b", c_void_p),
        ("qkv_w", c_void_p), ("qkv_b", c_void_p), ("proj_w", c_void_p), ("proj_b", c_void_p),
        ("norm2_g", c_void_p), ("norm2_b", c_void_p), ("fc1_w", c_void_p), ("fc1_b", c_void_p),
        ("fc2_w", c_void_p), ("fc2_b", c_void_p), ("cpe_conv_wimg", c_void_p),
        ("head_img", c_void_p), ("tail_img", c_void_p),
        ("attn_flags", c_int),
    ]


class BlockIO(Structure):
    _fields_ = [
        ("n", c_long), ("x", c_void_p), ("xc_in", c_void_p), ("xc_out", c_void_p), ("tbias", c_void_p),
        ("nbr", c_void_p), ("gidx", c_void_p), ("widx", c_void_p), ("patch_start", c_void_p),
        ("num_patches", c_int), ("max_len", c_int), ("scratch", c_void_p), ("scratch_bytes", c_size_t),
        ("sat_counter", c_void_p),
    ]


class PlanSpec(Structure):
    _fields_ = [
        ("nlev", c_int), ("cum", c_int * 9), ("ncurve", c_int), ("curve_rows", c_int * 3),
        ("nslot_curve", c_int), ("slot_curve", c_int * 4),
        ("nlink", c_int), ("link_a", c_int * 16), ("link_b", c_int * 16),
        ("npad", c_int), ("pad_patch", c_int * 4), ("pad_flash", c_int * 4),
    ]


class PlanBeginIO(Structure):
    _fields_ = [
        ("grid", c_void_p), ("grid_elem_bytes", c_int), ("offset", c_void_p), ("nb", c_int), ("n", c_long),
        ("depth", c_int), ("end_bit", c_int), ("i32", c_void_p), ("i64", c_void_p), ("ws", c_void_p),
        ("ws_bytes", c_size_t), ("gmax_host", c_void_p), ("meta_host", c_void_p),
    ]


class PlanFinishIO(Structure):
    _fields_ = [
        ("n", c_long), ("nb", c_int), ("depth", c_int), ("m_host", POINTER(c_long)), ("offs_host", POINTER(c_int)),
        ("grid0", c_void_p), ("bat0", c_void_p), ("code0", c_void_p), ("cluster", c_void_p), ("seg", c_void_p),
        ("orders0", c_void_p), ("i32", c_void_p), ("i64", c_void_p), ("ws", c_void_p), ("ws_bytes", c_size_t),
        ("pads_host", c_void_p),
    ]


# name -> (restype, argtypes); every symbol include/cdseg.h declares
SIGNATURES = {
    "cdseg_abi_version": (c_int, []),
    "cdseg_build_info": (c_char_p, []),
    "cdseg_grid_max": (c_int, [c_void_p, c_int, c_long, c_void_p, c_void_p]),
    "cdseg_offset2batch": (c_int, [c_void_p, c_int, c_long, c_void_p, c_void_p]),
    "cdseg_encode": (c_int, [c_void_p, c_int, c_void_p, c_int, c_long, c_int, c_int, c_void_p, c_void_p]),
    "cdseg_encode4": (c_int, [c_void_p, c_void_p, c_long, c_int, c_void_p, c_void_p]),
    "cdseg_sort_ws_bytes": (c_size_t, [c_long]),
    "cdseg_sort_pairs": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p, c_size_t, c_void_p]),
    "cdseg_sort_curves_ws_bytes": (c_size_t, [c_long, c_int]),
    "cdseg_sort_curves": (c_int, [c_void_p, POINTER(c_int), c_int, c_long, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "cdseg_invert_perm": (c_int, [c_void_p, c_long, c_void_p, c_void_p]),
    "cdseg_widen_i32": (c_int, [c_void_p, c_long, c_void_p, c_void_p]),
    "cdseg_gather_rows": (c_int, [c_void_p, c_void_p, c_long, c_int, c_void_p, c_void_p]),
    "cdseg_scatter_rows": (c_int, [c_void_p, c_void_p, c_long, c_int, c_void_p, c_void_p]),
    "cdseg_gather_i32": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p]),
    "cdseg_plan_gather_grid": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_long, c_int, c_void_p, c_void_p, c_void_p]),
    "cdseg_pool_level": (c_int, [c_void_p, c_long, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "cdseg_pool_gather": (c_int, [c_void_p, c_long, c_long, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p]),
    "cdseg_pool_levels_ws_bytes": (c_size_t, [c_long, c_int]),
    "cdseg_pool_levels": (c_int, [c_void_p, c_long, POINTER(c_int), c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_size_t, c_void_p]),
    "cdseg_link_derive": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_void_p]),
    "cdseg_coarse_orders_ws_bytes": (c_size_t, [c_long, c_int, c_int]),
    "cdseg_coarse_orders": (c_int, [POINTER(c_void_p), c_int, POINTER(c_void_p), c_int, c_long, c_void_p, c_void_p,
                                    c_size_t, c_void_p]),
    "cdseg_nbr_table": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p]),
    "cdseg_nbr_table_from_parent": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_long, c_int, c_int,
                                            c_int, c_void_p, c_void_p]),
    "cdseg_nbr_table_from_info": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_long, c_int, c_int, c_int, c_void_p,
                                          c_void_p]),
    "cdseg_pad_plan": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_long, c_void_p, c_void_p, c_void_p]),
    "cdseg_pad_plan_batch": (c_int, [c_int, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int),
                                     POINTER(c_long), c_int, c_void_p, c_void_p, c_void_p]),
    "cdseg_plan_begin_layout": (c_int, [POINTER(PlanSpec), c_long, c_int, POINTER(c_long), POINTER(c_long)]),
    "cdseg_plan_begin": (c_int, [POINTER(PlanSpec), POINTER(PlanBeginIO), c_int, c_void_p]),
    "cdseg_plan_finish_layout": (c_int, [POINTER(PlanSpec), c_long, c_int, POINTER(c_long), POINTER(c_int), POINTER(c_long),
                                         POINTER(c_long)]),
    "cdseg_plan_finish": (c_int, [POINTER(PlanSpec), POINTER(PlanFinishIO), c_void_p]),
    "cdseg_voxelize": (c_int, [c_void_p, ctypes.c_double, c_long, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cdseg_voxelize_f64": (c_int, [c_void_p, ctypes.c_double, c_long, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cdseg_center_shift": (c_int, [c_void_p, c_int, c_long, c_int, c_void_p, c_void_p, c_void_p]),
    "cdseg_tta_apply": (c_int, [c_void_p, c_long, POINTER(ctypes.c_double), ctypes.c_double, c_int, c_int, c_void_p, c_void_p]),
    "cdseg_div_add": (c_int, [c_void_p, c_float, c_float, c_long, c_void_p, c_void_p]),
    "cdseg_collect_feat": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_long, c_void_p, c_void_p]),
    "cdseg_max_run": (c_int, [c_void_p, c_long, c_void_p, c_void_p]),
    "cdseg_fragment_select": (c_int, [c_void_p, c_void_p, c_long, c_int, c_void_p, c_void_p]),
    "cdseg_softmax_vote": (c_int, [c_void_p, c_int, c_void_p, c_long, c_int, c_void_p, c_int, c_void_p]),
    "cdseg_argmax_rows": (c_int, [c_void_p, c_int, c_long, c_int, c_void_p, c_void_p]),
    "cdseg_knn": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_long, c_int, c_int, POINTER(c_float), c_float,
                          c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "cdseg_knn1_ws_bytes": (c_size_t, [c_long]),
    "cdseg_knn1": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_long, c_int, POINTER(c_float), c_float,
                           c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "cdseg_iou_counts": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_void_p, c_void_p]),
    "cdseg_gemm": (c_int, [POINTER(GemmArgs), c_void_p]),
    "cdseg_mlp_fused": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                c_long, c_int, c_int, c_void_p]),
    "cdseg_attn_tail_fused": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_long, c_int, c_int, c_void_p]),
    "cdseg_cpe_head_fused": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                     c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int, c_long, c_int, c_int,
                                     c_int, c_void_p]),
    "cdseg_subm_conv3_wimg_bytes": (c_size_t, [c_int]),
    "cdseg_subm_conv3_pack": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "cdseg_subm_conv3": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p, c_int, c_void_p]),
    "cdseg_child_info": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p]),
    "cdseg_stem5_wimg_bytes": (c_size_t, []),
    "cdseg_stem5_pack": (c_int, [c_void_p, c_void_p, c_void_p]),
    "cdseg_stem5": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_long,
                            c_int, c_void_p, c_void_p, c_void_p]),
    "cdseg_block_rr_img_bytes": (c_size_t, [c_int, c_int]),
    "cdseg_block_rr_pack": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cdseg_cpe_head_rr": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                  c_void_p, c_float, c_void_p, c_void_p, c_int, c_long, c_int, c_int, c_void_p]),
    "cdseg_attn_tail_rr": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                   c_void_p, c_int, c_void_p, c_int, c_long, c_int, c_void_p]),
    "cdseg_cpe_head_rr2": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                   c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_long, c_int, c_int, c_void_p]),
    "cdseg_attn_tail_rr2": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                    c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_long, c_int, c_void_p, c_size_t, c_void_p]),
    "cdseg_block_scratch_bytes": (c_size_t, [POINTER(BlockDesc), c_long]),
    "cdseg_block_forward": (c_int, [POINTER(BlockDesc), POINTER(BlockIO), c_void_p]),
    "cdseg_layernorm": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_int, c_void_p,
                                c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_long, c_int, c_void_p]),
    "cdseg_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_int, c_int, c_void_p]),
    "cdseg_attention_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_int, c_int, c_int, c_void_p]),
    "cdseg_attention_schedule": (c_long, [c_int, c_int, c_int, c_int, c_void_p, c_long]),
    "cdseg_attention_bwd_ws_bytes": (c_size_t, [c_long, c_int]),
    "cdseg_attention_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_int, c_int, c_long, c_int, c_float, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                    c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "cdseg_layernorm_bwd": (c_int, [c_void_p, c_int, c_void_p, c_float, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p,
                                    c_void_p, c_long, c_int, c_void_p]),
    "cdseg_gelu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_void_p]),
    "cdseg_conv_wgrad": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_long, c_int, c_int, c_void_p, c_void_p,
                                 c_void_p]),
    "cdseg_linear_wgrad": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_long, c_int, c_int, c_void_p, c_int, c_void_p,
                                   c_void_p]),
    "cdseg_prof_enable": (c_int, [c_int]),
    "cdseg_prof_summary": (c_int, [POINTER(ctypes.c_double), POINTER(c_long)]),
    "cdseg_prof_summary_class": (c_int, [c_int, POINTER(ctypes.c_double), POINTER(c_long)]),
    "cdseg_segment_max": (c_int, [c_void_p, c_int, c_int, c_void_p, c_long, c_int, c_void_p, c_void_p, c_int,
                                  c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "cdseg_segment_mean": (c_int, [c_void_p, c_int, c_void_p, c_long, c_int, c_void_p, c_int, c_void_p]),
    "cdseg_pool_fused_img_bytes": (c_size_t, [c_int, c_int]),
    "cdseg_pool_fused_pack": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "cdseg_pool_fused": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                 c_void_p, c_int, c_int, c_int, c_void_p]),
    "cdseg_gemv": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "cdseg_randn": (c_int, [c_void_p, c_long, c_uint64, c_uint64, c_void_p]),
    "cdseg_cast": (c_int, [c_void_p, c_int, c_void_p, c_int, c_long, c_void_p]),
    "cdseg_gather_pad_cast": (c_int, [c_void_p, c_int, c_void_p, c_long, c_int, c_int, c_void_p, c_int, c_void_p]),
    "cdseg_ddim_update": (c_int, [c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_int, c_void_p, c_long,
                                  c_void_p]),
    "cdseg_axpy": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_long, c_void_p]),
    "cdseg_count_saturated": (c_int, [c_void_p, c_long, c_int, c_int, c_void_p, c_void_p]),
    "cdseg_subm_conv3_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p, c_int, c_float, c_int,
                                     c_void_p]),
    "cdseg_split16": (c_int, [c_void_p, c_int, c_long, c_int, c_void_p, c_void_p, c_int, c_float, c_void_p]),
}

_libs = {}
_tls = threading.local()  # .variant: which build the calling host thread's ops go to ("bf16" default, "f16")
VARIANTS = ("bf16", "f16")


def active():
    return getattr(_tls, "variant", "bf16")


class use:
    """`with _lib.use("f16"):` routes the calling thread's ops to the IEEE-half build of the library."""

    def __init__(self, variant):
        if variant not in VARIANTS:
            raise ValueError(variant)
        self.variant = variant

    def __enter__(self):
        self.prev = active()
        _tls.variant = self.variant
        return self

    def __exit__(self, *exc):
        _tls.variant = self.prev
        return False


def activate(variant):
    """Make `variant` the calling thread's build from here on (a process that runs one precision throughout, e.g. bench.py)."""
    if variant not in VARIANTS:
        raise ValueError(variant)
    _tls.variant = variant


def load(variant=None):
    """Load the HIP library (the calling thread's active build); raises CdsegError (never falls back) if it is absent."""
    v = variant or active()
    lib = _libs.get(v)
    if lib is not None:
        return lib
    path = LIB_PATH if v == "bf16" else LIB_PATH_F16
    if not os.path.exists(path):
        raise CdsegError(
            f"{path} is missing: the MI355X HIP extension was not built. "
            "Run `python -m cdsegnet_amd.build` (hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    _libs[v] = lib
    return lib


def check(status, what):
    if status != 0:
        raise CdsegError(f"{what} failed: {ERRORS.get(status, status)}")
