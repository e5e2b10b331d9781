"""ctypes binding of ``libvmv_hip_{f16,bf16}.so`` (the C ABI declared in ``include/vmv.h``).

The structures below mirror the header field-for-field.  Loading fails loudly (``RuntimeError``) when the
shared library is missing: there is NO CPU / PyTorch fallback for the hot path.

The kernels are built once per 16-bit element type (storage + MFMA operands; fp32 accumulate in both).  A process uses
ONE of them: ``VMV_DTYPE`` = ``fp16`` (default — 11 significand bits, what the stated parity tolerances need, and the
reference's own half mode: ``use_fp16`` / autocast) or ``bf16``; ``set_elem()`` overrides it until the first ``load()``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_ELEM_NAMES = {"fp16": "f16", "f16": "f16", "float16": "f16", "half": "f16", "bf16": "bf16", "bfloat16": "bf16"}
_elem = _ELEM_NAMES[os.environ.get("VMV_DTYPE", "fp16").lower()]
ELEM_F16, ELEM_BF16 = 0, 1


def set_elem(name: str):
    """Choose the element type ("fp16" | "bf16") before the library is first loaded."""
    global _elem
    new = _ELEM_NAMES[str(name).lower()]
    if _lib is not None and new != _elem:
        raise RuntimeError(f"the {_elem} kernels are already loaded; set VMV_DTYPE / call set_elem() before first use")
    _elem = new


def elem_name() -> str:
    """"fp16" | "bf16" """
    return "fp16" if _elem == "f16" else "bf16"


def elem():
    """torch dtype of the 16-bit storage type of the loaded / selected library."""
    import torch
    return torch.float16 if _elem == "f16" else torch.bfloat16


def lib_path() -> str:
    # VMV_LIB_DIR: an alternative build of the same ABI (A/B experiments: tools/experiments/*_ab.sh)
    return os.path.join(os.environ.get("VMV_LIB_DIR") or os.path.join(_HERE, "lib"), f"libvmv_hip_{_elem}.so")

VMV_MAX_SEGS = 24
SEG_LINEAR, SEG_SPATIAL, SEG_TEMPORAL = 0, 1, 2
EPI_NONE, EPI_GEGLU, EPI_TATTN = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2
TILE_AUTO, TILE_128x128, TILE_128x160, TILE_128x64, TILE_64x64, TILE_256x128, TILE_256x160 = 0, 1, 2, 3, 4, 5, 6
TILE_G128x128, TILE_G128x160, TILE_P256x128, TILE_P256x160, TILE_PP256x128, TILE_PP256x160 = 7, 8, 9, 10, 11, 12
TILE_Q128x128, TILE_Q96x160 = 13, 14
TILE_S256x128, TILE_S192x160, TILE_S256x160 = 15, 16, 17
TILE_A128x160, TILE_A128x128 = 18, 19
TILE_X256x320, TILE_X256x256, TILE_X256x128 = 20, 21, 22
TILE_RS, TILE_RS512, TILE_RS256, TILE_HALO, TILE_TFR, TILE_TQA, TILE_W256x256, TILE_X512x128, TILE_Y256x128 = 23, 24, 25, 26, 27, 28, 29, 30, 31
OP_GEMM, OP_GN_STATS, OP_GN_APPLY, OP_LAYERNORM, OP_ATTENTION, OP_SOFTMAX, OP_COPY, OP_GN_FUSED, OP_FF, OP_GN_TABLE, OP_COMM = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11
COMM_ALL_TO_ALL, COMM_ALL_GATHER, COMM_ID_BYTES = 0, 1, 128
ABI_VERSION = 11
GN_FUSED_BYTES = 131072


class GemmSeg(C.Structure):
    _fields_ = [("src", C.c_void_p), ("ld", C.c_int32), ("k", C.c_int32), ("mode", C.c_int32),
                ("d0", C.c_int32), ("d1", C.c_int32), ("_pad", C.c_int32)]


class GemmParams(C.Structure):
    _fields_ = [("M", C.c_int32), ("N", C.c_int32), ("nseg", C.c_int32), ("ktot", C.c_int32),
                ("seg", GemmSeg * VMV_MAX_SEGS),
                ("W", C.c_void_p), ("bias", C.c_void_p), ("rowvec", C.c_void_p),
                ("rowvec_div", C.c_int32), ("rowvec_ld", C.c_int32),
                ("residual", C.c_void_p), ("ldr", C.c_int32),
                ("epilogue", C.c_int32), ("act", C.c_int32), ("out_fp32", C.c_int32),
                ("out", C.c_void_p), ("ldo", C.c_int32),
                ("OH", C.c_int32), ("OW", C.c_int32), ("IH", C.c_int32), ("IW", C.c_int32),
                ("stride", C.c_int32), ("ups", C.c_int32),
                ("F", C.c_int32), ("P", C.c_int32),
                ("ksplit", C.c_int32), ("workspace", C.c_void_p),
                ("tile", C.c_int32), ("res_scale", C.c_float), ("rowstat", C.c_void_p), ("colsum", C.c_void_p),
                ("ln_eps", C.c_float), ("wgroup_rows", C.c_int32), ("wgroup_stride", C.c_int64),
                ("gn_table", C.c_void_p), ("gn_rows_per_stat", C.c_int32), ("gn_silu", C.c_int32), ("epi_scale", C.c_float)]


class GroupNormParams(C.Structure):
    _fields_ = [("x", C.c_void_p), ("x1", C.c_void_p), ("ld", C.c_int32), ("ld1", C.c_int32),
                ("C0", C.c_int32), ("C1", C.c_int32), ("rows", C.c_int32), ("rows_per_stat", C.c_int32),
                ("chunk_rows", C.c_int32), ("partial", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("eps", C.c_float), ("silu", C.c_int32), ("y", C.c_void_p), ("ldy", C.c_int32), ("fold_ranks", C.c_int32),
                ("totals", C.c_void_p), ("totals_clear", C.c_void_p), ("clear_count", C.c_int32), ("_pad", C.c_int32)]


class FfParams(C.Structure):
    _fields_ = [("M", C.c_int32), ("C", C.c_int32), ("x", C.c_void_p), ("ldx", C.c_int32), ("_pad0", C.c_int32),
                ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
                ("residual", C.c_void_p), ("ldr", C.c_int32), ("ln_eps", C.c_float),
                ("out", C.c_void_p), ("ldo", C.c_int32), ("_pad1", C.c_int32)]


class GsParams(C.Structure):
    _fields_ = [("gaussians", C.c_void_p), ("N", C.c_int32), ("size", C.c_int32), ("view", C.c_void_p),
                ("view_proj", C.c_void_p), ("tan_half_fov", C.c_float), ("bg", C.c_float * 3),
                ("depth", C.c_void_p), ("xy", C.c_void_p), ("conic_opacity", C.c_void_p), ("rect", C.c_void_p),
                ("tiles_touched", C.c_void_p), ("offsets", C.c_void_p), ("scan_temp", C.c_void_p),
                ("scan_temp_bytes", C.c_size_t), ("keys", C.c_void_p), ("keys_sorted", C.c_void_p), ("vals", C.c_void_p),
                ("vals_sorted", C.c_void_p), ("num_rendered", C.c_int32), ("_pad", C.c_int32), ("sort_temp", C.c_void_p),
                ("sort_temp_bytes", C.c_size_t), ("ranges", C.c_void_p), ("out_color", C.c_void_p), ("out_alpha", C.c_void_p)]


class GsBatchParams(C.Structure):
    _fields_ = [("gaussians", C.c_void_p), ("B", C.c_int32), ("N", C.c_int32), ("V", C.c_int32), ("size", C.c_int32),
                ("views", C.c_void_p), ("view_projs", C.c_void_p), ("tan_half_fov", C.c_float), ("bg", C.c_float * 3),
                ("depth", C.c_void_p), ("xy", C.c_void_p), ("conic_opacity", C.c_void_p), ("rect", C.c_void_p),
                ("tiles_touched", C.c_void_p), ("offsets", C.c_void_p), ("scan_temp", C.c_void_p),
                ("scan_temp_bytes", C.c_size_t), ("keys", C.c_void_p), ("keys_sorted", C.c_void_p), ("vals", C.c_void_p),
                ("vals_sorted", C.c_void_p), ("num_rendered", C.c_int32), ("_pad", C.c_int32), ("sort_temp", C.c_void_p),
                ("sort_temp_bytes", C.c_size_t), ("ranges", C.c_void_p), ("out_color", C.c_void_p), ("out_alpha", C.c_void_p)]


class CopyParams(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("n0", C.c_int32), ("n1", C.c_int32), ("n2", C.c_int32),
                ("inner16", C.c_int32), ("ss0", C.c_int64), ("ss1", C.c_int64), ("ss2", C.c_int64)]


class CommParams(C.Structure):
    _fields_ = [("comm", C.c_void_p), ("kind", C.c_int32), ("_pad", C.c_int32), ("send", C.c_void_p), ("recv", C.c_void_p),
                ("bytes", C.c_int64)]


class LayerNormParams(C.Structure):
    _fields_ = [("x", C.c_void_p), ("ldx", C.c_int32), ("y", C.c_void_p), ("ldy", C.c_int32),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("rows", C.c_int32), ("C", C.c_int32),
                ("eps", C.c_float), ("_pad", C.c_int32), ("stats_out", C.c_void_p)]


class SeqMap(C.Structure):
    _fields_ = [("s_outer", C.c_int64), ("s_inner", C.c_int64), ("s_row", C.c_int64),
                ("inner", C.c_int32), ("_pad", C.c_int32)]


class AttnParams(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p),
                ("qm", SeqMap), ("km", SeqMap), ("vm", SeqMap), ("om", SeqMap),
                ("n_outer", C.c_int32), ("kv_div", C.c_int32), ("heads", C.c_int32),
                ("Nq", C.c_int32), ("Nk", C.c_int32), ("scale", C.c_float), ("head_dim", C.c_int32), ("causal", C.c_int32)]


class SoftmaxParams(C.Structure):
    _fields_ = [("s", C.c_void_p), ("lds", C.c_int32), ("p", C.c_void_p), ("ldp", C.c_int32),
                ("rows", C.c_int32), ("n", C.c_int32), ("scale", C.c_float), ("_pad", C.c_int32)]


class DdimParams(C.Structure):
    _fields_ = [("eps_rows", C.c_void_p), ("ld", C.c_int32), ("C", C.c_int32), ("F", C.c_int32), ("HW", C.c_int32),
                ("guide_scale", C.c_float), ("c_recip", C.c_float), ("c_recipm1", C.c_float),
                ("c_sqrt_ac", C.c_float), ("c_sqrt_1mac", C.c_float), ("a_prev", C.c_float),
                ("v_pred", C.c_int32), ("xt", C.c_void_p), ("x0_out", C.c_void_p),
                ("clamp", C.c_float), ("sigma", C.c_float), ("noise", C.c_void_p)]


# every symbol include/vmv.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "vmv_abi_version": (C.c_int, []),
    "vmv_elem_type": (C.c_int, []),
    "vmv_sizeof": (C.c_int, [C.c_int]),
    "vmv_error_string": (C.c_char_p, [C.c_int]),
    "vmv_gemm": (C.c_int, [C.POINTER(GemmParams), _P]),
    "vmv_gemm_ln_inline_ok": (C.c_int, [C.POINTER(GemmParams)]),
    "vmv_gemm_rs_ok": (C.c_int, [C.POINTER(GemmParams)]),
    "vmv_gemm_tfr_ok": (C.c_int, [C.POINTER(GemmParams)]),
    "vmv_gemm_tqa_ok": (C.c_int, [C.POINTER(GemmParams)]),
    "vmv_has_experiments": (C.c_int, []),
    "vmv_ff_fused": (C.c_int, [C.POINTER(FfParams), _P]),
    "vmv_ff_fused_ok": (C.c_int, [C.POINTER(FfParams)]),
    "vmv_gemm_pick_tile": (C.c_int, [C.POINTER(GemmParams)]),
    "vmv_gemm_validate": (C.c_int, [C.POINTER(GemmParams)]),
    "vmv_groupnorm_stats": (C.c_int, [C.POINTER(GroupNormParams), _P]),
    "vmv_groupnorm_apply": (C.c_int, [C.POINTER(GroupNormParams), _P]),
    "vmv_groupnorm_table": (C.c_int, [C.POINTER(GroupNormParams), _P]),
    "vmv_groupnorm_fused": (C.c_int, [C.POINTER(GroupNormParams), C.c_int32, _P]),
    "vmv_layernorm": (C.c_int, [C.POINTER(LayerNormParams), _P]),
    "vmv_attention": (C.c_int, [C.POINTER(AttnParams), _P]),
    "vmv_softmax_rows": (C.c_int, [C.POINTER(SoftmaxParams), _P]),
    "vmv_permute_copy": (C.c_int, [C.POINTER(CopyParams), _P]),
    "vmv_gaussian_activation": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, _P]),
    "vmv_lgm_x0_views": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, C.c_float, C.c_float, C.c_float, _P, _P]),
    "vmv_lgm_pack_input": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P]),
    "vmv_lgm_render_to_vae": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "vmv_ddim_x0_step": (C.c_int, [_P, _P, _P, C.c_long, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P, _P]),
    "vmv_gs_workspace_bytes": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "vmv_gs_preprocess": (C.c_int, [C.POINTER(GsParams), _P]),
    "vmv_gs_render": (C.c_int, [C.POINTER(GsParams), _P]),
    "vmv_gs_batch_key_bits": (C.c_int, [C.c_int, C.c_int]),
    "vmv_gs_batch_workspace_bytes": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "vmv_gs_batch_preprocess": (C.c_int, [C.POINTER(GsBatchParams), _P]),
    "vmv_gs_batch_render": (C.c_int, [C.POINTER(GsBatchParams), _P]),
    "vmv_latent_to_rows": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "vmv_latent_to_rows_keep": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "vmv_i2v_temporal_adapter": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_float, _P]),
    "vmv_adaptive_avgpool_rows": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "vmv_rows_to_nchw": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P]),
    "vmv_cfg_ddim_step": (C.c_int, [C.POINTER(DdimParams), _P]),
    "vmv_posterior_sample": (C.c_int, [_P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, _P]),
    "vmv_emb_combine_silu": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "vmv_sinusoidal": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "vmv_plan_create": (_P, []),
    "vmv_plan_destroy": (None, [_P]),
    "vmv_plan_add": (C.c_int, [_P, C.c_int, _P, C.c_size_t]),
    "vmv_plan_size": (C.c_int, [_P]),
    "vmv_plan_run": (C.c_int, [_P, _P]),
    "vmv_plan_run_range": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "vmv_plan_capture": (_P, [_P, _P]),
    "vmv_graph_launch": (C.c_int, [_P, _P]),
    "vmv_graph_nodes": (C.c_int, [_P]),
    "vmv_graph_destroy": (None, [_P]),
    "vmv_comm_load": (C.c_int, [C.c_char_p]),
    "vmv_comm_loaded": (C.c_int, []),
    "vmv_comm_unique_id": (C.c_int, [_P]),
    "vmv_comm_create": (_P, [_P, C.c_int, C.c_int]),
    "vmv_comm_create_sim": (_P, [C.c_int, C.c_int]),
    "vmv_comm_destroy": (None, [_P]),
    "vmv_comm_world": (C.c_int, [_P]),
    "vmv_comm_rank": (C.c_int, [_P]),
    "vmv_comm_is_sim": (C.c_int, [_P]),
    "vmv_comm_run": (C.c_int, [C.POINTER(CommParams), _P]),
}

_lib = None


def load():
    """Load (once) and type the shared library.  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    LIB_PATH = lib_path()
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; "
            f"g.build()' or make -C videomv_amd/csrc).  There is no CPU fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI drifted
        fn.restype = res
        fn.argtypes = args
    if lib.vmv_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH}: ABI version mismatch")
    if lib.vmv_elem_type() != (ELEM_F16 if _elem == "f16" else ELEM_BF16):
        raise RuntimeError(f"{LIB_PATH} was built for another element type")
    for which, st in ((OP_GEMM, GemmParams), (OP_GN_STATS, GroupNormParams), (OP_LAYERNORM, LayerNormParams),
                      (OP_ATTENTION, AttnParams), (OP_SOFTMAX, SoftmaxParams), (OP_COPY, CopyParams), (OP_FF, FfParams), (OP_COMM, CommParams), (103, GsParams), (104, GsBatchParams), (100, DdimParams), (101, GemmSeg), (102, SeqMap)):
        if lib.vmv_sizeof(which) != C.sizeof(st):
            raise RuntimeError(f"struct layout drift for {st.__name__}: C {lib.vmv_sizeof(which)} vs ctypes "
                               f"{C.sizeof(st)}")
    _lib = lib
    return lib


class VmvError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        msg = load().vmv_error_string(rc)
        raise VmvError(f"{what}: rc={rc} ({msg.decode() if msg else '?'})")
