"""ctypes binding of libsais_hip.so (the C ABI declared in include/sais_hip.h).

The product path has NO fallback: if the HIP library is missing or a kernel launch fails, this
module raises.  (tests/ use oracle/ as the checker; nothing here imports it.)
"""
import ctypes
import os

import torch  # noqa: F401  -- must come first: libsais_hip.so has to bind to the HIP runtime torch ships,
#                             otherwise two runtimes coexist and launches fail with "no ROCm-capable device"

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SAIS_HIP_LIB") or os.path.join(_HERE, "libsais_hip.so")      # override: profiling builds

c_void_p, c_int, c_long, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float


class SaisGemm(ctypes.Structure):
    _fields_ = [("A", c_void_p), ("lda", c_int), ("B", c_void_p), ("ldb", c_int),
                ("M", c_int), ("N", c_int), ("K", c_int), ("epilogue", c_int),
                ("bias", c_void_p), ("out", c_void_p), ("ldo", c_int),
                ("out2", c_void_p), ("ldo2", c_int), ("aux", c_void_p), ("ldaux", c_int),
                ("grp_in", c_int), ("grp_out", c_int), ("grp_off", c_int), ("rowscale", c_void_p),
                ("p_drop", c_float), ("rng_state", c_void_p), ("site", ctypes.c_uint)]


class SaisTGemm(ctypes.Structure):
    _fields_ = [("A", c_void_p), ("lda", c_long), ("W", c_void_p), ("ldw", c_long), ("M", c_int), ("N", c_int),
                ("K", c_int), ("epilogue", c_int), ("nsplit", c_int), ("bias", c_void_p), ("aux", c_void_p),
                ("ldaux", c_long), ("out", c_void_p), ("ldo", c_long), ("p_drop", c_float), ("rng_state", c_void_p),
                ("site", ctypes.c_uint)]


TN_MAX_ITEMS = 48            # SAIS_TN_MAX_ITEMS


class SaisTnItem(ctypes.Structure):
    _fields_ = [("P", c_void_p), ("ldp", c_int), ("Q", c_void_p), ("ldq", c_int), ("N1", c_int), ("N2", c_int),
                ("dW", c_void_p), ("ldw", c_int), ("db", c_void_p)]


class SaisGemmLn(ctypes.Structure):
    _fields_ = [("A", c_void_p), ("lda", c_int), ("W", c_void_p), ("ldw", c_int), ("M", c_int), ("K", c_int),
                ("bias", c_void_p), ("resid", c_void_p), ("ldr", c_int), ("out32", c_void_p), ("ldo32", c_int),
                ("out16", c_void_p), ("ldo16", c_int), ("gamma", c_void_p), ("beta", c_void_p), ("eps", c_float),
                ("mean", c_void_p), ("rstd", c_void_p), ("dres", c_void_p), ("lddres", c_int),
                ("dgamma", c_void_p), ("dbeta", c_void_p), ("rowscale", c_void_p), ("rowscale16", c_void_p),
                ("dres_period", c_int), ("xn16", c_void_p), ("ldxn16", c_int)]


class SaisMlp(ctypes.Structure):
    _fields_ = [("X", c_void_p), ("ldx", c_int), ("W1", c_void_p), ("ldw1", c_int), ("bias1", c_void_p),
                ("W2", c_void_p), ("ldw2", c_int), ("M", c_int), ("H", c_int), ("h", c_void_p), ("ldh", c_int),
                ("g", c_void_p), ("ldg", c_int), ("tail", SaisGemmLn)]


class SaisVitBlockParams(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in (
        "qkv_w", "qkv_b", "proj_w", "proj_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "norm1_g", "norm1_b", "norm2_g", "norm2_b",
        "next_norm_g", "next_norm_b", "qkv_wt", "proj_wt", "fc1_wt", "fc2_wt", "d_qkv_w", "d_qkv_b", "d_proj_w", "d_proj_b",
        "d_fc1_w", "d_fc1_b", "d_fc2_w", "d_fc2_b", "d_norm1_g", "d_norm1_b", "d_norm2_g", "d_norm2_b")]


class SaisVitBlockFwd(ctypes.Structure):
    _fields_ = [("frames", c_int), ("ntok", c_int)] + [(n, c_void_p) for n in (
        "xn1", "x_in", "qkv", "attn_out", "lse", "x_mid", "xn2", "mean2", "rstd2", "h", "gelu_grad", "x_out", "xn_next",
        "mean_next", "rstd_next", "rowscale_attn", "rowscale_mlp")]


class SaisVitBlockBwd(ctypes.Structure):
    _fields_ = [("frames", c_int), ("ntok", c_int)] + [(n, c_void_p) for n in (
        "x_in", "mean1", "rstd1", "xn1", "qkv", "attn_out", "lse", "x_mid", "mean2", "rstd2", "xn2", "h", "gelu_grad", "dx",
        "dx16_in", "dx16_out", "rowscale_attn", "rowscale_prev")] + [("defer_dw", c_int)]


class SaisTemporalLayerParams(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in (
        "in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "linear1_w", "linear1_b", "linear2_w", "linear2_b", "norm1_g",
        "norm1_b", "norm2_g", "norm2_b", "in_proj_wt", "out_proj_wt", "linear1_wt", "linear2_wt", "d_in_proj_w", "d_in_proj_b",
        "d_out_proj_w", "d_out_proj_b", "d_linear1_w", "d_linear1_b", "d_linear2_w", "d_linear2_b", "d_norm1_g", "d_norm1_b",
        "d_norm2_g", "d_norm2_b")]


class SaisTemporalLayerFwd(ctypes.Structure):
    _fields_ = [("B", c_int), ("S", c_int)] + [(n, c_void_p) for n in (
        "z", "key_pad", "qkv", "ctx", "attn_avg", "y1", "z1", "mean1", "rstd1", "h", "y2", "z_out", "mean2", "rstd2")] + \
        [("p_drop", c_float), ("rng_state", c_void_p), ("site0", ctypes.c_uint)]


class SaisTemporalLayerBwd(ctypes.Structure):
    _fields_ = [("B", c_int), ("S", c_int)] + [(n, c_void_p) for n in (
        "z", "qkv", "ctx", "y1", "mean1", "rstd1", "z1", "h", "y2", "mean2", "rstd2", "key_pad", "dz_slabs")] + \
        [("nslab", c_int), ("slab_stride", c_long), ("dz_add", c_void_p), ("dx_slabs", c_void_p), ("dx_add", c_void_p),
         ("p_drop", c_float), ("rng_state", c_void_p), ("site0", ctypes.c_uint), ("dw_items_out", c_void_p)]


OP_VIT_BLOCK_FWD, OP_VIT_BLOCK_BWD, OP_TEMPORAL_LAYER_FWD, OP_TEMPORAL_LAYER_BWD = 0, 1, 2, 3


class SaisOptChunk(ctypes.Structure):
    _fields_ = [("off", c_long), ("len", c_int), ("seg", c_int)]


class SaisAdamW(ctypes.Structure):
    _fields_ = [("param", c_void_p), ("grad", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p),
                ("teacher", c_void_p), ("param16", c_void_p), ("teacher16", c_void_p),
                ("chunks", c_void_p), ("nchunks", c_int), ("seg_flags", c_void_p), ("norms", c_void_p),
                ("clip", c_float), ("lr", c_float), ("weight_decay", c_float), ("beta1", c_float), ("beta2", c_float),
                ("eps", c_float), ("bc1", c_float * 2), ("sqrt_bc2", c_float * 2), ("frozen1", c_int), ("ema_m", c_float), ("grad_scale", c_float)]


OPT_DECAY, OPT_CLASS1, OPT_NO_GRAD = 1, 2, 4

EPI_BIAS_BF16, EPI_BIAS_RELU_BF16, EPI_BIAS_F32, EPI_BIAS_RESID_F32 = 0, 1, 2, 3
EPI_BIAS_GELU_BF16, EPI_DGELU_BF16, EPI_DRELU_BF16, EPI_PATCH_F32 = 4, 5, 6, 7
EPI_BIAS_RELU_F32, EPI_DRELU_F32 = 8, 9
EPI_BIAS_GELU_GRAD_BF16, EPI_MUL_BF16, EPI_RAW_SLABS_F32 = 10, 11, 12
EPI_BIAS_GELU_GRADQ_BF16, EPI_MULQ_BF16 = 13, 14      # GELU' as one-byte codes (ABI 12)
TG_RAW, TG_BIAS, TG_BIAS_RELU, TG_DRELU = 0, 1, 2, 3

# name -> argtypes; every symbol include/sais_hip.h declares (checked by tests/test_abi.py)
SIGNATURES = {
    "sais_abi_version": [],
    "sais_gemm_nt": [ctypes.POINTER(SaisGemm), c_void_p],
    "sais_gemm_nt_f32": [ctypes.POINTER(SaisGemm), c_void_p],
    "sais_splitk_finish": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                           c_int, c_void_p],
    "sais_gemm_ln_fwd": [ctypes.POINTER(SaisGemmLn), c_void_p],
    "sais_gemm_ln_bwd": [ctypes.POINTER(SaisGemmLn), c_void_p],
    "sais_workspace_bytes": [c_int, c_int, c_int],
    "sais_gelu_grad_bytes": [],
    "sais_vit_block_fwd": [ctypes.POINTER(SaisVitBlockParams), ctypes.POINTER(SaisVitBlockFwd), c_void_p, ctypes.c_size_t,
                           c_void_p],
    "sais_vit_block_bwd": [ctypes.POINTER(SaisVitBlockParams), ctypes.POINTER(SaisVitBlockBwd), c_void_p, ctypes.c_size_t,
                           c_void_p],
    "sais_vit_blocks_dw": [ctypes.POINTER(ctypes.POINTER(SaisVitBlockParams)), ctypes.POINTER(ctypes.POINTER(SaisVitBlockBwd)),
                           ctypes.POINTER(c_void_p), ctypes.c_size_t, c_int, ctypes.POINTER(SaisTnItem), c_int, c_void_p],
    "sais_temporal_layer_fwd": [ctypes.POINTER(SaisTemporalLayerParams), ctypes.POINTER(SaisTemporalLayerFwd), c_void_p,
                                ctypes.c_size_t, c_void_p],
    "sais_temporal_layer_bwd": [ctypes.POINTER(SaisTemporalLayerParams), ctypes.POINTER(SaisTemporalLayerBwd), c_void_p,
                                ctypes.c_size_t, c_void_p],
    "sais_mlp_fwd": [ctypes.POINTER(SaisMlp), c_void_p],
    "sais_mlp_bwd": [ctypes.POINTER(SaisMlp), c_void_p],
    "sais_gemm_tn_f32": [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p],
    "sais_transpose_f32": [c_void_p, c_int, c_int, c_void_p, c_void_p],
    "sais_gemm_tn_grouped": [ctypes.POINTER(SaisTnItem), c_int, c_int, c_int, c_void_p],
    "sais_gemm_tn_grouped_f32": [ctypes.POINTER(SaisTnItem), c_int, c_int, c_int, c_void_p],
    "sais_gemm_tn_grouped_ws": [ctypes.POINTER(SaisTnItem), c_int, c_int, c_int, c_void_p, ctypes.c_size_t, c_void_p],
    "sais_gemm_tn_grouped_slab_bytes": [ctypes.POINTER(SaisTnItem), c_int, c_int],
    "sais_gemm_tn": [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p],
    "sais_layernorm_fwd": [c_void_p, c_long, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_long, c_void_p,
                           c_long, c_void_p, c_void_p, c_void_p],
    "sais_layernorm_bwd": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_void_p,
                           c_void_p, c_long, c_int, c_int, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p,
                           c_void_p, c_void_p, c_float, c_void_p, ctypes.c_uint, c_void_p],
    "sais_vit_attn_fwd": [c_void_p, c_long, c_int, c_int, c_void_p, c_long, c_void_p, c_void_p, c_void_p],
    "sais_vit_attn_bwd": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_int, c_int, c_void_p,
                          c_long, c_void_p],
    "sais_vit_attn_cls_fwd": [c_void_p, c_long, c_int, c_int, c_void_p, c_long, c_void_p],
    "sais_vit_attn_cls_bwd": [c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_void_p, c_long, c_void_p],
    "sais_raft_corr_pool": [c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "sais_raft_lookup": [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "sais_im2col_f32": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p],
    "sais_patchify": [c_void_p, c_int, c_int, c_void_p, c_void_p],
    "sais_vit_cls_rows": [c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_void_p],
    "sais_vit_embed_bwd": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "sais_sgd_step": [c_void_p, c_void_p, c_void_p, c_long, c_float, c_float, c_void_p],
    "sais_cast_bf16": [c_void_p, c_void_p, c_long, c_void_p],
    "sais_transpose_cast_bf16": [c_void_p, c_int, c_int, c_void_p, c_void_p],
    "sais_transpose_batch": [c_void_p, c_int, c_int, c_int, c_void_p],
    "sais_preprocess_plan_create": [c_int, c_int, ctypes.c_double, ctypes.c_double, c_void_p, c_void_p,
                                    ctypes.POINTER(c_void_p)],
    "sais_preprocess_plan_box": [c_void_p, c_void_p],
    "sais_preprocess_run": [c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "sais_preprocess_plan_destroy": [c_void_p],
    "sais_scale_f32": [c_void_p, c_long, c_float, c_void_p],
    "sais_touch": [c_void_p, c_long, c_void_p],
    "sais_temporal_prepare_fwd": [c_void_p, c_long, c_long, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                  c_void_p],
    "sais_temporal_prepare_bwd": [c_void_p, c_void_p, c_int, c_long, c_int, c_int, c_void_p, c_long, c_long, c_int, c_void_p,
                                  c_void_p, c_void_p],
    "sais_tgemm": [ctypes.POINTER(SaisTGemm), c_void_p],
    "sais_tgemm_nsplit": [c_int, c_int, c_int],
    "sais_temporal_ln_fwd": [c_void_p, c_int, c_long, c_void_p, c_void_p, c_int, c_float, c_void_p, ctypes.c_uint, c_void_p,
                             c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p],
    "sais_temporal_ln_bwd": [c_void_p, c_int, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                             c_void_p, c_float, c_void_p, ctypes.c_uint, c_void_p, c_void_p, c_void_p],
    "sais_temporal_attn_fwd": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, ctypes.c_uint, c_void_p],
    "sais_temporal_attn_bwd": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_long, c_void_p, c_float, c_void_p,
                               ctypes.c_uint, c_void_p],
    "sais_rng_advance": [c_void_p, c_void_p],
    "sais_droppath_scales": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, ctypes.c_uint, c_void_p],
    "sais_cast_bf16_rows": [c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p],
    "sais_dropout_f32": [c_void_p, c_void_p, c_void_p, c_long, c_float, c_void_p, ctypes.c_uint, c_void_p],
    "sais_dropout_mask": [c_void_p, c_long, c_float, c_void_p, ctypes.c_uint, c_void_p],
    "sais_head_fwd": [c_void_p, c_void_p, c_long, c_long, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                      c_void_p, c_void_p, c_void_p],
    "sais_head_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_long, c_int, c_int, c_void_p, c_void_p,
                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "sais_mil_forward": [c_void_p, c_long, c_void_p, c_int, c_int, c_void_p, c_void_p],
    "sais_mil_head": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "sais_importance_fwd": [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "sais_importance_bwd": [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "sais_importance_loss": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p],
    "sais_nce": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                 c_float, c_void_p],
    # DINO pre-training objective
    "sais_dino_row_lse": [c_void_p, c_long, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p],
    "sais_dino_loss_partials": [c_int, c_int],
    "sais_dino_loss": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                       c_float, c_void_p, c_long, c_void_p, c_void_p, c_void_p],
    "sais_dino_colsum": [c_void_p, c_long, c_int, c_int, c_void_p, c_void_p],
    "sais_dino_center_ema": [c_void_p, c_void_p, c_int, c_float, c_float, c_void_p],
    "sais_gelu_fwd_f32": [c_void_p, c_void_p, c_long, c_void_p],
    "sais_split_bf16x3": [c_void_p, c_long, c_int, c_int, c_void_p, c_int, c_void_p],
    "sais_gelu_bwd_f32": [c_void_p, c_void_p, c_void_p, c_long, c_void_p],
    "sais_l2norm_fwd": [c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p],
    "sais_l2norm_bwd": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p],
    "sais_weight_norm_fwd": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "sais_weight_norm_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "sais_pos_interp_fwd": [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p],
    "sais_pos_interp_bwd": [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p],
    "sais_opt_chunk_elems": [],
    "sais_grad_norms": [c_void_p, c_void_p, c_int, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p],
    "sais_adamw_ema_step": [ctypes.POINTER(SaisAdamW), c_void_p],
}

_lib = None


class SaisHipError(RuntimeError):
    pass


def load():
    """dlopen libsais_hip.so and bind every entry point.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SaisHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C sais_amd/csrc`).  There is no CPU fallback for the product path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the ABI drifted
        fn.argtypes = argtypes
        fn.restype = c_int
    lib.sais_preprocess_plan_destroy.restype = None
    lib.sais_workspace_bytes.restype = ctypes.c_size_t
    lib.sais_gemm_tn_grouped_slab_bytes.restype = ctypes.c_size_t
    lib.sais_last_error.restype = ctypes.c_char_p
    lib.sais_last_error.argtypes = []
    _lib = lib
    return lib


def call(name, *args):
    rc = getattr(load(), name)(*args)
    if rc != 0:
        why = load().sais_last_error().decode() if rc == -2 else "invalid argument"
        raise SaisHipError(f"{name} failed with code {rc}: {why}")
    return rc
