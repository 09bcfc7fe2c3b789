"""CPU-only: the C-ABI library builds, loads, and exports every symbol include/sais_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    from sais_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        ge.build()
    hdr = open(os.path.join(ROOT, "include", "sais_hip.h")).read()
    declared = set(re.findall(r"^(?:int|void|size_t)\s+(sais_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib.sais_abi_version.restype = ctypes.c_int
    assert lib.sais_abi_version() == 13
    assert lib.sais_gelu_grad_bytes() == 1                       # GELU' of the block API as one-byte codes by default


def test_bad_arguments_are_rejected_without_a_gpu():
    from sais_amd import _lib
    lib = _lib.load()
    # NULL pointers / bad shapes must return SAIS_ERR_ARG (-1) before any launch
    assert lib.sais_gemm_nt(None, None) == -1
    g = _lib.SaisGemm()
    assert lib.sais_gemm_nt(ctypes.byref(g), None) == -1
    assert lib.sais_gemm_ln_fwd(None, None) == -1 and lib.sais_gemm_ln_bwd(ctypes.byref(_lib.SaisGemmLn()), None) == -1
    assert lib.sais_tgemm(None, None) == -1 and lib.sais_tgemm(ctypes.byref(_lib.SaisTGemm()), None) == -1
    assert lib.sais_temporal_ln_fwd(None, 1, 0, None, None, 8, 0.0, None, 0, None, None, None, 1e-5, None, None, None, None) == -1
    assert lib.sais_mil_head(None, 1, 1, 2, None, None, None, None, None, None, None, None, None, None, None, None) == -1
    # the split-K sizing rule of the temporal GEMMs is a host function of the library (no GPU): benchmark shapes
    assert [lib.sais_tgemm_nsplit(264, 384, k) for k in (384, 1152, 2048)] == [3, 6, 8]
    assert lib.sais_tgemm_nsplit(264, 2048, 384) == 1 and lib.sais_tgemm_nsplit(0, 384, 384) == 1
    assert lib.sais_layernorm_fwd(None, 384, 4, 384, None, None, 1e-6, None, 384, None, 384, None, None, None) == -1
    assert lib.sais_vit_attn_fwd(None, 1152, 1, 197, None, 384, None, None, None) == -1
    assert lib.sais_temporal_attn_fwd(None, None, 1, 1000, None, None, 0.0, None, 0, None) == -1
    assert lib.sais_dropout_f32(None, None, None, 10, 0.1, None, 0, None) == -1 and lib.sais_rng_advance(None, None) == -1
    assert lib.sais_preprocess_plan_create(0, 10, 0.8, 0.8, None, None, None) == -1
    assert lib.sais_preprocess_run(None, None, 1, None, None) == -1
    # DINO objective entries
    assert lib.sais_vit_attn_fwd(ctypes.c_void_p(16), 1152, 1, 50, ctypes.c_void_p(16), 384, None, None, None) == -1   # ntok
    assert lib.sais_patchify(ctypes.c_void_p(16), 1, 100, ctypes.c_void_p(16), None) == -1                            # side % 16
    assert lib.sais_dino_row_lse(None, 8, 1, 8, 1.0, None, None, None) == -1
    assert lib.sais_dino_loss(None, 8, None, 8, None, None, None, 1, 2, 8, 0.1, 0.04, None, 8, None, None, None) == -1
    assert lib.sais_dino_loss_partials(64, 65536) == 64 * 64 and lib.sais_dino_loss_partials(2, 1000) == 2
    assert lib.sais_opt_chunk_elems() == 8192
    assert lib.sais_adamw_ema_step(None, None) == -1 and lib.sais_adamw_ema_step(ctypes.byref(_lib.SaisAdamW()), None) == -1
    assert lib.sais_grad_norms(None, None, 0, None, 0, 1.0, None, None, None) == -1
    assert lib.sais_weight_norm_fwd(None, None, 1, 256, None, None, None) == -1
    assert lib.sais_split_bf16x3(None, 256, 1, 256, None, 0, None) == -1
    assert lib.sais_pos_interp_fwd(None, 36, 196, None, 384, None, None) == -1
    # round 4 (ABI 8): block-level entries, CLS-only attention, split-K finish, RAFT correlation volume
    assert lib.sais_mlp_fwd(None, None) == -1 and lib.sais_mlp_bwd(ctypes.byref(_lib.SaisMlp()), None) == -1
    assert lib.sais_vit_block_fwd(None, None, None, 0, None) == -1
    assert lib.sais_vit_block_bwd(ctypes.byref(_lib.SaisVitBlockParams()), ctypes.byref(_lib.SaisVitBlockBwd()), None, 0, None) == -1
    M = 256 * 197
    assert lib.sais_workspace_bytes(_lib.OP_VIT_BLOCK_FWD, 256, 197) >= M * 1536 * 2
    assert lib.sais_workspace_bytes(_lib.OP_VIT_BLOCK_BWD, 256, 197) >= M * (1536 + 3 * 384 + 1152) * 2
    assert lib.sais_workspace_bytes(99, 256, 197) == 0
    # round 5 (ABI 10): slab workspace of the grouped weight-gradient launch (argument checks only: no GPU here)
    items = (_lib.SaisTnItem * 4)()
    for it, (n1, n2) in zip(items, ((384, 1536), (1536, 384), (384, 384), (1152, 384))):
        it.N1, it.N2 = n1, n2
    lib.sais_gemm_tn_grouped_slab_bytes.restype = ctypes.c_size_t
    # round 6: the ViT block's launch runs on 192 x 384 tiles: 24 tiles x 10 M-splits, raw partial tile + bias tile per wave
    assert lib.sais_gemm_tn_grouped_slab_bytes(items, 4, M) == 24 * 10 * (192 * 384 * 4 + 4 * 32 * 32 * 4)
    assert lib.sais_gemm_tn_grouped_slab_bytes(items, 4, 300) == 0 and lib.sais_gemm_tn_grouped_slab_bytes(None, 4, M) == 0
    assert lib.sais_workspace_bytes(_lib.OP_VIT_BLOCK_BWD, 256, 197) >= M * (1536 + 3 * 384 + 1152) * 2 + 24 * 10 * 192 * 384 * 4
    assert lib.sais_gemm_tn_grouped_ws(None, 4, M, 7, None, 0, None) == -1
    assert lib.sais_im2col_f32(None, 3, 8, 8, 3, 3, 1, 1, 1, 1, None, 64, 128, None) == -1
    assert lib.sais_workspace_bytes(_lib.OP_TEMPORAL_LAYER_FWD, 8, 33) >= 8 * 264 * 384 * 4
    assert lib.sais_workspace_bytes(_lib.OP_TEMPORAL_LAYER_BWD, 8, 33) >= 264 * (3 * 384 + 2048 + 1152) * 4
    assert lib.sais_temporal_layer_fwd(None, None, None, 0, None) == -1
    assert lib.sais_temporal_layer_bwd(ctypes.byref(_lib.SaisTemporalLayerParams()), ctypes.byref(_lib.SaisTemporalLayerBwd()),
                                       None, 0, None) == -1
    assert lib.sais_vit_attn_cls_fwd(None, 1152, 1, 197, None, 384, None) == -1
    assert lib.sais_vit_attn_cls_bwd(ctypes.c_void_p(16), 1152, ctypes.c_void_p(16), 384, 1, 500, ctypes.c_void_p(16), 1152, None) == -1
    assert lib.sais_splitk_finish(None, 2, 4, 384, 384, None, None, None, 0, None, 0, None, 0, None) == -1
    assert lib.sais_touch(None, 64, None) == -1
    assert lib.sais_raft_corr_pool(None, 64, 1, 8, 8, None, None, None, None) == -1
    assert lib.sais_raft_lookup(None, 64, None, None, None, None, 1, 8, 8, 4, None, None) == -1
