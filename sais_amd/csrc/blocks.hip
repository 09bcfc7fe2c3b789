// Block-level entry points (include/sais_hip.h, "block-level entry points"): one Block of the DINO ViT
// (dino-main/vision_transformer.py:95-113) per call, forward or backward, as C-side sequencing of the GEMM-level entries of
// this library — the launch plan that sais_amd/vit.py otherwise spells out in Python.  Host code only: no kernels here.
//
// Two dispatch regimes, chosen by M = frames * ntok exactly as the Python host does (ops.ROW_GEMM_MIN_M): from 8192 rows on
// the LayerNorms live in the epilogues of the row-owning GEMMs (sais_gemm_ln_fwd / _bwd), below that the stand-alone
// LayerNorm kernels run behind plain GEMMs.  Both give the reference's Block; only the launch count differs.
#include <string.h>
#include "common.hpp"
#include "../../include/sais_hip.h"

namespace {
constexpr int D = 384, HID = 1536, QKV = 1152;
constexpr int ROW_GEMM_MIN_M = 8192;
constexpr size_t ALIGN = 256;

size_t up(size_t b) { return (b + ALIGN - 1) / ALIGN * ALIGN; }

int gemm(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int epi, const float* bias, void* out, int ldo,
         void* out2, int ldo2, const void* aux, int ldaux, const float* rowscale, void* stream) {
    SaisGemm g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.M = M; g.N = N; g.K = K; g.epilogue = epi; g.bias = bias;
    g.out = out; g.ldo = ldo; g.out2 = out2; g.ldo2 = ldo2; g.aux = aux; g.ldaux = ldaux; g.rowscale = rowscale;
    return sais_gemm_nt(&g, stream);
}

#define TRY(x) do { int rc_ = (x); if (rc_ != SAIS_OK) return rc_; } while (0)
}  // namespace

// slab workspace of the block's grouped weight-gradient launch (sais_gemm_tn_grouped_ws; 0 outside its wide-tile regime)
static size_t block_dw_slab_bytes(int M) {
    const SaisTnItem shape[4] = {{nullptr, D, nullptr, HID, D, HID, nullptr, HID, nullptr},
                                 {nullptr, HID, nullptr, D, HID, D, nullptr, D, nullptr},
                                 {nullptr, D, nullptr, D, D, D, nullptr, D, nullptr},
                                 {nullptr, QKV, nullptr, D, QKV, D, nullptr, D, nullptr}};
    return sais_gemm_tn_grouped_slab_bytes(shape, 4, M);
}

extern "C" size_t sais_workspace_bytes(int op, int frames, int ntok) {
    if (frames <= 0 || ntok <= 0) return 0;
    const size_t M = (size_t)frames * ntok;
    switch (op) {
        case SAIS_OP_VIT_BLOCK_FWD:                    // GELU(u) when the caller does not keep it (inference)
            return up(M * HID * 2);
        case SAIS_OP_VIT_BLOCK_BWD:                    // du, d(mid) bf16, d(attention out), dqkv, dxn (small-M regime), dW slabs
            return up(M * HID * 2) + 3 * up(M * D * 2) + up(M * QKV * 2) + up(block_dw_slab_bytes((int)M));
        case SAIS_OP_TEMPORAL_LAYER_FWD: {             // raw split-K slabs of out_proj, then of linear2 (the larger)
            const int m = frames * ntok;
            const int ns = sais_tgemm_nsplit(m, D, 2048) > sais_tgemm_nsplit(m, D, D) ? sais_tgemm_nsplit(m, D, 2048)
                                                                                       : sais_tgemm_nsplit(m, D, D);
            return up((size_t)ns * M * D * 4);
        }
        case SAIS_OP_TEMPORAL_LAYER_BWD: {             // dy2, dt2, dy-drop1, dh, dqkv + the slabs of dh . W1 / dt1 . Wo
            const int m = frames * ntok;
            const size_t slabs = (size_t)(sais_tgemm_nsplit(m, D, 2048) + sais_tgemm_nsplit(m, D, D)) * M * D * 4;
            return 3 * up(M * D * 4) + up(M * 2048 * 4) + up(M * QKV * 4) + up(slabs);
        }
        default:
            return 0;
    }
}

namespace {
// scratch of one backward call (sais_workspace_bytes(SAIS_OP_VIT_BLOCK_BWD)): the gradient tensors that live inside the block
struct BwdScratch { void* du; void* dxb; void* dao; void* dxn; void* dqkv; void* slabs; };
BwdScratch carve(void* workspace, int M) {
    char* ws = (char*)workspace;
    BwdScratch s;
    s.du = ws;             ws += up((size_t)M * HID * 2);
    s.dxb = ws;            ws += up((size_t)M * D * 2);
    s.dao = ws;            ws += up((size_t)M * D * 2);
    s.dxn = ws;            ws += up((size_t)M * D * 2);
    s.dqkv = ws;           ws += up((size_t)M * QKV * 2);
    s.slabs = ws;
    return s;
}
// the four weight / bias gradients of a block as items of the grouped dW launch
void dw_items(const SaisVitBlockParams* w, const SaisVitBlockBwd* a, const BwdScratch& sc, SaisTnItem* items) {
    items[0] = SaisTnItem{a->dx16_in, D, a->h, HID, D, HID, w->d_fc2_w, HID, w->d_fc2_b};
    items[1] = SaisTnItem{sc.du, HID, a->xn2, D, HID, D, w->d_fc1_w, D, w->d_fc1_b};
    items[2] = SaisTnItem{sc.dxb, D, a->attn_out, D, D, D, w->d_proj_w, D, w->d_proj_b};
    items[3] = SaisTnItem{sc.dqkv, QKV, a->xn1, D, QKV, D, w->d_qkv_w, D, w->d_qkv_b};
}
// M-splits for the kernels that take the number from the caller (the 192 x 384 kernel plans its own)
int dw_nsplit(int M, int nblocks) {
    const int tiles = nblocks * ((D / 128) * (HID / 128) * 2 + (D / 128) * (D / 128) + (QKV / 128) * (D / 128));
    int nsplit = (432 + tiles - 1) / tiles;
    const int cap = (M + 255) / 256;
    if (nsplit > cap) nsplit = cap;
    return nsplit < 1 ? 1 : nsplit;
}
}  // namespace

extern "C" int sais_gelu_grad_bytes(void) {
    static const int v = [] { const char* e = getenv("SAIS_GELU_GRAD_Q8"); return (e && atoi(e) == 0) ? 2 : 1; }();
    return v;
}

extern "C" int sais_vit_block_fwd(const SaisVitBlockParams* w, const SaisVitBlockFwd* a, void* workspace, size_t ws_bytes,
                                  void* stream) {
    SAIS_ENTER();
    if (!w || !a || a->frames <= 0 || (a->ntok != 197 && a->ntok != 37)) return SAIS_ERR_ARG;
    if (!a->xn1 || !a->x_in || !a->qkv || !a->attn_out || !a->x_mid || !a->xn2 || !a->x_out) return SAIS_ERR_ARG;
    if (!w->qkv_w || !w->proj_w || !w->fc1_w || !w->fc2_w || !w->norm2_g || !w->norm2_b) return SAIS_ERR_ARG;
    if (a->gelu_grad && !a->h) return SAIS_ERR_ARG;
    if (w->next_norm_g && (!w->next_norm_b || !a->xn_next)) return SAIS_ERR_ARG;
    const int M = a->frames * a->ntok;
    void* h = a->h;
    if (!h) {
        if (!workspace || ws_bytes < sais_workspace_bytes(SAIS_OP_VIT_BLOCK_FWD, a->frames, a->ntok)) return SAIS_ERR_ARG;
        h = workspace;
    }
    const bool fused = M >= ROW_GEMM_MIN_M;
    // attention branch: qkv -> softmax(q k^T / 8) v -> proj, + residual (DropPath row scale), then norm2
    TRY(gemm(a->xn1, D, w->qkv_w, D, M, QKV, D, SAIS_EPI_BIAS_BF16, w->qkv_b, a->qkv, QKV, nullptr, 0, nullptr, 0, nullptr, stream));
    TRY(sais_vit_attn_fwd(a->qkv, QKV, a->frames, a->ntok, a->attn_out, D, a->lse, nullptr, stream));
    if (fused) {
        SaisGemmLn g;
        memset(&g, 0, sizeof(g));
        g.A = a->attn_out; g.lda = D; g.W = w->proj_w; g.ldw = D; g.M = M; g.K = D; g.bias = w->proj_b;
        g.resid = a->x_in; g.ldr = D; g.out32 = a->x_mid; g.ldo32 = D; g.out16 = a->xn2; g.ldo16 = D;
        g.gamma = w->norm2_g; g.beta = w->norm2_b; g.eps = 1e-6f; g.mean = a->mean2; g.rstd = a->rstd2;
        g.rowscale = a->rowscale_attn;
        TRY(sais_gemm_ln_fwd(&g, stream));
    } else {
        TRY(gemm(a->attn_out, D, w->proj_w, D, M, D, D, SAIS_EPI_BIAS_RESID_F32, w->proj_b, a->x_mid, D, nullptr, 0, a->x_in, D,
                 a->rowscale_attn, stream));
        TRY(sais_layernorm_fwd(a->x_mid, D, M, D, w->norm2_g, w->norm2_b, 1e-6f, a->xn2, D, nullptr, 0, a->mean2, a->rstd2, stream));
    }
    // MLP branch: fc1 + GELU (+ GELU' for the backward) -> fc2 + residual (+ the next block's norm1)
    const bool gq8 = sais_gelu_grad_bytes() == 1;
    TRY(gemm(a->xn2, D, w->fc1_w, D, M, HID, D,
             !a->gelu_grad ? SAIS_EPI_BIAS_GELU_BF16 : gq8 ? SAIS_EPI_BIAS_GELU_GRADQ_BF16 : SAIS_EPI_BIAS_GELU_GRAD_BF16, w->fc1_b,
             h, HID, a->gelu_grad, HID, nullptr, 0, nullptr, stream));
    if (fused && w->next_norm_g) {
        SaisGemmLn g;
        memset(&g, 0, sizeof(g));
        g.A = h; g.lda = HID; g.W = w->fc2_w; g.ldw = HID; g.M = M; g.K = HID; g.bias = w->fc2_b;
        g.resid = a->x_mid; g.ldr = D; g.out32 = a->x_out; g.ldo32 = D; g.out16 = a->xn_next; g.ldo16 = D;
        g.gamma = w->next_norm_g; g.beta = w->next_norm_b; g.eps = 1e-6f; g.mean = a->mean_next; g.rstd = a->rstd_next;
        g.rowscale = a->rowscale_mlp;
        TRY(sais_gemm_ln_fwd(&g, stream));
    } else {
        TRY(gemm(h, HID, w->fc2_w, HID, M, D, HID, SAIS_EPI_BIAS_RESID_F32, w->fc2_b, a->x_out, D, nullptr, 0, a->x_mid, D,
                 a->rowscale_mlp, stream));
        if (w->next_norm_g)
            TRY(sais_layernorm_fwd(a->x_out, D, M, D, w->next_norm_g, w->next_norm_b, 1e-6f, a->xn_next, D, nullptr, 0,
                                   a->mean_next, a->rstd_next, stream));
    }
    return SAIS_OK;
}

extern "C" int sais_vit_block_bwd(const SaisVitBlockParams* w, const SaisVitBlockBwd* a, void* workspace, size_t ws_bytes,
                                  void* stream) {
    SAIS_ENTER();
    if (!w || !a || a->frames <= 0 || (a->ntok != 197 && a->ntok != 37)) return SAIS_ERR_ARG;
    if (!a->x_in || !a->mean1 || !a->rstd1 || !a->xn1 || !a->qkv || !a->attn_out || !a->lse || !a->x_mid || !a->mean2 ||
        !a->rstd2 || !a->xn2 || !a->h || !a->gelu_grad || !a->dx || !a->dx16_in || !a->dx16_out)
        return SAIS_ERR_ARG;
    if (!w->qkv_wt || !w->proj_wt || !w->fc1_wt || !w->fc2_wt || !w->norm1_g || !w->norm2_g || !w->d_qkv_w || !w->d_proj_w ||
        !w->d_fc1_w || !w->d_fc2_w || !w->d_norm1_g || !w->d_norm1_b || !w->d_norm2_g || !w->d_norm2_b)
        return SAIS_ERR_ARG;
    if (!workspace || ws_bytes < sais_workspace_bytes(SAIS_OP_VIT_BLOCK_BWD, a->frames, a->ntok) || ((uintptr_t)workspace & 15))
        return SAIS_ERR_ARG;
    const int M = a->frames * a->ntok;
    const BwdScratch sc = carve(workspace, M);
    void* const du = sc.du; void* const dxb = sc.dxb; void* const dao = sc.dao; void* const dxn = sc.dxn; void* const dqkv = sc.dqkv;
    void* const slabs = sc.slabs;
    const size_t slab_bytes = block_dw_slab_bytes(M);
    if (a->defer_dw && a->dx16_in == a->dx16_out) return SAIS_ERR_ARG;
    const bool fused = M >= ROW_GEMM_MIN_M;
    // MLP branch: du = (d . W2) * GELU'(u);  d(norm2 out) = du . W1;  norm2's backward adds the residual gradient
    TRY(gemm(a->dx16_in, D, w->fc2_wt, D, M, HID, D, sais_gelu_grad_bytes() == 1 ? SAIS_EPI_MULQ_BF16 : SAIS_EPI_MUL_BF16, nullptr, du, HID,
             nullptr, 0, a->gelu_grad, HID, nullptr, stream));
    if (fused) {
        SaisGemmLn g;
        memset(&g, 0, sizeof(g));
        g.A = du; g.lda = HID; g.W = w->fc1_wt; g.ldw = HID; g.M = M; g.K = HID;
        g.resid = a->x_mid; g.ldr = D; g.out32 = a->dx; g.ldo32 = D; g.out16 = dxb; g.ldo16 = D;
        g.gamma = w->norm2_g; g.mean = (float*)a->mean2; g.rstd = (float*)a->rstd2; g.dres = a->dx; g.lddres = D;
        g.dgamma = w->d_norm2_g; g.dbeta = w->d_norm2_b; g.rowscale16 = a->rowscale_attn;
        g.xn16 = a->xn2; g.ldxn16 = D; g.beta = w->norm2_b;       // xhat from the saved bf16 norm2 output (ABI 11)
        TRY(sais_gemm_ln_bwd(&g, stream));
    } else {
        TRY(gemm(du, HID, w->fc1_wt, HID, M, D, HID, SAIS_EPI_BIAS_BF16, nullptr, dxn, D, nullptr, 0, nullptr, 0, nullptr, stream));
        TRY(sais_layernorm_bwd(dxn, D, nullptr, 0, a->x_mid, D, a->mean2, a->rstd2, w->norm2_g, a->dx, D, M, D, a->dx, D, dxb, D,
                               w->d_norm2_g, w->d_norm2_b, a->rowscale_attn, nullptr, 0.f, nullptr, 0, stream));
    }
    // attention branch
    TRY(gemm(dxb, D, w->proj_wt, D, M, D, D, SAIS_EPI_BIAS_BF16, nullptr, dao, D, nullptr, 0, nullptr, 0, nullptr, stream));
    TRY(sais_vit_attn_bwd(a->qkv, QKV, dao, D, a->attn_out, D, a->lse, nullptr, a->frames, a->ntok, dqkv, QKV, stream));
    // the four weight / bias gradients of the block in one launch
    if (!a->defer_dw) {
        SaisTnItem items[4];
        dw_items(w, a, sc, items);
        TRY(sais_gemm_tn_grouped_ws(items, 4, M, dw_nsplit(M, 1), slab_bytes ? slabs : nullptr, slab_bytes, stream));
    }
    // dX of qkv + norm1's backward: the gradient of the block input
    if (fused) {
        SaisGemmLn g;
        memset(&g, 0, sizeof(g));
        g.A = dqkv; g.lda = QKV; g.W = w->qkv_wt; g.ldw = QKV; g.M = M; g.K = QKV;
        g.resid = a->x_in; g.ldr = D; g.out32 = a->dx; g.ldo32 = D; g.out16 = a->dx16_out; g.ldo16 = D;
        g.gamma = w->norm1_g; g.mean = (float*)a->mean1; g.rstd = (float*)a->rstd1; g.dres = a->dx; g.lddres = D;
        g.dgamma = w->d_norm1_g; g.dbeta = w->d_norm1_b; g.rowscale16 = a->rowscale_prev;
        if (w->norm1_b) { g.xn16 = a->xn1; g.ldxn16 = D; g.beta = w->norm1_b; }
        TRY(sais_gemm_ln_bwd(&g, stream));
    } else {
        TRY(gemm(dqkv, QKV, w->qkv_wt, QKV, M, D, QKV, SAIS_EPI_BIAS_BF16, nullptr, dxn, D, nullptr, 0, nullptr, 0, nullptr, stream));
        TRY(sais_layernorm_bwd(dxn, D, nullptr, 0, a->x_in, D, a->mean1, a->rstd1, w->norm1_g, a->dx, D, M, D, a->dx, D,
                               a->dx16_out, D, w->d_norm1_g, w->d_norm1_b, a->rowscale_prev, nullptr, 0.f, nullptr, 0, stream));
    }
    return SAIS_OK;
}

extern "C" int sais_vit_blocks_dw(const SaisVitBlockParams* const* w, const SaisVitBlockBwd* const* a, void* const* workspaces,
                                  size_t ws_bytes, int nblocks, const SaisTnItem* extra, int nextra, void* stream) {
    SAIS_ENTER();
    if (!w || !a || !workspaces || nblocks <= 0 || nextra < 0 || (nextra && !extra) || 4 * nblocks + nextra > SAIS_TN_MAX_ITEMS)
        return SAIS_ERR_ARG;
    SaisTnItem items[SAIS_TN_MAX_ITEMS];
    const int M = a[0] ? a[0]->frames * a[0]->ntok : 0;
    for (int i = 0; i < nblocks; ++i) {
        if (!w[i] || !a[i] || !workspaces[i] || a[i]->frames * a[i]->ntok != M || M <= 0) return SAIS_ERR_ARG;
        if (ws_bytes < sais_workspace_bytes(SAIS_OP_VIT_BLOCK_BWD, a[i]->frames, a[i]->ntok)) return SAIS_ERR_ARG;
        dw_items(w[i], a[i], carve(workspaces[i], M), items + 4 * i);
    }
    // partial tiles of the M-splits go to the slab region of the first workspace when it is large enough (it is sized for one block at
    // ten splits = 240 partial tiles; two blocks at five, five at two are as many); otherwise fp32 atomics
    for (int i = 0; i < nextra; ++i) items[4 * nblocks + i] = extra[i];
    const int n = 4 * nblocks + nextra;
    const size_t need = sais_gemm_tn_grouped_slab_bytes(items, n, M);
    const size_t have = block_dw_slab_bytes(M);
    void* slabs = need && need <= have ? carve(workspaces[0], M).slabs : nullptr;
    return sais_gemm_tn_grouped_ws(items, n, M, dw_nsplit(M, nblocks), slabs, slabs ? have : 0, stream);
}

// ---------------------------------------------------------------------------------------------- temporal encoder layer
namespace {
constexpr int FF = 2048;

int tg(const float* A, const float* W, int M, int N, int K, int epi, int nsplit, const float* bias, const float* aux, float* out,
       float p, const unsigned long long* rng, unsigned site, void* stream) {
    SaisTGemm g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.lda = K; g.W = W; g.ldw = K; g.M = M; g.N = N; g.K = K; g.epilogue = epi; g.nsplit = nsplit; g.bias = bias;
    g.aux = aux; g.ldaux = N; g.out = out; g.ldo = N; g.p_drop = p; g.rng_state = rng; g.site = site;
    return sais_tgemm(&g, stream);
}
}  // namespace

extern "C" int sais_temporal_layer_fwd(const SaisTemporalLayerParams* w, const SaisTemporalLayerFwd* a, void* workspace,
                                       size_t ws_bytes, void* stream) {
    SAIS_ENTER();
    if (!w || !a || a->B <= 0 || a->S <= 0 || !a->z || !a->key_pad || !a->qkv || !a->ctx || !a->z1 || !a->h || !a->z_out)
        return SAIS_ERR_ARG;
    if (!w->in_proj_w || !w->out_proj_w || !w->linear1_w || !w->linear2_w || !w->norm1_g || !w->norm1_b || !w->norm2_g || !w->norm2_b)
        return SAIS_ERR_ARG;
    if (a->p_drop < 0.f || a->p_drop >= 1.f || (a->p_drop > 0.f && !a->rng_state)) return SAIS_ERR_ARG;
    if (!workspace || ws_bytes < sais_workspace_bytes(SAIS_OP_TEMPORAL_LAYER_FWD, a->B, a->S) || ((uintptr_t)workspace & 15))
        return SAIS_ERR_ARG;
    const int M = a->B * a->S;
    float* slabs = (float*)workspace;
    const float p = a->p_drop;
    const unsigned long long* rng = p > 0.f ? a->rng_state : nullptr;
    // self-attention: in_proj -> per (sequence, head) softmax(q k^T / sqrt(96)) v with key-padding mask -> out_proj
    TRY(tg(a->z, w->in_proj_w, M, QKV, D, SAIS_TG_BIAS, 1, w->in_proj_b, nullptr, a->qkv, 0.f, nullptr, 0, stream));
    TRY(sais_temporal_attn_fwd(a->qkv, a->key_pad, a->B, a->S, a->ctx, a->attn_avg, p, rng, a->site0, stream));
    int ns = sais_tgemm_nsplit(M, D, D);
    TRY(tg(a->ctx, w->out_proj_w, M, D, D, SAIS_TG_RAW, ns, nullptr, nullptr, slabs, 0.f, nullptr, 0, stream));
    TRY(sais_temporal_ln_fwd(slabs, ns, (long)M * D, w->out_proj_b, a->z, M, p, rng, a->site0 + 1, a->y1, w->norm1_g, w->norm1_b,
                             1e-5f, a->z1, a->mean1, a->rstd1, stream));
    // feed-forward
    TRY(tg(a->z1, w->linear1_w, M, FF, D, SAIS_TG_BIAS_RELU, 1, w->linear1_b, nullptr, a->h, p, rng, a->site0 + 2, stream));
    ns = sais_tgemm_nsplit(M, D, FF);
    TRY(tg(a->h, w->linear2_w, M, D, FF, SAIS_TG_RAW, ns, nullptr, nullptr, slabs, 0.f, nullptr, 0, stream));
    TRY(sais_temporal_ln_fwd(slabs, ns, (long)M * D, w->linear2_b, a->z1, M, p, rng, a->site0 + 3, a->y2, w->norm2_g, w->norm2_b,
                             1e-5f, a->z_out, a->mean2, a->rstd2, stream));
    return SAIS_OK;
}

extern "C" int sais_temporal_layer_bwd(const SaisTemporalLayerParams* w, const SaisTemporalLayerBwd* a, void* workspace,
                                       size_t ws_bytes, void* stream) {
    SAIS_ENTER();
    if (!w || !a || a->B <= 0 || a->S <= 0) return SAIS_ERR_ARG;
    if (!a->z || !a->qkv || !a->ctx || !a->y1 || !a->mean1 || !a->rstd1 || !a->z1 || !a->h || !a->y2 || !a->mean2 || !a->rstd2 ||
        !a->key_pad || (!a->dz_slabs && !a->dz_add) || !a->dx_slabs || !a->dx_add)
        return SAIS_ERR_ARG;
    if (!w->in_proj_wt || !w->out_proj_wt || !w->linear1_wt || !w->linear2_wt || !w->norm1_g || !w->norm2_g || !w->d_in_proj_w ||
        !w->d_out_proj_w || !w->d_linear1_w || !w->d_linear2_w || !w->d_norm1_g || !w->d_norm1_b || !w->d_norm2_g || !w->d_norm2_b)
        return SAIS_ERR_ARG;
    if (a->p_drop < 0.f || a->p_drop >= 1.f || (a->p_drop > 0.f && !a->rng_state)) return SAIS_ERR_ARG;
    if (!workspace || ws_bytes < sais_workspace_bytes(SAIS_OP_TEMPORAL_LAYER_BWD, a->B, a->S) || ((uintptr_t)workspace & 15))
        return SAIS_ERR_ARG;
    const int M = a->B * a->S;
    const float p = a->p_drop;
    const unsigned long long* rng = p > 0.f ? a->rng_state : nullptr;
    char* ws = (char*)workspace;
    float* dy2 = (float*)ws;   ws += up((size_t)M * D * 4);
    float* dt2 = (float*)ws;   ws += up((size_t)M * D * 4);          // dropout2's backward of dy2 (p = 0: dy2 itself is used)
    float* dt1 = (float*)ws;   ws += up((size_t)M * D * 4);
    float* dh = (float*)ws;    ws += up((size_t)M * FF * 4);
    float* dqkv = (float*)ws;  ws += up((size_t)M * QKV * 4);
    float* slab1 = (float*)ws;                                        // dh . W1   [ns1][M][384]
    const int ns1 = sais_tgemm_nsplit(M, D, FF), nso = sais_tgemm_nsplit(M, D, D), nsq = sais_tgemm_nsplit(M, D, QKV);
    float* slabo = slab1 + (size_t)ns1 * M * D;                       // dt1 . Wo  [nso][M][384]
    // norm2 backward (the gradient of the layer output arrives as slabs + add), dropout2 backward as a second output
    float* g2 = p > 0.f ? dt2 : dy2;
    TRY(sais_temporal_ln_bwd(a->dz_slabs, a->dz_slabs ? a->nslab : 0, a->slab_stride, a->dz_add, a->y2, a->mean2, a->rstd2,
                             w->norm2_g, M, dy2, p > 0.f ? dt2 : nullptr, p, rng, a->site0 + 3, w->d_norm2_g, w->d_norm2_b, stream));
    // FFN: dh = drop'(relu'(.)) (g2 . W2);  d(norm1 out) = dh . W1 + dy2 (residual): left as slabs + add for norm1's backward
    TRY(tg(g2, w->linear2_wt, M, FF, D, SAIS_TG_DRELU, 1, nullptr, a->h, dh, p, rng, a->site0 + 2, stream));
    TRY(tg(dh, w->linear1_wt, M, D, FF, SAIS_TG_RAW, ns1, nullptr, nullptr, slab1, 0.f, nullptr, 0, stream));
    float* g1 = p > 0.f ? dt1 : a->dx_add;
    TRY(sais_temporal_ln_bwd(slab1, ns1, (long)M * D, dy2, a->y1, a->mean1, a->rstd1, w->norm1_g, M, a->dx_add,
                             p > 0.f ? dt1 : nullptr, p, rng, a->site0 + 1, w->d_norm1_g, w->d_norm1_b, stream));
    // attention: d ctx = g1 . Wo (raw slabs, summed on load by the attention backward)
    TRY(tg(g1, w->out_proj_wt, M, D, D, SAIS_TG_RAW, nso, nullptr, nullptr, slabo, 0.f, nullptr, 0, stream));
    TRY(sais_temporal_attn_bwd(a->qkv, a->key_pad, a->B, a->S, slabo, nso, (long)M * D, dqkv, p, rng, a->site0, stream));
    // the four weight / bias gradients of the layer in one launch (one M-split: owner-computes, no atomics)
    SaisTnItem items[4] = {
        {g2, D, a->h, FF, D, FF, w->d_linear2_w, FF, w->d_linear2_b},
        {dh, FF, a->z1, D, FF, D, w->d_linear1_w, D, w->d_linear1_b},
        {g1, D, a->ctx, D, D, D, w->d_out_proj_w, D, w->d_out_proj_b},
        {dqkv, QKV, a->z, D, QKV, D, w->d_in_proj_w, D, w->d_in_proj_b}};
    if (a->dw_items_out) {                     // deferred: the caller batches the layers' weight gradients into one launch
        for (int i = 0; i < 4; ++i) a->dw_items_out[i] = items[i];
    } else {
        TRY(sais_gemm_tn_grouped_f32(items, 4, M, 1, stream));
    }
    // gradient wrt the layer input = dx_add (residual path, written by norm1's backward) + dqkv . Win (raw slabs)
    TRY(tg(dqkv, w->in_proj_wt, M, D, QKV, SAIS_TG_RAW, nsq, nullptr, nullptr, a->dx_slabs, 0.f, nullptr, 0, stream));
    return SAIS_OK;
}
