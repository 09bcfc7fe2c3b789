/* libsais_hip.so — C ABI of the MI355X (gfx950) kernels behind the SAIS ViT-over-video hot path.
 *
 * The reference (danikiyasseh/SAIS) has no FFI / plugin interface: every op on this path is a stock
 * torch.nn module executed by ATen on the CPU.  The entry points below are therefore the operator
 * set a maintainer would bind in place of those modules; each one cites the reference code it
 * replaces (paths relative to SAIS/scripts/).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions (all entry points):
 *   - raw DEVICE pointers owned by the caller; no allocation, no host sync, no global state;
 *   - asynchronous on the passed hipStream_t (`stream`, passed as void*); re-entrant across streams;
 *   - returns 0 (SAIS_OK) or a negative error code, never throws;
 *   - "bf16" = bfloat16 storage, "f32" = IEEE float; all accumulation / statistics in f32;
 *   - row-major tensors, `ld*` = leading dimension in ELEMENTS.
 */
#ifndef SAIS_HIP_H
#define SAIS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAIS_ABI_VERSION 13
int sais_abi_version(void);
/* text of the HIP error behind the calling thread's last SAIS_ERR_LAUNCH (-2) return */
const char* sais_last_error(void);

/* ---------------------------------------------------------------- GEMM  C = A . B^T (+epilogue)
 * Replaces nn.Linear.forward everywhere on the path:
 *   dino-main/vision_transformer.py:59-65 (Mlp fc1/fc2), :80-92 (Attention qkv/proj), :128-131
 *   (PatchEmbed conv == GEMM on patches); prepare_model.py:74-81 (TransformerEncoderLayer in_proj /
 *   out_proj / linear1 / linear2).  With the transposed weight it is also dX = dY . W.          */
enum {
    SAIS_EPI_BIAS_BF16 = 0,       /* out bf16 = acc + bias                                           */
    SAIS_EPI_BIAS_RELU_BF16 = 1,  /* out bf16 = relu(acc + bias)                                     */
    SAIS_EPI_BIAS_F32 = 2,        /* out f32  = acc + bias                                           */
    SAIS_EPI_BIAS_RESID_F32 = 3,  /* out f32  = aux(f32) + acc + bias ; out2 (optional) same in bf16 */
    SAIS_EPI_BIAS_GELU_BF16 = 4,  /* out bf16 = gelu_erf(acc + bias) ; out2 (optional) = acc + bias  */
    SAIS_EPI_DGELU_BF16 = 5,      /* out bf16 = acc * gelu'(aux bf16)                                */
    SAIS_EPI_DRELU_BF16 = 6,      /* out bf16 = acc * (aux bf16 > 0)                                 */
    SAIS_EPI_BIAS_RELU_F32 = 8,   /* (f32 GEMM only) out f32 = relu(acc + bias)                      */
    SAIS_EPI_DRELU_F32 = 9,       /* (f32 GEMM only) out f32 = acc * (aux f32 > 0)                   */
    SAIS_EPI_BIAS_GELU_GRAD_BF16 = 10, /* out bf16 = gelu_erf(acc + bias) ; out2 bf16 = gelu'(acc + bias): the training
                                     forward stores the derivative instead of the pre-activation, so that ...     */
    SAIS_EPI_MUL_BF16 = 11,       /* ... the backward is out bf16 = acc * aux(bf16), with no erf/exp in its epilogue */
    SAIS_EPI_PATCH_F32 = 7,       /* patch-embed: row f*grp_in+q -> token row f*grp_out+q+grp_off,
                                     out f32 = acc + bias + aux(f32 pos)[q+grp_off]                  */
    SAIS_EPI_BIAS_GELU_GRADQ_BF16 = 13, /* (ABI 12) as 10, but out2 holds gelu'(acc + bias) as ONE BYTE per element: code q =
                                     clamp(rint(26 + 203 d), 0, 255), i.e. d = (q - 26) / 203 — 0 and 1 are exact, the range
                                     [-0.128, 1.128] covers gelu' (min -0.129, max 1.129), step 0.0049 (rms error 0.0014; a bf16
                                     near 0.5 has 0.0011): half the bytes of the tensor the backward reads again; ldo2 in bytes */
    SAIS_EPI_MULQ_BF16 = 14,      /* (ABI 12) out bf16 = acc * (aux(u8) - 26) / 203: the backward of 13; ldaux in bytes        */
    SAIS_EPI_RAW_SLABS_F32 = 12   /* (ABI 8, M < 8192 only) split-K: K is cut into grp_in slices, slice z writes its raw partial
                                     sums to out f32 [grp_in][M][ldo]; sais_splitk_finish sums them in a fixed order      */
};

typedef struct SaisGemm {
    const void* A; int lda;        /* bf16 [M,K]                         */
    const void* B; int ldb;        /* bf16 [N,K]  (nn.Linear weight)     */
    int M, N, K;                   /* N % 128 == 0, K % 64 == 0          */
    int epilogue;                  /* SAIS_EPI_*                         */
    const float* bias;             /* f32 [N] or NULL                    */
    void* out; int ldo;
    void* out2; int ldo2;
    const void* aux; int ldaux;
    int grp_in, grp_out, grp_off;  /* SAIS_EPI_PATCH_F32 only            */
    const float* rowscale;         /* SAIS_EPI_BIAS_RESID_F32 only, optional: DropPath (vision_transformer.py:27-46,111,113)
                                      out = aux + rowscale[m] * (acc + bias), rowscale[m] = keep / (1 - p) of row m's sample */
    /* sais_gemm_nt_f32 only: train-mode dropout of the temporal encoder layer fused into the epilogue (p_drop = 0: none);
     * mask element = m * N + n of site `site` (see sais_dropout_f32):  _BIAS_RESID_F32: out = aux + drop(acc + bias)
     * (dropout1 / dropout2);  _BIAS_RELU_F32: out = drop(relu(acc + bias)) (the FFN dropout);  _DRELU_F32: out =
     * drop(acc) where aux > 0 (its backward, same site).                                                          */
    float p_drop; const unsigned long long* rng_state; unsigned site;
} SaisGemm;

int sais_gemm_nt(const SaisGemm* g, void* stream);
/* Second half of a SAIS_EPI_RAW_SLABS_F32 product (the [frames, 384] GEMMs of the CLS-only last ViT block: a K = 1536 loop
 * over 6 output tiles is 24 serial steps; cut 12 ways it is 72 workgroups of 2):
 *   y = sum_z slabs[z][m][n] + bias[n];  y *= rowscale[m] (optional);  y += aux[m][n] (optional, f32);
 *   out32 (optional) = y, out16 (optional) = bf16(y).  Deterministic (fixed summation order, no atomics).            */
int sais_splitk_finish(const float* slabs, int nslabs, int M, int N, int lds, const float* bias, const float* rowscale,
                       const float* aux, int ldaux, float* out32, int ldo32, void* out16, int ldo16, void* stream);

/* Same contract with FP32 operands (A f32 [M,K], B f32 [N,K], out f32) at ~fp32 accuracy: each operand is
 * split hi/lo into two bf16 and three MFMA products are accumulated ("bf16x3").  Epilogues:
 * SAIS_EPI_BIAS_F32, _BIAS_RESID_F32 (aux f32), _BIAS_RELU_F32, _DRELU_F32 (aux f32).  This is what the
 * temporal TransformerEncoder's nn.Linear layers run on (prepare_model.py:74-81, called at :213): its
 * activations feed the <=1e-3 logit parity bar directly and are tiny.
 * Optional split-K for these few-row problems: out2 = caller workspace f32 [ldo2][M][N], ldo2 = number of K
 * splits (must divide K/64); partial sums land there and a second tiny kernel reduces + applies the epilogue. */
int sais_gemm_nt_f32(const SaisGemm* g, void* stream);

/* ---------------------------------------------------------------- the temporal encoder's linear layers (round 3)
 * The same "bf16x3" arithmetic as sais_gemm_nt_f32, tiled for M = clips x (T + 1) = a few hundred rows: 64 x 64 output
 * tiles, two K-steps of loads in flight, optional split-K into RAW partial slabs out[z][M][N] (z < nsplit, nsplit divides
 * K / 64) that the row kernels below consume — there is no separate reduce launch.  N % 64 == 0, K % 64 == 0.
 *   SAIS_TG_RAW        out[z] = partial A . W^T                      (out_proj / linear2 forward; every N = 384 dX)
 *   SAIS_TG_BIAS       out = A . W^T + bias (bias may be NULL)       (in_proj, prepare_model.py:213 -> MultiheadAttention)
 *   SAIS_TG_BIAS_RELU  out = drop(relu(A . W^T + bias))              (linear1 + activation + dropout)
 *   SAIS_TG_DRELU      out = drop(A . W^T) where aux > 0, else 0     (its backward; aux = the saved, dropped relu output)
 * drop(): train-mode dropout, mask element m * N + n of `site`, p_drop = 0 for none.                                  */
#define SAIS_TG_RAW 0
#define SAIS_TG_BIAS 1
#define SAIS_TG_BIAS_RELU 2
#define SAIS_TG_DRELU 3
typedef struct SaisTGemm {
    const float* A; long lda;      /* f32 [M,K]                          */
    const float* W; long ldw;      /* f32 [N,K]  (nn.Linear weight, or its transpose for dX) */
    int M, N, K;
    int epilogue;                  /* SAIS_TG_*                          */
    int nsplit;                    /* SAIS_TG_RAW: number of K splits = slabs written; otherwise 1 */
    const float* bias;             /* f32 [N] or NULL                    */
    const float* aux; long ldaux;  /* SAIS_TG_DRELU                      */
    float* out; long ldo;          /* f32 [M,N] (RAW: [nsplit][M][N], ldo = N) */
    float p_drop; const unsigned long long* rng_state; unsigned site;
} SaisTGemm;
int sais_tgemm(const SaisTGemm* g, void* stream);
/* the K split this library recommends for a SAIS_TG_RAW launch of that shape (>= 1; the slab workspace the caller passes
 * as `out` is nsplit * M * N floats): the sizing rule lives here, not in the host language                               */
int sais_tgemm_nsplit(int M, int N, int K);
/* Row kernels over D = 384 that CONSUME the slabs (split-K reduce + epilogue + LayerNorm in one pass):
 *   sais_temporal_ln_fwd:  y = resid + drop(sum_z slabs[z] + bias) ;  z = LayerNorm(y; gamma, beta, eps) ; mean / rstd saved
 *       = `src = src + dropout1(out_proj(..))` / `src = src + dropout2(linear2(..))` followed by norm1 / norm2 of the
 *         torch-1.8 post-norm TransformerEncoderLayer (prepare_model.py:74-81).  y may be NULL (inference).
 *   sais_temporal_ln_bwd:  dy = sum_z slabs[z] + add (either may be NULL) ;  dx = autograd of that LayerNorm at (x, mean,
 *       rstd) ;  dx_drop = drop(dx) (optional second output: the gradient entering the residual BRANCH) ;
 *       dgamma += sum_m dy xhat ; dbeta += sum_m dy.                                                                  */
int sais_temporal_ln_fwd(const float* slabs, int nslab, long slab_stride, const float* bias, const float* resid, int rows,
                         float p_drop, const unsigned long long* rng_state, unsigned site, float* y, const float* gamma,
                         const float* beta, float eps, float* z, float* mean, float* rstd, void* stream);
int sais_temporal_ln_bwd(const float* slabs, int nslab, long slab_stride, const float* add, const float* x,
                         const float* mean, const float* rstd, const float* gamma, int rows, float* dx, float* dx_drop,
                         float p_drop, const unsigned long long* rng_state, unsigned site, float* dgamma, float* dbeta,
                         void* stream);

/* dW[N1,N2] += P[M,N1]^T . Q[M,N2]  and (db != NULL)  db[N1] += column sums of P.
 * Weight / bias gradients of every nn.Linear above (autograd of F.linear).  Accumulates with f32
 * atomics into dW/db (caller zeroes them: optimizer.zero_grad(), perform_training.py:155).        */
int sais_gemm_tn(const void* P, int ldp, const void* Q, int ldq, int M, int N1, int N2,
                 float* dW, int ldw, float* db, int nsplit, void* stream);
/* Several weight-gradient GEMMs over the same M rows in ONE launch (all bf16): the four nn.Linear of a ViT
 * block give 108 output tiles, so a few M-splits fill the chip and the atomic traffic shrinks accordingly. */
#define SAIS_TN_MAX_ITEMS 48
typedef struct SaisTnItem {
    const void* P; int ldp;        /* bf16 [M,N1]  (dY)  */
    const void* Q; int ldq;        /* bf16 [M,N2]  (X)   */
    int N1, N2;
    float* dW; int ldw;            /* f32 [N1,N2], accumulated */
    float* db;                     /* f32 [N1] or NULL         */
} SaisTnItem;
int sais_gemm_tn_grouped(const SaisTnItem* items, int nitems, int M, int nsplit, void* stream);
/* ABI 10: the same launch with a caller-provided slab workspace.  In the large-tile regime (round 6, csrc/gemm_tn_xl.hip: every
 * N1 % 192 == 0 and N2 % 384 == 0, M % 32 == 0, M >= 8192, >= 48 32-row steps per M-split — the four nn.Linear of a ViT block at
 * training size) the M-splits store their raw 192 x 384 partial tiles into `slabs` (plain 16-B stores) and a second launch sums
 * them in a FIXED order into dW / db: no fp32 atomics (10 splits x 7.1 MB of them per ViT block otherwise, ~50 us of the launch),
 * bit-reproducible weight gradients.  sais_gemm_tn_grouped_slab_bytes returns the bytes that launch uses (0: no slab form applies
 * and `slabs` is ignored); slabs 16-B aligned; NULL = fp32 atomics.  Environment, read once per process: SAIS_TN_XL = 0 selects
 * the 128 x 384 kernel of rounds 2-5 (whose own slab form stays opt-in: SAIS_TN_SLABS = 1), SAIS_TN_XL = 8 the eight-wave form
 * of the large tile, SAIS_TN_XL_SLABS = 0 its atomics (LABNOTES R6.1).                                                      */
size_t sais_gemm_tn_grouped_slab_bytes(const SaisTnItem* items, int nitems, int M);
int sais_gemm_tn_grouped_ws(const SaisTnItem* items, int nitems, int M, int nsplit, void* slabs, size_t slab_bytes, void* stream);
/* the same with FP32 P and Q (rounded to bf16 while staging, like sais_gemm_tn_f32): one launch for the four weight
 * gradients of a temporal-encoder layer                                                                        */
int sais_gemm_tn_grouped_f32(const SaisTnItem* items, int nitems, int M, int nsplit, void* stream);
/* same as sais_gemm_tn with f32 P and Q (rounded to bf16 while staging; f32 accumulation) */
int sais_gemm_tn_f32(const void* P, int ldp, const void* Q, int ldq, int M, int N1, int N2,
                     float* dW, int ldw, float* db, int nsplit, void* stream);

/* ---------------------------------------------------------------- GEMM with LayerNorm in the epilogue (N = 384)
 * The ViT block's residual stream is D = 384 wide, so a workgroup that owns a 128-row x 384-column output tile owns
 * whole rows and can normalise them before they leave the registers (SURVEY §7 steps 4-5, §8b `proj_residual`,
 * `fc2_residual`, `ln_qkv`, `ln_fc1_gelu`):
 *   sais_gemm_ln_fwd:  x_out = A . W^T + bias + resid ;  out16 = LayerNorm(x_out; gamma, beta, eps) ; mean/rstd saved.
 *       = `x = x + attn.proj(...)` / `x = x + mlp.fc2(...)` followed by the NEXT `norm2` / `norm1`
 *         (Block.forward, dino-main/vision_transformer.py:107-113; nn.LayerNorm(eps=1e-6) via vit_small :243-247).
 *   sais_gemm_ln_bwd:  dy = A . W^T (the dX of fc1 / qkv: autograd of F.linear, :59-65 / :80-92) followed by autograd
 *       of that LayerNorm:  dx = dres + rstd (dy g - mean(dy g) - xhat mean(dy g xhat)),  dgamma += sum_m dy xhat,
 *       dbeta += sum_m dy.  `resid` is the LayerNorm INPUT x saved by the forward, `mean`/`rstd` its statistics.
 * W is bf16 [384,K] (K % 64 == 0), A bf16 [M,K]; out32 / dres may alias (in-place residual-gradient update).       */
typedef struct SaisGemmLn {
    const void* A; int lda;        /* bf16 [M,K]                                              */
    const void* W; int ldw;        /* bf16 [384,K]                                            */
    int M, K;
    const float* bias;             /* fwd: f32 [384] or NULL; bwd: unused                     */
    const float* resid; int ldr;   /* fwd: residual f32 [M,384]; bwd: LayerNorm input x f32   */
    float* out32; int ldo32;       /* fwd: x_out; bwd: dx (f32, optional)                     */
    void* out16; int ldo16;        /* fwd: LayerNorm(x_out) bf16; bwd: dx bf16 (optional)     */
    const float* gamma; const float* beta; float eps;
    float* mean; float* rstd;      /* fwd: outputs [M] (optional); bwd: inputs                */
    const float* dres; int lddres; /* bwd: residual-stream gradient added to dx (optional)    */
    float* dgamma; float* dbeta;   /* bwd: f32 [384], accumulated (both or neither)           */
    const float* rowscale;         /* fwd, optional (DropPath): x_out = resid + rowscale[m] * (A.W^T + bias)            */
    const float* rowscale16;       /* bwd, optional (DropPath): out16 = bf16(rowscale16[m] * dx): the gradient that enters
                                      the NEXT branch's backward GEMMs; out32 (the residual-stream gradient) is not scaled */
    int dres_period;               /* bwd (ABI 8): 0 = dres is [M,384].  > 0: dres is COMPACT [ceil(M / period), 384]: row m of the
                                      residual-stream gradient is dres[m / period] when m % period == 0 and ZERO otherwise — the
                                      gradient entering the last ViT block exists on the CLS rows only (period = tokens per frame;
                                      VisionTransformer.forward returns x[:, 0], vision_transformer.py:212-214)            */
    const void* xn16; int ldxn16;  /* bwd, optional (ABI 11): the bf16 LayerNorm OUTPUT y = xhat * gamma + beta the forward saved for the
                                      next GEMM and for dW.  With it (and `beta`) the epilogue rebuilds xhat = (y - beta) / gamma from
                                      768 B per row instead of reading the 1536-B fp32 `resid` row (-38.7 MB per launch at config 2);
                                      `resid` must still be valid: a workgroup falls back to it when some |gamma_c| < 1e-3 or
                                      |beta_c| > 64 |gamma_c| (the division would amplify y's bf16 rounding).  xhat enters dx only
                                      through xhat * mean(dy g xhat), a term ~1/sqrt(384) of dy: measured effect on the ViT parameter
                                      gradients <= 1e-3 relative (tests).  SAIS_LN_BWD_X16=0 in the environment ignores the field. */
} SaisGemmLn;
int sais_gemm_ln_fwd(const SaisGemmLn* g, void* stream);
int sais_gemm_ln_bwd(const SaisGemmLn* g, void* stream);

/* ---------------------------------------------------------------- the MLP branch of a ViT block as ONE launch (ABI 8)
 * Block-level entry points (ln_fc1_gelu + fc2_residual of Block.forward in one call).
 *   sais_mlp_fwd:  Mlp.forward (dino-main/vision_transformer.py:49-65: fc1 -> nn.GELU() exact erf -> fc2, dropout p = 0)
 *       + the residual add of Block.forward (:111-112: x = x + drop_path(mlp(norm2(x)))) + the NEXT LayerNorm of the
 *       residual stream (the next block's norm1, :107-108; eps 1e-6 via vit_small :243-247):
 *         u = X . W1^T + bias1;  h = GELU(u);  g = GELU'(u);
 *         out32 = resid + rowscale[m] * (h . W2^T + tail.bias);  out16 = LayerNorm(out32);  mean / rstd saved.
 *       X = LayerNorm2(x_mid) bf16 [M,384]; W1 = fc1.weight bf16 [H,384]; W2 = fc2.weight bf16 [384,H]; H % 128 == 0.
 *       h / g: bf16 [M,H] outputs the backward pass needs (dW2 = d^T h, du = (d W2) g).  g == NULL (inference): GELU'
 *       is not evaluated, and with h == NULL too the hidden activation is never written to HBM at all.
 *       tail.gamma == NULL: no following LayerNorm (the last block): only out32 is written.
 *   sais_mlp_bwd:  autograd of the same branch up to the block's norm2 (backward of :59-65 and of nn.LayerNorm):
 *         du = (X . W1^T) * g  (X = bf16 gradient entering the branch [M,384], W1 = fc2.weight^T bf16 [H,384], g = GELU'(u)
 *         saved by the forward) -> written to h (dW1 = du^T xn2 needs it);  dxn = du . W2^T (W2 = fc1.weight^T bf16 [384,H]);
 *         then exactly sais_gemm_ln_bwd's epilogue on dxn with tail.{resid = x_mid, mean, rstd, gamma, dres, dgamma, dbeta,
 *         out32, out16, rowscale16}.
 * `tail` is read like the argument of sais_gemm_ln_fwd / _bwd; its A / lda / W / ldw / M / K fields are ignored.        */
typedef struct SaisMlp {
    const void* X; int ldx;
    const void* W1; int ldw1;
    const float* bias1;            /* fwd: f32 [H] or NULL; bwd: unused */
    const void* W2; int ldw2;
    int M, H;
    void* h; int ldh;
    void* g; int ldg;
    SaisGemmLn tail;
} SaisMlp;
int sais_mlp_fwd(const SaisMlp* a, void* stream);
int sais_mlp_bwd(const SaisMlp* a, void* stream);

/* ================================================================ block-level entry points (ABI 8)
 * One call = one Block of the ViT (dino-main/vision_transformer.py:95-113: x = x + drop_path(attn(norm1(x)));
 * x = x + drop_path(mlp(norm2(x)))) in either direction: C-side sequencing of the GEMM-level entries above, so that a host
 * in any language drives the encoder with 12 + 12 calls per step instead of re-implementing the ~25-launch plan of
 * sais_amd/vit.py.  Conventions as everywhere: raw device pointers owned by the caller, no allocation, no host sync,
 * asynchronous on `stream`, 0 / negative return.  Scratch comes from the caller: `workspace` of at least
 * sais_workspace_bytes(op, frames, ntok) bytes, 256-B aligned.  The LayerNorms sit where the kernels fuse them: a forward call
 * CONSUMES norm1(x_in) (xn1, written by the previous block's call, or by sais_layernorm_fwd for block 0) and PRODUCES the next
 * block's norm1(x_out) (xn_next; next_norm_g == NULL for the last block).                                               */
enum SaisOp { SAIS_OP_VIT_BLOCK_FWD = 0, SAIS_OP_VIT_BLOCK_BWD = 1, SAIS_OP_TEMPORAL_LAYER_FWD = 2, SAIS_OP_TEMPORAL_LAYER_BWD = 3 };
size_t sais_workspace_bytes(int op, int frames, int ntok);   /* temporal ops: (op, sequences B, tokens S) */
/* (ABI 12) bytes per element of the GELU' tensor sais_vit_block_fwd writes and sais_vit_block_bwd reads: 1 = one-byte codes
 * (default; epilogues 13 / 14), 2 = bf16 (environment SAIS_GELU_GRAD_Q8=0, read once per process; epilogues 10 / 11). */
int sais_gelu_grad_bytes(void);

typedef struct SaisVitBlockParams {
    /* bf16 weight shadows [out,in] + f32 biases (nn.Linear of Attention :68-92 and Mlp :49-65), f32 LayerNorm parameters */
    const void* qkv_w;  const float* qkv_b;       /* [1152,384] */
    const void* proj_w; const float* proj_b;      /* [384,384]  */
    const void* fc1_w;  const float* fc1_b;       /* [1536,384] */
    const void* fc2_w;  const float* fc2_b;       /* [384,1536] */
    const float* norm1_g; const float* norm1_b;   /* backward only (norm1_b: ABI 11, may be NULL = fp32 LayerNorm input path) */
    const float* norm2_g; const float* norm2_b;
    const float* next_norm_g; const float* next_norm_b;   /* forward: norm1 of the NEXT block, or NULL */
    /* backward only: transposed bf16 shadows [in,out] (dX = dY . W runs on the same NT kernels) and f32 gradient accumulators */
    const void* qkv_wt; const void* proj_wt; const void* fc1_wt; const void* fc2_wt;
    float* d_qkv_w; float* d_qkv_b; float* d_proj_w; float* d_proj_b; float* d_fc1_w; float* d_fc1_b; float* d_fc2_w; float* d_fc2_b;
    float* d_norm1_g; float* d_norm1_b; float* d_norm2_g; float* d_norm2_b;
} SaisVitBlockParams;

typedef struct SaisVitBlockFwd {
    int frames, ntok;                /* M = frames * ntok token rows; ntok = 197 or 37 */
    const void* xn1;                 /* in : bf16 [M,384] norm1(x_in)                                             */
    const float* x_in;               /* in : f32 [M,384] residual stream                                          */
    void* qkv;                       /* out: bf16 [M,1152]   (saved for backward)                                 */
    void* attn_out;                  /* out: bf16 [M,384]    (saved)                                              */
    float* lse;                      /* out: f32 [frames,6,ntok] (saved; NULL in inference)                        */
    float* x_mid;                    /* out: f32 [M,384] x after the attention branch (may alias x_in in inference) */
    void* xn2;                       /* out: bf16 [M,384] norm2(x_mid) (saved)                                     */
    float* mean2; float* rstd2;      /* out: f32 [M] (saved; NULL in inference)                                    */
    void* h;                         /* out: bf16 [M,1536] GELU(u) (saved; NULL in inference: workspace is used)    */
    void* gelu_grad;                 /* out: [M,1536] GELU'(u), sais_gelu_grad_bytes() bytes per element: one-byte codes (ABI 12,
                                        SAIS_EPI_BIAS_GELU_GRADQ_BF16) or bf16 (saved; NULL in inference: not evaluated)       */
    float* x_out;                    /* out: f32 [M,384] (may alias x_mid in inference)                            */
    void* xn_next;                   /* out: bf16 [M,384] next block's norm1(x_out) (with next_norm_g)             */
    float* mean_next; float* rstd_next;   /* out: f32 [M] (optional)                                              */
    const float* rowscale_attn; const float* rowscale_mlp;   /* DropPath row scales f32 [M] of the two branches, or NULL */
} SaisVitBlockFwd;
int sais_vit_block_fwd(const SaisVitBlockParams* w, const SaisVitBlockFwd* a, void* workspace, size_t ws_bytes, void* stream);

typedef struct SaisVitBlockBwd {
    int frames, ntok;
    /* what the forward call saved */
    const float* x_in; const float* mean1; const float* rstd1; const void* xn1; const void* qkv; const void* attn_out;
    const float* lse; const float* x_mid; const float* mean2; const float* rstd2; const void* xn2; const void* h;
    const void* gelu_grad;
    float* dx;                       /* in/out: f32 [M,384] gradient of the residual stream (x_out's on entry, x_in's on return) */
    const void* dx16_in;             /* in : bf16 [M,384] = bf16(rowscale_mlp * dx): what enters the MLP branch           */
    void* dx16_out;                  /* out: bf16 [M,384] = bf16(rowscale_prev * dx on return): the next call's dx16_in    */
    const float* rowscale_attn;      /* DropPath scale of THIS block's attention branch, or NULL                           */
    const float* rowscale_prev;      /* DropPath scale of the PREVIOUS block's MLP branch (applied to dx16_out), or NULL   */
    int defer_dw;                    /* (ABI 13) != 0: the weight / bias gradients of this block are NOT launched by this call;
                                        sais_vit_blocks_dw launches them for several blocks at once.  Until then the caller keeps
                                        this struct's tensors, the workspace of this call and dx16_in untouched (dx16_out must
                                        then be a different buffer than dx16_in)                                             */
} SaisVitBlockBwd;
int sais_vit_block_bwd(const SaisVitBlockParams* w, const SaisVitBlockBwd* a, void* workspace, size_t ws_bytes, void* stream);
/* (ABI 13) The weight / bias gradients of `nblocks` blocks whose sais_vit_block_bwd calls ran with defer_dw, as ONE grouped launch of
 * 4 nblocks GEMMs (dW += P^T Q over the M = frames x ntok rows; LABNOTES R6.8).  One launch per block cuts M into ten splits so
 * that its 24 tiles fill the chip, and every split writes a partial tile that a second launch sums; two blocks per launch need five
 * splits, ten blocks none: 187 -> 160 -> 149 us per block at M = 50 432.  w[i], a[i], workspaces[i] = what call i was given
 * (each workspace >= sais_workspace_bytes(SAIS_OP_VIT_BLOCK_BWD, ...) bytes = ws_bytes). */
int sais_vit_blocks_dw(const SaisVitBlockParams* const* w, const SaisVitBlockBwd* const* a, void* const* workspaces, size_t ws_bytes,
                       int nblocks, const SaisTnItem* extra, int nextra, void* stream);
/* extra / nextra: further GEMMs over the same M rows to ride in the launch (the k / v weight gradient of the CLS-only last block:
 * 4 more tiles); 4 nblocks + nextra <= SAIS_TN_MAX_ITEMS. */

/* One layer of the temporal encoder per call: the torch-1.8 POST-norm nn.TransformerEncoderLayer(d 384, 4 heads, FF 2048,
 * ReLU, dropout 0.1, LayerNorm eps 1e-5) of prepare_model.py:74-81 as patched by README.md:43-48 (returns the attention map):
 *   src = norm1(src + dropout1(out_proj(MHA(src))));  src = norm2(src + dropout2(linear2(dropout(relu(linear1(src)))))).
 * 7 launches forward, 8 backward, sequenced by the library; fp32 tensors, linears on the bf16x3 path (sais_tgemm).
 * Dropout: p_drop = 0 for eval(); sites site0 .. site0 + 3 = attention weights, dropout1, FFN dropout, dropout2.        */
typedef struct SaisTemporalLayerParams {
    const float* in_proj_w;  const float* in_proj_b;    /* [1152,384], [1152] */
    const float* out_proj_w; const float* out_proj_b;   /* [384,384]          */
    const float* linear1_w;  const float* linear1_b;    /* [2048,384]         */
    const float* linear2_w;  const float* linear2_b;    /* [384,2048]         */
    const float* norm1_g; const float* norm1_b; const float* norm2_g; const float* norm2_b;
    /* backward only: fp32 transposed copies [in,out] and gradient accumulators */
    const float* in_proj_wt; const float* out_proj_wt; const float* linear1_wt; const float* linear2_wt;
    float* d_in_proj_w; float* d_in_proj_b; float* d_out_proj_w; float* d_out_proj_b; float* d_linear1_w; float* d_linear1_b;
    float* d_linear2_w; float* d_linear2_b; float* d_norm1_g; float* d_norm1_b; float* d_norm2_g; float* d_norm2_b;
} SaisTemporalLayerParams;

typedef struct SaisTemporalLayerFwd {
    int B, S;                            /* B sequences of S tokens: M = B * S rows; S <= 96                                */
    const float* z;                      /* in : f32 [M,384]                                                               */
    const unsigned char* key_pad;        /* in : u8 [B,S], 1 = masked key                                                  */
    float* qkv; float* ctx;              /* out: f32 [M,1152], [M,384] (saved for backward)                                */
    float* attn_avg;                     /* out, optional: f32 [B,S,S] head-averaged attention weights (the returned map)  */
    float* y1;                           /* out, optional (saved): f32 [M,384] input of norm1                               */
    float* z1; float* mean1; float* rstd1;   /* out: norm1 output (saved) and its statistics (optional)                     */
    float* h;                            /* out: f32 [M,2048] drop(relu(linear1(z1))) (saved)                               */
    float* y2;                           /* out, optional (saved): input of norm2                                           */
    float* z_out; float* mean2; float* rstd2;
    float p_drop; const unsigned long long* rng_state; unsigned site0;
} SaisTemporalLayerFwd;
int sais_temporal_layer_fwd(const SaisTemporalLayerParams* w, const SaisTemporalLayerFwd* a, void* workspace, size_t ws_bytes,
                            void* stream);

typedef struct SaisTemporalLayerBwd {
    int B, S;
    const float* z; const float* qkv; const float* ctx; const float* y1; const float* mean1; const float* rstd1;
    const float* z1; const float* h; const float* y2; const float* mean2; const float* rstd2; const unsigned char* key_pad;
    /* gradient wrt the layer OUTPUT = sum of nslab raw split-K slabs (slab_stride floats apart; the next layer's dx_slabs) +
     * dz_add; either part may be NULL / 0                                                                                  */
    const float* dz_slabs; int nslab; long slab_stride; const float* dz_add;
    /* gradient wrt the layer INPUT, in the same two-part form: dx_slabs f32 [sais_tgemm_nsplit(M,384,1152)][M][384] (raw dX of
     * in_proj) and dx_add f32 [M,384] (the residual path)                                                                  */
    float* dx_slabs; float* dx_add;
    float p_drop; const unsigned long long* rng_state; unsigned site0;
    /* optional (ABI 9): SaisTnItem[4] on the HOST.  When set, the layer's four weight / bias gradient GEMMs are NOT launched:
     * their items (operands inside `workspace` and the saved tensors) are written here, and the caller launches them later
     * with sais_gemm_tn_grouped_f32(items, n, B * S, 1, stream) — e.g. all layers of an encoder in ONE launch — and must keep
     * this call's `workspace` untouched until then (one workspace per layer).                                              */
    SaisTnItem* dw_items_out;
} SaisTemporalLayerBwd;
int sais_temporal_layer_bwd(const SaisTemporalLayerParams* w, const SaisTemporalLayerBwd* a, void* workspace, size_t ws_bytes,
                            void* stream);

/* ---------------------------------------------------------------- LayerNorm over dim = 384
 * nn.LayerNorm in Block / final norm (vision_transformer.py:99,103,107-113,212; eps 1e-6 from
 * vit_small :243-247) and norm1/norm2 of the post-norm TransformerEncoderLayer (prepare_model.py:74-81;
 * eps 1e-5).  Row r of x starts at x + r*ldx (so the CLS-only final norm is ldx = 197*384).
 * Outputs are optional (NULL): y_bf16 feeds the next MFMA GEMM, y_f32 is the post-norm residual,
 * mean/rstd are saved for backward.                                                              */
int sais_layernorm_fwd(const float* x, long ldx, int rows, int dim, const float* gamma, const float* beta,
                       float eps, void* y_bf16, long ldy16, float* y_f32, long ldy32, float* mean, float* rstd,
                       void* stream);
/* dy = dy_bf16 (optional) + dy_f32 (optional);  dx = dres (optional) + dLN(dy);  dgamma/dbeta += (atomics). */
int sais_layernorm_bwd(const void* dy_bf16, long lddy16, const float* dy_f32, long lddy32, const float* x, long ldx,
                       const float* mean, const float* rstd, const float* gamma, const float* dres, long lddres,
                       int rows, int dim, float* dx_f32, long lddx32, void* dx_bf16, long lddx16, float* dgamma,
                       float* dbeta, const float* rowscale16 /*optional: dx_bf16 = bf16(rowscale16[row] * dx), DropPath*/,
                       float* dx_f32_drop /*optional, ld = lddx32: dropout(dx) with mask element row * 384 + col of `site`:
                       the gradient that enters a dropped branch of the temporal encoder layer*/,
                       float p_drop, const unsigned long long* rng_state, unsigned site, void* stream);

/* ---------------------------------------------------------------- ViT spatial attention (197 tokens, 6 heads x 64)
 * Attention.forward core, vision_transformer.py:83-90: softmax(q k^T / 8) v per (frame, head).
 * qkv bf16 [frames*ntok, 3*384] laid out as the qkv Linear writes it (q | k | v, head-major inside).
 * out bf16 [frames*ntok, 384]; lse f32 [frames,6,ntok] (saved for backward, optional);
 * probs f32 [frames,6,ntok,ntok] (optional: get_last_selfattention, :216-223).
 * ntok = 197 (224 x 224 frames) or 37 (the 96 x 96 local crops of DINO pre-training, main_dino.py:658-663;
 * prepare_tokens at that size, vision_transformer.py:196-207); anything else is SAIS_ERR_ARG.     */
int sais_vit_attn_fwd(const void* qkv, long ldqkv, int frames, int ntok, void* out, long ldo, float* lse, float* probs,
                      void* stream);
/* dqkv bf16 [frames*ntok, 1152] from dout bf16 [frames*ntok, 384] and the saved forward output `out` (same shape:
 * delta_q = sum_d dout[q,d] out[q,d] is the softmax-backward row term, computed in the kernel).  One single-pass
 * kernel: P is rebuilt once per (query, key) from lse.  delta_ws is unused since ABI 2 (may be NULL).              */
int sais_vit_attn_bwd(const void* qkv, long ldqkv, const void* dout, long lddo, const void* out, long ldout,
                      const float* lse, float* delta_ws, int frames, int ntok, void* dqkv, long lddqkv, void* stream);
/* The LAST block's attention (ABI 8).  VisionTransformer.forward returns norm(x)[:, 0] (vision_transformer.py:209-214): of the
 * last block only the CLS row of every frame is read, so of its Attention.forward (:80-92) only the CLS QUERY is needed (keys
 * and values of all tokens still are).  out bf16 [frames, 384] (compact: one row per frame).  The backward takes the compact
 * dout bf16 [frames, 384], recomputes P and writes the WHOLE dqkv bf16 [frames*ntok, 1152]: dk, dv for every token, dq on the
 * CLS rows and zeros on the others.  Results equal sais_vit_attn_fwd / _bwd restricted to those rows (fp32 softmax here). */
int sais_vit_attn_cls_fwd(const void* qkv, long ldqkv, int frames, int ntok, void* out, long ldo, void* stream);
int sais_vit_attn_cls_bwd(const void* qkv, long ldqkv, const void* dout, long lddo, int frames, int ntok, void* dqkv,
                          long lddqkv, void* stream);

/* ---------------------------------------------------------------- optical-flow stage: RAFT correlation volume (ABI 8)
 * The reference renders its flow maps with ptlflow's `raft` ('things' checkpoint): SAIS/scripts/extract_representations.py
 * :33,62-67,221-252,267 and main.sh:18.  ptlflow 0.2.5 is a third-party dependency that is ABSENT from the reference tree
 * and this image: PARITY UNPINNED — these entries follow the published model (Teed & Deng, ECCV 2020) and are tested against
 * oracle/raft_oracle.py.  The all-pairs correlation itself is sais_gemm_nt_f32 on the [H W, 256] feature matrices.
 *   sais_raft_corr_pool: levels 1-3 of the correlation pyramid (2 x 2 average pooling over the last two dims of
 *       [rows, 1, H, W]; corr0 row r starts at corr0 + r * ld0) in one pass.  l1 / l2 / l3: f32 [rows, (H>>l) * (W>>l)].
 *   sais_raft_lookup: out f32 [B, 4 (2r+1)^2, H, W]; channel l (2r+1)^2 + a (2r+1) + b = level l sampled (bilinear, zeros
 *       outside, pixel coordinates) at (x / 2^l + a - r, y / 2^l + b - r) for (x, y) = coords[b', :, y', x'].          */
int sais_raft_corr_pool(const float* corr0, long ld0, int rows, int H, int W, float* l1, float* l2, float* l3, void* stream);
int sais_raft_lookup(const float* l0, long ld0, const float* l1, const float* l2, const float* l3, const float* coords,
                     int B, int H, int W, int radius, float* out, void* stream);
/* Convolution as a matrix-core GEMM (ABI 10; the RAFT encoders / update block, extract_representations.py:221-252 via ptlflow's
 * `raft`): cols f32 [rows, ld] <- im2col of ONE image x f32 [C, H, W]; row = output position oy * Wo + ox, column
 * (c * kh + ky) * kw + kx = the order of weight.view(Cout, C * kh * kw); column C*kh*kw holds 1.0 (append the bias to the weight
 * matrix as that column), the remaining columns and the rows >= Ho * Wo are zero (pad ld to a multiple of 64 and rows to a
 * multiple of 128 for sais_gemm_nt_f32).  y[Cout, Ho * Wo] = sais_gemm_nt_f32(A = weights [Cout, ld], B = cols) is NCHW.    */
int sais_im2col_f32(const float* x, int C, int H, int W, int kh, int kw, int sh, int sw, int ph, int pw,
                    float* cols, int ld, int rows, void* stream);

/* ---------------------------------------------------------------- ViT embedding glue
 * PatchEmbed + prepare_tokens, vision_transformer.py:116-131,196-207.                            */
int sais_patchify(const float* frames_f32 /*[F,3,side,side]*/, int frames, int side /* % 16 == 0 */,
                  void* patches_bf16 /*[F*(side/16)^2,768]*/, void* stream);
int sais_vit_cls_rows(const float* cls, const float* pos0, float* tokens, long frame_stride, int frames, int dim,
                      void* stream);
/* dcls/dpos += reductions of dtokens f32 [F,ntok,dim]; dpatch_bf16 [F*(ntok-1), dim] = rows 1.. (dY of patch GEMM) */
int sais_vit_embed_bwd(const float* dtokens, int frames, int ntok, int dim, float* dcls, float* dpos,
                       void* dpatch_bf16, void* stream);

/* Prefetch hint (ABI 8): reads [p, p + bytes) once with discarded loads so that the range is cache-resident (L2 /
 * Infinity Cache) for the kernels that follow — the temporal encoder's ~70 small launches otherwise each pay a first-touch
 * HBM round trip for weights the ViT's traffic evicted.  No effect on results.  p 16-B aligned.                       */
int sais_touch(const void* p, long bytes, void* stream);

/* ---------------------------------------------------------------- optimizer + weight shadows
 * optim.SGD(params, lr) — prepare_model.py:566-567, perform_training.py:155-158 (no momentum / wd).
 * param -= lr * grad_scale * grad; shadow_bf16 (optional) refreshed in the same pass.            */
int sais_sgd_step(float* param, const float* grad, void* shadow_bf16, long n, float lr, float grad_scale, void* stream);
int sais_cast_bf16(const float* src, void* dst_bf16, long n, void* stream);
int sais_transpose_cast_bf16(const float* src, int rows, int cols, void* dst_bf16 /*[cols,rows]*/, void* stream);
int sais_transpose_f32(const float* src, int rows, int cols, float* dst /*[cols,rows]*/, void* stream);
/* All transposed weight shadows of a model in ONE launch (the per-weight launches above are ~5 us each and a ViT-S
 * has 48 of them per optimizer step).  `items` is a DEVICE array, built once per model: the addresses are fixed.  */
typedef struct SaisTransposeItem {
    const float* src;        /* [rows, cols] f32                                              */
    void*        dst;        /* [cols, rows] bf16 (dst_is_f32 == 0) or f32                     */
    int rows, cols;
    int tile_begin;          /* prefix sum of ceil(rows/32)*ceil(cols/32) over preceding items */
    int reserved;
} SaisTransposeItem;
int sais_transpose_batch(const SaisTransposeItem* items_dev, int nitems, int total_tiles, int dst_is_f32, void* stream);
int sais_scale_f32(float* p, long n, float s, void* stream);

/* ---------------------------------------------------------------- frame preprocessing (uint8 frames -> ViT input)
 * SurgDataset.__getitem__, dino-main/main_dino.py:295-316 + the transform of extract_representations.py:158-162:
 * CenterCrop((height_frac*H, width_frac*W)) -> Resize((224,224)) -> ToTensor -> Normalize(mean, std), bit-identical to
 * torchvision 0.9.0 + Pillow's 8-bit bilinear resampler.  A plan holds the per-geometry coefficient tables on the
 * device (built once: allocation + upload); sais_preprocess_run only launches, so it can be graph-captured.
 * frames: device uint8 [F, H, W, 3] (RGB, as decoded); out: device float32 [F, 3, 224, 224].                      */
typedef struct SaisPreprocessPlan SaisPreprocessPlan;
int  sais_preprocess_plan_create(int H, int W, double height_frac, double width_frac, const float* mean3,
                                 const float* std3, SaisPreprocessPlan** plan);
int  sais_preprocess_plan_box(const SaisPreprocessPlan* plan, int* box4 /* left, top, right, bottom */);
int  sais_preprocess_run(const SaisPreprocessPlan* plan, const unsigned char* frames, int nframes, float* out, void* stream);
void sais_preprocess_plan_destroy(SaisPreprocessPlan* plan);

/* ---------------------------------------------------------------- temporal encoder glue (dim 384, 4 heads x 96)
 * prepareInputForTransformer, prepare_model.py:179-195: z[b,0] = frame_cls, z[b,1+t] = x[b,t] + pos[t]
 * (out of place: the reference's in-place += on the caller's tensor is NOT reproduced).          */
int sais_temporal_prepare_fwd(const float* x, long x_clip_stride, long x_frame_stride, const float* pos /*[T,384]*/,
                              const float* cls, int B, int T, float* z_f32, void* z_bf16 /*optional*/, void* stream);
/* backward: dz = dz_f32 + sum_z slabs[z] (either may be NULL; slabs = the raw split-K output of the in_proj dX GEMM) */
int sais_temporal_prepare_bwd(const float* dz_f32, const float* slabs, int nslab, long slab_stride, int B, int T, float* dx,
                              long dx_clip_stride, long dx_frame_stride, int accumulate, float* dpos, float* dcls,
                              void* stream);
/* nn.MultiheadAttention core of the torch-1.8 post-norm TransformerEncoderLayer (prepare_model.py:74-81,
 * called at :213): q scaled by 96^-0.5, key_pad[b][j] != 0 -> -inf, softmax, (dropout,) P v — exact fp32 on the matrix
 * cores (v_mfma_f32_16x16x4_f32), one workgroup per (sequence, head).
 * attn_avg f32 [B,S,S] (optional) = P averaged over the 4 heads = the README.md:43-48 attention map. */
#define SAIS_TEMPORAL_MAX_S_FWD 96
#define SAIS_TEMPORAL_MAX_S_BWD 96
int sais_temporal_attn_fwd(const float* qkv /*[B*S,1152]*/, const unsigned char* key_pad /*[B,S]*/, int B, int S,
                           float* ctx /*[B*S,384]*/, float* attn_avg, float p_drop, const unsigned long long* rng_state,
                           unsigned site, void* stream);
/* dctx = sum over nslab raw split-K slabs (slab_stride floats apart) of the out_proj dX GEMM; nslab = 1: a plain tensor */
int sais_temporal_attn_bwd(const float* qkv, const unsigned char* key_pad, int B, int S, const float* dctx, int nslab,
                           long slab_stride, float* dqkv /*[B*S,1152]*/, float p_drop, const unsigned long long* rng_state,
                           unsigned site, void* stream);
/* Train mode (model.train(), train.py:59): nn.TransformerEncoderLayer's default dropout = 0.1 (prepare_model.py:75) acts at
 * four sites per layer: on the attention weights (p_drop above: after the softmax, before P v; the returned map is the
 * dropped one, as in torch 1.8), after out_proj (dropout1), after the ReLU (dropout) and after linear2 (dropout2).
 * Masks come from Philox4x32-10 keyed by rng_state = {seed, offset} in DEVICE memory (graph replays draw fresh masks:
 * sais_rng_advance is a graph node), the site id and the element index; forward and backward regenerate them, nothing is
 * stored.  sais_dropout_f32: out = (resid ? resid : 0) + x * keep / (1 - p) (out may alias x); applied to a gradient with
 * the same (state, site) it is the backward of itself.  sais_dropout_mask exports the keep mask (tests feed it to the oracle).
 * The stream of numbers is this library's, not torch's: a torch run with the same seed draws different masks.            */
int sais_rng_advance(unsigned long long* state /*{seed, offset}*/, void* stream);
int sais_dropout_f32(const float* x, const float* resid /*optional*/, float* out, long n, float p,
                     const unsigned long long* rng_state, unsigned site, void* stream);
int sais_dropout_mask(unsigned char* mask, long n, float p, const unsigned long long* rng_state, unsigned site, void* stream);
/* DropPath / stochastic depth of the ViT blocks in train mode (vision_transformer.py:27-46,105-113; rates
 * linspace(0, drop_path_rate, depth), :150): x = x + keep_f / (1 - p) * branch(x), one Bernoulli(1 - p) draw per sample f.
 * sais_droppath_scales fills the per-ROW scale arrays of `nbranch` branches in one launch (out f32 [nbranch][samples *
 * rows_per_sample]; keep from the same Philox state as the dropout, site = site0 + branch, element = sample); they are
 * the `rowscale` of SaisGemm (SAIS_EPI_BIAS_RESID_F32) / SaisGemmLn forward and the `rowscale16` of the LayerNorm backward
 * entry points.  sais_cast_bf16_rows: dst = bf16(rowscale[row] * src) (the first branch gradient of the backward pass). */
int sais_droppath_scales(float* out, const float* rates_dev /*[nbranch]*/, int nbranch, int samples, int rows_per_sample,
                         const unsigned long long* rng_state, unsigned site0, void* stream);
int sais_cast_bf16_rows(const float* src, const float* rowscale, void* dst_bf16, long rows, int dim, void* stream);

/* ---------------------------------------------------------------- head + SupCon / prototype loss
 * fullModel.forward Prototypes branch, prepare_model.py:215,220,381-416:
 *   rep = mean_s relu(z_rgb[b,s,0]) (+ mean_s relu(z_flow[b,s,0]));  emb = linear(relu(rep))   (256 outputs);
 *   s runs over the `nsnippets` snippets of clip b (:381-382; sequences b*nsnippets + s of the encoder output).
 * The two streams may have different padded lengths (sequence strides): at inference the flow stream
 * has 1-2 frames per 15-frame window (prepare_dataset.py:2660-2666).                              */
int sais_head_fwd(const float* z_rgb, const float* z_flow, long clip_stride, long clip_stride_flow, int B,
                  int nsnippets, const float* W /*[256,384]*/, const float* bias,
                  const unsigned char* use_b /*[B] or NULL*/, const float* WB, const float* biasB,
                  float* rep /*[B,384] saved*/, float* emb /*[B,256]*/, void* stream);
/* use_b: multi-domain models ('+' in the domain name, two-stream, prepare_model.py:405-414): clips flagged 1 (domain !=
 * 'NH_02') go through linearB = (WB, biasB) instead of linear = (W, bias); the backward routes dz through the same head
 * and accumulates each clip's weight / bias gradient into the head it used (dW / dbias or dWB / dbiasB).              */
int sais_head_bwd(const float* demb, const float* W, const float* rep, const float* z_rgb, const float* z_flow,
                  long clip_stride, long clip_stride_flow, int B, int nsnippets, const unsigned char* use_b,
                  const float* WB, float* dW, float* dbias, float* dWB, float* dbiasB, float* dz_rgb, float* dz_flow,
                  void* stream);
/* MIL pathway, inference direction (fullModel.forward task 'MIL', prepare_model.py:356-361; its training raises inside the
 * reference).  sais_mil_forward: tokens[b, s] = relu(z_rgb[(b * ns + s), CLS row]) + clip_pos[s] = the input of
 * getClipReps' clip-level encoder (:452-460; z_rgb = frame-encoder output, seq_stride floats between sequences).  The
 * encoder itself is the same kernel chain as the frame encoder on transEncoderClip's weights (no mask, no CLS).
 * sais_mil_head: reps = relu(enc) (:465), then MIL_Head (:470-488): gated attention over the snippets per class
 * (calcAttention :131-138), video representation and score (:140-148).  attention is [nclasses][B][nsnippets].        */
int sais_mil_forward(const float* z_rgb, long seq_stride, const float* clip_pos /*[ns,384]*/, int B, int nsnippets,
                     float* tokens /*[B*ns,384]*/, void* stream);
int sais_mil_head(const float* enc /*[B*ns,384]*/, int B, int nsnippets, int nclasses, const float* WA, const float* bA,
                  const float* WB, const float* bB, const float* w_att /*[3,256]*/, const float* b_att /*[3]*/,
                  const float* w_fin /*[3,384]*/, const float* b_fin /*[3]*/, float* reps, float* logits /*[B,C]*/,
                  float* attention, void* stream);
/* Optional importance head (-il): importance_function = Linear(384 -> 1) on the ReLU'd encoder output sequence,
 * prepare_model.py:55-56,419-421.  out[m] = w . relu(z[m,:]) + b.  bwd ACCUMULATES into dz / dw / db.        */
int sais_importance_fwd(const float* z /*[M,384] pre-ReLU*/, const float* w, const float* b, int M, float* out, void* stream);
int sais_importance_bwd(const float* dlogit /*[M]*/, const float* z, const float* w, int M, float* dz, float* dw,
                        float* db, void* stream);
/* calcImportanceLoss, prepare_miscellaneous.py:48-60 (quirks kept: scalar mean BCE broadcast against the inverted
 * mask with its last entry dropped; mean over label-0 samples, NaN if none).  logits [B,T+1] (slot 0 = CLS),
 * target [B,T], ipad [B,T+1] (1 = masked).  dlogits (optional) [B,T+1] = scale * d loss / d logits.           */
int sais_importance_loss(const float* logits, const float* target, const unsigned char* ipad, const int* labels,
                         int B, int T, float* loss, float* dlogits, float scale, void* stream);

/* calcNCELoss / getProbs, prepare_miscellaneous.py:14-46,111-126: sim = s_hat p_hat^T (the class logits),
 * probs = softmax(sim), loss = -mean log probs[i, label_col[i]].  With demb != NULL also the gradients:
 * demb (written) and dprotos (accumulated), both scaled by loss_scale.                            */
int sais_nce(const float* emb /*[B,256]*/, const float* protos /*[C,256]*/, const int* label_col, int B, int C,
             float* sim, float* probs, float* loss, float* demb, float* dprotos, float loss_scale, void* stream);

/* ================================================================ DINO pre-training objective
 * SAIS/scripts/dino-main/main_dino.py trains the ViT-S/16 encoder by self-distillation: train_one_epoch (:517-576) runs
 * teacher(images[:2]) and student(images) through utils.MultiCropWrapper (utils.py:595-630; 2 global 224 x 224 crops + N
 * local 96 x 96 crops), DINOHead (vision_transformer.py:257-291), DINOLoss (:579-630), utils.clip_gradients
 * (utils.py:132-141), utils.cancel_gradients_last_layer (:144-149), AdamW on utils.get_params_groups (:633-645) and the
 * EMA teacher update (:563-566).  The ViT and the head's nn.Linear layers run on the GEMM / attention entry points
 * above (sais_vit_attn_* with ntok 197 / 37, sais_gemm_nt_f32, sais_gemm_tn_f32); the entries below are the rest.  */

/* lse[r] = log sum_k exp((x[r][k] - center[k]) * scale), center may be NULL.  Teacher rows: scale = 1 / temp with the
 * centre (F.softmax((teacher_output - self.center) / temp), :605); student rows: scale = 1 / student_temp, no centre
 * (F.log_softmax(student_out[v]), :600,614).  n % 4 == 0.                                                            */
int sais_dino_row_lse(const float* x, long ld, int rows, int n, float scale, const float* center, float* lse, void* stream);
/* DINOLoss.forward (:596-619) and its gradient.  student f32 [ncrops*B, n] (row v*B + b = view v of sample b, the
 * chunk(ncrops) layout), teacher f32 [2*B, n], s_lse / t_lse from sais_dino_row_lse.  Writes
 *   loss[0]  = 1/n_terms sum_{iq < 2} sum_{v != iq} mean_b sum_k -q_iq[b,k] log_softmax(s_v[b] / student_temp)[k]
 *   dlogits  = d loss / d student   (f32 [ncrops*B, n])
 * partials: workspace of sais_dino_loss_partials(B, n) floats (fixed-order reduction, no atomics).                  */
int sais_dino_loss_partials(int B, int n);
int sais_dino_loss(const float* student, long lds, const float* teacher, long ldt, const float* center, const float* s_lse,
                   const float* t_lse, int B, int ncrops, int n, float student_temp, float teacher_temp, float* dlogits,
                   long ldd, float* partials, float* loss, void* stream);
/* DINOLoss.update_center (:621-630) in two steps so that the data-parallel all-reduce of the [1, n] column sums
 * (dist.all_reduce(batch_center), :627) sits between them:  out[k] = sum_r x[r][k];  then
 * center = center * momentum + colsum * inv_count * (1 - momentum),  inv_count = 1 / (rows * world_size).           */
int sais_dino_colsum(const float* x, long ld, int rows, int n, float* out, void* stream);
int sais_dino_center_ema(float* center, const float* colsum, int n, float momentum, float inv_count, void* stream);

/* DINOHead.forward pieces (vision_transformer.py:287-291): nn.GELU() between the MLP's Linear layers (f32 in / out; the
 * backward is du = dh * gelu'(u)), F.normalize(x, dim=-1, p=2) (eps 1e-12) and nn.utils.weight_norm of the last layer
 * (:277-281: w = g v / ||v||_row; _bwd ACCUMULATES dv and, when dg != NULL, dg).  One wave per row, dim <= 1024.      */
int sais_gelu_fwd_f32(const float* u, float* h, long n, void* stream);
/* bf16x3 operand image of an f32 matrix [rows, cols] for sais_gemm_nt: dst bf16 [rows, 3 * cols] = [hi | hi | lo] (A side,
 * b_side = 0) or [hi | lo | hi] (B side), hi = bf16(x), lo = bf16(x - hi): one bf16 GEMM over K' = 3 K then sums the
 * three products of the "bf16x3" arithmetic in fp32.  Used for the head's 256 -> 65536 last layer (the logits).        */
int sais_split_bf16x3(const float* src, long ld, int rows, int cols, void* dst_bf16, int b_side, void* stream);
int sais_gelu_bwd_f32(const float* dh, const float* u, float* du, long n, void* stream);
int sais_l2norm_fwd(const float* z, int rows, int dim, float eps, float* out, float* inv, void* stream);
int sais_l2norm_bwd(const float* dout, const float* out, const float* inv, int rows, int dim, float eps, float* dz, void* stream);
int sais_weight_norm_fwd(const float* v, const float* g, int rows, int dim, float* w, float* inv, void* stream);
int sais_weight_norm_bwd(const float* dw, const float* v, const float* g, const float* inv, int rows, int dim, float* dv,
                         float* dg, void* stream);

/* VisionTransformer.interpolate_pos_encoding (vision_transformer.py:174-194) for crops that are not 224 x 224: the
 * bicubic resampling of the 14 x 14 positional grid is a fixed linear map Wm f32 [nout, nin] (built once per
 * resolution by the host, sais_amd/vit.py); out [1 + nout, dim]: row 0 = pos row 0 (the CLS position, :179),
 * rows 1.. = Wm . pos[1..].  _bwd ACCUMULATES the transpose into dpos [1 + nin, dim].                              */
int sais_pos_interp_fwd(const float* Wm, int nout, int nin, const float* pos, int dim, float* out, void* stream);
int sais_pos_interp_bwd(const float* Wm, int nout, int nin, const float* dout, int dim, float* dpos, void* stream);

/* The optimizer tail of train_one_epoch over one flat parameter buffer (sais_amd/flat.py), cut by the host into chunks
 * of <= sais_opt_chunk_elems() elements that never straddle a tensor ("segment").
 * sais_grad_norms: norms[seg] = scale * ||grad of tensor seg||_2 — what utils.clip_gradients computes per parameter
 *   (:135-136); seg_first_chunk int [nseg + 1]; partial_ws f32 [nchunks].  scale / SaisAdamW.grad_scale = 1 / world_size
 *   when `grad` holds the all-reduced SUM over the data-parallel ranks (DDP averages, main_dino.py:413), else 1.
 * sais_adamw_ema_step, per element of every chunk, in this order:
 *   scale  g *= grad_scale
 *   clip   g *= min(1, clip / (norm + 1e-6))                       utils.py:137-140  (clip = 0: off, main_dino.py:547)
 *   AdamW  p *= 1 - lr * wd (segments flagged SAIS_OPT_DECAY: the `regularized` group, utils.py:633-645; the others
 *          decay 0);  m = lerp(m, g, 1 - beta1);  v = beta2 v + (1 - beta2) g^2;
 *          p -= lr / bc1 * m / (sqrt(v) / sqrt_bc2 + eps)          torch.optim.AdamW, main_dino.py:446,556
 *          bc1 = 1 - beta1^t, sqrt_bc2 = sqrt(1 - beta2^t) per step-count class: class 1 (SAIS_OPT_CLASS1) = the last
 *          layer, whose step count lags while frozen1 != 0 (utils.cancel_gradients_last_layer, utils.py:144-149:
 *          p.grad = None -> the optimizer skips the tensor entirely); SAIS_OPT_NO_GRAD: requires_grad False (weight_g)
 *   EMA    teacher = teacher * ema_m + (1 - ema_m) * p              main_dino.py:563-566 (teacher may be NULL)
 *   and the bf16 MFMA shadows of both (param16 / teacher16, may be NULL) are refreshed in the same pass.            */
#define SAIS_OPT_DECAY 1
#define SAIS_OPT_CLASS1 2
#define SAIS_OPT_NO_GRAD 4
typedef struct SaisOptChunk { long off; int len; int seg; } SaisOptChunk;
typedef struct SaisAdamW {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;     /* flat f32 buffers, same layout   */
    float* teacher; void* param16; void* teacher16;                         /* optional                        */
    const SaisOptChunk* chunks; int nchunks;                                /* device table                    */
    const int* seg_flags; const float* norms;                               /* device, per segment             */
    float clip, lr, weight_decay, beta1, beta2, eps;
    float bc1[2], sqrt_bc2[2]; int frozen1;
    float ema_m;
    float grad_scale;
} SaisAdamW;
int sais_opt_chunk_elems(void);
int sais_grad_norms(const float* grad, const SaisOptChunk* chunks, int nchunks, const int* seg_first_chunk, int nseg,
                    float scale, float* partial_ws, float* norms, void* stream);
int sais_adamw_ema_step(const SaisAdamW* a, void* stream);

#ifdef __cplusplus
}
#endif
#endif
