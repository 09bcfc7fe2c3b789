// Fused MLP branch of a ViT block for gfx950 (Mlp.forward + the residual add + the following LayerNorm of Block.forward,
// dino-main/vision_transformer.py:49-65,107-113) — and its dX backward — as ONE row-owning kernel per direction:
//
//   forward :  u = xn2 . W1^T + b1 ;  h = GELU(u) (exact erf), g' = GELU'(u) ;  x_out = x_mid + s_m (h . W2^T + b2) ;
//              xn = LayerNorm(x_out) (the NEXT block's norm1) — h and g' are still written (the backward pass needs them:
//              dW2 = d^T h, du = (d W2) g'), but h is never READ back from HBM, and in inference it is never written.
//   backward:  du = (d . W2) g' ;  dxn = du . W1 ;  dx = dres + LayerNorm'(dxn) — du is still written (dW1 = du^T xn2).
//
// Replaces gemm_nt<GELU_GRAD> (write-bound: 310 MB out, MFMAs idle) + gemm_ln_fwd (MFMA / read-bound, stores idle), which as
// two launches add up (119 + 100 us per block at M = 50 432), and gemm_nt<MUL> + gemm_ln_bwd (109 + 111 us).
//
// Shape of the kernel.  A workgroup owns <= 112 whole rows (7 MFMA row tiles) and keeps them for the whole hidden dimension;
// four waves, ONE per SIMD, 512 registers each:
//   * the [112, 384] bf16 input tile is loaded ONCE into LDS (84 KiB, LDS-DMA, XOR-swizzled 128-B rows) and is the
//     MFMA "B" operand of the first GEMM for every hidden chunk;
//   * the hidden dimension is walked in chunks of 128: wave w computes u[:, 32 w .. 32 w + 31] (7 x 2 tiles, K = 384),
//     applies the chunk epilogue (bias + GELU / GELU', or x g'), writes the bf16 chunk to HBM (16 B per lane and row) AND
//     into a double-buffered LDS image (28 KiB each), from where all four waves read it as the "B" operand of the
//     second GEMM: out[:, 96 w .. 96 w + 95] += chunk . W2[:, chunk]^T (7 x 6 tiles = 168 accumulators per wave);
//   * the weights never touch LDS: every wave loads exactly the W1 rows / W2 rows it multiplies with, in MFMA fragment
//     layout, straight from global memory (L2-resident: 2.4 MB for both) into registers — 96 + 96 VGPRs, each set
//     reloaded during the OTHER GEMM's phase.  No LDS-DMA ring, no per-K-step barrier: ONE barrier per hidden chunk
//     (336 MFMAs per wave), against one per 16-28 MFMAs in the stand-alone kernels;
//   * phases are software-pipelined inside the single instruction stream of a wave: the chunk epilogue of chunk c (VALU:
//     ~19 issue slots per element pair, and the stores) is issued together with the second GEMM of chunk c - 1 (MFMA),
//     then the first GEMM of chunk c + 1 follows;
//   * the final epilogue is the row-streaming LayerNorm epilogue of the row-owning GEMMs (gemm_row_epi.hpp).
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"
#include "../../include/sais_hip.h"
#include "gemm_row_epi.hpp"

namespace {

constexpr int HC = 128;                      // hidden columns per chunk
constexpr int KA = 384;                      // K of the first GEMM (the residual-stream width)
constexpr int XBLK = 112 * 128;              // one 64-deep k block of a 112-row tile: 112 rows x 128 B
constexpr int NXB = KA / 64;                 // 6 blocks of the resident input tile
constexpr int MLP_LDS = NXB * XBLK + 2 * 2 * XBLK;      // 143 360 B

enum { MLP_FWD_SAVE = 0, MLP_FWD = 1, MLP_BWD = 2 };

// SAIS_MLP_ABL (timing-only builds, tools/gpu_mlp_abl.sh; results are wrong by construction): 1 = the weights are loaded once
// and never reloaded; 2 = no LDS fragment reads inside the chunk loop (the first fragments are reused); 4 = chunk epilogue
// without its arithmetic; 8 = no global stores of the chunk; 16 = no MFMAs
#ifndef SAIS_MLP_ABL
#define SAIS_MLP_ABL 0
#endif

struct MlpParams {
    RowParams r;                    // second GEMM + final epilogue: r.W = W2 [384, H] (ld r.ldw), r.K = H; r.A unused
    const bf16* X; int ldx;         // bf16 [M, 384]
    const bf16* Wa; int ldwa;       // bf16 [H, 384]
    const float* bias_a;            // [H] or null
    bf16* h; int ldh;               // chunk output [M, H] (null: not materialised, MLP_FWD only)
    bf16* g; int ldg;               // MLP_FWD_SAVE: out GELU'(u); MLP_BWD: in GELU'(u)
};

template <int MODE, int EPI, bool DP>
__global__ __launch_bounds__(256, 1) void mlp_fused_kernel(MlpParams q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const RowParams& p = q.r;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int m0 = blockIdx.x * p.rows_per_tile;
    const int mend = min(p.M, m0 + p.rows_per_tile);
    char* const sX = smem;
    char* const sH = smem + NXB * XBLK;
    const int nc = p.K / HC;

    // ---- the input tile: 84 LDS-DMA pieces (8 rows x 128 B), 21 per wave; swizzle on the source address
    {
        const int sub = lane >> 3, spos = lane & 7, schunk = spos ^ sub;
        for (int pc = w; pc < NXB * 14; pc += 4) {
            const int b = pc / 14, pr = pc - 14 * b;
            // tile rows past the workgroup's last row are COPIES of that row: their chunk results are then bit-identical to
            // it, and the chunk epilogue can store every lane's row unconditionally to min(row, last) — no divergent branch
            // between the MFMAs (a branch is a scheduling boundary: the VALU work of the epilogue would not interleave)
            const int m = min(m0 + 8 * pr + sub, mend - 1);
            glds16((const char*)q.X + ((size_t)m * q.ldx + 64 * b + schunk * 8) * 2, sX + b * XBLK + pr * 1024);
        }
    }

    // ---- weights in registers, MFMA "A" fragment layout: lane (li, g) holds W[row(t, li)][32 s + 8 g .. + 7].
    // Row permutation inside a 32-row slice: MFMA row i of tile t <- slice row 8 (i >> 2) + 4 t + (i & 3), so that lane
    // group g ends up with 8 CONTIGUOUS result columns 8 g .. 8 g + 7 (tile 0: + 0..3, tile 1: + 4..7).
    const int prow = 8 * (li >> 2) + (li & 3);
    bf16x8 w1[2][12], w2[6][4];
    // The two weight sets are reloaded PIECEWISE inside the other GEMM's k-steps, in the order the next phase consumes
    // them: a fragment's registers are dead after its k-step, so the incoming set takes over the outgoing set's registers
    // (a whole-set prefetch keeps 96 + 96 weight registers live beside the 224 accumulators and spills ~125 dwords).
    // addresses = wave-uniform base (SGPRs, moves with the chunk) + 32-bit per-lane byte offset (loop-invariant VGPR) +
    // immediate: 64-bit per-lane pointers for the 8 weight rows and 14 output rows would cost ~45 registers
    unsigned wo1[2], wo2[6], ro[RMT];
#pragma unroll
    for (int t = 0; t < 2; ++t) wo1[t] = ((unsigned)(32 * w + prow + 4 * t) * (unsigned)q.ldwa + 8 * g) * 2u;
#pragma unroll
    for (int t = 0; t < 6; ++t) wo2[t] = ((unsigned)(96 * w + 32 * (t >> 1) + prow + 4 * (t & 1)) * (unsigned)p.ldw + 8 * g) * 2u;
#pragma unroll
    for (int r = 0; r < RMT; ++r)                        // chunk outputs: row min(row, last) (duplicates carry identical values)
        ro[r] = ((unsigned)min(m0 + 16 * r + li, mend - 1) * (unsigned)q.ldh + 32 * w + 8 * g) * 2u;
    auto ld_w1 = [&](int c, int s0, int s1) {            // k-steps [s0, s1) of both column tiles of chunk c
        if ((SAIS_MLP_ABL & 1) && c > 1) return;
        const char* base = (const char*)q.Wa + (size_t)c * HC * q.ldwa * 2;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = s0; s < s1; ++s) w1[t][s] = *(const bf16x8*)(base + wo1[t] + 64 * s);
    };
    auto ld_w2 = [&](int c, int s0, int s1) {            // k-steps [s0, s1) of the six column tiles, hidden chunk c
        if ((SAIS_MLP_ABL & 1) && c > 0) return;
        const char* base = (const char*)p.W + (size_t)c * HC * 2;
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int s = s0; s < s1; ++s) w2[t][s] = *(const bf16x8*)(base + wo2[t] + 64 * s);
    };

    f32x4 acc[RMT][6], u[RMT][2];
#pragma unroll
    for (int i = 0; i < RMT; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    bf16x8 fa0[RMT];                                     // SAIS_MLP_ABL & 2 only
    if (SAIS_MLP_ABL & 2) {
#pragma unroll
        for (int r = 0; r < RMT; ++r) fa0[r] = *(const bf16x8*)(sX + swz(16 * r + li, g));
    }
    // A / chunk fragments of k32-step s from an LDS image (blocks of 64 k: 112 rows x 128 B, XOR swizzle)
    auto rd_frags = [&](const char* img, int s, bf16x8 (&fa)[RMT]) {
        const char* sb = img + (s >> 1) * XBLK;
#pragma unroll
        for (int r = 0; r < RMT; ++r) fa[r] = (SAIS_MLP_ABL & 2) ? fa0[r] : *(const bf16x8*)(sb + swz(16 * r + li, (s & 1) * 4 + g));
    };
    // first GEMM of one chunk: u[112, 32 (this wave)] = X[112, 384] . W1[chunk rows]^T.  One wave per SIMD: nothing hides an
    // LDS round trip, so the fragments of k-step s + 1 are read while the 14 MFMAs of step s issue (two register sets, the
    // interleave pinned with sched_group_barrier: one ds_read_b128 per two MFMAs).  LW2: the second GEMM's weights of hidden
    // chunk cw2 come in behind k-steps 2, 5, 8, 11 (one k-step of theirs each).  The chunk epilogue's own operands (bias /
    // GELU' rows of chunk cnext) are requested here, a whole phase before they are used.
    float ba[8];
    bf16x8 gin[RMT];
    auto chunk_loads = [&](int c) {
        if constexpr (MODE != MLP_BWD) {
            if (q.bias_a) {
                const float* b = q.bias_a + HC * c + 32 * w + 8 * g;
                const f32x4 b0 = *(const f32x4*)b, b1 = *(const f32x4*)(b + 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) { ba[i] = b0[i]; ba[4 + i] = b1[i]; }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) ba[i] = 0.f;
            }
        } else {
            const char* gb = (const char*)q.g + (size_t)c * HC * 2;
#pragma unroll
            for (int r = 0; r < RMT; ++r) gin[r] = *(const bf16x8*)(gb + ro[r]);
        }
    };
    auto gemm1 = [&](int cnext, int cw2, auto lw2c) {
        constexpr bool LW2 = decltype(lw2c)::value;
        chunk_loads(cnext);
#pragma unroll
        for (int i = 0; i < RMT; ++i) { u[i][0] = f32x4{0, 0, 0, 0}; u[i][1] = f32x4{0, 0, 0, 0}; }
        bf16x8 fa[2][RMT];
        rd_frags(sX, 0, fa[0]);
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            if (s + 1 < 12) rd_frags(sX, s + 1, fa[(s + 1) & 1]);
#pragma unroll
            for (int r = 0; r < RMT; ++r) {
                u[r][0] = mfma16(w1[0][s], fa[s & 1][r], u[r][0]);
                u[r][1] = mfma16(w1[1][s], fa[s & 1][r], u[r][1]);
            }
            if constexpr (LW2) {
                if (s % 3 == 2) ld_w2(cw2, s / 3, s / 3 + 1);
            }
        }
        if constexpr ((SAIS_MLP_ABL & 2) == 0) {
            __builtin_amdgcn_sched_group_barrier(0x100, RMT, 0);                 // step 0's fragments
#pragma unroll
            for (int i = 0; i < 11 * RMT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);               // 2 MFMAs of step s
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);               // 1 fragment of step s + 1
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * RMT, 0);
        }
    };
    // k32-step s of the second GEMM (single fragment set: 7 reads feed 42 MFMAs, and the register budget is spent)
    auto gemm2_step = [&](const char* hb, int s) {
        bf16x8 fa[RMT];
        rd_frags(hb, s, fa);
#pragma unroll
        for (int r = 0; r < RMT; ++r)
#pragma unroll
            for (int t = 0; t < 6; ++t) acc[r][t] = mfma16(w2[t][s], fa[r], acc[r][t]);
    };
    // chunk epilogue of row tile r: 8 values per lane = row 16 r + li, hidden columns HC c + 32 w + 8 g + 0..7
    auto chunk_epi = [&](int c, char* hb, int r) {
        float v[8], d[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = u[r][0][i]; v[4 + i] = u[r][1][i]; }
        bf16x8 hv, gv;
        if constexpr ((SAIS_MLP_ABL & 4) != 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { hv[i] = (bf16)v[i]; gv[i] = (bf16)v[i]; }
        } else if constexpr (MODE == MLP_BWD) {
#pragma unroll
            for (int i = 0; i < 8; ++i) hv[i] = (bf16)(v[i] * (float)gin[r][i]);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += ba[i];
            if constexpr (MODE == MLP_FWD_SAVE) {
                gelu_and_grad_n(v, d);
#pragma unroll
                for (int i = 0; i < 8; ++i) gv[i] = (bf16)d[i];
            } else {
                gelu_erf_n(v);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) hv[i] = (bf16)v[i];
        }
        *(bf16x8*)(hb + (w >> 1) * XBLK + swz(16 * r + li, 4 * (w & 1) + g)) = hv;
        char* const hob = (char*)q.h + (size_t)c * HC * 2;
        if constexpr ((SAIS_MLP_ABL & 8) != 0) {
        } else if constexpr (MODE == MLP_FWD) {
            if (q.h) *(bf16x8*)(hob + ro[r]) = hv;                           // wave-uniform, loop-invariant
        } else {
            *(bf16x8*)(hob + ro[r]) = hv;
            if constexpr (MODE == MLP_FWD_SAVE) *(bf16x8*)((char*)q.g + (size_t)c * HC * 2 + ro[r]) = gv;
        }
    };
    // chunk epilogue of chunk c, issued together with the second GEMM of chunk c - 1 (FC2) and the piecewise reload of the
    // first GEMM's weights for chunk c + 1 (LW1: three of their k-steps behind each k-step of the second GEMM)
    auto phase_e = [&](int c, auto fc2c, auto lw1c) {
        constexpr bool FC2 = decltype(fc2c)::value, LW1 = decltype(lw1c)::value;
        const char* hprev = sH + ((c + 1) & 1) * 2 * XBLK;
        char* hcur = sH + (c & 1) * 2 * XBLK;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if constexpr (FC2) gemm2_step(hprev, s);
            if constexpr (LW1) ld_w1(c + 1, 3 * s, 3 * s + 3);
            chunk_epi(c, hcur, 2 * s);
            if (2 * s + 1 < RMT) chunk_epi(c, hcur, 2 * s + 1);
        }
    };
    using T = std::true_type;
    using F = std::false_type;

    ld_w1(0, 0, 12);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    gemm1(0, 0, F{});                                    // chunk 0
    if (nc > 1) ld_w1(1, 0, 12);
    phase_e(0, F{}, F{});
    __syncthreads();
    if (nc > 1) gemm1(1, 0, T{});                        // chunk 1, with W2(chunk 0) coming in
    else ld_w2(0, 0, 4);
    for (int c = 1; c + 1 < nc; ++c) {
        phase_e(c, T{}, T{});                            // epilogue(c) || gemm2(c - 1) || W1(c + 1) coming in
        __syncthreads();                                 // chunk c is complete in LDS; chunk c - 1's image is free
        gemm1(c + 1, c, T{});                            // chunk c + 1 || W2(c) coming in
    }
    if (nc > 1) {
        phase_e(nc - 1, T{}, F{});
        __syncthreads();
        ld_w2(nc - 1, 0, 4);
    }
    {
        const char* hlast = sH + ((nc - 1) & 1) * 2 * XBLK;
#pragma unroll
        for (int s = 0; s < 4; ++s) gemm2_step(hlast, s);
    }
    __syncthreads();                                     // every wave is done with the LDS images: the slabs may overwrite them
    row_epilogue<EPI, DP, 4, 4>(p, acc, smem, m0, mend, 0);
}

int mlp_rows_per_tile(int M) {
    // one round of 2 tiles per CU when M allows it (M = 50 432 -> 99 rows, 510 tiles), whole 112-row tiles for small M
    const int rounds = (M + 512 * 112 - 1) / (512 * 112);
    int rows = (M + 512 * rounds - 1) / (512 * rounds);
    if (rows < 64) rows = 112;
    return rows;
}

template <int MODE, int EPI, bool DP>
int launch_mlp(MlpParams& q, void* stream) {
    static thread_local bool set = false;
    if (!set) {
        if (hipFuncSetAttribute((const void*)mlp_fused_kernel<MODE, EPI, DP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                MLP_LDS) != hipSuccess)
            return SAIS_ERR_LAUNCH;
        set = true;
    }
    q.r.rows_per_tile = mlp_rows_per_tile(q.r.M);
    const int grid = (q.r.M + q.r.rows_per_tile - 1) / q.r.rows_per_tile;
    hipLaunchKernelGGL((mlp_fused_kernel<MODE, EPI, DP>), dim3(grid), dim3(256), MLP_LDS, (hipStream_t)stream, q);
    return sais_check_launch();
}

int fill(const SaisMlp* a, MlpParams& q) {
    if (!a || !a->X || !a->W1 || !a->W2 || a->M <= 0 || a->H <= 0 || a->H % HC) return SAIS_ERR_ARG;
    if (a->ldx % 8 || a->ldw1 % 8 || a->ldw2 % 8 || a->ldh % 8 || a->ldg % 8) return SAIS_ERR_ARG;
    const SaisGemmLn& t = a->tail;
    if (!t.resid || t.ldr % 4 || t.ldo32 % 4 || t.ldo16 % 8) return SAIS_ERR_ARG;
    // 32-bit byte offsets inside the kernel; the chunk outputs h and g share one row-offset table
    if ((double)a->M * a->ldx * 2.0 >= 4294967296.0 || (double)a->M * a->ldh * 2.0 >= 4294967296.0 ||
        (double)a->H * a->ldw1 * 2.0 >= 4294967296.0 || (double)RBN * a->ldw2 * 2.0 >= 4294967296.0)
        return SAIS_ERR_ARG;
    if (a->g && a->ldg != a->ldh) return SAIS_ERR_ARG;
    q.X = (const bf16*)a->X; q.ldx = a->ldx; q.Wa = (const bf16*)a->W1; q.ldwa = a->ldw1; q.bias_a = a->bias1;
    q.h = (bf16*)a->h; q.ldh = a->ldh; q.g = (bf16*)a->g; q.ldg = a->ldg;
    RowParams& p = q.r;
    p.W = (const bf16*)a->W2; p.ldw = a->ldw2; p.M = a->M; p.N = RBN; p.K = a->H;
    p.out = t.out32; p.ldo = t.ldo32; p.out2 = t.out16; p.ldo2 = t.ldo16; p.aux = t.resid; p.ldaux = t.ldr;
    p.gamma = t.gamma; p.beta = t.beta; p.eps = t.eps; p.mean = t.mean; p.rstd = t.rstd;
    return SAIS_OK;
}

}  // namespace

extern "C" int sais_mlp_fwd(const SaisMlp* a, void* stream) {
    SAIS_ENTER();
    MlpParams q{};
    if (fill(a, q) != SAIS_OK) return SAIS_ERR_ARG;
    const SaisGemmLn& t = a->tail;
    if (!t.out32 || (a->g && !a->h)) return SAIS_ERR_ARG;
    q.r.bias = t.bias;
    q.r.rowscale = t.rowscale;
    const bool save = a->g != nullptr, ln = t.gamma != nullptr, dp = t.rowscale != nullptr;
    if (ln && (!t.beta || !t.out16)) return SAIS_ERR_ARG;
    if (!ln) q.r.out2 = nullptr;
#define MLP_GO(MODE)                                                                                               \
    return ln ? (dp ? launch_mlp<MODE, ROW_LN_FWD, true>(q, stream) : launch_mlp<MODE, ROW_LN_FWD, false>(q, stream)) \
              : (dp ? launch_mlp<MODE, ROW_RESID_F32, true>(q, stream) : launch_mlp<MODE, ROW_RESID_F32, false>(q, stream));
    if (save) { MLP_GO(MLP_FWD_SAVE) }
    MLP_GO(MLP_FWD)
#undef MLP_GO
}

extern "C" int sais_mlp_bwd(const SaisMlp* a, void* stream) {
    SAIS_ENTER();
    MlpParams q{};
    if (fill(a, q) != SAIS_OK) return SAIS_ERR_ARG;
    const SaisGemmLn& t = a->tail;
    if (!a->g || !a->h || !t.gamma || !t.mean || !t.rstd || (!t.out32 && !t.out16)) return SAIS_ERR_ARG;
    if ((t.dgamma == nullptr) != (t.dbeta == nullptr) || t.lddres % 4) return SAIS_ERR_ARG;
    RowParams& p = q.r;
    p.dres = t.dres; p.lddres = t.lddres; p.dres_period = t.dres_period; p.dgamma = t.dgamma; p.dbeta = t.dbeta;
    p.rowscale = t.rowscale16;
    return t.rowscale16 ? launch_mlp<MLP_BWD, ROW_LN_BWD, true>(q, stream) : launch_mlp<MLP_BWD, ROW_LN_BWD, false>(q, stream);
}
