// NOT PART OF libsais_hip.so — kept as the record of a round-2 experiment (correct: it passed test_gemm_nt_epilogues
// when wired into sais_gemm_nt; measured inside the training step on MI355X: fc1+GELU 142.5 vs 145.8 us, dX fc2 121.0 vs
// 114.4 us, qkv 72.4 vs 69.7 us against the 128x128 persistent kernel, i.e. no gain: these GEMMs are not bound by the
// LDS fill stream).  Include path when rebuilding: -I sais_amd/csrc.
//
// Weight-stationary bf16 MFMA GEMM for the short-K, wide-N GEMMs of a ViT block (gfx950):
//     C[M,N] = A[M,K] . W[N,K]^T + epilogue,   K <= 384, N >= 1152   (qkv; fc1 + GELU / GELU'; dX of fc2 x GELU')
//     — Attention.qkv / Mlp.fc1 of dino-main/vision_transformer.py:59-65,80-92 and the dX of Mlp.fc2.
//
// With K = D = 384 a 128 x 128 output tile needs 2 x 96 KiB of operands for 12.6 MFLOP: the tiled kernels of gemm.hip
// spend their time re-filling LDS (W is re-streamed for every 128-row tile: 0.93 GB of LDS-DMA per launch at M = 50 432,
// N = 1536) and re-starting a six-step pipeline per tile.  Here a persistent workgroup OWNS a 128-column panel of W
// (128 x 384 bf16 = 96 KiB, loaded once, resident in LDS for the whole launch) and streams A row tiles through a
// four-slot ring of 128 x 64 k pieces (64 KiB) that keeps running across tile boundaries: only A enters LDS
// (M x K x 2 B per panel: 465 MB per launch at N = 1536, half the bytes), there is no per-tile prologue, and the next
// tile's first three K-steps are already in flight while the epilogue of the current tile stores.
//
// 160 KiB of LDS -> one 512-thread workgroup per CU: 8 waves as 2 x 4 (64 x 32 per wave: the tile, LDS image, weight-row
// permutation and 8-columns-per-lane epilogue of gemm.hip's eight-wave kernel).  Grid = (N/128) panels x wpp workgroups
// per panel (252 for N = 1536 and 1152); the workgroups of different panels walk the row tiles in the same order, so a
// row tile is fetched from HBM once and re-read from L2 / MALL by the other panels.
//
// Counted waits (vmcnt is in-order and counts stores): A(s+3) is issued at step s; at the end of step s the wait leaves
// exactly the younger operations outstanding — the two later A pieces, the epilogue stores of the previous tile while
// they are younger than the piece needed (its first two steps), and the multiplier loads of the MUL epilogue issued
// two steps before the end of a tile.  The bias is loaded once per workgroup (a panel's columns never change).
#include "gemm_nt_epi.hpp"

namespace {

constexpr int WT = 128 * 64 * 2;            // 16 KiB: 128 rows x 64 k bf16
constexpr int WS_RING = 4;

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_wstat_kernel(NtParams p, int wpp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // W panel: nk x 16 KiB | A ring: 4 x 16 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3, g = lane >> 4, li = lane & 15;
    const int nk = p.K / 64;
    const int npanel = p.N / 128;
    const int panel = blockIdx.x % npanel, r = blockIdx.x / npanel;
    const int n0 = panel * 128;
    const int ntm = (p.M + 127) / 128;
    if (r >= ntm) return;
    const int ntiles = (ntm - r + wpp - 1) / wpp;                    // row tiles r, r + wpp, ...
    const int S = ntiles * nk;                                       // K-steps this workgroup executes
    char* const sW = smem;
    char* const sA = smem + nk * WT;

    // staging: wave w issues pieces 2w, 2w+1 (8 LDS rows each) of every 16-KiB piece
    const int sub = lane >> 3, spos = lane & 7, schunk = spos ^ sub;
    {
        const bf16* bsrc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rr = 8 * (2 * wid + j) + sub;
            bsrc[j] = p.B + (size_t)(n0 + perm_row32(rr)) * p.ldb + schunk * 8;
        }
        for (int kt = 0; kt < nk; ++kt)
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(bsrc[j] + kt * 64, sW + kt * WT + (2 * wid + j) * 1024);
    }
    // A issue cursor (step s_iss = tile i_iss, k-step kt_iss)
    int kt_iss = 0, t_iss = r, s_iss = 0;
    unsigned aoff[2];
    auto set_rows = [&](int t) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int m = t * 128 + 8 * (2 * wid + j) + sub;
            m = m < p.M ? m : p.M - 1;                               // clamp: rows >= M are never stored
            aoff[j] = ((unsigned)m * (unsigned)p.lda + schunk * 8) * 2u;
        }
    };
    set_rows(t_iss);
    const char* const Ab = (const char*)p.A;
    auto issue_next = [&]() {                                        // A(s_iss) -> ring slot s_iss & 3
        char* s = sA + (s_iss & (WS_RING - 1)) * WT + (2 * wid) * 1024;
        const char* b = Ab + (size_t)kt_iss * 128;
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(b + aoff[j], s + j * 1024);
        ++s_iss;
        if (++kt_iss == nk) { kt_iss = 0; t_iss += wpp; set_rows(t_iss); }
    };
    issue_next();
    if (S > 1) issue_next();
    if (S > 2) issue_next();
    // bias of this panel's columns: once per workgroup
    float bias[8];
    {
        const int n = n0 + wc * 32 + 8 * g;
        if (p.bias) {
            const f32x4 t0 = *(const f32x4*)(p.bias + n), t1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { bias[i] = t0[i]; bias[4 + i] = t1[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) bias[i] = 0.f;
        }
    }
    // W panel, A(0) and the bias have landed; A(1), A(2) may fly
    if (S > 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    constexpr int NST = EPI == SAIS_EPI_BIAS_GELU_GRAD_BF16 ? 8 : 4;      // store instructions per wave and full tile
    int s = 0;
    for (int i = 0; i < ntiles; ++i) {
        const int m0 = (r + i * wpp) * 128;
        f32x4 acc[4][2];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
        EpiAux8 aux;
        for (int kt = 0; kt < nk; ++kt, ++s) {
            const bool steady = s + 3 < S;
            if constexpr (EPI == SAIS_EPI_MUL_BF16) {
                // the multiplier rows of this tile (4 loads per lane), two steps before the epilogue and BEFORE this step's
                // A piece: by the time the epilogue needs them only the two youngest A pieces are younger
                if (kt == nk - 2) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        int m = m0 + wr * 64 + mt * 16 + li;
                        m = m < p.M ? m : p.M - 1;
                        aux.u[mt] = *(const bf16x8*)((const bf16*)p.aux + (size_t)m * p.ldaux + n0 + wc * 32 + 8 * g);
                    }
                }
            }
            if (steady) issue_next();                                // A(s+3)
            const char* sa = sA + (s & (WS_RING - 1)) * WT;
            const char* sb = sW + kt * WT;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 fa[4], fb[2];
#pragma unroll
                for (int t = 0; t < 4; ++t) fa[t] = *(const bf16x8*)(sa + swz(wr * 64 + t * 16 + li, ks * 4 + g));
#pragma unroll
                for (int t = 0; t < 2; ++t) fb[t] = *(const bf16x8*)(sb + swz(wc * 32 + t * 16 + li, ks * 4 + g));
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma16(fb[nt], fa[mt], acc[mt][nt]);
            }
            // A(s+1) must have landed; what is younger than it may stay in flight
            if (!steady) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            } else if (i > 0 && kt < 2) {                            // + the previous tile's stores
                if constexpr (NST == 8) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            } else if (EPI == SAIS_EPI_MUL_BF16 && kt >= nk - 2) {   // + this tile's multiplier loads
                asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = m0 + wr * 64 + mt * 16 + li;
            if (m >= p.M) continue;
            float vv[8];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) vv[4 * nt + e] = acc[mt][nt][e];
            epilogue8<EPI>(p, m, n0 + wc * 32 + 8 * g, vv, bias, aux, mt);
        }
    }
}

template <int EPI>
int launch_wstat(const NtParams& p, void* stream) {
    const int nk = p.K / 64, lds = (nk + WS_RING) * WT;
    static thread_local int granted = 0;
    if (lds > granted) {
        if (hipFuncSetAttribute((const void*)gemm_nt_wstat_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
            hipSuccess)
            return SAIS_ERR_LAUNCH;
        granted = lds;
    }
    const int npanel = p.N / 128;
    const int wpp = 256 / npanel;                                    // one workgroup per CU
    hipLaunchKernelGGL(gemm_nt_wstat_kernel<EPI>, dim3(npanel * wpp), dim3(512), lds, (hipStream_t)stream, p, wpp);
    return sais_check_launch();
}

}  // namespace

// called by sais_gemm_nt (gemm.hip) for M >= 8192, K in {256, 320, 384}, N >= 1152 (N % 128 == 0, at most 256 panels)
extern "C" int sais_gemm_nt_wstat_(const SaisGemm* g, void* stream) {
    const int nk = g->K / 64;
    if (g->K % 64 || nk < 4 || nk > 6 || g->N % 128 || g->N / 128 > 256) return SAIS_ERR_ARG;
    if ((double)g->M * g->lda * 2.0 >= 4294967296.0) return SAIS_ERR_ARG;      // 32-bit byte offsets into A
    NtParams p{(const bf16*)g->A, (const bf16*)g->B, g->lda, g->ldb, g->M, g->N, g->K, g->bias,
               g->out, g->ldo, g->out2, g->ldo2, g->aux, g->ldaux, 0, 0, 0};
    switch (g->epilogue) {
        case SAIS_EPI_BIAS_BF16: return launch_wstat<SAIS_EPI_BIAS_BF16>(p, stream);
        case SAIS_EPI_BIAS_GELU_GRAD_BF16: return launch_wstat<SAIS_EPI_BIAS_GELU_GRAD_BF16>(p, stream);
        case SAIS_EPI_MUL_BF16: return launch_wstat<SAIS_EPI_MUL_BF16>(p, stream);
        default: return SAIS_ERR_ARG;
    }
}
