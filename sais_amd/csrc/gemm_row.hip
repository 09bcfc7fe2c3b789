// Row-owning bf16 MFMA GEMM for the residual stream of the ViT blocks (gfx950):
//     C[M, 384] = A[M,K] . W[384,K]^T + epilogue,   one (<= 112)-row x 384-column output tile per 256-thread workgroup.
// With N = D = 384 a workgroup owns WHOLE rows of the output, which lets LayerNorm live in the epilogue:
//
//   LN_FWD  (proj / fc2 of Block.forward, vision_transformer.py:107-113):
//           x_out = A.W^T + bias + x ;  xn = LayerNorm(x_out) (eps 1e-6, the NEXT norm of the residual stream) ;
//           mean / rstd saved for backward.  Replaces gemm_nt<resid_f32> + ln_fwd_kernel.
//   LN_BWD  (dX of fc1 / qkv followed by autograd of norm2 / norm1):
//           dy = A.W^T ; dx = dres + rstd (dy g - mean(dy g) - xhat mean(dy g xhat)) ; dgamma += sum dy xhat ;
//           dbeta += sum dy.  Replaces gemm_nt<bias_bf16> + ln_bwd_kernel (the bf16 dy round trip disappears).
//   BIAS_BF16 / RESID_F32: the two plain N = 384 GEMMs of a block (dX of proj; the last block's fc2), same tile and
//           streaming epilogue.  (N = 384 g works too — column groups — but for N = 1152 / 1536 the 128x128 persistent
//           kernel of gemm.hip measured faster inside the step: 116 vs 134 us for dX fc2, 69 vs 72 us for qkv.)
//
// Tile.  M = 50 432 rows over 256 CUs x 2 workgroups is 98.5 rows per workgroup: the host picks rows_per_tile =
// ceil(M / 512) (99 -> 510 equal tiles, one round, two workgroups on every CU) and the kernel computes 7 MFMA row
// tiles (112 rows) of which the last 13 are not stored.  4 waves side by side in N: a wave owns all 112 rows x 96
// columns = 7 x 6 MFMA 16x16x32 tiles = 168 fp32 accumulator VGPRs (+ 36 fragment registers: fits 256 with headroom,
// two waves per SIMD = TWO workgroups per CU, so one workgroup's HBM-bound epilogue runs under the other's K loop).
//
// K loop.  LDS (80 KiB): A tile (128 rows x 64 k, 16 KiB) in two slots; W streamed as 16-KiB chunks through a
// three-slot ring.  Chunk c of a K-step holds, for EACH wave, 32 of its 96 columns (LDS rows 32 w .. 32 w + 31 =
// columns 96 w + 32 c ..): every sub-step all four waves do 7 x 2 x 2 = 28 MFMAs on 18 ds_read_b128.  Per K-step
// 16 KiB of A + 48 KiB of W enter LDS for 2*112*384*64 flop (86 flop per LDS-DMA byte; the 128x128 kernel: 65).  All
// staging is LDS-DMA (global_load_lds_dwordx4), XOR swizzle and weight-row permutation on the SOURCE address; waits are
// counted (vmcnt is in-order): W runs two sub-steps ahead, A (first-touch HBM data) three.
//
// Epilogue.  Operands are swapped in the MFMA (weights as "A"), so a lane owns one output row per row tile and 8
// contiguous columns per chunk.  For the LayerNorm epilogues the tile is handed, 32 rows at a time, through LDS (fp32
// slab, rows padded to 1552 B) to a streaming phase in which the 8 half-waves of the workgroup treat whole rows exactly
// like the stand-alone LayerNorm kernels (norm.hip): 12 columns per lane, statistics by half-wave shuffles, 512-B
// coalesced row segments, the next row's operands prefetched BEFORE the current row's stores are issued (vmcnt counts
// stores).  LN_BWD reads x once; dgamma / dbeta are per-lane column sums reduced through LDS, one atomic per column
// and workgroup.
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"
#include "../../include/sais_hip.h"
#include "gemm_row_epi.hpp"

namespace {

constexpr int RTILE = 128 * 64 * 2;                 // 16 KiB: 128 rows x 64 k bf16
// NW = waves per workgroup.  4: one 112-row half, two workgroups per CU (small / odd M).  8: TWO 112-row halves that share
// every W chunk (waves 0-3 rows 0..111, waves 4-7 rows 112..223), ONE workgroup per CU: the W fill stream per flop halves
// (145 flop per LDS-DMA byte instead of 84), and M = 50 432 = 256 x 197 gives every CU exactly one frame's rows.
// SPEC (eight waves): staging roles are split — waves 0-3 issue every W chunk, waves 4-7 every A tile, into a THREE-slot A
// ring, so that the A stream (first-touch HBM rows) runs TWO K-steps ahead.  vmcnt is one in-order counter per wave: a
// wave that issues both operands has to retire its A loads whenever it waits for the (younger, sooner-needed) W chunk
// behind them, which caps the A lead at one K-step = 28-32 KB in flight per CU; measured (r3b): with that cap the K = 1536
// kernels take the same time at 84 and at 145 flop per LDS-DMA byte, with and without the stagger — they are paced by the
// latency x bytes-in-flight of the A stream, not by the fill rate.
template <int NW, bool SPEC = false> constexpr int row_lds() { return (SPEC ? 3 : 2) * (NW / 4) * RTILE + 3 * RTILE; }

template <int N> DEVINL void wait_vm_lgkm0() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(N) : "memory"); }
template <int N> DEVINL void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// weight-row permutation inside a 32-column wave slice of a chunk: LDS row 16 t + 4 a + b <- slice column 8 a + 4 t + b,
// so that lane group a = lane>>4 ends up with 8 CONTIGUOUS output columns (t = MFMA tile 0/1, b = accumulator register)
DEVINL int perm32(int r) { return (((r >> 2) & 3) << 3) | (((r >> 4) & 1) << 2) | (r & 3); }

template <int EPI, bool DP, int NW, bool STAG = false, bool SPEC = false>
__global__ __launch_bounds__(64 * NW, 2) void gemm_nt_row_kernel(RowParams p) {
    static_assert(!STAG || NW == 8, "the stagger pairs waves w and w + 4 of one SIMD");
    static_assert(!SPEC || NW == 8, "role-split staging needs the two wave groups");
    constexpr int ASLOTS = SPEC ? 3 : 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CLK_STAMP(4 + (EPI == 2 ? 0 : EPI == 3 ? 1 : 2) * 2 + (p.K > 1200 ? 1 : 0));    // LN_FWD 4/5, LN_BWD 6/7, plain 8/9 (K <= / > 1200)
    constexpr int HALVES = NW / 4;                   // 112-row halves of the tile
    constexpr int ATILE = HALVES * RTILE;            // one A slot: 128 LDS rows per half
    constexpr int WP = 16 / NW;                      // LDS-DMA pieces (8 rows) of a W chunk per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wid >> 2, wq = wid & 3;         // row half, column quarter (96 columns)
    const int g = lane >> 4, li = lane & 15;
    // tiles are numbered column-group fastest, so the N/384 workgroups that share an A row panel run side by side on
    // one XCD (xcd_remap hands every XCD a contiguous run of tiles) and the panel is fetched from HBM once
    const int ngrp = p.N / RBN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (tile % ngrp) * RBN;
    const int m0 = (tile / ngrp) * p.rows_per_tile;
    const int mend = min(p.M, m0 + p.rows_per_tile);                 // rows [m0, mend) are this workgroup's

    // staging: wave w issues pieces 4w..4w+3 (8 LDS rows each) of the A slot (LDS row 128 h + r' <- tile row 112 h + r')
    // and pieces WP w .. WP w + WP - 1 of every W chunk.  W chunk c: LDS rows 32 q + j (q = owning column quarter) <- weight
    // rows 96 q + 32 c + perm32(j)
    const int sub = lane >> 3, spos = lane & 7, schunk = spos ^ sub;
    // per-lane byte offsets from the uniform bases.  SPEC: one array, A offsets (8 pieces) on waves 4-7, W offsets (4
    // pieces of every chunk) on waves 0-3
    constexpr int NA = SPEC ? 8 : 4, NWP = SPEC ? 4 : WP;
    unsigned aoff[NA], woff_[SPEC ? 1 : WP];
    unsigned (&woff)[SPEC ? NA : WP] = *(unsigned (*)[SPEC ? NA : WP])(SPEC ? aoff : woff_);
    if (!SPEC || half) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int r = 8 * (NA * (SPEC ? wq : wid) + j) + sub;    // LDS row of the slot
            const int rt = 112 * (r >> 7) + (r & 127);               // tile row (LDS rows 112..127 of a half are never read)
            int m = ((r & 127) < 112 && rt < p.rows_per_tile) ? m0 + rt : m0;   // rows past the tile: one L2-hot row
            m = m < p.M ? m : p.M - 1;                               // clamp: rows outside the tile are never stored
            aoff[j] = ((unsigned)m * (unsigned)p.lda + schunk * 8) * 2u;
        }
    }
    if (!SPEC || !half) {
#pragma unroll
        for (int j = 0; j < NWP; ++j) {
            const int r = 8 * (NWP * (SPEC ? wq : wid) + j) + sub;
            woff[j] = ((unsigned)(n0 + 96 * (r >> 5) + perm32(r & 31)) * (unsigned)p.ldw + schunk * 8) * 2u;
        }
    }
    char* const sA = smem;
    char* const sW = smem + ASLOTS * ATILE;
    const char* const Ab = (const char*)p.A;
    const char* const Wb = (const char*)p.W;
    const size_t wchunk = (size_t)32 * p.ldw * 2;                    // bytes between chunk c and c + 1 of a wave's columns
    // pieces [J0, J1) of this wave's share of A(kt) into ring slot `slot`
    auto issue_a_part = [&](int kt, int slot, auto j0c, auto j1c) {
        constexpr int J0 = decltype(j0c)::value, J1 = decltype(j1c)::value;
        char* s = sA + slot * ATILE + (NA * (SPEC ? wq : wid)) * 1024;
        const char* b = Ab + (size_t)kt * (RBK * 2);
#pragma unroll
        for (int j = J0; j < J1; ++j) glds16(b + aoff[j], s + j * 1024);
    };
    auto issue_a = [&](int kt) { issue_a_part(kt, SPEC ? kt % 3 : (kt & 1), std::integral_constant<int, 0>{}, std::integral_constant<int, NA>{}); };
    auto issue_w = [&](int kt, int c) {
        char* s = sW + c * RTILE + (NWP * (SPEC ? wq : wid)) * 1024;
        const char* b = Wb + c * wchunk + (size_t)kt * (RBK * 2);
#pragma unroll
        for (int j = 0; j < NWP; ++j) glds16(b + woff[j], s + j * 1024);
    };

    f32x4 acc[RMT][6];
#pragma unroll
    for (int i = 0; i < RMT; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    // fragments of one 32-deep k-half of chunk C: 2 W + 7 A ds_read_b128; then 14 MFMAs
    bf16x8 fa[RMT], fb[2];
#define ROW_READ(C, KS)                                                                         \
    {                                                                                           \
        const char* sa = sA + aslot * ATILE + half * RTILE;                                     \
        const char* sb = sW + (C) * RTILE;                                                      \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                           \
            fb[t] = *(const bf16x8*)(sb + swz(32 * wq + 16 * t + li, (KS) * 4 + g));            \
        _Pragma("unroll") for (int t = 0; t < RMT; ++t)                                         \
            fa[t] = *(const bf16x8*)(sa + swz(16 * t + li, (KS) * 4 + g));                      \
    }
#define ROW_MMA(C)                                                                              \
    _Pragma("unroll") for (int mt = 0; mt < RMT; ++mt)                                          \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                           \
            acc[mt][2 * (C) + t] = mfma16(fb[t], fa[mt], acc[mt][2 * (C) + t]);
    // STAG (eight waves): waves 4-7, the SIMD partners of waves 0-3, run HALF A PHASE behind inside every barrier interval —
    // they open the interval with the 14 MFMAs of the previous chunk's second k-half (its fragments stay in registers across
    // the barrier, so the LDS slot may be refilled) while their partners read fragments, and alternate from there: one wave
    // of each SIMD is in the matrix pipe while the other is in the LDS (MI355X_MICROARCH.md, two waves per SIMD, item 9).
#define ROW_WAIT(MORE, NMORE)                                                                   \
    if (more) wait_vm_lgkm0<(MORE)>();                                                          \
    else wait_vm_lgkm0<(NMORE)>();                                                              \
    __builtin_amdgcn_s_barrier();
    // LATE = true (STAG, waves 4-7): a barrier interval opens with the 14 MFMAs of the PREVIOUS chunk's second k-half
#define ROW_COMPUTE(C, CPREV)                                                                   \
    if constexpr (LATE) {                                                                       \
        ROW_MMA(CPREV)                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        ROW_READ(C, 0)                                                                          \
        ROW_MMA(C)                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        ROW_READ(C, 1)                                                                          \
    } else {                                                                                    \
        ROW_READ(C, 0)                                                                          \
        ROW_MMA(C)                                                                              \
        ROW_READ(C, 1)                                                                          \
        ROW_MMA(C)                                                                              \
    }

    const int nk = p.K / RBK;
    if constexpr (SPEC) {
        if (half) {                                                  // A waves: two tiles ahead from the start
            issue_a(0);
            if (nk > 1) { issue_a(1); wait_vm<8>(); } else wait_vm<0>();
        } else {
            issue_w(0, 0);
            issue_w(0, 1);
            wait_vm<4>();                                            // W(0,0) landed; W(0,1) may fly
        }
    } else {
        issue_a(0);
        issue_w(0, 0);
        issue_w(0, 1);
        wait_vm<WP>();                                               // A(0), W(0,0) landed; W(0,1) may fly
    }
    __builtin_amdgcn_s_barrier();
    // STAG (eight waves): waves 4-7, the SIMD partners of waves 0-3, run HALF A PHASE behind inside every barrier interval:
    // the fragments of a chunk's second k-half stay in registers across the barrier (so the LDS slot may be refilled) and
    // their 14 MFMAs open the next interval while the partner wave reads its fragments, and the two alternate from there —
    // one wave of each SIMD in the matrix pipe, the other in the LDS (MI355X_MICROARCH.md, two waves per SIMD, item 9).
    // The two wave groups run two complete copies of the loop (same barrier count), chosen once per wave.
    auto kloop = [&](auto grp_c) {
        constexpr bool GRP1 = decltype(grp_c)::value;                // waves 4-7
        constexpr bool LATE = STAG && GRP1;
        using I = std::integral_constant<int, 0>;
        if constexpr (LATE) {                                        // the first "previous k-half" adds zeros
#pragma unroll
            for (int t = 0; t < RMT; ++t) fa[t] = zero8();
            fb[0] = zero8(); fb[1] = zero8();
        }
        int aslot = 0, anext = SPEC ? 2 : 1;                         // ring slots of A(kt) and of the tile issued in step kt
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + 1 < nk;
            if constexpr (!SPEC) {
                // sub-step 0: needs A(kt), W(kt,0).  Issue W(kt,2) then A(kt+1) (A last: it may stay in flight longest)
                issue_w(kt, 2);
                if (more) issue_a(kt + 1);
                ROW_COMPUTE(0, 2)
                ROW_WAIT(WP + 4, WP)                                 // W(kt,1) landed; W(kt,2) [+ A(kt+1)] in flight
                if (more) issue_w(kt + 1, 0);
                ROW_COMPUTE(1, 0)
                ROW_WAIT(4 + WP, 0)                                  // W(kt,2) landed; A(kt+1), W(kt+1,0) in flight
                if (more) issue_w(kt + 1, 1);
                ROW_COMPUTE(2, 1)
                ROW_WAIT(WP, 0)                                      // A(kt+1), W(kt+1,0) landed; W(kt+1,1) in flight
                aslot ^= 1;
            } else if constexpr (GRP1) {
                // A waves: A(kt+2) goes out in three parts (3 + 3 + 2 pieces) into the slot A(kt-1) left; the only wait is
                // at the end of the K-step: A(kt+1) landed, the 8 pieces of A(kt+2) may fly
                const bool more2 = kt + 2 < nk;
                (void)I{};
                if (more2) issue_a_part(kt + 2, anext, std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
                ROW_COMPUTE(0, 2)
                wait_vm_lgkm0<63>();                                 // (vmcnt 63 = no wait on loads)
                __builtin_amdgcn_s_barrier();
                if (more2) issue_a_part(kt + 2, anext, std::integral_constant<int, 3>{}, std::integral_constant<int, 6>{});
                ROW_COMPUTE(1, 0)
                wait_vm_lgkm0<63>();
                __builtin_amdgcn_s_barrier();
                if (more2) issue_a_part(kt + 2, anext, std::integral_constant<int, 6>{}, std::integral_constant<int, 8>{});
                ROW_COMPUTE(2, 1)
                if (more2) wait_vm_lgkm0<8>(); else wait_vm_lgkm0<0>();
                __builtin_amdgcn_s_barrier();
                aslot = aslot == 2 ? 0 : aslot + 1;
                anext = anext == 2 ? 0 : anext + 1;
            } else {
                // W waves: chunk q + 2 goes out in sub-step q; at its end chunk q + 1 has landed, the one just issued may fly
                issue_w(kt, 2);
                ROW_COMPUTE(0, 2)
                wait_vm_lgkm0<4>();
                __builtin_amdgcn_s_barrier();
                if (more) issue_w(kt + 1, 0);
                ROW_COMPUTE(1, 0)
                ROW_WAIT(4, 0)
                if (more) issue_w(kt + 1, 1);
                ROW_COMPUTE(2, 1)
                ROW_WAIT(4, 0)
                aslot = aslot == 2 ? 0 : aslot + 1;
            }
        }
        if constexpr (LATE) { ROW_MMA(2) }                           // the last chunk's second k-half
    };
    if ((STAG || SPEC) && half) kloop(std::true_type{});
    else kloop(std::false_type{});
#undef ROW_READ
#undef ROW_MMA
#undef ROW_COMPUTE
#undef ROW_WAIT

    // ------------------------------------------------------------------------------------------- epilogues
    row_epilogue<EPI, DP, NW>(p, acc, smem, m0, mend, n0);
}

// rows per workgroup.  NW = 4: one round of 2 workgroups per CU when M allows it (M = 50 432 -> 99 rows, 510 tiles), whole
// 112-row tiles for small M, r rounds of 512 tiles for M beyond 57 344.  NW = 8: one workgroup per CU, 224 rows at most
// (M = 50 432 -> 197 rows = one frame, 256 tiles).
template <int NW>
int rows_per_tile(int M) {
    constexpr int cap = 112 * (NW / 4), slots = NW == 4 ? 512 : 256;
    const int rounds = (M + slots * cap - 1) / (slots * cap);
    int rows = (M + slots * rounds - 1) / (slots * rounds);
    // few rows per workgroup waste MFMA row tiles (all RMT are always computed); many-row tiles leave workgroup slots
    // empty.  SAIS_ROW_MINROWS overrides the switch-over for A/B measurements.
    static const int minrows = [] { const char* e = getenv("SAIS_ROW_MINROWS"); return e ? atoi(e) : 0; }();
    if (rows < (minrows > 0 ? minrows : cap / 2 + 8)) rows = cap;
    return rows;
}

// the eight-wave tile pays when a launch fills the chip with whole 1-workgroup-per-CU rounds: the ViT GEMMs of a training
// step and of large extraction batches.  SAIS_ROW_WAVES=4 / 8 forces one form (A/B measurements).
bool use_eight_waves(int M) {
    static const int forced = [] { const char* e = getenv("SAIS_ROW_WAVES"); return e ? atoi(e) : 0; }();
    if (forced == 4) return false;
    if (forced == 8) return true;
    return M >= 256 * 112;
}

template <int EPI, bool DP, int NW, bool STAG = false, bool SPEC = false>
int launch_row_nw(RowParams& p, void* stream) {
    static thread_local bool set = false;
    if (!set) {
        constexpr int lds_max = row_lds<NW, SPEC>();
        if (hipFuncSetAttribute((const void*)gemm_nt_row_kernel<EPI, DP, NW, STAG, SPEC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                lds_max) != hipSuccess)
            return SAIS_ERR_LAUNCH;
        set = true;
    }
    // 32-bit byte offsets inside the kernel
    if ((double)p.M * p.lda * 2.0 >= 4294967296.0 || (double)p.N * p.ldw * 2.0 >= 4294967296.0) return SAIS_ERR_ARG;
    p.rows_per_tile = rows_per_tile<NW>(p.M);
    const int grid = (p.N / RBN) * ((p.M + p.rows_per_tile - 1) / p.rows_per_tile);
    constexpr int lds = row_lds<NW, SPEC>();
    hipLaunchKernelGGL((gemm_nt_row_kernel<EPI, DP, NW, STAG, SPEC>), dim3(grid), dim3(64 * NW), lds, (hipStream_t)stream, p);
    return sais_check_launch();
}

template <int EPI, bool DP>
int launch_row_dp(RowParams& p, void* stream) {
    static const bool stag = [] { const char* e = getenv("SAIS_ROW_STAG"); return e ? atoi(e) != 0 : true; }();
    if (!use_eight_waves(p.M)) return launch_row_nw<EPI, DP, 4>(p, stream);
    static const bool spec = [] { const char* e = getenv("SAIS_ROW_SPEC"); return e ? atoi(e) != 0 : true; }();
    if (spec) return stag ? launch_row_nw<EPI, DP, 8, true, true>(p, stream) : launch_row_nw<EPI, DP, 8, false, true>(p, stream);
    return stag ? launch_row_nw<EPI, DP, 8, true>(p, stream) : launch_row_nw<EPI, DP, 8, false>(p, stream);
}

template <int EPI>
int launch_row(RowParams& p, void* stream) {
    if constexpr (EPI != ROW_BIAS_BF16) {
        if (p.rowscale) return launch_row_dp<EPI, true>(p, stream);
    }
    return launch_row_dp<EPI, false>(p, stream);
}

}  // namespace

// plain epilogues: called by sais_gemm_nt (gemm.hip) for the large-M ViT GEMMs whose N is a multiple of 384
extern "C" int sais_gemm_nt_row_(const SaisGemm* g, void* stream) {
    if (g->N % RBN) return SAIS_ERR_ARG;
    RowParams p{};
    p.A = (const bf16*)g->A; p.W = (const bf16*)g->B; p.lda = g->lda; p.ldw = g->ldb;
    p.M = g->M; p.N = g->N; p.K = g->K; p.bias = g->bias;
    p.out = g->out; p.ldo = g->ldo; p.out2 = g->out2; p.ldo2 = g->ldo2; p.aux = g->aux; p.ldaux = g->ldaux;
    p.rowscale = g->rowscale;
    switch (g->epilogue) {
        case SAIS_EPI_BIAS_BF16: return launch_row<ROW_BIAS_BF16>(p, stream);
        case SAIS_EPI_BIAS_RESID_F32: return g->out2 ? SAIS_ERR_ARG : launch_row<ROW_RESID_F32>(p, stream);
        default: return SAIS_ERR_ARG;
    }
}

static int check_ln(const SaisGemmLn* g) {
    if (!g || !g->A || !g->W || !g->resid || !g->gamma || g->M <= 0 || g->K <= 0 || g->K % RBK) return SAIS_ERR_ARG;
    if (g->lda % 8 || g->ldw % 8 || g->ldr % 4 || g->ldo32 % 4 || g->ldo16 % 8) return SAIS_ERR_ARG;
    return SAIS_OK;
}

extern "C" int sais_gemm_ln_fwd(const SaisGemmLn* g, void* stream) {
    SAIS_ENTER();
    if (check_ln(g) != SAIS_OK || !g->out32 || !g->out16 || !g->beta) return SAIS_ERR_ARG;
    RowParams p{};
    p.A = (const bf16*)g->A; p.W = (const bf16*)g->W; p.lda = g->lda; p.ldw = g->ldw;
    p.M = g->M; p.N = RBN; p.K = g->K; p.bias = g->bias;
    p.out = g->out32; p.ldo = g->ldo32; p.out2 = g->out16; p.ldo2 = g->ldo16; p.aux = g->resid; p.ldaux = g->ldr;
    p.gamma = g->gamma; p.beta = g->beta; p.eps = g->eps; p.mean = g->mean; p.rstd = g->rstd;
    p.rowscale = g->rowscale;
    return launch_row<ROW_LN_FWD>(p, stream);
}

extern "C" int sais_gemm_ln_bwd(const SaisGemmLn* g, void* stream) {
    SAIS_ENTER();
    if (check_ln(g) != SAIS_OK || !g->mean || !g->rstd || (!g->out32 && !g->out16)) return SAIS_ERR_ARG;
    if ((g->dgamma == nullptr) != (g->dbeta == nullptr) || g->lddres % 4) return SAIS_ERR_ARG;
    RowParams p{};
    p.A = (const bf16*)g->A; p.W = (const bf16*)g->W; p.lda = g->lda; p.ldw = g->ldw;
    p.M = g->M; p.N = RBN; p.K = g->K;
    p.out = g->out32; p.ldo = g->ldo32; p.out2 = g->out16; p.ldo2 = g->ldo16; p.aux = g->resid; p.ldaux = g->ldr;
    p.gamma = g->gamma; p.mean = g->mean; p.rstd = g->rstd;
    p.dres = g->dres; p.lddres = g->lddres; p.dres_period = g->dres_period; p.dgamma = g->dgamma; p.dbeta = g->dbeta;
    p.rowscale = g->rowscale16;
    static const bool x16 = [] { const char* e = getenv("SAIS_LN_BWD_X16"); return e ? atoi(e) != 0 : true; }();
    if (x16 && g->xn16 && g->beta) {
        if ((g->ldxn16 & 3) || ((uintptr_t)g->xn16 & 7)) return SAIS_ERR_ARG;
        p.xn16 = (const bf16*)g->xn16; p.ldxn16 = g->ldxn16; p.beta = g->beta;
    }
    return launch_row<ROW_LN_BWD>(p, stream);
}

CLK_EXPORT(row)
