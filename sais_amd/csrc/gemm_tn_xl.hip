// Large-tile weight-gradient GEMM (round 6): dW[N1,N2] += sum_m P[m,N1] Q[m,N2], db[N1] += colsum(P), bf16 operands.
//
// Why another dW kernel.  The 128 x 384 ping-pong kernel of gemm.hip is paced by bytes through the per-CU vector L1 (LABNOTES
// R5.5-R5.7: 1.78 GB of LDS fills per launch for 627 MB of unique operands, 16.6 B/clk per CU, TCP stalled on L2 data 62 % of
// its cycles), and five re-schedulings of that tile changed nothing.  This kernel changes the bytes per flop instead:
//
//   tile     192 (P columns = dW rows) x 384 (Q columns) per workgroup: 36 KiB of operands per 32-row step for
//            2 x 192 x 384 x 32 flop = 0.75 x the fill bytes per flop of the 128 x 384 tile (a 256-row tile would give 0.625 x
//            but 384, 1152 and the 384-row fc2 gradient are not multiples of 256; 192 divides every N1 of a ViT block)
//   waves    NW = 4: ONE wave per SIMD, 96 x 192 per wave = 3 x 6 tiles of v_mfma_f32_32x32x16_bf16 (288 accumulator registers,
//            AGPRs + VGPRs of the unified 512-entry file; this file is built WITHOUT -amdgpu-mfma-vgpr-form); per 16-deep
//            k-step a wave reads 9 fragments for 18 MFMAs (the 64 x 96 wave tile of the old kernel: 10 for 24 16x16x32 = 12
//            32x32x16-equivalents), and a 32-cycle MFMA leaves 24 issue cycles for the LDS reads and LDS-DMA issues of the
//            same wave.  NW = 8 (2 x 4 waves of 96 x 96, two per SIMD, <= 256 registers) is the same code, kept for the A/B.
//   staging  LDS-DMA only (global_load_lds_dwordx4, no staging registers), a ring of FOUR 36-KiB stages, pieces of three
//            steps in flight per wave (counted vmcnt), ONE s_barrier per 32-row step (36 MFMAs per wave at NW = 4).
//   image    rows of 384 B (P) / 768 B (Q), 32-B units XOR-swizzled on the SOURCE address so that the transposed fragment
//            reads (ds_read_b64_tr_b16: m is the slow index of both operands) are conflict-free: a 32-lane half reads
//            4 rows x 2 units, P unit u of row r sits at u ^ 2((r >> 1) & 1), Q unit u at u ^ 2(r & 3).
//   output   fp32 atomics straight from the 32x32 accumulator layout: one register = 32 consecutive dW columns of two rows
//            (two 128-B segments per wave-instruction: the full-rate shape of MI355X_MICROARCH.md, global float atomics).
//   bias     db = colsum(P) on the matrix pipe: one MFMA per P fragment against a one-hot B column, accumulated into ONE
//            extra 32x32 tile (column t <- fragment t); the k-steps are shared out over the waves of a row so that no wave
//            carries more than 3 (NW = 4) extra MFMAs per 36.
//
// M-splits: floor(256 / tiles) splits of (almost) equal numbers of 32-row steps, one workgroup per CU, one round.
// Reference math: the weight / bias gradients of F.linear — autograd of vision_transformer.py:59-65 (Mlp), :80-92 (Attention).
#include <type_traits>
#include <utility>
#include "common.hpp"
#include "../../include/sais_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int XP = 192, XQ = 384, XK = 32, XNST = 4;
constexpr int XPROW = XP * 2, XQROW = XQ * 2;              // image row bytes
constexpr int XPIMG = XK * XPROW;                          // 12 KiB
constexpr int XQIMG = XK * XQROW;                          // 24 KiB
constexpr int XSTAGE = XPIMG + XQIMG;                      // 36 KiB
constexpr int XLDS = XNST * XSTAGE;                        // 144 KiB

struct XlItem {
    const bf16* P; const bf16* Q; float* dW; float* db;
    int ldp, ldq, ldw, nt2;
};
struct XlGroup {
    XlItem item[SAIS_TN_MAX_ITEMS];
    int tile_end[SAIS_TN_MAX_ITEMS];
    int nitems, ntiles, nsteps, nsplit;                    // nsteps = M / 32 over all splits
};

DEVINL f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
template <class Fn, int... I> DEVINL void static_for_impl(Fn&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class Fn> DEVINL void static_for(Fn&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
template <int N> DEVINL void xl_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// SAIS_XL_ABL (timing ablations, WRONG results): 1 no LDS-DMA, 2 no MFMAs, 4 no fragment reads, 8 no atomics
#ifndef SAIS_XL_ABL
#define SAIS_XL_ABL 0
#endif
#ifndef SAIS_EXPERIMENTAL
#define SAIS_EXPERIMENTAL 0
#endif

// SLAB: instead of fp32 atomics (75 MB per ViT block at 10 splits: ~50 us of the launch, LABNOTES R6.1: the chip retires ~1.5 TB/s
// of them and every workgroup flushes at the same moment) each workgroup stores its raw partial tile ONCE, in register order
// (16 B per lane, 1 KiB per wave-instruction: plain stores at HBM speed), and xl_finish_kernel sums the splits in a fixed order
// into dW / db: deterministic gradients as a by-product.  Slab of workgroup z = split * ntiles + tile:
//   [z][wave][tile k of the wave][register quad r4][lane] f32x4   +   bias part [z][wave][r4][lane] f32x4 behind all tiles
template <int NW, bool SLAB>
__global__ __launch_bounds__(64 * NW, NW / 4) void gemm_tn_xl_kernel(XlGroup gp, float* slabs) {
    extern __shared__ __attribute__((aligned(1024))) char xsm[];
    CLK_STAMP(3);
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int split = wg / gp.ntiles;
    int t = wg - split * gp.ntiles, it0 = 0;
    while (it0 + 1 < gp.nitems && t >= gp.tile_end[it0]) ++it0;
    if (it0 > 0) t -= gp.tile_end[it0 - 1];
    const XlItem& p = gp.item[it0];
    const int n1_0 = (t / p.nt2) * XP, n2_0 = (t % p.nt2) * XQ;
    const int sbase = gp.nsteps / gp.nsplit, srem = gp.nsteps % gp.nsplit;
    const int s0 = split * sbase + min(split, srem);       // first 32-row step of this split
    const int nsteps = sbase + (split < srem ? 1 : 0);     // >= 4 (host)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NWC = NW / 2;                            // waves along Q
    constexpr int AI = 3, BJ = XQ / 32 / NWC;              // 32x32 tiles per wave: 3 x 6 (NW = 4) or 3 x 3
    const int wr = wid / NWC, wc = wid % NWC;
    const int ldp = p.ldp, ldq = p.ldq;

    // transposed fragment reads: 16-lane group gi -> columns 16 (gi & 1) .. + 15 of a 32-column block, k half h = gi >> 1;
    // lane 4 q + pq of the group supplies row q, columns 4 pq .. 4 pq + 3.  k slot (h, e) of k-step s <-> m = 16 s + 8 h + e.
    const int gi = lane >> 4, h = gi >> 1, g1 = gi & 1, q = (lane >> 2) & 3, pq = lane & 3;
    unsigned pa[AI], qa[BJ];
#pragma unroll
    for (int i = 0; i < AI; ++i)
        pa[i] = (8 * h + q) * XPROW + (((2 * (wr * AI + i) + g1) ^ (2 * (q >> 1))) << 5) + 8 * pq;
#pragma unroll
    for (int j = 0; j < BJ; ++j)
        qa[j] = XPIMG + (8 * h + q) * XQROW + (((2 * (wc * BJ + j) + g1) ^ (2 * q)) << 5) + 8 * pq;

    // NW = 4: 18 accumulator tiles = 288 registers.  The encoding gives a wave 256 VGPRs + 256 AGPRs, and left to itself hipcc
    // shuffles tiles between the two files inside the loop (816 v_accvgpr_read + 1089 _write in the first build), so the MFMAs of
    // this variant are inline asm with the file pinned per tile: tiles 0-15 in AGPRs ("+a"), tiles 16-17 and the bias tile in VGPRs.
    constexpr int NT = AI * BJ, NTA = NW == 4 ? 16 : 0;
    f32x16 acc[NT], accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
#pragma unroll
    for (int k = 0; k < NT; ++k) acc[k] = accb;
    auto mma_tile = [&](auto k_c, const bf16x8& a, const bf16x8& b) {
        constexpr int K = decltype(k_c)::value;
        if constexpr (NW == 4 && K < NTA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[K]) : "v"(a), "v"(b));
        else if constexpr (NW == 4) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[K]) : "v"(a), "v"(b));
        else acc[K] = mfma32(a, b, acc[K]);
    };
    auto mma_bias = [&](const bf16x8& a, const bf16x8& b) {
        if constexpr (NW == 4) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accb) : "v"(a), "v"(b));
        else accb = mfma32(a, b, accb);
    };
    const bool do_bias = p.db != nullptr && n2_0 == 0;
    bf16x8 oneh[AI];                                       // B operand with ones in output column i
#pragma unroll
    for (int i = 0; i < AI; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) oneh[i][e] = (bf16)((lane & 31) == i ? 1.0f : 0.0f);
    // (fragment i, k-step s) of the bias product runs on wave column (NW = 4: s; NW = 8: (3 s + i) & 3)
    bool bflag[2][AI];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < AI; ++i) bflag[s][i] = do_bias && (NW == 4 ? s == wc : ((3 * s + i) & 3) == wc);

    const char* const Pb = (const char*)(p.P + (size_t)s0 * XK * ldp + n1_0);
    const char* const Qb = (const char*)(p.Q + (size_t)s0 * XK * ldq + n2_0);
    const size_t pstep = (size_t)XK * ldp * 2, qstep = (size_t)XK * ldq * 2;

    auto body = [&](auto npp_c) {
        constexpr int NPP = decltype(npp_c)::value;        // P pieces (1 KiB) of a stage this wave issues; 12 in all
        constexpr int NQP = 24 / NW;                       // Q pieces: 24 in all
        constexpr int NPW = NPP + NQP;
        const int pp0 = NW == 4 ? 3 * wid : (NPP == 2 ? 2 * wid : 4 + wid);
        const int qp0 = NQP * wid;
        unsigned poff[NPP], qoff[NQP];
#pragma unroll
        for (int j = 0; j < NPP; ++j) {
            const int c = (pp0 + j) * 64 + lane, row = c / 24, ch = c - row * 24;
            const int u = (ch >> 1) ^ (2 * ((row >> 1) & 1));
            poff[j] = (unsigned)(row * ldp + u * 16 + (ch & 1) * 8) * 2u;
        }
#pragma unroll
        for (int j = 0; j < NQP; ++j) {
            const int c = (qp0 + j) * 64 + lane, row = c / 48, ch = c - row * 48;
            const int u = (ch >> 1) ^ (2 * (row & 3));
            qoff[j] = (unsigned)(row * ldq + u * 16 + (ch & 1) * 8) * 2u;
        }
        auto issue = [&](int step) {
            if (SAIS_XL_ABL & 1) return;
            char* s = xsm + (step & (XNST - 1)) * XSTAGE;
            const char* pb = Pb + step * pstep;
            const char* qb = Qb + step * qstep;
#pragma unroll
            for (int j = 0; j < NPP; ++j) glds16(pb + poff[j], s + (pp0 + j) * 1024);
#pragma unroll
            for (int j = 0; j < NQP; ++j) glds16(qb + qoff[j], s + XPIMG + (qp0 + j) * 1024);
        };
        // Fragments as 8-byte halves (one ds_read_b64_tr_b16 each).  The transposed reads are INLINE ASM: behind an LDS-DMA issue
        // hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of every LDS read it knows about (the builtin carries an LDS memory
        // operand; SIInsertWaitcnts orders it behind every pending LDS-DMA store), which drains the three-step prefetch once per
        // step.  An asm read is invisible to that pass — and to its lgkmcnt bookkeeping, so the waits are written by hand and name
        // the registers they validate ("+v": nothing the compiler derives from a fragment can be placed before its wait).
        struct Frags { u32x2 a[AI][2]; u32x2 b[BJ][2]; } f0, f1;
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)xsm;
        auto tr_asm = [&](u32x2& dst, unsigned addr, auto off_c) {
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(decltype(off_c)::value));
        };
        // transposed read number E (0 .. 2 (AI + BJ) - 1) of k-step S of the stage at LDS offset `st`: fragments in the order the
        // MFMAs first need them (A0, B0 .. B5, A1, A2), low half (k rows +0..3) then high half (+4..7)
        auto rd1 = [&](auto e_c, Frags& f, unsigned st, auto s_c) {
            constexpr int E = decltype(e_c)::value, F = E >> 1, HI = E & 1, S = decltype(s_c)::value;
            if (SAIS_XL_ABL & 4) return;
            if constexpr (F == 0) tr_asm(f.a[0][HI], st + pa[0], std::integral_constant<int, (16 * S + 4 * HI) * XPROW>{});
            else if constexpr (F <= BJ) tr_asm(f.b[F - 1][HI], st + qa[F - 1], std::integral_constant<int, (16 * S + 4 * HI) * XQROW>{});
            else tr_asm(f.a[F - BJ][HI], st + pa[F - BJ], std::integral_constant<int, (16 * S + 4 * HI) * XPROW>{});
        };
        auto rd = [&](Frags& f, unsigned st, auto s_c) {
            static_for<2 * (AI + BJ)>([&](auto e_c) { rd1(e_c, f, st, s_c); });
        };
        // lgkmcnt(N) naming the fragment halves it validates
        auto wait_a = [&](auto n_c, Frags& f, auto i_c) {
            constexpr int I = decltype(i_c)::value;
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.a[I][0]), "+v"(f.a[I][1]) : "i"(decltype(n_c)::value));
        };
        auto wait_b = [&](auto n_c, Frags& f, auto j_c) {
            constexpr int J = decltype(j_c)::value;
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.b[J][0]), "+v"(f.b[J][1]) : "i"(decltype(n_c)::value));
        };
        auto wait_all = [&](Frags& f) {
            using Z = std::integral_constant<int, 0>;
            static_for<AI>([&](auto i_c) { wait_a(Z{}, f, i_c); });
            static_for<BJ>([&](auto j_c) { wait_b(Z{}, f, j_c); });
        };
        auto frag8 = [&](const u32x2& lo, const u32x2& hi) {
            return __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
        };
        // MFMA number K (0 .. NT - 1) of a k-step: tile (K / BJ, K % BJ); the bias product of fragment i behind its last tile
        auto mm1 = [&](auto k_c, const Frags& f, int s) {
            constexpr int K = decltype(k_c)::value, I = K / BJ, J = K % BJ;
            if (SAIS_XL_ABL & 2) return;
            const bf16x8 a = frag8(f.a[I][0], f.a[I][1]);
            mma_tile(k_c, a, frag8(f.b[J][0], f.b[J][1]));
            if constexpr (J == BJ - 1) {
                if (bflag[s][I]) mma_bias(a, oneh[I]);
            }
        };
        auto stage_of = [&](int step) { return xsm + (step & (XNST - 1)) * XSTAGE; };
        auto stage_lds = [&](int step) { return lds0 + (unsigned)((step & (XNST - 1)) * XSTAGE); };
        auto top_sync = [&](auto next_c, auto wait_c) {
            if constexpr (decltype(next_c)::value && !(SAIS_XL_ABL & 1)) xl_wait_vm<decltype(wait_c)::value>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        // One 32-row step.  On entry the fragments of (step, k-step 0) are in f0 (possibly still in flight).
        //   wait: this wave's pieces of step + 1 have landed (those of step + 2 may fly)
        //   barrier: step + 1 is complete for every wave, and every wave is done reading stage (step - 1) & 3
        //   then issue step + 3 into that stage.  Reads of (step, 1) run under the MFMAs of (step, 0), reads of (step + 1, 0)
        //   under the MFMAs of (step, 1).
        // NW = 8: two waves per SIMD cover each other; source order = phases, the compiler schedules inside them.
        auto step8 = [&](int st, auto next_c, auto issue_c, auto wait_c) {
            using S0 = std::integral_constant<int, 0>;
            using S1 = std::integral_constant<int, 1>;
            top_sync(next_c, wait_c);
            if constexpr (decltype(issue_c)::value) issue(st + 3);
            rd(f1, stage_lds(st), S1{});
            static_for<NT>([&](auto k_c) { mm1(k_c, f0, 0); });
            wait_all(f1);
            if constexpr (decltype(next_c)::value) rd(f0, stage_lds(st + 1), S0{});
            static_for<NT>([&](auto k_c) { mm1(k_c, f1, 1); });
            if constexpr (decltype(next_c)::value) wait_all(f0);
        };
        // NW = 4: one wave per SIMD, so the wave's own stream has to keep the matrix pipe fed: hand-placed.  The MFMAs are
        // volatile asm (they stay in source order, and every LDS read / LDS-DMA issue stays where it is written between them):
        // gap g (0 .. 35) of a step carries one transposed read — the 18 of the NEXT k-step, in the order its MFMAs need them —
        // and, from gap 9 on, every third gap one LDS-DMA piece of step + 3.  The wait + barrier sit in gap 9: the reads of
        // gaps 0-17 touch the CURRENT stage (visible since the previous step's barrier), those of gaps 18-35 the next one.
        auto step4 = [&](int st, auto next_c, auto issue_c, auto wait_c) {
            constexpr bool NEXT = decltype(next_c)::value, ISSUE = decltype(issue_c)::value;
            const unsigned cur = stage_lds(st), nxt = stage_lds(st + 1);
            char* dst = stage_of(st + 3);
            const char* pb = Pb + (st + 3) * pstep;
            const char* qb = Qb + (st + 3) * qstep;
            static_for<2 * NT>([&](auto g_c) {
                constexpr int G = decltype(g_c)::value, S = G / NT, K = G % NT;
                if constexpr (G == 9) top_sync(next_c, wait_c);
                // the batch this k-step consumes was read in the previous k-step, one read per gap, this k-step has issued K
                // reads of the next batch so far: fragment with last read number e is valid at lgkmcnt <= (17 - e) + K
                // (no reads in flight behind the batch in the very last k-step: plain lgkmcnt(0) there)
                constexpr bool LASTK = S == 1 && !NEXT;
                Frags& fc = S == 0 ? f0 : f1;
                if constexpr (LASTK) {
                    if constexpr (K == 0) wait_all(fc);
                } else if constexpr (K == 0) {
                    asm volatile("s_waitcnt lgkmcnt(14)" : "+v"(fc.a[0][0]), "+v"(fc.a[0][1]), "+v"(fc.b[0][0]), "+v"(fc.b[0][1]));   // e = 3
                } else if constexpr (K < BJ) {
                    wait_b(std::integral_constant<int, 14 - K>{}, fc, std::integral_constant<int, K>{});  // e = 3 + 2 K
                } else if constexpr (K % BJ == 0) {                                                       // A1 (e = 15), A2 (e = 17)
                    wait_a(std::integral_constant<int, (K == BJ ? 2 : 0) + K>{}, fc, std::integral_constant<int, K / BJ>{});
                }
                mm1(std::integral_constant<int, K>{}, fc, S);
                if constexpr (S == 0) rd1(std::integral_constant<int, K>{}, f1, cur, std::integral_constant<int, 1>{});
                else if constexpr (NEXT) rd1(std::integral_constant<int, K>{}, f0, nxt, std::integral_constant<int, 0>{});
                if constexpr (ISSUE && G >= 9 && (G - 9) % 3 == 0 && (G - 9) / 3 < NPW && !(SAIS_XL_ABL & 1)) {
                    constexpr int PC = (G - 9) / 3;
                    if constexpr (PC < NPP) glds16(pb + poff[PC], dst + (pp0 + PC) * 1024);
                    else glds16(qb + qoff[PC - NPP], dst + XPIMG + (qp0 + PC - NPP) * 1024);
                }
            });
        };
        auto step = [&](int st, auto next_c, auto issue_c, auto wait_c) {
            if constexpr (NW == 4) step4(st, next_c, issue_c, wait_c);
            else step8(st, next_c, issue_c, wait_c);
        };
        using T = std::true_type;
        using F = std::false_type;
        issue(0);
        issue(1);
        issue(2);
        if (!(SAIS_XL_ABL & 1)) xl_wait_vm<2 * NPW>();
        __builtin_amdgcn_s_barrier();
        rd(f0, stage_lds(0), std::integral_constant<int, 0>{});
        if constexpr (NW == 8) wait_all(f0);
        int st = 0;
        for (; st < nsteps - 3; ++st) step(st, T{}, T{}, std::integral_constant<int, NPW>{});
        step(st, T{}, F{}, std::integral_constant<int, NPW>{});          // nsteps - 3: step + 2 is the last one in flight
        step(st + 1, T{}, F{}, std::integral_constant<int, 0>{});        // nsteps - 2
        step(st + 2, F{}, F{}, std::integral_constant<int, 0>{});        // nsteps - 1
    };
    if constexpr (NW == 4) body(std::integral_constant<int, 3>{});
    else if (wid < 4) body(std::integral_constant<int, 2>{});
    else body(std::integral_constant<int, 1>{});
    // NW = 4: the compiler does not know that the asm statements above are MFMAs still in flight (first build: it spilled the
    // bias tile and copied accumulators right behind the last MFMA and got stale registers).  Every accumulator passes
    // through this statement, so nothing that reads one can be placed in front of the wait states.
    if constexpr (NW == 4)
        asm volatile("s_nop 15\n\ts_nop 15"
                     : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]),
                       "+a"(acc[8]), "+a"(acc[9]), "+a"(acc[10]), "+a"(acc[11]), "+a"(acc[12]), "+a"(acc[13]), "+a"(acc[14]),
                       "+a"(acc[15]), "+v"(acc[NT - 2]), "+v"(acc[NT - 1]), "+v"(accb));

    if (SAIS_XL_ABL & 8) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < AI; ++i)
#pragma unroll
            for (int j = 0; j < BJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i * BJ + j][r];
        if (s == 12345.678f) p.dW[0] = s;
        return;
    }
    const int hh = lane >> 5, col = lane & 31;
    if constexpr (SLAB) {
        const int z = split * gp.ntiles + (wg - split * gp.ntiles);
        f32x4* o = (f32x4*)slabs + ((size_t)(z * NW + wid) * NT) * 4 * 64 + lane;
#pragma unroll
        for (int k = 0; k < NT; ++k)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                o[(k * 4 + r4) * 64] = f32x4{acc[k][4 * r4], acc[k][4 * r4 + 1], acc[k][4 * r4 + 2], acc[k][4 * r4 + 3]};
        if (do_bias) {
            f32x4* ob = (f32x4*)slabs + (size_t)gridDim.x * NW * NT * 4 * 64 + (size_t)(z * NW + wid) * 4 * 64 + lane;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) ob[r4 * 64] = f32x4{accb[4 * r4], accb[4 * r4 + 1], accb[4 * r4 + 2], accb[4 * r4 + 3]};
        }
        return;
    }
    // D[n1][n2]: register r of tile (i, j) = rows 32 i + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column 32 j + (lane & 31)
    float* const out = p.dW + (size_t)(n1_0 + wr * 96 + 4 * hh) * p.ldw + n2_0 + wc * (32 * BJ) + col;
    if (gp.nsplit == 1) {                                    // the tile has ONE owner: plain read-add-write, 128-B row segments
#pragma unroll
        for (int i = 0; i < AI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* row = out + (size_t)(32 * i + (r & 3) + 8 * (r >> 2)) * p.ldw;
                float old[BJ];
#pragma unroll
                for (int j = 0; j < BJ; ++j) old[j] = row[32 * j];
#pragma unroll
                for (int j = 0; j < BJ; ++j) row[32 * j] = old[j] + acc[i * BJ + j][r];
            }
    } else {
#pragma unroll
        for (int i = 0; i < AI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* row = out + (size_t)(32 * i + (r & 3) + 8 * (r >> 2)) * p.ldw;
#pragma unroll
                for (int j = 0; j < BJ; ++j) atomicAdd(row + 32 * j, acc[i * BJ + j][r]);
            }
    }
    if (do_bias && col < AI) {                               // column i of accb = the sums of fragment i's 32 rows
        float* b = p.db + n1_0 + wr * 96 + 32 * col + 4 * hh;
#pragma unroll
        for (int r = 0; r < 16; ++r) atomicAdd(b + (r & 3) + 8 * (r >> 2), accb[r]);
    }
}

// Fixed-order sum of the split slabs into dW / db.  One thread per (tile, wave, wave-tile k, register quad r4, lane): nsplit
// independent 16-B loads (all in flight), then four read-add-writes of 128-B row segments.  grid (ntiles * NW * NT * 4 * 64 / 256)
// + one block row for the bias parts.
template <int NW>
__global__ __launch_bounds__(256) void xl_finish_kernel(XlGroup gp, const float* slabs, int nwg) {
    constexpr int NWC = NW / 2, AI = 3, BJ = XQ / 32 / NWC, NT = AI * BJ;
    constexpr int PER_TILE = NW * NT * 4 * 64;                 // f32x4 elements of one workgroup's slab
    const int nbody = gp.ntiles * PER_TILE / 256;
    const int ns = gp.nsplit;
    if ((int)blockIdx.x < nbody) {
        const int e = blockIdx.x * 256 + threadIdx.x;
        int t = e / PER_TILE;
        const int w = e - t * PER_TILE, lane = w & 63, r4 = (w >> 6) & 3, k = (w >> 8) % NT, wid = (w >> 8) / NT;
        const f32x4* src = (const f32x4*)slabs + (size_t)t * PER_TILE + w;
        const size_t zs = (size_t)gp.ntiles * PER_TILE;
        f32x4 v[12];
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        // splits in a fixed order, 12 loads in flight (10 splits at training size): the loads are UNCONDITIONAL (index clamped, value selected afterwards) — a load
        // under a run-time condition makes hipcc branch around it and wait for each one (cdna_hip_programming.md, "three .s-level traps" (c))
        for (int z0 = 0; z0 < ns; z0 += 12) {
#pragma unroll
            for (int u = 0; u < 12; ++u) v[u] = src[(size_t)min(z0 + u, ns - 1) * zs];
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                const float keep = z0 + u < ns ? 1.f : 0.f;
                sum += v[u] * keep;
            }
        }
        int it0 = 0;
        while (it0 + 1 < gp.nitems && t >= gp.tile_end[it0]) ++it0;
        if (it0 > 0) t -= gp.tile_end[it0 - 1];
        const XlItem& p = gp.item[it0];
        const int n1_0 = (t / p.nt2) * XP, n2_0 = (t % p.nt2) * XQ;
        const int wr = wid / NWC, wc = wid % NWC, i = k / BJ, j = k % BJ;
        float* row = p.dW + (size_t)(n1_0 + wr * 96 + 32 * i + 8 * r4 + 4 * (lane >> 5)) * p.ldw + n2_0 + wc * (32 * BJ) + 32 * j + (lane & 31);
#pragma unroll
        for (int c = 0; c < 4; ++c) row[(size_t)c * p.ldw] += sum[c];
        return;
    }
    // bias parts: [z][wave][r4][lane]; column i (lane & 31 == i < 3) of accb holds the sums of fragment i's 32 rows.  The k-steps
    // of the bias product are shared out over the waves of a row, so one thread sums the NWC waves and the splits, fixed order.
    const int e = (blockIdx.x - nbody) * 256 + threadIdx.x;
    if (e >= gp.ntiles * 2 * 4 * 64) return;
    const int t = e / (2 * 4 * 64);
    const int w = e - t * (2 * 4 * 64), lane = w & 63, r4 = (w >> 6) & 3, wr = w >> 8;
    if ((lane & 31) >= AI) return;
    int it0 = 0, tt = t;
    while (it0 + 1 < gp.nitems && tt >= gp.tile_end[it0]) ++it0;
    if (it0 > 0) tt -= gp.tile_end[it0 - 1];
    const XlItem& p = gp.item[it0];
    if (p.db == nullptr || tt % p.nt2 != 0) return;
    const f32x4* src = (const f32x4*)slabs + (size_t)nwg * NW * NT * 4 * 64 + (size_t)t * (NW * 4 * 64) + (wr * NWC * 4 + r4) * 64 + lane;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < ns; ++z)
        for (int wc = 0; wc < NWC; ++wc) sum += src[(size_t)z * gp.ntiles * (NW * 4 * 64) + wc * 4 * 64];
    float* b = p.db + (tt / p.nt2) * XP + wr * 96 + 32 * (lane & 31) + 8 * r4 + 4 * (lane >> 5);
#pragma unroll
    for (int c = 0; c < 4; ++c) b[c] += sum[c];
}

}  // namespace

// Applies when every item has N1 % 192 == 0 and N2 % 384 == 0, M % 32 == 0 and every split gets a worthwhile number of steps
// (a launch with 4 tiles would be 64 splits of 24 steps, each flushing a whole tile: the 128 x 384 kernel is faster there).
static bool xl_plan(const SaisTnItem* items, int nitems, int M, XlGroup& gp) {
    if (M % XK || M < 8192 || nitems > SAIS_TN_MAX_ITEMS) return false;
    int nt = 0;
    for (int i = 0; i < nitems; ++i) {
        const SaisTnItem& t = items[i];
        if (t.N1 % XP || t.N2 % XQ || t.ldp % 8 || t.ldq % 8) return false;
        if (((uintptr_t)t.P & 15) || ((uintptr_t)t.Q & 15)) return false;
        gp.item[i] = XlItem{(const bf16*)t.P, (const bf16*)t.Q, t.dW, t.db, t.ldp, t.ldq, t.ldw, t.N2 / XQ};
        nt += (t.N1 / XP) * (t.N2 / XQ);
        gp.tile_end[i] = nt;
    }
    gp.nitems = nitems;
    gp.ntiles = nt;
    gp.nsteps = M / XK;
    const int ns = 256 / nt < 1 ? 1 : 256 / nt;
    if (gp.nsteps / ns < 48) return false;
    gp.nsplit = ns;
    return true;
}

static size_t xl_slab_need(const XlGroup& gp, int nwaves) {
    if (!SAIS_EXPERIMENTAL) nwaves = 4;
    const int nt_wave = nwaves == 8 ? 9 : 18;
    const size_t nwg = (size_t)gp.ntiles * gp.nsplit;
    return nwg * nwaves * nt_wave * 4 * 64 * 16 + nwg * nwaves * 4 * 64 * 16;
}

// bytes of slab workspace the launch would use (0: the regime does not apply)
extern "C" size_t sais_gemm_tn_xl_slab_bytes_(const SaisTnItem* items, int nitems, int M, int nwaves) {
    XlGroup gp;
    if (!xl_plan(items, nitems, M, gp) || gp.nsplit < 2) return 0;
    return xl_slab_need(gp, nwaves);
}

template <int NW, bool SLAB>
static int xl_launch(const XlGroup& gp, float* slabs, hipStream_t stream) {
    static thread_local bool set = false;
    if (!set) {
        if (hipFuncSetAttribute((const void*)gemm_tn_xl_kernel<NW, SLAB>, hipFuncAttributeMaxDynamicSharedMemorySize, XLDS) != hipSuccess)
            return SAIS_ERR_LAUNCH;
        set = true;
    }
    const int nwg = gp.ntiles * gp.nsplit;
    hipLaunchKernelGGL((gemm_tn_xl_kernel<NW, SLAB>), dim3(nwg), dim3(64 * NW), XLDS, stream, gp, slabs);
    if constexpr (SLAB) {
        constexpr int NT = NW == 8 ? 9 : 18;
        const int nbody = gp.ntiles * (NW * NT * 4 * 64) / 256, nbias = (gp.ntiles * 2 * 4 * 64 + 255) / 256;
        hipLaunchKernelGGL(xl_finish_kernel<NW>, dim3(nbody + nbias), dim3(256), 0, stream, gp, (const float*)slabs, nwg);
    }
    return sais_check_launch() == SAIS_OK ? 1 : SAIS_ERR_LAUNCH;
}

// Returns 1 when the launch was made, 0 when the regime does not apply (caller falls back), < 0 on error.
// slabs (16-B aligned, >= sais_gemm_tn_xl_slab_bytes_) selects the atomics-free form; NULL = fp32 atomics.
extern "C" int sais_gemm_tn_xl_(const SaisTnItem* items, int nitems, int M, int nwaves, void* slabs, size_t slab_bytes, void* stream) {
    XlGroup gp;
    if (!xl_plan(items, nitems, M, gp)) return 0;
    const bool slab = slabs != nullptr && gp.nsplit >= 2;
    if (slab && (slab_bytes < xl_slab_need(gp, nwaves) || ((uintptr_t)slabs & 15))) return SAIS_ERR_ARG;
#if SAIS_EXPERIMENTAL      // the eight-wave form of the same tile (two waves per SIMD, compiler-scheduled): slower, LABNOTES R6.1
    if (nwaves == 8) return slab ? xl_launch<8, true>(gp, (float*)slabs, (hipStream_t)stream) : xl_launch<8, false>(gp, nullptr, (hipStream_t)stream);
#endif
    return slab ? xl_launch<4, true>(gp, (float*)slabs, (hipStream_t)stream) : xl_launch<4, false>(gp, nullptr, (hipStream_t)stream);
}
