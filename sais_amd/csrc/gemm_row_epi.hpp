// Epilogues of the row-owning GEMMs (gemm_row.hip) and of the fused MLP kernels (mlp_fused.hip): a workgroup owns whole
// rows of a [rows, 384] fp32 accumulator tile (4 waves side by side in N per 112-row half, 7 x 6 MFMA 16x16 tiles per wave,
// operands swapped so that a lane owns one output row per row tile and 8 contiguous columns per 32-column chunk).  The
// tile is handed, 32 rows per half at a time, through LDS (fp32 slab, rows padded to 1552 B) to a streaming phase in which
// the half-waves of the workgroup treat whole rows exactly like the stand-alone LayerNorm kernels (norm.hip).
#pragma once
#include "common.hpp"

namespace {

constexpr int RBN = 384, RBK = 64, RMT = 7;         // 7 row tiles of 16 = 112 rows computed per workgroup half


enum { ROW_BIAS_BF16 = 0, ROW_RESID_F32, ROW_LN_FWD, ROW_LN_BWD };

struct RowParams {
    const bf16* A; const bf16* W;
    int lda, ldw, M, N, K, rows_per_tile;
    const float* bias;              // [N] or null
    void* out; int ldo;             // bf16 out (BIAS) | f32 x_out (RESID, LN_FWD) | f32 dx (LN_BWD)
    void* out2; int ldo2;           // bf16: xn (LN_FWD) | dx (LN_BWD)
    const void* aux; int ldaux;     // f32 residual (RESID, LN_FWD) | f32 LN input x (LN_BWD)
    const float* gamma; const float* beta; float eps;
    float* mean; float* rstd;       // LN_FWD: out (nullable) | LN_BWD: in
    const float* dres; int lddres;  // LN_BWD: residual-stream gradient added to dx (may alias out)
    int dres_period;                // LN_BWD: > 0 = dres is COMPACT: row m of the residual-stream gradient is dres[m / period] when
                                    // m % period == 0 and zero otherwise (the gradient entering the last ViT block: CLS rows only)
    float* dgamma; float* dbeta;    // LN_BWD: +=
    // DropPath (stochastic depth, train mode; kernels instantiated with DP = true): per-row scale s[m] = keep / (1 - p) of
    // the sample the row belongs to.  RESID / LN_FWD: x_out = residual + s (acc + bias).  LN_BWD: the bf16 copy of dx that
    // feeds the NEXT branch's backward GEMMs is s dx (the fp32 residual-stream gradient is not scaled).
    const float* rowscale;
    const bf16* xn16; int ldxn16;   // LN_BWD, optional: the saved bf16 LayerNorm output; xhat = (xn16 - beta) / gamma instead of (aux - mean) rstd
};

// acc: this wave's 7 x 6 accumulator tiles; smem: the workgroup's dynamic LDS (>= 32 (NW/4) x 1552 B + 6 KiB; the operand
// slots are free by now — the caller has passed a barrier after its last LDS read); rows [m0, mend) are the workgroup's,
// n0 = first output column (a multiple of 384).
// PF = rows of epilogue operands (residual / LayerNorm input / gradient rows: first-touch HBM data) a half-wave keeps in
// flight.  1: the next row is fetched while the current one is processed (the row-owning GEMMs: a second workgroup or
// wave group on the CU hides the rest).  > 1 (fused MLP kernel: ONE wave per SIMD, 512 registers, nothing else on the CU to
// hide a round trip behind): a ring of PF rows, the row loops fully unrolled.
template <int EPI, bool DP, int NW, int PF = 1>
DEVINL void row_epilogue(const RowParams& p, f32x4 (&acc)[RMT][6], char* smem, int m0, int mend, int n0) {
    constexpr int HALVES = NW / 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wid >> 2, wq = wid & 3;
    const int g = lane >> 4, li = lane & 15;
    // lane: tile rows 112 half + 16 mt + li (mt = 0..6); per chunk c the 8 columns 96 wq + 32 c + 8 g + (4 t + e)
    const int cbase = 96 * wq + 8 * g;                               // + 32 c
    {
        // ---- every epilogue: (32 HALVES)-row fp32 slabs through LDS, then a row-streaming phase -------------------
        constexpr int SLD = 388;                                     // floats per slab row (1552 B: conflict-free dumps)
        constexpr int SROWS = 32 * HALVES;                           // rows per slab
        constexpr int NHW = 2 * NW;                                  // half-waves that stream rows
        float* const slab = (float*)smem;                            // [SROWS][SLD]; the operand slots are free now
        float* const scr = slab;                                     // LN_BWD column sums [NHW][2][384]: after the last slab
        const int l32 = tid & 31, hw = tid >> 5;
        auto clampm = [&](int m) { return m < p.M ? m : p.M - 1; };
        auto ld12 = [&](const float* q, float (&v)[12]) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const f32x4 t = *(const f32x4*)(q + 128 * i + 4 * l32);
                v[4 * i] = t[0]; v[4 * i + 1] = t[1]; v[4 * i + 2] = t[2]; v[4 * i + 3] = t[3];
            }
        };
        auto st12 = [&](float* q, const float (&v)[12]) {
#pragma unroll
            for (int i = 0; i < 3; ++i) *(f32x4*)(q + 128 * i + 4 * l32) = f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
        };
        auto st12_bf16 = [&](bf16* q, const float (&v)[12]) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                bf16x4 t;
                t[0] = (bf16)v[4 * i]; t[1] = (bf16)v[4 * i + 1]; t[2] = (bf16)v[4 * i + 2]; t[3] = (bf16)v[4 * i + 3];
                *(bf16x4*)(q + 128 * i + 4 * l32) = t;
            }
        };
        auto half_sum = [&](float v) {
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
            return v;
        };
        // slab s = local row tiles 2 s, 2 s + 1 of EVERY half (the last slab: local tile 6): slab rows 32 h .. 32 h + 31
        // belong to half h, so all waves free their accumulators at the same pace
        auto dump = [&](int s) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int mt = 2 * s + h;
                if (mt >= RMT) break;
                float* row = slab + (32 * half + 16 * h + li) * SLD + cbase;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    *(f32x4*)(row + 32 * c) = acc[mt][2 * c];
                    *(f32x4*)(row + 32 * c + 4) = acc[mt][2 * c + 1];
                }
            }
        };
        // the rows a half-wave streams: iteration it = 4 s + q' (slab 3: q' = 0, 1).  q' -> q (slab 3: q = HALVES q'),
        // half hh = q / QPH, local row = hw + NHW (q % QPH): slab row 32 hh + local, tile row 112 hh + 32 s + local
        constexpr int QPH = 4 / HALVES;
        auto qof = [&](int it) { return it < 12 ? (it & 3) : (it - 12) * HALVES; };
        auto srow = [&](int it) { const int q = qof(it); return 32 * (q / QPH) + hw + NHW * (q % QPH); };
        auto trow = [&](int it) { const int q = qof(it); return 112 * (q / QPH) + 32 * (it >> 2) + hw + NHW * (q % QPH); };
        constexpr int NIT = 14;                                      // 3 x 4 + 2

        if constexpr (EPI == ROW_RESID_F32 && PF > 1) {
            float bs[12], rq[PF][12];
            auto ldaux = [&](int it, float (&v)[12]) {
                ld12((const float*)p.aux + (size_t)clampm(m0 + trow(it)) * p.ldaux + n0, v);
            };
#pragma unroll
            for (int j = 0; j < PF; ++j) ldaux(j, rq[j]);
            if (p.bias) ld12(p.bias + n0, bs);
            else {
#pragma unroll
                for (int i = 0; i < 12; ++i) bs[i] = 0.f;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                dump(s);
                __syncthreads();
#pragma unroll
                for (int q = 0; q < (s < 3 ? 4 : 2); ++q) {
                    const int it = 4 * s + q, m = m0 + trow(it);
                    float v[12], acur[12];
                    ld12(slab + srow(it) * SLD, v);
#pragma unroll
                    for (int i = 0; i < 12; ++i) acur[i] = rq[it % PF][i];
                    if (it + PF < NIT) ldaux(it + PF, rq[it % PF]);
#pragma unroll
                    for (int i = 0; i < 12; ++i) v[i] += bs[i];
                    if constexpr (DP) {
                        const float sc = p.rowscale[clampm(m)];
#pragma unroll
                        for (int i = 0; i < 12; ++i) v[i] *= sc;
                    }
                    if (m < mend) {
#pragma unroll
                        for (int i = 0; i < 12; ++i) v[i] += acur[i];
                        st12((float*)p.out + (size_t)m * p.ldo + n0, v);
                    }
                }
                if (s < 3) __syncthreads();
            }
        } else if constexpr (EPI == ROW_BIAS_BF16 || EPI == ROW_RESID_F32) {
            // out[m, n0 .. n0+383] = acc + bias [+ residual row]; the residual of the NEXT row is in flight during this one
            constexpr bool HAS_AUX = EPI == ROW_RESID_F32;
            float bs[12], anext[12];
            auto ldaux = [&](int m, float (&v)[12]) {
                if constexpr (EPI == ROW_RESID_F32) ld12((const float*)p.aux + (size_t)m * p.ldaux + n0, v);
            };
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                dump(s);
                if (s == 0) {
                    if constexpr (HAS_AUX) ldaux(clampm(m0 + trow(0)), anext);
                    if (p.bias) ld12(p.bias + n0, bs);
                    else {
#pragma unroll
                        for (int i = 0; i < 12; ++i) bs[i] = 0.f;
                    }
                }
                __syncthreads();
                const int nq = s < 3 ? 4 : 2;
#pragma nounroll
                for (int q = 0; q < nq; ++q) {
                    const int it = 4 * s + q, m = m0 + trow(it);
                    float v[12], acur[12];
                    ld12(slab + srow(it) * SLD, v);
                    if constexpr (HAS_AUX) {
#pragma unroll
                        for (int i = 0; i < 12; ++i) acur[i] = anext[i];
                        if (it + 1 < NIT) ldaux(clampm(m0 + trow(it + 1)), anext);
                    }
#pragma unroll
                    for (int i = 0; i < 12; ++i) v[i] += bs[i];
                    if constexpr (DP) {
                        const float sc = p.rowscale[clampm(m)];
#pragma unroll
                        for (int i = 0; i < 12; ++i) v[i] *= sc;
                    }
                    if (m < mend) {
                        if constexpr (EPI == ROW_BIAS_BF16) {
                            st12_bf16((bf16*)p.out + (size_t)m * p.ldo + n0, v);
                        } else {
#pragma unroll
                            for (int i = 0; i < 12; ++i) v[i] += acur[i];
                            st12((float*)p.out + (size_t)m * p.ldo + n0, v);
                        }
                    }
                }
                if (s < 3) __syncthreads();
            }
        } else if constexpr (EPI == ROW_LN_FWD && PF > 1) {
            float gm[12], bt[12], bs[12], rq[PF][12];
            auto ldres = [&](int it, float (&v)[12]) {
                ld12((const float*)p.aux + (size_t)clampm(m0 + trow(it)) * p.ldaux, v);
            };
#pragma unroll
            for (int j = 0; j < PF; ++j) ldres(j, rq[j]);
            ld12(p.gamma, gm);
            ld12(p.beta, bt);
            if (p.bias) ld12(p.bias, bs);
            else {
#pragma unroll
                for (int i = 0; i < 12; ++i) bs[i] = 0.f;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                dump(s);
                __syncthreads();
#pragma unroll
                for (int q = 0; q < (s < 3 ? 4 : 2); ++q) {
                    const int it = 4 * s + q, m = m0 + trow(it);
                    float v[12], rcur[12];
                    ld12(slab + srow(it) * SLD, v);
#pragma unroll
                    for (int i = 0; i < 12; ++i) rcur[i] = rq[it % PF][i];
                    if (it + PF < NIT) ldres(it + PF, rq[it % PF]);
                    float sum = 0.f;
                    if constexpr (DP) {
                        const float sc = p.rowscale[clampm(m)];
#pragma unroll
                        for (int i = 0; i < 12; ++i) { v[i] = (v[i] + bs[i]) * sc + rcur[i]; sum += v[i]; }
                    } else {
#pragma unroll
                        for (int i = 0; i < 12; ++i) { v[i] += bs[i] + rcur[i]; sum += v[i]; }
                    }
                    const float mu = half_sum(sum) * (1.0f / RBN);
                    float sq = 0.f;
#pragma unroll
                    for (int i = 0; i < 12; ++i) { const float d = v[i] - mu; sq += d * d; }
                    const float rs = rsqrtf(half_sum(sq) * (1.0f / RBN) + p.eps);
                    if (m < mend) {
                        st12((float*)p.out + (size_t)m * p.ldo, v);
#pragma unroll
                        for (int i = 0; i < 12; ++i) v[i] = (v[i] - mu) * rs * gm[i] + bt[i];
                        st12_bf16((bf16*)p.out2 + (size_t)m * p.ldo2, v);
                        if (l32 == 0) {
                            if (p.mean) p.mean[m] = mu;
                            if (p.rstd) p.rstd[m] = rs;
                        }
                    }
                }
                if (s < 3) __syncthreads();
            }
        } else if constexpr (EPI == ROW_LN_FWD) {
            float gm[12], bt[12], bs[12], rnext[12];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                dump(s);
                if (s == 0) {                                        // after the first dump: 48 accumulators are dead
                    ld12((const float*)p.aux + (size_t)clampm(m0 + trow(0)) * p.ldaux, rnext);
                    ld12(p.gamma, gm);
                    ld12(p.beta, bt);
                    if (p.bias) ld12(p.bias, bs);
                    else {
#pragma unroll
                        for (int i = 0; i < 12; ++i) bs[i] = 0.f;
                    }
                }
                __syncthreads();
                const int nq = s < 3 ? 4 : 2;
#pragma nounroll
                for (int q = 0; q < nq; ++q) {
                    const int it = 4 * s + q, m = m0 + trow(it);
                    float v[12], rcur[12];
                    ld12(slab + srow(it) * SLD, v);
#pragma unroll
                    for (int i = 0; i < 12; ++i) rcur[i] = rnext[i];
                    if (it + 1 < NIT)                                // next row's residual: issued before this row's stores
                        ld12((const float*)p.aux + (size_t)clampm(m0 + trow(it + 1)) * p.ldaux, rnext);
                    float sum = 0.f;
                    if constexpr (DP) {
                        const float sc = p.rowscale[clampm(m)];
#pragma unroll
                        for (int i = 0; i < 12; ++i) { v[i] = (v[i] + bs[i]) * sc + rcur[i]; sum += v[i]; }
                    } else {
#pragma unroll
                        for (int i = 0; i < 12; ++i) { v[i] += bs[i] + rcur[i]; sum += v[i]; }
                    }
                    const float mu = half_sum(sum) * (1.0f / RBN);
                    float sq = 0.f;
#pragma unroll
                    for (int i = 0; i < 12; ++i) { const float d = v[i] - mu; sq += d * d; }
                    const float rs = rsqrtf(half_sum(sq) * (1.0f / RBN) + p.eps);
                    if (m < mend) {
                        st12((float*)p.out + (size_t)m * p.ldo, v);
#pragma unroll
                        for (int i = 0; i < 12; ++i) v[i] = (v[i] - mu) * rs * gm[i] + bt[i];
                        st12_bf16((bf16*)p.out2 + (size_t)m * p.ldo2, v);
                        if (l32 == 0) {
                            if (p.mean) p.mean[m] = mu;
                            if (p.rstd) p.rstd[m] = rs;
                        }
                    }
                }
                if (s < 3) __syncthreads();                          // the slab is rewritten by the next dump
            }
        } else {  // ROW_LN_BWD: acc = dy
            float ag[12], ab[12];
            float* const gls = slab + SROWS * SLD;                   // gamma, kept in LDS (no registers to spare)
            for (int i = tid; i < RBN; i += 64 * NW) gls[i] = p.gamma[i];
#pragma unroll
            for (int i = 0; i < 12; ++i) { ag[i] = 0.f; ab[i] = 0.f; }
            float dnext[12], munext, rsnext;
            unsigned xnext[12];                                      // the prefetched LayerNorm-input row: 12 fp32 or (first six dwords) 12 bf16, as raw bits
            // bf16 form of the LayerNorm input: the forward's saved OUTPUT y = xhat gamma + beta (half the bytes of the fp32 row).
            // Used by the whole workgroup or not at all: every column needs a gamma the division does not blow up on.
            float* const igls = gls + RBN;                           // 1 / gamma
            float* const bls = igls + RBN;                           // - beta / gamma
            bool use16 = false;
            if constexpr (PF == 1) {
                if (p.xn16 != nullptr) {
                    int ok = 1;
                    for (int i = tid; i < RBN; i += 64 * NW) {
                        const float gv = p.gamma[i], bv = p.beta[i];
                        igls[i] = 1.0f / gv;
                        bls[i] = -bv / gv;
                        ok &= (fabsf(gv) >= 1e-3f && fabsf(bv) <= 64.f * fabsf(gv)) ? 1 : 0;
                    }
                    use16 = __syncthreads_and(ok) != 0;
                }
            }
            // the bf16 row travels in the first six dwords of the fp32 row's registers (no second prefetch set)
            auto ld12h = [&](const bf16* q, unsigned (&v)[12]) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const u32x2 t = *(const u32x2*)(q + 128 * i + 4 * l32);
                    v[2 * i] = t[0];
                    v[2 * i + 1] = t[1];
                }
            };
            auto ld12u = [&](const float* q, unsigned (&v)[12]) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const u32x4 t = *(const u32x4*)(q + 128 * i + 4 * l32);
                    v[4 * i] = t[0]; v[4 * i + 1] = t[1]; v[4 * i + 2] = t[2]; v[4 * i + 3] = t[3];
                }
            };
            auto ld_dres = [&](int m, float (&v)[12]) {
                if (!p.dres) return;
                if (p.dres_period > 0) {                             // compact: only every period-th row carries a gradient
                    const int qd = m / p.dres_period;
                    if (m - qd * p.dres_period == 0) ld12(p.dres + (size_t)qd * p.lddres, v);
                    else {
#pragma unroll
                        for (int i = 0; i < 12; ++i) v[i] = 0.f;
                    }
                } else {
                    ld12(p.dres + (size_t)m * p.lddres, v);
                }
            };
            if constexpr (PF > 1) {
                float xq[PF][12], dq[PF][12], muq[PF], rsq[PF];
                auto ldrow = [&](int it, int slot) {
                    const int m = clampm(m0 + trow(it));
                    ld12((const float*)p.aux + (size_t)m * p.ldaux, xq[slot]);
                    ld_dres(m, dq[slot]);
                    muq[slot] = p.mean[m];
                    rsq[slot] = p.rstd[m];
                };
#pragma unroll
                for (int j = 0; j < PF; ++j) ldrow(j, j);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    dump(s);
                    __syncthreads();
#pragma unroll
                    for (int q = 0; q < (s < 3 ? 4 : 2); ++q) {
                        const int it = 4 * s + q, m = m0 + trow(it);
                        float dy[12], xv[12], dr[12], gm[12];
                        ld12(slab + srow(it) * SLD, dy);
                        ld12(gls, gm);
                        const float mu = muq[it % PF], rs = rsq[it % PF];
#pragma unroll
                        for (int i = 0; i < 12; ++i) { xv[i] = xq[it % PF][i]; dr[i] = p.dres ? dq[it % PF][i] : 0.f; }
                        if (it + PF < NIT) ldrow(it + PF, it % PF);
                        const bool live = m < mend;
                        float c1 = 0.f, c2 = 0.f;
#pragma unroll
                        for (int i = 0; i < 12; ++i) {
                            xv[i] = (xv[i] - mu) * rs;               // xhat
                            if (live) { ag[i] += dy[i] * xv[i]; ab[i] += dy[i]; }
                            dy[i] *= gm[i];
                            c1 += dy[i];
                            c2 += dy[i] * xv[i];
                        }
                        c1 = half_sum(c1) * (1.0f / RBN);
                        c2 = half_sum(c2) * (1.0f / RBN);
#pragma unroll
                        for (int i = 0; i < 12; ++i) dy[i] = rs * (dy[i] - c1 - xv[i] * c2) + dr[i];
                        if (live) {
                            if (p.out) st12((float*)p.out + (size_t)m * p.ldo, dy);
                            if constexpr (DP) {
                                const float sc = p.rowscale[m];
#pragma unroll
                                for (int i = 0; i < 12; ++i) dy[i] *= sc;
                            }
                            if (p.out2) st12_bf16((bf16*)p.out2 + (size_t)m * p.ldo2, dy);
                        }
                    }
                    if (s < 3) __syncthreads();
                }
            } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                dump(s);
                if (s == 0) {
                    const int m = clampm(m0 + trow(0));
                    if (use16) ld12h(p.xn16 + (size_t)m * p.ldxn16, xnext);
                    else ld12u((const float*)p.aux + (size_t)m * p.ldaux, xnext);
                    ld_dres(m, dnext);
                    munext = p.mean[m];
                    rsnext = p.rstd[m];
                }
                __syncthreads();
                const int nq = s < 3 ? 4 : 2;
#pragma nounroll
                for (int q = 0; q < nq; ++q) {
                    const int it = 4 * s + q, m = m0 + trow(it);
                    float dy[12], xv[12], dr[12], gm[12];
                    ld12(slab + srow(it) * SLD, dy);
                    ld12(gls, gm);
                    const float mu = munext, rs = rsnext;
                    if (use16) {                                     // xhat = (y - beta) / gamma from the bf16 row
                        float hv[12];
#pragma unroll
                        for (int i = 0; i < 6; ++i) {                // bf16 -> f32: the 16 bits moved up
                            const unsigned u = xnext[i];
                            hv[2 * i] = __builtin_bit_cast(float, u << 16);
                            hv[2 * i + 1] = __builtin_bit_cast(float, u & 0xffff0000u);
                        }
                        ld12(igls, xv);                              // 1 / gamma, then - beta / gamma: xhat = y / gamma - beta / gamma
#pragma unroll
                        for (int i = 0; i < 12; ++i) hv[i] *= xv[i];
                        ld12(bls, xv);
#pragma unroll
                        for (int i = 0; i < 12; ++i) xv[i] = hv[i] + xv[i];
                    } else {
#pragma unroll
                        for (int i = 0; i < 12; ++i) xv[i] = (__builtin_bit_cast(float, xnext[i]) - mu) * rs;
                    }
#pragma unroll
                    for (int i = 0; i < 12; ++i) dr[i] = p.dres ? dnext[i] : 0.f;
                    if (it + 1 < NIT) {
                        const int mn = clampm(m0 + trow(it + 1));
                        if (use16) ld12h(p.xn16 + (size_t)mn * p.ldxn16, xnext);
                        else ld12u((const float*)p.aux + (size_t)mn * p.ldaux, xnext);
                        ld_dres(mn, dnext);
                        munext = p.mean[mn];
                        rsnext = p.rstd[mn];
                    }
                    const bool live = m < mend;
                    float c1 = 0.f, c2 = 0.f;
#pragma unroll
                    for (int i = 0; i < 12; ++i) {
                        if (live) { ag[i] += dy[i] * xv[i]; ab[i] += dy[i]; }
                        dy[i] *= gm[i];
                        c1 += dy[i];
                        c2 += dy[i] * xv[i];
                    }
                    c1 = half_sum(c1) * (1.0f / RBN);
                    c2 = half_sum(c2) * (1.0f / RBN);
#pragma unroll
                    for (int i = 0; i < 12; ++i) dy[i] = rs * (dy[i] - c1 - xv[i] * c2) + dr[i];
                    if (live) {
                        if (p.out) st12((float*)p.out + (size_t)m * p.ldo, dy);
                        if constexpr (DP) {
                            const float sc = p.rowscale[m];
#pragma unroll
                            for (int i = 0; i < 12; ++i) dy[i] *= sc;
                        }
                        if (p.out2) st12_bf16((bf16*)p.out2 + (size_t)m * p.ldo2, dy);
                    }
                }
                if (s < 3) __syncthreads();
            }
            }
            if (p.dgamma) {
                __syncthreads();                                     // the column sums go where the last slab was
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        scr[(hw * 2 + 0) * RBN + 128 * i + 4 * l32 + e] = ag[4 * i + e];
                        scr[(hw * 2 + 1) * RBN + 128 * i + 4 * l32 + e] = ab[4 * i + e];
                    }
                __syncthreads();
                for (int c = tid; c < 2 * RBN; c += 64 * NW) {
                    const int which = c / RBN, col = c - which * RBN;
                    float t = 0.f;
#pragma unroll
                    for (int h = 0; h < NHW; ++h) t += scr[(h * 2 + which) * RBN + col];
                    atomicAdd((which ? p.dbeta : p.dgamma) + col, t);
                }
            }
        }
    }
}

}  // namespace
