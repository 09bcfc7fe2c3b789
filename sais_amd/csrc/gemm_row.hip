// Row-owning bf16 MFMA GEMM for the ViT blocks (gfx950):  C[M, 384 g .. 384 g + 383] = A[M,K] . W[N,K]^T + epilogue,
// one 128-row x 384-column output tile per 256-thread workgroup.  With N = D = 384 a workgroup owns WHOLE rows of the
// output, which is what lets LayerNorm live in the epilogue:
//
//   LN_FWD  (proj / fc2 of Block.forward, vision_transformer.py:107-113):
//           x_out = A.W^T + bias + x ;  xn = LayerNorm(x_out) (eps 1e-6, the NEXT norm of the residual stream) ;
//           mean / rstd saved for backward.  Replaces gemm_nt<resid_f32> + ln_fwd_kernel.
//   LN_BWD  (dX of fc1 / qkv followed by autograd of norm2 / norm1):
//           dy = A.W^T ; dx = dres + rstd (dy g - mean(dy g) - xhat mean(dy g xhat)) ; dgamma += sum dy xhat ;
//           dbeta += sum dy.  Replaces gemm_nt<bias_bf16> + ln_bwd_kernel (the bf16 dy round trip disappears).
//   plain epilogues (bias / x aux / GELU + GELU' / + residual) for the other ViT GEMMs whose N is a multiple of 384.
//
// Structure.  4 waves as 2 (M) x 2 (N): a wave owns 64 rows x 192 columns = 4 x 12 MFMA 16x16x32 tiles = 192 fp32
// accumulator VGPRs (<= 256 VGPRs: two waves per SIMD, i.e. TWO workgroups per CU, so one workgroup's HBM-bound
// epilogue runs under the other's K loop).  LDS (80 KiB): A tile 128 x 64 k in two slots, W streamed as 128-column
// chunks (16 KiB) through a three-slot ring.  A K-step is three sub-steps (chunk c = 0,1,2: 32 MFMAs per wave each,
// A fragments re-read from LDS); per K-step 16 KiB of A + 48 KiB of W enter LDS for 2*128*384*64 flop = 98 flop per
// LDS-DMA byte (the 128x128 kernel: 65).  All staging is LDS-DMA (global_load_lds_dwordx4) with the XOR swizzle and
// the weight-row permutation on the SOURCE address; waits are counted (vmcnt is in-order): W runs two sub-steps
// ahead, A (first-touch HBM data) three.
//
// Operands are swapped in the MFMA (weights as "A", activations as "B") so a lane owns, per 16-row sub-tile mt,
// ONE output row (m = 16 mt + lane&15) and, per chunk c, 16 contiguous columns (n = 128 c + 64 wc + 16 (lane>>4) ..):
// loads / stores are 64-B (fp32) or 32-B (bf16) runs per lane, row sums are 48 in-register adds + 2 shuffles + one
// LDS exchange between the two N-waves, column sums (dgamma, dbeta) are DPP row reductions + LDS atomics.
#include "common.hpp"
#include "../../include/sais_hip.h"

namespace {

constexpr int RBM = 128, RBN = 384, RBK = 64;
constexpr int RTILE = 128 * 64 * 2;                 // 16 KiB: 128 rows x 64 k bf16
constexpr int ROW_LDS = 5 * RTILE;                  // A x 2, W x 3

enum { ROW_BIAS_BF16 = 0, ROW_MUL_BF16, ROW_GELU_GRAD_BF16, ROW_RESID_F32, ROW_LN_FWD, ROW_LN_BWD };

struct RowParams {
    const bf16* A; const bf16* W;
    int lda, ldw, M, N, K;
    const float* bias;              // [N] or null
    void* out; int ldo;             // bf16 out (BIAS / MUL / GELU) | f32 x_out (RESID, LN_FWD) | f32 dx (LN_BWD)
    void* out2; int ldo2;           // bf16: gelu' (GELU_GRAD) | xn (LN_FWD) | dx (LN_BWD)
    const void* aux; int ldaux;     // bf16 multiplier (MUL) | f32 residual (RESID, LN_FWD) | f32 LN input x (LN_BWD)
    const float* gamma; const float* beta; float eps;
    float* mean; float* rstd;       // LN_FWD: out (nullable) | LN_BWD: in
    const float* dres; int lddres;  // LN_BWD: residual-stream gradient added to dx (may alias out)
    float* dgamma; float* dbeta;    // LN_BWD: +=
};

DEVINL float sum4(const f32x4& v) { return (v[0] + v[1]) + (v[2] + v[3]); }

DEVINL void store16_bf16(bf16* o, const float (&z)[16]) {
    bf16x8 lo, hi;
#pragma unroll
    for (int i = 0; i < 8; ++i) { lo[i] = (bf16)z[i]; hi[i] = (bf16)z[8 + i]; }
    *(bf16x8*)o = lo;
    *(bf16x8*)(o + 8) = hi;
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_row_kernel(RowParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1, g = lane >> 4, li = lane & 15;
    const int ngrp = p.N / RBN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (tile % ngrp) * RBN, m0 = (tile / ngrp) * RBM;

    // wave w issues pieces 4w..4w+3 (8 LDS rows each) of every 128-row tile
    const int sub = lane >> 3, spos = lane & 7, schunk = spos ^ sub;
    // per-lane BYTE offsets (32-bit) from the wave-uniform bases: the LDS-DMA address is SGPR base + VGPR offset, which
    // keeps 8 instead of 16 address registers live next to the 192 accumulators
    unsigned aoff[4], woff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 8 * (4 * wid + j) + sub;
        int m = m0 + r;
        m = m < p.M ? m : p.M - 1;                                   // clamp: rows >= M are never stored
        aoff[j] = ((unsigned)m * (unsigned)p.lda + schunk * 8) * 2u;
        woff[j] = ((unsigned)(n0 + perm_row(r)) * (unsigned)p.ldw + schunk * 8) * 2u;
    }
    char* const sA = smem;
    char* const sW = smem + 2 * RTILE;
    const char* const Ab = (const char*)p.A;
    const char* const Wb = (const char*)p.W;
    const size_t wchunk = (size_t)128 * p.ldw * 2;                   // bytes between two 128-column chunks of W
    auto issue_a = [&](int kt) {
        char* s = sA + (kt & 1) * RTILE + (4 * wid) * 1024;
        const char* b = Ab + (size_t)kt * (RBK * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(b + aoff[j], s + j * 1024);
    };
    auto issue_w = [&](int kt, int c) {
        char* s = sW + c * RTILE + (4 * wid) * 1024;
        const char* b = Wb + c * wchunk + (size_t)kt * (RBK * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(b + woff[j], s + j * 1024);
    };

    f32x4 acc[4][12];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 12; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

#define ROW_COMPUTE(C)                                                                          \
    {                                                                                           \
        const char* sa = sA + (kt & 1) * RTILE;                                                 \
        const char* sb = sW + (C) * RTILE;                                                      \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                      \
            bf16x8 fa[4], fb[4];                                                                \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                     \
                fa[t] = *(const bf16x8*)(sa + swz(wr * 64 + t * 16 + li, ks * 4 + g));          \
                fb[t] = *(const bf16x8*)(sb + swz(wc * 64 + t * 16 + li, ks * 4 + g));          \
            }                                                                                   \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                    \
                _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                \
                    acc[mt][4 * (C) + nt] = mfma16(fb[nt], fa[mt], acc[mt][4 * (C) + nt]);      \
        }                                                                                       \
    }
#define ROW_WAIT(MORE, NMORE)                                                                   \
    if (more) asm volatile("s_waitcnt vmcnt(" #MORE ") lgkmcnt(0)" ::: "memory");               \
    else asm volatile("s_waitcnt vmcnt(" #NMORE ") lgkmcnt(0)" ::: "memory");                   \
    __builtin_amdgcn_s_barrier();

    const int nk = p.K / RBK;
    issue_a(0);
    issue_w(0, 0);
    issue_w(0, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                 // A(0), W(0,0) landed; W(0,1) may fly
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        // sub-step 0: needs A(kt), W(kt,0).  Issue W(kt,2) then A(kt+1) (A last: it may stay in flight longest)
        issue_w(kt, 2);
        if (more) issue_a(kt + 1);
        ROW_COMPUTE(0)
        ROW_WAIT(8, 4)                                               // W(kt,1) landed; W(kt,2) [+ A(kt+1)] in flight
        // sub-step 1
        if (more) issue_w(kt + 1, 0);
        ROW_COMPUTE(1)
        ROW_WAIT(8, 0)                                               // W(kt,2) landed; A(kt+1), W(kt+1,0) in flight
        // sub-step 2
        if (more) issue_w(kt + 1, 1);
        ROW_COMPUTE(2)
        ROW_WAIT(4, 0)                                               // A(kt+1), W(kt+1,0) landed; W(kt+1,1) in flight
    }
#undef ROW_COMPUTE
#undef ROW_WAIT

    // ------------------------------------------------------------------------------------------- epilogues
    // lane: rows m0 + wr*64 + mt*16 + li (mt = 0..3); per chunk c the 16 columns n0 + 128 c + 64 wc + 16 g + (4 nt + e)
    const int rloc = wr * 64 + li;                                   // + 16 mt: row inside the tile
    const int cbase = n0 + 64 * wc + 16 * g;                         // + 128 c
    float* const red = (float*)smem;                                 // the operand slots are free after the last barrier

    if constexpr (EPI == ROW_BIAS_BF16 || EPI == ROW_MUL_BF16 || EPI == ROW_GELU_GRAD_BF16 || EPI == ROW_RESID_F32) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = m0 + rloc + 16 * mt;
            if (m >= p.M) continue;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int n = cbase + 128 * c;
                float y[16];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 b = p.bias ? *(const f32x4*)(p.bias + n + 4 * i) : f32x4{0, 0, 0, 0};
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[4 * i + e] = acc[mt][4 * c + i][e] + b[e];
                }
                if constexpr (EPI == ROW_BIAS_BF16) {
                    store16_bf16((bf16*)p.out + (size_t)m * p.ldo + n, y);
                } else if constexpr (EPI == ROW_MUL_BF16) {
                    const bf16* u = (const bf16*)p.aux + (size_t)m * p.ldaux + n;
                    const bf16x8 u0 = *(const bf16x8*)u, u1 = *(const bf16x8*)(u + 8);
#pragma unroll
                    for (int i = 0; i < 8; ++i) { y[i] *= (float)u0[i]; y[8 + i] *= (float)u1[i]; }
                    store16_bf16((bf16*)p.out + (size_t)m * p.ldo + n, y);
                } else if constexpr (EPI == ROW_GELU_GRAD_BF16) {
                    float d[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) gelu_and_grad(y[i], y[i], d[i]);
                    store16_bf16((bf16*)p.out2 + (size_t)m * p.ldo2 + n, d);
                    store16_bf16((bf16*)p.out + (size_t)m * p.ldo + n, y);
                } else {
                    const float* r = (const float*)p.aux + (size_t)m * p.ldaux + n;
                    float* o = (float*)p.out + (size_t)m * p.ldo + n;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x4 rr = *(const f32x4*)(r + 4 * i);
                        *(f32x4*)(o + 4 * i) = f32x4{y[4 * i] + rr[0], y[4 * i + 1] + rr[1], y[4 * i + 2] + rr[2], y[4 * i + 3] + rr[3]};
                    }
                }
            }
        }
    } else if constexpr (EPI == ROW_LN_FWD) {
        // x_out = acc + bias + residual (kept in the accumulators), then LayerNorm over the 384 columns of each row
        float mu[4], rs[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            int m = m0 + rloc + 16 * mt;
            m = m < p.M ? m : p.M - 1;
            const float* rp = (const float*)p.aux + (size_t)m * p.ldaux + cbase;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 v = acc[mt][4 * c + i] + *(const f32x4*)(rp + 128 * c + 4 * i);
                    if (p.bias) v += *(const f32x4*)(p.bias + cbase + 128 * c + 4 * i);
                    acc[mt][4 * c + i] = v;
                    s += sum4(v);
                }
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            mu[mt] = s;
        }
        if (g == 0) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) red[wc * 128 + rloc + 16 * mt] = mu[mt];
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            mu[mt] = (red[rloc + 16 * mt] + red[128 + rloc + 16 * mt]) * (1.0f / RBN);
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 12; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = acc[mt][j][e] - mu[mt]; q += d * d; }
            q += __shfl_xor(q, 16);
            q += __shfl_xor(q, 32);
            rs[mt] = q;
        }
        if (g == 0) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) red[256 + wc * 128 + rloc + 16 * mt] = rs[mt];
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = m0 + rloc + 16 * mt;
            rs[mt] = rsqrtf((red[256 + rloc + 16 * mt] + red[384 + rloc + 16 * mt]) * (1.0f / RBN) + p.eps);
            if (m >= p.M) continue;
            float* xo = (float*)p.out + (size_t)m * p.ldo + cbase;
            bf16* no = (bf16*)p.out2 + (size_t)m * p.ldo2 + cbase;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float y[16];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 v = acc[mt][4 * c + i];
                    *(f32x4*)(xo + 128 * c + 4 * i) = v;
                    const f32x4 gm = *(const f32x4*)(p.gamma + cbase + 128 * c + 4 * i);
                    const f32x4 bt = *(const f32x4*)(p.beta + cbase + 128 * c + 4 * i);
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[4 * i + e] = (v[e] - mu[mt]) * rs[mt] * gm[e] + bt[e];
                }
                store16_bf16(no + 128 * c, y);
            }
            if (wc == 0 && g == 0) {
                if (p.mean) p.mean[m] = mu[mt];
                if (p.rstd) p.rstd[m] = rs[mt];
            }
        }
    } else {  // ROW_LN_BWD: acc = dy
        // LDS after the K loop: red[0..511] row sums (c1 | c2, per N-wave), rowst[512..767] mean | rstd of the tile's
        // rows, csum[768..1535] dgamma | dbeta partial column sums
        float* const rowst = red + 512;
        float* const csum = red + 768;
        for (int i = tid; i < 2 * RBN; i += 256) csum[i] = 0.f;
        if (tid < 128) {
            int m = m0 + tid;
            m = m < p.M ? m : p.M - 1;
            rowst[tid] = p.mean[m];
            rowst[128 + tid] = p.rstd[m];
        }
        __syncthreads();
        // phase A: c1 = mean(dy g), c2 = mean(dy g xhat) per row
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            int m = m0 + rloc + 16 * mt;
            m = m < p.M ? m : p.M - 1;
            const float mu = rowst[rloc + 16 * mt], rs = rowst[128 + rloc + 16 * mt];
            const float* xp = (const float*)p.aux + (size_t)m * p.ldaux + cbase;
            float sa = 0.f, sb = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 xv = *(const f32x4*)(xp + 128 * c + 4 * i);
                    const f32x4 gm = *(const f32x4*)(p.gamma + cbase + 128 * c + 4 * i);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gy = acc[mt][4 * c + i][e] * gm[e];
                        sa += gy;
                        sb += gy * ((xv[e] - mu) * rs);
                    }
                }
            sa += __shfl_xor(sa, 16); sa += __shfl_xor(sa, 32);
            sb += __shfl_xor(sb, 16); sb += __shfl_xor(sb, 32);
            if (g == 0) {
                red[wc * 128 + rloc + 16 * mt] = sa;
                red[256 + wc * 128 + rloc + 16 * mt] = sb;
            }
        }
        __syncthreads();
        // phase B, 8 columns at a time (register budget: 192 accumulators stay live): dx, stores, column sums.
        // x is read a second time here (L2 / MALL-resident: the same workgroup read it a few microseconds ago).
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int col = cbase + 128 * c + 8 * h;
                const f32x4 gm0 = *(const f32x4*)(p.gamma + col), gm1 = *(const f32x4*)(p.gamma + col + 4);
                float dg[8], db[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { dg[k] = 0.f; db[k] = 0.f; }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int mr = m0 + rloc + 16 * mt;
                    const bool live = mr < p.M;
                    const int m = live ? mr : p.M - 1;
                    const float mu = rowst[rloc + 16 * mt], rs = rowst[128 + rloc + 16 * mt];
                    const float c1 = (red[rloc + 16 * mt] + red[128 + rloc + 16 * mt]) * (1.0f / RBN);
                    const float c2 = (red[256 + rloc + 16 * mt] + red[384 + rloc + 16 * mt]) * (1.0f / RBN);
                    const float* xp = (const float*)p.aux + (size_t)m * p.ldaux + col;
                    const f32x4 x0 = *(const f32x4*)xp, x1 = *(const f32x4*)(xp + 4);
                    f32x4 d0 = f32x4{0, 0, 0, 0}, d1 = f32x4{0, 0, 0, 0};
                    if (p.dres) {
                        const float* dp = p.dres + (size_t)m * p.lddres + col;
                        d0 = *(const f32x4*)dp;
                        d1 = *(const f32x4*)(dp + 4);
                    }
                    float dx[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float dy = acc[mt][4 * c + 2 * h + (e >> 2)][e & 3];
                        const float xv = e < 4 ? x0[e & 3] : x1[e & 3];
                        const float gm = e < 4 ? gm0[e & 3] : gm1[e & 3];
                        const float dr = e < 4 ? d0[e & 3] : d1[e & 3];
                        const float xh = (xv - mu) * rs;
                        dx[e] = rs * (dy * gm - c1 - xh * c2) + dr;
                        if (live) { dg[e] += dy * xh; db[e] += dy; }
                    }
                    if (live) {
                        if (p.out) {
                            float* o = (float*)p.out + (size_t)m * p.ldo + col;
                            *(f32x4*)o = f32x4{dx[0], dx[1], dx[2], dx[3]};
                            *(f32x4*)(o + 4) = f32x4{dx[4], dx[5], dx[6], dx[7]};
                        }
                        if (p.out2) {
                            bf16x8 o16;
#pragma unroll
                            for (int e = 0; e < 8; ++e) o16[e] = (bf16)dx[e];
                            *(bf16x8*)((bf16*)p.out2 + (size_t)m * p.ldo2 + col) = o16;
                        }
                    }
                }
                if (p.dgamma) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) { dg[k] = row16_sum(dg[k]); db[k] = row16_sum(db[k]); }
                    if (li == 0) {
                        float* cs = csum + (col - n0);
#pragma unroll
                        for (int k = 0; k < 8; ++k) { atomicAdd(cs + k, dg[k]); atomicAdd(cs + RBN + k, db[k]); }
                    }
                }
            }
        if (p.dgamma) {
            __syncthreads();
            for (int i = tid; i < 2 * RBN; i += 256) atomicAdd((i < RBN ? p.dgamma + i : p.dbeta + (i - RBN)), csum[i]);
        }
    }
}

template <int EPI>
int launch_row(const RowParams& p, void* stream) {
    static thread_local bool set = false;
    if (!set) {
        if (hipFuncSetAttribute((const void*)gemm_nt_row_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                ROW_LDS) != hipSuccess)
            return SAIS_ERR_LAUNCH;
        set = true;
    }
    const int grid = (p.N / RBN) * ((p.M + RBM - 1) / RBM);
    hipLaunchKernelGGL(gemm_nt_row_kernel<EPI>, dim3(grid), dim3(256), ROW_LDS, (hipStream_t)stream, p);
    return sais_check_launch();
}

}  // namespace

// plain epilogues: called by sais_gemm_nt (gemm.hip) for the large-M ViT GEMMs whose N is a multiple of 384
extern "C" int sais_gemm_nt_row_(const SaisGemm* g, void* stream) {
    RowParams p{};
    p.A = (const bf16*)g->A; p.W = (const bf16*)g->B; p.lda = g->lda; p.ldw = g->ldb;
    p.M = g->M; p.N = g->N; p.K = g->K; p.bias = g->bias;
    p.out = g->out; p.ldo = g->ldo; p.out2 = g->out2; p.ldo2 = g->ldo2; p.aux = g->aux; p.ldaux = g->ldaux;
    switch (g->epilogue) {
        case SAIS_EPI_BIAS_BF16: return launch_row<ROW_BIAS_BF16>(p, stream);
        case SAIS_EPI_MUL_BF16: return launch_row<ROW_MUL_BF16>(p, stream);
        case SAIS_EPI_BIAS_GELU_GRAD_BF16: return launch_row<ROW_GELU_GRAD_BF16>(p, stream);
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
    p.dres = g->dres; p.lddres = g->lddres; p.dgamma = g->dgamma; p.dbeta = g->dbeta;
    return launch_row<ROW_LN_BWD>(p, stream);
}
