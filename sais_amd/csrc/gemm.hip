// bf16 MFMA GEMMs for the SAIS hot path (gfx950).
//
//  sais_gemm_nt : C[M,N] = A[M,K] . B[N,K]^T  (+ fused epilogue)  — every nn.Linear forward
//                 (vision_transformer.py:59-65,80-92; prepare_model.py:74-81,416) and, fed with the
//                 pre-transposed weight, every dX = dY . W.
//  sais_gemm_tn : dW[N1,N2] += P[M,N1]^T . Q[M,N2],  db[N1] += colsum(P) — every weight/bias gradient.
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4 MFMA
// 16x16x32 tiles, 64 fp32 accumulator VGPRs), BK = 64, two LDS stages (64 KiB), one barrier per K-step,
// global->register->LDS staging issued before the MFMAs of the current step.
// LDS image: 128-B rows, 16-B chunk index XOR (row & 7)  -> conflict-free ds_read_b128 fragment reads.
// Operands are swapped in the MFMA (weights as "A", activations as "B") and weight rows are permuted
// while staging so that every lane ends up with 16 CONTIGUOUS output columns of one output row:
// epilogue stores are 32-B (bf16) / 64-B (fp32) per lane, a full 128-B line per row per wave.
#include <stdlib.h>
#include "common.hpp"
#include "../../include/sais_hip.h"
#include "gemm_nt_epi.hpp"
#include "philox.hpp"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;      // 16 KiB per operand per stage


// Epilogue in two phases.  vmcnt is in-order and counts stores on CDNA4, so a load issued after a store cannot be
// consumed before that store has been acknowledged: phase A issues EVERY load a lane needs (bias once, the
// residual / pre-activation rows of all four 16-row sub-tiles), phase B only does arithmetic and stores.
struct EpiAux {
    f32x4 r[4][4];        // f32 aux (residual / position rows): 16 columns x 4 sub-tiles
    bf16x8 u[4][2];       // bf16 aux (pre-activation)
    u32x4 q[4];           // one-byte GELU' codes
};

template <int EPI>
DEVINL void epilogue_loads(const NtParams& p, int mbase, int li, int n, float (&b)[16], EpiAux& a) {
    if (p.bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 t = *(const f32x4*)(p.bias + n + 4 * i);
            b[4 * i] = t[0]; b[4 * i + 1] = t[1]; b[4 * i + 2] = t[2]; b[4 * i + 3] = t[3];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) b[i] = 0.f;
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        int m = mbase + mt * 16 + li;
        m = m < p.M ? m : p.M - 1;
        if constexpr (EPI == SAIS_EPI_BIAS_RESID_F32 || EPI == SAIS_EPI_PATCH_F32) {
            size_t row = m;
            if constexpr (EPI == SAIS_EPI_PATCH_F32) row = (m % p.grp_in) + p.grp_off;
            const float* r = (const float*)p.aux + row * p.ldaux + n;
#pragma unroll
            for (int i = 0; i < 4; ++i) a.r[mt][i] = *(const f32x4*)(r + 4 * i);
        } else if constexpr (EPI == SAIS_EPI_DGELU_BF16 || EPI == SAIS_EPI_DRELU_BF16 || EPI == SAIS_EPI_MUL_BF16) {
            const bf16* u = (const bf16*)p.aux + (size_t)m * p.ldaux + n;
            a.u[mt][0] = *(const bf16x8*)u;
            a.u[mt][1] = *(const bf16x8*)(u + 8);
        } else if constexpr (EPI == SAIS_EPI_MULQ_BF16) {
            a.q[mt] = *(const u32x4*)((const unsigned char*)p.aux + (size_t)m * p.ldaux + n);
        }
    }
}

template <int EPI>
DEVINL void epilogue(const NtParams& p, int m, int n, const float (&v)[16], const float (&b)[16], const EpiAux& a, int mt) {
    // one output row m, 16 contiguous columns n..n+15 (n multiple of 16)
    float y[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) y[i] = v[i] + b[i];

    auto store_bf16 = [&](void* base, int ld, const float (&z)[16]) {
        bf16x8 lo, hi;
#pragma unroll
        for (int i = 0; i < 8; ++i) { lo[i] = (bf16)z[i]; hi[i] = (bf16)z[8 + i]; }
        bf16* o = (bf16*)base + (size_t)m * ld + n;
        *(bf16x8*)o = lo;
        *(bf16x8*)(o + 8) = hi;
    };
    auto store_f32 = [&](void* base, int ld, size_t row, const float (&z)[16]) {
        float* o = (float*)base + row * ld + n;
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4*)(o + 4 * i) = f32x4{z[4 * i], z[4 * i + 1], z[4 * i + 2], z[4 * i + 3]};
    };

    if constexpr (EPI == SAIS_EPI_BIAS_BF16) {
        store_bf16(p.out, p.ldo, y);
    } else if constexpr (EPI == SAIS_EPI_BIAS_RELU_BF16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) y[i] = fmaxf(y[i], 0.f);
        store_bf16(p.out, p.ldo, y);
    } else if constexpr (EPI == SAIS_EPI_BIAS_F32) {
        store_f32(p.out, p.ldo, m, y);
    } else if constexpr (EPI == SAIS_EPI_BIAS_RESID_F32 || EPI == SAIS_EPI_PATCH_F32) {
        if constexpr (EPI == SAIS_EPI_BIAS_RESID_F32) {
            if (p.rowscale) {                                   // DropPath: residual + s_m (acc + bias)
                const float sc = p.rowscale[m];
#pragma unroll
                for (int i = 0; i < 16; ++i) y[i] *= sc;
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) y[i] += a.r[mt][i >> 2][i & 3];
        size_t orow = m;
        if constexpr (EPI == SAIS_EPI_PATCH_F32) orow = (size_t)(m / p.grp_in) * p.grp_out + (m % p.grp_in) + p.grp_off;
        store_f32(p.out, p.ldo, orow, y);
        if constexpr (EPI == SAIS_EPI_BIAS_RESID_F32)
            if (p.out2) store_bf16(p.out2, p.ldo2, y);
    } else if constexpr (EPI == SAIS_EPI_BIAS_GELU_BF16) {
        if (p.out2) store_bf16(p.out2, p.ldo2, y);          // pre-activation u (training)
        gelu_erf_n(y);
        store_bf16(p.out, p.ldo, y);
    } else if constexpr (EPI == SAIS_EPI_BIAS_GELU_GRAD_BF16) {
        float d[16];
        gelu_and_grad_n(y, d);
        store_bf16(p.out2, p.ldo2, d);
        store_bf16(p.out, p.ldo, y);
    } else if constexpr (EPI == SAIS_EPI_BIAS_GELU_GRADQ_BF16) {
        float d[16];
        gelu_and_grad_n(y, d);
        *(u32x4*)((unsigned char*)p.out2 + (size_t)m * p.ldo2 + n) =
            u32x4{gq8_pack4(d[0], d[1], d[2], d[3]), gq8_pack4(d[4], d[5], d[6], d[7]), gq8_pack4(d[8], d[9], d[10], d[11]),
                  gq8_pack4(d[12], d[13], d[14], d[15])};
        store_bf16(p.out, p.ldo, y);
    } else if constexpr (EPI == SAIS_EPI_MUL_BF16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) y[i] *= (float)a.u[mt][i >> 3][i & 7];
        store_bf16(p.out, p.ldo, y);
    } else if constexpr (EPI == SAIS_EPI_MULQ_BF16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) y[i] *= gq8_decode(a.q[mt][i >> 2], i & 3);
        store_bf16(p.out, p.ldo, y);
    } else if constexpr (EPI == SAIS_EPI_DGELU_BF16) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            f32x2 g;
            dgelu_erf2(f32x2{(float)a.u[mt][i >> 3][i & 7], (float)a.u[mt][i >> 3][(i & 7) + 1]}, g);
            y[i] *= g.x, y[i + 1] *= g.y;
        }
        store_bf16(p.out, p.ldo, y);
    } else if constexpr (EPI == SAIS_EPI_DRELU_BF16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) y[i] = (float)a.u[mt][i >> 3][i & 7] > 0.f ? y[i] : 0.f;
        store_bf16(p.out, p.ldo, y);
    }
}

// Staging is LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write): one wave-instruction fills
// 1 KiB = 8 LDS rows of 128 B linearly, so the XOR swizzle (and the weight-row permutation) is applied to the
// per-lane SOURCE address: LDS slot (row r, position c') receives global chunk c' ^ (r & 7).
template <int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(NtParams p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1, g = lane >> 4, li = lane & 15;
    const int ntn = p.N / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (tile % ntn) * BN, m0 = (tile / ntn) * BM;

    // wave w issues pieces 4w..4w+3 of each operand tile; piece q = LDS rows 8q..8q+7
    const int sub = lane >> 3, spos = lane & 7, schunk = spos ^ sub;
    const bf16* asrc[4]; const bf16* bsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 8 * (4 * wid + j) + sub;
        int m = m0 + r;
        m = m < p.M ? m : p.M - 1;                                   // clamp: rows >= M are never stored
        asrc[j] = p.A + (size_t)m * p.lda + schunk * 8;
        bsrc[j] = p.B + (size_t)(n0 + perm_row(r)) * p.ldb + schunk * 8;
    }
    auto issue = [&](int stage, int k0) {
        char* s = smem + stage * 2 * TILE_BYTES + (4 * wid) * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            glds16(asrc[j] + k0, s + j * 1024);
            glds16(bsrc[j] + k0, s + TILE_BYTES + j * 1024);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    int nk = p.K / BK, kbase = 0;
    if constexpr (EPI == SAIS_EPI_RAW_SLABS_F32) {                   // split-K: slice blockIdx.y of the K range
        const int per = (nk + (int)gridDim.y - 1) / (int)gridDim.y;
        kbase = blockIdx.y * per;
        nk = min(per, nk - kbase);
        if (nk <= 0) nk = 0;
    }
    if (nk > 0) issue(0, kbase * BK);
    __syncthreads();
    // the epilogue's own loads (bias, residual / pre-activation rows: first-touch HBM data) are issued before the MFMAs
    // of the LAST K-step, so their latency runs under that step instead of in front of the stores
    float bias[16];
    EpiAux aux;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue(cur ^ 1, (kbase + kt + 1) * BK);
        const char* sa = smem + cur * 2 * TILE_BYTES;
        const char* sb = sa + TILE_BYTES;
        if constexpr (EPI != SAIS_EPI_RAW_SLABS_F32)
            if (kt == nk - 1) epilogue_loads<EPI>(p, m0 + wr * 64, li, n0 + wc * 64 + 16 * g, bias, aux);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fa[t] = *(const bf16x8*)(sa + swz(wr * 64 + t * 16 + li, ks * 4 + g));
                fb[t] = *(const bf16x8*)(sb + swz(wc * 64 + t * 16 + li, ks * 4 + g));
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16(fb[nt], fa[mt], acc[mt][nt]);
        }
        __syncthreads();                      // drains this wave's LDS-DMA (vmcnt(0)) and fences the buffer swap
    }

    // lane holds, for row m = m0 + wr*64 + mt*16 + li, columns n0 + wc*64 + 16 g + (4 nt + r)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        int m = m0 + wr * 64 + mt * 16 + li;
        if (m >= p.M) continue;
        float v[16];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * nt + r] = acc[mt][nt][r];
        if constexpr (EPI == SAIS_EPI_RAW_SLABS_F32) {
            float* o = (float*)p.out + ((size_t)blockIdx.y * p.M + m) * p.ldo + n0 + wc * 64 + 16 * g;
#pragma unroll
            for (int i = 0; i < 4; ++i) *(f32x4*)(o + 4 * i) = f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
        } else {
            epilogue<EPI>(p, m, n0 + wc * 64 + 16 * g, v, bias, aux, mt);
        }
    }
}

// y = sum_z slabs[z] + bias; y *= rowscale[m]; y += aux; -> f32 and / or bf16 (fixed summation order: deterministic)
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* ws, int ks, int M, int N, int lds, const float* bias,
                                                            const float* rowscale, const float* aux, int ldaux, float* out32,
                                                            int ldo32, bf16* out16, int ldo16) {
    const int n4 = N >> 2;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < (long)M * n4; i += (long)gridDim.x * 256) {
        const int m = (int)(i / n4), n = (int)(i - (long)m * n4) * 4;
        f32x4 y = *(const f32x4*)(ws + (size_t)m * lds + n);
        for (int z = 1; z < ks; ++z) y += *(const f32x4*)(ws + ((size_t)z * M + m) * lds + n);
        if (bias) y += *(const f32x4*)(bias + n);
        if (rowscale) y *= rowscale[m];
        if (aux) y += *(const f32x4*)(aux + (size_t)m * ldaux + n);
        if (out32) *(f32x4*)(out32 + (size_t)m * ldo32 + n) = y;
        if (out16) {
            bf16x4 o;
            o[0] = (bf16)y[0]; o[1] = (bf16)y[1]; o[2] = (bf16)y[2]; o[3] = (bf16)y[3];
            *(bf16x4*)(out16 + (size_t)m * ldo16 + n) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Persistent form of the eight-wave kernel: a workgroup walks tiles b, b + G, ... and issues the first LDS-DMA loads of
// its NEXT tile (A'(0), W'(0), A'(1)) before the epilogue of the current one, so the per-tile prologue (the first
// K-tile's round trip, ~19 % of a K = 384 tile) runs under the epilogue's arithmetic and stores.  vmcnt is in-order
// and counts stores: the wait that follows the epilogue allows exactly the stores it issued (+ the two A'(1) pieces)
// to stay outstanding, which is only known for full tiles, so a ragged tile waits for everything.
// SAIS_NT_STAMP (debug builds only, tools/nt_stamp.py): lane 0 of every wave of workgroup 0 records the shader clock at the
// phase boundaries of its THIRD tile; sais_debug_nt_stamps() copies the table out.
#ifdef SAIS_NT_STAMP
__device__ unsigned long long g_nt_stamps[8][16];
#define NTSTAMP(i) do { if (blockIdx.x == 0 && titer == 2 && lane == 0) g_nt_stamps[wid][i] = __builtin_readcyclecounter(); } while (0)
#else
#define NTSTAMP(i) do { } while (0)
#endif
// SAIS_NT_ABL (timing ablations, results are WRONG when set; LABNOTES R4.4): 1 no K loop (no operand loads, no MFMAs),
// 2 no epilogue (no epilogue loads, arithmetic, stores), 4 / 8 role split — even / odd workgroups (4) or the lower / upper
// 256 workgroups (8) run ONLY the K loops resp. ONLY the epilogues of their tiles: do the two phases overlap on a CU at all?
#ifndef SAIS_NT_ABL
#define SAIS_NT_ABL 0
#endif
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_w8p_kernel(NtParams p, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // A ring: 3 x 16 KiB, then W: 2 x 16 KiB
    CLK_STAMP(EPI == SAIS_EPI_BIAS_GELU_GRAD_BF16 ? 1 : EPI == SAIS_EPI_MUL_BF16 ? 2 : 0);
    [[maybe_unused]] const int abl_role = (SAIS_NT_ABL & 4) ? (blockIdx.x & 1) : (SAIS_NT_ABL & 8) ? ((blockIdx.x >> 8) & 1) : -1;
    const bool do_k = !(SAIS_NT_ABL & 1) && abl_role != 1, do_e = !(SAIS_NT_ABL & 2) && abl_role != 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3, g = lane >> 4, li = lane & 15;
    const int ntn = p.N / BN;
    const int sub = lane >> 3, spos = lane & 7, schunk = spos ^ sub;
    const bf16* asrc[2]; const bf16* bsrc[2];
    auto set_tile = [&](int v, int& m0, int& n0) {
        const int tile = xcd_remap(v, ntiles);
        n0 = (tile % ntn) * BN; m0 = (tile / ntn) * BM;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = 8 * (2 * wid + j) + sub;
            int m = m0 + r;
            m = m < p.M ? m : p.M - 1;                               // clamp: rows >= M are never stored
            asrc[j] = p.A + (size_t)m * p.lda + schunk * 8;
            bsrc[j] = p.B + (size_t)(n0 + perm_row32(r)) * p.ldb + schunk * 8;
        }
    };
    char* const sW = smem + 3 * TILE_BYTES;
    auto issue_a = [&](int kt) {
        char* s = smem + (kt % 3) * TILE_BYTES + (2 * wid) * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(asrc[j] + kt * BK, s + j * 1024);
    };
    auto issue_w = [&](int kt) {
        char* s = sW + (kt & 1) * TILE_BYTES + (2 * wid) * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(bsrc[j] + kt * BK, s + j * 1024);
    };
    const int nk = p.K / BK;
    // store instructions one wave issues in a full tile's epilogue
    constexpr int SROW = (EPI == SAIS_EPI_BIAS_F32) ? 2 : (EPI == SAIS_EPI_BIAS_RESID_F32) ? 2 : (EPI == SAIS_EPI_PATCH_F32) ? 2
                       : (EPI == SAIS_EPI_BIAS_GELU_GRAD_BF16 || EPI == SAIS_EPI_BIAS_GELU_GRADQ_BF16) ? 2 : 1;
    const int nstores = 4 * (SROW + ((EPI == SAIS_EPI_BIAS_RESID_F32 || EPI == SAIS_EPI_BIAS_GELU_BF16) && p.out2 ? 1 : 0));

    // A-operand prefetch into L2 (K = 384, N >= 1024).  The ntn column tiles of a row tile run side by side on one XCD and
    // all of them wait for the same first-touch fetch of the A rows; once the epilogues of the other CUs keep HBM busy
    // with writes that fetch takes several microseconds, far more than the two K-steps of lead the LDS ring gives
    // (measured: the GELU + GELU' GEMM takes 145 us, 123 us with an L2-resident A).  So while a workgroup is in the
    // epilogue of tile i, wave 0 touches its 1/ntn share of the cache lines of tile i+1's A rows that the in-loop loads
    // would only ask for later (k >= 128: lines 2-5 of every 768-B row; lines 0-1 are being fetched by A'(0), A'(1)):
    // one global_load_dword, 44-60 active lanes, result never used.  It is the youngest load at the tile switch (one
    // more allowed in that wait) and is retired by the counted wait of K-step 0.  fc1 + GELU' 129 -> 124 us, qkv 68.5 ->
    // 66.5 us inside the step.  (Prefetching two tiles ahead is no better; the same trick on the bf16 aux tile of the
    // MUL epilogue is WORSE, 128 vs 118 us: that operand is bandwidth, not latency.)
    const bool do_pf = nk == 6 && ntn >= 8 && ntn <= 16 && wid == 0;
    unsigned pf_keep = 0;
    auto prefetch_a = [&](int pm0, int tcol) {
        const int per = (BM + ntn - 1) / ntn;                      // rows per sharing workgroup
        const int row = tcol * per + (lane >> 2);
        if ((lane >> 2) < per && row < BM) {
            int m = pm0 + row;
            m = m < p.M ? m : p.M - 1;
            const bf16* q = p.A + (size_t)m * p.lda + (2 + (lane & 3)) * 64;
            asm volatile("global_load_dword %0, %1, off" : "=v"(pf_keep) : "v"(q) : "memory");
        }
    };
    int v = blockIdx.x, m0, n0;
    if (v >= ntiles) return;
    set_tile(v, m0, n0);
    if (do_k) {
        issue_a(0);
        issue_w(0);
        if (nk > 1) issue_a(1);
    }
    if (nk > 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    [[maybe_unused]] int titer = -1;
    for (;;) {
        ++titer;
        NTSTAMP(0);
        f32x4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        float bias[8];
        EpiAux8 aux;
        // The barrier-paced K loop outranks the epilogue of the OTHER workgroup on this CU (round 3): the two share the SIMDs'
        // issue ports, and at equal priority the epilogue's VALU stream (GELU: ~19 slots per element) delays the waves the
        // whole workgroup waits for at the next barrier.  fc1 + GELU' 124 -> 117-120 us, dX fc2 114 -> 108-112, qkv 67 -> 63-65
        // on two boxes; priorities 1, 2 and 3 measure the same.
        __builtin_amdgcn_s_setprio(2);
        for (int kt = 0; kt < (do_k ? nk : 0); ++kt) {
            if (kt + 1 < nk) issue_w(kt + 1);
            if (kt + 2 < nk) issue_a(kt + 2);
            const char* sa = smem + (kt % 3) * TILE_BYTES;
            const char* sb = sW + (kt & 1) * TILE_BYTES;
            if (kt == nk - 1 && do_e) epilogue_loads8<EPI>(p, m0 + wr * 64, li, n0 + wc * 32 + 8 * g, bias, aux);
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
            if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
            else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // last step: only the epilogue's loads are out
            NTSTAMP(1 + 2 * kt);                                         // MFMAs issued, operands of the next step awaited
            __builtin_amdgcn_s_barrier();
            NTSTAMP(2 + 2 * kt);
        }
        __builtin_amdgcn_s_setprio(0);
        // the next tile's first loads go out before this tile's epilogue
        const int cm0 = m0, cn0 = n0;
        const int nv = v + gridDim.x;
        const bool more = nv < ntiles;
        asm volatile("" ::"v"(pf_keep));                               // the previous prefetch has been retired by now
        if (more) {
            set_tile(nv, m0, n0);
            if (do_k) {
                issue_a(0);
                issue_w(0);
                if (nk > 1) issue_a(1);
                if (do_pf) prefetch_a(m0, xcd_remap(nv, ntiles) % ntn);
            }
        }
        if (SAIS_NT_ABL) {
            if (!do_k) epilogue_loads8<EPI>(p, cm0 + wr * 64, li, cn0 + wc * 32 + 8 * g, bias, aux);
            if (!do_e) {                                               // keep the MFMAs alive
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) asm volatile("" ::"v"(acc[mt][nt]));
            }
        }
#pragma unroll
        for (int mt = 0; mt < (do_e ? 4 : 0); ++mt) {
            const int m = cm0 + wr * 64 + mt * 16 + li;
            if (m >= p.M) continue;
            float vv[8];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) vv[4 * nt + r] = acc[mt][nt][r];
            epilogue8<EPI>(p, m, cn0 + wc * 32 + 8 * g, vv, bias, aux, mt);
        }
        NTSTAMP(13);                                                     // epilogue arithmetic done, stores issued
        if (!more) break;
        v = nv;
        // A'(0) and W'(0) must have landed; the two A'(1) pieces and this epilogue's stores may stay in flight
        const int allow = (cm0 + BM <= p.M && nk > 1) ? nstores + 2 + (do_pf ? 1 : 0) : 0;
        if (allow == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (allow == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else if (allow == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (allow == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
        else if (allow == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        else if (allow == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        NTSTAMP(14);                                                     // the next tile's first operands have landed
        __builtin_amdgcn_s_barrier();
        NTSTAMP(15);
    }
}

// ---- EXPERIMENT RECORDS (built only with -DSAIS_EXPERIMENTAL=1: tools/build_variant.sh exp -DSAIS_EXPERIMENTAL=1) ----------------
// Three other organisations of the persistent K = 384 GEMM, each correct, each measured slower than gemm_nt_w8p_kernel
// (LABNOTES R5.1, R5.2, R5.6).  The default library does not contain them; the switches SAIS_NT_W8R / _W16 / _W4 are then ignored.
#ifndef SAIS_EXPERIMENTAL
#define SAIS_EXPERIMENTAL 0
#endif
#if SAIS_EXPERIMENTAL
// ---------------------------------------------------------------------------------------------
// W in registers (round 5, LABNOTES R5.6; K = 384 only, the K loop fully unrolled).  The stamps of R5.2 show a K-step of the
// kernel above taking 1 200-1 700 cycles of a wave's time for 256 cycles of its MFMAs: W(kt + 1) is requested at the top of step
// kt and awaited at its end (two-slot W ring: one exposed L2 round trip of LDS-DMA per step), and a third W slot does not fit two
// workgroups per CU.  Here W does not pass through LDS at all: a wave loads the MFMA fragments of ITS 32 weight columns
// straight from global memory (L2-resident: 16-B per lane, four loads per step) TWO steps ahead into a rotating triple of register
// sets (+ 32 VGPRs), and the 80 KiB of LDS become a five-slot A ring with four steps of lead.  Per step and wave: 2 LDS-DMA issues
// instead of 4, 8 fragment reads instead of 12, no W to publish at the barrier.  vmcnt is one in-order counter, so the issue
// order inside a step is W first, then A, and the counted wait at the end of step kt leaves exactly A(kt + 3), W(kt + 2), A(kt + 4)
// in flight.
template <int EPI>
__global__ __launch_bounds__(512, 4) void gemm_nt_w8r_kernel(NtParams p, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // A ring: 5 x 16 KiB
    constexpr int NK = 6, NA = 5;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3, g = lane >> 4, li = lane & 15;
    const int ntn = p.N / BN;
    const int sub = lane >> 3, spos = lane & 7, schunk = spos ^ sub;
    const bf16* asrc[2]; const bf16* wsrc[2];
    auto set_tile = [&](int v, int& m0, int& n0) {
        const int tile = xcd_remap(v, ntiles);
        n0 = (tile % ntn) * BN; m0 = (tile / ntn) * BM;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = 8 * (2 * wid + j) + sub;
            int m = m0 + r;
            m = m < p.M ? m : p.M - 1;
            asrc[j] = p.A + (size_t)m * p.lda + schunk * 8;
            // fragment tile j of this wave's 32 columns: MFMA row li <-> weight row 8 (li >> 2) + 4 j + (li & 3) (perm_row32: a
            // lane then owns 8 contiguous output columns), k = 8 g .. 8 g + 7 of a 32-deep half-step
            wsrc[j] = p.B + (size_t)(n0 + wc * 32 + 8 * (li >> 2) + 4 * j + (li & 3)) * p.ldb + 8 * g;
        }
    };
    auto issue_a = [&](int kt) {
        char* s = smem + (kt % NA) * TILE_BYTES + (2 * wid) * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(asrc[j] + kt * BK, s + j * 1024);
    };
    bf16x8 wf[3][2][2];                                               // [set][k-half][column tile]
    // The W loads are inline asm: the compiler's own s_waitcnt insertion does not see the counted waits below and put vmcnt(0)
    // in front of the MFMAs of steps 0 and 3 (first version: 201 instead of 141 us).  Invisible to it, they are ordered by hand:
    // every counted wait names the register set it makes valid as an in / out operand, so no MFMA that reads the set can be
    // scheduled above the wait.
    auto load_w = [&](int kt, bf16x8 (&dst)[2][2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[ks][nt]) : "v"(wsrc[nt] + kt * BK + ks * 32) : "memory");
    };
#define W8R_WAIT(N, SET) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)"                                                 \
                                      : "+v"(wf[SET][0][0]), "+v"(wf[SET][0][1]), "+v"(wf[SET][1][0]), "+v"(wf[SET][1][1]) :: "memory")
    constexpr int SROW = (EPI == SAIS_EPI_BIAS_F32) ? 2 : (EPI == SAIS_EPI_BIAS_RESID_F32) ? 2 : (EPI == SAIS_EPI_PATCH_F32) ? 2
                       : (EPI == SAIS_EPI_BIAS_GELU_GRAD_BF16 || EPI == SAIS_EPI_BIAS_GELU_GRADQ_BF16) ? 2 : 1;
    const int nstores = 4 * (SROW + ((EPI == SAIS_EPI_BIAS_RESID_F32 || EPI == SAIS_EPI_BIAS_GELU_BF16) && p.out2 ? 1 : 0));
    auto prologue = [&] {                                             // W(0), A(0), W(1), A(1), A(2), A(3): the order the waits count on
        load_w(0, wf[0]);
        issue_a(0);
        load_w(1, wf[1]);
        issue_a(1);
        issue_a(2);
        issue_a(3);
        __builtin_amdgcn_sched_barrier(0);
    };
    int v = blockIdx.x, m0, n0;
    if (v >= ntiles) return;
    set_tile(v, m0, n0);
    prologue();
    W8R_WAIT(10, 0);                                                  // W(0) and A(0) are in; W(1), A(1..3) may be in flight
    __builtin_amdgcn_s_barrier();
    int carry = 0;                           // stores of the previous tile's epilogue that may still be in flight at step 0
    for (;;) {
        f32x4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        float bias[8];
        EpiAux8 aux;
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            if (kt + 2 < NK) load_w(kt + 2, wf[(kt + 2) % 3]);
            if (kt + 4 < NK) issue_a(kt + 4);
            __builtin_amdgcn_sched_barrier(0);
            const char* sa = smem + (kt % NA) * TILE_BYTES;
            if (kt == NK - 1) epilogue_loads8<EPI>(p, m0 + wr * 64, li, n0 + wc * 32 + 8 * g, bias, aux);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 fa[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) fa[t] = *(const bf16x8*)(sa + swz(wr * 64 + t * 16 + li, ks * 4 + g));
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma16(wf[kt % 3][ks][nt], fa[mt], acc[mt][nt]);
            }
            // in flight after this point (oldest first): [kt = 0: A(2), A(3), the previous tile's stores] W(kt+2), A(kt+4) and,
            // before them, A(kt+3) — everything older, i.e. W(kt+1) and A(kt+1), has to be in
            if (kt == 0) {                                            // makes W(1) = set 1 valid
                const int allow = 10 + carry;
                if (allow == 10) W8R_WAIT(10, 1);
                else if (allow == 14) W8R_WAIT(14, 1);
                else if (allow == 18) W8R_WAIT(18, 1);
                else W8R_WAIT(4, 1);
            } else if (kt == 1) W8R_WAIT(8, 2);
            else if (kt == 2) W8R_WAIT(6, 0);
            else if (kt == 3) W8R_WAIT(4, 1);
            else if (kt == 4) W8R_WAIT(0, 2);
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // last step: only the epilogue's loads are out
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_setprio(0);
        const int cm0 = m0, cn0 = n0;
        const int nv = v + gridDim.x;
        const bool more = nv < ntiles;
        if (more) {
            set_tile(nv, m0, n0);
            prologue();
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = cm0 + wr * 64 + mt * 16 + li;
            if (m >= p.M) continue;
            float vv[8];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) vv[4 * nt + r] = acc[mt][nt][r];
            epilogue8<EPI>(p, m, cn0 + wc * 32 + 8 * g, vv, bias, aux, mt);
        }
        if (!more) break;
        v = nv;
        // W'(0) and A'(0) must be in; W'(1), A'(1..3) and this epilogue's stores may stay in flight (a ragged tile issues fewer
        // stores than counted: wait for everything)
        const int allow = (cm0 + BM <= p.M) ? nstores + 10 : 0;
        carry = allow ? nstores : 0;
        if (allow == 14) W8R_WAIT(14, 0);                             // makes W'(0) = set 0 valid
        else if (allow == 18) W8R_WAIT(18, 0);
        else { W8R_WAIT(0, 0); carry = 0; }
        __builtin_amdgcn_s_barrier();
    }
#undef W8R_WAIT
}

// ---------------------------------------------------------------------------------------------
// Two eight-wave groups of ONE 1024-thread workgroup in ENFORCED anti-phase (round 5, LABNOTES R5.2).  Measured on the kernel
// above (SAIS_NT_GRID, SAIS_NT_ABL builds): K loops alone 62.5 us with two workgroups per CU and 81 us with one, epilogues alone
// 58 us (HBM-bound) either way, the whole kernel 141 us = MORE than their sum — the two workgroups of a CU run the same program
// from the same start, so both are in their K loops together (each slowed by the other) and in their epilogues together (the
// store path and HBM saturated, the matrix pipe idle), and a tile's first-touch A rows are fetched while every CU writes.
// Here the two tile pipelines of a CU are two wave groups of one workgroup that share every s_barrier: group 0 runs the nk
// K-steps of its tile while group 1 runs the epilogue of ITS previous tile in nk slices (one 16-row sub-tile per interval,
// then idle intervals), and vice versa.  At any moment eight waves feed the matrix pipe and eight drain to HBM, the next
// tile's first operands are requested a whole half-period ahead, and HBM sees a steady write stream.
// Same tile, LDS image (2 x 80 KiB), epilogues and registers as the eight-wave kernel.  Needs nk >= 5.
#ifdef SAIS_NT_STAMP
__device__ unsigned long long g_nt16_stamps[16][36];
#endif
template <int EPI>
__global__ __launch_bounds__(1024) void gemm_nt_w16_kernel(NtParams p, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w16 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = w16 >> 3, wid = w16 & 7;
    char* const smem = smem_all + grp * (5 * TILE_BYTES);            // this group's A ring (3 x 16 KiB) + W ring (2 x 16 KiB)
    const int wr = wid >> 2, wc = wid & 3, g = lane >> 4, li = lane & 15;
    const int ntn = p.N / BN;
    const int sub = lane >> 3, spos = lane & 7, schunk = spos ^ sub;
    const bf16* asrc[2]; const bf16* bsrc[2];
    auto set_tile = [&](int v, int& m0, int& n0) {
        const int tile = xcd_remap(v, ntiles);
        n0 = (tile % ntn) * BN; m0 = (tile / ntn) * BM;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = 8 * (2 * wid + j) + sub;
            int m = m0 + r;
            m = m < p.M ? m : p.M - 1;
            asrc[j] = p.A + (size_t)m * p.lda + schunk * 8;
            bsrc[j] = p.B + (size_t)(n0 + perm_row32(r)) * p.ldb + schunk * 8;
        }
    };
    char* const sW = smem + 3 * TILE_BYTES;
    auto issue_a = [&](int kt) {
        char* s = smem + (kt % 3) * TILE_BYTES + (2 * wid) * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(asrc[j] + kt * BK, s + j * 1024);
    };
    auto issue_w = [&](int kt) {
        char* s = sW + (kt & 1) * TILE_BYTES + (2 * wid) * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(bsrc[j] + kt * BK, s + j * 1024);
    };
    const int nk = p.K / BK;
    constexpr int SROW = (EPI == SAIS_EPI_BIAS_F32) ? 2 : (EPI == SAIS_EPI_BIAS_RESID_F32) ? 2 : (EPI == SAIS_EPI_PATCH_F32) ? 2
                       : (EPI == SAIS_EPI_BIAS_GELU_GRAD_BF16 || EPI == SAIS_EPI_BIAS_GELU_GRADQ_BF16) ? 2 : 1;
    const int nstores = 4 * (SROW + ((EPI == SAIS_EPI_BIAS_RESID_F32 || EPI == SAIS_EPI_BIAS_GELU_BF16) && p.out2 ? 1 : 0));
    // virtual workgroup ids: group 0 = blockIdx.x, group 1 = blockIdx.x + gridDim.x (same XCD); both walk with stride 2 G
    const int G2 = 2 * (int)gridDim.x;
    auto count = [&](int v0) { return v0 < ntiles ? (ntiles - v0 + G2 - 1) / G2 : 0; };
    const int nA = count(blockIdx.x), nB = count(blockIdx.x + gridDim.x);
    const int mine = grp ? nB : nA;
    const int totA = 2 * nk * nA, totB = nB ? nk + 2 * nk * nB : 0;
    const int total = totA > totB ? totA : totB;                     // barriers every wave of the workgroup takes
    int done = 0;
#ifdef SAIS_NT_STAMP
    // lane 0 of every wave of workgroup 0 stamps the shader clock BEFORE and AFTER each of 18 consecutive barriers (from the
    // 24th on: both groups are in steady state): arrival and release times of every interval (tools/nt16_stamp.py)
    auto bar = [&] {
        const int k = done - 24;
        if (blockIdx.x == 0 && lane == 0 && k >= 0 && k < 18) g_nt16_stamps[w16][2 * k] = __builtin_readcyclecounter();
        __builtin_amdgcn_s_barrier();
        if (blockIdx.x == 0 && lane == 0 && k >= 0 && k < 18) g_nt16_stamps[w16][2 * k + 1] = __builtin_readcyclecounter();
        ++done;
    };
#else
    auto bar = [&] { __builtin_amdgcn_s_barrier(); ++done; };
#endif
    int v = blockIdx.x + grp * gridDim.x, m0 = 0, n0 = 0;
    if (mine > 0) {
        set_tile(v, m0, n0);
        issue_a(0);
        issue_w(0);
        issue_a(1);
    }
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1 && mine > 0)
        for (int i = 0; i < nk; ++i) bar();                          // group 1 runs half a period behind
    for (int t = 0; t < mine; ++t) {
        f32x4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        float bias[8];
        EpiAux8 aux;
        __builtin_amdgcn_s_setprio(2);
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) issue_w(kt + 1);
            if (kt + 2 < nk) issue_a(kt + 2);
            const char* sa = smem + (kt % 3) * TILE_BYTES;
            const char* sb = sW + (kt & 1) * TILE_BYTES;
            if (kt == nk - 1) epilogue_loads8<EPI>(p, m0 + wr * 64, li, n0 + wc * 32 + 8 * g, bias, aux);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 fa[4], fb[2];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) fa[tt] = *(const bf16x8*)(sa + swz(wr * 64 + tt * 16 + li, ks * 4 + g));
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) fb[tt] = *(const bf16x8*)(sb + swz(wc * 32 + tt * 16 + li, ks * 4 + g));
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma16(fb[nt], fa[mt], acc[mt][nt]);
            }
            if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
            else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // last step: only the epilogue's loads are out
            bar();
        }
        __builtin_amdgcn_s_setprio(0);
        // epilogue phase = nk intervals beside the OTHER group's K loop.  The next tile's first operands go out first: they
        // have the whole phase to arrive.
        const int cm0 = m0, cn0 = n0;
        const bool more = t + 1 < mine;
        if (more) {
            v += G2;
            set_tile(v, m0, n0);
            issue_a(0);
            issue_w(0);
            issue_a(1);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = cm0 + wr * 64 + mt * 16 + li;
            if (m < p.M) {
                float vv[8];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) vv[4 * nt + r] = acc[mt][nt][r];
                epilogue8<EPI>(p, m, cn0 + wc * 32 + 8 * g, vv, bias, aux, mt);
            }
            bar();
        }
        for (int i = 4; i < nk - 1; ++i) bar();
        // A'(0) and W'(0) must have landed before the phase's last barrier; the two A'(1) pieces and this epilogue's stores may
        // stay in flight (vmcnt is in-order: they are younger)
        const int allow = (more && cm0 + BM <= p.M) ? nstores + 2 : 0;
        if (allow == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (allow == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (allow == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bar();
    }
    while (done < total) bar();
}

// ---------------------------------------------------------------------------------------------
// Four workgroups per CU (round 5).  LABNOTES R4.4: a workgroup of the eight-wave kernel above is a latency CHAIN (K loop ->
// epilogue -> K loop; neither phase is slowed by what the CU's other workgroup does), so the launch takes tiles-per-workgroup x
// chain length and what shortens it is more chains per CU.  Same 128 x 128 tile, same fill bytes per flop, but FOUR waves of
// 64 x 64 (16 MFMAs per wave between barriers, as before; 8 instead of 12 fragment reads for them) and K in steps of 32:
// 8-KiB stages, A ring of three + W ring of two = 40 KiB per workgroup, <= 128 VGPRs -> four workgroups = four chains per CU,
// and no two waves of a workgroup share a SIMD (the barrier skew of the eight-wave form was the SIMD sibling).
// LDS image: 64-B rows; the 16-B chunk c of row r sits at position c ^ qmap(r), which makes the ds_read_b128 fragment reads
// conflict-free for the hardware's lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS).
constexpr int QK = 32;
constexpr int QTILE = 128 * QK * 2;              // 8 KiB per operand per stage
DEVINL int qmap(int r) { const int q = (r >> 2) & 3; return (((q ^ (q >> 1)) & 1) << 1) | (q >> 1); }
DEVINL int swz64(int row, int chunk) { return row * 64 + ((chunk ^ qmap(row)) << 4); }

template <int EPI>
__global__ __launch_bounds__(256, 4) void gemm_nt_w4q_kernel(NtParams p, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // A ring: 3 x 8 KiB, then W: 2 x 8 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1, g = lane >> 4, li = lane & 15;
    const int ntn = p.N / BN;
    const int srow = lane >> 2, spos = lane & 3;
    const bf16* asrc[2]; const bf16* bsrc[2];
    auto set_tile = [&](int v, int& m0, int& n0) {
        const int tile = xcd_remap(v, ntiles);
        n0 = (tile % ntn) * BN; m0 = (tile / ntn) * BM;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = 16 * (2 * wid + j) + srow;
            const int c = spos ^ qmap(r);
            int m = m0 + r;
            m = m < p.M ? m : p.M - 1;
            asrc[j] = p.A + (size_t)m * p.lda + c * 8;
            bsrc[j] = p.B + (size_t)(n0 + perm_row(r)) * p.ldb + c * 8;
        }
    };
    char* const sW = smem + 3 * QTILE;
    auto issue_a = [&](int kt) {
        char* s = smem + (kt % 3) * QTILE + (2 * wid) * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(asrc[j] + kt * QK, s + j * 1024);
    };
    auto issue_w = [&](int kt) {
        char* s = sW + (kt & 1) * QTILE + (2 * wid) * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(bsrc[j] + kt * QK, s + j * 1024);
    };
    const int nk = p.K / QK;
    constexpr bool LATE = EPI == SAIS_EPI_MUL_BF16 || EPI == SAIS_EPI_DGELU_BF16 || EPI == SAIS_EPI_DRELU_BF16;
    [[maybe_unused]] const int tk = nk >= 4 ? nk - 4 : 0;
    [[maybe_unused]] unsigned pf_keep = 0;
    constexpr int SROW = (EPI == SAIS_EPI_BIAS_F32 || EPI == SAIS_EPI_BIAS_RESID_F32 || EPI == SAIS_EPI_PATCH_F32) ? 4
                       : (EPI == SAIS_EPI_BIAS_GELU_GRAD_BF16) ? 4 : 2;
    const int nstores = 4 * (SROW + ((EPI == SAIS_EPI_BIAS_RESID_F32 || EPI == SAIS_EPI_BIAS_GELU_BF16) && p.out2 ? 2 : 0));
    int v = blockIdx.x, m0, n0;
    if (v >= ntiles) return;
    set_tile(v, m0, n0);
    issue_a(0);
    issue_w(0);
    issue_a(1);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (;;) {
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        float bias[16];
        EpiAux aux;
        __builtin_amdgcn_s_setprio(2);
        for (int kt = 0; kt < nk; ++kt) {
            if constexpr (LATE) {
                // the bf16 aux tile of this wave (64 rows x 128 B) is first-touch HBM data and there are no registers to hold
                // it during the K loop (64 accumulators + 32 fragment registers): one discarded dword per row pulls the 64
                // lines into L2 three steps early (retired by this step's counted wait), the real loads follow the loop
                if (kt == tk) {
                    int m = m0 + wr * 64 + lane;
                    m = m < p.M ? m : p.M - 1;
                    const bf16* q = (const bf16*)p.aux + (size_t)m * p.ldaux + n0 + wc * 64;
                    asm volatile("global_load_dword %0, %1, off" : "=v"(pf_keep) : "v"(q) : "memory");
                }
            }
            if (kt + 1 < nk) issue_w(kt + 1);
            if (kt + 2 < nk) issue_a(kt + 2);
            const char* sa = smem + (kt % 3) * QTILE;
            const char* sb = sW + (kt & 1) * QTILE;
            if constexpr (!LATE) {
                if (kt == nk - 1) epilogue_loads<EPI>(p, m0 + wr * 64, li, n0 + wc * 64 + 16 * g, bias, aux);
            }
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fa[t] = *(const bf16x8*)(sa + swz64(wr * 64 + t * 16 + li, g));
                fb[t] = *(const bf16x8*)(sb + swz64(wc * 64 + t * 16 + li, g));
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16(fb[nt], fa[mt], acc[mt][nt]);
            if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
            else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // last step: only the epilogue's loads are out
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_setprio(0);
        const int cm0 = m0, cn0 = n0;
        const int nv = v + gridDim.x;
        const bool more = nv < ntiles;
        if constexpr (LATE) {
            asm volatile("" ::"v"(pf_keep));
            epilogue_loads<EPI>(p, cm0 + wr * 64, li, cn0 + wc * 64 + 16 * g, bias, aux);
        }
        if (more) {
            set_tile(nv, m0, n0);
            issue_a(0);
            issue_w(0);
            issue_a(1);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = cm0 + wr * 64 + mt * 16 + li;
            if (m >= p.M) continue;
            float vv[16];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) vv[4 * nt + r] = acc[mt][nt][r];
            epilogue<EPI>(p, m, cn0 + wc * 64 + 16 * g, vv, bias, aux, mt);
        }
        if (!more) break;
        v = nv;
        // A'(0) and W'(0) must have landed; the two A'(1) pieces and this epilogue's stores may stay in flight
        const int allow = (cm0 + BM <= p.M) ? nstores + 2 : 0;
        if (allow == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (allow == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else if (allow == 26) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

#endif  // SAIS_EXPERIMENTAL

#ifdef SAIS_NT_STAMP
#if SAIS_EXPERIMENTAL
extern "C" int sais_debug_nt16_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_nt16_stamps), sizeof(unsigned long long) * 16 * 36) == hipSuccess ? 0 : -2;
}
#endif
extern "C" int sais_debug_nt_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_nt_stamps), sizeof(unsigned long long) * 8 * 16) == hipSuccess ? 0 : -2;
}
#endif

// ---------------------------------------------------------------------------------------------
// NT with fp32 operands at ~fp32 accuracy on the bf16 matrix cores ("bf16x3"): every operand is split
// while staging into hi = bf16(x), lo = bf16(x - hi) and the product is accumulated as
// a_hi b_hi + a_hi b_lo + a_lo b_hi (the dropped lo*lo term is ~2^-18 relative).  Used for the temporal
// encoder, whose activations feed the <=1e-3 logit parity bar directly and are tiny (M = clips*(T+1)),
// so 3x the MFMA work is irrelevant.  Single LDS stage (4 x 16 KiB), same swizzle / operand swap /
// 16-contiguous-columns-per-lane epilogue as the bf16 kernel.
template <int EPI>
DEVINL void epilogue_f32(const NtParams& p, int m, int n, const float (&v)[16]) {
    if (p.grp_in > 1) {
        // split-K: raw partial sums go to the workspace slab of this split; splitk_reduce_kernel applies the epilogue
        float* o = (float*)p.out2 + ((size_t)blockIdx.z * p.M + m) * p.N + n;
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4*)(o + 4 * i) = f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
        return;
    }
    // four columns at a time (the dropout draws would otherwise push the 64-accumulator kernel into scratch).
    // Train-mode dropout of the encoder layer, fused: relu -> dropout (FFN), dropout -> + residual (dropout1 / dropout2),
    // and in the backward drelu -> the same FFN mask.
    const bool dropping = p.p_drop > 0.f;
    const unsigned thr = drop_threshold(p.p_drop);
    const float inv = dropping ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    float* o = (float*)p.out + (size_t)m * p.ldo + n;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 t;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = 4 * q + j;
            t[j] = v[i] + (p.bias ? p.bias[n + i] : 0.f);
        }
        f32x4 keep = {1.f, 1.f, 1.f, 1.f};
        if (dropping) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                keep[j] = philox_keep(p.rng, p.site, (unsigned long long)m * p.N + n + 4 * q + j, thr) ? inv : 0.f;
        }
        if constexpr (EPI == SAIS_EPI_BIAS_RELU_F32) {
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = fmaxf(t[j], 0.f) * keep[j];
        } else if constexpr (EPI == SAIS_EPI_BIAS_RESID_F32) {
            const f32x4 r = *(const f32x4*)((const float*)p.aux + (size_t)m * p.ldaux + n + 4 * q);
            t = t * keep + r;
        } else if constexpr (EPI == SAIS_EPI_DRELU_F32) {
            const f32x4 u = *(const f32x4*)((const float*)p.aux + (size_t)m * p.ldaux + n + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = u[j] > 0.f ? t[j] * keep[j] : 0.f;
        }
        *(f32x4*)(o + 4 * q) = t;
    }
}

DEVINL void split8(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo) {
    bf16x8 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = (bf16)a[i]; l[i] = (bf16)(a[i] - (float)h[i]);
        h[4 + i] = (bf16)b[i]; l[4 + i] = (bf16)(b[i] - (float)h[4 + i]);
    }
    hi = __builtin_bit_cast(u32x4, h);
    lo = __builtin_bit_cast(u32x4, l);
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_nt_f32x3_kernel(NtParams p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];      // A_hi | A_lo | B_hi | B_lo
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, g = lane >> 4, li = lane & 15;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const float* A = (const float*)p.A;
    const float* B = (const float*)p.B;
    const int sc = tid & 7, sr = tid >> 3;
    f32x4 ra[4][2], rb[4][2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = sr + 32 * i, m = m0 + r;
            const float* pa = A + (size_t)(m < p.M ? m : 0) * p.lda + k0 + sc * 8;
            const float* pb = B + (size_t)(n0 + r) * p.ldb + k0 + sc * 8;
            bool ok = m < p.M;
            ra[i][0] = ok ? *(const f32x4*)pa : f32x4{0, 0, 0, 0};
            ra[i][1] = ok ? *(const f32x4*)(pa + 4) : f32x4{0, 0, 0, 0};
            rb[i][0] = *(const f32x4*)pb;
            rb[i][1] = *(const f32x4*)(pb + 4);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = sr + 32 * i;
            u32x4 hi, lo;
            split8(ra[i][0], ra[i][1], hi, lo);
            *(u32x4*)(smem + swz(r, sc)) = hi;
            *(u32x4*)(smem + TILE_BYTES + swz(r, sc)) = lo;
            split8(rb[i][0], rb[i][1], hi, lo);
            *(u32x4*)(smem + 2 * TILE_BYTES + swz(perm_row(r), sc)) = hi;
            *(u32x4*)(smem + 3 * TILE_BYTES + swz(perm_row(r), sc)) = lo;
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    // split-K: grp_in = number of K splits (gridDim.z); this workgroup owns K-tiles [kbeg, kbeg + nk)
    const int nk = p.K / BK / (p.grp_in > 1 ? p.grp_in : 1);
    const int kbeg = blockIdx.z * nk;
    gload(kbeg * BK);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                       // previous tile fully consumed
        lstore();
        __syncthreads();
        if (kt + 1 < nk) gload((kbeg + kt + 1) * BK);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                int oa = swz(wr * 64 + t * 16 + li, ks * 4 + g), ob = swz(wc * 64 + t * 16 + li, ks * 4 + g);
                ah[t] = *(const bf16x8*)(smem + oa);
                al[t] = *(const bf16x8*)(smem + TILE_BYTES + oa);
                bh[t] = *(const bf16x8*)(smem + 2 * TILE_BYTES + ob);
                bl[t] = *(const bf16x8*)(smem + 3 * TILE_BYTES + ob);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    f32x4 c = acc[mt][nt];
                    c = mfma16(bl[nt], ah[mt], c);
                    c = mfma16(bh[nt], al[mt], c);
                    c = mfma16(bh[nt], ah[mt], c);
                    acc[mt][nt] = c;
                }
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        int m = m0 + wr * 64 + mt * 16 + li;
        if (m >= p.M) continue;
        float v[16];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * nt + r] = acc[mt][nt][r];
        epilogue_f32<EPI>(p, m, n0 + wc * 64 + 16 * g, v);
    }
}

// ---------------------------------------------------------------------------------------------
// TN: dW[N1,N2] += sum_m P[m,N1] Q[m,N2].  Reduction index m is the SLOW dimension of both
// operands, so MFMA fragments (8 consecutive k per lane) are column reads of the row-major LDS
// tiles: ds_read_b64_tr_b16 (two per fragment).  LDS rows are padded 256 -> 288 B so the 8 rows a
// half-wave touches per read fall on distinct banks.  Split over M (gridDim.z) with fp32
// atomicAdd of the partial tiles; db via one extra MFMA column of ones in the n2-tile-0 blocks.
constexpr int TK = 64;                 // m rows per step
constexpr int TROW = 288;              // padded LDS row bytes (128 bf16 + 16 pad)
constexpr int TTILE = TK * TROW;       // 18 KiB

struct TnParams {
    const void* P; const void* Q; int ldp, ldq, M, N1, N2;
    float* dW; int ldw; float* db; int rows_per_split;
};

// 8 consecutive elements -> packed bf16x8 (f32 inputs are rounded to bf16 while staging)
DEVINL u32x4 load8_bf16(const bf16* p) { return *(const u32x4*)p; }
DEVINL u32x4 load8_bf16(const float* p) {
    f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
    bf16x8 v;
    v[0] = (bf16)a[0]; v[1] = (bf16)a[1]; v[2] = (bf16)a[2]; v[3] = (bf16)a[3];
    v[4] = (bf16)b[0]; v[5] = (bf16)b[1]; v[6] = (bf16)b[2]; v[7] = (bf16)b[3];
    return __builtin_bit_cast(u32x4, v);
}

// one 128x128 output tile over rows [mbeg, mend) of P / Q
// OWNED: the workgroup is the only writer of its output tile in this launch (one M-split), so the accumulation into dW / db
// is a plain read-add-write instead of 16 k atomics per tile (the few-row temporal dW GEMMs were atomics-bound: 37 -> 23 us)
// NP = P columns (= dW rows) per tile: 128, or 64 for the few-row temporal dW launches (twice the workgroups, half the
// read-add-write epilogue per workgroup)
template <typename T, bool OWNED = false, int NP = 128>
DEVINL void tn_tile(const TnParams& p, int n1_0, int n2_0, int mbeg, int mend, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, g = lane >> 4, li = lane & 15;

    // staging: tile = 64 rows x 128 cols bf16 = 64 x 16 chunks; thread -> chunk tid&15, rows tid>>4 + 16 i
    const int sc = tid & 15, sr = tid >> 4;
    u32x4 rp[4], rq[4];
    auto gload = [&](int mb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int m = mb + sr + 16 * i;
            bool ok = m < mend;
            rp[i] = (ok && sc < NP / 8) ? load8_bf16((const T*)p.P + (size_t)m * p.ldp + n1_0 + sc * 8) : u32x4{0, 0, 0, 0};
            rq[i] = ok ? load8_bf16((const T*)p.Q + (size_t)m * p.ldq + n2_0 + sc * 8) : u32x4{0, 0, 0, 0};
        }
    };
    auto lstore = [&](int stage) {
        char* s = smem + stage * 2 * TTILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int off = (sr + 16 * i) * TROW + sc * 16;
            *(u32x4*)(s + off) = rp[i];
            *(u32x4*)(s + TTILE + off) = rq[i];
        }
    };

    constexpr int PT = NP / 32;                    // 16-column P tiles per wave (the wave owns NP / 2 dW rows)
    f32x4 acc[PT][4];
    f32x4 accb[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        accb[i] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    }
    const bool do_bias = p.db != nullptr && n2_0 == 0 && wc == 0;
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (bf16)1.0f;

    // tr16 read address: lane-in-group = 4q + p supplies row q, cols c0 + 4p..4p+3 of the 4x16 block
    const int q4 = li >> 2, p4 = li & 3;
    const int nsteps = (mend - mbeg + TK - 1) / TK;
    gload(mbeg);
    lstore(0);
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const int cur = st & 1;
        if (st + 1 < nsteps) gload(mbeg + (st + 1) * TK);
        const char* sp = smem + cur * 2 * TTILE;
        const char* sq = sp + TTILE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fp[PT], fq[4];
            // k-slot (g, e) <-> m = 32 ks + 16 (e>>2) + 4 g + (e&3): a half-wave touches 8 CONSECUTIVE rows
            // per read (conflict-free with the 288-B row stride); P and Q use the same slot map.
            const int rbase = (ks * 32 + 4 * g + q4) * TROW + p4 * 8;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                int cq = (wc * 64 + t * 16) * 2;
                fq[t] = cat4(lds_read_tr16(sq + rbase + cq), lds_read_tr16(sq + rbase + 16 * TROW + cq));
            }
#pragma unroll
            for (int t = 0; t < PT; ++t) {
                int cp = (wr * (NP / 2) + t * 16) * 2;
                fp[t] = cat4(lds_read_tr16(sp + rbase + cp), lds_read_tr16(sp + rbase + 16 * TROW + cp));
            }
#pragma unroll
            for (int it = 0; it < PT; ++it)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) acc[it][jt] = mfma16(fp[it], fq[jt], acc[it][jt]);
            if (do_bias) {
#pragma unroll
                for (int it = 0; it < PT; ++it) accb[it] = mfma16(fp[it], ones, accb[it]);
            }
        }
        if (st + 1 < nsteps) lstore(cur ^ 1);
        __syncthreads();
    }
    // D[i = n1][j = n2]: lane holds n2 = tile + li, n1 = tile + 4g + r
#pragma unroll
    for (int it = 0; it < PT; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int n1 = n1_0 + wr * (NP / 2) + it * 16 + 4 * g + r;
            float* row = p.dW + (size_t)n1 * p.ldw + n2_0 + wc * 64 + li;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                if constexpr (OWNED) row[jt * 16] += acc[it][jt][r];
                else atomicAdd(row + jt * 16, acc[it][jt][r]);
            }
            if (do_bias && li == 0) {
                if constexpr (OWNED) p.db[n1] += accb[it][r];
                else atomicAdd(p.db + n1, accb[it][r]);
            }
        }
}

// LDS-DMA variant of tn_tile for bf16 operands when every M-split is a whole number of 64-row steps:
// unpadded 256-B rows, 32-B units XOR-swizzled by (row & 7) on the SOURCE address (a half-wave's transposed read
// touches 8 consecutive rows x 32 B -> 8 distinct units = all 64 banks), two 32-KiB stages.
DEVINL const char* tr_addr(const char* tile, int row, int col) {       // col multiple of 4
    return tile + row * 256 + ((((col >> 4) ^ (row & 7)) << 5) | ((col & 15) << 1));
}

DEVINL void tn_tile_dma(const TnParams& p, int n1_0, int n2_0, int mbeg, int mend, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1, g = lane >> 4, li = lane & 15;
    constexpr int T16 = 64 * 256;                                       // 16 KiB per operand per stage
    // pieces 4w..4w+3 of each operand: piece = 4 rows; lane -> row 4*piece + (lane>>4), position lane&15
    const bf16* psrc[4]; const bf16* qsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 4 * (4 * wid + j) + (lane >> 4);
        const int c = lane & 15, u = (c >> 1) ^ (r & 7);
        psrc[j] = (const bf16*)p.P + (size_t)(mbeg + r) * p.ldp + n1_0 + u * 16 + (c & 1) * 8;
        qsrc[j] = (const bf16*)p.Q + (size_t)(mbeg + r) * p.ldq + n2_0 + u * 16 + (c & 1) * 8;
    }
    auto issue = [&](int stage, int step) {
        char* s = smem + stage * 2 * T16 + (4 * wid) * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            glds16(psrc[j] + (size_t)step * 64 * p.ldp, s + j * 1024);
            glds16(qsrc[j] + (size_t)step * 64 * p.ldq, s + T16 + j * 1024);
        }
    };
    f32x4 acc[4][4];
    f32x4 accb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        accb[i] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    }
    const bool do_bias = p.db != nullptr && n2_0 == 0 && wc == 0;
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (bf16)1.0f;
    const int q4 = li >> 2, p4 = li & 3;
    const int nsteps = (mend - mbeg) / TK;
    issue(0, 0);
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const int cur = st & 1;
        if (st + 1 < nsteps) issue(cur ^ 1, st + 1);
        const char* sp = smem + cur * 2 * T16;
        const char* sq = sp + T16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fp[4], fq[4];
            const int row = ks * 32 + 4 * g + q4;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int cp = wr * 64 + t * 16 + 4 * p4, cq = wc * 64 + t * 16 + 4 * p4;
                fp[t] = cat4(lds_read_tr16(tr_addr(sp, row, cp)), lds_read_tr16(tr_addr(sp, row + 16, cp)));
                fq[t] = cat4(lds_read_tr16(tr_addr(sq, row, cq)), lds_read_tr16(tr_addr(sq, row + 16, cq)));
            }
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) acc[it][jt] = mfma16(fp[it], fq[jt], acc[it][jt]);
            if (do_bias) {
#pragma unroll
                for (int it = 0; it < 4; ++it) accb[it] = mfma16(fp[it], ones, accb[it]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int n1 = n1_0 + wr * 64 + it * 16 + 4 * g + r;
            float* row = p.dW + (size_t)n1 * p.ldw + n2_0 + wc * 64 + li;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) atomicAdd(row + jt * 16, acc[it][jt][r]);
            if (do_bias && li == 0) atomicAdd(p.db + n1, accb[it][r]);
        }
}

template <typename T>
__global__ __launch_bounds__(256) void gemm_tn_kernel(TnParams p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * TTILE];   // 72 KiB
    // 1-D grid, XCD-aware order with the M-split as the slow index: the (N1/128)*(N2/128) tiles of one split
    // run on ONE XCD back to back and share that split's P and Q row slabs through its L2 (the slabs are then
    // fetched from HBM once instead of once per tile).
    const int nt2 = p.N2 / 128, ntile = (p.N1 / 128) * nt2;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int split = wg / ntile, t12 = wg - split * ntile;
    const int mbeg = split * p.rows_per_split;
    const int mend = min(p.M, mbeg + p.rows_per_split);
    if (mbeg >= mend) return;
    tn_tile<T>(p, (t12 / nt2) * 128, (t12 % nt2) * 128, mbeg, mend, smem);
}

// Several weight-gradient GEMMs over the SAME M rows in one launch (the four nn.Linear of a ViT block): 108 tiles
// instead of 9-36, so 4 M-splits fill the chip where the per-GEMM launches needed 12-48, and the fp32 atomic
// traffic (64 KiB per workgroup) drops by the same factor.
struct TnGroup {
    TnParams item[SAIS_TN_MAX_ITEMS];
    int tile_end[SAIS_TN_MAX_ITEMS];          // prefix sums of tiles per item
    int nitems, ntiles;
};

__global__ __launch_bounds__(256) void gemm_tn_grouped_kernel(TnGroup gp) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * TTILE];
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int split = wg / gp.ntiles;
    int t = wg - split * gp.ntiles, it = 0;
    while (it + 1 < gp.nitems && t >= gp.tile_end[it]) ++it;
    if (it > 0) t -= gp.tile_end[it - 1];
    const TnParams& p = gp.item[it];
    const int nt2 = p.N2 / 128;
    const int mbeg = split * p.rows_per_split;
    const int mend = min(p.M, mbeg + p.rows_per_split);
    if (mbeg >= mend) return;
    if ((mend - mbeg) % TK == 0) tn_tile_dma(p, (t / nt2) * 128, (t % nt2) * 128, mbeg, mend, smem);
    else tn_tile<bf16>(p, (t / nt2) * 128, (t % nt2) * 128, mbeg, mend, smem);
}

// the same grouping for fp32 operands (rounded to bf16 while staging): the four dW of a temporal-encoder layer, M = a few
// hundred rows, where the launch count rather than the arithmetic is what costs
template <bool OWNED, int NP = 128>
__global__ __launch_bounds__(256) void gemm_tn_grouped_f32_kernel(TnGroup gp) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * TTILE];
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int split = wg / gp.ntiles;
    int t = wg - split * gp.ntiles, it = 0;
    while (it + 1 < gp.nitems && t >= gp.tile_end[it]) ++it;
    if (it > 0) t -= gp.tile_end[it - 1];
    const TnParams& p = gp.item[it];
    const int nt2 = p.N2 / 128;
    const int mbeg = split * p.rows_per_split;
    const int mend = min(p.M, mbeg + p.rows_per_split);
    if (mbeg >= mend) return;
    tn_tile<float, OWNED, NP>(p, (t / nt2) * NP, (t % nt2) * 128, mbeg, mend, smem);
}

// Wide variant of the grouped dW kernel: 128 (P columns) x 384 (Q columns) output tile per 512-thread workgroup
// (8 waves as 2 x 4, 64 x 96 per wave), used when every item has N2 % 384 == 0 (all four dW of a ViT block do).
// The 128x128 kernel above is paced by its global->LDS fill stream (ablation in LABNOTES.md 4.1: 2.8 GB of fills per
// launch, DMA-only 223 us vs 166 us of MFMA work); this tile needs a third fewer fill bytes per flop: 64 KiB per
// 64-row step (P 16 KiB + three 128-column blocks of Q) for 2 x 128 x 384 x 64 flop.  Two 64-KiB stages = 128 KiB of
// LDS, one workgroup per CU; 36 tiles x 7 M-splits = 252 workgroups fill the 256 CUs in one round.
constexpr int WQ = 384;
constexpr int WBLK = 64 * 256;                 // one 64-row x 128-column block, 16 KiB
constexpr int WSTAGE = 4 * WBLK;               // P block + 3 Q blocks

struct TnWideGroup {
    TnParams item[SAIS_TN_MAX_ITEMS];
    int tile_end[SAIS_TN_MAX_ITEMS];
    int nitems, ntiles;
};

// The wide dW kernel: 128 x 384 tile, 8 waves (2 x 4 of 64 x 96), global -> VGPR -> LDS staging (a plain vector load does
// not hold the wave the way an LDS-DMA issue does) with two tiles in flight in registers (a first-touch row slab comes
// from HBM, and one step is not enough to cover that latency), transposed fragment reads.
// Ping-pong schedule.  Round 1's version had all eight waves read fragments together, run their 48 MFMAs
// together and meet at one barrier per step, so the MFMA pipe of a SIMD idled while both of its waves were in the LDS
// phase (PMC: MFMA busy 39 %).  Here a step is four barrier intervals per wave,
//     R0: fragments of k-half 0 + first half of the next tile's LDS writes / global loads
//     M0: 24 MFMAs          R1: fragments of k-half 1 + second half of the writes / loads          M1: 24 MFMAs
// and the waves 4-7 (the SIMD partners of 0-3) run ONE INTERVAL BEHIND (one extra barrier before the loop, the other
// group takes it after): in every interval one wave of each SIMD owns the MFMA pipe while its partner is in the LDS.
// Hazards (intervals numbered globally; group A's step s is 4s..4s+3, group B's 4s+1..4s+4): tile s+1 is written into
// buffer (s+1)&1 during 4s..4s+3 and first read in 4s+4; the old contents (tile s-1) were last read in 4s-2 (A) and
// 4s-1 (B), and every R interval ends with lgkmcnt(0) BEFORE its barrier, so those reads have returned.
// (212 -> 197 us per block inside the step.  Measured and dropped on this kernel: one bias MFMA per wave instead of four
// on the wc = 0 waves, hand-counted vmcnt(12) instead of the compiler's vmcnt(7..4): no change either way — the kernel
// is paced by the global fill stream, LABNOTES.md 4.2.)
// SLAB (round 5): instead of 96 fp32 atomicAdd instructions per wave at the very end (7 M-splits x 7.1 MB = 49.8 MB of
// atomics that all 252 workgroups issue at the same moment; the chip retires ~1.3 TB/s of them), every workgroup stores its
// raw 128 x 384 partial tile ONCE, in register order (16 B per lane, 1 KiB per wave-instruction), into the slab of its split;
// tn_slab_finish_kernel sums the splits in a fixed order and adds the result to dW / db: deterministic gradients.
// NI (round 5, opt-in SAIS_TN_NI=2): barrier intervals per 64-row step.  4 = the schedule above.  2 = one LDS interval (the
// fragments of BOTH k-halves, the whole next tile's LDS writes, the loads of the tile after it) and one interval of 48 MFMAs per
// step: half the barriers — the bare MFMA + barrier skeleton of the 4-interval form already takes 130 of the kernel's 198 us —
// paid for with 80 instead of 40 fragment registers, which leaves room for ONE staging register set (a tile is requested one
// step before its LDS write instead of two).
template <bool SLAB, int NI = 4>
__global__ __launch_bounds__(512) void gemm_tn_pp_kernel(TnWideGroup gp, float* slabs) {
    extern __shared__ __attribute__((aligned(16))) char wsmem[];          // 2 x WSTAGE
    CLK_STAMP(3);
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int split = wg / gp.ntiles;
    int t = wg - split * gp.ntiles, it0 = 0;
    while (it0 + 1 < gp.nitems && t >= gp.tile_end[it0]) ++it0;
    if (it0 > 0) t -= gp.tile_end[it0 - 1];
    const TnParams& p = gp.item[it0];
    const int nt2 = p.N2 / WQ;
    const int n1_0 = (t / nt2) * 128, n2_0 = (t % nt2) * WQ;
    const int mbeg = split * p.rows_per_split;
    const int mend = min(p.M, mbeg + p.rows_per_split);
    if (mbeg >= mend) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3, g = lane >> 4, li = lane & 15;
    const int blk = wid >> 1;
    const bf16* src0 = blk == 0 ? (const bf16*)p.P + n1_0 : (const bf16*)p.Q + n2_0 + (blk - 1) * 128;
    const int ld = blk == 0 ? p.ldp : p.ldq;
    const bf16* pbase = src0 + (size_t)(mbeg + 32 * (wid & 1) + (lane >> 4)) * ld + (lane & 15) * 8;
    const int nsteps = (mend - mbeg) / TK;
    u32x4 stg[2][8];
    auto gload4 = [&](int step, u32x4 (&dst)[8], int h) {
        step = step < nsteps ? step : nsteps - 1;
#pragma unroll
        for (int j = 4 * h; j < 4 * h + 4; ++j) dst[j] = *(const u32x4*)(pbase + (size_t)(step * TK + 4 * j) * ld);
    };
    auto lwrite4 = [&](int stage, const u32x4 (&src)[8], int h) {
        char* s = wsmem + stage * WSTAGE + blk * WBLK;
#pragma unroll
        for (int j = 4 * h; j < 4 * h + 4; ++j) {
            const int r = 4 * (8 * (wid & 1) + j) + (lane >> 4), c = lane & 15;
            *(u32x4*)(s + r * 256 + ((((c >> 1) ^ (r & 7)) << 5) | ((c & 1) << 4))) = src[j];
        }
    };
    f32x4 acc[4][6];
    f32x4 accb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        accb[i] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    }
    const bool do_bias = p.db != nullptr && n2_0 == 0 && wc == 0;
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (bf16)1.0f;
    const int q4 = li >> 2, p4 = li & 3;
    bf16x8 fp[4], fq[6];
    auto frags = [&](int cur, int ks) {
        const char* sp = wsmem + cur * WSTAGE;
        const char* sq = sp + WBLK;
        const int row = ks * 32 + 4 * g + q4;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int cp = wr * 64 + tt * 16 + 4 * p4;
            fp[tt] = cat4(lds_read_tr16(tr_addr(sp, row, cp)), lds_read_tr16(tr_addr(sp, row + 16, cp)));
        }
#pragma unroll
        for (int tt = 0; tt < 6; ++tt) {
            const int c = wc * 96 + tt * 16;
            const char* qb = sq + (c >> 7) * WBLK;
            const int cq = (c & 127) + 4 * p4;
            fq[tt] = cat4(lds_read_tr16(tr_addr(qb, row, cq)), lds_read_tr16(tr_addr(qb, row + 16, cq)));
        }
    };
    auto fence = [&] {                                            // end of an LDS interval
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mma = [&] {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i][j] = mfma16(fp[i], fq[j], acc[i][j]);
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < 4; ++i) accb[i] = mfma16(fp[i], ones, accb[i]);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // one step on buffer `cur`: stg[s] holds tile st+1 (goes to the other buffer), tile st+3 is loaded into it afterwards
    auto step = [&](int cur, int st, u32x4 (&sreg)[8]) {
        frags(cur, 0);
        lwrite4(cur ^ 1, sreg, 0);
        gload4(st + 3, sreg, 0);
        fence();
        mma();
        frags(cur, 1);
        lwrite4(cur ^ 1, sreg, 1);
        gload4(st + 3, sreg, 1);
        fence();
        mma();
    };
    if constexpr (NI == 2) {
        bf16x8 fp2[2][4], fq2[2][6];
        auto frags2 = [&](int cur) {
            const char* sp = wsmem + cur * WSTAGE;
            const char* sq = sp + WBLK;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int row = ks * 32 + 4 * g + q4;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int cp = wr * 64 + tt * 16 + 4 * p4;
                    fp2[ks][tt] = cat4(lds_read_tr16(tr_addr(sp, row, cp)), lds_read_tr16(tr_addr(sp, row + 16, cp)));
                }
#pragma unroll
                for (int tt = 0; tt < 6; ++tt) {
                    const int c = wc * 96 + tt * 16;
                    const char* qb = sq + (c >> 7) * WBLK;
                    const int cq = (c & 127) + 4 * p4;
                    fq2[ks][tt] = cat4(lds_read_tr16(tr_addr(qb, row, cq)), lds_read_tr16(tr_addr(qb, row + 16, cq)));
                }
            }
        };
        auto mma2 = [&] {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc[i][j] = mfma16(fp2[ks][i], fq2[ks][j], acc[i][j]);
                if (do_bias) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) accb[i] = mfma16(fp2[ks][i], ones, accb[i]);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        gload4(0, stg[0], 0); gload4(0, stg[0], 1);
        lwrite4(0, stg[0], 0); lwrite4(0, stg[0], 1);
        gload4(1, stg[0], 0); gload4(1, stg[0], 1);
        __syncthreads();
        if (wr == 1) __builtin_amdgcn_s_barrier();                // waves 4-7 run one interval behind
        for (int st = 0; st < nsteps; ++st) {
            const int cur = st & 1;
            frags2(cur);
            lwrite4(cur ^ 1, stg[0], 0); lwrite4(cur ^ 1, stg[0], 1);     // tile st + 1 (a repeat of the last tile at the end)
            gload4(st + 2, stg[0], 0); gload4(st + 2, stg[0], 1);
            fence();
            mma2();
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();
    } else {
    gload4(0, stg[0], 0); gload4(0, stg[0], 1);
    gload4(1, stg[1], 0); gload4(1, stg[1], 1);
    lwrite4(0, stg[0], 0); lwrite4(0, stg[0], 1);
    gload4(2, stg[0], 0); gload4(2, stg[0], 1);
    __syncthreads();
    if (wr == 1) __builtin_amdgcn_s_barrier();                    // waves 4-7 run one interval behind
    for (int st = 0; st < nsteps; st += 2) {
        step(0, st, stg[1]);
        if (st + 1 < nsteps) step(1, st + 1, stg[0]);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    }
    if constexpr (SLAB) {
        const int tg = (wg - split * gp.ntiles), zt = split * gp.ntiles + tg;
        f32x4* o = (f32x4*)slabs + ((size_t)zt * 8 + wid) * (24 * 64) + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) o[(i * 6 + j) * 64] = acc[i][j];
        if (do_bias && li == 0) {
            float* ob = slabs + (size_t)gridDim.x * (8 * 24 * 64 * 4) + (size_t)zt * 128 + wr * 64 + 4 * g;
#pragma unroll
            for (int i = 0; i < 4; ++i) *(f32x4*)(ob + i * 16) = accb[i];
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n1 = n1_0 + wr * 64 + i * 16 + 4 * g + r;
            float* row = p.dW + (size_t)n1 * p.ldw + n2_0 + wc * 96 + li;
#pragma unroll
            for (int j = 0; j < 6; ++j) atomicAdd(row + j * 16, acc[i][j][r]);
            if (do_bias && li == 0) atomicAdd(p.db + n1, accb[i][r]);
        }
}

// One thread per (tile, wave, accumulator tile, lane): the nsplit partial f32x4 of its position are loaded together (up to 16
// loads in flight per thread; round 5's form walked them one dependent load at a time from 288 workgroups and cost more than the
// atomics it replaced), summed in a fixed order and added to dW; the bias parts by the last blocks of the grid.
__global__ __launch_bounds__(256) void tn_slab_finish_kernel(TnWideGroup gp, const float* slabs, int nsplit) {
    constexpr int PER_TILE = 8 * 24 * 64;                                     // f32x4 elements of one workgroup's slab
    const int nbody = gp.ntiles * PER_TILE / 256;
    const size_t zstride = (size_t)gp.ntiles * PER_TILE;                      // f32x4 per split
    if ((int)blockIdx.x < nbody) {
        const int e = blockIdx.x * 256 + threadIdx.x;
        int t = e / PER_TILE;
        const int w8 = e - t * PER_TILE, lane = w8 & 63, ij = (w8 >> 6) % 24, w = (w8 >> 6) / 24;
        const f32x4* src = (const f32x4*)slabs + (size_t)t * PER_TILE + w8;
        f32x4 v[16];
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int z0 = 0; z0 < nsplit; z0 += 16) {              // unconditional loads (clamped index), values selected afterwards
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = src[(size_t)min(z0 + u, nsplit - 1) * zstride];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float keep = z0 + u < nsplit ? 1.f : 0.f;
                sum += v[u] * keep;
            }
        }
        int it0 = 0;
        while (it0 + 1 < gp.nitems && t >= gp.tile_end[it0]) ++it0;
        if (it0 > 0) t -= gp.tile_end[it0 - 1];
        const TnParams& p = gp.item[it0];
        const int nt2 = p.N2 / WQ;
        const int n1_0 = (t / nt2) * 128, n2_0 = (t % nt2) * WQ;
        const int wr = w >> 2, wc = w & 3, i = ij / 6, j = ij - 6 * i, g = lane >> 4, li = lane & 15;
        float* row = p.dW + (size_t)(n1_0 + wr * 64 + i * 16 + 4 * g) * p.ldw + n2_0 + wc * 96 + j * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) row[(size_t)r * p.ldw] += sum[r];
        return;
    }
    const int e = (blockIdx.x - nbody) * 256 + threadIdx.x;
    if (e >= gp.ntiles * 128) return;
    int t = e >> 7;
    const int tg = t, c = e & 127;
    int it0 = 0;
    while (it0 + 1 < gp.nitems && t >= gp.tile_end[it0]) ++it0;
    if (it0 > 0) t -= gp.tile_end[it0 - 1];
    const TnParams& p = gp.item[it0];
    const int nt2 = p.N2 / WQ;
    if (p.db == nullptr || t % nt2 != 0) return;
    const float* bb = slabs + (size_t)nsplit * zstride * 4 + (size_t)tg * 128 + c;
    float sum = 0.f;
    for (int z = 0; z < nsplit; ++z) sum += bb[(size_t)z * gp.ntiles * 128];
    p.db[(t / nt2) * 128 + c] += sum;
}

}  // namespace

// epilogues the four-workgroups-per-CU kernel is built for (the fp32-aux ones need 64 more registers than it has)
static constexpr bool w4_epi(int e) {
    return e == SAIS_EPI_BIAS_BF16 || e == SAIS_EPI_BIAS_GELU_GRAD_BF16 || e == SAIS_EPI_MUL_BF16 || e == SAIS_EPI_BIAS_GELU_BF16 ||
           e == SAIS_EPI_BIAS_RELU_BF16;
}
#if SAIS_EXPERIMENTAL
#define LAUNCH_NT_EXP(E)                                                                    \
        if (big && nt_w8r && w4_epi(E) && g->K == 6 * BK) {                                 \
            static thread_local bool set8r = false;                                         \
            if (!set8r) {                                                                   \
                if (hipFuncSetAttribute((const void*)gemm_nt_w8r_kernel<E>,                 \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 5 * TILE_BYTES) != hipSuccess) \
                    return SAIS_ERR_LAUNCH;                                                 \
                set8r = true;                                                               \
            }                                                                               \
            const int nt_ = (int)grid.x;                                                    \
            hipLaunchKernelGGL(gemm_nt_w8r_kernel<E>, dim3(nt_ < nt_grid ? nt_ : nt_grid), dim3(512), 5 * TILE_BYTES, \
                               (hipStream_t)stream, p, nt_);                                \
        } else if (big && nt_w16 && g->K / BK >= 5) {                                       \
            static thread_local bool set16 = false;                                         \
            if (!set16) {                                                                   \
                if (hipFuncSetAttribute((const void*)gemm_nt_w16_kernel<E>,                 \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 10 * TILE_BYTES) != hipSuccess) \
                    return SAIS_ERR_LAUNCH;                                                 \
                set16 = true;                                                               \
            }                                                                               \
            const int nt_ = (int)grid.x, half_ = (nt_ + 1) / 2;                             \
            hipLaunchKernelGGL(gemm_nt_w16_kernel<E>, dim3(half_ < 256 ? half_ : 256), dim3(1024), 10 * TILE_BYTES, \
                               (hipStream_t)stream, p, nt_);                                \
        } else if (big && nt_w4 && w4_epi(E) && g->K >= 2 * QK) {                           \
            const int nt_ = (int)grid.x;                                                    \
            hipLaunchKernelGGL(gemm_nt_w4q_kernel<E>, dim3(nt_ < 1024 ? nt_ : 1024), dim3(256), 5 * QTILE, \
                               (hipStream_t)stream, p, nt_);                                \
        } else
#else
#define LAUNCH_NT_EXP(E)
#endif
#define LAUNCH_NT(E)                                                                        \
    case E:                                                                                 \
        LAUNCH_NT_EXP(E)                                                                    \
        if (big) {                                                                          \
            static thread_local bool set8p = false;                                         \
            if (!set8p) {                                                                   \
                if (hipFuncSetAttribute((const void*)gemm_nt_w8p_kernel<E>,                 \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 5 * TILE_BYTES) != hipSuccess) \
                    return SAIS_ERR_LAUNCH;                                                 \
                set8p = true;                                                               \
            }                                                                               \
            const int nt_ = (int)grid.x;                                                    \
            hipLaunchKernelGGL(gemm_nt_w8p_kernel<E>, dim3(nt_ < nt_grid ? nt_ : nt_grid), dim3(512), 5 * TILE_BYTES, \
                               (hipStream_t)stream, p, nt_);                                \
        } else                                                                              \
            hipLaunchKernelGGL(gemm_nt_kernel<E>, grid, dim3(256), 0, (hipStream_t)stream, p);  \
        break;

extern "C" int sais_gemm_nt_row_(const SaisGemm* g, void* stream);      // gemm_row.hip: row-owning tiles, N = 384
extern "C" int sais_gemm_tn_xl_(const SaisTnItem* items, int nitems, int M, int nwaves, void* slabs, size_t slab_bytes, void* stream);   // gemm_tn_xl.hip: 192 x 384 dW tiles
extern "C" size_t sais_gemm_tn_xl_slab_bytes_(const SaisTnItem* items, int nitems, int M, int nwaves);
// SAIS_TN_XL = 4 | 8 waves (default 4), 0 = the 128 x 384 kernel; SAIS_TN_XL_SLABS = 0: fp32 atomics instead of slabs + finish;
// SAIS_TN_SLABS = 1: the round-5 slab form of the 128 x 384 kernel (implies SAIS_TN_XL = 0)
static int tn_xl_waves() {
    static const int v = [] {
        const char* s = getenv("SAIS_TN_SLABS");
        if (s && atoi(s) != 0) return 0;
        const char* e = getenv("SAIS_TN_XL");
        return e ? atoi(e) : 4;
    }();
    return v;
}
static bool tn_xl_slabs() {
    static const bool v = [] { const char* e = getenv("SAIS_TN_XL_SLABS"); return e ? atoi(e) != 0 : true; }();
    return v;
}
CLK_EXPORT(gemm)


extern "C" int sais_gemm_nt(const SaisGemm* g, void* stream) {
    SAIS_ENTER();
    if (!g || !g->A || !g->B || !g->out) return SAIS_ERR_ARG;
    if (g->M <= 0 || g->N % BN || g->K % BK || g->lda % 8 || g->ldb % 8 || g->ldo % 8) return SAIS_ERR_ARG;
    if ((g->epilogue == SAIS_EPI_BIAS_GELU_GRAD_BF16 || g->epilogue == SAIS_EPI_BIAS_GELU_GRADQ_BF16) && !g->out2) return SAIS_ERR_ARG;
    if (g->epilogue == SAIS_EPI_BIAS_GELU_GRADQ_BF16 && (g->ldo2 % 16 || ((uintptr_t)g->out2 & 15))) return SAIS_ERR_ARG;
    if (g->epilogue == SAIS_EPI_MULQ_BF16 && (!g->aux || g->ldaux % 16 || ((uintptr_t)g->aux & 15))) return SAIS_ERR_ARG;
    if ((g->epilogue == SAIS_EPI_MUL_BF16 || g->epilogue == SAIS_EPI_DGELU_BF16 || g->epilogue == SAIS_EPI_DRELU_BF16 ||
         g->epilogue == SAIS_EPI_BIAS_RESID_F32) && !g->aux)
        return SAIS_ERR_ARG;
    NtParams p{(const bf16*)g->A, (const bf16*)g->B, g->lda, g->ldb, g->M, g->N, g->K, g->bias,
               g->out, g->ldo, g->out2, g->ldo2, g->aux, g->ldaux, g->grp_in, g->grp_out, g->grp_off, g->rowscale};
    if (g->rowscale && g->epilogue != SAIS_EPI_BIAS_RESID_F32) return SAIS_ERR_ARG;
    dim3 grid((g->N / BN) * ((g->M + BM - 1) / BM));
    // Two kernels, chosen by M alone: the four-wave 128x128 kernel for small M (inference batches, tests, the
    // patch-embed epilogue) and the persistent eight-wave A-ring kernel for the ViT GEMMs of a training step
    // (M >= 8192).  Round 1's other variants (wave-specialised, register-stationary, non-persistent eight-wave,
    // four-wave A-ring) were measured slower inside the step and are gone from the library (LABNOTES.md 4.1).
    const bool big = g->M >= 8192;
    // persistent workgroups of the eight-wave kernel (2 per CU); SAIS_NT_GRID=256 = one per CU (diagnostic: LABNOTES R5.2)
    static const int nt_grid = [] { const char* e = getenv("SAIS_NT_GRID"); return e && atoi(e) > 0 ? atoi(e) : 512; }();
    static const bool nt_w8r = [] { const char* e = getenv("SAIS_NT_W8R"); return e ? atoi(e) != 0 : false; }();
    static const bool nt_w16 = [] { const char* e = getenv("SAIS_NT_W16"); return e ? atoi(e) != 0 : false; }();
    static const bool nt_w4 = [] { const char* e = getenv("SAIS_NT_W4"); return e ? atoi(e) != 0 : false; }();
    if (g->epilogue == SAIS_EPI_RAW_SLABS_F32) {                // split-K over grp_in slices: small M only, raw fp32 slabs
        if (big || g->grp_in < 1 || g->grp_in > g->K / BK || g->ldo % 4) return SAIS_ERR_ARG;
        hipLaunchKernelGGL(gemm_nt_kernel<SAIS_EPI_RAW_SLABS_F32>, dim3(grid.x, g->grp_in), dim3(256), 0, (hipStream_t)stream, p);
        return sais_check_launch();
    }
    // the plain N = 384 GEMMs of a training step (dX of proj, the last block's fc2): balanced row tiles of gemm_row.hip
    if (big && g->N == 384 && (g->epilogue == SAIS_EPI_BIAS_BF16 || (g->epilogue == SAIS_EPI_BIAS_RESID_F32 && !g->out2)))
        return sais_gemm_nt_row_(g, stream);
    if (big && g->rowscale) return SAIS_ERR_ARG;               // the eight-wave kernel has no row-scale epilogue
    switch (g->epilogue) {
        LAUNCH_NT(SAIS_EPI_BIAS_BF16)
        LAUNCH_NT(SAIS_EPI_BIAS_RELU_BF16)
        LAUNCH_NT(SAIS_EPI_BIAS_F32)
        LAUNCH_NT(SAIS_EPI_BIAS_RESID_F32)
        LAUNCH_NT(SAIS_EPI_BIAS_GELU_BF16)
        LAUNCH_NT(SAIS_EPI_DGELU_BF16)
        LAUNCH_NT(SAIS_EPI_DRELU_BF16)
        LAUNCH_NT(SAIS_EPI_PATCH_F32)
        LAUNCH_NT(SAIS_EPI_BIAS_GELU_GRAD_BF16)
        LAUNCH_NT(SAIS_EPI_MUL_BF16)
        LAUNCH_NT(SAIS_EPI_BIAS_GELU_GRADQ_BF16)
        LAUNCH_NT(SAIS_EPI_MULQ_BF16)
        default: return SAIS_ERR_ARG;
    }
    return sais_check_launch();
}

extern "C" int sais_splitk_finish(const float* slabs, int nslabs, int M, int N, int lds, const float* bias,
                                  const float* rowscale, const float* aux, int ldaux, float* out32, int ldo32, void* out16,
                                  int ldo16, void* stream) {
    SAIS_ENTER();
    if (!slabs || nslabs < 1 || M <= 0 || N <= 0 || N % 4 || lds % 4 || (!out32 && !out16)) return SAIS_ERR_ARG;
    if ((aux && ldaux % 4) || (out32 && ldo32 % 4) || (out16 && ldo16 % 4)) return SAIS_ERR_ARG;
    const long n = (long)M * (N / 4);
    const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(splitk_finish_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, slabs, nslabs, M, N, lds, bias,
                       rowscale, aux, ldaux, out32, ldo32, (bf16*)out16, ldo16);
    return sais_check_launch();
}

// out[m][n] = epilogue( sum_z ws[z][m][n] + bias[n] , aux[m][n] )  — second half of the split-K fp32 GEMM
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* ws, int ks, int M, int N, const float* bias,
                                                            const float* aux, int ldaux, float* out, int ldo, float p_drop,
                                                            const unsigned long long* rng, unsigned site) {
    const int n4 = N >> 2;
    const unsigned thr = drop_threshold(p_drop);
    const float inv = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < (long)M * n4; i += (long)gridDim.x * 256) {
        const int m = i / n4, n = (i - (long)m * n4) * 4;
        f32x4 y = *(const f32x4*)(ws + (size_t)m * N + n);
        for (int z = 1; z < ks; ++z) y += *(const f32x4*)(ws + ((size_t)z * M + m) * N + n);
        if (bias) y += *(const f32x4*)(bias + n);
        f32x4 keep = {1.f, 1.f, 1.f, 1.f};
        if (p_drop > 0.f) {
#pragma unroll
            for (int j = 0; j < 4; ++j) keep[j] = philox_keep(rng, site, (unsigned long long)m * N + n + j, thr) ? inv : 0.f;
        }
        if constexpr (EPI == SAIS_EPI_BIAS_RESID_F32) y = y * keep + *(const f32x4*)(aux + (size_t)m * ldaux + n);
        if constexpr (EPI == SAIS_EPI_BIAS_RELU_F32) {
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = fmaxf(y[j], 0.f) * keep[j];
        }
        if constexpr (EPI == SAIS_EPI_DRELU_F32) {
            const f32x4 u = *(const f32x4*)(aux + (size_t)m * ldaux + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = u[j] > 0.f ? y[j] * keep[j] : 0.f;
        }
        *(f32x4*)(out + (size_t)m * ldo + n) = y;
    }
}

#define LAUNCH_RED(E)                                                                                             \
    case E:                                                                                                       \
        hipLaunchKernelGGL(splitk_reduce_kernel<E>, dim3(rgrid), dim3(256), 0, (hipStream_t)stream,              \
                           (const float*)g->out2, ks, g->M, g->N, g->bias, (const float*)g->aux, g->ldaux,        \
                           (float*)g->out, g->ldo, g->p_drop, g->rng_state, g->site);                             \
        break;

#define LAUNCH_NT32(E)                                                                            \
    case E:                                                                                       \
        hipLaunchKernelGGL(gemm_nt_f32x3_kernel<E>, grid, dim3(256), 0, (hipStream_t)stream, p);  \
        break;

extern "C" int sais_gemm_nt_f32(const SaisGemm* g, void* stream) {
    SAIS_ENTER();
    if (!g || !g->A || !g->B || !g->out) return SAIS_ERR_ARG;
    if (g->M <= 0 || g->N % BN || g->K % BK || g->lda % 4 || g->ldb % 4 || g->ldo % 4) return SAIS_ERR_ARG;
    NtParams p{(const bf16*)g->A, (const bf16*)g->B, g->lda, g->ldb, g->M, g->N, g->K, g->bias,
               g->out, g->ldo, g->out2, g->ldo2, g->aux, g->ldaux, 1, 0, 0, nullptr, g->p_drop, g->rng_state, g->site};
    if (g->p_drop < 0.f || g->p_drop >= 1.f || (g->p_drop > 0.f && (!g->rng_state || g->epilogue == SAIS_EPI_BIAS_F32)))
        return SAIS_ERR_ARG;
    dim3 grid(g->N / BN, (g->M + BM - 1) / BM);
    // Few output tiles (M = clips*(T+1) rows): split K over gridDim.z into the caller's workspace (out2 = f32
    // [ldo2][M][N], ldo2 = number of splits) and finish with a tiny reduce+epilogue kernel, so that dozens of CUs
    // work instead of <= 9 and the exposed per-K-tile load latency is paid K/64/ks times instead of K/64.
    int ks = 1;
    if (g->out2 && g->ldo2 > 1) {
        ks = g->ldo2;
        if ((g->K / BK) % ks) return SAIS_ERR_ARG;
        p.grp_in = ks;
        grid.z = ks;
    }
    switch (g->epilogue) {
        LAUNCH_NT32(SAIS_EPI_BIAS_F32)
        LAUNCH_NT32(SAIS_EPI_BIAS_RESID_F32)
        LAUNCH_NT32(SAIS_EPI_BIAS_RELU_F32)
        LAUNCH_NT32(SAIS_EPI_DRELU_F32)
        default: return SAIS_ERR_ARG;
    }
    if (ks > 1) {
        long n = (long)g->M * (g->N / 4);
        int rgrid = (int)((n + 255) / 256);
        switch (g->epilogue) {
            LAUNCH_RED(SAIS_EPI_BIAS_F32)
            LAUNCH_RED(SAIS_EPI_BIAS_RESID_F32)
            LAUNCH_RED(SAIS_EPI_BIAS_RELU_F32)
            LAUNCH_RED(SAIS_EPI_DRELU_F32)
        }
    }
    return sais_check_launch();
}

static int launch_tn(const void* P, int ldp, const void* Q, int ldq, int M, int N1, int N2, float* dW, int ldw,
                     float* db, int nsplit, void* stream, bool f32) {
    if (!P || !Q || !dW || M <= 0 || N1 % 128 || N2 % 128 || ldp % 8 || ldq % 8 || nsplit <= 0) return SAIS_ERR_ARG;
    int rows = (M + nsplit - 1) / nsplit;
    rows = (rows + TK - 1) / TK * TK;
    int ns = (M + rows - 1) / rows;
    TnParams p{P, Q, ldp, ldq, M, N1, N2, dW, ldw, db, rows};
    dim3 grid((N2 / 128) * (N1 / 128) * ns);
    if (f32) hipLaunchKernelGGL(gemm_tn_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(gemm_tn_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, p);
    return sais_check_launch();
}

extern "C" size_t sais_gemm_tn_grouped_slab_bytes(const SaisTnItem* items, int nitems, int M) {
    if (!items || nitems <= 0 || nitems > SAIS_TN_MAX_ITEMS) return 0;
    if (tn_xl_waves()) {
        const size_t need = tn_xl_slabs() ? sais_gemm_tn_xl_slab_bytes_(items, nitems, M, tn_xl_waves()) : 0;
        if (need) return need;
    }
    // the 128 x 384 kernel's slab form is opt-in (SAIS_TN_SLABS = 1): without the switch no workspace is asked for
    static const bool old_slabs = [] { const char* e = getenv("SAIS_TN_SLABS"); return e ? atoi(e) != 0 : false; }();
    if (!old_slabs || M % TK || M < 8192) return 0;
    int wt = 0;
    for (int i = 0; i < nitems; ++i) {
        if (items[i].N1 % 128 || items[i].N2 % WQ) return 0;
        wt += (items[i].N1 / 128) * (items[i].N2 / WQ);
    }
    int wns = 256 / wt < 1 ? 1 : 256 / wt;
    const int wrows = ((M + wns - 1) / wns + TK - 1) / TK * TK;
    wns = (M + wrows - 1) / wrows;
    return wns > 1 ? (size_t)wt * wns * (8 * 24 * 64 * 16 + 128 * 4) : 0;
}

extern "C" int sais_gemm_tn_grouped(const SaisTnItem* items, int nitems, int M, int nsplit, void* stream) {
    return sais_gemm_tn_grouped_ws(items, nitems, M, nsplit, nullptr, 0, stream);
}

extern "C" int sais_gemm_tn_grouped_ws(const SaisTnItem* items, int nitems, int M, int nsplit, void* slabs, size_t slab_bytes,
                                       void* stream) {
    SAIS_ENTER();
    if (!items || nitems <= 0 || nitems > SAIS_TN_MAX_ITEMS || M <= 0 || nsplit <= 0) return SAIS_ERR_ARG;
    int rows = (M + nsplit - 1) / nsplit;
    rows = (rows + TK - 1) / TK * TK;
    const int ns = (M + rows - 1) / rows;
    TnGroup gp;
    gp.nitems = nitems;
    int total = 0;
    for (int i = 0; i < nitems; ++i) {
        const SaisTnItem& t = items[i];
        if (!t.P || !t.Q || !t.dW || t.N1 % 128 || t.N2 % 128 || t.ldp % 8 || t.ldq % 8) return SAIS_ERR_ARG;
        gp.item[i] = TnParams{t.P, t.Q, t.ldp, t.ldq, M, t.N1, t.N2, t.dW, t.ldw, t.db, rows};
        total += (t.N1 / 128) * (t.N2 / 128);
        gp.tile_end[i] = total;
    }
    gp.ntiles = total;
    // large tiles (192 x 384, gemm_tn_xl.hip) when every N1 % 192 == 0, N2 % 384 == 0 and M % 32 == 0
    if (tn_xl_waves()) {
        const bool sl = slabs != nullptr && tn_xl_slabs();
        const int r = sais_gemm_tn_xl_(items, nitems, M, tn_xl_waves(), sl ? slabs : nullptr, sl ? slab_bytes : 0, stream);
        if (r != 0) return r > 0 ? SAIS_OK : r;
    }
    // wide tiles (128 x 384) when every item allows them and M is a whole number of 64-row steps
    bool wide = M % TK == 0 && M >= 8192;
    for (int i = 0; i < nitems && wide; ++i) wide = items[i].N2 % WQ == 0;
    if (wide) {
        TnWideGroup wg;
        wg.nitems = nitems;
        int wt = 0;
        for (int i = 0; i < nitems; ++i) {
            wt += (items[i].N1 / 128) * (items[i].N2 / WQ);
            wg.tile_end[i] = wt;
        }
        wg.ntiles = wt;
        // one workgroup per CU: as many M-splits as keep the grid within one round of 256
        int wns = 256 / wt < 1 ? 1 : 256 / wt;
        int wrows = ((M + wns - 1) / wns + TK - 1) / TK * TK;
        wns = (M + wrows - 1) / wrows;
        for (int i = 0; i < nitems; ++i) { wg.item[i] = gp.item[i]; wg.item[i].rows_per_split = wrows; }
        static thread_local bool lds_set = false;
        if (!lds_set) {
            if (hipFuncSetAttribute((const void*)gemm_tn_pp_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WSTAGE) != hipSuccess ||
                hipFuncSetAttribute((const void*)gemm_tn_pp_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WSTAGE) != hipSuccess)
                return SAIS_ERR_LAUNCH;
            lds_set = true;
        }
        const size_t need = (size_t)wt * wns * (8 * 24 * 64 * 16 + 128 * 4);
        // opt-in (SAIS_TN_SLABS=1): bit-reproducible weight gradients.  Measured SLOWER than the atomics (LABNOTES R5.1: 254 vs 240 us
        // stand-alone, 12.86 vs 12.76 ms per step) — the atomic tail this was built to remove is not there.
        static const bool use_slabs = [] { const char* e = getenv("SAIS_TN_SLABS"); return e ? atoi(e) != 0 : false; }();
        if (slabs && wns > 1 && use_slabs) {
            if (slab_bytes < need || ((uintptr_t)slabs & 15)) return SAIS_ERR_ARG;
            hipLaunchKernelGGL(gemm_tn_pp_kernel<true>, dim3(wt * wns), dim3(512), 2 * WSTAGE, (hipStream_t)stream, wg, (float*)slabs);
            hipLaunchKernelGGL(tn_slab_finish_kernel, dim3(wt * (8 * 24 * 64) / 256 + (wt * 128 + 255) / 256), dim3(256), 0, (hipStream_t)stream, wg, (const float*)slabs, wns);
        } else {
#if SAIS_EXPERIMENTAL
            static const int tn_ni = [] { const char* e = getenv("SAIS_TN_NI"); return e ? atoi(e) : 4; }();
            if (tn_ni == 2) {
                static thread_local bool set2 = false;
                if (!set2) {
                    if (hipFuncSetAttribute((const void*)gemm_tn_pp_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WSTAGE) != hipSuccess)
                        return SAIS_ERR_LAUNCH;
                    set2 = true;
                }
                hipLaunchKernelGGL((gemm_tn_pp_kernel<false, 2>), dim3(wt * wns), dim3(512), 2 * WSTAGE, (hipStream_t)stream, wg, (float*)nullptr);
            } else
#endif
            hipLaunchKernelGGL(gemm_tn_pp_kernel<false>, dim3(wt * wns), dim3(512), 2 * WSTAGE, (hipStream_t)stream, wg, (float*)nullptr);
        }
        return sais_check_launch();
    }
    hipLaunchKernelGGL(gemm_tn_grouped_kernel, dim3(total * ns), dim3(256), 0, (hipStream_t)stream, gp);
    return sais_check_launch();
}

extern "C" int sais_gemm_tn_grouped_f32(const SaisTnItem* items, int nitems, int M, int nsplit, void* stream) {
    SAIS_ENTER();
    if (!items || nitems <= 0 || nitems > SAIS_TN_MAX_ITEMS || M <= 0 || nsplit <= 0) return SAIS_ERR_ARG;
    int rows = (M + nsplit - 1) / nsplit;
    rows = (rows + TK - 1) / TK * TK;
    const int ns = (M + rows - 1) / rows;
    TnGroup gp;
    gp.nitems = nitems;
    int total = 0;
    for (int i = 0; i < nitems; ++i) {
        const SaisTnItem& t = items[i];
        if (!t.P || !t.Q || !t.dW || t.N1 % 128 || t.N2 % 128 || t.ldp % 4 || t.ldq % 4) return SAIS_ERR_ARG;
        gp.item[i] = TnParams{t.P, t.Q, t.ldp, t.ldq, M, t.N1, t.N2, t.dW, t.ldw, t.db, rows};
        total += (t.N1 / 128) * (t.N2 / 128);
        gp.tile_end[i] = total;
    }
    gp.ntiles = total;
    if (ns == 1 && total < 200) {
        // one M-split and fewer tiles than CUs (the temporal layers: 132): 64-row dW tiles = twice the workgroups
        int t64 = 0;
        for (int i = 0; i < nitems; ++i) {
            t64 += (items[i].N1 / 64) * (items[i].N2 / 128);
            gp.tile_end[i] = t64;
        }
        gp.ntiles = t64;
        hipLaunchKernelGGL((gemm_tn_grouped_f32_kernel<true, 64>), dim3(t64), dim3(256), 0, (hipStream_t)stream, gp);
        return sais_check_launch();
    }
    if (ns == 1) hipLaunchKernelGGL(gemm_tn_grouped_f32_kernel<true>, dim3(total), dim3(256), 0, (hipStream_t)stream, gp);
    else hipLaunchKernelGGL(gemm_tn_grouped_f32_kernel<false>, dim3(total * ns), dim3(256), 0, (hipStream_t)stream, gp);
    return sais_check_launch();
}

extern "C" int sais_gemm_tn_f32(const void* P, int ldp, const void* Q, int ldq, int M, int N1, int N2,
                                float* dW, int ldw, float* db, int nsplit, void* stream) {
    SAIS_ENTER();
    return launch_tn(P, ldp, Q, ldq, M, N1, N2, dW, ldw, db, nsplit, stream, true);
}

extern "C" int sais_gemm_tn(const void* P, int ldp, const void* Q, int ldq, int M, int N1, int N2,
                            float* dW, int ldw, float* db, int nsplit, void* stream) {
    SAIS_ENTER();
    return launch_tn(P, ldp, Q, ldq, M, N1, N2, dW, ldw, db, nsplit, stream, false);
}
