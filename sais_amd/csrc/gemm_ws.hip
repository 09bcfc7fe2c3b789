// Wave-specialised persistent NT GEMM for gfx950: C[M,N] = A[M,K] . B[N,K]^T + epilogue, 256 x 128 tiles.
//
// Why (measured): in the 128x128 kernel every wave does everything in order — issue LDS-DMA
// (~70-85 issue cycles per 1-KiB piece), wait, read fragments, MFMA, epilogue — and the two waves of a SIMD do it
// in lockstep, so the phases add up and the matrix pipe is busy 26-30 %.  Here the roles are split:
//   waves 4-7  LOADERS   : only issue global_load_lds (12 pieces each per K-tile) and wait for them (counted vmcnt)
//   waves 0-3  CONSUMERS : one per SIMD, only ds_read fragments + MFMA (64 x 128 sub-tile = 4 x 8 MFMA tiles,
//                          128 accumulator VGPRs) and the epilogue
// One s_barrier per K-tile couples them through a 3-stage LDS ring (3 x 48 KiB): at barrier q the loaders have
// seen their pieces of K-tile q land, the consumers have finished K-tile q-1, so the loaders may refill that stage
// with K-tile q+2.  The workgroup is PERSISTENT (grid = min(tiles, CUs)): the loaders run ahead across tile
// boundaries, so the next tile's first K-tiles are already in LDS while the consumers run the epilogue, and
// kernel-argument loads / address setup / pipeline fill are paid once per CU instead of once per tile.
#include "common.hpp"
#include "../../include/sais_hip.h"

namespace {
constexpr int BM = 256, BN = 128, BK = 64;
constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;   // 32 + 16 KiB
constexpr int PPW = 12;                         // LDS-DMA pieces per loader wave per K-tile (8 of A, 4 of B)
constexpr int STG_BYTES = BM * BN * 2;          // bf16 output staging of one tile (hand-off variant): 64 KiB

// Store hand-off (bf16-output epilogues).  Measured: the consumers spent ~7.8k cycles per tile issuing their 16
// epilogue stores (HBM-write back-pressure) with the matrix pipe idle, while each loader wave idled 1-2k cycles per
// K-tile at the barrier.  So the consumers only finish the arithmetic and drop the bf16 tile into an LDS staging
// buffer (chunk index XOR row: conflict-free both ways); the LOADERS write it to HBM as whole 256-B row segments,
// a few per K-tile of the NEXT tile, behind their LDS-DMA issue (the counted vmcnt leaves exactly those stores in
// flight).  For the GELU epilogue the loaders also evaluate GELU on the staged pre-activation and write both
// tensors.  LDS: 2-stage ring (96 KiB) + staging (64 KiB) = 160 KiB.
template <bool HANDOFF> struct Cfg { static constexpr int NSTAGE = HANDOFF ? 2 : 3; };
DEVINL int stg_off(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }

struct NtParams {
    const bf16* A; const bf16* B;
    int lda, ldb, M, N, K;
    const float* bias;
    void* out; int ldo;
    void* out2; int ldo2;
    const void* aux; int ldaux;
    int grp_in, grp_out, grp_off;
    int ntm, ntn, ntiles;
};

DEVINL int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
// weight-row permutation of the 128-column panel: LDS row p = 16 nt + 4 g + r  <-  column 32 g + 4 nt + r
DEVINL int perm128(int p) { return 32 * ((p >> 2) & 3) + 4 * (p >> 4) + (p & 3); }
DEVINL int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}
DEVINL void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

DEVINL void st_bf16(void* base, int ld, size_t row, int n, const float (&y)[32]) {
    bf16* o = (bf16*)base + row * ld + n;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        bf16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (bf16)y[8 * c + i];
        *(bf16x8*)(o + 8 * c) = v;
    }
}
DEVINL void st_f32(void* base, int ld, size_t row, int n, const float (&y)[32]) {
    float* o = (float*)base + row * ld + n;
#pragma unroll
    for (int c = 0; c < 8; ++c) *(f32x4*)(o + 4 * c) = f32x4{y[4 * c], y[4 * c + 1], y[4 * c + 2], y[4 * c + 3]};
}

// one output row m, 32 contiguous columns n..n+31: loads first (vmcnt is in-order and counts stores), then stores
template <int EPI>
DEVINL void epilogue_row(const NtParams& p, int m, int n, float (&y)[32], const float (&b)[32]) {
#pragma unroll
    for (int i = 0; i < 32; ++i) y[i] += b[i];
    if constexpr (EPI == SAIS_EPI_BIAS_BF16) {
        st_bf16(p.out, p.ldo, m, n, y);
    } else if constexpr (EPI == SAIS_EPI_BIAS_RELU_BF16) {
#pragma unroll
        for (int i = 0; i < 32; ++i) y[i] = fmaxf(y[i], 0.f);
        st_bf16(p.out, p.ldo, m, n, y);
    } else if constexpr (EPI == SAIS_EPI_BIAS_F32) {
        st_f32(p.out, p.ldo, m, n, y);
    } else if constexpr (EPI == SAIS_EPI_BIAS_RESID_F32 || EPI == SAIS_EPI_PATCH_F32) {
        size_t arow = m, orow = m;
        if constexpr (EPI == SAIS_EPI_PATCH_F32) {
            const int f = m / p.grp_in, q = m - f * p.grp_in;
            arow = q + p.grp_off;
            orow = (size_t)f * p.grp_out + q + p.grp_off;
        }
        const float* r = (const float*)p.aux + arow * p.ldaux + n;
        f32x4 t[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) t[c] = *(const f32x4*)(r + 4 * c);
#pragma unroll
        for (int i = 0; i < 32; ++i) y[i] += t[i >> 2][i & 3];
        st_f32(p.out, p.ldo, orow, n, y);
        if constexpr (EPI == SAIS_EPI_BIAS_RESID_F32)
            if (p.out2) st_bf16(p.out2, p.ldo2, m, n, y);
    } else if constexpr (EPI == SAIS_EPI_BIAS_GELU_BF16) {
        if (p.out2) st_bf16(p.out2, p.ldo2, m, n, y);
#pragma unroll
        for (int i = 0; i < 32; ++i) y[i] = gelu_erf(y[i]);
        st_bf16(p.out, p.ldo, m, n, y);
    } else if constexpr (EPI == SAIS_EPI_DGELU_BF16 || EPI == SAIS_EPI_DRELU_BF16) {
        const bf16* u = (const bf16*)p.aux + (size_t)m * p.ldaux + n;
        bf16x8 t[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) t[c] = *(const bf16x8*)(u + 8 * c);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const float uu = (float)t[i >> 3][i & 7];
            if constexpr (EPI == SAIS_EPI_DGELU_BF16) y[i] *= dgelu_erf(uu);
            else y[i] = uu > 0.f ? y[i] : 0.f;
        }
        st_bf16(p.out, p.ldo, m, n, y);
    }
}

template <int EPI, bool HANDOFF>
__global__ __launch_bounds__(512) void gemm_nt_ws_kernel(NtParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // ring (NSTAGE * 48 KiB) [+ 64 KiB staging]
    constexpr int NSTAGE = Cfg<HANDOFF>::NSTAGE;
    char* const stg = smem + NSTAGE * STAGE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);         // 0..3 consumers, 4..7 loaders
    const int nk = p.K / BK;
    const int G = gridDim.x;
    const int first = xcd_remap(blockIdx.x, G);                       // tiles first, first + G, first + 2G, ...
    const int cnt = first < p.ntiles ? (p.ntiles - first + G - 1) / G : 0;
    const int total = cnt * nk;                                       // K-tile stream length of this workgroup

    if (wid >= 4) {
        // ------------------------------------------------------------------ LOADER
        const int lw = wid - 4;
        const int sub = lane >> 3, schunk = (lane & 7) ^ sub;         // piece = 8 rows x 128 B, swizzled source
        const bf16* src[PPW];
        int dst[PPW];
        auto set_tile = [&](int tile) {
            const int m0 = (tile / p.ntn) * BM, n0 = (tile % p.ntn) * BN;
#pragma unroll
            for (int j = 0; j < 8; ++j) {                             // A: 32 pieces, this wave 8 lw .. 8 lw + 7
                const int piece = 8 * lw + j;
                int m = m0 + 8 * piece + sub;
                m = m < p.M ? m : p.M - 1;
                src[j] = p.A + (size_t)m * p.lda + schunk * 8;
                dst[j] = piece * 1024;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {                             // B: 16 pieces, this wave 4 lw .. 4 lw + 3
                const int piece = 4 * lw + j;
                src[8 + j] = p.B + (size_t)(n0 + perm128(8 * piece + sub)) * p.ldb + schunk * 8;
                dst[8 + j] = A_BYTES + piece * 1024;
            }
        };
        int it = 0, ikt = 0, istage = 0;                              // issue cursor: tile index, K-tile, ring stage
        auto issue_next = [&]() {
            if (ikt == 0) set_tile(first + it * G);
            char* s = smem + istage * STAGE;
#pragma unroll
            for (int j = 0; j < PPW; ++j) glds16(src[j] + ikt * BK, s + dst[j]);
            if (++ikt == nk) { ikt = 0; ++it; }
            istage = istage == NSTAGE - 1 ? 0 : istage + 1;
        };
        int issued = 0;
        for (; issued < NSTAGE - 1 && issued < total; ++issued) issue_next();
        if constexpr (!HANDOFF) {
            for (int q = 0; q < total; ++q) {
                // my pieces of K-tile q have landed (those of q+1 may still be in flight)
                if (q + 1 < issued) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                         // consumers are done with K-tile q-1
                if (issued < total) { issue_next(); ++issued; }       // refill its stage with K-tile q+2
            }
        } else {
            // one K-tile ahead + store job of the previous tile spread over the K-tiles of the current one
            const int Q = nk > 16 ? 1 : (nk > 8 ? 2 : (nk > 4 ? 4 : 16)); // chunk rounds per K-tile: Q (nk-1) >= 16
            const int srow = lw * 64 + (lane >> 4), schk = lane & 15;  // round c: row srow + 4 c, 16-B chunk schk
            int done = 16, m0s = 0, n0s = 0, prev_st = 0, kt = 0, ti = 0;
            auto store_rounds = [&](int nrounds) {
                int n = 0;
                for (int c = done; c < done + nrounds; ++c) {
                    const int row = srow + 4 * c, m = m0s + row;
                    const bf16x8 v = *(const bf16x8*)(stg + stg_off(row, schk));
                    if (m < p.M) {
                        if constexpr (EPI == SAIS_EPI_BIAS_GELU_BF16) {
                            if (p.out2) *(bf16x8*)((bf16*)p.out2 + (size_t)m * p.ldo2 + n0s + 8 * schk) = v;
                            bf16x8 h;
#pragma unroll
                            for (int e = 0; e < 8; ++e) h[e] = (bf16)gelu_erf((float)v[e]);
                            *(bf16x8*)((bf16*)p.out + (size_t)m * p.ldo + n0s + 8 * schk) = h;
                        } else {
                            *(bf16x8*)((bf16*)p.out + (size_t)m * p.ldo + n0s + 8 * schk) = v;
                        }
                    }
                    ++n;
                }
                done += nrounds;
                return n;
            };
            const int per_round = (EPI == SAIS_EPI_BIAS_GELU_BF16 && p.out2) ? 2 : 1;
            for (int q = 0; q < total; ++q) {
                // K-tile q landed; the (at most 8) stores issued after its DMA may stay in flight
                switch (prev_st) {
                    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                    default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                }
                __builtin_amdgcn_s_barrier();                         // consumers done with K-tile q-1 (+ staging of the
                if (issued < total) { issue_next(); ++issued; }       //   previous tile complete when kt == 0)
                if (kt == 0 && ti > 0) {                              // previous tile's output is staged: new store job
                    const int tile = first + (ti - 1) * G;
                    m0s = (tile / p.ntn) * BM; n0s = (tile % p.ntn) * BN; done = 0;
                }
                prev_st = 0;
                if (done < 16 && kt < nk - 1) {                       // never in the tile's last K-tile: the consumers
                    const int r = (16 - done) < Q ? (16 - done) : Q;  //   overwrite the staging right after it
                    prev_st = store_rounds(r) * per_round;
                    if (prev_st == 3 || (prev_st > 4 && prev_st != 8)) prev_st = 0;   // unreachable with Q in {1,2,4}
                    if (m0s + BM > p.M) prev_st = 0;      // ragged last panel: some store instructions may be fully masked
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                if (++kt == nk) { kt = 0; ++ti; }
            }
            __builtin_amdgcn_s_barrier();                             // last tile staged
            if (cnt > 0) {
                const int tile = first + (cnt - 1) * G;
                m0s = (tile / p.ntn) * BM; n0s = (tile % p.ntn) * BN; done = 0;
                store_rounds(16);
            }
        }
        return;
    }

    // ---------------------------------------------------------------------- CONSUMER (wave w: rows 64 w .. 64 w + 63)
    const int g = lane >> 4, li = lane & 15;
    int stage = 0;
    for (int i = 0; i < cnt; ++i) {
        const int tile = first + i * G;
        const int m0 = (tile / p.ntn) * BM, n0 = (tile % p.ntn) * BN;
        f32x4 acc[4][8];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
        for (int kt = 0; kt < nk; ++kt) {
            __builtin_amdgcn_s_barrier();                             // K-tile landed for every loader
            const char* sa = smem + stage * STAGE;
            const char* sb = sa + A_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 fa[4], fb[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) fa[t] = *(const bf16x8*)(sa + swz(wid * 64 + t * 16 + li, ks * 4 + g));
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) fb[t] = *(const bf16x8*)(sb + swz(h * 64 + t * 16 + li, ks * 4 + g));
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) acc[mt][4 * h + t] = mfma16(fb[t], fa[mt], acc[mt][4 * h + t]);
                }
            }
            stage = stage == NSTAGE - 1 ? 0 : stage + 1;
        }
        // epilogue: lane holds, for row m0 + 64 w + 16 mt + li, the 32 columns n0 + 32 g + (4 nt + r), nt = 0..7
        const int n = n0 + 32 * g;
        float bias[32];
        if (p.bias) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x4 t = *(const f32x4*)(p.bias + n + 4 * c);
                bias[4 * c] = t[0]; bias[4 * c + 1] = t[1]; bias[4 * c + 2] = t[2]; bias[4 * c + 3] = t[3];
            }
        } else {
#pragma unroll
            for (int c = 0; c < 32; ++c) bias[c] = 0.f;
        }
        if constexpr (HANDOFF) {
            // arithmetic only; the bf16 row segment (32 columns = 4 chunks) goes to the staging buffer
            bf16x8 uaux[4][4];
            if constexpr (EPI == SAIS_EPI_DGELU_BF16 || EPI == SAIS_EPI_DRELU_BF16) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    int m = m0 + wid * 64 + mt * 16 + li;
                    m = m < p.M ? m : p.M - 1;
                    const bf16* u = (const bf16*)p.aux + (size_t)m * p.ldaux + n;
#pragma unroll
                    for (int c = 0; c < 4; ++c) uaux[mt][c] = *(const bf16x8*)(u + 8 * c);
                }
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int row = wid * 64 + mt * 16 + li;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    bf16x8 v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float y = acc[mt][2 * c + (e >> 2)][e & 3] + bias[8 * c + e];
                        if constexpr (EPI == SAIS_EPI_BIAS_RELU_BF16) y = fmaxf(y, 0.f);
                        if constexpr (EPI == SAIS_EPI_DGELU_BF16) y *= dgelu_erf((float)uaux[mt][c][e]);
                        if constexpr (EPI == SAIS_EPI_DRELU_BF16) y = (float)uaux[mt][c][e] > 0.f ? y : 0.f;
                        v[e] = (bf16)y;
                    }
                    *(bf16x8*)(stg + stg_off(row, 4 * g + c)) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // staged before the next barrier
        } else {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int m = m0 + wid * 64 + mt * 16 + li;
                if (m >= p.M) continue;
                float y[32];
#pragma unroll
                for (int nt = 0; nt < 8; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[4 * nt + r] = acc[mt][nt][r];
                epilogue_row<EPI>(p, m, n, y, bias);
            }
        }
    }
    if constexpr (HANDOFF) __builtin_amdgcn_s_barrier();              // tells the loaders the last tile is staged
}

template <int EPI>
int launch(const NtParams& p, hipStream_t stream) {
    constexpr bool HANDOFF = EPI == SAIS_EPI_BIAS_BF16 || EPI == SAIS_EPI_BIAS_RELU_BF16 || EPI == SAIS_EPI_BIAS_GELU_BF16 ||
                             EPI == SAIS_EPI_DGELU_BF16 || EPI == SAIS_EPI_DRELU_BF16;
    static thread_local bool ready[16] = {false};
    static thread_local int ncu[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return SAIS_ERR_LAUNCH;
    constexpr int lds = Cfg<HANDOFF>::NSTAGE * STAGE + (HANDOFF ? STG_BYTES : 0);
    if (p.K / BK < 2) return SAIS_ERR_ARG;
    if (!ready[dev]) {
        if (hipFuncSetAttribute((const void*)gemm_nt_ws_kernel<EPI, HANDOFF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
            hipSuccess)
            return SAIS_ERR_LAUNCH;
        ready[dev] = true;
    }
    if (!ncu[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return SAIS_ERR_LAUNCH;
        ncu[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int grid = p.ntiles < ncu[dev] ? p.ntiles : ncu[dev];
    hipLaunchKernelGGL((gemm_nt_ws_kernel<EPI, HANDOFF>), dim3(grid), dim3(512), lds, stream, p);
    return SAIS_OK;
}
}  // namespace

// internal (not part of the public ABI): called by sais_gemm_nt for large M
extern "C" int sais_gemm_nt_ws_(const SaisGemm* g, void* stream) {
    const int ntm = (g->M + BM - 1) / BM, ntn = g->N / BN;
    NtParams p{(const bf16*)g->A, (const bf16*)g->B, g->lda, g->ldb, g->M, g->N, g->K, g->bias,
               g->out, g->ldo, g->out2, g->ldo2, g->aux, g->ldaux, g->grp_in, g->grp_out, g->grp_off,
               ntm, ntn, ntm * ntn};
    hipStream_t s = (hipStream_t)stream;
    switch (g->epilogue) {
        case SAIS_EPI_BIAS_BF16: return launch<SAIS_EPI_BIAS_BF16>(p, s);
        case SAIS_EPI_BIAS_RELU_BF16: return launch<SAIS_EPI_BIAS_RELU_BF16>(p, s);
        case SAIS_EPI_BIAS_F32: return launch<SAIS_EPI_BIAS_F32>(p, s);
        case SAIS_EPI_BIAS_RESID_F32: return launch<SAIS_EPI_BIAS_RESID_F32>(p, s);
        case SAIS_EPI_BIAS_GELU_BF16: return launch<SAIS_EPI_BIAS_GELU_BF16>(p, s);
        case SAIS_EPI_DGELU_BF16: return launch<SAIS_EPI_DGELU_BF16>(p, s);
        case SAIS_EPI_DRELU_BF16: return launch<SAIS_EPI_DRELU_BF16>(p, s);
        case SAIS_EPI_PATCH_F32: return launch<SAIS_EPI_PATCH_F32>(p, s);
        default: return SAIS_ERR_ARG;
    }
}
