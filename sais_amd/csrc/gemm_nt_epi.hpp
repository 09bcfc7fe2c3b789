// Shared by the NT GEMM kernels of gemm.hip and gemm_wstat.hip: launch parameters and the 8-columns-per-lane epilogue
// of the eight-wave (2 x 4 waves of 64 x 32) 128 x 128 tile.
#pragma once
#include "common.hpp"
#include "../../include/sais_hip.h"

namespace {

struct NtParams {
    const bf16* A; const bf16* B;
    int lda, ldb, M, N, K;
    const float* bias;
    void* out; int ldo;
    void* out2; int ldo2;
    const void* aux; int ldaux;
    int grp_in, grp_out, grp_off;
    const float* rowscale;          // SAIS_EPI_BIAS_RESID_F32, small-M kernel only (DropPath)
    // fp32 GEMM (temporal encoder) only: train-mode dropout fused into the epilogue, mask element = m * N + n
    float p_drop; const unsigned long long* rng; unsigned site;
};

// Eight-wave tile: the 128x128x64 tile and LDS image of gemm_nt_kernel, but 512 threads (2 x 4 waves of
// 64 x 32), so that each wave issues 4 instead of 8 LDS-DMA instructions per K-step (an issue holds the wave for
// 65-85 cycles) and four waves per SIMD (two workgroups per CU) cover each other's stalls.  Needs <= 128 VGPRs.
// Weight rows are permuted inside each 32-row panel (LDS row 16 t + 4 a + b <- panel column 8 a + 4 t + b) so that a
// lane owns 8 CONTIGUOUS output columns of a row: one 16-B bf16 store or two fp32 ones.
DEVINL int perm_row32(int n) { return (n & ~31) | (((n >> 2) & 3) << 3) | (((n >> 4) & 1) << 2) | (n & 3); }

struct EpiAux8 {
    f32x4 r[4][2];        // f32 aux: 8 columns x 4 sub-tiles
    bf16x8 u[4];          // bf16 aux
    u32x2 q[4];           // one-byte GELU' codes (SAIS_EPI_MULQ_BF16)
};

template <int EPI>
DEVINL void epilogue_loads8(const NtParams& p, int mbase, int li, int n, float (&b)[8], EpiAux8& a) {
    if (p.bias) {
        const f32x4 t0 = *(const f32x4*)(p.bias + n), t1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { b[i] = t0[i]; b[4 + i] = t1[i]; }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) b[i] = 0.f;
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        int m = mbase + mt * 16 + li;
        m = m < p.M ? m : p.M - 1;
        if constexpr (EPI == SAIS_EPI_BIAS_RESID_F32 || EPI == SAIS_EPI_PATCH_F32) {
            size_t row = m;
            if constexpr (EPI == SAIS_EPI_PATCH_F32) row = (m % p.grp_in) + p.grp_off;       // position row of the patch
            const float* r = (const float*)p.aux + row * p.ldaux + n;
            a.r[mt][0] = *(const f32x4*)r;
            a.r[mt][1] = *(const f32x4*)(r + 4);
        } else if constexpr (EPI == SAIS_EPI_DGELU_BF16 || EPI == SAIS_EPI_DRELU_BF16 || EPI == SAIS_EPI_MUL_BF16) {
            a.u[mt] = *(const bf16x8*)((const bf16*)p.aux + (size_t)m * p.ldaux + n);
        } else if constexpr (EPI == SAIS_EPI_MULQ_BF16) {
            a.q[mt] = *(const u32x2*)((const unsigned char*)p.aux + (size_t)m * p.ldaux + n);
        }
    }
}

template <int EPI>
DEVINL void epilogue8(const NtParams& p, int m, int n, const float (&v)[8], const float (&b)[8], const EpiAux8& a, int mt) {
    float y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = v[i] + b[i];
#ifndef SAIS_NT_STORE
#define SAIS_NT_STORE 0          // experiment switch: 1 = GELU' (read again only in backward) non-temporal, 2 = every bf16 output
#endif
    auto store_bf16 = [&](void* base, int ld, const float (&z)[8], int level = 2) {
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (bf16)z[i];
        bf16x8* dst = (bf16x8*)((bf16*)base + (size_t)m * ld + n);
        if (SAIS_NT_STORE >= level) __builtin_nontemporal_store(o, dst);
        else *dst = o;
    };
    auto store_f32 = [&](void* base, int ld, const float (&z)[8]) {
        float* o = (float*)base + (size_t)m * ld + n;
        *(f32x4*)o = f32x4{z[0], z[1], z[2], z[3]};
        *(f32x4*)(o + 4) = f32x4{z[4], z[5], z[6], z[7]};
    };
    if constexpr (EPI == SAIS_EPI_BIAS_BF16) {
        store_bf16(p.out, p.ldo, y);
    } else if constexpr (EPI == SAIS_EPI_BIAS_RELU_BF16) {
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] = fmaxf(y[i], 0.f);
        store_bf16(p.out, p.ldo, y);
    } else if constexpr (EPI == SAIS_EPI_BIAS_F32) {
        store_f32(p.out, p.ldo, y);
    } else if constexpr (EPI == SAIS_EPI_BIAS_RESID_F32) {
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] += a.r[mt][i >> 2][i & 3];
        store_f32(p.out, p.ldo, y);
        if (p.out2) store_bf16(p.out2, p.ldo2, y);
    } else if constexpr (EPI == SAIS_EPI_PATCH_F32) {
        // patch embedding: + position row, token row (m / grp_in) * grp_out + m % grp_in + grp_off of the [F, 197, 384] stream
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] += a.r[mt][i >> 2][i & 3];
        float* o = (float*)p.out + ((size_t)(m / p.grp_in) * p.grp_out + (m % p.grp_in) + p.grp_off) * p.ldo + n;
        *(f32x4*)o = f32x4{y[0], y[1], y[2], y[3]};
        *(f32x4*)(o + 4) = f32x4{y[4], y[5], y[6], y[7]};
    } else if constexpr (EPI == SAIS_EPI_BIAS_GELU_BF16) {
        if (p.out2) store_bf16(p.out2, p.ldo2, y);
        gelu_erf_n(y);
        store_bf16(p.out, p.ldo, y);
    } else if constexpr (EPI == SAIS_EPI_BIAS_GELU_GRAD_BF16) {
        float d[8];
        gelu_and_grad_n(y, d);
        store_bf16(p.out2, p.ldo2, d, 1);
        store_bf16(p.out, p.ldo, y);
    } else if constexpr (EPI == SAIS_EPI_BIAS_GELU_GRADQ_BF16) {
        float d[8];
        gelu_and_grad_n(y, d);
        *(u32x2*)((unsigned char*)p.out2 + (size_t)m * p.ldo2 + n) =
            u32x2{gq8_pack4(d[0], d[1], d[2], d[3]), gq8_pack4(d[4], d[5], d[6], d[7])};
        store_bf16(p.out, p.ldo, y);
    } else if constexpr (EPI == SAIS_EPI_MUL_BF16) {
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] *= (float)a.u[mt][i];
        store_bf16(p.out, p.ldo, y);
    } else if constexpr (EPI == SAIS_EPI_MULQ_BF16) {
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] *= gq8_decode(a.q[mt][i >> 2], i & 3);
        store_bf16(p.out, p.ldo, y);
    } else if constexpr (EPI == SAIS_EPI_DGELU_BF16) {
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            f32x2 g;
            dgelu_erf2(f32x2{(float)a.u[mt][i], (float)a.u[mt][i + 1]}, g);
            y[i] *= g.x, y[i + 1] *= g.y;
        }
        store_bf16(p.out, p.ldo, y);
    } else if constexpr (EPI == SAIS_EPI_DRELU_BF16) {
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] = (float)a.u[mt][i] > 0.f ? y[i] : 0.f;
        store_bf16(p.out, p.ldo, y);
    }
}


}  // namespace
