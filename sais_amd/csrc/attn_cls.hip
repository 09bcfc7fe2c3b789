// CLS-query attention for the LAST block of the DINO ViT (gfx950).
//
// VisionTransformer.forward returns norm(x)[:, 0] (dino-main/vision_transformer.py:209-214): of the last block's output only
// the CLS row of every frame is ever read.  Everything in that block that is row-local (proj, norm2, the MLP, both residual
// adds) therefore only has to run on the CLS rows, and of its attention (Attention.forward, :80-92) only the CLS QUERY is
// needed — keys and values of all tokens still are.  Outputs and every parameter gradient are unchanged: the rows that are
// not computed feed nothing, and the gradient that enters the block is zero outside the CLS rows.
//
//   forward :  out[f, h*64 .. ] = softmax(q_cls k^T / 8) v                     one wave per (frame, head)
//   backward:  dq_cls = dS k / 8,  dk = dS^T q_cls / 8,  dv = P^T dO  (P, dS: 1 x ntok rows, fp32 throughout; P is
//              recomputed, nothing is saved); the q part of dqkv is ZERO for the other rows and is written as such
//              (the buffer is reused between blocks).
//
// These are 2 x ntok x 64 MACs per problem: read- / store-bound (the forward reads K and V once, 77 MB at config 2; the
// backward also writes the whole [M, 1152] bf16 gradient, 116 MB), against 48 / 131 us for the full kernels they replace in
// this block.  Everything moves in whole 128-B rows, 8 rows per wave instruction; reductions are lane shuffles.
#include "common.hpp"
#include "../../include/sais_hip.h"

namespace {
constexpr int HD = 64, NH = 6, DM = 384;
constexpr int MAXTOK = 200;                      // 25 row groups of 8
constexpr int NI = MAXTOK / 8;
constexpr float SCALE = 0.125f;                  // 64^-0.5

// Lane (r = lane >> 3, c = lane & 7) owns the 16-B piece c (features 8 c .. 8 c + 7) of the rows 8 i + r: one wave
// instruction moves 8 whole 128-B rows of K or V, every dot product over the 64 features is a reduction over the 8 lanes of
// a row group (xor 1, 2, 4), every sum over the keys a reduction over the row groups (xor 8, 16, 32).  No LDS.
DEVINL float red_c(float v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); return v + __shfl_xor(v, 4); }
DEVINL float red_r(float v) { v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); return v + __shfl_xor(v, 32); }
DEVINL float max_r(float v) { v = fmaxf(v, __shfl_xor(v, 8)); v = fmaxf(v, __shfl_xor(v, 16)); return fmaxf(v, __shfl_xor(v, 32)); }
DEVINL float dot8(const bf16x8& a, const float (&b)[8]) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += (float)a[e] * b[e];
    return s;
}
DEVINL void unpack8(const bf16x8& a, float (&b)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) b[e] = (float)a[e];
}

// P of the CLS query (unnormalised exp, 25 keys per lane: key 8 i + r) and 1 / sum; optionally keeps the K pieces
template <bool KEEPK>
DEVINL void cls_softmax(const bf16* qkv, long ld, int row0, int ntok, int h, int r, int c, const float (&q)[8],
                        bf16x8 (&kf)[NI], float (&p)[NI], float& inv) {
    const bf16* kp = qkv + (size_t)row0 * ld + DM + h * HD + 8 * c;
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int k = 8 * i + r;
        const bf16x8 v = *(const bf16x8*)(kp + (size_t)(k < ntok ? k : ntok - 1) * ld);
        if (KEEPK) kf[i] = v;
        const float sc = red_c(dot8(v, q)) * SCALE;
        p[i] = k < ntok ? sc : -INFINITY;
        mx = fmaxf(mx, p[i]);
    }
    mx = max_r(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) { p[i] = __expf(p[i] - mx); sum += p[i]; }      // exp(-inf) = 0 past the last token
    inv = 1.0f / red_r(sum);
}

__global__ __launch_bounds__(256) void attn_cls_fwd_kernel(const bf16* qkv, long ld, int frames, int ntok, bf16* out, long ldo) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int prob = blockIdx.x * 4 + w;
    if (prob >= frames * NH) return;
    const int f = prob / NH, h = prob - f * NH, row0 = f * ntok, r = lane >> 3, c = lane & 7;
    float q[8];
    unpack8(*(const bf16x8*)(qkv + (size_t)row0 * ld + h * HD + 8 * c), q);
    bf16x8 kf[NI];
    float p[NI], inv;
    cls_softmax<false>(qkv, ld, row0, ntok, h, r, c, q, kf, p, inv);
    const bf16* vp = qkv + (size_t)row0 * ld + 2 * DM + h * HD + 8 * c;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int k = 8 * i + r;
        const bf16x8 v = *(const bf16x8*)(vp + (size_t)(k < ntok ? k : ntok - 1) * ld);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += p[i] * (float)v[e];                // p = 0 past the last token
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16)(red_r(acc[e]) * inv);
    if (r == 0) *(bf16x8*)(out + (size_t)f * ldo + h * HD + 8 * c) = o;
}

__global__ __launch_bounds__(256, 2) void attn_cls_bwd_kernel(const bf16* qkv, long ld, const bf16* dout, long lddo, int frames,
                                                           int ntok, bf16* dqkv, long lddq) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int prob = blockIdx.x * 4 + w;
    if (prob >= frames * NH) return;
    const int f = prob / NH, h = prob - f * NH, row0 = f * ntok, r = lane >> 3, c = lane & 7;
    float q[8], dO[8];
    unpack8(*(const bf16x8*)(qkv + (size_t)row0 * ld + h * HD + 8 * c), q);
    unpack8(*(const bf16x8*)(dout + (size_t)f * lddo + h * HD + 8 * c), dO);
    bf16x8 kf[NI];
    float p[NI], inv;
    cls_softmax<true>(qkv, ld, row0, ntok, h, r, c, q, kf, p, inv);
    // dP_k = dO . v_k, delta = sum_k P_k dP_k, dS_k = P_k (dP_k - delta) / 8
    const bf16* vp = qkv + (size_t)row0 * ld + 2 * DM + h * HD + 8 * c;
    float dP[NI], delta = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int k = 8 * i + r;
        const bf16x8 v = *(const bf16x8*)(vp + (size_t)(k < ntok ? k : ntok - 1) * ld);
        p[i] *= inv;
        dP[i] = red_c(dot8(v, dO));
        delta += p[i] * dP[i];
    }
    delta = red_r(delta);
    // row k of dqkv: dq (zero unless k = 0: written as zeros, the buffer is reused between blocks), dk = dS_k q, dv = P_k dO
    bf16* op = dqkv + (size_t)row0 * lddq + h * HD + 8 * c;
    float dq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bf16x8 zero;
#pragma unroll
    for (int e = 0; e < 8; ++e) zero[e] = (bf16)0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int k = 8 * i + r;
        const float ds = p[i] * (dP[i] - delta) * SCALE;
        bf16x8 dk, dv;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            dq[e] += ds * (float)kf[i][e];                                       // ds = 0 past the last token (p = 0)
            dk[e] = (bf16)(ds * q[e]);
            dv[e] = (bf16)(p[i] * dO[e]);
        }
        if (k < ntok) {
            bf16* o = op + (size_t)k * lddq;
            if (k) *(bf16x8*)o = zero;
            *(bf16x8*)(o + DM) = dk;
            *(bf16x8*)(o + 2 * DM) = dv;
        }
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16)red_r(dq[e]);
    if (r == 0) *(bf16x8*)op = o;
}
}  // namespace

extern "C" int sais_vit_attn_cls_fwd(const void* qkv, long ldqkv, int frames, int ntok, void* out, long ldo, void* stream) {
    SAIS_ENTER();
    if (!qkv || !out || frames <= 0 || ntok <= 0 || ntok > MAXTOK || (ldqkv & 7) || (ldo & 7)) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(attn_cls_fwd_kernel, dim3((frames * NH + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)qkv,
                       ldqkv, frames, ntok, (bf16*)out, ldo);
    return sais_check_launch();
}

extern "C" int sais_vit_attn_cls_bwd(const void* qkv, long ldqkv, const void* dout, long lddo, int frames, int ntok,
                                     void* dqkv, long lddqkv, void* stream) {
    SAIS_ENTER();
    if (!qkv || !dout || !dqkv || frames <= 0 || ntok <= 0 || ntok > MAXTOK || (ldqkv & 7) || (lddo & 7) || (lddqkv & 7)) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(attn_cls_bwd_kernel, dim3((frames * NH + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)qkv,
                       ldqkv, (const bf16*)dout, lddo, frames, ntok, (bf16*)dqkv, lddqkv);
    return sais_check_launch();
}
