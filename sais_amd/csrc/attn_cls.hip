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
// These are 2 x ntok x 64 MACs per problem: latency- and store-bound (the backward writes the whole [M, 1152] bf16 gradient,
// 116 MB at config 2), against 131 us for the full single-pass backward it replaces in this block.
#include "common.hpp"
#include "../../include/sais_hip.h"

namespace {
constexpr int HD = 64, NH = 6, DM = 384;
constexpr int MAXTOK = 256;
constexpr float SCALE = 0.125f;                  // 64^-0.5

// scores of the CLS query against keys lane, lane + 64, ...: lane-per-key, each lane reads whole 128-B K rows
DEVINL void cls_scores(const bf16* qkv, long ld, int row0, int ntok, int h, int lane, float (&s)[MAXTOK / 64], float& mx) {
    const bf16* qp = qkv + (size_t)row0 * ld + h * HD;
    float q[HD];
#pragma unroll
    for (int c = 0; c < HD / 8; ++c) {
        const bf16x8 v = *(const bf16x8*)(qp + 8 * c);              // same address in every lane: a broadcast load
#pragma unroll
        for (int e = 0; e < 8; ++e) q[8 * c + e] = (float)v[e];
    }
    mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXTOK / 64; ++i) {
        const int k = lane + 64 * i;
        float a = -INFINITY;
        if (k < ntok) {
            const bf16* kp = qkv + (size_t)(row0 + k) * ld + DM + h * HD;
            a = 0.f;
#pragma unroll
            for (int c = 0; c < HD / 8; ++c) {
                const bf16x8 v = *(const bf16x8*)(kp + 8 * c);
#pragma unroll
                for (int e = 0; e < 8; ++e) a += q[8 * c + e] * (float)v[e];
            }
            a *= SCALE;
        }
        s[i] = a;
        mx = fmaxf(mx, a);
    }
    mx = wave_max(mx);
}

__global__ __launch_bounds__(256) void attn_cls_fwd_kernel(const bf16* qkv, long ld, int frames, int ntok, bf16* out, long ldo) {
    __shared__ float sp[4][MAXTOK];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int prob = blockIdx.x * 4 + w;
    if (prob >= frames * NH) return;
    const int f = prob / NH, h = prob - f * NH, row0 = f * ntok;
    float s[MAXTOK / 64], mx;
    cls_scores(qkv, ld, row0, ntok, h, lane, s, mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXTOK / 64; ++i) {
        const float p = lane + 64 * i < ntok ? __expf(s[i] - mx) : 0.f;
        sp[w][lane + 64 * i] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    // out[d = lane] = sum_k p_k v[k][d]: one coalesced 128-B row of V per key, four independent partial sums
    const bf16* vp = qkv + (size_t)row0 * ld + 2 * DM + h * HD + lane;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 4 <= ntok; k += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += sp[w][k + j] * (float)vp[(size_t)(k + j) * ld];
    }
    for (; k < ntok; ++k) acc[0] += sp[w][k] * (float)vp[(size_t)k * ld];
    out[(size_t)f * ldo + h * HD + lane] = (bf16)(((acc[0] + acc[1]) + (acc[2] + acc[3])) * inv);
}

__global__ __launch_bounds__(256) void attn_cls_bwd_kernel(const bf16* qkv, long ld, const bf16* dout, long lddo, int frames,
                                                           int ntok, bf16* dqkv, long lddq) {
    __shared__ float sp[4][MAXTOK], sds[4][MAXTOK];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int prob = blockIdx.x * 4 + w;
    if (prob >= frames * NH) return;
    const int f = prob / NH, h = prob - f * NH, row0 = f * ntok;
    float s[MAXTOK / 64], mx;
    cls_scores(qkv, ld, row0, ntok, h, lane, s, mx);
    float p[MAXTOK / 64], sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXTOK / 64; ++i) {
        p[i] = lane + 64 * i < ntok ? __expf(s[i] - mx) : 0.f;
        sum += p[i];
    }
    const float inv = 1.0f / wave_sum(sum);
    // dP_k = dO . v_k (lane per key), delta = sum_k P_k dP_k, dS_k = P_k (dP_k - delta)
    const bf16* dop = dout + (size_t)f * lddo + h * HD;
    float dO[HD];
#pragma unroll
    for (int c = 0; c < HD / 8; ++c) {
        const bf16x8 v = *(const bf16x8*)(dop + 8 * c);
#pragma unroll
        for (int e = 0; e < 8; ++e) dO[8 * c + e] = (float)v[e];
    }
    float dP[MAXTOK / 64], delta = 0.f;
#pragma unroll
    for (int i = 0; i < MAXTOK / 64; ++i) {
        const int k = lane + 64 * i;
        float a = 0.f;
        p[i] *= inv;
        if (k < ntok) {
            const bf16* vp = qkv + (size_t)(row0 + k) * ld + 2 * DM + h * HD;
#pragma unroll
            for (int c = 0; c < HD / 8; ++c) {
                const bf16x8 v = *(const bf16x8*)(vp + 8 * c);
#pragma unroll
                for (int e = 0; e < 8; ++e) a += dO[8 * c + e] * (float)v[e];
            }
        }
        dP[i] = a;
        delta += p[i] * a;
    }
    delta = wave_sum(delta);
#pragma unroll
    for (int i = 0; i < MAXTOK / 64; ++i) {
        sp[w][lane + 64 * i] = p[i];
        sds[w][lane + 64 * i] = p[i] * (dP[i] - delta) * SCALE;       // d(q . k) = dS / 8
    }
    // lane = feature d from here on: dk[k][d] = dS_k q[d], dv[k][d] = P_k dO[d], dq[d] = sum_k dS_k k[k][d]
    const float qd = (float)qkv[(size_t)row0 * ld + h * HD + lane];
    const float dod = (float)dop[lane];
    const bf16* kp = qkv + (size_t)row0 * ld + DM + h * HD + lane;
    bf16* o = dqkv + (size_t)row0 * lddq + h * HD + lane;
    float dq[4] = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 4 <= ntok; k += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float ds = sds[w][k + j];
            dq[j] += ds * (float)kp[(size_t)(k + j) * ld];
            bf16* r = o + (size_t)(k + j) * lddq;
            if (k + j) r[0] = (bf16)0.f;                               // q rows other than CLS receive no gradient
            r[DM] = (bf16)(ds * qd);
            r[2 * DM] = (bf16)(sp[w][k + j] * dod);
        }
    }
    for (; k < ntok; ++k) {
        const float ds = sds[w][k];
        dq[0] += ds * (float)kp[(size_t)k * ld];
        bf16* r = o + (size_t)k * lddq;
        if (k) r[0] = (bf16)0.f;
        r[DM] = (bf16)(ds * qd);
        r[2 * DM] = (bf16)(sp[w][k] * dod);
    }
    o[0] = (bf16)((dq[0] + dq[1]) + (dq[2] + dq[3]));
}
}  // namespace

extern "C" int sais_vit_attn_cls_fwd(const void* qkv, long ldqkv, int frames, int ntok, void* out, long ldo, void* stream) {
    SAIS_ENTER();
    if (!qkv || !out || frames <= 0 || ntok <= 0 || ntok > MAXTOK || (ldqkv & 7)) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(attn_cls_fwd_kernel, dim3((frames * NH + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)qkv,
                       ldqkv, frames, ntok, (bf16*)out, ldo);
    return sais_check_launch();
}

extern "C" int sais_vit_attn_cls_bwd(const void* qkv, long ldqkv, const void* dout, long lddo, int frames, int ntok,
                                     void* dqkv, long lddqkv, void* stream) {
    SAIS_ENTER();
    if (!qkv || !dout || !dqkv || frames <= 0 || ntok <= 0 || ntok > MAXTOK || (ldqkv & 7) || (lddo & 7)) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(attn_cls_bwd_kernel, dim3((frames * NH + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)qkv,
                       ldqkv, (const bf16*)dout, lddo, frames, ntok, (bf16*)dqkv, lddqkv);
    return sais_check_launch();
}
