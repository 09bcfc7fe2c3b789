// LayerNorm forward / backward over D = 384 columns (gfx950).  HBM-bound row kernels:
// one row per half-wave (32 lanes x 3 x float4), all statistics in fp32, 16-B loads/stores.
//   forward : nn.LayerNorm in Block (vision_transformer.py:99,103,107-113; eps 1e-6 via vit_small :243-247),
//             the final self.norm (:212), and norm1/norm2 of the post-norm TransformerEncoderLayer
//             (prepare_model.py:74-81; eps 1e-5).
//   backward: autograd of the same.
#include "common.hpp"
#include "philox.hpp"
#include "../../include/sais_hip.h"

namespace {
constexpr int D = 384;

struct Row12 { float v[12]; };

DEVINL float half_sum(float v) {          // reduce over the 32 lanes of a half-wave
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

DEVINL int col_of(int l32, int i) { return 128 * (i >> 2) + 4 * l32 + (i & 3); }      // column of register i of lane l32
DEVINL void load_f32(const float* p, int l32, float (&v)[12]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        f32x4 t = *(const f32x4*)(p + 128 * i + 4 * l32);
        v[4 * i] = t[0]; v[4 * i + 1] = t[1]; v[4 * i + 2] = t[2]; v[4 * i + 3] = t[3];
    }
}
DEVINL void load_bf16(const bf16* p, int l32, float (&v)[12]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        bf16x4 t = *(const bf16x4*)(p + 128 * i + 4 * l32);
        v[4 * i] = (float)t[0]; v[4 * i + 1] = (float)t[1]; v[4 * i + 2] = (float)t[2]; v[4 * i + 3] = (float)t[3];
    }
}
DEVINL void store_f32(float* p, int l32, const float (&v)[12]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) *(f32x4*)(p + 128 * i + 4 * l32) = f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
}
DEVINL void store_bf16(bf16* p, int l32, const float (&v)[12]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        bf16x4 t;
        t[0] = (bf16)v[4 * i]; t[1] = (bf16)v[4 * i + 1]; t[2] = (bf16)v[4 * i + 2]; t[3] = (bf16)v[4 * i + 3];
        *(bf16x4*)(p + 128 * i + 4 * l32) = t;
    }
}

__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* x, long ldx, int rows, const float* gamma,
                                                     const float* beta, float eps, bf16* y16, long ldy16,
                                                     float* y32, long ldy32, float* mean, float* rstd) {
    const int l32 = threadIdx.x & 31;
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (row >= rows) return;
    float v[12], gm[12], bt[12];
    load_f32(x + (size_t)row * ldx, l32, v);
    load_f32(gamma, l32, gm);
    load_f32(beta, l32, bt);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += v[i];
    const float mu = half_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) { float d = v[i] - mu; q += d * d; }
    const float rs = rsqrtf(half_sum(q) * (1.0f / D) + eps);
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = (v[i] - mu) * rs * gm[i] + bt[i];
    if (y16) store_bf16(y16 + (size_t)row * ldy16, l32, v);
    if (y32) store_f32(y32 + (size_t)row * ldy32, l32, v);
    if (l32 == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
}

// dy = dy16 (bf16, optional) + dy32 (f32, optional); dx = dres (f32, optional) + LN'(dy)
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16* dy16, long lddy16, const float* dy32, long lddy32,
                                                     const float* x, long ldx, const float* mean, const float* rstd,
                                                     const float* gamma, const float* dres, long lddres, int rows,
                                                     float* dx32, long lddx32, bf16* dx16, long lddx16,
                                                     float* dgamma, float* dbeta, const float* rowscale16, float* dx32_drop,
                                                     float p_drop, const unsigned long long* rng, unsigned site) {
    __shared__ float red[2][8][D];
    const int l32 = threadIdx.x & 31, hw = threadIdx.x >> 5;
    float gm[12], ag[12], ab[12];
    load_f32(gamma, l32, gm);
#pragma unroll
    for (int i = 0; i < 12; ++i) { ag[i] = 0.f; ab[i] = 0.f; }
    for (int row = blockIdx.x * 8 + hw; row < rows; row += gridDim.x * 8) {
        float dy[12], xv[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) dy[i] = 0.f;
        if (dy16) load_bf16(dy16 + (size_t)row * lddy16, l32, dy);
        if (dy32) {
            float t[12];
            load_f32(dy32 + (size_t)row * lddy32, l32, t);
#pragma unroll
            for (int i = 0; i < 12; ++i) dy[i] += t[i];
        }
        load_f32(x + (size_t)row * ldx, l32, xv);
        const float mu = mean[row], rs = rstd[row];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            xv[i] = (xv[i] - mu) * rs;               // xhat
            ag[i] += dy[i] * xv[i];
            ab[i] += dy[i];
            dy[i] *= gm[i];                          // dy * gamma
            c1 += dy[i];
            c2 += dy[i] * xv[i];
        }
        c1 = half_sum(c1) * (1.0f / D);
        c2 = half_sum(c2) * (1.0f / D);
#pragma unroll
        for (int i = 0; i < 12; ++i) dy[i] = rs * (dy[i] - c1 - xv[i] * c2);
        if (dres) {
            float t[12];
            load_f32(dres + (size_t)row * lddres, l32, t);
#pragma unroll
            for (int i = 0; i < 12; ++i) dy[i] += t[i];
        }
        if (dx32) store_f32(dx32 + (size_t)row * lddx32, l32, dy);
        if (dx32_drop) {                                           // dropout backward of the branch this gradient enters
            const unsigned thr = drop_threshold(p_drop);
            const float inv = 1.0f / (1.0f - p_drop);
            float t[12];
#pragma unroll
            for (int i = 0; i < 12; ++i)
                t[i] = philox_keep(rng, site, (unsigned long long)row * D + col_of(l32, i), thr) ? dy[i] * inv : 0.f;
            store_f32(dx32_drop + (size_t)row * lddx32, l32, t);
        }
        if (dx16) {
            if (rowscale16) {                                      // DropPath: the next branch's backward sees s dx
                const float sc = rowscale16[row];
#pragma unroll
                for (int i = 0; i < 12; ++i) dy[i] *= sc;
            }
            store_bf16(dx16 + (size_t)row * lddx16, l32, dy);
        }
    }
    if (dgamma) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                red[0][hw][128 * i + 4 * l32 + j] = ag[4 * i + j];
                red[1][hw][128 * i + 4 * l32 + j] = ab[4 * i + j];
            }
        __syncthreads();
        for (int c = threadIdx.x; c < 2 * D; c += 256) {
            int which = c / D, col = c - which * D;
            float s = 0.f;
#pragma unroll
            for (int h = 0; h < 8; ++h) s += red[which][h][col];
            atomicAdd((which ? dbeta : dgamma) + col, s);
        }
    }
}
}  // namespace

extern "C" int sais_layernorm_fwd(const float* x, long ldx, int rows, int dim, const float* gamma, const float* beta,
                                  float eps, void* y_bf16, long ldy16, float* y_f32, long ldy32, float* mean,
                                  float* rstd, void* stream) {
    SAIS_ENTER();
    if (!x || !gamma || !beta || dim != D || rows <= 0 || (ldx & 3) || (ldy16 & 3) || (ldy32 & 3)) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(ln_fwd_kernel, dim3((rows + 7) / 8), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, gamma,
                       beta, eps, (bf16*)y_bf16, ldy16, y_f32, ldy32, mean, rstd);
    return sais_check_launch();
}

extern "C" int sais_layernorm_bwd(const void* dy_bf16, long lddy16, const float* dy_f32, long lddy32, const float* x,
                                  long ldx, const float* mean, const float* rstd, const float* gamma,
                                  const float* dres, long lddres, int rows, int dim, float* dx_f32, long lddx32,
                                  void* dx_bf16, long lddx16, float* dgamma, float* dbeta, const float* rowscale16,
                                  float* dx_f32_drop, float p_drop, const unsigned long long* rng_state, unsigned site,
                                  void* stream) {
    SAIS_ENTER();
    if (!x || !mean || !rstd || !gamma || dim != D || rows <= 0 || (!dy_bf16 && !dy_f32)) return SAIS_ERR_ARG;
    if ((dgamma == nullptr) != (dbeta == nullptr)) return SAIS_ERR_ARG;
    if (dx_f32_drop && (!rng_state || p_drop <= 0.f || p_drop >= 1.f)) return SAIS_ERR_ARG;
    int grid = (rows + 7) / 8;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)dy_bf16, lddy16,
                       dy_f32, lddy32, x, ldx, mean, rstd, gamma, dres, lddres, rows, dx_f32, lddx32, (bf16*)dx_bf16,
                       lddx16, dgamma, dbeta, rowscale16, dx_f32_drop, p_drop, rng_state, site);
    return sais_check_launch();
}
