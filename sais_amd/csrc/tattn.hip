// Masked multi-head self-attention of the temporal TransformerEncoder (4 heads x 96, S = T + 1 <= 96 tokens per sequence):
// the nn.MultiheadAttention core inside the torch-1.8 post-norm TransformerEncoderLayer, with key_padding_mask, train-mode
// dropout on the attention weights and the README.md:43-48 head-averaged attention-map return
// (prepare_model.py:74-81,197-221).  Forward and backward, exact fp32 on the matrix cores.
//
// Round 2 computed every dot product with VALU FMAs out of LDS (two LDS reads per FMA: 16x below the VALU rate, 15 / 28 us
// per layer for 33 x 33 x 96 problems).  Here every product runs on v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate:
// bit-for-bit an fmaf chain, at the fp32 VALU peak rate but with one LDS dword per 32 flop).  One 256-thread workgroup per
// (sequence, head); q, k, v (and dctx) of the head sit in LDS with 100-float rows.
//
// Operand trick: the 16x16x4 MFMA sums over 4 "k slots" (lane >> 4).  The summation order is free, so slot g of step s is
// mapped to feature d = 24 g + s: a lane then needs 24 CONTIGUOUS floats of its row (six ds_read_b128, conflict-free with the
// 100-float stride) for the whole 96-deep dot product instead of 24 scattered dwords.
// Orientation: scores are computed transposed, S^T[key][query] = K Q^T, so a lane owns one QUERY (column) and 4 keys per
// 16-key tile: softmax is an in-lane loop plus two shuffles, and P^T is already the B operand of ctx^T = V^T P^T (slot g of
// step r <-> key 4 g + r) with no lane movement.  The backward needs P and dS with the KEY on the lane as well (dV, dK sum
// over queries): the waves re-own key tiles after a barrier and rebuild both from the per-query statistics left in LDS.
#include "common.hpp"
#include "philox.hpp"
#include "../../include/sais_hip.h"

namespace {
constexpr int D = 384, TH = 4, THD = 96, TLD = 100;       // TLD: floats per LDS row

DEVINL f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// row[24 g .. 24 g + 23] -> registers (times `mul`)
DEVINL void load24(const float* row, int g, float (&f)[24], float mul = 1.0f) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const f32x4 t = *(const f32x4*)(row + 24 * g + 4 * i);
        f[4 * i] = t[0] * mul; f[4 * i + 1] = t[1] * mul; f[4 * i + 2] = t[2] * mul; f[4 * i + 3] = t[3] * mul;
    }
}
// T[i][j] = sum_d X[i][d] Y[j][d] for the 16 x 16 tile whose rows i / columns j are the rows `x` / `y` were loaded from:
// lane (li, g) register r holds T[4 g + r][li]
DEVINL f32x4 dot_tile(const float (&x)[24], const float (&y)[24]) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 24; ++s) acc = mfma4(x[s], y[s], acc);
    return acc;
}
// two tiles at once: the two accumulator chains are independent, so the MFMAs issue back to back (a single chain waits
// out the 40-cycle dependent latency of the 32-cycle instruction)
DEVINL void dot_tile2(const float (&x0)[24], const float (&y0)[24], const float (&x1)[24], const float (&y1)[24], f32x4& a0,
                      f32x4& a1) {
    a0 = f32x4{0.f, 0.f, 0.f, 0.f};
    a1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 24; ++s) {
        a0 = mfma4(x0[s], y0[s], a0);
        a1 = mfma4(x1[s], y1[s], a1);
    }
}
DEVINL float max4g(float v) { v = fmaxf(v, __shfl_xor(v, 16)); return fmaxf(v, __shfl_xor(v, 32)); }
DEVINL float sum4g(float v) { v += __shfl_xor(v, 16); return v + __shfl_xor(v, 32); }

// head slices q, k, v of qkv [B*S, 1152] -> sQ / sK / sV [S_pad][TLD], rows >= S zero.  Addresses are clamped instead of
// branched on and a lane issues the loads of SU consecutive passes (3 SU float4) before the first LDS write: the
// `if (s < S) load` loop this replaces compiled to one dependent round trip per pass and matrix (~15 per launch for S = 33,
// each ~0.5 us of a 16-us kernel).
constexpr int SU = 3;
DEVINL void stage_qkv(const float* qkv, int b, int h, int S, int Spad, float* sQ, float* sK, float* sV, int tid) {
    const int total = Spad * (THD / 4);
    for (int i0 = tid; i0 < total; i0 += 256 * SU) {
        f32x4 v[SU][3];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int i = min(i0 + 256 * u, total - 1), s = i / (THD / 4), c4 = i % (THD / 4);
            const float* src = qkv + ((size_t)b * S + min(s, S - 1)) * (3 * D) + h * THD + 4 * c4;
#pragma unroll
            for (int w = 0; w < 3; ++w) v[u][w] = *(const f32x4*)(src + w * D);
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int i = i0 + 256 * u;
            if (i < total) {
                const int s = i / (THD / 4), c4 = i % (THD / 4);
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                *(f32x4*)(sQ + s * TLD + 4 * c4) = s < S ? v[u][0] : z;
                *(f32x4*)(sK + s * TLD + 4 * c4) = s < S ? v[u][1] : z;
                *(f32x4*)(sV + s * TLD + 4 * c4) = s < S ? v[u][2] : z;
            }
        }
    }
}
// key flags of sequence b into LDS: 1 = masked (key_padding_mask) or past the sequence; read back per (key tile, lane)
// without the serialised global byte loads the `key >= S || pad[key]` form compiled to
DEVINL void stage_pad(const unsigned char* pad, int S, int Spad, unsigned char* sP, int tid) {
    if (tid < Spad) sP[tid] = tid >= S ? (unsigned char)1 : pad[tid];
}

constexpr int MAXT = 6;                                  // 16-token tiles: S <= 96

// Train mode (p > 0): the attention weights are dropped AFTER the softmax and BEFORE P v, and the returned map is the
// dropped one (torch-1.8 F.multi_head_attention_forward).  Mask element index: ((b * 4 + h) * S + i) * S + j.
__global__ __launch_bounds__(256) void tattn_fwd_kernel(const float* qkv, const unsigned char* key_pad, int S, float* ctx,
                                                       float* attn_avg, float p_drop, const unsigned long long* rng,
                                                       unsigned sid) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nt = (S + 15) >> 4, Spad = nt * 16;
    float* sQ = (float*)smem;
    float* sK = sQ + Spad * TLD;
    float* sV = sK + Spad * TLD;
    unsigned char* sP = (unsigned char*)(sV + Spad * TLD);
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    stage_pad(key_pad + (size_t)b * S, S, Spad, sP, tid);
    stage_qkv(qkv, b, h, S, Spad, sQ, sK, sV, tid);
    __syncthreads();
    const bool dropping = p_drop > 0.f;
    const unsigned thr = drop_threshold(p_drop);
    const float inv_keep = dropping ? 1.0f / (1.0f - p_drop) : 1.0f;
    const unsigned long long base = ((unsigned long long)b * TH + h) * S * S;
    for (int qt = wid; qt < nt; qt += 4) {
        const int q = 16 * qt + li;                      // this lane's query (column of S^T)
        float qf[24];
        load24(sQ + q * TLD, g, qf, 0.10206207261596577f /* 96^-0.5 */);
        f32x4 p[MAXT];
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < MAXT; kt += 2) {           // two key tiles per pass: two independent accumulator chains
            if (kt >= nt) break;
            float kf[24], kg[24];
            load24(sK + (16 * kt + li) * TLD, g, kf);
            if (kt + 1 < nt) {
                load24(sK + (16 * (kt + 1) + li) * TLD, g, kg);
                dot_tile2(kf, qf, kg, qf, p[kt], p[kt + 1]);          // register r: key 16 kt + 4 g + r, query q
            } else {
                p[kt] = dot_tile(kf, qf);
            }
        }
#pragma unroll
        for (int kt = 0; kt < MAXT; ++kt) {
            if (kt >= nt) break;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (sP[16 * kt + 4 * g + r]) p[kt][r] = -INFINITY;
                m = fmaxf(m, p[kt][r]);
            }
        }
        m = max4g(m);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < MAXT; ++kt) {
            if (kt >= nt) break;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = __expf(p[kt][r] - m); p[kt][r] = e; sum += e; }
        }
        const float inv = 1.0f / sum4g(sum);
#pragma unroll
        for (int kt = 0; kt < MAXT; ++kt) {
            if (kt >= nt) break;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * kt + 4 * g + r;
                float v = p[kt][r] * inv;
                if (dropping && q < S && key < S)
                    v = philox_keep(rng, sid, base + (unsigned long long)q * S + key, thr) ? v * inv_keep : 0.f;
                p[kt][r] = v;
                if (attn_avg && q < S && key < S) atomicAdd(attn_avg + ((size_t)b * S + q) * S + key, v * (1.0f / TH));
            }
        }
        // ctx^T[d][query] = sum_key v[key][d] P^T[key][query]: six independent accumulator chains (one per 16 features)
        f32x4 acc[THD / 16];
#pragma unroll
        for (int dt = 0; dt < THD / 16; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < MAXT; ++kt) {
            if (kt >= nt) break;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* vrow = sV + (16 * kt + 4 * g + r) * TLD + li;
#pragma unroll
                for (int dt = 0; dt < THD / 16; ++dt) acc[dt] = mfma4(vrow[16 * dt], p[kt][r], acc[dt]);
            }
        }
        if (q < S) {
#pragma unroll
            for (int dt = 0; dt < THD / 16; ++dt)
                *(f32x4*)(ctx + ((size_t)b * S + q) * D + h * THD + 16 * dt + 4 * g) = acc[dt];
        }
    }
}

// With dropout: ctx = P' v, P' = P m / (1 - p).  dV = P'^T dctx; dP = (dctx v^T) m / (1 - p); dS = P (dP - rowsum(P dP)) scale;
// dq = dS k; dk = dS^T q.  dctx = sum of nslab raw split-K slabs (slab_stride floats apart) of the out_proj dX GEMM.
__global__ __launch_bounds__(256) void tattn_bwd_kernel(const float* qkv, const unsigned char* key_pad, int S,
                                                       const float* dctx, int nslab, long slab_stride, float* dqkv,
                                                       float p_drop, const unsigned long long* rng, unsigned sid) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nt = (S + 15) >> 4, Spad = nt * 16;
    float* sQ = (float*)smem;
    float* sK = sQ + Spad * TLD;
    float* sV = sK + Spad * TLD;
    float* sG = sV + Spad * TLD;                          // dctx
    float* sM = sG + Spad * TLD;                          // per query: row max, 1 / row sum, rowsum(P dP)
    float* sI = sM + Spad;
    float* sDot = sI + Spad;
    unsigned char* sP = (unsigned char*)(sDot + Spad);
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const float scale = 0.10206207261596577f;
    stage_pad(key_pad + (size_t)b * S, S, Spad, sP, tid);
    {   // dctx = sum of the out_proj dX slabs: the loads of SU passes x one slab in flight together
        const int total = Spad * (THD / 4);
        for (int i0 = tid; i0 < total; i0 += 256 * SU) {
            f32x4 v[SU];
            const float* src[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int i = min(i0 + 256 * u, total - 1), sr = i / (THD / 4), c4 = i % (THD / 4);
                src[u] = dctx + ((size_t)b * S + min(sr, S - 1)) * D + h * THD + 4 * c4;
                v[u] = *(const f32x4*)src[u];
            }
            for (int z = 1; z < nslab; ++z) {
#pragma unroll
                for (int u = 0; u < SU; ++u) v[u] += *(const f32x4*)(src[u] + (size_t)z * slab_stride);
            }
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int i = i0 + 256 * u;
                if (i < total) {
                    const int sr = i / (THD / 4), c4 = i % (THD / 4);
                    *(f32x4*)(sG + sr * TLD + 4 * c4) = sr < S ? v[u] : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
    }
    stage_qkv(qkv, b, h, S, Spad, sQ, sK, sV, tid);
    __syncthreads();
    const bool dropping = p_drop > 0.f;
    const unsigned thr = drop_threshold(p_drop);
    const float inv_keep = dropping ? 1.0f / (1.0f - p_drop) : 1.0f;
    const unsigned long long base = ((unsigned long long)b * TH + h) * S * S;
    auto keep_of = [&](int q, int key) {
        return (!dropping || philox_keep(rng, sid, base + (unsigned long long)q * S + key, thr)) ? inv_keep : 0.f;
    };

    // ---- phase 1: a wave owns 16 queries (on the lane), all keys: statistics, dS^T, dq
    for (int qt = wid; qt < nt; qt += 4) {
        const int q = 16 * qt + li;
        float qf[24], gf[24];
        load24(sQ + q * TLD, g, qf, scale);
        load24(sG + q * TLD, g, gf);
        f32x4 p[MAXT], dp[MAXT];
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < MAXT; ++kt) {
            if (kt >= nt) break;
            float kf[24], vf[24];
            load24(sK + (16 * kt + li) * TLD, g, kf);
            load24(sV + (16 * kt + li) * TLD, g, vf);
            dot_tile2(kf, qf, vf, gf, p[kt], dp[kt]);    // scores^T and dP'[query][key] = dctx_q . v_key (key 4 g + r on the registers)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * kt + 4 * g + r;
                if (sP[key]) p[kt][r] = -INFINITY;
                m = fmaxf(m, p[kt][r]);
            }
        }
        m = max4g(m);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < MAXT; ++kt) {
            if (kt >= nt) break;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = __expf(p[kt][r] - m); p[kt][r] = e; sum += e; }
        }
        const float inv = 1.0f / sum4g(sum);
        float dot = 0.f;
#pragma unroll
        for (int kt = 0; kt < MAXT; ++kt) {
            if (kt >= nt) break;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * kt + 4 * g + r;
                const float pv = p[kt][r] * inv;
                const float kp = (q < S && key < S) ? keep_of(q, key) : 0.f;
                p[kt][r] = pv;
                dp[kt][r] *= kp;                         // dP = dP' m / (1 - p)
                dot += pv * dp[kt][r];
            }
        }
        dot = sum4g(dot);
        if (g == 0) { sM[q] = m; sI[q] = inv; sDot[q] = dot; }
#pragma unroll
        for (int kt = 0; kt < MAXT; ++kt) {
            if (kt >= nt) break;
#pragma unroll
            for (int r = 0; r < 4; ++r) p[kt][r] = p[kt][r] * (dp[kt][r] - dot) * scale;      // dS^T[key][query]
        }
        // dq^T[d][query] = sum_key k[key][d] dS^T[key][query]
        f32x4 acc[THD / 16];
#pragma unroll
        for (int dt = 0; dt < THD / 16; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < MAXT; ++kt) {
            if (kt >= nt) break;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* krow = sK + (16 * kt + 4 * g + r) * TLD + li;
#pragma unroll
                for (int dt = 0; dt < THD / 16; ++dt) acc[dt] = mfma4(krow[16 * dt], p[kt][r], acc[dt]);
            }
        }
        if (q < S) {
#pragma unroll
            for (int dt = 0; dt < THD / 16; ++dt)
                *(f32x4*)(dqkv + ((size_t)b * S + q) * (3 * D) + h * THD + 16 * dt + 4 * g) = acc[dt];
        }
    }
    __syncthreads();

    // ---- phase 2: a wave owns 16 keys (on the lane), all queries: P' and dS with the key on the lane -> dV, dK
    for (int kt = wid; kt < nt; kt += 4) {
        const int key = 16 * kt + li;
        const bool key_ok = !sP[key];
        float kf[24], vf[24];
        load24(sK + key * TLD, g, kf);
        load24(sV + key * TLD, g, vf);
        f32x4 adv[THD / 16], adk[THD / 16];
#pragma unroll
        for (int dt = 0; dt < THD / 16; ++dt) { adv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; adk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        for (int qt = 0; qt < nt; ++qt) {
            float xf[24], yf[24];
            load24(sQ + (16 * qt + li) * TLD, g, xf, scale);
            load24(sG + (16 * qt + li) * TLD, g, yf);
            f32x4 s, dpp;                                // register r: query 16 qt + 4 g + r, key on the lane
            dot_tile2(xf, kf, yf, vf, s, dpp);           // scores and dP'[query][key]
            const f32x4 m4 = *(const f32x4*)(sM + 16 * qt + 4 * g);
            const f32x4 i4 = *(const f32x4*)(sI + 16 * qt + 4 * g);
            const f32x4 d4 = *(const f32x4*)(sDot + 16 * qt + 4 * g);
            f32x4 pp, ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = 16 * qt + 4 * g + r;
                const bool ok = key_ok && q < S;
                const float pv = ok ? __expf(s[r] - m4[r]) * i4[r] : 0.f;
                const float kp = ok ? keep_of(q, key) : 0.f;
                pp[r] = pv * kp;                                              // P'
                ds[r] = ok ? pv * (dpp[r] * kp - d4[r]) * scale : 0.f;        // dS
            }
            // dV^T[d][key] += sum_q dctx[q][d] P'[q][key] ;  dK^T[d][key] += sum_q q[q][d] dS[q][key]   (slot g of step r <-> query 4 g + r)
#pragma unroll
            for (int dt = 0; dt < THD / 16; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = (16 * qt + 4 * g + r) * TLD + 16 * dt + li;
                    adv[dt] = mfma4(sG[row], pp[r], adv[dt]);
                    adk[dt] = mfma4(sQ[row], ds[r], adk[dt]);
                }
        }
        if (key < S) {
            float* o = dqkv + ((size_t)b * S + key) * (3 * D) + D + h * THD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < THD / 16; ++dt) {
                *(f32x4*)(o + 16 * dt) = adk[dt];
                *(f32x4*)(o + D + 16 * dt) = adv[dt];
            }
        }
    }
}

// raise the dynamic-LDS limit of a kernel once per process and device (not per call: keeps the launch
// path free of runtime-API calls so it can be captured into a hipGraph); the limit only ever grows.
template <typename K>
int set_lds(K kernel, int bytes) {
    static thread_local int granted[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return SAIS_ERR_LAUNCH;
    if (bytes <= granted[dev]) return SAIS_OK;
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
        return SAIS_ERR_LAUNCH;
    granted[dev] = bytes;
    return SAIS_OK;
}

// zero-fill as a KERNEL node: the head-averaged map is accumulated with atomics over the four heads.  (Round 6: as a
// hipMemsetAsync node inside a captured graph the fill was right on the first replay and left garbage in the map on later ones —
// tools/scratch/win_dbg3.py, the opt-in hipGraph form of run_windows; a kernel node replays like every other launch.)
__global__ __launch_bounds__(256) void zero_f32_kernel(float* p, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) p[i] = 0.f;
}
}  // namespace

extern "C" int sais_temporal_attn_fwd(const float* qkv, const unsigned char* key_pad, int B, int S, float* ctx,
                                      float* attn_avg, float p_drop, const unsigned long long* rng_state,
                                      unsigned site, void* stream) {
    SAIS_ENTER();
    if (!qkv || !key_pad || !ctx || B <= 0 || S <= 0 || S > SAIS_TEMPORAL_MAX_S_FWD) return SAIS_ERR_ARG;
    if (p_drop < 0.f || p_drop >= 1.f || (p_drop > 0.f && !rng_state)) return SAIS_ERR_ARG;
    const int Spad = (S + 15) / 16 * 16;
    const int lds = 3 * Spad * TLD * 4 + Spad;
    if (set_lds(tattn_fwd_kernel, lds)) return SAIS_ERR_LAUNCH;
    hipStream_t s = (hipStream_t)stream;
    if (attn_avg) {
        const long n = (long)B * S * S;
        hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024)), dim3(256), 0, s, attn_avg, n);
    }
    hipLaunchKernelGGL(tattn_fwd_kernel, dim3(TH, B), dim3(256), lds, s, qkv, key_pad, S, ctx, attn_avg, p_drop, rng_state,
                       site);
    return sais_check_launch();
}

extern "C" int sais_temporal_attn_bwd(const float* qkv, const unsigned char* key_pad, int B, int S,
                                      const float* dctx, int nslab, long slab_stride, float* dqkv, float p_drop,
                                      const unsigned long long* rng_state, unsigned site, void* stream) {
    SAIS_ENTER();
    if (!qkv || !key_pad || !dctx || !dqkv || B <= 0 || S <= 0 || S > SAIS_TEMPORAL_MAX_S_BWD || nslab <= 0 || (slab_stride & 3))
        return SAIS_ERR_ARG;
    if (p_drop < 0.f || p_drop >= 1.f || (p_drop > 0.f && !rng_state)) return SAIS_ERR_ARG;
    const int Spad = (S + 15) / 16 * 16;
    const int lds = (4 * Spad * TLD + 3 * Spad) * 4 + Spad;
    if (set_lds(tattn_bwd_kernel, lds)) return SAIS_ERR_LAUNCH;
    hipLaunchKernelGGL(tattn_bwd_kernel, dim3(TH, B), dim3(256), lds, (hipStream_t)stream, qkv, key_pad, S, dctx, nslab,
                       slab_stride, dqkv, p_drop, rng_state, site);
    return sais_check_launch();
}
