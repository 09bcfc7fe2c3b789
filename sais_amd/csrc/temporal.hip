// Temporal encoder glue + SupCon / prototype head of SAIS for gfx950.  These tensors are tiny
// (S = T+1 <= 97 tokens per clip, 4 heads x 96), latency-bound, and feed the <=1e-3 logit parity
// budget directly, so everything here is exact fp32 VALU math staged through LDS; the linear layers
// around them go through the MFMA GEMMs in gemm.hip.
//
//   prepareInputForTransformer  prepare_model.py:179-195   (sais_temporal_prepare_*)
//   (the attention core: tattn.hip; the linear layers and LayerNorms: tgemm.hip)
//   ReLU -> CLS row -> (+flow) -> ReLU -> Linear(384->256)   prepare_model.py:215,220,381-416 (sais_head_*)
//   calcNCELoss / getProbs      prepare_miscellaneous.py:14-46,111-126   (sais_nce_*)
#include "common.hpp"
#include "philox.hpp"
#include "../../include/sais_hip.h"

namespace {
constexpr int D = 384, TH = 4, THD = 96, EMB = 256, QS = 97;   // QS: padded LDS row (floats)

// ---------------------------------------------------------------- prepare: z[b, 0] = cls ; z[b, 1+t] = x[b, t] + pos[t]
__global__ void prepare_fwd_kernel(const float* x, long x_clip_stride, long x_frame_stride, const float* pos,
                                   const float* cls, int B, int T, float* z32, bf16* z16) {
    const int S = T + 1;
    long total = (long)B * S * (D / 4);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c4 = i % (D / 4);
        long r = i / (D / 4);
        int b = r / S, s = r % S;
        f32x4 v;
        if (s == 0) v = *(const f32x4*)(cls + 4 * c4);
        else {
            v = *(const f32x4*)(x + (size_t)b * x_clip_stride + (size_t)(s - 1) * x_frame_stride + 4 * c4);
            v += *(const f32x4*)(pos + (size_t)(s - 1) * D + 4 * c4);
        }
        *(f32x4*)(z32 + (size_t)r * D + 4 * c4) = v;
        if (z16) {
            bf16x4 o;
            o[0] = (bf16)v[0]; o[1] = (bf16)v[1]; o[2] = (bf16)v[2]; o[3] = (bf16)v[3];
            *(bf16x4*)(z16 + (size_t)r * D + 4 * c4) = o;
        }
    }
}

// dz = dz32 + sum_z slabs[z] (either may be absent) ; dx[b,t] (+)= dz[b,1+t] ; dpos[t] += sum_b ; dcls += sum_b dz[b,0].
// One thread per (sequence, token, 4 columns): every load of a thread is independent (a per-column loop over the sequences
// was B x nslab dependent round trips: 16 us), the sums over the sequences are float atomics (B per element).
__global__ void prepare_bwd_kernel(const float* dz32, const float* slabs, int nslab, long slab_stride, int B, int T,
                                   float* dx, long dx_clip_stride, long dx_frame_stride, int accumulate, float* dpos,
                                   float* dcls) {
    const int S = T + 1;
    const long i = blockIdx.x * 256L + threadIdx.x;            // over B * S * (D / 4)
    if (i >= (long)B * S * (D / 4)) return;
    const int c = 4 * (int)(i % (D / 4));
    const long r = i / (D / 4);
    const int b = (int)(r / S), s = (int)(r % S);
    const size_t idx = (size_t)r * D + c;
    f32x4 v = dz32 ? *(const f32x4*)(dz32 + idx) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < nslab; ++z) v += *(const f32x4*)(slabs + (size_t)z * slab_stride + idx);
    if (s > 0 && dx) {
        float* o = dx + (size_t)b * dx_clip_stride + (size_t)(s - 1) * dx_frame_stride + c;
        f32x4 out = v;
        if (accumulate) out += *(const f32x4*)o;
        *(f32x4*)o = out;
    }
    float* acc = s == 0 ? dcls + c : dpos + (size_t)(s - 1) * D + c;
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(acc + e, v[e]);
}

// (masked multi-head attention: tattn.hip)

// ---------------------------------------------------------------- head
// rep = relu(z_rgb[b,0]) (+ relu(z_flow[b,0]));  emb = W relu(rep) + bias
// useB (optional, [B]): clips flagged 1 go through (WB, biasB) instead of (W, bias) — the per-sample linear / linearB
// selection of multi-domain models (prepare_model.py:405-414)
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* zr, const float* zf, long clip_stride,
                                                       long clip_stride_f, int ns, const float* W, const float* bias,
                                                       const unsigned char* useB, const float* WB, const float* biasB,
                                                       float* rep, float* emb) {
    __shared__ float sr[D];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float inv = 1.0f / ns;
    for (int c = tid; c < D; c += 256) {
        // mean over the ns snippets of a clip of relu(CLS row) (prepare_model.py:215,220,381-382), per stream
        float vr = 0.f, vf = 0.f;
        for (int s = 0; s < ns; ++s) {
            if (zr) vr += fmaxf(zr[(size_t)(b * ns + s) * clip_stride + c], 0.f);
            if (zf) vf += fmaxf(zf[(size_t)(b * ns + s) * clip_stride_f + c], 0.f);
        }
        const float v = vr * inv + vf * inv;
        if (blockIdx.y == 0) rep[(size_t)b * D + c] = v;
        sr[c] = fmaxf(v, 0.f);
    }
    __syncthreads();
    const bool second = useB && useB[b];
    const float* Ws = second ? WB : W;
    const float* bs = second ? biasB : bias;
    // grid (B, 4): a workgroup owns 64 of the 256 outputs of its clip, a wave 16 of them, the 384 inputs across its lanes
    // (every weight row is read as one contiguous 1.5-KB run; a thread per output walked 64 different rows per load
    // instruction), EIGHT outputs per pass so that 48 independent loads are in flight (one output per pass is a chain of
    // dependent round trips: 56 us measured)
    const int lane = tid & 63, wid = tid >> 6;
    float x[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = sr[lane + 64 * i];
    const int obeg = 64 * blockIdx.y + 16 * wid;
    for (int o0 = obeg; o0 < obeg + 16; o0 += 8) {
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float* w = Ws + (size_t)(o0 + u) * D + lane;
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 6; ++i) t += w[64 * i] * x[i];
            a[u] = t;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float t = wave_sum(a[u]);
            if (lane == 0) emb[(size_t)b * EMB + o0 + u] = t + bs[o0 + u];
        }
    }
}

// demb [B,256] -> dz_rgb[b,s,0,:] / dz_flow[b,s,0,:] (one workgroup per clip), then dW += demb^T relu(rep) and
// db += sum_b demb with one OWNER thread per output element (a loop over the clips instead of B x 256 x 384 atomics:
// 54 us -> a few us at B = 8)
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* demb, const float* W, const float* rep,
                                                       const float* zr, const float* zf, long clip_stride,
                                                       long clip_stride_f, int B, int ns, const unsigned char* useB,
                                                       const float* WB, float* dzr, float* dzf) {
    __shared__ float sd[EMB];
    const int b = blockIdx.x, tid = threadIdx.x;
    sd[tid] = demb[(size_t)b * EMB + tid];
    if (useB && useB[b]) W = WB;
    __syncthreads();
    for (int c = tid; c < D; c += 256) {
        const float r = rep[(size_t)b * D + c];
        float a = 0.f;
        for (int o = 0; o < EMB; ++o) a += W[(size_t)o * D + c] * sd[o];
        const float drep = (r > 0.f ? a : 0.f) / ns;
        for (int s = 0; s < ns; ++s) {
            const size_t ir = (size_t)(b * ns + s) * clip_stride + c, jf = (size_t)(b * ns + s) * clip_stride_f + c;
            if (zr) dzr[ir] = zr[ir] > 0.f ? drep : 0.f;
            if (zf) dzf[jf] = zf[jf] > 0.f ? drep : 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void head_bwd_dw_kernel(const float* demb, const float* rep, int B,
                                                          const unsigned char* useB, float* dW, float* dbias, float* dWB,
                                                          float* dbiasB) {
    const int c = blockIdx.x, o = threadIdx.x;               // grid D (+1 block for the bias), 256 threads = EMB
    float a = 0.f, a2 = 0.f;                                 // a: clips of the first head, a2: clips flagged for linearB
    for (int b = 0; b < B; ++b) {
        const float t = demb[(size_t)b * EMB + o] * (c == D ? 1.f : fmaxf(rep[(size_t)b * D + c], 0.f));
        if (useB && useB[b]) a2 += t; else a += t;
    }
    if (c == D) {
        dbias[o] += a;
        if (useB) dbiasB[o] += a2;
    } else {
        dW[(size_t)o * D + c] += a;
        if (useB) dWB[(size_t)o * D + c] += a2;
    }
}

// ---------------------------------------------------------------- MIL pathway, inference (prepare_model.py:356-361)
// getClipReps input (:452-460): tokens[b, s] = relu(z[b * ns + s, CLS row]) + clip_pos[s]   (one sequence of ns snippets per clip)
__global__ __launch_bounds__(128) void mil_prepare_kernel(const float* z, long seq_stride, const float* clip_pos, int ns,
                                                          float* tokens) {
    const int r = blockIdx.x, s = r % ns;                    // r = b * ns + s
    for (int c = threadIdx.x; c < D; c += 128)
        tokens[(size_t)r * D + c] = fmaxf(z[(size_t)r * seq_stride + c], 0.f) + clip_pos[(size_t)s * D + c];
}

// MIL_Head (:470-488) on reps = relu(clip encoder output) [B, ns, 384], one workgroup per clip, thread e = one of the 256
// gate units: gated[s][e] = tanh(A reps_s + a)[e] * sigmoid(Bm reps_s + bm)[e]  (calcAttention :131-138); per class c:
// att_c = softmax_s(w_c . gated[s] + b_c), video_c = sum_s att_c[s] reps_s, logit_c = f_c . video_c + g_c (:140-148).
// Writes reps (the relu), logits [B, C], attention [C, B, ns].  ns <= 128, C <= 3.
__global__ __launch_bounds__(256) void mil_head_kernel(const float* enc, int ns, int C, const float* WA, const float* bA,
                                                       const float* WB, const float* bB, const float* wAtt,
                                                       const float* bAtt, const float* wFin, const float* bFin,
                                                       float* reps, float* logits, float* attention, int B) {
    __shared__ float row[D];
    __shared__ float red[3][4];
    __shared__ float score[3][128];
    const int b = blockIdx.x, e = threadIdx.x, lane = e & 63, w = e >> 6;
    for (int s = 0; s < ns; ++s) {
        __syncthreads();
        for (int c = e; c < D; c += 256) {
            const float v = fmaxf(enc[((size_t)b * ns + s) * D + c], 0.f);
            row[c] = v;
            reps[((size_t)b * ns + s) * D + c] = v;
        }
        __syncthreads();
        float a = bA[e], g = bB[e];
        for (int c = 0; c < D; c += 4) {
            const f32x4 ta = *(const f32x4*)(WA + (size_t)e * D + c), tb = *(const f32x4*)(WB + (size_t)e * D + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) { a += ta[j] * row[c + j]; g += tb[j] * row[c + j]; }
        }
        const float gated = tanhf(a) * (1.0f / (1.0f + __expf(-g)));
        for (int c = 0; c < C; ++c) {
            const float t = wave_sum(gated * wAtt[(size_t)c * EMB + e]);
            if (lane == 0) red[c][w] = t;
        }
        __syncthreads();
        if (e < C) score[e][s] = red[e][0] + red[e][1] + red[e][2] + red[e][3] + bAtt[e];
    }
    __syncthreads();
    if (e < C) {                                             // softmax over the snippets
        float m = -INFINITY, sum = 0.f;
        for (int s = 0; s < ns; ++s) m = fmaxf(m, score[e][s]);
        for (int s = 0; s < ns; ++s) { const float t = __expf(score[e][s] - m); score[e][s] = t; sum += t; }
        for (int s = 0; s < ns; ++s) {
            score[e][s] /= sum;
            attention[((size_t)e * B + b) * ns + s] = score[e][s];
        }
    }
    __syncthreads();
    for (int c = 0; c < C; ++c) {
        float part = 0.f;
        for (int d = e; d < D; d += 256) {
            float v = 0.f;
            for (int s = 0; s < ns; ++s) v += score[c][s] * fmaxf(enc[((size_t)b * ns + s) * D + d], 0.f);
            part += v * wFin[(size_t)c * D + d];
        }
        part = wave_sum(part);
        __syncthreads();
        if (lane == 0) red[0][w] = part;
        __syncthreads();
        if (e == 0) logits[(size_t)b * C + c] = red[0][0] + red[0][1] + red[0][2] + red[0][3] + bFin[c];
    }
}

// ---------------------------------------------------------------- importance head (-il): Linear(384 -> 1) on relu(z)
__global__ __launch_bounds__(256) void importance_fwd_kernel(const float* z, const float* w, const float* b, int M, float* out) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float a = 0.f;
    for (int c = lane; c < D; c += 64) a += w[c] * fmaxf(z[(size_t)row * D + c], 0.f);
    a = wave_sum(a);
    if (lane == 0) out[row] = a + b[0];
}

// dz[m,c] += dl[m] w[c] (z>0) ; dw[c] += sum_m dl[m] relu(z[m,c]) ; db += sum_m dl[m]
__global__ __launch_bounds__(384) void importance_bwd_kernel(const float* dl, const float* z, const float* w, int M,
                                                            float* dz, float* dw, float* db) {
    const int c = threadIdx.x;
    float aw = 0.f, ab = 0.f;
    const float wc = w[c];
    for (int m = blockIdx.x; m < M; m += gridDim.x) {
        const float g = dl[m], zz = z[(size_t)m * D + c];
        if (zz > 0.f) { dz[(size_t)m * D + c] += g * wc; aw += g * zz; }
        ab += g;
    }
    atomicAdd(dw + c, aw);
    if (c == 0) atomicAdd(db, ab);
}

// calcImportanceLoss (prepare_miscellaneous.py:48-60), quirks kept; one workgroup.  logits [B,S] (slot 0 = CLS),
// target [B,T], ipad [B,S] (1 = masked), labels [B].  loss = mean_bce * sum_{b low, t<T} !ipad[b,t] / (nlow T);
// dlogits[b,1+t] = frac * (sigmoid(x) - target) / (B T) * scale, dlogits[b,0] = 0.
__global__ __launch_bounds__(256) void importance_loss_kernel(const float* logits, const float* target,
                                                              const unsigned char* ipad, const int* labels, int B, int T,
                                                              float* loss, float* dlogits, float scale) {
    __shared__ float red[256];
    __shared__ float s_frac, s_bce;
    const int S = T + 1, tid = threadIdx.x;
    float a = 0.f, cnt = 0.f, nlow = 0.f;
    for (int i = tid; i < B * T; i += 256) {
        const int b = i / T, t = i - b * T;
        const float x = logits[b * S + 1 + t], y = target[i];
        a += fmaxf(x, 0.f) - x * y + log1pf(__expf(-fabsf(x)));
        if (labels[b] == 0 && !ipad[b * S + t]) cnt += 1.f;
    }
    for (int b = tid; b < B; b += 256) nlow += labels[b] == 0 ? 1.f : 0.f;
    for (int pass = 0; pass < 3; ++pass) {
        red[tid] = pass == 0 ? a : (pass == 1 ? cnt : nlow);
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
        if (tid == 0) { if (pass == 0) a = red[0]; else if (pass == 1) cnt = red[0]; else nlow = red[0]; }
        __syncthreads();
    }
    if (tid == 0) {
        s_bce = a / (float)(B * T);
        s_frac = cnt / (nlow * (float)T);          // 0/0 = NaN when no low-skill sample: torch.mean of an empty tensor
        if (loss) *loss = s_bce * s_frac;
    }
    __syncthreads();
    if (!dlogits) return;
    const float k = s_frac * scale / (float)(B * T);
    for (int i = tid; i < B * S; i += 256) {
        const int b = i / S, s = i - b * S;
        float g = 0.f;
        if (s > 0) {
            const float x = logits[i];
            g = k * (1.0f / (1.0f + __expf(-x)) - target[b * T + s - 1]);
        }
        dlogits[i] = g;
    }
}

// ---------------------------------------------------------------- SupCon / prototype loss; one workgroup
// Latency is the whole cost of this kernel (B x C = 8 x 2 outputs): round 3's 4-wave form walked the B + C rows and the
// B C pairs in rounds of four with a dependent global load in each (17.9 us inside the step).  Now 16 waves, and the
// (B + C) x 256 operands are staged ONCE into LDS by all threads (one round trip); every later phase is one round out of
// LDS for B + C <= 16.  The arithmetic (summation order of every dot product, the loss sum) is unchanged.
constexpr int NCE_THREADS = 1024;
__global__ __launch_bounds__(NCE_THREADS) void nce_kernel(const float* emb, const float* protos, const int* label_col, int B,
                                                         int C, float* sim, float* probs, float* loss, float* demb,
                                                         float* dprotos, float loss_scale, int staged) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* en = (float*)smem;                 // [B] |emb|
    float* pn = en + B;                       // [C] |p|
    float* ss = pn + C;                       // [B*C] sim
    float* dsim = ss + B * C;                 // [B*C]
    float* lrow = dsim + B * C;               // [B]
    float* stage = lrow + B + ((4 - ((2 * B + C + 2 * B * C) & 3)) & 3);        // 16-B aligned: [B + C][EMB] when staged
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int NWV = NCE_THREADS / 64;
    const float* E = emb;
    const float* P = protos;
    if (staged) {
        for (int i = tid; i < (B + C) * (EMB / 4); i += NCE_THREADS) {
            const int r = i / (EMB / 4), c4 = i - r * (EMB / 4);
            const float* v = r < B ? emb + (size_t)r * EMB : protos + (size_t)(r - B) * EMB;
            *(f32x4*)(stage + (size_t)r * EMB + 4 * c4) = *(const f32x4*)(v + 4 * c4);
        }
        E = stage;
        P = stage + (size_t)B * EMB;
        __syncthreads();
    }
    for (int r = w; r < B + C; r += NWV) {
        const float* v = r < B ? E + (size_t)r * EMB : P + (size_t)(r - B) * EMB;
        float a = 0.f;
        for (int c = lane; c < EMB; c += 64) a += v[c] * v[c];
        a = sqrtf(wave_sum(a));
        if (lane == 0) (r < B ? en[r] : pn[r - B]) = a;
    }
    __syncthreads();
    for (int idx = w; idx < B * C; idx += NWV) {
        int i = idx / C, j = idx % C;
        float a = 0.f;
        for (int c = lane; c < EMB; c += 64) a += E[(size_t)i * EMB + c] * P[(size_t)j * EMB + c];
        a = wave_sum(a) / (en[i] * pn[j]);
        if (lane == 0) { ss[idx] = a; if (sim) sim[idx] = a; }
    }
    __syncthreads();
    for (int i = tid; i < B; i += NCE_THREADS) {
        float den = 0.f;
        for (int j = 0; j < C; ++j) den += __expf(ss[i * C + j]);
        int y = label_col ? label_col[i] : 0;
        for (int j = 0; j < C; ++j) {
            float p = __expf(ss[i * C + j]) / den;
            if (probs) probs[i * C + j] = p;
            dsim[i * C + j] = (p - (j == y ? 1.f : 0.f)) * loss_scale / B;
        }
        lrow[i] = -__logf(__expf(ss[i * C + y]) / den);
    }
    __syncthreads();
    if (tid == 0 && loss) {
        float a = 0.f;
        for (int i = 0; i < B; ++i) a += lrow[i];
        *loss = a / B;
    }
    if (!demb) return;
    // d s_hat_i = sum_j dsim[i][j] p_hat_j ;  d emb_i = (d s_hat_i - s_hat_i (s_hat_i . d s_hat_i)) / |emb_i|
    for (int r = w; r < B + C; r += NWV) {
        const bool is_e = r < B;
        const int i = is_e ? r : r - B;
        const float* self = is_e ? E + (size_t)i * EMB : P + (size_t)i * EMB;
        const float nself = is_e ? en[i] : pn[i];
        float g[EMB / 64], sh[EMB / 64];
        float dot = 0.f;
#pragma unroll
        for (int u = 0; u < EMB / 64; ++u) {
            int c = lane + 64 * u;
            float a = 0.f;
            if (is_e) for (int j = 0; j < C; ++j) a += dsim[i * C + j] * P[(size_t)j * EMB + c] / pn[j];
            else      for (int k = 0; k < B; ++k) a += dsim[k * C + i] * E[(size_t)k * EMB + c] / en[k];
            g[u] = a;
            sh[u] = self[c] / nself;
            dot += a * sh[u];
        }
        dot = wave_sum(dot);
        float* out = is_e ? demb + (size_t)i * EMB : dprotos + (size_t)i * EMB;
#pragma unroll
        for (int u = 0; u < EMB / 64; ++u) {
            float v = (g[u] - sh[u] * dot) / nself;
            if (is_e) out[lane + 64 * u] = v; else out[lane + 64 * u] += v;
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
}  // namespace

extern "C" int sais_temporal_prepare_fwd(const float* x, long x_clip_stride, long x_frame_stride, const float* pos,
                                         const float* cls, int B, int T, float* z_f32, void* z_bf16, void* stream) {
    SAIS_ENTER();
    if (!x || !pos || !cls || !z_f32 || B <= 0 || T <= 0 || (x_clip_stride & 3) || (x_frame_stride & 3))
        return SAIS_ERR_ARG;
    long total = (long)B * (T + 1) * (D / 4);
    hipLaunchKernelGGL(prepare_fwd_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       x_clip_stride, x_frame_stride, pos, cls, B, T, z_f32, (bf16*)z_bf16);
    return sais_check_launch();
}

extern "C" int sais_temporal_prepare_bwd(const float* dz_f32, const float* slabs, int nslab, long slab_stride, int B, int T,
                                         float* dx, long dx_clip_stride, long dx_frame_stride, int accumulate, float* dpos,
                                         float* dcls, void* stream) {
    SAIS_ENTER();
    if ((!dz_f32 && !slabs) || (slabs && nslab <= 0) || !dpos || !dcls || B <= 0 || T <= 0) return SAIS_ERR_ARG;
    if ((dx_clip_stride & 3) || (dx_frame_stride & 3) || (slab_stride & 3)) return SAIS_ERR_ARG;
    const long nthr = (long)B * (T + 1) * (D / 4);
    hipLaunchKernelGGL(prepare_bwd_kernel, dim3((int)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dz_f32,
                       slabs, slabs ? nslab : 0, slab_stride, B, T, dx, dx_clip_stride, dx_frame_stride, accumulate, dpos,
                       dcls);
    return sais_check_launch();
}

extern "C" int sais_head_fwd(const float* z_rgb, const float* z_flow, long clip_stride, long clip_stride_flow, int B,
                             int nsnippets, const float* W, const float* bias, const unsigned char* use_b, const float* WB,
                             const float* biasB, float* rep, float* emb, void* stream) {
    SAIS_ENTER();
    if ((!z_rgb && !z_flow) || !W || !bias || !rep || !emb || B <= 0 || nsnippets <= 0) return SAIS_ERR_ARG;
    if (use_b && (!WB || !biasB)) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(head_fwd_kernel, dim3(B, 4), dim3(256), 0, (hipStream_t)stream, z_rgb, z_flow, clip_stride,
                       clip_stride_flow, nsnippets, W, bias, use_b, WB, biasB, rep, emb);
    return sais_check_launch();
}

extern "C" int sais_mil_forward(const float* z_rgb, long seq_stride, const float* clip_pos, int B, int nsnippets, float* tokens,
                                void* stream) {
    SAIS_ENTER();
    if (!z_rgb || !clip_pos || !tokens || B <= 0 || nsnippets <= 0) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(mil_prepare_kernel, dim3(B * nsnippets), dim3(128), 0, (hipStream_t)stream, z_rgb, seq_stride,
                       clip_pos, nsnippets, tokens);
    return sais_check_launch();
}

extern "C" int sais_mil_head(const float* enc, int B, int nsnippets, int nclasses, const float* WA, const float* bA,
                             const float* WB, const float* bB, const float* w_att, const float* b_att, const float* w_fin,
                             const float* b_fin, float* reps, float* logits, float* attention, void* stream) {
    SAIS_ENTER();
    if (!enc || !WA || !bA || !WB || !bB || !w_att || !b_att || !w_fin || !b_fin || !reps || !logits || !attention)
        return SAIS_ERR_ARG;
    if (B <= 0 || nsnippets <= 0 || nsnippets > 128 || nclasses <= 0 || nclasses > 3) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(mil_head_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, enc, nsnippets, nclasses, WA, bA, WB, bB,
                       w_att, b_att, w_fin, b_fin, reps, logits, attention, B);
    return sais_check_launch();
}

extern "C" int sais_head_bwd(const float* demb, const float* W, const float* rep, const float* z_rgb,
                             const float* z_flow, long clip_stride, long clip_stride_flow, int B, int nsnippets,
                             const unsigned char* use_b, const float* WB, float* dW, float* dbias, float* dWB, float* dbiasB,
                             float* dz_rgb, float* dz_flow, void* stream) {
    SAIS_ENTER();
    if (!demb || !W || !rep || !dW || !dbias || B <= 0 || nsnippets <= 0) return SAIS_ERR_ARG;
    if ((z_rgb && !dz_rgb) || (z_flow && !dz_flow)) return SAIS_ERR_ARG;
    if (use_b && (!WB || !dWB || !dbiasB)) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(head_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, demb, W, rep, z_rgb, z_flow,
                       clip_stride, clip_stride_flow, B, nsnippets, use_b, WB, dz_rgb, dz_flow);
    hipLaunchKernelGGL(head_bwd_dw_kernel, dim3(D + 1), dim3(EMB), 0, (hipStream_t)stream, demb, rep, B, use_b, dW, dbias,
                       dWB, dbiasB);
    return sais_check_launch();
}

extern "C" int sais_importance_fwd(const float* z, const float* w, const float* b, int M, float* out, void* stream) {
    SAIS_ENTER();
    if (!z || !w || !b || !out || M <= 0) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(importance_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, z, w, b, M, out);
    return sais_check_launch();
}

extern "C" int sais_importance_bwd(const float* dlogit, const float* z, const float* w, int M, float* dz, float* dw,
                                   float* db, void* stream) {
    SAIS_ENTER();
    if (!dlogit || !z || !w || !dz || !dw || !db || M <= 0) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(importance_bwd_kernel, dim3(M < 64 ? M : 64), dim3(384), 0, (hipStream_t)stream, dlogit, z, w, M, dz,
                       dw, db);
    return sais_check_launch();
}

extern "C" int sais_importance_loss(const float* logits, const float* target, const unsigned char* ipad,
                                    const int* labels, int B, int T, float* loss, float* dlogits, float scale,
                                    void* stream) {
    SAIS_ENTER();
    if (!logits || !target || !ipad || !labels || B <= 0 || T <= 0) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(importance_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, target, ipad, labels, B,
                       T, loss, dlogits, scale);
    return sais_check_launch();
}

extern "C" int sais_nce(const float* emb, const float* protos, const int* label_col, int B, int C, float* sim,
                        float* probs, float* loss, float* demb, float* dprotos, float loss_scale, void* stream) {
    SAIS_ENTER();
    if (!emb || !protos || B <= 0 || C <= 0 || (demb && (!dprotos || !label_col))) return SAIS_ERR_ARG;
    int lds = (B + C + 2 * B * C + B + 3) * 4;
    if (lds > 64 * 1024) return SAIS_ERR_ARG;
    const long stage = (long)(B + C) * EMB * 4;
    const int staged = lds + stage <= 60 * 1024 ? 1 : 0;              // within the default dynamic-LDS limit: no attribute call
    if (staged) lds += (int)stage;
    hipLaunchKernelGGL(nce_kernel, dim3(1), dim3(NCE_THREADS), lds, (hipStream_t)stream, emb, protos, label_col, B, C, sim,
                       probs, loss, demb, dprotos, loss_scale, staged);
    return sais_check_launch();
}
