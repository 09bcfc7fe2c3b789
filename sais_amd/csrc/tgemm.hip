// The linear layers of the temporal TransformerEncoder (prepare_model.py:74-81, called at :213) on gfx950.
//
// The problem: M = clips x (T + 1) is a few hundred rows (264 at the benchmark), N, K in {384, 1152, 2048}: every GEMM is
// a few GFLOP and a few MB of fp32 weights, i.e. a LATENCY problem — what matters is how many CUs pull on the weights at
// once and how many dependent round trips a workgroup makes.  Round 2 ran these on the 128 x 128 tile of gemm.hip
// (48 workgroups for the FFN, one K-step of latency exposed per step: 33 us for 1.2 GFLOP) plus a split-K reduce kernel
// and a LayerNorm kernel behind every N = 384 GEMM (~110 launches per step).  Here:
//
//   tgemm_kernel      64 x 64 tile per 256-thread workgroup (4 waves of 32 x 32), fp32 operands split hi/lo into bf16
//                     while staging ("bf16x3": a_hi w_hi + a_hi w_lo + a_lo w_hi, ~2^-17 relative), TWO K-steps of
//                     global loads in flight in registers, two LDS stages, one barrier per K-step; optional split-K
//                     over gridDim.z into raw partial slabs.  160-240 workgroups for every shape of the layer.
//   tln_fwd_kernel    the consumer of the N = 384 slabs in the forward: y = resid + dropout(sum_z slab_z + bias) and
//                     z = LayerNorm(y) in one pass (the split-K reduce, the bias / dropout / residual epilogue and norm1 /
//                     norm2 of the post-norm layer).
//   tln_bwd_kernel    the same for the backward: dy = sum_z slab_z + add, then autograd of LayerNorm, the residual-branch
//                     dropout as a second output, dgamma / dbeta.
// Dropout masks: Philox element index m * N + n of the site, exactly as sais_dropout_f32 / sais_dropout_mask define them.
#include "common.hpp"
#include "philox.hpp"
#include "../../include/sais_hip.h"

namespace {
constexpr int D = 384;
constexpr int TT = 64 * 128;                  // one operand half (hi or lo) of a 64 x 64 tile: 64 rows x 128 B

struct TgParams {
    const float* A; const float* W; long lda, ldw;
    int M, N, K, nsplit;
    const float* bias; const float* aux; long ldaux;
    float* out; long ldo;
    float p_drop; const unsigned long long* rng; unsigned site;
};

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

// keep / (1 - p) factors of the four consecutive mask elements idx .. idx + 3 (idx % 4 == 0): ONE Philox block
DEVINL f32x4 keep4(const unsigned long long* rng, unsigned site, unsigned long long idx, unsigned thr, float inv) {
    unsigned r[4];
    philox_u32x4(rng, site, idx, r);
    return f32x4{r[0] >= thr ? inv : 0.f, r[1] >= thr ? inv : 0.f, r[2] >= thr ? inv : 0.f, r[3] >= thr ? inv : 0.f};
}

template <int EPI>
__global__ __launch_bounds__(256) void tgemm_kernel(TgParams p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 4 * TT];       // 2 stages x {A_hi, A_lo, W_hi, W_lo}
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, g = lane >> 4, li = lane & 15;
    const int n0 = blockIdx.x * 64, m0 = blockIdx.y * 64;
    // staging: thread -> tile row tid >> 2, 16 consecutive k = two 16-B bf16 chunks (2 c, 2 c + 1)
    const int sr = tid >> 2, sc = tid & 3;
    int ma = m0 + sr;
    ma = ma < p.M ? ma : p.M - 1;                                         // rows >= M: loaded (clamped), never stored
    const float* pa = p.A + (size_t)ma * p.lda + sc * 16;
    const float* pw = p.W + (size_t)(n0 + sr) * p.ldw + sc * 16;
    const int nk = p.K / 64 / p.nsplit, kbeg = blockIdx.z * nk;

    f32x4 ra[2][4], rw[2][4];
#define TG_LOAD(SET, KT)                                                                        \
    {                                                                                           \
        const float* qa = pa + (size_t)(kbeg + (KT)) * 64;                                      \
        const float* qw = pw + (size_t)(kbeg + (KT)) * 64;                                      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                         \
            ra[SET][i] = *(const f32x4*)(qa + 4 * i);                                           \
            rw[SET][i] = *(const f32x4*)(qw + 4 * i);                                           \
        }                                                                                       \
    }
#define TG_STORE(SET, STAGE)                                                                    \
    {                                                                                           \
        char* s = smem + (STAGE) * 4 * TT;                                                      \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                         \
            u32x4 hi, lo;                                                                       \
            split8(ra[SET][2 * h], ra[SET][2 * h + 1], hi, lo);                                 \
            *(u32x4*)(s + swz(sr, 2 * sc + h)) = hi;                                            \
            *(u32x4*)(s + TT + swz(sr, 2 * sc + h)) = lo;                                       \
            split8(rw[SET][2 * h], rw[SET][2 * h + 1], hi, lo);                                 \
            *(u32x4*)(s + 2 * TT + swz(sr, 2 * sc + h)) = hi;                                   \
            *(u32x4*)(s + 3 * TT + swz(sr, 2 * sc + h)) = lo;                                   \
        }                                                                                       \
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
#define TG_COMPUTE(STAGE)                                                                       \
    {                                                                                           \
        const char* s = smem + (STAGE) * 4 * TT;                                                \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                      \
            bf16x8 ah[2], al[2], wh[2], wl[2];                                                  \
            _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                     \
                const int oa = swz(wr * 32 + t * 16 + li, ks * 4 + g);                          \
                const int ow = swz(wc * 32 + t * 16 + li, ks * 4 + g);                          \
                ah[t] = *(const bf16x8*)(s + oa);                                               \
                al[t] = *(const bf16x8*)(s + TT + oa);                                          \
                wh[t] = *(const bf16x8*)(s + 2 * TT + ow);                                      \
                wl[t] = *(const bf16x8*)(s + 3 * TT + ow);                                      \
            }                                                                                   \
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                    \
                _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                              \
                    f32x4 c = acc[mt][nt];                                                      \
                    c = mfma16(wl[nt], ah[mt], c);                                              \
                    c = mfma16(wh[nt], al[mt], c);                                              \
                    c = mfma16(wh[nt], ah[mt], c);                                              \
                    acc[mt][nt] = c;                                                            \
                }                                                                               \
        }                                                                                       \
    }
    // step kt lives in register set / LDS stage kt & 1; loads run two steps ahead of the MFMAs.  Stage s is rewritten at
    // step kt + 2 by waves that have passed the barrier of step kt + 1, i.e. after every wave finished computing step kt.
    TG_LOAD(0, 0)
    if (nk > 1) TG_LOAD(1, 1)
    for (int kt = 0; kt < nk; kt += 2) {
        TG_STORE(0, 0)
        __syncthreads();
        if (kt + 2 < nk) TG_LOAD(0, kt + 2)
        TG_COMPUTE(0)
        if (kt + 1 < nk) {
            TG_STORE(1, 1)
            __syncthreads();
            if (kt + 3 < nk) TG_LOAD(1, kt + 3)
            TG_COMPUTE(1)
        }
    }
#undef TG_LOAD
#undef TG_STORE
#undef TG_COMPUTE

    // D[i = n][j = m]: the lane holds, per (mt, nt), row m0 + 32 wr + 16 mt + li and the 4 columns n0 + 32 wc + 16 nt + 4 g ..
    const bool dropping = p.p_drop > 0.f;
    const unsigned thr = drop_threshold(p.p_drop);
    const float inv = dropping ? 1.0f / (1.0f - p.p_drop) : 1.0f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = m0 + wr * 32 + mt * 16 + li;
        if (m >= p.M) continue;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = n0 + wc * 32 + nt * 16 + 4 * g;
            f32x4 v = acc[mt][nt];
            if constexpr (EPI == SAIS_TG_RAW) {
                *(f32x4*)(p.out + ((size_t)blockIdx.z * p.M + m) * p.ldo + n) = v;
            } else {
                if (p.bias) v += *(const f32x4*)(p.bias + n);
                f32x4 keep = {1.f, 1.f, 1.f, 1.f};
                if (dropping) keep = keep4(p.rng, p.site, (unsigned long long)m * p.N + n, thr, inv);
                if constexpr (EPI == SAIS_TG_BIAS_RELU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f) * keep[j];
                } else if constexpr (EPI == SAIS_TG_DRELU) {
                    const f32x4 u = *(const f32x4*)(p.aux + (size_t)m * p.ldaux + n);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = u[j] > 0.f ? v[j] * keep[j] : 0.f;
                }
                *(f32x4*)(p.out + (size_t)m * p.ldo + n) = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------- row kernels
DEVINL float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
DEVINL void ld12(const float* q, int l32, float (&v)[12]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const f32x4 t = *(const f32x4*)(q + 128 * i + 4 * l32);
        v[4 * i] = t[0]; v[4 * i + 1] = t[1]; v[4 * i + 2] = t[2]; v[4 * i + 3] = t[3];
    }
}
DEVINL void st12(float* q, int l32, const float (&v)[12]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) *(f32x4*)(q + 128 * i + 4 * l32) = f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
}
// lane l32 of a half-wave owns columns 128 i + 4 l32 .. + 3 (i = 0..2) of its row: three aligned groups of four
DEVINL void drop12(float (&v)[12], const unsigned long long* rng, unsigned site, int row, int l32, unsigned thr, float inv) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const f32x4 k = keep4(rng, site, (unsigned long long)row * D + 128 * i + 4 * l32, thr, inv);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * i + j] *= k[j];
    }
}

// v = sum_z slabs[z][row] with FOUR slab rows in flight (a one-at-a-time loop is nslab dependent round trips: 8 us for 8 slabs)
DEVINL void sum_slabs(const float* slabs, int nslab, long slab_stride, size_t row_off, int l32, float (&v)[12]) {
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = 0.f;
    int s = 0;
    for (; s + 4 <= nslab; s += 4) {
        float t0[12], t1[12], t2[12], t3[12];
        ld12(slabs + (size_t)s * slab_stride + row_off, l32, t0);
        ld12(slabs + (size_t)(s + 1) * slab_stride + row_off, l32, t1);
        ld12(slabs + (size_t)(s + 2) * slab_stride + row_off, l32, t2);
        ld12(slabs + (size_t)(s + 3) * slab_stride + row_off, l32, t3);
#pragma unroll
        for (int i = 0; i < 12; ++i) v[i] += (t0[i] + t1[i]) + (t2[i] + t3[i]);
    }
    for (; s < nslab; ++s) {
        float t[12];
        ld12(slabs + (size_t)s * slab_stride + row_off, l32, t);
#pragma unroll
        for (int i = 0; i < 12; ++i) v[i] += t[i];
    }
}

// y = resid + dropout(sum_z slab_z + bias) ;  z = LayerNorm(y).  One row per half-wave, 8 rows per workgroup.
__global__ __launch_bounds__(256) void tln_fwd_kernel(const float* slabs, int nslab, long slab_stride, const float* bias,
                                                      const float* resid, int rows, float p_drop,
                                                      const unsigned long long* rng, unsigned site, float* y,
                                                      const float* gamma, const float* beta, float eps, float* z,
                                                      float* mean, float* rstd) {
    const int l32 = threadIdx.x & 31;
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (row >= rows) return;
    float v[12], t[12];
    sum_slabs(slabs, nslab, slab_stride, (size_t)row * D, l32, v);
    if (bias) {
        ld12(bias, l32, t);
#pragma unroll
        for (int i = 0; i < 12; ++i) v[i] += t[i];
    }
    if (p_drop > 0.f) drop12(v, rng, site, row, l32, drop_threshold(p_drop), 1.0f / (1.0f - p_drop));
    if (resid) {
        ld12(resid + (size_t)row * D, l32, t);
#pragma unroll
        for (int i = 0; i < 12; ++i) v[i] += t[i];
    }
    if (y) st12(y + (size_t)row * D, l32, v);
    float gm[12], bt[12];
    ld12(gamma, l32, gm);
    ld12(beta, l32, bt);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += v[i];
    const float mu = half_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) { const float d = v[i] - mu; q += d * d; }
    const float rs = rsqrtf(half_sum(q) * (1.0f / D) + eps);
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = (v[i] - mu) * rs * gm[i] + bt[i];
    st12(z + (size_t)row * D, l32, v);
    if (l32 == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
}

// dy = sum_z slab_z + add ;  dx = rstd (dy g - mean(dy g) - xhat mean(dy g xhat)) ;  dx_drop = dropout(dx) (the gradient
// that enters the residual BRANCH; the residual path keeps dx) ;  dgamma += sum dy xhat ;  dbeta += sum dy
__global__ __launch_bounds__(256) void tln_bwd_kernel(const float* slabs, int nslab, long slab_stride, const float* add,
                                                      const float* x, const float* mean, const float* rstd,
                                                      const float* gamma, int rows, float* dx, float* dx_drop, float p_drop,
                                                      const unsigned long long* rng, unsigned site, float* dgamma,
                                                      float* dbeta) {
    __shared__ float red[2][8][D];
    const int l32 = threadIdx.x & 31, hw = threadIdx.x >> 5;
    float gm[12], ag[12], ab[12];
    ld12(gamma, l32, gm);
#pragma unroll
    for (int i = 0; i < 12; ++i) { ag[i] = 0.f; ab[i] = 0.f; }
    for (int row = blockIdx.x * 8 + hw; row < rows; row += gridDim.x * 8) {
        float dy[12], xv[12], t[12];
        sum_slabs(slabs, nslab, slab_stride, (size_t)row * D, l32, dy);
        if (add) {
            ld12(add + (size_t)row * D, l32, t);
#pragma unroll
            for (int i = 0; i < 12; ++i) dy[i] += t[i];
        }
        ld12(x + (size_t)row * D, l32, xv);
        const float mu = mean[row], rs = rstd[row];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            xv[i] = (xv[i] - mu) * rs;
            ag[i] += dy[i] * xv[i];
            ab[i] += dy[i];
            dy[i] *= gm[i];
            c1 += dy[i];
            c2 += dy[i] * xv[i];
        }
        c1 = half_sum(c1) * (1.0f / D);
        c2 = half_sum(c2) * (1.0f / D);
#pragma unroll
        for (int i = 0; i < 12; ++i) dy[i] = rs * (dy[i] - c1 - xv[i] * c2);
        st12(dx + (size_t)row * D, l32, dy);
        if (dx_drop) {
            drop12(dy, rng, site, row, l32, drop_threshold(p_drop), 1.0f / (1.0f - p_drop));
            st12(dx_drop + (size_t)row * D, l32, dy);
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
            const int which = c / D, col = c - which * D;
            float s = 0.f;
#pragma unroll
            for (int h = 0; h < 8; ++h) s += red[which][h][col];
            atomicAdd((which ? dbeta : dgamma) + col, s);
        }
    }
}
}  // namespace

#define LAUNCH_TG(E)                                                                              \
    case E:                                                                                       \
        hipLaunchKernelGGL(tgemm_kernel<E>, grid, dim3(256), 0, (hipStream_t)stream, p);          \
        break;

extern "C" int sais_tgemm(const SaisTGemm* g, void* stream) {
    SAIS_ENTER();
    if (!g || !g->A || !g->W || !g->out) return SAIS_ERR_ARG;
    if (g->M <= 0 || g->N <= 0 || g->N % 64 || g->K <= 0 || g->K % 64 || g->lda % 4 || g->ldw % 4 || g->ldo % 4) return SAIS_ERR_ARG;
    const int ns = g->nsplit > 0 ? g->nsplit : 1;
    if ((g->K / 64) % ns) return SAIS_ERR_ARG;
    if (ns > 1 && g->epilogue != SAIS_TG_RAW) return SAIS_ERR_ARG;            // only raw partial sums can be split
    if (g->epilogue == SAIS_TG_DRELU && (!g->aux || g->ldaux % 4)) return SAIS_ERR_ARG;
    if (g->p_drop < 0.f || g->p_drop >= 1.f) return SAIS_ERR_ARG;
    if (g->p_drop > 0.f && (!g->rng_state || g->epilogue == SAIS_TG_RAW || g->epilogue == SAIS_TG_BIAS)) return SAIS_ERR_ARG;
    TgParams p{g->A, g->W, g->lda, g->ldw, g->M, g->N, g->K, ns, g->bias, g->aux, g->ldaux, g->out, g->ldo,
               g->p_drop, g->rng_state, g->site};
    dim3 grid(g->N / 64, (g->M + 63) / 64, ns);
    switch (g->epilogue) {
        LAUNCH_TG(SAIS_TG_RAW)
        LAUNCH_TG(SAIS_TG_BIAS)
        LAUNCH_TG(SAIS_TG_BIAS_RELU)
        LAUNCH_TG(SAIS_TG_DRELU)
        default: return SAIS_ERR_ARG;
    }
    return sais_check_launch();
}

// Recommended number of K splits for a SAIS_TG_RAW launch: as many workgroups as fit one round of the chip (256 CUs), at
// least two K-steps per split.  The caller allocates nsplit * M * N floats for the slabs.  Returns 1 for bad arguments.
extern "C" int sais_tgemm_nsplit(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || N % 64 || K % 64) return 1;
    const int tiles = ((M + 63) / 64) * (N / 64), nk = K / 64;
    int best = 1;
    for (int ns = 1; ns <= nk; ++ns)
        if (nk % ns == 0 && nk / ns >= 2 && tiles * ns <= 256) best = ns;
    return best;
}

extern "C" int sais_temporal_ln_fwd(const float* slabs, int nslab, long slab_stride, const float* bias, const float* resid,
                                    int rows, float p_drop, const unsigned long long* rng_state, unsigned site, float* y,
                                    const float* gamma, const float* beta, float eps, float* z, float* mean, float* rstd,
                                    void* stream) {
    SAIS_ENTER();
    if (!slabs || nslab <= 0 || rows <= 0 || !gamma || !beta || !z || (slab_stride & 3)) return SAIS_ERR_ARG;
    if (p_drop < 0.f || p_drop >= 1.f || (p_drop > 0.f && !rng_state)) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(tln_fwd_kernel, dim3((rows + 7) / 8), dim3(256), 0, (hipStream_t)stream, slabs, nslab, slab_stride,
                       bias, resid, rows, p_drop, rng_state, site, y, gamma, beta, eps, z, mean, rstd);
    return sais_check_launch();
}

extern "C" int sais_temporal_ln_bwd(const float* slabs, int nslab, long slab_stride, const float* add, const float* x,
                                    const float* mean, const float* rstd, const float* gamma, int rows, float* dx,
                                    float* dx_drop, float p_drop, const unsigned long long* rng_state, unsigned site,
                                    float* dgamma, float* dbeta, void* stream) {
    SAIS_ENTER();
    if ((!slabs && !add) || (slabs && nslab <= 0) || rows <= 0 || !x || !mean || !rstd || !gamma || !dx) return SAIS_ERR_ARG;
    if ((dgamma == nullptr) != (dbeta == nullptr) || (slab_stride & 3)) return SAIS_ERR_ARG;
    if (dx_drop && (!rng_state || p_drop <= 0.f || p_drop >= 1.f)) return SAIS_ERR_ARG;
    int grid = (rows + 7) / 8;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(tln_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, slabs, slabs ? nslab : 0, slab_stride,
                       add, x, mean, rstd, gamma, rows, dx, dx_drop, p_drop, rng_state, site, dgamma, dbeta);
    return sais_check_launch();
}
