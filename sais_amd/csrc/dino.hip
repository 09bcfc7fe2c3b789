// DINO pre-training objective on gfx950 (SAIS/scripts/dino-main/main_dino.py): everything around the ViT that the
// self-distillation step needs and that is not a GEMM —
//   * DINOLoss.forward / update_center (main_dino.py:579-630): centred + sharpened teacher softmax, student
//     log-softmax, the (teacher view, student view) cross-entropy sum and its gradient, the [1, out_dim] centre EMA;
//   * DINOHead's non-GEMM pieces (vision_transformer.py:257-291): exact-erf GELU, F.normalize, nn.utils.weight_norm;
//   * interpolate_pos_encoding (vision_transformer.py:174-194) as a fixed [P_out, P_in] linear map and its transpose;
//   * the optimizer tail of train_one_epoch (main_dino.py:541-566): per-parameter gradient clipping
//     (utils.clip_gradients), AdamW on two parameter groups (utils.get_params_groups), the frozen last layer
//     (utils.cancel_gradients_last_layer) and the EMA teacher — one pass over the flat parameter buffers.
// All of it is HBM-bound streaming work over [rows, out_dim = 65536] logits or the 44 M-element parameter buffers:
// 16-B accesses, one pass per tensor, reductions in LDS, fixed summation order (no float atomics: bit-reproducible).
#include "common.hpp"
#include "../../include/sais_hip.h"

namespace {

DEVINL float block_sum(float v, float* red, int tid, int nthreads) {      // every thread gets the total
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (nthreads >> 6); ++w) t += red[w];
    return t;
}
DEVINL float block_max(float v, float* red, int tid, int nthreads) {
    v = wave_max(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float t = -INFINITY;
    for (int w = 0; w < (nthreads >> 6); ++w) t = fmaxf(t, red[w]);
    return t;
}

// ------------------------------------------------------------------------------------------ DINOLoss
// lse[r] = log sum_k exp((x[r][k] - center[k]) * scale): the normaliser of softmax((t - c) / temp) (teacher rows,
// main_dino.py:605) and of log_softmax(s / student_temp) (student rows, center = NULL, :600,614).
// One 1024-thread workgroup per row; the row (256 KiB at out_dim 65536) is read twice, the second time from L2.
__global__ __launch_bounds__(1024) void dino_row_lse_kernel(const float* x, long ld, int n, float scale, const float* center,
                                                            float* lse) {
    __shared__ float red[16];
    const int tid = threadIdx.x;
    const float* row = x + (size_t)blockIdx.x * ld;
    float m = -INFINITY;
    for (int i = 4 * tid; i < n; i += 4096) {
        f32x4 v = *(const f32x4*)(row + i);
        if (center) v -= *(const f32x4*)(center + i);
        m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
    }
    m = block_max(m, red, tid, 1024) * scale;                  // scale > 0
    float s = 0.f;
    for (int i = 4 * tid; i < n; i += 4096) {
        f32x4 v = *(const f32x4*)(row + i);
        if (center) v -= *(const f32x4*)(center + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) s += __expf(__builtin_fmaf(v[e], scale, -m));
    }
    s = block_sum(s, red, tid, 1024);
    if (tid == 0) lse[blockIdx.x] = m + __logf(s);
}

// Workgroup (column chunk of 1024, sample b): reads the 2 teacher rows and the ncrops student rows of sample b once,
// writes the ncrops gradient rows and one partial of the loss.
//   q_i = exp((t_i - c) / temp - lse_t_i)            i = 0, 1  (teacher views = global crops, rows i B + b)
//   logp_v = s_v / tau - lse_s_v,  p_v = exp(logp_v)  v < ncrops (student views, rows v B + b)
//   loss   = -1 / (n_terms B) sum_b sum_k sum_{i != v} q_i logp_v                              (:609-617)
//   dloss / ds_v = 1 / (n_terms B tau) (n_i(v) p_v - sum_{i != v} q_i),  n_i(v) = #{i != v}   (sum_k q_i = 1)
__global__ __launch_bounds__(256) void dino_loss_grad_kernel(const float* student, long lds_, const float* teacher, long ldt,
                                                             const float* center, const float* s_lse, const float* t_lse,
                                                             int B, int ncrops, int n, float inv_tau, float inv_temp,
                                                             float coef, float* dlogits, long ldd, float* partials) {
    __shared__ float red[4];
    const int tid = threadIdx.x, b = blockIdx.y;
    const int k = blockIdx.x * 1024 + 4 * tid;
    float acc = 0.f;
    if (k < n) {
        const f32x4 c4 = *(const f32x4*)(center + k);
        f32x4 q[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x4 t = *(const f32x4*)(teacher + (size_t)(i * B + b) * ldt + k);
            const float l = t_lse[i * B + b];
#pragma unroll
            for (int e = 0; e < 4; ++e) q[i][e] = __expf(__builtin_fmaf(t[e] - c4[e], inv_temp, -l));
        }
        for (int v = 0; v < ncrops; ++v) {
            const size_t r = (size_t)v * B + b;
            const f32x4 s = *(const f32x4*)(student + r * lds_ + k);
            const float l = s_lse[r];
            const float w0 = v != 0 ? 1.f : 0.f, w1 = v != 1 ? 1.f : 0.f;
            f32x4 d;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lp = __builtin_fmaf(s[e], inv_tau, -l);
                const float qs = w0 * q[0][e] + w1 * q[1][e];
                acc = __builtin_fmaf(qs, lp, acc);
                d[e] = coef * ((w0 + w1) * __expf(lp) - qs);
            }
            *(f32x4*)(dlogits + r * ldd + k) = d;
        }
    }
    acc = block_sum(acc, red, tid, 256);
    if (tid == 0) partials[(size_t)b * gridDim.x + blockIdx.x] = acc;
}

// loss = scale * sum(partials), summed in double in a fixed order by one workgroup
__global__ __launch_bounds__(256) void dino_loss_reduce_kernel(const float* partials, int count, float scale, float* loss) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += 256) s += (double)partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(red[0] * (double)scale);
}

// batch_center = sum over the teacher rows (main_dino.py:626): 64 column groups x 4 row groups per workgroup,
// the row groups combined through LDS in a fixed order
__global__ __launch_bounds__(256) void dino_colsum_kernel(const float* x, long ld, int rows, int n, float* out) {
    __shared__ f32x4 red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int k = (blockIdx.x * 64 + tx) * 4;
    f32x4 s = {0, 0, 0, 0};
    if (k < n)
        for (int r = ty; r < rows; r += 4) s += *(const f32x4*)(x + (size_t)r * ld + k);
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && k < n) *(f32x4*)(out + k) = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}

// center = center * momentum + colsum * inv_count * (1 - momentum)   (:627-630; inv_count = 1 / (rows * world))
__global__ __launch_bounds__(256) void dino_center_ema_kernel(float* center, const float* colsum, int n, float momentum,
                                                              float inv_count) {
    const int k = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (k >= n) return;
    const f32x4 c = *(const f32x4*)(center + k), s = *(const f32x4*)(colsum + k);
    *(f32x4*)(center + k) = c * momentum + (s * inv_count) * (1.0f - momentum);
}

// ------------------------------------------------------------------------------------------ DINOHead pieces
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* u, float* h, long n4) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = *(const f32x4*)(u + 4 * i);
        f32x2 a, b;
        gelu_erf2(f32x2{v[0], v[1]}, a);
        gelu_erf2(f32x2{v[2], v[3]}, b);
        *(f32x4*)(h + 4 * i) = f32x4{a.x, a.y, b.x, b.y};
    }
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* dh, const float* u, float* du, long n4) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = *(const f32x4*)(u + 4 * i), g = *(const f32x4*)(dh + 4 * i);
        f32x2 a, b;
        dgelu_erf2(f32x2{v[0], v[1]}, a);
        dgelu_erf2(f32x2{v[2], v[3]}, b);
        *(f32x4*)(du + 4 * i) = f32x4{g[0] * a.x, g[1] * a.y, g[2] * b.x, g[3] * b.y};
    }
}

// bf16x3 operand images for the bf16 MFMA GEMM (the 65536-wide last layer): x = hi + lo with hi = bf16(x), lo = bf16(x - hi);
// A side [hi | hi | lo], B side [hi | lo | hi] along K, so that ONE bf16 GEMM with K' = 3 K accumulates hi.hi + hi.lo + lo.hi
// in fp32 — the three products of sais_gemm_nt_f32 on the tuned 128 x 128 bf16 kernel instead of the small-M fp32 one.
__global__ __launch_bounds__(256) void split3_kernel(const float* src, long ld, int rows, int K, bf16* dst, int b_side) {
    const int k4 = K >> 2;
    const long total = (long)rows * k4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / k4), c = 4 * (int)(i - (long)r * k4);
        const f32x4 v = *(const f32x4*)(src + (size_t)r * ld + c);
        bf16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi[e] = (bf16)v[e];
            lo[e] = (bf16)(v[e] - (float)hi[e]);
        }
        bf16* d = dst + (size_t)r * 3 * K + c;
        *(bf16x4*)d = hi;
        *(bf16x4*)(d + K) = b_side ? lo : hi;
        *(bf16x4*)(d + 2 * K) = b_side ? hi : lo;
    }
}

// one wave per row of a [rows, dim] matrix (dim % 4 == 0, dim <= 1024: <= 4 float4 per lane)
constexpr int ROWV = 4;
DEVINL int load_row(const float* p, int dim, int lane, f32x4 (&v)[ROWV]) {
    int n = 0;
#pragma unroll
    for (int j = 0; j < ROWV; ++j) {
        const int c = 4 * (lane + 64 * j);
        v[j] = c < dim ? *(const f32x4*)(p + c) : f32x4{0, 0, 0, 0};
        n += c < dim;
    }
    return n;
}
DEVINL void store_row(float* p, int dim, int lane, const f32x4 (&v)[ROWV]) {
#pragma unroll
    for (int j = 0; j < ROWV; ++j) {
        const int c = 4 * (lane + 64 * j);
        if (c < dim) *(f32x4*)(p + c) = v[j];
    }
}
DEVINL float dot_rows(const f32x4 (&a)[ROWV], const f32x4 (&b)[ROWV]) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < ROWV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) s = __builtin_fmaf(a[j][e], b[j][e], s);
    return wave_sum(s);
}

// F.normalize(z, dim=-1, p=2) (vision_transformer.py:289): out = z / max(||z||, eps); inv = 1 / max(||z||, eps)
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* z, int rows, int dim, float eps, float* out, float* inv) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    f32x4 v[ROWV];
    load_row(z + (size_t)r * dim, dim, lane, v);
    const float iv = 1.0f / fmaxf(sqrtf(dot_rows(v, v)), eps);
#pragma unroll
    for (int j = 0; j < ROWV; ++j) v[j] *= iv;
    store_row(out + (size_t)r * dim, dim, lane, v);
    if (lane == 0) inv[r] = iv;
}
// dz = inv (dout - out (dout . out))   (the clamp branch ||z|| < eps has dz = dout / eps: out . dout term dropped)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* dout, const float* out, const float* inv, int rows, int dim,
                                                         float eps, float* dz) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    f32x4 g[ROWV], o[ROWV];
    load_row(dout + (size_t)r * dim, dim, lane, g);
    load_row(out + (size_t)r * dim, dim, lane, o);
    const float iv = inv[r];
    const float d = iv * eps >= 1.0f ? 0.f : dot_rows(g, o);
#pragma unroll
    for (int j = 0; j < ROWV; ++j) g[j] = (g[j] - o[j] * d) * iv;
    store_row(dz + (size_t)r * dim, dim, lane, g);
}

// nn.utils.weight_norm(Linear(256, out_dim, bias=False)) (vision_transformer.py:277-281): w = g v / ||v||_row
__global__ __launch_bounds__(256) void weight_norm_fwd_kernel(const float* v, const float* g, int rows, int dim, float* w,
                                                              float* inv) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    f32x4 x[ROWV];
    load_row(v + (size_t)r * dim, dim, lane, x);
    const float iv = 1.0f / sqrtf(dot_rows(x, x));
    const float s = g[r] * iv;
#pragma unroll
    for (int j = 0; j < ROWV; ++j) x[j] *= s;
    store_row(w + (size_t)r * dim, dim, lane, x);
    if (lane == 0) inv[r] = iv;
}
// dv += g / ||v|| (dw - v (dw . v) / ||v||^2) ;  dg += (dw . v) / ||v||   (dg may be NULL: norm_last_layer freezes g)
__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(const float* dw, const float* v, const float* g, const float* inv,
                                                              int rows, int dim, float* dv, float* dg) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    f32x4 d[ROWV], x[ROWV], o[ROWV];
    load_row(dw + (size_t)r * dim, dim, lane, d);
    load_row(v + (size_t)r * dim, dim, lane, x);
    load_row(dv + (size_t)r * dim, dim, lane, o);
    const float iv = inv[r], dot = dot_rows(d, x), s = g[r] * iv, t = dot * iv * iv;
#pragma unroll
    for (int j = 0; j < ROWV; ++j) o[j] += (d[j] - x[j] * t) * s;
    store_row(dv + (size_t)r * dim, dim, lane, o);
    if (dg && lane == 0) dg[r] += dot * iv;
}

// ------------------------------------------------------------------------------------------ positional table
// out[0] = pos[0];  out[1 + o] = sum_j Wm[o][j] pos[1 + j]      (the bicubic map has 16 non-zeros per row)
__global__ __launch_bounds__(128) void pos_interp_fwd_kernel(const float* Wm, int nout, int nin, const float* pos, int dim,
                                                             float* out) {
    const int o = blockIdx.x, c = 4 * threadIdx.x;
    if (c >= dim) return;
    f32x4 s = {0, 0, 0, 0};
    if (o == 0) {
        s = *(const f32x4*)(pos + c);
    } else {
        const float* wr = Wm + (size_t)(o - 1) * nin;
        for (int j = 0; j < nin; ++j) {
            const float w = wr[j];
            if (w != 0.f) s += *(const f32x4*)(pos + (size_t)(1 + j) * dim + c) * w;
        }
    }
    *(f32x4*)(out + (size_t)o * dim + c) = s;
}
// dpos[0] += dout[0];  dpos[1 + j] += sum_o Wm[o][j] dout[1 + o]
__global__ __launch_bounds__(128) void pos_interp_bwd_kernel(const float* Wm, int nout, int nin, const float* dout, int dim,
                                                             float* dpos) {
    const int j = blockIdx.x, c = 4 * threadIdx.x;
    if (c >= dim) return;
    f32x4 s = {0, 0, 0, 0};
    if (j == 0) {
        s = *(const f32x4*)(dout + c);
    } else {
        for (int o = 0; o < nout; ++o) {
            const float w = Wm[(size_t)o * nin + j - 1];
            if (w != 0.f) s += *(const f32x4*)(dout + (size_t)(1 + o) * dim + c) * w;
        }
    }
    float* p = dpos + (size_t)j * dim + c;
    *(f32x4*)p = *(const f32x4*)p + s;
}

// ------------------------------------------------------------------------------------------ optimizer tail
// The flat parameter buffer is cut into chunks of <= CHUNK elements that never straddle a tensor: chunk c covers
// [off, off + len) of tensor seg.  (Tensor starts are 16-B aligned and padded with zeros: sais_amd/flat.py.)
constexpr int CHUNK = 8192;

__global__ __launch_bounds__(256) void seg_sqnorm_partial_kernel(const float* grad, const SaisOptChunk* chunks, float* partial) {
    __shared__ float red[4];
    const SaisOptChunk ch = chunks[blockIdx.x];
    const float* g = grad + ch.off;
    float s = 0.f;
    for (int i = 4 * threadIdx.x; i < ch.len; i += 1024) {
        const f32x4 v = *(const f32x4*)(g + i);
        s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    s = block_sum(s, red, threadIdx.x, 256);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// norms[seg] = sqrt(sum of the tensor's chunk partials), one wave per tensor, fixed order, double accumulation
__global__ __launch_bounds__(64) void seg_sqnorm_final_kernel(const float* partial, const int* seg_first_chunk, float scale,
                                                              float* norms) {
    const int seg = blockIdx.x, c0 = seg_first_chunk[seg], c1 = seg_first_chunk[seg + 1];
    double s = 0.0;
    for (int c = c0 + threadIdx.x; c < c1; c += 64) s += (double)partial[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) norms[seg] = (float)sqrt(s) * scale;
}

__global__ __launch_bounds__(256) void adamw_ema_kernel(SaisAdamW a) {
    const SaisOptChunk ch = a.chunks[blockIdx.x];
    const int flags = a.seg_flags[ch.seg];
    const int cls = (flags & SAIS_OPT_CLASS1) ? 1 : 0;
    const bool update = !(flags & SAIS_OPT_NO_GRAD) && !(cls && a.frozen1);
    float coef = 1.0f;
    if (a.clip > 0.f) {                                        // utils.clip_gradients: per-parameter
        const float c = a.clip / (a.norms[ch.seg] + 1e-6f);
        coef = c < 1.0f ? c : 1.0f;
    }
    const float decay = 1.0f - a.lr * ((flags & SAIS_OPT_DECAY) ? a.weight_decay : 0.f);
    const float step = a.lr / a.bc1[cls], rs2 = 1.0f / a.sqrt_bc2[cls];
    const float om1 = 1.0f - a.beta1, om2 = 1.0f - a.beta2, ome = 1.0f - a.ema_m;
    for (int i = 4 * threadIdx.x; i < ch.len; i += 1024) {
        const long o = ch.off + i;
        f32x4 p = *(const f32x4*)(a.param + o);
        if (update) {
            const f32x4 g = *(const f32x4*)(a.grad + o) * (coef * a.grad_scale);
            f32x4 m = *(const f32x4*)(a.exp_avg + o), v = *(const f32x4*)(a.exp_avg_sq + o);
            p *= decay;
            m += (g - m) * om1;
            v = v * a.beta2 + (g * g) * om2;
#pragma unroll
            for (int e = 0; e < 4; ++e) p[e] -= step * m[e] / (sqrtf(v[e]) * rs2 + a.eps);
            *(f32x4*)(a.exp_avg + o) = m;
            *(f32x4*)(a.exp_avg_sq + o) = v;
            *(f32x4*)(a.param + o) = p;
            if (a.param16) {
                bf16x4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (bf16)p[e];
                *(bf16x4*)((bf16*)a.param16 + o) = h;
            }
        }
        if (a.teacher) {                                       // main_dino.py:563-566
            f32x4 t = *(const f32x4*)(a.teacher + o);
            t = t * a.ema_m + p * ome;
            *(f32x4*)(a.teacher + o) = t;
            if (a.teacher16) {
                bf16x4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (bf16)t[e];
                *(bf16x4*)((bf16*)a.teacher16 + o) = h;
            }
        }
    }
}
}  // namespace

// ============================================================================================ C ABI
extern "C" int sais_dino_row_lse(const float* x, long ld, int rows, int n, float scale, const float* center, float* lse,
                                 void* stream) {
    SAIS_ENTER();
    if (!x || !lse || rows <= 0 || n <= 0 || (n & 3) || (ld & 3) || !(scale > 0.f)) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(dino_row_lse_kernel, dim3(rows), dim3(1024), 0, (hipStream_t)stream, x, ld, n, scale, center, lse);
    return sais_check_launch();
}

extern "C" int sais_dino_loss_partials(int B, int n) { return B > 0 && n > 0 ? B * ((n + 1023) / 1024) : 0; }

extern "C" int sais_dino_loss(const float* student, long lds, const float* teacher, long ldt, const float* center,
                              const float* s_lse, const float* t_lse, int B, int ncrops, int n, float student_temp,
                              float teacher_temp, float* dlogits, long ldd, float* partials, float* loss, void* stream) {
    SAIS_ENTER();
    if (!student || !teacher || !center || !s_lse || !t_lse || !dlogits || !partials || !loss) return SAIS_ERR_ARG;
    if (B <= 0 || ncrops < 2 || n <= 0 || (n & 3) || (lds & 3) || (ldt & 3) || (ldd & 3) || !(student_temp > 0.f) ||
        !(teacher_temp > 0.f))
        return SAIS_ERR_ARG;
    const int n_terms = 2 * ncrops - 2;                        // (teacher view, student view) pairs with v != iq
    const int nchunk = (n + 1023) / 1024;
    const float coef = 1.0f / ((float)n_terms * (float)B * student_temp);
    hipLaunchKernelGGL(dino_loss_grad_kernel, dim3(nchunk, B), dim3(256), 0, (hipStream_t)stream, student, lds, teacher, ldt,
                       center, s_lse, t_lse, B, ncrops, n, 1.0f / student_temp, 1.0f / teacher_temp, coef, dlogits, ldd,
                       partials);
    hipLaunchKernelGGL(dino_loss_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, nchunk * B,
                       -1.0f / ((float)n_terms * (float)B), loss);
    return sais_check_launch();
}

extern "C" int sais_dino_colsum(const float* x, long ld, int rows, int n, float* out, void* stream) {
    SAIS_ENTER();
    if (!x || !out || rows <= 0 || n <= 0 || (n & 3) || (ld & 3)) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(dino_colsum_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, ld, rows, n, out);
    return sais_check_launch();
}

extern "C" int sais_dino_center_ema(float* center, const float* colsum, int n, float momentum, float inv_count,
                                    void* stream) {
    SAIS_ENTER();
    if (!center || !colsum || n <= 0 || (n & 3)) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(dino_center_ema_kernel, dim3((n + 1023) / 1024), dim3(256), 0, (hipStream_t)stream, center, colsum, n,
                       momentum, inv_count);
    return sais_check_launch();
}

static int ew_grid(long n4) {
    long g = (n4 + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

extern "C" int sais_gelu_fwd_f32(const float* u, float* h, long n, void* stream) {
    SAIS_ENTER();
    if (!u || !h || n <= 0 || (n & 3)) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, u, h, n / 4);
    return sais_check_launch();
}
extern "C" int sais_gelu_bwd_f32(const float* dh, const float* u, float* du, long n, void* stream) {
    SAIS_ENTER();
    if (!dh || !u || !du || n <= 0 || (n & 3)) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, dh, u, du, n / 4);
    return sais_check_launch();
}

extern "C" int sais_split_bf16x3(const float* src, long ld, int rows, int cols, void* dst_bf16, int b_side, void* stream) {
    SAIS_ENTER();
    if (!src || !dst_bf16 || rows <= 0 || cols <= 0 || (cols & 3) || (ld & 3)) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(split3_kernel, dim3(ew_grid((long)rows * cols / 4)), dim3(256), 0, (hipStream_t)stream, src, ld, rows,
                       cols, (bf16*)dst_bf16, b_side);
    return sais_check_launch();
}

extern "C" int sais_l2norm_fwd(const float* z, int rows, int dim, float eps, float* out, float* inv, void* stream) {
    SAIS_ENTER();
    if (!z || !out || !inv || rows <= 0 || dim <= 0 || (dim & 3) || dim > 1024) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, z, rows, dim, eps, out, inv);
    return sais_check_launch();
}
extern "C" int sais_l2norm_bwd(const float* dout, const float* out, const float* inv, int rows, int dim, float eps, float* dz,
                               void* stream) {
    SAIS_ENTER();
    if (!dout || !out || !inv || !dz || rows <= 0 || dim <= 0 || (dim & 3) || dim > 1024) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, dout, out, inv, rows, dim,
                       eps, dz);
    return sais_check_launch();
}

extern "C" int sais_weight_norm_fwd(const float* v, const float* g, int rows, int dim, float* w, float* inv, void* stream) {
    SAIS_ENTER();
    if (!v || !g || !w || !inv || rows <= 0 || dim <= 0 || (dim & 3) || dim > 1024) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(weight_norm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, v, g, rows, dim, w, inv);
    return sais_check_launch();
}
extern "C" int sais_weight_norm_bwd(const float* dw, const float* v, const float* g, const float* inv, int rows, int dim,
                                    float* dv, float* dg, void* stream) {
    SAIS_ENTER();
    if (!dw || !v || !g || !inv || !dv || rows <= 0 || dim <= 0 || (dim & 3) || dim > 1024) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, dw, v, g, inv, rows,
                       dim, dv, dg);
    return sais_check_launch();
}

extern "C" int sais_pos_interp_fwd(const float* Wm, int nout, int nin, const float* pos, int dim, float* out, void* stream) {
    SAIS_ENTER();
    if (!Wm || !pos || !out || nout <= 0 || nin <= 0 || dim <= 0 || (dim & 3) || dim > 512) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(pos_interp_fwd_kernel, dim3(nout + 1), dim3(128), 0, (hipStream_t)stream, Wm, nout, nin, pos, dim, out);
    return sais_check_launch();
}
extern "C" int sais_pos_interp_bwd(const float* Wm, int nout, int nin, const float* dout, int dim, float* dpos, void* stream) {
    SAIS_ENTER();
    if (!Wm || !dout || !dpos || nout <= 0 || nin <= 0 || dim <= 0 || (dim & 3) || dim > 512) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(pos_interp_bwd_kernel, dim3(nin + 1), dim3(128), 0, (hipStream_t)stream, Wm, nout, nin, dout, dim, dpos);
    return sais_check_launch();
}

extern "C" int sais_opt_chunk_elems(void) { return CHUNK; }

extern "C" int sais_grad_norms(const float* grad, const SaisOptChunk* chunks, int nchunks, const int* seg_first_chunk,
                               int nseg, float scale, float* partial_ws, float* norms, void* stream) {
    SAIS_ENTER();
    if (!grad || !chunks || !seg_first_chunk || !partial_ws || !norms || nchunks <= 0 || nseg <= 0) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(seg_sqnorm_partial_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, grad, chunks, partial_ws);
    hipLaunchKernelGGL(seg_sqnorm_final_kernel, dim3(nseg), dim3(64), 0, (hipStream_t)stream, partial_ws, seg_first_chunk, scale, norms);
    return sais_check_launch();
}

extern "C" int sais_adamw_ema_step(const SaisAdamW* a, void* stream) {
    SAIS_ENTER();
    if (!a || !a->param || !a->grad || !a->exp_avg || !a->exp_avg_sq || !a->chunks || !a->seg_flags || a->nchunks <= 0)
        return SAIS_ERR_ARG;
    if (a->clip > 0.f && !a->norms) return SAIS_ERR_ARG;
    if (!(a->bc1[0] > 0.f) || !(a->sqrt_bc2[0] > 0.f) || !(a->bc1[1] > 0.f) || !(a->sqrt_bc2[1] > 0.f)) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(adamw_ema_kernel, dim3(a->nchunks), dim3(256), 0, (hipStream_t)stream, *a);
    return sais_check_launch();
}
