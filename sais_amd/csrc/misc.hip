// Small HBM-bound helpers of the SAIS hot path (gfx950): patch gather, CLS rows, SGD, weight shadows.
#include "common.hpp"
#include "philox.hpp"
#include "../../include/sais_hip.h"

namespace {

// PatchEmbed (vision_transformer.py:116-131): Conv2d(3->384, k=16, s=16) over NON-overlapping patches
// is a GEMM on the [F*196, 768] patch matrix; column order = (c, py, px) = conv weight flatten.
// One thread = one 16-pixel patch row segment (64-B f32 read, 32-B bf16 write).
// side = frame height = width in pixels (224: 14 x 14 patches; 96: DINO's local crops, 6 x 6).
__global__ __launch_bounds__(256) void patchify_kernel(const float* img, bf16* out, int frames, int side) {
    const int G = side >> 4;
    const long total = (long)frames * G * G * 48;        // 48 = 3 channels * 16 rows
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        // consecutive threads walk px-segments of one image row: idx -> (f, c, y, gx)
        long t = i;
        const int gx = t % G; t /= G;
        const int y = t % side; t /= side;
        const int c = t % 3;
        const int f = t / 3;
        const float* src = img + (((size_t)f * 3 + c) * side + y) * side + gx * 16;
        const int gy = y >> 4, py = y & 15;
        bf16* dst = out + ((size_t)f * G * G + gy * G + gx) * 768 + c * 256 + py * 16;
        bf16x8 lo, hi;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            f32x4 a = *(const f32x4*)(src + 8 * k), b = *(const f32x4*)(src + 8 * k + 4);
            bf16x8& d = k ? hi : lo;
            d[0] = (bf16)a[0]; d[1] = (bf16)a[1]; d[2] = (bf16)a[2]; d[3] = (bf16)a[3];
            d[4] = (bf16)b[0]; d[5] = (bf16)b[1]; d[6] = (bf16)b[2]; d[7] = (bf16)b[3];
        }
        *(bf16x8*)dst = lo;
        *(bf16x8*)(dst + 8) = hi;
    }
}

// prepare_tokens (vision_transformer.py:196-207): token row 0 of every frame = cls_token + pos_embed[0]
__global__ void cls_rows_kernel(const float* cls, const float* pos0, float* tokens, long frame_stride, int frames, int dim) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= frames * dim) return;
    int f = i / dim, c = i - f * dim;
    tokens[(size_t)f * frame_stride + c] = cls[c] + pos0[c];
}

// d cls_token = d pos_embed[0] = sum over frames of d tokens[f, 0, :]
__global__ void cls_rows_bwd_kernel(const float* dtokens, long frame_stride, int frames, int dim, float* dcls, float* dpos0) {
    int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= dim) return;
    float s = 0.f;
    for (int f = 0; f < frames; ++f) s += dtokens[(size_t)f * frame_stride + c];
    atomicAdd(dcls + c, s);
    atomicAdd(dpos0 + c, s);
}

// d pos_embed[1 + q][:] += sum_f dtok[f, 1 + q, :]   (rows 1..196)
__global__ void pos_bwd_kernel(const float* dtokens, int frames, int ntok, int dim, float* dpos) {
    int i = blockIdx.x * 256 + threadIdx.x;            // over (ntok-1)*dim
    if (i >= (ntok - 1) * dim) return;
    int q = i / dim + 1, c = i % dim;
    float s = 0.f;
    for (int f = 0; f < frames; ++f) s += dtokens[((size_t)f * ntok + q) * dim + c];
    atomicAdd(dpos + (size_t)q * dim + c, s);
}

// The three kernels above and below fused: one pass over dtokens [F, ntok, dim].  Workgroup (token q, frame chunk):
// every thread owns 4 columns, sums its chunk of frames (d pos_embed[q], and d cls for q = 0) and writes the bf16
// patch-gradient row on the way; one atomicAdd per column and workgroup at the end.
constexpr int EMB_CHUNKS = 4;
__global__ __launch_bounds__(128) void embed_bwd_kernel(const float* dtokens, int frames, int ntok, int dim, float* dcls,
                                                        float* dpos, bf16* dpatch) {
    const int q = blockIdx.x, c4 = threadIdx.x;
    if (4 * c4 >= dim) return;
    const int per = (frames + EMB_CHUNKS - 1) / EMB_CHUNKS;
    const int f0 = blockIdx.y * per, f1 = min(frames, f0 + per);
    f32x4 s = {0, 0, 0, 0};
#pragma unroll 4
    for (int f = f0; f < f1; ++f) {
        const f32x4 v = *(const f32x4*)(dtokens + ((size_t)f * ntok + q) * dim + 4 * c4);
        s += v;
        if (q > 0) {
            bf16x4 o;
            o[0] = (bf16)v[0]; o[1] = (bf16)v[1]; o[2] = (bf16)v[2]; o[3] = (bf16)v[3];
            *(bf16x4*)(dpatch + ((size_t)f * (ntok - 1) + q - 1) * dim + 4 * c4) = o;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        atomicAdd(dpos + (size_t)q * dim + 4 * c4 + e, s[e]);
        if (q == 0) atomicAdd(dcls + 4 * c4 + e, s[e]);
    }
}

// gather the 196 patch-token rows of every frame into a dense bf16 [F*196, dim] matrix (dY of the patch GEMM)
__global__ void gather_patch_rows_kernel(const float* dtokens, int frames, int ntok, int dim, bf16* out) {
    long total = (long)frames * (ntok - 1) * (dim / 4);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c4 = i % (dim / 4);
        long r = i / (dim / 4);
        int f = r / (ntok - 1), q = r % (ntok - 1);
        f32x4 v = *(const f32x4*)(dtokens + ((size_t)f * ntok + q + 1) * dim + 4 * c4);
        bf16x4 o;
        o[0] = (bf16)v[0]; o[1] = (bf16)v[1]; o[2] = (bf16)v[2]; o[3] = (bf16)v[3];
        *(bf16x4*)(out + (size_t)r * dim + 4 * c4) = o;
    }
}

// vanilla SGD (prepare_model.py:566-567: optim.SGD(params, lr), no momentum / weight decay) fused with the
// refresh of the bf16 shadow the MFMA kernels read.
__global__ __launch_bounds__(256) void sgd_kernel(float* p, const float* g, bf16* shadow, long n, float lr, float gscale) {
    long n4 = n >> 2;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 w = *(f32x4*)(p + 4 * i), d = *(const f32x4*)(g + 4 * i);
        w -= lr * gscale * d;
        *(f32x4*)(p + 4 * i) = w;
        if (shadow) {
            bf16x4 o;
            o[0] = (bf16)w[0]; o[1] = (bf16)w[1]; o[2] = (bf16)w[2]; o[3] = (bf16)w[3];
            *(bf16x4*)(shadow + 4 * i) = o;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        long i = (n4 << 2) + threadIdx.x;
        float w = p[i] - lr * gscale * g[i];
        p[i] = w;
        if (shadow) shadow[i] = (bf16)w;
    }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* src, bf16* dst, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = (bf16)src[i];
}

// dst[m, :] = bf16(rowscale[m] * src[m, :]) over rows of `dim` elements (DropPath: the branch gradient)
__global__ __launch_bounds__(256) void cast_bf16_rows_kernel(const float* src, const float* rowscale, bf16* dst, long rows,
                                                            int dim) {
    const long n = rows * dim;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = (bf16)(src[i] * rowscale[i / dim]);
}

// DropPath (vision_transformer.py:27-46): per sample a Bernoulli(1 - p) keep, output scaled by 1 / (1 - p).  One launch
// fills the per-ROW scale arrays of all branches: out[j][f * rows_per_sample + t] = keep(j, f) / (1 - rate[j]),
// keep from Philox (site = site0 + j, element = sample f).  rate[j] = 0 -> 1.0.
__global__ __launch_bounds__(256) void droppath_scales_kernel(float* out, const float* rates, int nbranch, int samples,
                                                             int rows_per_sample, const unsigned long long* rng,
                                                             unsigned site0) {
    const long per = (long)samples * rows_per_sample, n = per * nbranch;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int j = (int)(i / per);
        const int f = (int)((i - (long)j * per) / rows_per_sample);
        const float p = rates[j];
        float sc = 1.0f;
        if (p > 0.f) sc = philox_keep(rng, site0 + j, (unsigned long long)f, drop_threshold(p)) ? 1.0f / (1.0f - p) : 0.f;
        out[i] = sc;
    }
}

// dst[C,R] = src[R,C]^T (f32 in, bf16 or f32 out); 32x32 tiles through LDS
template <typename TO>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* src, TO* dst, int R, int C) {
    __shared__ float tile[32][33];
    int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < R && c < C) ? src[(size_t)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        int c = c0 + j, r = r0 + tx;
        if (c < C && r < R) dst[(size_t)c * R + r] = (TO)tile[tx][j];
    }
}

// every transposed shadow of a model in one launch: workgroup -> (item, 32x32 tile) through the tile prefix sums
template <typename TO>
__global__ __launch_bounds__(256) void transpose_batch_kernel(const SaisTransposeItem* items, int nitems) {
    __shared__ float tile[32][33];
    const int wg = blockIdx.x;
    int lo = 0, hi = nitems - 1;                       // last item whose tile_begin <= wg
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].tile_begin <= wg) lo = mid; else hi = mid - 1;
    }
    const SaisTransposeItem it = items[lo];
    const int R = it.rows, C = it.cols, tc = (C + 31) / 32, t = wg - it.tile_begin;
    const int c0 = (t % tc) * 32, r0 = (t / tc) * 32;
    const float* src = it.src;
    TO* dst = (TO*)it.dst;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < R && c < C) ? src[(size_t)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        int c = c0 + j, r = r0 + tx;
        if (c < C && r < R) dst[(size_t)c * R + r] = (TO)tile[tx][j];
    }
}

__global__ void scale_kernel(float* p, long n, float s) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] *= s;
}

// out = (resid ? resid : 0) + x * keep / (1 - p): nn.Dropout in train mode, optionally with the residual add that follows it
// in the post-norm TransformerEncoderLayer (src + dropout1(src2)).  out may alias x.
__global__ __launch_bounds__(256) void dropout_kernel(const float* x, const float* resid, float* out, long n, float p,
                                                      const unsigned long long* rng, unsigned sid) {
    const unsigned thr = drop_threshold(p);
    const float inv = 1.0f / (1.0f - p);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = philox_keep(rng, sid, (unsigned long long)i, thr) ? x[i] * inv : 0.f;
        out[i] = resid ? resid[i] + v : v;
    }
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(unsigned char* mask, long n, float p, const unsigned long long* rng,
                                                           unsigned sid) {
    const unsigned thr = drop_threshold(p);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        mask[i] = philox_keep(rng, sid, (unsigned long long)i, thr) ? 1 : 0;
}

__global__ void rng_advance_kernel(unsigned long long* state) { state[1] += 1; }

int grid_for(long n, int per_block = 256) {
    long b = (n + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}
}  // namespace

extern "C" int sais_abi_version(void) { return SAIS_ABI_VERSION; }

static thread_local int g_last_hip_error = 0;
extern "C" void sais_set_last_error(int e) { g_last_hip_error = e; }
extern "C" const char* sais_last_error(void) { return hipGetErrorString((hipError_t)g_last_hip_error); }

extern "C" int sais_patchify(const float* frames_f32, int frames, int side, void* patches_bf16, void* stream) {
    SAIS_ENTER();
    if (!frames_f32 || !patches_bf16 || frames <= 0 || side < 16 || (side & 15)) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_for((long)frames * (side >> 4) * (side >> 4) * 48)), dim3(256), 0,
                       (hipStream_t)stream, frames_f32, (bf16*)patches_bf16, frames, side);
    return sais_check_launch();
}

extern "C" int sais_vit_cls_rows(const float* cls, const float* pos0, float* tokens, long frame_stride, int frames,
                                 int dim, void* stream) {
    SAIS_ENTER();
    if (!cls || !pos0 || !tokens || frames <= 0) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(cls_rows_kernel, dim3((frames * dim + 255) / 256), dim3(256), 0, (hipStream_t)stream, cls, pos0,
                       tokens, frame_stride, frames, dim);
    return sais_check_launch();
}

extern "C" int sais_vit_embed_bwd(const float* dtokens, int frames, int ntok, int dim, float* dcls, float* dpos,
                                  void* dpatch_bf16, void* stream) {
    SAIS_ENTER();
    if (!dtokens || !dcls || !dpos || !dpatch_bf16 || frames <= 0 || (dim & 3)) return SAIS_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dim <= 512) {                 // fused single pass (the ViT case: dim 384)
        hipLaunchKernelGGL(embed_bwd_kernel, dim3(ntok, EMB_CHUNKS), dim3(128), 0, s, dtokens, frames, ntok, dim, dcls, dpos,
                           (bf16*)dpatch_bf16);
        return sais_check_launch();
    }
    hipLaunchKernelGGL(cls_rows_bwd_kernel, dim3((dim + 255) / 256), dim3(256), 0, s, dtokens, (long)ntok * dim, frames,
                       dim, dcls, dpos);
    hipLaunchKernelGGL(pos_bwd_kernel, dim3(((ntok - 1) * dim + 255) / 256), dim3(256), 0, s, dtokens, frames, ntok, dim, dpos);
    hipLaunchKernelGGL(gather_patch_rows_kernel, dim3(grid_for((long)frames * (ntok - 1) * dim / 4)), dim3(256), 0, s,
                       dtokens, frames, ntok, dim, (bf16*)dpatch_bf16);
    return sais_check_launch();
}

extern "C" int sais_sgd_step(float* param, const float* grad, void* shadow_bf16, long n, float lr, float grad_scale,
                             void* stream) {
    SAIS_ENTER();
    if (!param || !grad || n <= 0) return SAIS_ERR_ARG;
    if (((uintptr_t)param | (uintptr_t)grad) & 15) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, param, grad,
                       (bf16*)shadow_bf16, n, lr, grad_scale);
    return sais_check_launch();
}

extern "C" int sais_cast_bf16(const float* src, void* dst_bf16, long n, void* stream) {
    SAIS_ENTER();
    if (!src || !dst_bf16 || n <= 0) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst_bf16, n);
    return sais_check_launch();
}

extern "C" int sais_transpose_cast_bf16(const float* src, int rows, int cols, void* dst_bf16, void* stream) {
    SAIS_ENTER();
    if (!src || !dst_bf16 || rows <= 0 || cols <= 0) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(transpose_cast_kernel<bf16>, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0,
                       (hipStream_t)stream, src, (bf16*)dst_bf16, rows, cols);
    return sais_check_launch();
}

extern "C" int sais_transpose_f32(const float* src, int rows, int cols, float* dst, void* stream) {
    SAIS_ENTER();
    if (!src || !dst || rows <= 0 || cols <= 0) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(transpose_cast_kernel<float>, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0,
                       (hipStream_t)stream, src, dst, rows, cols);
    return sais_check_launch();
}

extern "C" int sais_transpose_batch(const SaisTransposeItem* items_dev, int nitems, int total_tiles, int dst_is_f32,
                                    void* stream) {
    SAIS_ENTER();
    if (!items_dev || nitems <= 0 || total_tiles <= 0) return SAIS_ERR_ARG;
    if (dst_is_f32)
        hipLaunchKernelGGL(transpose_batch_kernel<float>, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream,
                           items_dev, nitems);
    else
        hipLaunchKernelGGL(transpose_batch_kernel<bf16>, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream,
                           items_dev, nitems);
    return sais_check_launch();
}

// Pull a byte range towards the GPU (L2 / Infinity Cache) without using it: one discarded 16-B load per lane.  Used in front
// of the temporal encoder, whose ~70 launches are a few microseconds each and otherwise pay a first-touch HBM round trip for
// their weights in every one of them (the ViT's gigabytes evicted them since the last step).
__global__ __launch_bounds__(256) void touch_kernel(const u32x4* p, long n16) {
    u32x4 acc = {0, 0, 0, 0};
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n16; i += (long)gridDim.x * 256) acc |= p[i];
    asm volatile("" ::"v"(acc));
}

extern "C" int sais_touch(const void* p, long bytes, void* stream) {
    SAIS_ENTER();
    if (!p || bytes <= 0 || ((uintptr_t)p & 15)) return SAIS_ERR_ARG;
    const long n16 = bytes / 16;
    if (n16 == 0) return SAIS_OK;
    hipLaunchKernelGGL(touch_kernel, dim3(grid_for(n16)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)p, n16);
    return sais_check_launch();
}

extern "C" int sais_scale_f32(float* p, long n, float s, void* stream) {
    SAIS_ENTER();
    if (!p || n <= 0) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, n, s);
    return sais_check_launch();
}

// ---- train-mode dropout (philox.hpp) ---------------------------------------------------------------------------
extern "C" int sais_rng_advance(unsigned long long* state, void* stream) {
    SAIS_ENTER();
    if (!state) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
    return sais_check_launch();
}

extern "C" int sais_dropout_f32(const float* x, const float* resid, float* out, long n, float p,
                                const unsigned long long* rng_state, unsigned site, void* stream) {
    SAIS_ENTER();
    if (!x || !out || !rng_state || n <= 0 || p < 0.f || p >= 1.f) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, resid, out, n, p, rng_state,
                       site);
    return sais_check_launch();
}

extern "C" int sais_dropout_mask(unsigned char* mask, long n, float p, const unsigned long long* rng_state, unsigned site,
                                 void* stream) {
    SAIS_ENTER();
    if (!mask || !rng_state || n <= 0 || p < 0.f || p >= 1.f) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, mask, n, p, rng_state, site);
    return sais_check_launch();
}

extern "C" int sais_cast_bf16_rows(const float* src, const float* rowscale, void* dst_bf16, long rows, int dim, void* stream) {
    SAIS_ENTER();
    if (!src || !rowscale || !dst_bf16 || rows <= 0 || dim <= 0) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(cast_bf16_rows_kernel, dim3(grid_for(rows * dim)), dim3(256), 0, (hipStream_t)stream, src, rowscale,
                       (bf16*)dst_bf16, rows, dim);
    return sais_check_launch();
}

extern "C" int sais_droppath_scales(float* out, const float* rates_dev, int nbranch, int samples, int rows_per_sample,
                                    const unsigned long long* rng_state, unsigned site0, void* stream) {
    SAIS_ENTER();
    if (!out || !rates_dev || !rng_state || nbranch <= 0 || samples <= 0 || rows_per_sample <= 0) return SAIS_ERR_ARG;
    hipLaunchKernelGGL(droppath_scales_kernel, dim3(grid_for((long)nbranch * samples * rows_per_sample)), dim3(256), 0,
                       (hipStream_t)stream, out, rates_dev, nbranch, samples, rows_per_sample, rng_state, site0);
    return sais_check_launch();
}
