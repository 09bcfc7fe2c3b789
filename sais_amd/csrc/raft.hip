// Correlation pyramid and correlation lookup of RAFT for gfx950 — the two pieces of the optical-flow stage
// (SAIS/scripts/extract_representations.py:62-67,221-252: ptlflow's `raft`, Teed & Deng ECCV 2020) that are not plain
// convolutions.  PARITY UNPINNED: ptlflow 0.2.5 is absent from the reference tree and from this image; the arithmetic follows
// the published model and is tested against oracle/raft_oracle.py (see its header).
//
//   all-pairs correlation  C[i, j] = <f1[:, i], f2[:, j]> / sqrt(256): sais_gemm_nt_f32 (gemm.hip: fp32 operands split into
//       bf16 hi / lo on the matrix cores, fp32-grade) on [H W, 256] feature matrices, one launch per frame pair;
//   sais_raft_corr_pool    the three coarser pyramid levels (2 x 2 average pooling over the LAST two dims of
//       [H W, 1, H, W]) from level 0 in ONE pass: a workgroup stages one correlation row (an H x W image, 32 KB at 540 x 960
//       input) in LDS and emits levels 1-3 — HBM-bound: level 0 is read once (266 MB at that size), 1/4 + 1/16 + 1/64 of it
//       written;
//   sais_raft_lookup       per position and level the (2 r + 1)^2 window around coords / 2^l, bilinear, zeros outside
//       (grid_sample with align_corners=True), with the published offset quirk (window index a runs along x): a gather out
//       of each position's own correlation row; lanes = consecutive positions, so the [B, 4 (2r+1)^2, H, W] output is
//       written in coalesced 256-B segments.
#include "common.hpp"
#include "../../include/sais_hip.h"

namespace {

__global__ __launch_bounds__(256) void corr_pool_kernel(const float* c0, long ld0, int H, int W, float* l1, float* l2, float* l3) {
    extern __shared__ __attribute__((aligned(16))) float sm[];       // level 0 row | level 1 | level 2
    const int row = blockIdx.x, tid = threadIdx.x;
    const int H1 = H >> 1, W1 = W >> 1, H2 = H1 >> 1, W2 = W1 >> 1, H3 = H2 >> 1, W3 = W2 >> 1;
    float* s0 = sm;
    float* s1 = s0 + H * W;
    float* s2 = s1 + H1 * W1;
    const float* src = c0 + (size_t)row * ld0;
    for (int i = tid; i < H * W; i += 256) s0[i] = src[i];
    __syncthreads();
    float* o1 = l1 + (size_t)row * H1 * W1;
    for (int i = tid; i < H1 * W1; i += 256) {
        const int y = i / W1, x = i - y * W1;
        const float* p = s0 + 2 * y * W + 2 * x;
        const float v = (p[0] + p[1] + p[W] + p[W + 1]) * 0.25f;
        s1[i] = v;
        o1[i] = v;
    }
    __syncthreads();
    float* o2 = l2 + (size_t)row * H2 * W2;
    for (int i = tid; i < H2 * W2; i += 256) {
        const int y = i / W2, x = i - y * W2;
        const float* p = s1 + 2 * y * W1 + 2 * x;
        const float v = (p[0] + p[1] + p[W1] + p[W1 + 1]) * 0.25f;
        s2[i] = v;
        o2[i] = v;
    }
    __syncthreads();
    float* o3 = l3 + (size_t)row * H3 * W3;
    for (int i = tid; i < H3 * W3; i += 256) {
        const int y = i / W3, x = i - y * W3;
        const float* p = s2 + 2 * y * W2 + 2 * x;
        o3[i] = (p[0] + p[1] + p[W2] + p[W2 + 1]) * 0.25f;
    }
}

struct LookupParams {
    const float* lvl[4];
    long stride[4];              // floats between consecutive correlation rows of a level
    int Hl[4], Wl[4];
    const float* coords;         // [B, 2, H, W] (x, y)
    float* out;                  // [B, 4 (2r+1)^2, H, W]
    int B, H, W, radius;
};

// 256 threads = 64 consecutive positions x 4 levels (wave w = level w)
__global__ __launch_bounds__(256) void lookup_kernel(LookupParams p) {
    const int lane = threadIdx.x & 63;
    const int l = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int HW = p.H * p.W;
    const long gpos = (long)blockIdx.x * 64 + lane;                  // position index over B * H * W
    if (gpos >= (long)p.B * HW) return;
    const int b = (int)(gpos / HW), pos = (int)(gpos - (long)b * HW);
    const float inv = 1.0f / (float)(1 << l);
    const float cx = p.coords[((size_t)b * 2 + 0) * HW + pos] * inv;
    const float cy = p.coords[((size_t)b * 2 + 1) * HW + pos] * inv;
    const float* img = p.lvl[l] + (size_t)gpos * p.stride[l];
    const int Hl = p.Hl[l], Wl = p.Wl[l], r = p.radius, n = 2 * r + 1;
    // the window offsets are integers: every sample of the window has the same fractional part
    const float fx0 = floorf(cx), fy0 = floorf(cy);
    const float fx = cx - fx0, fy = cy - fy0;
    const int x0 = (int)fx0 - r, y0 = (int)fy0 - r;
    auto tap = [&](int x, int y) { return (x >= 0 && x < Wl && y >= 0 && y < Hl) ? img[y * Wl + x] : 0.f; };
    float* o = p.out + ((size_t)b * 4 * n * n + (size_t)l * n * n) * HW + pos;
    for (int a = 0; a < n; ++a) {                                    // a: along x (the published meshgrid order)
        const int x = x0 + a;
        float left0 = tap(x, y0), right0 = tap(x + 1, y0);
        for (int bb = 0; bb < n; ++bb) {
            const int y = y0 + bb;
            const float left1 = tap(x, y + 1), right1 = tap(x + 1, y + 1);
            const float top = left0 + fx * (right0 - left0), bot = left1 + fx * (right1 - left1);
            o[(size_t)(a * n + bb) * HW] = top + fy * (bot - top);
            left0 = left1;
            right0 = right1;
        }
    }
}

// im2col for the convolutions of the encoders and the update block (round 5: they run as bf16x3 matrix-core GEMMs on
// sais_gemm_nt_f32 instead of MIOpen): cols[pos][k] = x[c][oy * sh - ph + ky][ox * sw - pw + kx], k = (c * kh + ky) * kw + kx
// (the order of weight.view(Cout, Cin * kh * kw)), zero outside the image; column K = Cin * kh * kw holds 1.0 (the bias rides
// in the GEMM as one more weight column), columns K + 1 .. ld - 1 and rows pos >= Ho * Wo are zero (the GEMM's padding).
// One workgroup per 16 positions; consecutive threads write consecutive k of a row (coalesced 1-KiB segments), reads of x are
// gathers out of L2 (an input element is read kh * kw times).
struct Im2colParams {
    const float* x; float* cols;
    int C, H, W, kh, kw, sh, sw, ph, pw, Ho, Wo, K, ld, rows;
};
__global__ __launch_bounds__(256) void im2col_kernel(Im2colParams p) {
    const int npos = p.Ho * p.Wo, khw = p.kh * p.kw;
    for (int r = 0; r < 16; ++r) {
        const int pos = blockIdx.x * 16 + r;
        if (pos >= p.rows) return;
        float* dst = p.cols + (size_t)pos * p.ld;
        const int oy = pos / p.Wo, ox = pos - oy * p.Wo;
        const int y0 = oy * p.sh - p.ph, x0 = ox * p.sw - p.pw;
        for (int k = threadIdx.x; k < p.ld; k += 256) {
            float v = 0.f;
            if (pos < npos) {
                if (k < p.K) {
                    const int c = k / khw, rem = k - c * khw, ky = rem / p.kw, kx = rem - ky * p.kw;
                    const int y = y0 + ky, xx = x0 + kx;
                    if (y >= 0 && y < p.H && xx >= 0 && xx < p.W) v = p.x[((size_t)c * p.H + y) * p.W + xx];
                } else if (k == p.K) {
                    v = 1.0f;
                }
            }
            dst[k] = v;
        }
    }
}

}  // namespace

extern "C" int sais_im2col_f32(const float* x, int C, int H, int W, int kh, int kw, int sh, int sw, int ph, int pw,
                               float* cols, int ld, int rows, void* stream) {
    SAIS_ENTER();
    if (!x || !cols || C <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 || pw < 0) return SAIS_ERR_ARG;
    const int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
    const long K = (long)C * kh * kw;
    if (Ho <= 0 || Wo <= 0 || ld < K + 1 || rows < Ho * Wo) return SAIS_ERR_ARG;
    Im2colParams p{x, cols, C, H, W, kh, kw, sh, sw, ph, pw, Ho, Wo, (int)K, ld, rows};
    hipLaunchKernelGGL(im2col_kernel, dim3((rows + 15) / 16), dim3(256), 0, (hipStream_t)stream, p);
    return sais_check_launch();
}

extern "C" int sais_raft_corr_pool(const float* corr0, long ld0, int rows, int H, int W, float* l1, float* l2, float* l3,
                                   void* stream) {
    SAIS_ENTER();
    if (!corr0 || !l1 || !l2 || !l3 || rows <= 0 || H < 8 || W < 8 || ld0 < (long)H * W) return SAIS_ERR_ARG;
    const int lds = (H * W + (H / 2) * (W / 2) + (H / 4) * (W / 4)) * 4;
    if (lds > 160 * 1024) return SAIS_ERR_ARG;
    static thread_local int granted = 0;
    if (lds > granted) {
        if (hipFuncSetAttribute((const void*)corr_pool_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return SAIS_ERR_LAUNCH;
        granted = lds;
    }
    hipLaunchKernelGGL(corr_pool_kernel, dim3(rows), dim3(256), lds, (hipStream_t)stream, corr0, ld0, H, W, l1, l2, l3);
    return sais_check_launch();
}

extern "C" int sais_raft_lookup(const float* l0, long ld0, const float* l1, const float* l2, const float* l3,
                                const float* coords, int B, int H, int W, int radius, float* out, void* stream) {
    SAIS_ENTER();
    if (!l0 || !l1 || !l2 || !l3 || !coords || !out || B <= 0 || H < 8 || W < 8 || radius < 1 || radius > 8) return SAIS_ERR_ARG;
    LookupParams p{};
    p.lvl[0] = l0; p.lvl[1] = l1; p.lvl[2] = l2; p.lvl[3] = l3;
    int h = H, w = W;
    for (int l = 0; l < 4; ++l) {
        p.Hl[l] = h; p.Wl[l] = w;
        p.stride[l] = l == 0 ? ld0 : (long)h * w;
        h >>= 1; w >>= 1;
    }
    p.coords = coords; p.out = out; p.B = B; p.H = H; p.W = W; p.radius = radius;
    const long npos = (long)B * H * W;
    hipLaunchKernelGGL(lookup_kernel, dim3((unsigned)((npos + 63) / 64)), dim3(256), 0, (hipStream_t)stream, p);
    return sais_check_launch();
}
