// Frame preprocessing on the GPU: decoded uint8 frames -> the float32 [F,3,224,224] tensor the ViT
// takes, bit-identical to the reference's CPU pipeline
//     CenterCrop((0.8 H, 0.8 W)) -> Resize((224,224)) -> ToTensor -> Normalize(mean, std)
// (SurgDataset.__getitem__, dino-main/main_dino.py:295-316; transform at extract_representations.py:158-162), i.e.
// torchvision 0.9.0's center_crop box arithmetic and Pillow's 8-bit ImagingResample (separable antialiased bilinear,
// 22-bit fixed-point coefficients normalised in double, horizontal pass rounded to uint8 before the vertical pass).
//
// Integer byte work: one workgroup per (frame, band of output rows).  The horizontal pass of exactly the input
// rows that band needs goes to LDS as uint8 [rows][224][3]; the vertical pass reads it back, and the
// ToTensor+Normalize step is a [3][256] float table (256 possible bytes per channel: exact by construction).
// Coefficient tables are built on the host in double, as Pillow does, once per frame geometry (a "plan").
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "../../include/sais_hip.h"

namespace {
constexpr int OUT = 224;
constexpr int PRECISION_BITS = 32 - 8 - 2;
constexpr int ROW_BYTES = OUT * 3;              // one horizontally-resampled row in LDS
constexpr int LDS_BUDGET = 60 * 1024;

struct Axis {
    std::vector<int> bounds;   // [OUT][2]  first input index, tap count
    std::vector<int> coef;     // [OUT][ksize]
    int ksize;
};

// Resample.c precompute_coeffs + normalize_coeffs_8bpc, bilinear filter, box = whole axis
Axis make_axis(int in_size) {
    Axis a;
    const double scale = (double)in_size / OUT;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    a.ksize = (int)std::ceil(support) * 2 + 1;
    const double ss = 1.0 / filterscale;
    a.bounds.assign(OUT * 2, 0);
    a.coef.assign((size_t)OUT * a.ksize, 0);
    std::vector<double> w(a.ksize);
    for (int xx = 0; xx < OUT; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            double t = (x + xmin - center + 0.5) * ss;
            if (t < 0.0) t = -t;
            w[x] = t < 1.0 ? 1.0 - t : 0.0;
            ww += w[x];
        }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) w[x] /= ww;
            a.coef[(size_t)xx * a.ksize + x] =
                w[x] < 0 ? (int)(-0.5 + w[x] * (1 << PRECISION_BITS)) : (int)(0.5 + w[x] * (1 << PRECISION_BITS));
        }
        a.bounds[2 * xx] = xmin;
        a.bounds[2 * xx + 1] = xmax;
    }
    return a;
}

template <typename T>
T* upload(const std::vector<T>& v) {
    T* d = nullptr;
    if (hipMalloc(&d, v.size() * sizeof(T)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    return d;
}

struct Geometry {
    int H, W, left, top, cw, ch, kx, ky, band, nbands, max_rows;
    long total_bytes;             // F*H*W*3, set per launch: the last aligned dword of the last row may straddle the end
    const int* xb; const int* xc; const int* yb; const int* yc;
    const int* band_rows;      // [nbands][2] first crop row, row count
    const float* lut;          // [3][256]
};

DEVINL int clip8(int acc) {
    int v = acc >> PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

DEVINL unsigned load_dword(const unsigned char* frames, long a, long total) {
    if (a + 4 <= total) return *(const unsigned*)(frames + a);
    unsigned v = 0;
    for (int j = 0; j < 4 && a + j < total; ++j) v |= (unsigned)frames[a + j] << (8 * j);
    return v;
}

// Horizontal pass, one thread per output column xx, looping over the band's input rows.  The column's 4*G taps
// (Pillow zero-fills a coefficient row past its tap count, so every column can run the same 4*G taps) stay in
// registers; the 12*G bytes they cover are fetched as 3*G+1 aligned dwords and re-aligned with v_alignbyte, which
// cuts the memory instructions per tap from 4 (three byte loads + one coefficient load) to under one.
template <int G>
DEVINL void horizontal_pass(const Geometry& g, const unsigned char* frames, int f, int r0, int nrows, unsigned char* tmp,
                            int tid) {
    if (tid >= OUT) return;
    const int xx = tid;
    int k[4 * G];
#pragma unroll
    for (int x = 0; x < 4 * G; ++x) k[x] = x < g.kx ? g.xc[(size_t)xx * g.kx + x] : 0;
    const long col = ((long)g.left + g.xb[2 * xx]) * 3;
    for (int row = 0; row < nrows; ++row) {
        const long off = ((long)f * g.H + g.top + r0 + row) * g.W * 3 + col;
        const long a = off & ~3L;
        const unsigned sh = (unsigned)(off & 3);
        unsigned dw[3 * G + 1];
#pragma unroll
        for (int d = 0; d <= 3 * G; ++d) dw[d] = load_dword(frames, a + 4 * d, g.total_bytes);
        int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
#pragma unroll
        for (int q = 0; q < G; ++q) {
            const unsigned s0 = __builtin_amdgcn_alignbyte(dw[3 * q + 1], dw[3 * q], sh);
            const unsigned s1 = __builtin_amdgcn_alignbyte(dw[3 * q + 2], dw[3 * q + 1], sh);
            const unsigned s2 = __builtin_amdgcn_alignbyte(dw[3 * q + 3], dw[3 * q + 2], sh);
            const int k0 = k[4 * q], k1 = k[4 * q + 1], k2 = k[4 * q + 2], k3 = k[4 * q + 3];
            a0 += (int)(s0 & 255) * k0;          a1 += (int)((s0 >> 8) & 255) * k0;   a2 += (int)((s0 >> 16) & 255) * k0;
            a0 += (int)(s0 >> 24) * k1;          a1 += (int)(s1 & 255) * k1;          a2 += (int)((s1 >> 8) & 255) * k1;
            a0 += (int)((s1 >> 16) & 255) * k2;  a1 += (int)(s1 >> 24) * k2;          a2 += (int)(s2 & 255) * k2;
            a0 += (int)((s2 >> 8) & 255) * k3;   a1 += (int)((s2 >> 16) & 255) * k3;  a2 += (int)(s2 >> 24) * k3;
        }
        unsigned char* t = tmp + ((size_t)row * OUT + xx) * 3;
        t[0] = (unsigned char)clip8(a0); t[1] = (unsigned char)clip8(a1); t[2] = (unsigned char)clip8(a2);
    }
}

// fallback for very wide frames (more than 32 taps per column): one byte at a time
DEVINL void horizontal_pass_generic(const Geometry& g, const unsigned char* frames, int f, int r0, int nrows,
                                    unsigned char* tmp, int tid) {
    for (int item = tid; item < nrows * OUT; item += 256) {
        const int row = item / OUT, xx = item - row * OUT;
        const int xmin = g.xb[2 * xx], cnt = g.xb[2 * xx + 1];
        const int* k = g.xc + (size_t)xx * g.kx;
        const unsigned char* p = frames + (((size_t)f * g.H + g.top + r0 + row) * g.W + g.left + xmin) * 3;
        int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
        for (int x = 0; x < cnt; ++x) {
            const int kv = k[x];
            a0 += p[3 * x] * kv; a1 += p[3 * x + 1] * kv; a2 += p[3 * x + 2] * kv;
        }
        unsigned char* t = tmp + (size_t)item * 3;
        t[0] = (unsigned char)clip8(a0); t[1] = (unsigned char)clip8(a1); t[2] = (unsigned char)clip8(a2);
    }
}

__global__ __launch_bounds__(256) void preprocess_kernel(Geometry g, const unsigned char* frames, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tmp[];          // [rows][224][3] + lut
    const int band = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    const int r0 = g.band_rows[2 * band], nrows = g.band_rows[2 * band + 1];
    float* lut = (float*)(tmp + ((g.max_rows * ROW_BYTES + 15) & ~15));
    for (int i = tid; i < 3 * 256; i += 256) lut[i] = g.lut[i];

    // horizontal pass: crop rows r0 .. r0+nrows-1  ->  tmp[row][xx][c]
    switch ((g.kx + 3) >> 2) {
        case 1: horizontal_pass<1>(g, frames, f, r0, nrows, tmp, tid); break;
        case 2: horizontal_pass<2>(g, frames, f, r0, nrows, tmp, tid); break;
        case 3: horizontal_pass<3>(g, frames, f, r0, nrows, tmp, tid); break;
        case 4: horizontal_pass<4>(g, frames, f, r0, nrows, tmp, tid); break;
        case 5: horizontal_pass<5>(g, frames, f, r0, nrows, tmp, tid); break;
        case 6: horizontal_pass<6>(g, frames, f, r0, nrows, tmp, tid); break;
        case 7: horizontal_pass<7>(g, frames, f, r0, nrows, tmp, tid); break;
        case 8: horizontal_pass<8>(g, frames, f, r0, nrows, tmp, tid); break;
        default: horizontal_pass_generic(g, frames, f, r0, nrows, tmp, tid);
    }
    __syncthreads();

    // vertical pass + ToTensor/Normalize table: out[f][c][yy][xx], xx fastest; the output row is uniform per round,
    // so its coefficients are scalar loads
    const int y0 = band * g.band;
    const int ny = min(g.band, OUT - y0);
    for (int yl = 0; yl < ny; ++yl) {
        const int yy = y0 + yl;
        const int ymin = g.yb[2 * yy] - r0, cnt = g.yb[2 * yy + 1];
        const int* k = g.yc + (size_t)yy * g.ky;
        for (int i = tid; i < 3 * OUT; i += 256) {
            const int c = i / OUT, xx = i - c * OUT;
            int acc = 1 << (PRECISION_BITS - 1);
            for (int y = 0; y < cnt; ++y) acc += tmp[((size_t)(ymin + y) * OUT + xx) * 3 + c] * k[y];
            out[(((size_t)f * 3 + c) * OUT + yy) * OUT + xx] = lut[c * 256 + clip8(acc)];
        }
    }
}
}  // namespace

struct SaisPreprocessPlan {
    Geometry g;
    int lds_bytes;
    int* d_xb; int* d_xc; int* d_yb; int* d_yc; int* d_band; float* d_lut;
};

extern "C" int sais_preprocess_plan_create(int H, int W, double height_frac, double width_frac, const float* mean3,
                                           const float* std3, SaisPreprocessPlan** plan) {
    SAIS_ENTER();
    if (!plan || !mean3 || !std3 || H <= 0 || W <= 0 || !(height_frac > 0 && height_frac <= 1) ||
        !(width_frac > 0 && width_frac <= 1))
        return SAIS_ERR_ARG;
    // torchvision 0.9.0 center_crop with float sizes, then PIL's Image.crop rounding of every box edge
    // (Python round() = round-half-to-even = nearbyint in the default rounding mode)
    const double ch = height_frac * H, cw = width_frac * W;
    const int top = (int)std::nearbyint((H - ch) / 2.0), left = (int)std::nearbyint((W - cw) / 2.0);
    const int x0 = (int)std::nearbyint((double)left), y0 = (int)std::nearbyint((double)top);
    const int x1 = (int)std::nearbyint(left + cw), y1 = (int)std::nearbyint(top + ch);
    if (x0 < 0 || y0 < 0 || x1 > W || y1 > H || x1 <= x0 || y1 <= y0) return SAIS_ERR_ARG;
    Axis ax = make_axis(x1 - x0), ay = make_axis(y1 - y0);

    // band height: the largest of 8,4,2,1 output rows whose input rows fit the LDS budget
    int band = 8, max_rows = 0;
    std::vector<int> band_rows;
    for (;; band >>= 1) {
        band_rows.clear();
        max_rows = 0;
        for (int b0 = 0; b0 < OUT; b0 += band) {
            int lo = ay.bounds[2 * b0], hi = 0;
            for (int yy = b0; yy < OUT && yy < b0 + band; ++yy) {
                lo = std::min(lo, ay.bounds[2 * yy]);
                hi = std::max(hi, ay.bounds[2 * yy] + ay.bounds[2 * yy + 1]);
            }
            band_rows.push_back(lo);
            band_rows.push_back(hi - lo);
            max_rows = std::max(max_rows, hi - lo);
        }
        if (max_rows * ROW_BYTES + 16 + 3 * 256 * 4 <= LDS_BUDGET || band == 1) break;
    }
    const int fixed_bytes = ((max_rows * ROW_BYTES + 15) & ~15) + 3 * 256 * 4;
    if (fixed_bytes > LDS_BUDGET) return SAIS_ERR_ARG;                   // > ~85x vertical downscale

    std::vector<float> lut(3 * 256);
    for (int c = 0; c < 3; ++c)
        for (int v = 0; v < 256; ++v) {
            volatile float t = (float)v / 255.0f;          // float32 at every step, as torch does
            volatile float u = t - mean3[c];
            lut[c * 256 + v] = u / std3[c];
        }
    SaisPreprocessPlan* p = new SaisPreprocessPlan();
    p->d_xb = upload(ax.bounds); p->d_xc = upload(ax.coef);
    p->d_yb = upload(ay.bounds); p->d_yc = upload(ay.coef);
    p->d_band = upload(band_rows); p->d_lut = upload(lut);
    if (!p->d_xb || !p->d_xc || !p->d_yb || !p->d_yc || !p->d_band || !p->d_lut) {
        sais_preprocess_plan_destroy(p);
        return SAIS_ERR_LAUNCH;
    }
    p->g = Geometry{H, W, x0, y0, x1 - x0, y1 - y0, ax.ksize, ay.ksize, band, (OUT + band - 1) / band, max_rows,
                    0L,
                    p->d_xb, p->d_xc, p->d_yb, p->d_yc, p->d_band, p->d_lut};
    p->lds_bytes = fixed_bytes;
    *plan = p;
    return SAIS_OK;
}

extern "C" int sais_preprocess_plan_box(const SaisPreprocessPlan* plan, int* box4) {
    if (!plan || !box4) return SAIS_ERR_ARG;
    box4[0] = plan->g.left; box4[1] = plan->g.top;
    box4[2] = plan->g.left + plan->g.cw; box4[3] = plan->g.top + plan->g.ch;
    return SAIS_OK;
}

extern "C" int sais_preprocess_run(const SaisPreprocessPlan* plan, const unsigned char* frames, int nframes, float* out,
                                   void* stream) {
    SAIS_ENTER();
    if (!plan || !frames || !out || nframes <= 0) return SAIS_ERR_ARG;
    Geometry g = plan->g;
    g.total_bytes = (long)nframes * g.H * g.W * 3;
    hipLaunchKernelGGL(preprocess_kernel, dim3(g.nbands, nframes), dim3(256), plan->lds_bytes, (hipStream_t)stream,
                       g, frames, out);
    return sais_check_launch();
}

extern "C" void sais_preprocess_plan_destroy(SaisPreprocessPlan* p) {
    if (!p) return;
    (void)hipFree(p->d_xb); (void)hipFree(p->d_xc); (void)hipFree(p->d_yb); (void)hipFree(p->d_yc);
    (void)hipFree(p->d_band); (void)hipFree(p->d_lut);
    delete p;
}
