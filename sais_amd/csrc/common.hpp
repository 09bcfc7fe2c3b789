// Shared device helpers for the gfx950 (CDNA4 / MI355X) kernels of the SAIS hot path.
// Wave = 64 lanes; MFMA = v_mfma_f32_16x16x32_bf16 (fp32 accumulate) unless stated otherwise.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define SAIS_OK 0
#define SAIS_ERR_ARG (-1)
#define SAIS_ERR_LAUNCH (-2)

#define DEVINL __device__ __forceinline__

// D[i][j] += sum_k A[i][k] B[k][j]; lane l holds A[i=l&15][k=8*(l>>4)+e], B[k=8*(l>>4)+e][j=l&15],
// D[i=4*(l>>4)+r][j=l&15]  (cdna_hip_programming.md §3).
DEVINL f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-col block of 16-bit elements, delivered
// column-major: lane 4q+p supplies the address of row q, cols 4p..4p+3; lane i receives
// column i of the 4 rows (row q in element q).  EXEC must be all ones; address 8-B aligned.
DEVINL bf16x4 lds_read_tr16(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(p));
}

DEVINL bf16x8 cat4(bf16x4 a, bf16x4 b) {
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

DEVINL bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (bf16)0.0f;
    return z;
}

DEVINL float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
DEVINL float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// exact-erf GELU (nn.GELU(), vision_transformer.py:49-65) without libm's erff: Abramowitz-Stegun 7.1.26,
//   erf(x) = 1 - (a1 t + ... + a5 t^5) exp(-x^2),  t = 1/(1 + p x),  |error| <= 1.5e-7  (x >= 0, odd extension)
// and exp(-x^2) = exp(-u^2/2) is shared with the Gaussian pdf of GELU'.  The epilogue of the fc1 GEMM evaluates this
// 77 M times per launch, which is VALU time no MFMA hides (all waves of a workgroup are in the epilogue together), so
// the arithmetic is kept to 1 v_rcp_f32 + 1 v_exp_f32 (quarter rate) + packed-fp32 FMAs on PAIRS of elements:
// v_rcp_f32 (1 ulp) instead of the IEEE division sequence, exp2 with log2(e) folded into the argument scale.
typedef float f32x2 __attribute__((ext_vector_type(2)));
DEVINL void erf_parts2(f32x2 u, f32x2& erf_abs, f32x2& e) {     // erf(|u|/sqrt2), exp(-u^2/2) of two elements
    const f32x2 au = {fabsf(u.x), fabsf(u.y)};
    const f32x2 d = au * 0.23164189f + 1.0f;                    // 1 + p |u| / sqrt2,  p = 0.3275911
    const f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const f32x2 z = (u * -0.72134752f) * u;                     // -u^2/2 * log2(e)
    e = f32x2{__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)};
    f32x2 pl = t * 1.061405429f + -1.453152027f;
    pl = pl * t + 1.421413741f;
    pl = pl * t + -0.284496736f;
    pl = pl * t + 0.254829592f;
    erf_abs = 1.0f - (pl * t) * e;
}
DEVINL f32x2 cdf2(f32x2 u, f32x2 erf_abs) {                     // 0.5 (1 + sign(u) erf(|u|/sqrt2))
    const f32x2 s = {copysignf(0.5f, u.x), copysignf(0.5f, u.y)};
    return erf_abs * s + 0.5f;
}
DEVINL void gelu_erf2(f32x2 u, f32x2& y) {
    f32x2 ea, e;
    erf_parts2(u, ea, e);
    y = u * cdf2(u, ea);
}
DEVINL void dgelu_erf2(f32x2 u, f32x2& dy) {
    f32x2 ea, e;
    erf_parts2(u, ea, e);
    dy = (u * 0.3989422804014327f) * e + cdf2(u, ea);
}
DEVINL void gelu_and_grad2(f32x2 u, f32x2& y, f32x2& dy) {      // both from one erf/exp evaluation
    f32x2 ea, e;
    erf_parts2(u, ea, e);
    const f32x2 cdf = cdf2(u, ea);
    y = u * cdf;
    dy = (u * 0.3989422804014327f) * e + cdf;
}
// in-place forms over an even-length register array
template <int N> DEVINL void gelu_erf_n(float (&y)[N]) {
#pragma unroll
    for (int i = 0; i < N; i += 2) {
        f32x2 o;
        gelu_erf2(f32x2{y[i], y[i + 1]}, o);
        y[i] = o.x, y[i + 1] = o.y;
    }
}
template <int N> DEVINL void gelu_and_grad_n(float (&y)[N], float (&d)[N]) {
#pragma unroll
    for (int i = 0; i < N; i += 2) {
        f32x2 o, g;
        gelu_and_grad2(f32x2{y[i], y[i + 1]}, o, g);
        y[i] = o.x, y[i + 1] = o.y, d[i] = g.x, d[i + 1] = g.y;
    }
}
// GELU'(u) in one byte (SAIS_EPI_BIAS_GELU_GRADQ_BF16 / SAIS_EPI_MULQ_BF16): q = clamp(rint(26 + 203 d), 0, 255), d = (q - 26) / 203
DEVINL unsigned gq8_code(float d) {
    return (unsigned)__builtin_amdgcn_fmed3f(__builtin_rintf(__builtin_fmaf(d, 203.f, 26.f)), 0.f, 255.f);
}
DEVINL unsigned gq8_pack4(float a, float b, float c, float d) {
    return gq8_code(a) | (gq8_code(b) << 8) | (gq8_code(c) << 16) | (gq8_code(d) << 24);
}
DEVINL float gq8_decode(unsigned word, int byte) {
    return ((float)((word >> (8 * byte)) & 0xffu) - 26.f) * (1.0f / 203.f);
}
DEVINL float dgelu_erf(float u) {
    f32x2 g;
    dgelu_erf2(f32x2{u, u}, g);
    return g.x;
}

// ---- shared by the GEMM kernels (gemm.hip, gemm_row.hip) --------------------------------------------------------
// LDS tile image: 128-B rows (64 bf16), 16-B chunk index XOR (row & 7) -> conflict-free ds_read_b128 fragment reads.
DEVINL int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// weight-row permutation inside a 64-row wave panel: n_local = 16a + 4t + b  ->  LDS row 16t + 4a + b, so that with
// the operands swapped in the MFMA a lane ends up with 16 CONTIGUOUS output columns of one output row
DEVINL int perm_row(int n) { return (n & 64) | ((n & 0x0c) << 2) | ((n & 0x30) >> 2) | (n & 3); }

// XCD-aware tile order (cdna_hip_programming.md T1, bijective form): workgroups are dealt round-robin over the
// 8 XCDs, so hand each XCD a CONTIGUOUS run of logical tiles.
DEVINL int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// one LDS-DMA wave-instruction: 64 lanes x 16 B from per-lane global addresses to 1 KiB of LDS at a wave-uniform base
DEVINL void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// sum over the 16 lanes of a DPP row (lanes 16g .. 16g+15); every lane of the row gets the total
template <int CTRL>
DEVINL float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
DEVINL float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);       // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);       // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);      // row_half_mirror
    v += dpp_mov<0x140>(v);      // row_mirror
    return v;
}

// hipGetLastError is sticky across the whole process (torch included): clear it on entry so that
// sais_check_launch() reports only this call's own launch status.
#define SAIS_ENTER() (void)hipGetLastError()

extern "C" void sais_set_last_error(int hip_error);      // misc.hip

static inline int sais_check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) sais_set_last_error((int)e);
    return e == hipSuccess ? SAIS_OK : SAIS_ERR_LAUNCH;
}

// SAIS_CLK_STAMP (diagnostic builds only, tools/clk_probe.py; MI355X_MICROARCH.md "DVFS give-back" item 6): workgroup 0 of a
// stamped kernel records the shader-clock and the 100-MHz real-time ticks of its own lifetime, so that
// in-kernel clock = d(memtime) / d(memrealtime) x 100 MHz.  The values go to a table no kernel reads.
#ifdef SAIS_CLK_STAMP
static __device__ unsigned long long g_sais_clk[16][2];
struct ClkStamp {
    int id; unsigned long long t0, r0;
    __device__ explicit ClkStamp(int i) : id(i), t0(__builtin_amdgcn_s_memtime()), r0(__builtin_amdgcn_s_memrealtime()) {}
    __device__ ~ClkStamp() {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            g_sais_clk[id][0] = __builtin_amdgcn_s_memtime() - t0;
            g_sais_clk[id][1] = __builtin_amdgcn_s_memrealtime() - r0;
        }
    }
};
#define CLK_STAMP(id) ClkStamp clk_stamp_(id)
#define CLK_EXPORT(name) extern "C" int sais_debug_clk_##name(unsigned long long* host_out) { \
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_sais_clk), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -2; }
#else
#define CLK_STAMP(id) do { } while (0)
#define CLK_EXPORT(name)
#endif
