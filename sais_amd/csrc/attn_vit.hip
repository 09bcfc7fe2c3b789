// Spatial self-attention of the DINO ViT (197 tokens, head_dim 64) for gfx950: forward and backward.
//   Attention.forward — dino-main/vision_transformer.py:80-92:
//       attn = softmax(q k^T * 64^-0.5);  x = attn v   (per frame, per head; no mask, dropout p = 0)
// One 256-thread workgroup per (frame, head): the whole K and V of a head (197 x 64 bf16 = 25 KiB
// each) sit in LDS, so the 197x197 score matrix never touches HBM and softmax is exact (no online
// rescale): every 16-query tile holds its full 16 x 208 score strip in 52 accumulator VGPRs.
//
// LDS image: 160-B rows (64 bf16 + 16 pad).  That stride is conflict-free BOTH for ds_read_b128 row
// fragments (K as MFMA operand over d) and for ds_read_b64_tr_b16 transposed fragments (V^T / K^T /
// Q^T / dO^T as MFMA operand over tokens), so one image serves both uses.
//
// Orientation: scores are computed transposed, S^T = K Q^T (key on accumulator rows, query on the
// lane), so that P^T is already the B operand of O^T = V^T P^T with no lane movement; the k-slot
// map of that product is  element e of lane group g  <->  key 32 s + 16 (e>>2) + 4 g + (e&3).
#include "common.hpp"
#include "../../include/sais_hip.h"

namespace {
constexpr int HD = 64, NH = 6, DM = 384;
constexpr int ROWB = 160;                 // LDS row stride in bytes
constexpr float LOG2E = 1.4426950408889634f;

// Token-count geometry.  197 = 224 x 224 frames (the SAIS extraction / training path and DINO's global crops);
// 37 = DINO's 96 x 96 local crops (main_dino.py:658-663 -> prepare_tokens, vision_transformer.py:196-207).
template <int NTOK_>
struct Geo {
    static constexpr int NTOK = NTOK_;
    static constexpr int NKT = (NTOK_ + 15) / 16;          // 16-key tiles            (197: 13, 37: 3)
    static constexpr int NKS = (NKT + 1) / 2;              // 32-key k-steps          (197: 7,  37: 2)
    static constexpr int TILE_ROWS = 32 * NKS;             //                         (197: 224, 37: 64)
    static constexpr int MAT_BYTES = TILE_ROWS * ROWB;     //                         (197: 35840)
    // backward workgroup: 16 waves for 197 tokens (13 own a key tile, 8 of them also do the dQ products); 4 waves for
    // 37 tokens (3 key tiles, the 8 dQ products shared two per wave) so that 3 workgroups fit a CU: these problems are
    // a few microseconds of latency each, and independent problems in flight are what hides it
    static constexpr int BWD_THREADS = NKT > 4 ? 1024 : 256;
};

// stage rows [0,197) x 64 bf16 of a [M, ld] matrix (column offset applied by caller) into LDS, zero-fill pad rows
// Both global loads of every row segment are issued before the first LDS write, with clamped (not branched-on) row
// addresses: a workgroup then has its whole K / V fill in flight at once instead of one dependent round trip per
// conditional load (the compiler serialised the `r < NTOK ? load : 0` form into ~10 of them).
template <class G>
DEVINL void stage_two(char* lds0, const bf16* src0, char* lds1, const bf16* src1, long ld, int tid) {
    const int c = tid & 7, r0 = tid >> 3;
    u32x4 v0[G::NKS], v1[G::NKS];
#pragma unroll
    for (int i = 0; i < G::NKS; ++i) {
        const int r = r0 + 32 * i, rc = r < G::NTOK ? r : G::NTOK - 1;
        v0[i] = *(const u32x4*)(src0 + (size_t)rc * ld + c * 8);
        v1[i] = *(const u32x4*)(src1 + (size_t)rc * ld + c * 8);
    }
#pragma unroll
    for (int i = 0; i < G::NKS; ++i) {
        const int r = r0 + 32 * i;
        const u32x4 z = {0, 0, 0, 0};
        *(u32x4*)(lds0 + r * ROWB + c * 16) = r < G::NTOK ? v0[i] : z;
        *(u32x4*)(lds1 + r * ROWB + c * 16) = r < G::NTOK ? v1[i] : z;
    }
}
DEVINL bf16x8 row_frag(const char* lds, int row, int chunk) { return *(const bf16x8*)(lds + row * ROWB + chunk * 16); }

// transposed fragment for k-step s (32 tokens) and 16-wide column tile ct
DEVINL bf16x8 tr_frag(const char* lds, int s, int ct, int g, int li) {
    const char* p = lds + (32 * s + 4 * g + (li >> 2)) * ROWB + (16 * ct + 4 * (li & 3)) * 2;
    return cat4(lds_read_tr16(p), lds_read_tr16(p + 16 * ROWB));
}

// raw v_exp_f32 (exp2f() adds a denormal-range fix-up of 4 VALU per element; arguments here are <= ~0 and a
// flush to zero of results below 2^-126 is exactly what softmax wants)
DEVINL float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// NKT sixteen-row tiles over 4 waves: wave w takes tiles w, w+4, ... (NKT / 4 each); the NKT % 4 left-over tiles go to
// the waves (w - extra) & 3 = 0, 1, ..: `extra` rotates with the (frame, head) index so no SIMD is systematically the last
// to finish (197 tokens: the 13th tile, 5 real rows)
template <class G> DEVINL int tiles_of_wave(int wid, int extra) { return G::NKT / 4 + ((((wid - extra) & 3) < G::NKT % 4) ? 1 : 0); }
template <class G> DEVINL int tile_id(int wid, int i, int extra) {
    return i < G::NKT / 4 ? wid + 4 * i : 4 * (G::NKT / 4) + ((wid - extra) & 3);
}

DEVINL float group_max(float v) { v = fmaxf(v, __shfl_xor(v, 16)); return fmaxf(v, __shfl_xor(v, 32)); }
DEVINL float group_sum(float v) { v += __shfl_xor(v, 16); return v + __shfl_xor(v, 32); }

DEVINL bf16x8 pack_p(const f32x4& a, const f32x4& b) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = (bf16)a[i]; r[4 + i] = (bf16)b[i]; }
    return r;
}

// S^T strip for one 16-query tile: s[t][r] = score(key 16 t + 4 g + r, query q0 + li), masked to -inf past 197
template <class G>
DEVINL void score_strip(const char* sK, const bf16x8 (&fq)[2], int g, int li, f32x4 (&s)[G::NKT]) {
    constexpr int NKT = G::NKT;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        f32x4 a = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a = mfma16(row_frag(sK, 16 * t + li, 4 * ks + g), fq[ks], a);
        s[t] = a;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (16 * (NKT - 1) + 4 * g + r >= G::NTOK) s[NKT - 1][r] = -INFINITY;
}

template <class G>
DEVINL void load_q_frags(const bf16* base, long ld, int q, int g, bf16x8 (&f)[2]) {
    const bf16* p = base + (size_t)(q < G::NTOK ? q : G::NTOK - 1) * ld + 8 * g;
    f[0] = *(const bf16x8*)p;
    f[1] = *(const bf16x8*)(p + 32);
}

// ------------------------------------------------------------------------------------------ forward
template <class G>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16* qkv, long ldq, bf16* out, long ldo, float* lse,
                                                       float* probs, float scale) {
    constexpr int NTOK = G::NTOK, NKT = G::NKT, NKS = G::NKS, MAT_BYTES = G::MAT_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CLK_STAMP(10);
    char* sK = smem;
    char* sV = smem + MAT_BYTES;
    const int h = blockIdx.x, f = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, li = lane & 15;
    const bf16* base = qkv + (size_t)f * NTOK * ldq + h * HD;
    stage_two<G>(sK, base + DM, sV, base + 2 * DM, ldq, tid);
    __syncthreads();
    const float c = scale * LOG2E;
    const int extra = (f * NH + h) & 3;
    const int nmine = tiles_of_wave<G>(wid, extra);
    bf16x8 fq[2], fq_next[2];
    load_q_frags<G>(base, ldq, tile_id<G>(wid, 0, extra) * 16 + li, g, fq);
    for (int it = 0; it < nmine; ++it) {
        const int qt = tile_id<G>(wid, it, extra);
        const int q = qt * 16 + li;
        if (it + 1 < nmine) load_q_frags<G>(base, ldq, tile_id<G>(wid, it + 1, extra) * 16 + li, g, fq_next);   // prefetch
        f32x4 s[NKT];
        score_strip<G>(sK, fq, g, li, s);
        fq[0] = fq_next[0]; fq[1] = fq_next[1];
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, s[t][r]);
        m = group_max(m);
        const float mc = -m * c;
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { float e = fast_exp2(__builtin_fmaf(s[t][r], c, mc)); s[t][r] = e; sum += e; }
        sum = group_sum(sum);
        const float inv = 1.0f / sum;
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0, 0, 0, 0};
        const f32x4 z4 = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            bf16x8 pf = pack_p(s[2 * ks], (2 * ks + 1 < NKT) ? s[2 * ks + 1] : z4);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = mfma16(tr_frag(sV, ks, dt, g, li), pf, o[dt]);
        }
        if (q < NTOK) {
            bf16* orow = out + ((size_t)f * NTOK + q) * ldo + h * HD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                bf16x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (bf16)(o[dt][r] * inv);
                *(bf16x4*)(orow + 16 * dt) = v;
            }
            if (lse && g == 0) lse[((size_t)f * NH + h) * NTOK + q] = m * scale + __logf(sum);
            if (probs) {
                float* pr = probs + (((size_t)f * NH + h) * NTOK + q) * NTOK;
#pragma unroll
                for (int t = 0; t < NKT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int key = 16 * t + 4 * g + r;
                        if (key < NTOK) pr[key] = s[t][r] * inv;
                    }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ backward, single pass
// One persistent 1024-thread workgroup per CU walks the (frame, head) problems.  Q, dO and K of a head are staged ONCE
// into LDS (160-B rows: row and transposed reads conflict-free); P is rebuilt ONCE per (query, key) from the saved
// log-sum-exp with the KEY on the lane (S = Q K^T, dP = dO V^T: K / V fragments of a wave's key tile live in its
// registers), so that P and dS are already the B operands of dV^T += dO^T P and dK^T += Q^T dS (accumulated in
// registers over the query sweep: no cross-workgroup sums).  Only dS crosses LDS, once, as a bf16 [key][query] image
// (double-buffered, 96-B rows: conflict-free transposed reads): after the barrier eight waves turn it into
// dQ^T = K^T dS^T for the 32 queries of the step — complete over all keys, so dQ goes straight to HBM.
// delta_q = sum_d dO O is computed while staging.  Five MFMA products per (query, key) tile instead of seven, one
// exponential instead of two, every operand staged once (the two-kernel version re-staged K, V, Q, dO: 470 vs 348 MB).
#ifndef SAIS_ATTN_SROW
#define SAIS_ATTN_SROW 96
#endif
constexpr int SROW = SAIS_ATTN_SROW;                       // bytes per key row of the dS image (32 queries bf16 + pad)
template <class G> constexpr int bwd_lds() {               // 197 tokens: 152320
    return 3 * G::MAT_BYTES + 2 * G::TILE_ROWS * 4 + 2 * G::TILE_ROWS * SROW;
}

// SAIS_ATTN_STAMP (debug builds only, tools/gpu_attn_stamp.sh): lane 0 of every wave of workgroup 0 records the shader clock
// at the phase boundaries of its SECOND problem; sais_debug_attn_stamps() copies the table out.
#ifdef SAIS_ATTN_STAMP
__device__ unsigned long long g_attn_stamps[16][40];
#define STAMP(i) do { if (blockIdx.x == 0 && prob == (int)gridDim.x && lane == 0) g_attn_stamps[wid][i] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
// SAIS_ATTN_ABL (timing ablations, results are WRONG when set; tools/gpu_attn_abl.sh): 1 no exponential, 2 no dQ phase,
// 4 no dV / dK products, 8 no S / dP products, 16 no query loop at all (staging + final stores), 32 no dS store
#ifndef SAIS_ATTN_ABL
#define SAIS_ATTN_ABL 0
#endif
#ifndef SAIS_ATTN_NB_ABL
#define SAIS_ATTN_NB_ABL 0          // timing ablations of the barrier-free backward (WRONG results): 1 no job K, 2 no job Q
#endif
#ifndef SAIS_ATTN_BWD_NB_DEFAULT
#define SAIS_ATTN_BWD_NB_DEFAULT false
#endif
template <class G>
__global__ __launch_bounds__(G::BWD_THREADS) void attn_bwd_kernel(const bf16* qkv, long ldq, const bf16* dout, long ldo,
                                                        const bf16* out, long ldout, const float* lse, int nprob,
                                                        bf16* dqkv, long lddq, float scale) {
    constexpr int NTOK = G::NTOK, NKT = G::NKT, NKS = G::NKS, TILE_ROWS = G::TILE_ROWS, MAT_BYTES = G::MAT_BYTES;
    constexpr int S_BYTES = TILE_ROWS * SROW;
    constexpr int NWAVES = G::BWD_THREADS / 64, RPP = G::BWD_THREADS / 8;      // RPP: rows staged per pass
    constexpr int DQ_FIRST = NWAVES >= 16 ? 8 : 0, DQ_WAVES = NWAVES - DQ_FIRST;
    static_assert(NKT <= NWAVES, "one key tile per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CLK_STAMP(11);
    char* const sQ = smem;
    char* const sO = smem + MAT_BYTES;
    char* const sK = smem + 2 * MAT_BYTES;
    float* const sL = (float*)(smem + 3 * MAT_BYTES);      // lse * log2e   [224]
    float* const sD = sL + TILE_ROWS;                      // delta         [224]
    char* const sS = (char*)(sD + TILE_ROWS);              // 2 x [224 keys][32 q] bf16
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float c = scale * LOG2E;
    for (int prob = blockIdx.x; prob < nprob; prob += gridDim.x) {
        const int f = prob / NH, h = prob - f * NH;
        const bf16* base = qkv + (size_t)f * NTOK * ldq + h * HD;
        const bf16* dob = dout + (size_t)f * NTOK * ldo + h * HD;
        const bf16* ob = out + (size_t)f * NTOK * ldout + h * HD;
        STAMP(0);
        // ---- stage Q, dO, K (rows >= NTOK zero) and delta = rowsum(dO * O); 8 threads per row, RPP rows per pass.
        // EVERY global load of the problem — the passes' four row segments each, the log-sum-exp values and this wave's K / V
        // fragments — is issued before the first use: addresses are clamped instead of branched on, so the compiler keeps
        // ~14 loads in flight per lane instead of five dependent round trips (loads, wait, lse, wait, second pass, ...),
        // which were most of the 49 us this phase takes on its own (tools/gpu_attn_abl.sh, ablation 16).
        const int kt = wid;                                // waves >= NKT own no key tile
        const int key = kt * 16 + li;
        bf16x8 fk[2], fv[2];
        {
            constexpr int NPASS = (TILE_ROWS + RPP - 1) / RPP;
            const int cch = tid & 7, r0 = tid >> 3;
            u32x4 vq[NPASS], vk[NPASS], vd[NPASS], vo[NPASS];
            float lv[NPASS];
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                const int r = r0 + RPP * i, rc = r < NTOK ? r : NTOK - 1;
                vq[i] = *(const u32x4*)(base + (size_t)rc * ldq + cch * 8);
                vk[i] = *(const u32x4*)(base + DM + (size_t)rc * ldq + cch * 8);
                vd[i] = *(const u32x4*)(dob + (size_t)rc * ldo + cch * 8);
                vo[i] = *(const u32x4*)(ob + (size_t)rc * ldout + cch * 8);
                lv[i] = lse[((size_t)f * NH + h) * NTOK + rc];
            }
            if (kt < NKT) {
                load_q_frags<G>(base + DM, ldq, key, g, fk);
                load_q_frags<G>(base + 2 * DM, ldq, key, g, fv);
            }
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                const int r = r0 + RPP * i;
                if (r < TILE_ROWS) {
                    const bool ok = r < NTOK;
                    const u32x4 z = {0, 0, 0, 0};
                    *(u32x4*)(sQ + r * ROWB + cch * 16) = ok ? vq[i] : z;
                    *(u32x4*)(sK + r * ROWB + cch * 16) = ok ? vk[i] : z;
                    *(u32x4*)(sO + r * ROWB + cch * 16) = ok ? vd[i] : z;
                    const bf16x8 a = __builtin_bit_cast(bf16x8, vd[i]), b = __builtin_bit_cast(bf16x8, vo[i]);
                    float dl = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) dl = __builtin_fmaf((float)a[e], (float)b[e], dl);
                    dl += __shfl_xor(dl, 1); dl += __shfl_xor(dl, 2); dl += __shfl_xor(dl, 4);
                    if (cch == 0) {
                        sD[r] = ok ? dl : 0.f;
                        sL[r] = ok ? lv[i] * LOG2E : INFINITY;               // exp2(-inf) = 0: pad queries
                    }
                }
            }
        }
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0, 0, 0, 0}; dv[dt] = f32x4{0, 0, 0, 0}; }
        STAMP(1);
        __syncthreads();
        STAMP(2);
#pragma unroll 1
        for (int qs = 0; qs < ((SAIS_ATTN_ABL & 16) ? 0 : NKS); ++qs) {
            char* const sb = sS + (qs & 1) * S_BYTES;
            if (kt < NKT) {
                f32x4 p[2], ds[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int qrow = 32 * qs + 16 * u;      // lane holds q = qrow + 4 g + r, key = 16 kt + li
                    f32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < ((SAIS_ATTN_ABL & 8) ? 0 : 2); ++ks) {
                        a = mfma16(row_frag(sQ, qrow + li, 4 * ks + g), fk[ks], a);     // S[q][key]
                        b = mfma16(row_frag(sO, qrow + li, 4 * ks + g), fv[ks], b);     // dP[q][key]
                    }
                    const f32x4 l4 = *(const f32x4*)(sL + qrow + 4 * g);
                    const f32x4 d4 = *(const f32x4*)(sD + qrow + 4 * g);
                    bf16x4 dsb;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = (SAIS_ATTN_ABL & 1) ? __builtin_fmaf(a[r], c, -l4[r]) : fast_exp2(__builtin_fmaf(a[r], c, -l4[r]));
                        p[u][r] = pv;
                        const float t = pv * (b[r] - d4[r]);             // x scale at the dK / dQ stores
                        ds[u][r] = t;
                        dsb[r] = (bf16)(key < NTOK ? t : 0.f);           // pad keys must not reach dQ
                    }
                    if (!(SAIS_ATTN_ABL & 32)) *(bf16x4*)(sb + key * SROW + (16 * u + 4 * g) * 2) = dsb;
                }
                STAMP(3 + 5 * qs);                          // S / dP, exponentials, dS written
                const bf16x8 pf = pack_p(p[0], p[1]), dsf = pack_p(ds[0], ds[1]);
#pragma unroll
                for (int dt = 0; dt < ((SAIS_ATTN_ABL & 4) ? 0 : 4); ++dt) {
                    dv[dt] = mfma16(tr_frag(sO, qs, dt, g, li), pf, dv[dt]);            // dV^T[d][key]
                    dk[dt] = mfma16(tr_frag(sQ, qs, dt, g, li), dsf, dk[dt]);           // dK^T[d][key]
                }
                if (SAIS_ATTN_ABL & 4) { dv[0] += p[0] + p[1]; dk[0] += ds[0] + ds[1]; }   // keep the producers alive
            } else if constexpr (NKT & 1) {                 // the last 16 rows of the dS image belong to no key tile
                if (qs < 2) {
                    for (int i = lane + 64 * (wid - NKT); i < 16 * SROW / 8; i += 64 * (NWAVES - NKT))
                        *(u32x2*)(sS + qs * S_BYTES + NKT * 16 * SROW + i * 8) = u32x2{0, 0};
                }
            }
            STAMP(4 + 5 * qs);                              // dV / dK products issued
            __syncthreads();                                // dS of this query step is complete
            STAMP(5 + 5 * qs);
            if (wid >= DQ_FIRST && !(SAIS_ATTN_ABL & 2))    // dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]  (uniform branch)
            for (int w = wid - DQ_FIRST; w < 8; w += DQ_WAVES) {
                const int qt = w >> 2, dt = w & 3;
                f32x4 o = {0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const char* ps = sb + (32 * ks + 4 * g + (li >> 2)) * SROW + (16 * qt + 4 * (li & 3)) * 2;
                    const bf16x8 fs = cat4(lds_read_tr16(ps), lds_read_tr16(ps + 16 * SROW));
                    o = mfma16(tr_frag(sK, ks, dt, g, li), fs, o);
                }
                const int q = 32 * qs + 16 * qt + li;       // lane: query q, d = 16 dt + 4 g + r
                if (q < NTOK) {
                    bf16x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (bf16)(o[r] * scale);
                    *(bf16x4*)(dqkv + ((size_t)f * NTOK + q) * lddq + h * HD + 16 * dt + 4 * g) = v;
                }
            }
            STAMP(6 + 5 * qs);                              // dQ product stored (dQ waves)
        }
        STAMP(38);
        if (kt < NKT && key < NTOK) {
            bf16* krow = dqkv + ((size_t)f * NTOK + key) * lddq + DM + h * HD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                bf16x4 a, b;
#pragma unroll
                for (int r = 0; r < 4; ++r) { a[r] = (bf16)(dk[dt][r] * scale); b[r] = (bf16)dv[dt][r]; }
                *(bf16x4*)(krow + 16 * dt) = a;
                *(bf16x4*)(krow + DM + 16 * dt) = b;
            }
        }
        __syncthreads();                                    // every read of this problem's images is done
        STAMP(39);
    }
}

#ifndef SAIS_EXPERIMENTAL
#define SAIS_EXPERIMENTAL 0
#endif
#if SAIS_EXPERIMENTAL
// ------------------------------------------------------------------------------------------ backward without the barrier chain (round 6)
// EXPERIMENT RECORD (built only with -DSAIS_EXPERIMENTAL=1; SAIS_ATTN_BWD_NB=1 selects it): correct at the first run and 5-7 % SLOWER than
// the single-pass kernel (138-140 vs 131-132 us stand-alone; ablations: staging 30 us, job K 59 us, job Q 52 us — LABNOTES R6.5).
// The single-pass kernel above pays one workgroup barrier per 32-query step: the key-tile waves hand dS to the dQ waves through
// LDS, every phase of a step waits for the slowest wave, and MFMA busy is 0.17 (LABNOTES 4.3, R5.7).  Here NO value crosses waves:
// one wave per 16-row tile does two independent jobs with every operand image (Q, dO, K, V: 4 x 35 KB) staged once —
//   job K (its KEY tile):   S = Q K^T and dP = dO V^T with the key on the lane (K / V fragments in registers), P from the saved
//                           log-sum-exp, dV^T += dO^T P, dK^T += Q^T dS over the seven 32-query steps: the loop above minus the dS
//                           store;
//   job Q (its QUERY tile): S^T = K Q^T and dP^T = V dO^T with the query on the lane (Q / dO fragments in registers: the forward
//                           kernel's orientation), P^T and dS^T in registers = the B operand of dQ^T += K^T dS^T over the seven
//                           32-key steps.
// S, dP and the exponential are computed twice (7 product groups instead of 5, 2 x the VALU work) — on pipes that idled — and
// the only barriers are the two around the staging of a problem.  13 waves (832 threads) for 197 tokens, one workgroup per CU.
template <class G> struct NbTag {};
template <class G> constexpr int bwd_nb_lds() { return 4 * G::MAT_BYTES + 2 * G::TILE_ROWS * 4; }

template <class G>
__global__ __launch_bounds__(G::NKT * 64) void attn_bwd_nb_kernel(const bf16* qkv, long ldq, const bf16* dout, long ldo,
                                                                  const bf16* out, long ldout, const float* lse, int nprob,
                                                                  bf16* dqkv, long lddq, float scale) {
    constexpr int NTOK = G::NTOK, NKT = G::NKT, NKS = G::NKS, TILE_ROWS = G::TILE_ROWS, MAT_BYTES = G::MAT_BYTES;
    constexpr int NTHREADS = NKT * 64, RPP = NTHREADS / 8, NPASS = (TILE_ROWS + RPP - 1) / RPP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CLK_STAMP(11);
    char* const sQ = smem;
    char* const sO = smem + MAT_BYTES;                     // dO
    char* const sK = smem + 2 * MAT_BYTES;
    char* const sV = smem + 3 * MAT_BYTES;
    float* const sL = (float*)(smem + 4 * MAT_BYTES);      // lse * log2e   [TILE_ROWS]
    float* const sD = sL + TILE_ROWS;                      // delta         [TILE_ROWS]
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float c = scale * LOG2E;
    const int row16 = wid * 16 + li;                       // this lane's key (job K) and query (job Q)
    for (int prob = blockIdx.x; prob < nprob; prob += gridDim.x) {
        const int f = prob / NH, h = prob - f * NH;
        const bf16* base = qkv + (size_t)f * NTOK * ldq + h * HD;
        const bf16* dob = dout + (size_t)f * NTOK * ldo + h * HD;
        const bf16* ob = out + (size_t)f * NTOK * ldout + h * HD;
        {
            // every global load of the problem is issued before the first use (clamped addresses, no branches)
            const int cch = tid & 7, r0 = tid >> 3;
            u32x4 vq[NPASS], vk[NPASS], vv[NPASS], vd[NPASS], vo[NPASS];
            float lv[NPASS];
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                const int r = r0 + RPP * i, rc = r < NTOK ? r : NTOK - 1;
                vq[i] = *(const u32x4*)(base + (size_t)rc * ldq + cch * 8);
                vk[i] = *(const u32x4*)(base + DM + (size_t)rc * ldq + cch * 8);
                vv[i] = *(const u32x4*)(base + 2 * DM + (size_t)rc * ldq + cch * 8);
                vd[i] = *(const u32x4*)(dob + (size_t)rc * ldo + cch * 8);
                vo[i] = *(const u32x4*)(ob + (size_t)rc * ldout + cch * 8);
                lv[i] = lse[((size_t)f * NH + h) * NTOK + rc];
            }
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                const int r = r0 + RPP * i;
                if (r < TILE_ROWS) {
                    const bool ok = r < NTOK;
                    const u32x4 z = {0, 0, 0, 0};
                    *(u32x4*)(sQ + r * ROWB + cch * 16) = ok ? vq[i] : z;
                    *(u32x4*)(sK + r * ROWB + cch * 16) = ok ? vk[i] : z;
                    *(u32x4*)(sV + r * ROWB + cch * 16) = ok ? vv[i] : z;
                    *(u32x4*)(sO + r * ROWB + cch * 16) = ok ? vd[i] : z;
                    const bf16x8 a = __builtin_bit_cast(bf16x8, vd[i]), b = __builtin_bit_cast(bf16x8, vo[i]);
                    float dl = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) dl = __builtin_fmaf((float)a[e], (float)b[e], dl);
                    dl += __shfl_xor(dl, 1); dl += __shfl_xor(dl, 2); dl += __shfl_xor(dl, 4);
                    if (cch == 0) {
                        sD[r] = ok ? dl : 0.f;
                        sL[r] = ok ? lv[i] * LOG2E : INFINITY;               // exp2(-inf) = 0: pad queries
                    }
                }
            }
        }
        __syncthreads();
        // ---- job K: this wave's key tile over the query steps (its K / V row fragments come from the staged images)
        if (!(SAIS_ATTN_NB_ABL & 1)) {
            bf16x8 fk[2], fv[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) { fk[ks] = row_frag(sK, row16, 4 * ks + g); fv[ks] = row_frag(sV, row16, 4 * ks + g); }
            f32x4 dk[4], dv[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0, 0, 0, 0}; dv[dt] = f32x4{0, 0, 0, 0}; }
#pragma unroll 1
            for (int qs = 0; qs < NKS; ++qs) {
                f32x4 p[2], ds[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int qrow = 32 * qs + 16 * u;      // lane holds q = qrow + 4 g + r, key = row16
                    f32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        a = mfma16(row_frag(sQ, qrow + li, 4 * ks + g), fk[ks], a);     // S[q][key]
                        b = mfma16(row_frag(sO, qrow + li, 4 * ks + g), fv[ks], b);     // dP[q][key]
                    }
                    const f32x4 l4 = *(const f32x4*)(sL + qrow + 4 * g);
                    const f32x4 d4 = *(const f32x4*)(sD + qrow + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = fast_exp2(__builtin_fmaf(a[r], c, -l4[r]));
                        p[u][r] = pv;
                        ds[u][r] = pv * (b[r] - d4[r]);                 // x scale at the dK store
                    }
                }
                const bf16x8 pf = pack_p(p[0], p[1]), dsf = pack_p(ds[0], ds[1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dv[dt] = mfma16(tr_frag(sO, qs, dt, g, li), pf, dv[dt]);            // dV^T[d][key]
                    dk[dt] = mfma16(tr_frag(sQ, qs, dt, g, li), dsf, dk[dt]);           // dK^T[d][key]
                }
            }
            if (row16 < NTOK) {
                bf16* krow = dqkv + ((size_t)f * NTOK + row16) * lddq + DM + h * HD + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    bf16x4 a, b;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { a[r] = (bf16)(dk[dt][r] * scale); b[r] = (bf16)dv[dt][r]; }
                    *(bf16x4*)(krow + 16 * dt) = a;
                    *(bf16x4*)(krow + DM + 16 * dt) = b;
                }
            }
        }
        // ---- job Q: this wave's query tile over the key steps (query on the lane: lse / delta are per-lane scalars)
        if (!(SAIS_ATTN_NB_ABL & 2)) {
            bf16x8 fq[2], fdo[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) { fq[ks] = row_frag(sQ, row16, 4 * ks + g); fdo[ks] = row_frag(sO, row16, 4 * ks + g); }
            f32x4 dq[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0, 0, 0, 0};
            const float lq = sL[row16], dlt = sD[row16];
#pragma unroll 1
            for (int ks2 = 0; ks2 < NKS; ++ks2) {
                f32x4 dst[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int krow = 32 * ks2 + 16 * t;     // lane holds key = krow + 4 g + r, q = row16; rows past NKT * 16 are zero
                    f32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        a = mfma16(row_frag(sK, krow + li, 4 * ks + g), fq[ks], a);     // S^T[key][q]
                        b = mfma16(row_frag(sV, krow + li, 4 * ks + g), fdo[ks], b);    // dP^T[key][q]
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = fast_exp2(__builtin_fmaf(a[r], c, -lq));
                        dst[t][r] = (krow + 4 * g + r < NTOK) ? pv * (b[r] - dlt) : 0.f;        // pad keys must not reach dQ
                    }
                }
                const bf16x8 dsf = pack_p(dst[0], dst[1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dq[dt] = mfma16(tr_frag(sK, ks2, dt, g, li), dsf, dq[dt]);   // dQ^T[d][q]
            }
            if (row16 < NTOK) {                             // lane: query row16, d = 16 dt + 4 g + r
                bf16* qrow = dqkv + ((size_t)f * NTOK + row16) * lddq + h * HD + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    bf16x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (bf16)(dq[dt][r] * scale);
                    *(bf16x4*)(qrow + 16 * dt) = v;
                }
            }
        }
        __syncthreads();                                    // every read of this problem's images is done
    }
}

#endif  // SAIS_EXPERIMENTAL

template <class G> constexpr int fwd_lds() { return 2 * G::MAT_BYTES; }

// raise the dynamic-LDS limit of a kernel once per process and device (not per call: keeps the launch
// path free of runtime-API calls so it can be captured into a hipGraph); the limit only ever grows.
template <class Tag, typename K>          // Tag: one `granted` table per kernel INSTANCE (K alone is only the signature)
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

template <class G>
int launch_fwd(const void* qkv, long ldqkv, int frames, void* out, long ldo, float* lse, float* probs, void* stream) {
    if (set_lds<G>(attn_fwd_kernel<G>, fwd_lds<G>())) return SAIS_ERR_LAUNCH;
    hipLaunchKernelGGL(attn_fwd_kernel<G>, dim3(NH, frames), dim3(256), fwd_lds<G>(), (hipStream_t)stream,
                       (const bf16*)qkv, ldqkv, (bf16*)out, ldo, lse, probs, 0.125f);
    return sais_check_launch();
}

template <class G>
int launch_bwd(const void* qkv, long ldqkv, const void* dout, long lddo, const void* out, long ldout, const float* lse,
               int frames, void* dqkv, long lddqkv, void* stream) {
    const int nprob = frames * NH;
#if SAIS_EXPERIMENTAL
    // SAIS_ATTN_BWD_NB (read once): 1 = the barrier-free form (197 tokens only), 0 = the single-pass kernel with the dS hand-off
    static const bool nb = [] { const char* e = getenv("SAIS_ATTN_BWD_NB"); return e ? atoi(e) != 0 : SAIS_ATTN_BWD_NB_DEFAULT; }();
    if constexpr (G::NKT > 4) {
        if (nb) {
            if (set_lds<NbTag<G>>(attn_bwd_nb_kernel<G>, bwd_nb_lds<G>())) return SAIS_ERR_LAUNCH;
            hipLaunchKernelGGL(attn_bwd_nb_kernel<G>, dim3(nprob < 256 ? nprob : 256), dim3(G::NKT * 64), bwd_nb_lds<G>(),
                               (hipStream_t)stream, (const bf16*)qkv, ldqkv, (const bf16*)dout, lddo, (const bf16*)out, ldout, lse,
                               nprob, (bf16*)dqkv, lddqkv, 0.125f);
            return sais_check_launch();
        }
    }
#endif
    if (set_lds<G>(attn_bwd_kernel<G>, bwd_lds<G>())) return SAIS_ERR_LAUNCH;
    // short sequences leave most of the LDS free: several workgroups per CU
    int per_cu = 160 * 1024 / bwd_lds<G>();                       // resident workgroups per CU (LDS-limited)
    per_cu = G::BWD_THREADS == 1024 ? 1 : (per_cu > 4 ? 4 : per_cu);
    const int cap = 256 * per_cu;
    hipLaunchKernelGGL(attn_bwd_kernel<G>, dim3(nprob < cap ? nprob : cap), dim3(G::BWD_THREADS), bwd_lds<G>(), (hipStream_t)stream,
                       (const bf16*)qkv, ldqkv, (const bf16*)dout, lddo, (const bf16*)out, ldout, lse, nprob,
                       (bf16*)dqkv, lddqkv, 0.125f);
    return sais_check_launch();
}
}  // namespace

#ifdef SAIS_ATTN_STAMP
extern "C" int sais_debug_attn_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_attn_stamps), sizeof(unsigned long long) * 16 * 40) == hipSuccess ? 0 : -2;
}
#endif

extern "C" int sais_vit_attn_fwd(const void* qkv, long ldqkv, int frames, int ntok, void* out, long ldo, float* lse,
                                 float* probs, void* stream) {
    SAIS_ENTER();
    if (!qkv || !out || frames <= 0 || (ldqkv & 7) || (ldo & 3)) return SAIS_ERR_ARG;
    if (ntok == 197) return launch_fwd<Geo<197>>(qkv, ldqkv, frames, out, ldo, lse, probs, stream);
    if (ntok == 37) return launch_fwd<Geo<37>>(qkv, ldqkv, frames, out, ldo, lse, probs, stream);
    return SAIS_ERR_ARG;
}

extern "C" int sais_vit_attn_bwd(const void* qkv, long ldqkv, const void* dout, long lddo, const void* out, long ldout,
                                 const float* lse, float* delta_ws, int frames, int ntok, void* dqkv, long lddqkv,
                                 void* stream) {
    SAIS_ENTER();
    (void)delta_ws;      // the single-pass kernel computes delta = rowsum(dO * O) while staging; kept in the ABI
    if (!qkv || !dout || !out || !lse || !dqkv || frames <= 0 || (ldqkv & 7) || (lddo & 7) ||
        (ldout & 7) || (lddqkv & 3))
        return SAIS_ERR_ARG;
    if (ntok == 197) return launch_bwd<Geo<197>>(qkv, ldqkv, dout, lddo, out, ldout, lse, frames, dqkv, lddqkv, stream);
    if (ntok == 37) return launch_bwd<Geo<37>>(qkv, ldqkv, dout, lddo, out, ldout, lse, frames, dqkv, lddqkv, stream);
    return SAIS_ERR_ARG;
}

CLK_EXPORT(attn)
