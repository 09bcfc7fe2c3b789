// REJECTED VARIANT, kept as a record (not built into the library): the ViT attention backward with TWO key tiles per wave.
// 512 threads, 225-236 VGPRs, no spills; passed every attention test.  Measured on MI355X, stand-alone at 256 frames: 129-132 us,
// the same as the 16-wave attn_bwd_kernel of sais_amd/csrc/attn_vit.hip (131-132 us), both in the form below (all S / dP products
// of a step first, then the exponentials, then dV / dK) and with the two 16-query halves processed one after the other; inside the
// training step 13.73-13.76 vs 13.78 ms.  Halving the operand re-reads from LDS and doubling the independent work per wave at half
// the waves per SIMD changes nothing: see the stamp timeline in LABNOTES.md 4.3.  To try it again: paste this block before `set_lds`
// in attn_vit.hip and launch it with 512 threads and bwd_lds<Geo<197>>() bytes of dynamic LDS (same arguments as attn_bwd_kernel).
// ------------------------------------------------------------------------------------------ backward, two key tiles per wave
// Same LDS images and arithmetic as attn_bwd_kernel with HALF the waves: 512 threads, waves 0-6 own key tiles 2w and 2w + 1
// (tile 13 is a phantom: keys >= NTOK, its P and dS never leave the wave except as the zero rows 208..223 of the dS image),
// every wave does one dQ product.  At two waves per SIMD a wave may use 256 VGPRs, so both tiles' chains are in flight
// together (8 independent S / dP accumulators, 16 for dV / dK) and the Q / dO row fragments and the transposed dO^T / Q^T
// fragments are read from LDS ONCE for the two tiles: the 1.46 MB of operand re-reads per problem halve.
template <class G>
__global__ __launch_bounds__(512) void attn_bwd2_kernel(const bf16* qkv, long ldq, const bf16* dout, long ldo,
                                                        const bf16* out, long ldout, const float* lse, int nprob,
                                                        bf16* dqkv, long lddq, float scale) {
    constexpr int NTOK = G::NTOK, NKS = G::NKS, TILE_ROWS = G::TILE_ROWS, MAT_BYTES = G::MAT_BYTES;
    constexpr int S_BYTES = TILE_ROWS * SROW;
    static_assert(G::NKT <= 14 && TILE_ROWS == 224, "seven waves x two key tiles");
    extern __shared__ __attribute__((aligned(16))) char smem[];
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
        const bool owner = wid < 7;
        bf16x8 fk[2][2], fv[2][2];
        {   // staging: 8 threads per row, 64 rows per pass, every load in flight before the first use
            constexpr int NPASS = 4;
            const int cch = tid & 7, r0 = tid >> 3;
            u32x4 vq[NPASS], vk[NPASS], vd[NPASS], vo[NPASS];
            float lv[NPASS];
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                const int r = r0 + 64 * i, rc = r < NTOK ? r : NTOK - 1;
                vq[i] = *(const u32x4*)(base + (size_t)rc * ldq + cch * 8);
                vk[i] = *(const u32x4*)(base + DM + (size_t)rc * ldq + cch * 8);
                vd[i] = *(const u32x4*)(dob + (size_t)rc * ldo + cch * 8);
                vo[i] = *(const u32x4*)(ob + (size_t)rc * ldout + cch * 8);
                lv[i] = lse[((size_t)f * NH + h) * NTOK + rc];
            }
            if (owner) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    load_q_frags<G>(base + DM, ldq, (2 * wid + t) * 16 + li, g, fk[t]);
                    load_q_frags<G>(base + 2 * DM, ldq, (2 * wid + t) * 16 + li, g, fv[t]);
                }
            }
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                const int r = r0 + 64 * i;
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
                        sL[r] = ok ? lv[i] * LOG2E : INFINITY;
                    }
                }
            }
        }
        f32x4 dk[2][4], dv[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) { dk[t][dt] = f32x4{0, 0, 0, 0}; dv[t][dt] = f32x4{0, 0, 0, 0}; }
        __syncthreads();
#pragma unroll 1
        for (int qs = 0; qs < NKS; ++qs) {
            char* const sb = sS + (qs & 1) * S_BYTES;
            if (owner) {
                f32x4 p[2][2], ds[2][2];                    // [tile][16-query half]
                // phase 1: every S / dP product of the step (16 MFMAs on 8 independent accumulators, 8 ds_read_b128 up front)
                f32x4 a[2][2], b[2][2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int t = 0; t < 2; ++t) { a[t][u] = f32x4{0, 0, 0, 0}; b[t][u] = f32x4{0, 0, 0, 0}; }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 qf[2], of[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        qf[u] = row_frag(sQ, 32 * qs + 16 * u + li, 4 * ks + g);
                        of[u] = row_frag(sO, 32 * qs + 16 * u + li, 4 * ks + g);
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            a[t][u] = mfma16(qf[u], fk[t][ks], a[t][u]);                 // S[q][key]
                            b[t][u] = mfma16(of[u], fv[t][ks], b[t][u]);                 // dP[q][key]
                        }
                }
                // phase 2: P, dS (lane holds q = 32 qs + 16 u + 4 g + r, key = 16 (2 wid + t) + li)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int qrow = 32 * qs + 16 * u;
                    const f32x4 l4 = *(const f32x4*)(sL + qrow + 4 * g);
                    const f32x4 d4 = *(const f32x4*)(sD + qrow + 4 * g);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int key = (2 * wid + t) * 16 + li;
                        bf16x4 dsb;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pv = fast_exp2(__builtin_fmaf(a[t][u][r], c, -l4[r]));
                            p[t][u][r] = pv;
                            const float tt = pv * (b[t][u][r] - d4[r]);                  // x scale at the dK / dQ stores
                            ds[t][u][r] = tt;
                            dsb[r] = (bf16)(key < NTOK ? tt : 0.f);                      // pad keys must not reach dQ
                        }
                        *(bf16x4*)(sb + key * SROW + (16 * u + 4 * g) * 2) = dsb;
                    }
                }
                bf16x8 pf[2], dsf[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) { pf[t] = pack_p(p[t][0], p[t][1]); dsf[t] = pack_p(ds[t][0], ds[t][1]); }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8 to = tr_frag(sO, qs, dt, g, li), tq = tr_frag(sQ, qs, dt, g, li);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        dv[t][dt] = mfma16(to, pf[t], dv[t][dt]);                        // dV^T[d][key]
                        dk[t][dt] = mfma16(tq, dsf[t], dk[t][dt]);                       // dK^T[d][key]
                    }
                }
            }
            __syncthreads();                                // dS of this query step is complete
            {                                               // dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]: one product per wave
                const int qt = wid >> 2, dt = wid & 3;
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
        }
        if (owner) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int key = (2 * wid + t) * 16 + li;
                if (key < NTOK) {
                    bf16* krow = dqkv + ((size_t)f * NTOK + key) * lddq + DM + h * HD + 4 * g;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        bf16x4 a, b;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { a[r] = (bf16)(dk[t][dt][r] * scale); b[r] = (bf16)dv[t][dt][r]; }
                        *(bf16x4*)(krow + 16 * dt) = a;
                        *(bf16x4*)(krow + DM + 16 * dt) = b;
                    }
                }
            }
        }
        __syncthreads();                                    // every read of this problem's images is done
    }
}

