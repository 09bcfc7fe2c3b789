// REJECTED VARIANT, kept as a record (not built into the library): the ViT attention backward with STREAMED queries.
// Measured on MI355X, stand-alone at 256 frames: 136-137 us vs 131-132 us for attn_bwd_kernel of sais_amd/csrc/attn_vit.hip;
// inside the training step 13.75-13.81 vs 13.75 ms.  It passed every attention test.  What it shows: the up-front fill of
// the staged-image kernel (49 us when timed alone, tools/gpu_attn_abl.sh) is NOT additive in the full kernel — taking 100
// of its 150 KB off the critical path and under the query loop changes nothing; the per-step chain of dependent phases is
// what paces the kernel (LABNOTES.md 4.3).  To try it again: paste this block before `set_lds` in attn_vit.hip and launch it
// with 1024 threads and bwd_stream_lds<Geo<197>>() bytes of dynamic LDS (same arguments as attn_bwd_kernel).
// ------------------------------------------------------------------------------------------ backward, streamed queries
// Same arithmetic and wave roles as attn_bwd_kernel, but Q, dO and O are never staged as whole images: the 32 query rows of
// a step arrive as three 4-KiB chunks by LDS-DMA (global_load_lds, unpadded 128-B rows, 16-B units XOR-swizzled by row on
// the SOURCE address) into a three-slot ring, issued TWO steps ahead by wave 13 — one of the three waves that own no key
// tile — which also turns each chunk's dO and O into delta = rowsum(dO * O) once it has landed.  The problem's up-front
// fill shrinks from 150 KB through registers (Q, dO, O, K images + fragments: the 49-us phase of the timing ablations
// during which no MFMA runs) to the K image, the K / V fragments and the log-sum-exp row; the rest of the fetch runs
// under the query loop.  LDS: K image 35 KiB + ring 36 KiB + dS 42 KiB + statistics = 114 KiB.
constexpr int CH_BYTES = 32 * 128;                         // one chunk: 32 query rows x 64 bf16
constexpr int RING = 3;
template <class G> constexpr int bwd_stream_lds() {
    return G::MAT_BYTES + RING * 3 * CH_BYTES + 2 * G::TILE_ROWS * SROW + G::TILE_ROWS * 4 + RING * 32 * 4;
}
DEVINL bf16x8 crow_frag(const char* buf, int row, int chunk) { return *(const bf16x8*)(buf + swz(row, chunk)); }
// transposed fragment of a chunk for the 16-wide column tile ct: rows 4 g + (li >> 2) and + 16, columns 16 ct + 4 (li & 3)
DEVINL bf16x8 ctr_frag(const char* buf, int ct, int g, int li) {
    const int r = 4 * g + (li >> 2), col = 16 * ct + 4 * (li & 3);
    const char* p0 = buf + swz(r, col >> 3) + (col & 7) * 2;
    const char* p1 = buf + swz(r + 16, col >> 3) + (col & 7) * 2;
    return cat4(lds_read_tr16(p0), lds_read_tr16(p1));
}

template <class G>
__global__ __launch_bounds__(1024) void attn_bwd_stream_kernel(const bf16* qkv, long ldq, const bf16* dout, long ldo,
                                                               const bf16* out, long ldout, const float* lse, int nprob,
                                                               bf16* dqkv, long lddq, float scale) {
    constexpr int NTOK = G::NTOK, NKT = G::NKT, NKS = G::NKS, TILE_ROWS = G::TILE_ROWS, MAT_BYTES = G::MAT_BYTES;
    constexpr int S_BYTES = TILE_ROWS * SROW;
    static_assert(NKT <= 13 && G::BWD_THREADS == 1024, "wave 13 is the producer");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sK = smem;
    char* const sC = smem + MAT_BYTES;                     // ring: slot s = {Q, dO, O} chunks
    char* const sS = sC + RING * 3 * CH_BYTES;             // 2 x [224 keys][32 q] bf16
    float* const sL = (float*)(sS + 2 * S_BYTES);          // lse * log2e   [224]
    float* const sD = sL + TILE_ROWS;                      // delta         [RING][32]
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float c = scale * LOG2E;
    // lane offsets into a chunk (the slot base is wave-uniform): row fragments of k-half ks (+ 2048 for the second 16 rows),
    // transposed fragments of column tile dt (+ 2048 for rows + 16)
    const int cr0 = swz(li, g), cr1 = swz(li, 4 + g);
    int ctf[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const int r = 4 * g + (li >> 2), col = 16 * dt + 4 * (li & 3);
        ctf[dt] = swz(r, col >> 3) + (col & 7) * 2;
    }
    for (int prob = blockIdx.x; prob < nprob; prob += gridDim.x) {
        const int f = prob / NH, h = prob - f * NH;
        const bf16* base = qkv + (size_t)f * NTOK * ldq + h * HD;
        const bf16* dob = dout + (size_t)f * NTOK * ldo + h * HD;
        const bf16* ob = out + (size_t)f * NTOK * ldout + h * HD;
        // producer (wave 13): chunk qs -> ring slot qs % 3: 12 LDS-DMA instructions of 8 rows x 128 B
        auto issue_chunk = [&](int qs) {
            char* slot = sC + (qs % RING) * 3 * CH_BYTES;
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {                   // not unrolled: the kernel sits at the 128-VGPR cap
                const int row = 8 * j + (lane >> 3), pos = lane & 7;
                int q = 32 * qs + row;
                q = q < NTOK ? q : NTOK - 1;               // pad queries: finite stand-in rows, their P and dS are exactly 0
                const int src = (pos ^ (row & 7)) * 8;
                glds16(base + (size_t)q * ldq + src, slot + j * 1024);
                glds16(dob + (size_t)q * ldo + src, slot + CH_BYTES + j * 1024);
                glds16(ob + (size_t)q * ldout + src, slot + 2 * CH_BYTES + j * 1024);
            }
        };
        // delta of a landed chunk: lane -> (row = lane >> 1, 32-column half = lane & 1)
        auto chunk_delta = [&](int qs) {
            const char* slot = sC + (qs % RING) * 3 * CH_BYTES;
            const int row = lane >> 1, hf = lane & 1;
            float dl = 0.f;
#pragma unroll 1
            for (int ch = 0; ch < 4; ++ch) {
                const bf16x8 a = crow_frag(slot + CH_BYTES, row, 4 * hf + ch), b = crow_frag(slot + 2 * CH_BYTES, row, 4 * hf + ch);
#pragma unroll
                for (int e = 0; e < 8; ++e) dl = __builtin_fmaf((float)a[e], (float)b[e], dl);
            }
            dl += __shfl_xor(dl, 1);
            if (hf == 0) sD[(qs % RING) * 32 + row] = dl;
        };
        // ---- up-front fill: K image, this wave's K / V fragments, log-sum-exp row; the producer starts the ring
        const int kt = wid;
        const int key = kt * 16 + li;
        bf16x8 fk[2], fv[2];
        if (wid == 13) { issue_chunk(0); issue_chunk(1); }
        {
            const int cch = tid & 7, r0 = tid >> 3;
            u32x4 vk[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = r0 + 128 * i, rc = r < NTOK ? r : NTOK - 1;
                vk[i] = *(const u32x4*)(base + DM + (size_t)rc * ldq + cch * 8);
            }
            float lv = 0.f;
            if (tid < TILE_ROWS) lv = lse[((size_t)f * NH + h) * NTOK + (tid < NTOK ? tid : NTOK - 1)];
            if (kt < NKT) {
                load_q_frags<G>(base + DM, ldq, key, g, fk);
                load_q_frags<G>(base + 2 * DM, ldq, key, g, fv);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = r0 + 128 * i;
                if (r < TILE_ROWS) *(u32x4*)(sK + r * ROWB + cch * 16) = r < NTOK ? vk[i] : u32x4{0, 0, 0, 0};
            }
            if (tid < TILE_ROWS) sL[tid] = tid < NTOK ? lv * LOG2E : INFINITY;       // exp2(-inf) = 0: pad queries
        }
        if (wid == 13) {
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");          // chunk 0 has landed (vmcnt is in-order)
            chunk_delta(0);
        }
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0, 0, 0, 0}; dv[dt] = f32x4{0, 0, 0, 0}; }
        __syncthreads();
#pragma unroll 1
        for (int qs = 0; qs < NKS; ++qs) {
            char* const sb = sS + (qs & 1) * S_BYTES;
            const char* const cq = sC + (qs % RING) * 3 * CH_BYTES;
            const char* const co = cq + CH_BYTES;
            const float* const dD = sD + (qs % RING) * 32;
            if (wid == 13) {                                // slot (qs + 2) % 3 was last read in step qs - 1: free since its barrier
                if (qs + 2 < NKS) issue_chunk(qs + 2);
                if (qs + 1 < NKS) {
                    if (qs + 2 < NKS) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    chunk_delta(qs + 1);
                }
            }
            if (kt < NKT) {
                f32x4 p[2], ds[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int qrow = 16 * u;                // chunk-local; lane holds q = 32 qs + qrow + 4 g + r, key = 16 kt + li
                    f32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const int off = (ks ? cr1 : cr0) + 2048 * u;
                        a = mfma16(*(const bf16x8*)(cq + off), fk[ks], a);               // S[q][key]
                        b = mfma16(*(const bf16x8*)(co + off), fv[ks], b);               // dP[q][key]
                    }
                    const f32x4 l4 = *(const f32x4*)(sL + 32 * qs + qrow + 4 * g);
                    const f32x4 d4 = *(const f32x4*)(dD + qrow + 4 * g);
                    bf16x4 dsb;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = fast_exp2(__builtin_fmaf(a[r], c, -l4[r]));
                        p[u][r] = pv;
                        const float t = pv * (b[r] - d4[r]);             // x scale at the dK / dQ stores
                        ds[u][r] = t;
                        dsb[r] = (bf16)(key < NTOK ? t : 0.f);           // pad keys must not reach dQ
                    }
                    *(bf16x4*)(sb + key * SROW + (16 * u + 4 * g) * 2) = dsb;
                }
                const bf16x8 pf = pack_p(p[0], p[1]), dsf = pack_p(ds[0], ds[1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dv[dt] = mfma16(cat4(lds_read_tr16(co + ctf[dt]), lds_read_tr16(co + ctf[dt] + 2048)), pf, dv[dt]);    // dV^T[d][key]
                    dk[dt] = mfma16(cat4(lds_read_tr16(cq + ctf[dt]), lds_read_tr16(cq + ctf[dt] + 2048)), dsf, dk[dt]);   // dK^T[d][key]
                }
            } else if constexpr (NKT & 1) {                 // the last 16 rows of the dS image belong to no key tile
                if (qs < 2) {
                    for (int i = lane + 64 * (wid - NKT); i < 16 * SROW / 8; i += 64 * (16 - NKT))
                        *(u32x2*)(sS + qs * S_BYTES + NKT * 16 * SROW + i * 8) = u32x2{0, 0};
                }
            }
            __syncthreads();                                // dS of this query step is complete; chunk qs + 1 and its delta too
            if (wid >= 8) {                                 // dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]  (uniform branch)
                const int w = wid - 8, qt = w >> 2, dt = w & 3;
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
    }
}

