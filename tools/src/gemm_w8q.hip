// NOT PART OF libsais_hip.so — kept as the record of a round-4 experiment (LABNOTES R4.4).  Correct (it passed the GEMM tests of
// tests/test_kernels_gpu.py and the bench line's parity gate when wired into sais_gemm_nt behind SAIS_NT_W8Q=1: dispatch
// `gemm_nt_w8q_kernel<E><<<min(ntiles, 768), 512, 6 * QSLOT>>>(p, ntiles)` for M >= 8192, K % 32 == 0) and SLOWER than the
// two-workgroups-per-CU kernel it was meant to beat: fc1 + GELU + GELU' 162-165 vs 140-142 us, dX fc2 x GELU' 150-151 vs
// 117-118, qkv 71-72 vs 60-62 stand-alone; 13.49-13.55 vs 12.76 ms per step.  It belongs inside gemm.hip's anonymous namespace
// (NtParams, BM, BN, glds16, perm_row32, xcd_remap, mfma16, epilogue_loads8 / epilogue8).
// ---------------------------------------------------------------------------------------------
// Three workgroups per CU (experiment, R4.4: a workgroup's K loop is a latency chain and two chains per CU fill neither the
// LDS / MFMA side nor HBM).  The same 128 x 128 tile, eight waves of 64 x 32, weight-row permutation and epilogue as
// gemm_nt_w8p_kernel, but K advances in steps of 32 through two three-slot rings of 8-KiB stages (A and W both two steps
// ahead): 48 KiB of LDS per workgroup, <= 80 VGPRs (six waves per SIMD).  A stage row is 32 bf16 = 64 B and a fragment
// read takes 16 rows x 64 B = 1 KiB contiguous, so the LDS image is linear (no swizzle): one LDS-DMA piece = 16 rows.
constexpr int QK = 32, QSLOT = BM * QK * 2;                          // 8 KiB
template <int EPI>
__global__ __launch_bounds__(512, 6) void gemm_nt_w8q_kernel(NtParams p, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // A ring: 3 x 8 KiB, then W ring: 3 x 8 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3, g = lane >> 4, li = lane & 15;
    const int ntn = p.N / BN;
    const bf16* asrc; const bf16* bsrc;
    auto set_tile = [&](int v, int& m0, int& n0) {
        const int tile = xcd_remap(v, ntiles);
        n0 = (tile % ntn) * BN; m0 = (tile / ntn) * BM;
        const int r = 16 * wid + (lane >> 2);
        int m = m0 + r;
        m = m < p.M ? m : p.M - 1;                                   // clamp: rows >= M are never stored
        asrc = p.A + (size_t)m * p.lda + (lane & 3) * 8;
        bsrc = p.B + (size_t)(n0 + perm_row32(r)) * p.ldb + (lane & 3) * 8;
    };
    char* const sW = smem + 3 * QSLOT;
    auto issue = [&](int kt, int slot) {                             // one A piece + one W piece per wave
        glds16(asrc + kt * QK, smem + slot * QSLOT + wid * 1024);
        glds16(bsrc + kt * QK, sW + slot * QSLOT + wid * 1024);
    };
    const int nk = p.K / QK;
    constexpr int SROW = (EPI == SAIS_EPI_BIAS_F32) ? 2 : (EPI == SAIS_EPI_BIAS_RESID_F32) ? 2 : (EPI == SAIS_EPI_PATCH_F32) ? 2
                       : (EPI == SAIS_EPI_BIAS_GELU_GRAD_BF16) ? 2 : 1;
    const int nstores = 4 * (SROW + ((EPI == SAIS_EPI_BIAS_RESID_F32 || EPI == SAIS_EPI_BIAS_GELU_BF16) && p.out2 ? 1 : 0));
    int v = blockIdx.x, m0, n0;
    if (v >= ntiles) return;
    set_tile(v, m0, n0);
    int slot = 0;                                                    // ring position of the current step
    auto nxt = [](int s, int d) { s += d; return s >= 3 ? s - 3 : s; };
    issue(0, 0);
    issue(1, 1);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (;;) {
        f32x4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        float bias[8];
        EpiAux8 aux;
        __builtin_amdgcn_s_setprio(2);
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 2 < nk) issue(kt + 2, nxt(slot, 2));
            const char* sa = smem + slot * QSLOT;
            const char* sb = sW + slot * QSLOT;
            if (kt == nk - 1) epilogue_loads8<EPI>(p, m0 + wr * 64, li, n0 + wc * 32 + 8 * g, bias, aux);
            bf16x8 fa[4], fb[2];
#pragma unroll
            for (int t = 0; t < 4; ++t) fa[t] = *(const bf16x8*)(sa + (wr * 64 + t * 16 + li) * 64 + g * 16);
#pragma unroll
            for (int t = 0; t < 2; ++t) fb[t] = *(const bf16x8*)(sb + (wc * 32 + t * 16 + li) * 64 + g * 16);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma16(fb[nt], fa[mt], acc[mt][nt]);
            if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
            else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // last step: only the epilogue's loads are out
            __builtin_amdgcn_s_barrier();
            slot = nxt(slot, 1);
        }
        __builtin_amdgcn_s_setprio(0);
        const int cm0 = m0, cn0 = n0;
        const int nv = v + gridDim.x;
        const bool more = nv < ntiles;
        if (more) {                                                    // the next tile's first two stages go out before the epilogue
            set_tile(nv, m0, n0);
            issue(0, slot);
            issue(1, nxt(slot, 1));
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = cm0 + wr * 64 + mt * 16 + li;
            if (m >= p.M) continue;
            float vv[8];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) vv[4 * nt + r] = acc[mt][nt][r];
            epilogue8<EPI>(p, m, cn0 + wc * 32 + 8 * g, vv, bias, aux, mt);
        }
        if (!more) break;
        v = nv;
        // stage 0 of the next tile must have landed; its stage 1 (two pieces) and this epilogue's stores may stay in flight
        const int allow = (cm0 + BM <= p.M) ? nstores + 2 : 0;
        if (allow == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (allow == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (allow == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

