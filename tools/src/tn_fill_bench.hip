// Fill-stream microbenchmark for the dW (TN) kernel on gfx950: the global -> LDS traffic of gemm_tn_pp_kernel for the four
// weight-gradient GEMMs of a ViT block at M = 50 432, with nothing else in the loop (no fragment reads, no MFMAs).
// Question: is the stream bound by latency (bytes in flight per CU) or by a bandwidth (HBM / L2 / LDS write path)?
//   reg<D>  : global_load_dwordx4 -> VGPR -> ds_write_b128, D tiles of 64 KiB in flight per workgroup
//   dma<D>  : global_load_lds_dwordx4 (LDS-DMA), D tiles in flight (counted vmcnt)
//   nolds   : loads only (kept live), D tiles in flight
// hipcc --offload-arch=gfx950 -O3 tools/src/tn_fill_bench.hip -o tools/bin/tn_fill_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

struct Item { const u16* P; const u16* Q; int ldp, ldq, N1, N2; };
struct Group { Item item[4]; int tile_end[4]; int ntiles, rows_per_split, M, tile_n1; const u16* flat; };

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

template <int V>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(V) : "memory"); }

// MODE 0 reg staging, 1 LDS-DMA, 2 loads only
template <int D, int MODE>
__global__ __launch_bounds__(512) void fill_kernel(Group gp, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];           // 2 x 64 KiB
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int split = wg / gp.ntiles;
    int t = wg - split * gp.ntiles, it0 = 0;
    while (it0 + 1 < 4 && t >= gp.tile_end[it0]) ++it0;
    if (it0 > 0) t -= gp.tile_end[it0 - 1];
    const Item& p = gp.item[it0];
    const int nt2 = p.N2 / 384;
    const int n1_0 = (t / nt2) * 128, n2_0 = (t % nt2) * 384;
    const int mbeg = split * gp.rows_per_split;
    const int mend = min(gp.M, mbeg + gp.rows_per_split);
    if (mbeg >= mend) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = wid >> 1;
    const u16* src0 = blk == 0 ? p.P + n1_0 : p.Q + n2_0 + (blk - 1) * 128;
    const int ld = blk == 0 ? p.ldp : p.ldq;
    const u16* pbase = src0 + (size_t)(mbeg + 32 * (wid & 1) + (lane >> 4)) * ld + (lane & 15) * 8;
    const int nsteps = (mend - mbeg) / 64;
    u32x4 stg[D][8];
    unsigned acc = 0;
    auto src = [&](int step, int j) {
        step = step < nsteps ? step : nsteps - 1;
        if (MODE == 3) step &= 3;                                       // L2-resident: every step re-reads the first 256 rows
        if (MODE == 4)                                                  // linear: 64 KiB contiguous per workgroup and step
            return gp.flat + (size_t)(blockIdx.x * nsteps + step) * 32768 + (wid * 8 + j) * 512 + lane * 8;
        return pbase + (size_t)(step * 64 + 4 * j) * ld;
    };
    auto lds_at = [&](int stage, int j) {
        const int r = 4 * (8 * (wid & 1) + j) + (lane >> 4), c = lane & 15;
        return smem + stage * 65536 + blk * 16384 + r * 256 + ((((c >> 1) ^ (r & 7)) << 5) | ((c & 1) << 4));
    };
    if constexpr (MODE == 1) {
        // LDS-DMA: a piece = 4 rows x 256 B = 1 KiB, lane-linear in LDS
        auto issue = [&](int step) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                char* dst = smem + (step & 1) * 65536 + blk * 16384 + (8 * (wid & 1) + j) * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src(step, j),
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        };
#pragma unroll
        for (int d = 0; d < D; ++d) issue(d);
        for (int st = 0; st < nsteps; ++st) {
            wait_vm<8 * (D - 1)>();
            __builtin_amdgcn_s_barrier();
            issue(st + D);
        }
        wait_vm<0>();
    } else {
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int j = 0; j < 8; ++j) stg[d][j] = *(const u32x4*)src(d, j);
        for (int st = 0; st < nsteps; st += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (st + d < nsteps) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (MODE == 0 || MODE >= 3) *(u32x4*)lds_at((st + d) & 1, j) = stg[d][j];
                        else acc += stg[d][j].x ^ stg[d][j].w;
                        stg[d][j] = *(const u32x4*)src(st + d + D, j);
                    }
                    if (MODE == 0 || MODE >= 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
            }
        }
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += stg[d][j].y;
    }
    if (MODE != 2) acc += *(unsigned*)(smem + tid * 4);
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int D, int MODE>
float run(const Group& g, int grid, unsigned* sink) {
    hipFuncSetAttribute((const void*)fill_kernel<D, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    fill_kernel<D, MODE><<<grid, 512, 131072>>>(g, sink);
    hipDeviceSynchronize();
    float tot = 0;
    const int n = 10;
    hipEventRecord(a);
    for (int it = 0; it < n; ++it) fill_kernel<D, MODE><<<grid, 512, 131072>>>(g, sink);
    hipEventRecord(b);
    hipEventSynchronize(b); hipEventElapsedTime(&tot, a, b);
    return tot / n;
}

int main() {
    const int M = 50432, D = 384, H = 1536;
    const int shp[4][2] = {{D, H}, {H, D}, {D, D}, {3 * D, D}};
    Group g; int wt = 0; double uniq = 0, fill = 0;
    for (int i = 0; i < 4; ++i) {
        u16 *P, *Q;
        hipMalloc(&P, (size_t)M * shp[i][0] * 2); hipMalloc(&Q, (size_t)M * shp[i][1] * 2);
        hipMemset(P, 0x3c, (size_t)M * shp[i][0] * 2); hipMemset(Q, 0x3c, (size_t)M * shp[i][1] * 2);
        g.item[i] = Item{P, Q, shp[i][0], shp[i][1], shp[i][0], shp[i][1]};
        wt += (shp[i][0] / 128) * (shp[i][1] / 384);
        g.tile_end[i] = wt;
        uniq += (double)M * (shp[i][0] + shp[i][1]) * 2;
        fill += (double)M * (shp[i][0] / 128) * (shp[i][1] / 384) * 512 * 2;
    }
    g.ntiles = wt; g.M = M;
    int wns = 256 / wt;
    int wrows = ((M + wns - 1) / wns + 63) / 64 * 64;
    wns = (M + wrows - 1) / wrows;
    g.rows_per_split = wrows;
    unsigned* sink; hipMalloc(&sink, 64);
    { u16* f; hipMalloc(&f, (size_t)256 * 120 * 65536); hipMemset(f, 0x3c, (size_t)256 * 120 * 65536); g.flat = f; }
    printf("tiles %d splits %d grid %d  unique %.0f MB  fill %.0f MB\n", wt, wns, wt * wns, uniq / 1e6, fill / 1e6);
#define R(D_, MODE_, name) { float ms = run<D_, MODE_>(g, wt * wns, sink); \
    printf("%-8s D=%d: %7.1f us  fill %5.2f TB/s  unique %5.2f TB/s\n", name, D_, ms * 1e3, fill / ms / 1e9, uniq / ms / 1e9); }
    R(1, 0, "reg") R(2, 0, "reg") R(3, 0, "reg") R(4, 0, "reg") R(6, 0, "reg")
    R(1, 2, "nolds") R(2, 2, "nolds") R(3, 2, "nolds") R(4, 2, "nolds") R(6, 2, "nolds")
    R(2, 3, "l2res") R(2, 4, "linear")
    R(1, 1, "dma") R(2, 1, "dma") R(3, 1, "dma") R(4, 1, "dma")
    return 0;
}
