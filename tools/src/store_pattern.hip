// Micro-benchmark: how fast can 2 x [50432, 1536] bf16 (the outputs of the fc1 + GELU / GELU' GEMM) be WRITTEN with the
// store pattern of gemm_nt_w8p_kernel's epilogue (per wave-instruction: 16 rows x 64 B) compared with full-line
// patterns?  No arithmetic, no loads: the answer bounds what an epilogue restructure could gain.
//   hipcc --offload-arch=gfx950 -O3 tools/src/store_pattern.hip -o tools/bin/store_pattern && tools/bin/store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int M = 50432, N = 1536;

// (a) the w8p epilogue: 128 x 128 tiles, 8 waves = 2 (rows of 64) x 4 (cols of 32); lane: row li (16), 8 columns at 8 g
__global__ __launch_bounds__(512, 2) void pat_w8p(unsigned short* o1, unsigned short* o2, int ntiles, int nout) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 2, wc = wid & 3, g = lane >> 4, li = lane & 15;
    const int ntn = N / 128;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int n0 = (t % ntn) * 128, m0 = (t / ntn) * 128;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = m0 + wr * 64 + mt * 16 + li, n = n0 + wc * 32 + 8 * g;
            const u32x4 v = {(unsigned)m, (unsigned)n, 1u, 2u};
            *(u32x4*)(o1 + (size_t)m * N + n) = v;
            if (nout > 1) *(u32x4*)(o2 + (size_t)m * N + n) = v;
        }
    }
}
// (b) same tiles, but a wave-instruction covers 4 rows x 256 B (lane: row lane >> 4, 16 B at 16 (lane & 15))
__global__ __launch_bounds__(512, 2) void pat_rows(unsigned short* o1, unsigned short* o2, int ntiles, int nout) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ntn = N / 128;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int n0 = (t % ntn) * 128, m0 = (t / ntn) * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wid * 16 + i * 4 + (lane >> 4), n = n0 + 8 * (lane & 15);
            const u32x4 v = {(unsigned)m, (unsigned)n, 1u, 2u};
            *(u32x4*)(o1 + (size_t)m * N + n) = v;
            if (nout > 1) *(u32x4*)(o2 + (size_t)m * N + n) = v;
        }
    }
}
// (c) linear fill of the same bytes
__global__ __launch_bounds__(256) void pat_linear(u32x4* o, size_t n16) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) o[i] = u32x4{1, 2, 3, 4};
}

int main() {
    unsigned short *a, *b;
    const size_t bytes = (size_t)M * N * 2;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int ntiles = (M / 128) * (N / 128);
    auto timeit = [&](const char* name, auto fn, double total) {
        for (int i = 0; i < 3; ++i) fn();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) fn();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %7.1f us  %6.2f TB/s\n", name, ms * 1e3 / 20, total / (ms * 1e-3 / 20) / 1e12);
    };
    for (int nout = 1; nout <= 2; ++nout) {
        printf("outputs: %d\n", nout);
        timeit("w8p epilogue pattern", [&] { hipLaunchKernelGGL(pat_w8p, dim3(512), dim3(512), 0, 0, a, b, ntiles, nout); }, (double)bytes * nout);
        timeit("4 rows x 256 B pattern", [&] { hipLaunchKernelGGL(pat_rows, dim3(512), dim3(512), 0, 0, a, b, ntiles, nout); }, (double)bytes * nout);
    }
    timeit("linear fill (1 output)", [&] { hipLaunchKernelGGL(pat_linear, dim3(2048), dim3(256), 0, 0, (u32x4*)a, bytes / 16); }, (double)bytes);
    return 0;
}
