// Store-pattern microbenchmark for the GEMM epilogue on gfx950: how fast can 256-thread workgroups write a
// [M, N] bf16 / fp32 matrix when every wave-instruction writes 16 rows x 4 pieces of 16 B, for several
// piece placements.  hipcc --offload-arch=gfx950 -O3 tools/src/store_bench.hip -o tools/bin/store_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// tile 128 rows x 256 bytes (128 bf16 columns).  wave (wr, wc) owns rows wr*64.., byte columns wc*128..+128.
// MODE 0: lane (li, g) writes bytes [32 g, 32 g + 32) of its row as two 16-B stores (today's bf16 epilogue)
// MODE 1: instruction i writes bytes [64 i + 16 g, +16)  -> 64 contiguous bytes per row per instruction
// MODE 2: rows remapped so that one instruction writes 8 rows x 128 B (whole wave half-row): lane -> row li>>1,
//         byte (li&1)*64 + 16 g   [needs a different accumulator layout; upper bound for row-major tiles]
// MODE 3: fully linear 1 KiB per instruction (tile stored as a contiguous 32 KiB block: upper bound)
template <int MODE>
__global__ __launch_bounds__(256) void st_kernel(char* out, int M, int ldo_bytes, int ntn, int reps) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1, g = lane >> 4, li = lane & 15;
    const int tile = blockIdx.x;
    const int n0b = (tile % ntn) * 256, m0 = (tile / ntn) * 128;
    u32x4 v = {(unsigned)tid, (unsigned)tile, 3u, 4u};
    for (int rep = 0; rep < reps; ++rep) {
        char* base = out + (size_t)rep * M * ldo_bytes;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            if (MODE == 0) {
                char* p = base + (size_t)(m0 + wr * 64 + mt * 16 + li) * ldo_bytes + n0b + wc * 128 + 32 * g;
                *(u32x4*)p = v; *(u32x4*)(p + 16) = v;
            } else if (MODE == 1) {
                char* p = base + (size_t)(m0 + wr * 64 + mt * 16 + li) * ldo_bytes + n0b + wc * 128 + 16 * g;
                *(u32x4*)p = v; *(u32x4*)(p + 64) = v;
            } else if (MODE == 2) {
                char* p = base + (size_t)(m0 + wr * 64 + mt * 16 + (li >> 1)) * ldo_bytes + n0b + wc * 128 + (li & 1) * 64 + 16 * g;
                *(u32x4*)p = v; *(u32x4*)(p + (size_t)8 * ldo_bytes) = v;
            } else {
                char* p = base + ((size_t)tile * 32768) + wid * 8192 + mt * 2048 + lane * 16;
                *(u32x4*)p = v; *(u32x4*)(p + 1024) = v;
            }
        }
    }
}

template <int MODE>
float run(char* buf, int M, int N, int reps) {
    int ntn = N * 2 / 256, ntm = M / 128;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    st_kernel<MODE><<<ntn * ntm, 256>>>(buf, M, N * 2, ntn, reps);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(a); st_kernel<MODE><<<ntn * ntm, 256>>>(buf, M, N * 2, ntn, reps); hipEventRecord(b);
        hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
    }
    return best;
}

int main() {
    const int M = 50432 / 128 * 128, N = 1536, reps = 2;
    char* buf; hipMalloc(&buf, (size_t)M * N * 2 * reps);
    float t[4] = {run<0>(buf, M, N, reps), run<1>(buf, M, N, reps), run<2>(buf, M, N, reps), run<3>(buf, M, N, reps)};
    double bytes = (double)M * N * 2 * reps;
    for (int i = 0; i < 4; ++i) printf("mode %d: %7.1f us  %6.2f TB/s\n", i, t[i] * 1e3, bytes / t[i] / 1e9);
    return 0;
}
