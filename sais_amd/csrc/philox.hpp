// Counter-based RNG for train-mode dropout: Philox4x32-10 (Salmon et al., SC'11; the generator behind torch's CUDA
// dropout as well, but the (seed, offset) -> element mapping here is this library's own, so masks are NOT torch's).
// state = {seed, offset} lives in device memory (two uint64) so that a captured hipGraph draws fresh masks at every
// replay: sais_rng_advance bumps `offset` inside the graph.  The mask of element `idx` of dropout site `sid`:
//   r = philox(counter = {idx / 4 (64 bit), sid, offset}, key = seed)[idx % 4],   keep  <=>  r >= p * 2^32.
// Forward and backward regenerate the same mask from (state, sid, idx): nothing is stored.
#pragma once
#include "common.hpp"

DEVINL void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned h0 = (unsigned)(p0 >> 32), l0 = (unsigned)p0, h1 = (unsigned)(p1 >> 32), l1 = (unsigned)p1;
    c[0] = h1 ^ c[1] ^ k0; c[1] = l1; c[2] = h0 ^ c[3] ^ k1; c[3] = l0;
}

DEVINL unsigned philox_u32(const unsigned long long* state, unsigned sid, unsigned long long idx) {
    const unsigned long long seed = state[0], off = state[1], q = idx >> 2;
    unsigned c[4] = {(unsigned)q, (unsigned)(q >> 32), sid, (unsigned)off};
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32) ^ (unsigned)(off >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const unsigned k = (unsigned)idx & 3u;                        // (a dynamic index would put c[] into scratch memory)
    return k == 0 ? c[0] : k == 1 ? c[1] : k == 2 ? c[2] : c[3];
}

// the whole block of four draws that covers elements idx .. idx + 3 (idx % 4 == 0): one tenth of the arithmetic per
// element of four philox_u32 calls
DEVINL void philox_u32x4(const unsigned long long* state, unsigned sid, unsigned long long idx, unsigned (&c)[4]) {
    const unsigned long long seed = state[0], off = state[1], q = idx >> 2;
    c[0] = (unsigned)q; c[1] = (unsigned)(q >> 32); c[2] = sid; c[3] = (unsigned)off;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32) ^ (unsigned)(off >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// p in [0, 1): threshold on the raw 32-bit draw; p = 0 keeps everything
// (4294967040 = 2^32 - 256 is the largest fp32 below 2^32: the clamp keeps the float -> unsigned conversion defined)
DEVINL unsigned drop_threshold(float p) { return (unsigned)fminf(p * 4294967296.0f, 4294967040.0f); }
DEVINL bool philox_keep(const unsigned long long* state, unsigned sid, unsigned long long idx, unsigned thr) {
    return philox_u32(state, sid, idx) >= thr;
}
