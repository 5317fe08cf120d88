// philox.h -- counter-based Philox4x32-10 uniform used by every sampling site (one definition).
#pragma once
#include <stdint.h>

__device__ __forceinline__ void tsm_philox_round(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3,
                                                 uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

// 4 x 32 random bits for (seed, counter)
__device__ __forceinline__ void tsm_philox4(uint64_t seed, uint64_t counter, uint32_t out[4]) {
    uint32_t c0 = (uint32_t)counter, c1 = (uint32_t)(counter >> 32), c2 = 0, c3 = 0;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        tsm_philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float tsm_u01(uint32_t bits) {  // 24-bit uniform in [0, 1)
    return (float)(bits >> 8) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ float tsm_philox_uniform(uint64_t seed, uint64_t counter) {
    uint32_t r[4];
    tsm_philox4(seed, counter, r);
    return tsm_u01(r[0]);
}
