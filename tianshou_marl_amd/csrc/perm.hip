// perm.hip -- minibatch permutations on the device.
//
// Replaces the `np.random.permutation(length)` that `Batch.split(size, shuffle=True)` draws once per repeat
// (/root/reference/tianshou/data/batch.py:1219, used by ppo.py:179) when the caller asks for device-side shuffling
// (PPO(shuffle="device")).  The parity mode (shuffle="numpy") still uploads numpy's permutation.
//
// A permutation of [0, n) is evaluated point-wise, out[i] = pi(i): a balanced 6-round Feistel network on the
// smallest even-width power-of-two domain >= n, keyed by Philox4x32-10(seed, counter + perm index), with cycle walking
// (re-apply while the image is >= n; the domain is < 4n, so < 4 applications on average).  Every thread is independent:
// one launch writes all permutations of an update (n_perm x n), instead of a key-sort per permutation.
// out[p][i] = pi_p(i) * scale + (p / group_size) * offset_mul  -- the affine part maps a row permutation to lane ids
// (lane = row * n_agent + agent) for per-agent dispatch (marl.py:233-246).
#include "common.h"
#include "philox.h"

namespace {

__device__ __forceinline__ uint32_t mix32(uint32_t x) {  // full-avalanche 32-bit finalizer
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

__device__ __forceinline__ void perm_point(int64_t n, int hb, uint64_t seed, uint64_t ctr, int64_t i, int p, int64_t scale,
                                           int32_t group_size, int64_t offset_mul, int64_t *__restrict__ out) {
    uint32_t k[4];
    tsm_philox4(seed, ctr + (uint64_t)p, k);
    const uint32_t mask = (1u << hb) - 1u;
    uint32_t x = (uint32_t)i;
    do {
        uint32_t l = x >> hb, r = x & mask;
#pragma unroll
        for (int round = 0; round < 6; ++round) {
            const uint32_t f = mix32(r + k[round & 3] + 0x9E3779B9u * (uint32_t)(round + 1)) & mask;
            const uint32_t t = l ^ f;
            l = r;
            r = t;
        }
        x = (l << hb) | r;
    } while ((int64_t)x >= n);
    out[(int64_t)p * n + i] = (int64_t)x * scale + (int64_t)(p / group_size) * offset_mul;
}

__global__ __launch_bounds__(256) void perm_kernel(int64_t n, int hb, uint64_t seed, uint64_t counter,
                                                    const uint64_t *__restrict__ counter_dev, int64_t scale,
                                                    int32_t group_size, int64_t offset_mul, int64_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    perm_point(n, hb, seed, counter + (counter_dev ? *counter_dev : 0), i, blockIdx.y, scale, group_size, offset_mul, out);
}

// ... and the draw counter advanced by the launch itself: every thread reads *counter_dev first, a workgroup arrives (one atomic)
// behind its barrier, and the last arrival adds `inc` -- no workgroup can still be reading the old value then
__global__ __launch_bounds__(256) void perm_advance_kernel(int64_t n, int hb, uint64_t seed, uint64_t *counter_dev, uint64_t inc,
                                                            uint32_t *done_ctr, int64_t scale, int32_t group_size,
                                                            int64_t offset_mul, int64_t *__restrict__ out) {
    const uint64_t ctr = __hip_atomic_load(counter_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) perm_point(n, hb, seed, ctr, i, blockIdx.y, scale, group_size, offset_mul, out);
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(done_ctr, 1u) == gridDim.x * gridDim.y - 1) {
        __hip_atomic_store(counter_dev, ctr + inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *done_ctr = 0u;
    }
}

}  // namespace

TSM_EXPORT int tsm_random_permutations(int64_t n, int32_t n_perm, uint64_t seed, uint64_t counter,
                                       const uint64_t *counter_dev, int64_t scale, int32_t group_size,
                                       int64_t offset_mul, int64_t *out, void *stream) {
    TSM_REQUIRE(n >= 0 && n < ((int64_t)1 << 30), "tsm_random_permutations: n=%lld out of range [0, 2^30)", (long long)n);
    TSM_REQUIRE(n_perm >= 0 && n_perm <= 65535, "tsm_random_permutations: n_perm out of range");
    TSM_REQUIRE(group_size >= 1, "tsm_random_permutations: group_size must be >= 1");
    if (n == 0 || n_perm == 0) return TSM_OK;
    TSM_REQUIRE(out, "tsm_random_permutations: null output");
    int bits = 0;
    while (((int64_t)1 << bits) < n) ++bits;
    const int hb = bits < 2 ? 1 : (bits + 1) / 2;
    hipLaunchKernelGGL(perm_kernel, dim3((unsigned)ceil_div(n, 256), (unsigned)n_perm), dim3(256), 0, tsm_stream(stream),
                       n, hb, seed, counter, counter_dev, scale, group_size, offset_mul, out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int tsm_random_permutations_advance(int64_t n, int32_t n_perm, uint64_t seed, uint64_t *counter_dev, uint64_t counter_inc,
                                               uint32_t *done_ctr, int64_t scale, int32_t group_size, int64_t offset_mul,
                                               int64_t *out, void *stream) {
    TSM_REQUIRE(n >= 1 && n < ((int64_t)1 << 30), "tsm_random_permutations_advance: n=%lld out of range [1, 2^30)", (long long)n);
    TSM_REQUIRE(n_perm >= 1 && n_perm <= 65535, "tsm_random_permutations_advance: n_perm out of range");
    TSM_REQUIRE(group_size >= 1, "tsm_random_permutations_advance: group_size must be >= 1");
    TSM_REQUIRE(out && counter_dev && done_ctr, "tsm_random_permutations_advance: null pointer");
    int bits = 0;
    while (((int64_t)1 << bits) < n) ++bits;
    const int hb = bits < 2 ? 1 : (bits + 1) / 2;
    hipLaunchKernelGGL(perm_advance_kernel, dim3((unsigned)ceil_div(n, 256), (unsigned)n_perm), dim3(256), 0, tsm_stream(stream),
                       n, hb, seed, counter_dev, counter_inc, done_ctr, scale, group_size, offset_mul, out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}
