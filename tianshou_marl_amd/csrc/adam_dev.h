// adam_dev.h -- device pieces of the slab reduction + Adam step shared by csrc/adam.hip (the optimizer kernels) and
// csrc/p2p.hip (the data-parallel form that sums the replicas' gradients between the two inside ONE launch): the same
// instructions in both, so the fused form is bit-identical to reduce_slabs -> all-reduce -> adam_step.
#pragma once
#include "common.h"

namespace {

constexpr int kCols = 64;  // parameters per workgroup
#ifndef TSM_ADAM_DEEP
#define TSM_ADAM_DEEP 0
#endif
constexpr bool tsm_adam_deep = TSM_ADAM_DEEP;

__device__ __forceinline__ double ipow(double b, int64_t e) {
    double r = 1.0;
    while (e > 0) {
        if (e & 1) r *= b;
        b *= b;
        e >>= 1;
    }
    return r;
}

// One lane's share of a parameter's slab sum: slabs sl, sl + 4, sl + 8, ... added sequentially in that order (16 independent
// loads in flight; 32 measured no faster on the 54 MB of the C3 step -- the headline's 11 MB are latency-bound: TSM_ADAM_DEEP).  `stride` = floats between two consecutive slabs.
__device__ __forceinline__ float slab_lane_sum(const float *__restrict__ slabs, int32_t n_slab, int64_t i, int64_t stride, int sl) {
    float acc = 0.f;
    int s = sl;
    if (tsm_adam_deep) {   // (TSM_ADAM_DEEP, set by csrc/adam.hip: 64 loads in flight -- the same additions in the same order)
#pragma unroll 1
        for (; s + 252 < n_slab; s += 256) {
            float t[64];
#pragma unroll
            for (int u = 0; u < 64; ++u) t[u] = slabs[(int64_t)(s + 4 * u) * stride + i];
#pragma unroll
            for (int u = 0; u < 64; ++u) acc += t[u];
        }
    }
#pragma unroll 1
    for (; s + 60 < n_slab; s += 64) {
        float t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = slabs[(int64_t)(s + 4 * u) * stride + i];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += t[u];
    }
    for (; s + 28 < n_slab; s += 32) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = slabs[(int64_t)(s + 4 * u) * stride + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += t[u];
    }
    for (; s < n_slab; s += 4) acc += slabs[(int64_t)s * stride + i];
    return acc;
}

// `n` = parameters covered (bounds).  256 threads = 64 parameters x 4 slab lanes; the lanes' sums meet in LDS as
// ((l0 + l1) + l2) + l3 -- ONE summation order for every consumer of gradient slabs (adam.hip, p2p.hip, the side reductions of
// critic_dw1.hip), so that where a sum is formed does not change its bits.
__device__ __forceinline__ float slab_sum_block(const float *__restrict__ slabs, int32_t n_slab, int64_t n,
                                                int64_t i, float *sm /* [4][64] */, int64_t stride) {
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const float acc = i < n ? slab_lane_sum(slabs, n_slab, i, stride, sl) : 0.f;
    sm[sl * 64 + lane] = acc;
    __syncthreads();
    return sm[lane] + sm[64 + lane] + sm[128 + lane] + sm[192 + lane];
}


// Adam on one parameter (torch single-tensor Adam, amsgrad off; bias corrections from the host or device-resident step count
// in f64 like torch's python scalars, beta^step by repeated squaring; the learning rate may live in HBM: LR schedulers,
// algorithm_base.py:626-627).  Returns the new parameter value (already stored).
__device__ __forceinline__ float adam_apply(float *__restrict__ p, float *__restrict__ m, float *__restrict__ v, int64_t i, float g,
                                            double lr_host, const double *__restrict__ lr_dev, double beta1d, double beta2d,
                                            int64_t step_host, const int64_t *__restrict__ step_dev, float eps, float weight_decay) {
    const int64_t step = step_dev ? *step_dev : step_host;
    const double lr = lr_dev ? *lr_dev : lr_host;
    const float step_size = (float)(lr / (1.0 - ipow(beta1d, step)));
    const float bc2_sqrt = (float)sqrt(1.0 - ipow(beta2d, step));
    const float beta1 = (float)beta1d, beta2 = (float)beta2d;
    const float pi = p[i];
    if (weight_decay != 0.f) g += weight_decay * pi;
    const float mi = beta1 * m[i] + (1.f - beta1) * g;
    const float vi = beta2 * v[i] + (1.f - beta2) * g * g;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    const float pn = pi - step_size * (mi / denom);
    p[i] = pn;
    return pn;
}

}  // namespace
