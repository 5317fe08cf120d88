// adam.hip -- gradient-slab reduction + clip_grad_norm_ + Adam on one flat f32 parameter vector.
//
// Replaces Algorithm.Optimizer.step (/root/reference/tianshou/algorithm/algorithm_base.py:485-498):
// zero_grad -> backward -> nn.utils.clip_grad_norm_(params, max_grad_norm) -> torch.optim.Adam.step
// (AdamOptimizerFactory, tianshou/algorithm/optim.py:91-111; torch single-tensor Adam, amsgrad off):
//   g   = sum_slab grad_slab                      (deterministic slab order)
//   g  *= min(1, max_norm / (||g||_2 + 1e-6))     (only when max_grad_norm > 0)
//   g  += weight_decay * p
//   m   = b1 m + (1-b1) g ;  v = b2 v + (1-b2) g^2
//   p  -= lr / (1-b1^t) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
// norm_scratch: f32[kNormBlocks] per-block partial sums of squares (fixed order -> deterministic).
#include "common.h"

namespace {

constexpr int kNormBlocks = 64;

__global__ __launch_bounds__(256) void gradnorm_kernel(const float *__restrict__ slabs, int32_t n_slab, int64_t n,
                                                       float *__restrict__ norm_scratch) {
    __shared__ double sm[256 / 64];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)kNormBlocks * 256) {
        float g = 0.f;
        for (int s = 0; s < n_slab; ++s) g += slabs[(int64_t)s * n + i];
        acc += (double)g * (double)g;
    }
    acc = block_sum<double, 256>(acc, sm);
    if (threadIdx.x == 0) norm_scratch[blockIdx.x] = (float)acc;
}

__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ slabs,
                                                   int32_t n_slab, int64_t n, float *__restrict__ m,
                                                   float *__restrict__ v, float step_size, float beta1,
                                                   float beta2, float bc2_sqrt, float eps, float weight_decay,
                                                   float max_norm, const float *__restrict__ norm_scratch) {
    float coef = 1.f;
    if (max_norm > 0.f) {
        float tot = 0.f;
        for (int b = 0; b < kNormBlocks; ++b) tot += norm_scratch[b];
        const float c = max_norm / (sqrtf(tot) + 1e-6f);
        coef = c < 1.f ? c : 1.f;
    }
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float g = 0.f;
    for (int s = 0; s < n_slab; ++s) g += slabs[(int64_t)s * n + i];
    g *= coef;
    const float pi = p[i];
    if (weight_decay != 0.f) g += weight_decay * pi;
    const float mi = beta1 * m[i] + (1.f - beta1) * g;
    const float vi = beta2 * v[i] + (1.f - beta2) * g * g;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float *__restrict__ slabs, int32_t n_slab,
                                                           int64_t n, float scale, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float g = 0.f;
    for (int s = 0; s < n_slab; ++s) g += slabs[(int64_t)s * n + i];
    out[i] = g * scale;
}

}  // namespace

// Sum per-workgroup gradient slabs into one flat gradient (slab order => deterministic), times `scale`.
// Used in front of the RCCL all-reduce of the data-parallel path (one flat buffer per gradient step).
TSM_EXPORT int tsm_reduce_slabs(const float *grad_slabs, int32_t n_slab, int64_t n, double scale, float *out,
                                void *stream) {
    TSM_REQUIRE(n >= 0 && n_slab >= 1, "tsm_reduce_slabs: bad sizes");
    if (n == 0) return TSM_OK;
    TSM_REQUIRE(grad_slabs && out, "tsm_reduce_slabs: null pointer");
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, tsm_stream(stream),
                       grad_slabs, n_slab, n, (float)scale, out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int tsm_adam_step(float *param, const float *grad_slabs, int32_t n_slab, int64_t n, float *exp_avg,
                             float *exp_avg_sq, int64_t step, double lr, double beta1, double beta2, double eps,
                             double weight_decay, double max_grad_norm, float *norm_scratch, void *stream) {
    TSM_REQUIRE(n >= 0 && n_slab >= 1 && step >= 1, "tsm_adam_step: bad sizes n=%lld n_slab=%d step=%lld",
                (long long)n, n_slab, (long long)step);
    if (n == 0) return TSM_OK;
    TSM_REQUIRE(param && grad_slabs && exp_avg && exp_avg_sq, "tsm_adam_step: null pointer");
    TSM_REQUIRE(max_grad_norm <= 0.0 || norm_scratch, "tsm_adam_step: clipping needs norm_scratch[64]");
    hipStream_t st = tsm_stream(stream);
    if (max_grad_norm > 0.0) {
        hipLaunchKernelGGL(gradnorm_kernel, dim3(kNormBlocks), dim3(256), 0, st, grad_slabs, n_slab, n,
                           norm_scratch);
        TSM_LAUNCH_CHECK();
    }
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, param, grad_slabs, n_slab,
                       n, exp_avg, exp_avg_sq, (float)(lr / bc1), (float)beta1, (float)beta2, (float)sqrt(bc2),
                       (float)eps, (float)weight_decay, (float)max_grad_norm, norm_scratch);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}
