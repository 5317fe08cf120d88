// categorical.hip -- Categorical(logits) head: sample / mode / log_prob / entropy.
//
// Replaces torch.distributions.Categorical as used by the reference's actor policy
// (/root/reference/tianshou/utils/net/discrete.py:22-24; algorithm/modelfree/reinforce.py:183-189;
//  ppo.py:160,187,210):  logits_n = logits - logsumexp(logits);  log_prob = logits_n[act];
//  entropy = -sum(softmax * logits_n);  sample ~ softmax;  mode = argmax.
// Sampling: inverse-CDF with one Philox4x32-10 uniform per row, counter = (offset + row), key = seed.
// One thread per row; A is small (5 for simple_spread), rows are A*4 bytes apart.
#include "common.h"
#include "philox.h"

namespace {

constexpr int kMaxA = 64;

__global__ void sample_kernel(const float *__restrict__ logits, int64_t B, int32_t A, uint64_t seed,
                              uint64_t offset, const uint64_t *__restrict__ offset_dev, int deterministic,
                              int32_t *__restrict__ act_out, float *__restrict__ logp_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    if (offset_dev) offset += *offset_dev;
    const float *lg = logits + i * A;
    float m = -INFINITY;
    int arg = 0;
    for (int j = 0; j < A; ++j) { const float v = lg[j]; if (v > m) { m = v; arg = j; } }
    float s = 0.f;
    for (int j = 0; j < A; ++j) s += expf(lg[j] - m);
    const float lse = m + logf(s);
    int a = arg;
    if (!deterministic) {
        const float u = tsm_philox_uniform(seed, offset + (uint64_t)i) * s;  // u in [0, s)
        float c = 0.f;
        a = A - 1;
        for (int j = 0; j < A; ++j) {
            c += expf(lg[j] - m);
            if (u < c) { a = j; break; }
        }
    }
    act_out[i] = a;
    if (logp_out) logp_out[i] = lg[a] - lse;
}

__global__ void logp_ent_kernel(const float *__restrict__ logits, const int32_t *__restrict__ act, int64_t B,
                                int32_t A, float *__restrict__ logp_out, float *__restrict__ ent_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const float *lg = logits + i * A;
    float m = -INFINITY;
    for (int j = 0; j < A; ++j) m = fmaxf(m, lg[j]);
    float s = 0.f;
    for (int j = 0; j < A; ++j) s += expf(lg[j] - m);
    const float lse = m + logf(s);
    if (logp_out) logp_out[i] = lg[act[i]] - lse;
    if (ent_out) {
        float h = 0.f;
        for (int j = 0; j < A; ++j) { const float l = lg[j] - lse; h -= expf(l) * l; }
        ent_out[i] = h;
    }
}

}  // namespace

TSM_EXPORT int tsm_categorical_sample(const float *logits, int64_t B, int32_t A, uint64_t seed, uint64_t offset,
                                      const uint64_t *offset_dev, int deterministic, int32_t *act_out, float *logp_out,
                                      void *stream) {
    TSM_REQUIRE(B >= 0 && A >= 1 && A <= kMaxA, "tsm_categorical_sample: bad sizes B=%lld A=%d", (long long)B, A);
    if (B == 0) return TSM_OK;
    TSM_REQUIRE(logits && act_out, "tsm_categorical_sample: null pointer");
    hipLaunchKernelGGL(sample_kernel, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0, tsm_stream(stream), logits, B,
                       A, seed, offset, offset_dev, deterministic, act_out, logp_out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int tsm_categorical_logp_entropy(const float *logits, const int32_t *act, int64_t B, int32_t A,
                                            float *logp_out, float *ent_out, void *stream) {
    TSM_REQUIRE(B >= 0 && A >= 1 && A <= kMaxA, "tsm_categorical_logp_entropy: bad sizes");
    if (B == 0) return TSM_OK;
    TSM_REQUIRE(logits && (act || !logp_out), "tsm_categorical_logp_entropy: null pointer");
    hipLaunchKernelGGL(logp_ent_kernel, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0, tsm_stream(stream), logits,
                       act, B, A, logp_out, ent_out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}
