// ctde.hip -- CTDE global-state construction from per-agent observation arrays.
//
// Replaces GlobalStateConstructor.build("concatenate" | "mean")
// (/root/reference/tianshou/algorithm/multiagent/ctde.py:291-300).  Agents are taken in the order
// given (env.agents order; reference quirk Q5: never dict-iteration order).
// concat: out[b][a*D + d] = obs_a[b][d]   mean: out[b][d] = (1/N) sum_a obs_a[b][d]
#include "common.h"

namespace {
constexpr int kMaxAgents = 64;
struct PtrTable { const float *p[kMaxAgents]; };

__global__ void concat_kernel(PtrTable t, int32_t N, int64_t B, int32_t D, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over B*N*D, output order
    if (i >= B * N * D) return;
    const int64_t b = i / ((int64_t)N * D);
    const int32_t rem = (int32_t)(i - b * (int64_t)N * D);
    const int32_t a = rem / D, d = rem - a * D;
    out[i] = t.p[a][b * D + d];
}

__global__ void mean_kernel(PtrTable t, int32_t N, int64_t B, int32_t D, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    float s = 0.f;
    for (int a = 0; a < N; ++a) s += t.p[a][i];  // torch.stack(...).mean(0): sequential f32 sum
    out[i] = s / (float)N;
}
}  // namespace

TSM_EXPORT int tsm_global_state(const float *const *obs_by_agent_host, int32_t n_agent, int64_t B, int32_t D,
                                int mode, float *out, void *stream) {
    TSM_REQUIRE(n_agent >= 1 && n_agent <= kMaxAgents && B >= 0 && D >= 1, "tsm_global_state: bad sizes");
    TSM_REQUIRE(mode == 0 || mode == 1, "tsm_global_state: mode must be 0 (concatenate) or 1 (mean)");
    if (B == 0) return TSM_OK;
    TSM_REQUIRE(obs_by_agent_host && out, "tsm_global_state: null pointer");
    PtrTable t;
    for (int a = 0; a < n_agent; ++a) {
        TSM_REQUIRE(obs_by_agent_host[a], "tsm_global_state: null agent array %d", a);
        t.p[a] = obs_by_agent_host[a];
    }
    if (mode == 0)
        hipLaunchKernelGGL(concat_kernel, dim3((unsigned)ceil_div(B * n_agent * D, 256)), dim3(256), 0,
                           tsm_stream(stream), t, n_agent, B, D, out);
    else
        hipLaunchKernelGGL(mean_kernel, dim3((unsigned)ceil_div(B * D, 256)), dim3(256), 0, tsm_stream(stream), t,
                           n_agent, B, D, out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}
