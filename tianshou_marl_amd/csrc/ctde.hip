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

// ---- CTDEPolicy.learn loss head (ctde.py:149-185) -----------------------------------------------------------
// Given the centralized critic's outputs q[B][n_out] on global_obs, q_next on global_obs_next and the actor's
// logits[B][A]:
//   values      = q.mean(1),  values_next = q_next.mean(1)              (:154-157; n_out == 1: the value itself)
//   td_target   = rew + gamma * values_next * (1 - terminated)          (:169)
//   critic_loss = mean((values - td_target)^2)                          (:172)
//   advantage   = td_target - values                                    (:181, detached)
//   log_probs   = -cross_entropy(logits, act, reduction="none")         (:184)
//   actor_loss  = -(log_probs[B] * advantage[B,1]).mean()               (:185)
// The product of a (B,) and a (B,1) tensor broadcasts to (B,B), so actor_loss = -mean(log_probs) * mean(advantage)
// (quirk Q7) -- restated exactly, including its gradient
//   d actor_loss / d logits[b][a] = -(mean(advantage) / B) * (1[a == act_b] - softmax(logits_b)[a]).
// Outputs: dq[B][n_out] = 2 (values - td_target) / (B n_out), dlogits[B][A], scalars = {actor_loss, critic_loss}.
namespace {
constexpr int kHeadThreads = 256;

__global__ __launch_bounds__(kHeadThreads) void ctde_head_rows(
    const float *__restrict__ q, const float *__restrict__ q_next, int32_t n_out, const float *__restrict__ rew,
    const uint8_t *__restrict__ terminated, float gamma, const float *__restrict__ logits,
    const int64_t *__restrict__ act, int32_t A, int64_t B, float *__restrict__ dq, double *__restrict__ partial) {
    __shared__ double red[3][kHeadThreads / 64];
    const int64_t b = (int64_t)blockIdx.x * kHeadThreads + threadIdx.x;
    double s_adv = 0.0, s_logp = 0.0, s_sq = 0.0;
    if (b < B) {
        float v = 0.f, vn = 0.f;
        for (int j = 0; j < n_out; ++j) {  // torch mean over dim 1: sequential f32 sum / n
            v += q[b * n_out + j];
            vn += q_next[b * n_out + j];
        }
        v /= (float)n_out;
        vn /= (float)n_out;
        const float td = rew[b] + gamma * vn * (terminated[b] ? 0.f : 1.f);
        const float diff = v - td;
        const float gq = 2.f * diff / ((float)B * (float)n_out);
        for (int j = 0; j < n_out; ++j) dq[b * n_out + j] = gq;
        const float *lg = logits + b * A;
        float mx = lg[0];
        for (int a = 1; a < A; ++a) mx = fmaxf(mx, lg[a]);
        float se = 0.f;
        for (int a = 0; a < A; ++a) se += expf(lg[a] - mx);
        const float logp = lg[act[b]] - mx - logf(se);
        s_adv = (double)(td - v);
        s_logp = (double)logp;
        s_sq = (double)diff * (double)diff;
    }
    s_adv = wave_sum(s_adv);
    s_logp = wave_sum(s_logp);
    s_sq = wave_sum(s_sq);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[0][w] = s_adv; red[1][w] = s_logp; red[2][w] = s_sq; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double s = 0.0;
        for (int i = 0; i < kHeadThreads / 64; ++i) s += red[threadIdx.x][i];
        partial[(int64_t)blockIdx.x * 3 + threadIdx.x] = s;
    }
}

__global__ __launch_bounds__(kHeadThreads) void ctde_head_actor_grad(
    const float *__restrict__ logits, const int64_t *__restrict__ act, int32_t A, int64_t B,
    const double *__restrict__ partial, int32_t n_part, float *__restrict__ dlogits, float *__restrict__ scalars) {
    __shared__ double tot[3];
    __shared__ double red[kHeadThreads / 64];
    // every block folds the partials with the same strided + tree order: identical result everywhere
    for (int q = 0; q < 3; ++q) {
        double s = 0.0;
        for (int i = threadIdx.x; i < n_part; i += kHeadThreads) s += partial[(int64_t)i * 3 + q];
        s = block_sum<double, kHeadThreads>(s, red);
        if (threadIdx.x == 0) tot[q] = s;
        __syncthreads();
    }
    const double mean_adv = tot[0] / (double)B, mean_logp = tot[1] / (double)B;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scalars[0] = (float)(-mean_logp * mean_adv);
        scalars[1] = (float)(tot[2] / (double)B);
    }
    const int64_t b = (int64_t)blockIdx.x * kHeadThreads + threadIdx.x;
    if (b >= B) return;
    const float coef = -(float)mean_adv / (float)B;
    const float *lg = logits + b * A;
    float mx = lg[0];
    for (int a = 1; a < A; ++a) mx = fmaxf(mx, lg[a]);
    float se = 0.f;
    for (int a = 0; a < A; ++a) se += expf(lg[a] - mx);
    const float inv = 1.f / se;
    const int64_t ab = act[b];
    for (int a = 0; a < A; ++a) dlogits[b * A + a] = coef * ((a == ab ? 1.f : 0.f) - expf(lg[a] - mx) * inv);
}
}  // namespace

TSM_EXPORT int64_t tsm_ctde_head_partial_elems(int64_t B) { return B < 0 ? -1 : 3 * ceil_div(B > 0 ? B : 1, kHeadThreads); }

TSM_EXPORT int tsm_ctde_td_head(const float *q, const float *q_next, int32_t n_out, const float *rew,
                                const uint8_t *terminated, float gamma, const float *logits, const int64_t *act,
                                int32_t n_act, int64_t B, float *dq, float *dlogits, double *partial, float *scalars,
                                void *stream) {
    TSM_REQUIRE(B >= 1 && n_out >= 1 && n_act >= 1, "tsm_ctde_td_head: bad sizes");
    TSM_REQUIRE(q && q_next && rew && terminated && logits && act && dq && dlogits && partial && scalars,
                "tsm_ctde_td_head: null pointer");
    const unsigned nb = (unsigned)ceil_div(B, kHeadThreads);
    hipLaunchKernelGGL(ctde_head_rows, dim3(nb), dim3(kHeadThreads), 0, tsm_stream(stream), q, q_next, n_out, rew,
                       terminated, gamma, logits, act, n_act, B, dq, partial);
    TSM_LAUNCH_CHECK();
    hipLaunchKernelGGL(ctde_head_actor_grad, dim3(nb), dim3(kHeadThreads), 0, tsm_stream(stream), logits, act, n_act,
                       B, partial, (int32_t)nb, dlogits, scalars);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}


// ---- the two scalars of CTDEPolicy.learn + mean(advantage) from the one-launch kernels' partial sums -----------------
namespace {
__global__ __launch_bounds__(64) void ctde_finalize_kernel(const double *__restrict__ pc, int32_t nb_c,
                                                           const double *__restrict__ pa, int32_t nb_a, int64_t B,
                                                           float *__restrict__ scalars, float *__restrict__ mean_adv) {
    // one wave: lane l takes entries l, l + 64, ... (all loads of a lane in flight), then the xor butterfly -- a fixed order
    const int lane = threadIdx.x;
    double s_adv = 0.0, s_sq = 0.0, s_logp = 0.0;
    for (int i = lane; i < nb_c; i += 64) { s_adv += pc[4 * i]; s_sq += pc[4 * i + 1]; }
    for (int i = lane; i < nb_a; i += 64) s_logp += pa[4 * i];
    s_adv = wave_sum(s_adv);
    s_sq = wave_sum(s_sq);
    s_logp = wave_sum(s_logp);
    if (lane != 0) return;
    const double m_adv = s_adv / (double)B, m_logp = s_logp / (double)B;
    scalars[0] = (float)(-m_logp * m_adv);
    scalars[1] = (float)(s_sq / (double)B);
    *mean_adv = (float)m_adv;
}
}  // namespace

TSM_EXPORT int tsm_ctde_finalize(const double *critic_partial, int32_t nb_c, const double *actor_partial, int32_t nb_a,
                                 int64_t B, float *scalars_out, float *mean_adv_out, void *stream) {
    TSM_REQUIRE(nb_c >= 1 && nb_a >= 1 && B >= 1, "tsm_ctde_finalize: bad sizes");
    TSM_REQUIRE(critic_partial && actor_partial && scalars_out && mean_adv_out, "tsm_ctde_finalize: null pointer");
    hipLaunchKernelGGL(ctde_finalize_kernel, dim3(1), dim3(64), 0, tsm_stream(stream), critic_partial, nb_c, actor_partial,
                       nb_a, B, scalars_out, mean_adv_out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}
