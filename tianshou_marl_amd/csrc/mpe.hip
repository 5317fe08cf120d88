// mpe.hip -- batched, device-resident PettingZoo-MPE `simple_spread` / `simple_tag` style worlds.
//
// Row (f)1 of SURVEY.md section 8: the reference drives third-party pettingzoo envs one Python object
// at a time through PettingZooEnv / EnhancedPettingZooEnv + DummyVectorEnv
// (/root/reference/tianshou/env/pettingzoo_env.py:71-120, enhanced_pettingzoo_env.py:130-222,
//  venvs.py:237-322); at 1k-8k envs that loop is the bottleneck, so the dynamics are restated here.
// The pettingzoo 1.24.2 MPE sources are NOT part of the reference tree and pettingzoo is not installed:
// the physics below is restated from the published MPE specification and its parity with pettingzoo is
// UNPINNED; tests pin this kernel against oracle/mpe_oracle.py (same spec, numpy f64).
//
// World (dim_p = 2, dt = 0.1, damping = 0.25, contact_force = 100, contact_margin = 1e-3):
//   action u in {noop, -x, +x, -y, +y} * accel;  agent-agent soft contact forces;
//   v <- v*(1-damping) + (f/mass)*dt ; clamp |v| <= max_speed (if > 0) ; p <- p + v*dt
// simple_spread: N agents (size .15, accel 5), N landmarks; reward = (1-local_ratio)*global + local_ratio*local,
//   global = -sum_landmarks min_agents dist, local = -#collisions(agent); obs = [vel, pos, landmarks rel,
//   others rel, comm(0)] (6N); truncation at max_cycles.
// One thread per environment (N <= 8: N^2 pair work in registers); state SoA [n_env][N][2].
#include "common.h"
#include "philox.h"

namespace {

constexpr int kMaxN = 8;

struct MpeCfg {
    int n_env, N, obs_dim, max_cycles;
    float dt, damping, contact_force, contact_margin, agent_size, landmark_size, accel, max_speed, local_ratio;
};

__device__ __forceinline__ float uni_pm1(uint32_t bits) { return tsm_u01(bits) * 2.f - 1.f; }

__device__ void reset_env(const MpeCfg &c, int e, uint64_t seed, uint64_t episode, float *apos, float *avel,
                          float *lpos, int32_t *steps) {
    // positions ~ U(-1, 1)^2, velocities 0; Philox counter = (episode * n_env + e) * 8 + draw
    const uint64_t base = (episode * (uint64_t)c.n_env + (uint64_t)e) * 8ull;
    for (int i = 0; i < c.N; ++i) {
        uint32_t r[4];
        tsm_philox4(seed, base + (uint64_t)i, r);
        float *ap = apos + ((int64_t)e * c.N + i) * 2, *av = avel + ((int64_t)e * c.N + i) * 2;
        float *lp = lpos + ((int64_t)e * c.N + i) * 2;
        ap[0] = uni_pm1(r[0]); ap[1] = uni_pm1(r[1]);
        lp[0] = uni_pm1(r[2]); lp[1] = uni_pm1(r[3]);
        av[0] = 0.f; av[1] = 0.f;
    }
    steps[e] = 0;
}

__device__ void write_obs(const MpeCfg &c, int e, const float *apos, const float *avel, const float *lpos,
                          float *obs_env /* this env's [N][obs_dim] block (global or LDS) */) {
    const float *ap = apos + (int64_t)e * c.N * 2, *av = avel + (int64_t)e * c.N * 2, *lp = lpos + (int64_t)e * c.N * 2;
    for (int i = 0; i < c.N; ++i) {
        float *o = obs_env + i * c.obs_dim;
        int k = 0;
        o[k++] = av[2 * i]; o[k++] = av[2 * i + 1];
        o[k++] = ap[2 * i]; o[k++] = ap[2 * i + 1];
        for (int l = 0; l < c.N; ++l) { o[k++] = lp[2 * l] - ap[2 * i]; o[k++] = lp[2 * l + 1] - ap[2 * i + 1]; }
        for (int j = 0; j < c.N; ++j) if (j != i) { o[k++] = ap[2 * j] - ap[2 * i]; o[k++] = ap[2 * j + 1] - ap[2 * i + 1]; }
        for (int j = 0; j < c.N; ++j) if (j != i) { o[k++] = 0.f; o[k++] = 0.f; }  // comm channel (silent agents)
    }
}

__global__ void mpe_reset_kernel(MpeCfg c, uint64_t seed, uint64_t *episode_ctr, const int64_t *env_ids, int64_t n,
                                 float *apos, float *avel, float *lpos, int32_t *steps, float *obs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int e = env_ids ? (int)env_ids[i] : (int)i;
    const uint64_t ep = episode_ctr[e]++;
    reset_env(c, e, seed, ep, apos, avel, lpos, steps);
    write_obs(c, e, apos, avel, lpos, obs + (int64_t)e * c.N * c.obs_dim);
}

// One joint step of every env.  Outputs (all [n_env][N] unless noted):
//   obs_next [n_env][N][obs_dim]  observation after the step (terminal observation for finished episodes)
//   obs_cur  [n_env][N][obs_dim]  observation the policy sees next (== obs_next, or the reset obs if done)
//   rew f32, terminated u8 (always 0 here), truncated u8;  done_env [n_env] u8
// auto_reset != 0: finished envs are re-initialised in place (Collector's env.reset(env_id=done ids),
// /root/reference/tianshou/data/collector.py:971).
__device__ void step_one_env(const MpeCfg &c, int e, uint64_t seed, uint64_t *episode_ctr,
                             const int32_t *__restrict__ act, float *apos, float *avel, float *lpos, int32_t *steps,
                             float *obs_next /* env block */, float *obs_cur /* env block or null */, float *rew,
                             uint8_t *term, uint8_t *trunc, uint8_t *done_env, int auto_reset) {
    const int N = c.N;
    float px[kMaxN], py[kMaxN], vx[kMaxN], vy[kMaxN], fx[kMaxN], fy[kMaxN];
    float *ap = apos + (int64_t)e * N * 2, *av = avel + (int64_t)e * N * 2;
    const float *lp = lpos + (int64_t)e * N * 2;
#pragma unroll
    for (int i = 0; i < kMaxN; ++i) {
        if (i < N) {
            px[i] = ap[2 * i]; py[i] = ap[2 * i + 1]; vx[i] = av[2 * i]; vy[i] = av[2 * i + 1];
            const int a = act[(int64_t)e * N + i];
            fx[i] = (a == 1 ? -1.f : (a == 2 ? 1.f : 0.f)) * c.accel;
            fy[i] = (a == 3 ? -1.f : (a == 4 ? 1.f : 0.f)) * c.accel;
        }
    }
    // soft contact forces between agent pairs
#pragma unroll
    for (int i = 0; i < kMaxN; ++i) {
#pragma unroll
        for (int j = i + 1; j < kMaxN; ++j) {
            if (j < N) {
                const float dx = px[i] - px[j], dy = py[i] - py[j];
                const float dist = sqrtf(dx * dx + dy * dy);
                const float dist_min = 2.f * c.agent_size;
                const float k = c.contact_margin;
                const float z = -(dist - dist_min) / k;
                const float pen = (z > 0.f ? z + log1pf(expf(-z)) : log1pf(expf(z))) * k;  // logaddexp(0, z) * k
                const float s = c.contact_force * pen / dist;
                fx[i] += s * dx; fy[i] += s * dy;
                fx[j] -= s * dx; fy[j] -= s * dy;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < kMaxN; ++i) {
        if (i < N) {
            vx[i] = vx[i] * (1.f - c.damping) + fx[i] * c.dt;
            vy[i] = vy[i] * (1.f - c.damping) + fy[i] * c.dt;
            if (c.max_speed > 0.f) {
                const float sp = sqrtf(vx[i] * vx[i] + vy[i] * vy[i]);
                if (sp > c.max_speed) { vx[i] = vx[i] / sp * c.max_speed; vy[i] = vy[i] / sp * c.max_speed; }
            }
            px[i] += vx[i] * c.dt;
            py[i] += vy[i] * c.dt;
            ap[2 * i] = px[i]; ap[2 * i + 1] = py[i]; av[2 * i] = vx[i]; av[2 * i + 1] = vy[i];
        }
    }
    // rewards
    float global = 0.f;
    for (int l = 0; l < N; ++l) {
        float m = INFINITY;
        for (int i = 0; i < N; ++i) {
            const float dx = px[i] - lp[2 * l], dy = py[i] - lp[2 * l + 1];
            m = fminf(m, sqrtf(dx * dx + dy * dy));
        }
        global -= m;
    }
    const int st = steps[e] + 1;
    steps[e] = st;
    const bool tr = st >= c.max_cycles;
    for (int i = 0; i < N; ++i) {
        float local = 0.f;
        for (int j = 0; j < N; ++j) {
            if (j != i) {
                const float dx = px[i] - px[j], dy = py[i] - py[j];
                if (sqrtf(dx * dx + dy * dy) < 2.f * c.agent_size) local -= 1.f;
            }
        }
        const int64_t li = (int64_t)e * N + i;
        rew[li] = global * (1.f - c.local_ratio) + local * c.local_ratio;
        term[li] = 0;
        trunc[li] = tr ? 1 : 0;
    }
    done_env[e] = tr ? 1 : 0;
    write_obs(c, e, apos, avel, lpos, obs_next);
    if (tr && auto_reset) {
        const uint64_t ep = episode_ctr[e]++;
        reset_env(c, e, seed, ep, apos, avel, lpos, steps);
    }
    if (obs_cur) write_obs(c, e, apos, avel, lpos, obs_cur);
}

constexpr int kEnvPerBlock = 16;  // small workgroups: 1024 envs spread over 64 CUs (the step is latency-bound)

// Lanes 0..15 of the single wave advance one env each and stage both observation blocks in LDS; then all 64
// lanes stream them out as one contiguous, coalesced segment per workgroup (obs rows of consecutive envs
// are adjacent in [n_env][N][obs_dim]).
__global__ __launch_bounds__(64) void mpe_step_kernel(MpeCfg c, uint64_t seed, uint64_t *episode_ctr,
                                                      const int32_t *__restrict__ act, float *apos, float *avel,
                                                      float *lpos, int32_t *steps, float *obs_next, float *obs_cur,
                                                      float *rew, uint8_t *term, uint8_t *trunc, uint8_t *done_env,
                                                      int auto_reset, uint64_t *tick, uint64_t tick_inc) {
    extern __shared__ float s_obs[];  // [2][kEnvPerBlock][N * obs_dim]
    const int row = c.N * c.obs_dim;
    const int e0 = blockIdx.x * kEnvPerBlock;
    const int e = e0 + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x == 0 && tick) *tick += tick_inc;
    float *s_next = s_obs, *s_cur = s_obs + kEnvPerBlock * row;
    if (threadIdx.x < kEnvPerBlock && e < c.n_env)
        step_one_env(c, e, seed, episode_ctr, act, apos, avel, lpos, steps, s_next + threadIdx.x * row,
                     obs_cur ? s_cur + threadIdx.x * row : nullptr, rew, term, trunc, done_env, auto_reset);
    __syncthreads();
    const int n_here = min(kEnvPerBlock, c.n_env - e0);
    float *g_next = obs_next + (int64_t)e0 * row;
    for (int i = threadIdx.x; i < n_here * row; i += 64) g_next[i] = s_next[i];
    if (obs_cur) {
        float *g_cur = obs_cur + (int64_t)e0 * row;
        for (int i = threadIdx.x; i < n_here * row; i += 64) g_cur[i] = s_cur[i];
    }
}

int check_cfg(const tsm_mpe_cfg *h, MpeCfg *c) {
    TSM_REQUIRE(h, "mpe: null cfg");
    TSM_REQUIRE(h->n_env >= 1 && h->n_agent >= 1 && h->n_agent <= kMaxN, "mpe: n_env >= 1, 1 <= n_agent <= %d", kMaxN);
    TSM_REQUIRE(h->max_cycles >= 1, "mpe: max_cycles must be >= 1");
    c->n_env = h->n_env; c->N = h->n_agent; c->obs_dim = 6 * h->n_agent; c->max_cycles = h->max_cycles;
    c->dt = (float)h->dt; c->damping = (float)h->damping; c->contact_force = (float)h->contact_force;
    c->contact_margin = (float)h->contact_margin; c->agent_size = (float)h->agent_size;
    c->landmark_size = (float)h->landmark_size; c->accel = (float)h->accel; c->max_speed = (float)h->max_speed;
    c->local_ratio = (float)h->local_ratio;
    return TSM_OK;
}

}  // namespace

TSM_EXPORT int tsm_mpe_spread_reset(const tsm_mpe_cfg *cfg_host, uint64_t seed, uint64_t *episode_ctr,
                                    const int64_t *env_ids, int64_t n_ids, float *agent_pos, float *agent_vel,
                                    float *landmark_pos, int32_t *steps, float *obs_out, void *stream) {
    MpeCfg c;
    if (int rc = check_cfg(cfg_host, &c)) return rc;
    const int64_t n = env_ids ? n_ids : c.n_env;
    if (n == 0) return TSM_OK;
    TSM_REQUIRE(episode_ctr && agent_pos && agent_vel && landmark_pos && steps && obs_out, "tsm_mpe_spread_reset: null pointer");
    hipLaunchKernelGGL(mpe_reset_kernel, dim3((unsigned)ceil_div(n, 64)), dim3(64), 0, tsm_stream(stream), c, seed,
                       episode_ctr, env_ids, n, agent_pos, agent_vel, landmark_pos, steps, obs_out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int tsm_mpe_spread_step(const tsm_mpe_cfg *cfg_host, uint64_t seed, uint64_t *episode_ctr,
                                   const int32_t *act, float *agent_pos, float *agent_vel, float *landmark_pos,
                                   int32_t *steps, float *obs_next_out, float *obs_cur_out, float *rew_out,
                                   uint8_t *terminated_out, uint8_t *truncated_out, uint8_t *done_env_out,
                                   int auto_reset, uint64_t *rng_tick, uint64_t rng_tick_inc, void *stream) {
    MpeCfg c;
    if (int rc = check_cfg(cfg_host, &c)) return rc;
    TSM_REQUIRE(episode_ctr && act && agent_pos && agent_vel && landmark_pos && steps && obs_next_out && rew_out &&
                    terminated_out && truncated_out && done_env_out,
                "tsm_mpe_spread_step: null pointer");
    const size_t shmem = (size_t)2 * kEnvPerBlock * c.N * c.obs_dim * sizeof(float);
    hipLaunchKernelGGL(mpe_step_kernel, dim3((unsigned)ceil_div(c.n_env, kEnvPerBlock)), dim3(64), shmem,
                       tsm_stream(stream), c, seed, episode_ctr, act, agent_pos, agent_vel, landmark_pos, steps, obs_next_out, obs_cur_out,
                       rew_out, terminated_out, truncated_out, done_env_out, auto_reset, rng_tick, rng_tick_inc);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}
