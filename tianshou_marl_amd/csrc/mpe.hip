// mpe.hip -- batched, device-resident PettingZoo-MPE `simple_spread` worlds (stand-alone env kernels).
//
// Row (f)1 of SURVEY.md section 8: the reference drives third-party pettingzoo envs one Python object
// at a time through PettingZooEnv / EnhancedPettingZooEnv + DummyVectorEnv
// (/root/reference/tianshou/env/pettingzoo_env.py:71-120, enhanced_pettingzoo_env.py:130-222,
//  venvs.py:237-322); at 1k-8k envs that loop is the bottleneck, so the dynamics are restated here.
// The pettingzoo 1.24.2 MPE sources are NOT part of the reference tree and pettingzoo is not installed:
// the physics (mpe_dev.h) is restated from the published MPE specification and its parity with pettingzoo is
// UNPINNED; tests pin this kernel against oracle/mpe_oracle.py (same spec, numpy f64).
//
// World (dim_p = 2, dt = 0.1, damping = 0.25, contact_force = 100, contact_margin = 1e-3):
//   action u in {noop, -x, +x, -y, +y} * accel;  agent-agent soft contact forces;
//   v <- v*(1-damping) + (f/mass)*dt ; clamp |v| <= max_speed (if > 0) ; p <- p + v*dt
// simple_spread: N agents (size .15, accel 5), N landmarks; reward = (1-local_ratio)*global + local_ratio*local,
//   global = -sum_landmarks min_agents dist, local = -#collisions(agent); obs = [vel, pos, landmarks rel,
//   others rel, comm(0)] (6N); truncation at max_cycles.
// State SoA [n_env][N][2].  The step is latency-bound (1024 envs): a workgroup owns 16 envs (64 workgroups over
// 64 CUs for 1024 envs), one lane per (env, agent); state and both observation blocks are staged in LDS and
// streamed in/out as contiguous, coalesced segments.
#include "common.h"
#include "mpe_dev.h"

namespace {

__global__ void mpe_reset_kernel(MpeCfg c, uint64_t seed, uint64_t *episode_ctr, const int64_t *env_ids, int64_t n,
                                 float *apos, float *avel, float *lpos, int32_t *steps, float *obs) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int e = env_ids ? (int)env_ids[idx] : (int)idx;
    const uint64_t ep = episode_ctr[e]++;
    float *ap = apos + (int64_t)e * c.N * 2, *av = avel + (int64_t)e * c.N * 2, *lp = lpos + (int64_t)e * c.N * 2;
    for (int i = 0; i < c.N; ++i) mpe_reset_agent(c, e, seed, ep, i, ap, av, lp);
    steps[e] = 0;
    float *o = obs + (int64_t)e * c.N * c.obs_dim;
    for (int i = 0; i < c.N; ++i)
        for (int k = 0; k < c.obs_dim; ++k) o[i * c.obs_dim + k] = mpe_obs_elem(c, ap, av, lp, i, k);
}

constexpr int kEnvPerBlock = 16;
constexpr int kStepThreads = 128;  // >= kEnvPerBlock * kMpeMaxN agent lanes

// One joint step of every env.  Outputs (all [n_env][N] unless noted):
//   obs_next [n_env][N][obs_dim]  observation after the step (terminal observation for finished episodes)
//   obs_cur  [n_env][N][obs_dim]  observation the policy sees next (== obs_next, or the reset obs if done)
//   rew f32, terminated u8 (always 0 here), truncated u8;  done_env [n_env] u8
// auto_reset != 0: finished envs are re-initialised in place (Collector's env.reset(env_id=done ids),
// /root/reference/tianshou/data/collector.py:971).
__global__ __launch_bounds__(kStepThreads) void mpe_step_kernel(
    MpeCfg c, uint64_t seed, uint64_t *episode_ctr, const int32_t *__restrict__ act, float *apos, float *avel,
    float *lpos, int32_t *steps, float *obs_next, float *obs_cur, float *rew, uint8_t *term, uint8_t *trunc,
    uint8_t *done_env, int auto_reset, uint64_t *tick, uint64_t tick_inc) {
    extern __shared__ float sm[];
    const int N = c.N, row = N * c.obs_dim, st = 2 * N;
    float *s_ap = sm;                              // [16][N][2]
    float *s_av = s_ap + kEnvPerBlock * st;
    float *s_lp = s_av + kEnvPerBlock * st;
    float *s_m = s_lp + kEnvPerBlock * st;         // [16][N] landmark minima
    int *s_tr = reinterpret_cast<int *>(s_m + kEnvPerBlock * N);  // [16] truncated flag
    float *s_next = reinterpret_cast<float *>(s_tr + kEnvPerBlock);  // [16][row]
    float *s_cur = s_next + kEnvPerBlock * row;
    const int e0 = blockIdx.x * kEnvPerBlock;
    const int n_here = min(kEnvPerBlock, c.n_env - e0);
    if (blockIdx.x == 0 && threadIdx.x == 0 && tick) *tick += tick_inc;
    // state in (contiguous block of n_here envs)
    for (int i = threadIdx.x; i < n_here * st; i += kStepThreads) {
        s_ap[i] = apos[(int64_t)e0 * st + i];
        s_av[i] = avel[(int64_t)e0 * st + i];
        s_lp[i] = lpos[(int64_t)e0 * st + i];
    }
    __syncthreads();
    const int lane_rows = n_here * N;
    const int el = threadIdx.x / N, i = threadIdx.x - el * N;  // agent lane -> (env, agent)
    const bool live = threadIdx.x < lane_rows;
    const int e = e0 + el;
    float npx = 0.f, npy = 0.f, nvx = 0.f, nvy = 0.f;
    if (live) mpe_agent_move(c, s_ap + el * st, s_av + el * st, i, act[(int64_t)e * N + i], npx, npy, nvx, nvy);
    __syncthreads();
    if (live) {
        s_ap[el * st + 2 * i] = npx; s_ap[el * st + 2 * i + 1] = npy;
        s_av[el * st + 2 * i] = nvx; s_av[el * st + 2 * i + 1] = nvy;
    }
    __syncthreads();
    float local = 0.f;
    if (live) {
        const MpePos pos = mpe_load_pos(c, s_ap + el * st);
        s_m[el * N + i] = mpe_landmark_min_dist(c, pos, s_lp + el * st, i);
        local = mpe_local_penalty(c, pos, s_ap + el * st, i);
        if (i == 0) {
            const int stp = steps[e] + 1;
            const int tr = stp >= c.max_cycles;
            s_tr[el] = tr;
            steps[e] = (tr && auto_reset) ? 0 : stp;
            done_env[e] = (uint8_t)tr;
        }
    }
    __syncthreads();
    if (live) {
        const int64_t li = (int64_t)e * N + i;
        rew[li] = mpe_reward(c, s_m + el * N, local);
        term[li] = 0;
        trunc[li] = (uint8_t)s_tr[el];
    }
    for (int k = threadIdx.x; k < n_here * row; k += kStepThreads) {
        const int r = k / c.obs_dim, kk = k - r * c.obs_dim, ee = r / N;
        s_next[k] = mpe_obs_elem(c, s_ap + ee * st, s_av + ee * st, s_lp + ee * st, r - ee * N, kk);
    }
    __syncthreads();
    uint64_t *s_ep = reinterpret_cast<uint64_t *>(s_cur);  // [16] episode counters of finished envs (64-B aligned)
    if (live && i == 0 && s_tr[el] && auto_reset) {
        const uint64_t ep = episode_ctr[e];
        s_ep[el] = ep;
        episode_ctr[e] = ep + 1;
    }
    __syncthreads();
    if (live && s_tr[el] && auto_reset)
        mpe_reset_agent(c, e, seed, s_ep[el], i, s_ap + el * st, s_av + el * st, s_lp + el * st);
    __syncthreads();
    float *g_next = obs_next + (int64_t)e0 * row;
    for (int k = threadIdx.x; k < n_here * row; k += kStepThreads) g_next[k] = s_next[k];
    if (obs_cur) {
        float *g_cur = obs_cur + (int64_t)e0 * row;
        for (int k = threadIdx.x; k < n_here * row; k += kStepThreads) {
            const int r = k / c.obs_dim, kk = k - r * c.obs_dim, ee = r / N;
            g_cur[k] = mpe_obs_elem(c, s_ap + ee * st, s_av + ee * st, s_lp + ee * st, r - ee * N, kk);
        }
    }
    for (int k = threadIdx.x; k < n_here * st; k += kStepThreads) {
        apos[(int64_t)e0 * st + k] = s_ap[k];
        avel[(int64_t)e0 * st + k] = s_av[k];
        lpos[(int64_t)e0 * st + k] = s_lp[k];
    }
}

}  // namespace

int tsm_mpe_check_cfg(const tsm_mpe_cfg *h, MpeCfg *c) {
    TSM_REQUIRE(h, "mpe: null cfg");
    TSM_REQUIRE(h->n_env >= 1 && h->n_agent >= 1 && h->n_agent <= kMpeMaxN, "mpe: n_env >= 1, 1 <= n_agent <= %d",
                kMpeMaxN);
    TSM_REQUIRE(h->max_cycles >= 1, "mpe: max_cycles must be >= 1");
    c->n_env = h->n_env; c->N = h->n_agent; c->obs_dim = 6 * h->n_agent; c->max_cycles = h->max_cycles;
    c->dt = (float)h->dt; c->damping = (float)h->damping; c->contact_force = (float)h->contact_force;
    c->contact_margin = (float)h->contact_margin; c->agent_size = (float)h->agent_size;
    c->landmark_size = (float)h->landmark_size; c->accel = (float)h->accel; c->max_speed = (float)h->max_speed;
    c->local_ratio = (float)h->local_ratio;
    return TSM_OK;
}

TSM_EXPORT int tsm_mpe_spread_reset(const tsm_mpe_cfg *cfg_host, uint64_t seed, uint64_t *episode_ctr,
                                    const int64_t *env_ids, int64_t n_ids, float *agent_pos, float *agent_vel,
                                    float *landmark_pos, int32_t *steps, float *obs_out, void *stream) {
    MpeCfg c;
    if (int rc = tsm_mpe_check_cfg(cfg_host, &c)) return rc;
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
    if (int rc = tsm_mpe_check_cfg(cfg_host, &c)) return rc;
    TSM_REQUIRE(episode_ctr && act && agent_pos && agent_vel && landmark_pos && steps && obs_next_out && rew_out &&
                    terminated_out && truncated_out && done_env_out,
                "tsm_mpe_spread_step: null pointer");
    const size_t shmem = ((size_t)kEnvPerBlock * (3 * 2 * c.N + c.N + 1) + (size_t)2 * kEnvPerBlock * c.N * c.obs_dim) *
                         sizeof(float);
    hipLaunchKernelGGL(mpe_step_kernel, dim3((unsigned)ceil_div(c.n_env, kEnvPerBlock)), dim3(kStepThreads), shmem,
                       tsm_stream(stream), c, seed, episode_ctr, act, agent_pos, agent_vel, landmark_pos, steps,
                       obs_next_out, obs_cur_out, rew_out, terminated_out, truncated_out, done_env_out, auto_reset,
                       rng_tick, rng_tick_inc);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}
