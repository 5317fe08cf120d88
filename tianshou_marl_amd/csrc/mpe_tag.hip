// mpe_tag.hip -- batched, device-resident PettingZoo-MPE `simple_tag` worlds (predator-prey, two teams).
//
// Row (f)1 of SURVEY.md section 8 / BASELINE configs[4]: the env the reference would drive through
// EnhancedPettingZooEnv + a vector env (/root/reference/tianshou/env/enhanced_pettingzoo_env.py:130-222,
// venvs.py:237-322) for grouped (team) policies and the self-play / league trainers
// (algorithm/multiagent/training_coordinator.py:413-747).  As for simple_spread, the pettingzoo sources are not in
// the reference tree: the dynamics are restated from the published MPE specification, parity with pettingzoo is
// UNPINNED, and the kernel is pinned against oracle/mpe_tag_oracle.py (same spec, numpy f64).
//
// World (dim_p = 2, dt 0.1, damping 0.25, contact force 100, margin 1e-3), entities = agents then landmarks:
//   adversaries (first n_adv agents): size .075, accel 3.0, max speed 1.0;  good agents: size .05, accel 4.0, 1.3
//   obstacles (landmarks): size .2, collide, do not move;  every entity pair exchanges the soft contact force
//   action u in {noop, -x, +x, -y, +y} * accel;  v <- v (1 - damping) + f dt, clamped to max speed;  p <- p + v dt
// Rewards (no local_ratio mixing): good agent: -10 per adversary touching it, minus the boundary penalty
//   bound(|x|) = 0 (<0.9), 10 (|x| - 0.9) (<1.0), min(exp(2|x| - 2), 10) per coordinate;
//   every adversary: +10 per (good agent, adversary) pair in contact (shared by the team).
// Observation: [vel 2, pos 2, obstacles rel 2 n_obst, other agents rel 2 (NA - 1), good agents' velocities
//   (others only) 2 each]; adversaries see 2 n_good more numbers than good agents do, so rows are zero-padded to
//   the common width obs_dim = 4 + 2 n_obst + 2 (NA - 1) + 2 n_good (PettingZooEnv requires identical spaces,
//   pettingzoo_env.py:55-67).  Truncation at max_cycles; never terminates.
// The step kernel runs one lane per (env, agent) on LDS-staged state (16 envs per workgroup); reset: one thread per env.
#include "common.h"
#include "mpe_tag_dev.h"

namespace {

constexpr int kMaxAgents = kTagMaxAgents, kMaxObst = kTagMaxObst;

__global__ void tag_reset_kernel(TagCfg c, uint64_t seed, uint64_t *episode_ctr, const int64_t *env_ids, int64_t n,
                                 float *apos, float *avel, float *lpos, int32_t *steps, float *obs) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int e = env_ids ? (int)env_ids[idx] : (int)idx;
    const int NA = c.n_adv + c.n_good;
    const uint64_t ep = episode_ctr[e]++;
    float *ap = apos + (int64_t)e * NA * 2, *av = avel + (int64_t)e * NA * 2, *lp = lpos + (int64_t)e * c.n_obst * 2;
    tag_reset_env(c, e, seed, ep, ap, av, lp);
    steps[e] = 0;
    for (int i = 0; i < NA; ++i) tag_obs(c, ap, av, lp, i, obs + ((int64_t)e * NA + i) * c.obs_dim);
}

constexpr int kTagEnvPerBlock = 16;
constexpr int kTagThreads = kTagEnvPerBlock * kMaxAgents;  // one lane per (env, agent)

// One joint step of every env, one lane per agent (same structure as mpe_step_kernel in mpe.hip): the env state of
// 16 envs is staged in LDS, every agent lane sums the soft contact forces it receives from the other entities in
// increasing entity index -- the order in which the serial pair loop (a < b) accumulates into f[i] -- so the result
// is bit-identical to the pair formulation; observation rows leave through LDS as coalesced writes.
__global__ __launch_bounds__(kTagThreads) void tag_step_kernel(
    TagCfg c, uint64_t seed, uint64_t *episode_ctr, const int32_t *__restrict__ act, float *apos, float *avel,
    float *lpos, int32_t *steps, float *obs_next, float *obs_cur, float *rew, uint8_t *term, uint8_t *trunc,
    uint8_t *done_env, int auto_reset, uint64_t *tick, uint64_t tick_inc) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int NA = c.n_adv + c.n_good;
    const int st = 2 * NA, lst = 2 * c.n_obst, row = NA * c.obs_dim;
    float *s_ap = sm;                                   // [16][NA][2]
    float *s_av = s_ap + kTagEnvPerBlock * st;
    float *s_lp = s_av + kTagEnvPerBlock * st;          // [16][n_obst][2]
    float *s_hit = s_lp + kTagEnvPerBlock * (lst > 0 ? lst : 2);  // [16][NA] 10 * (adversaries touching good agent g)
    int *s_tr = reinterpret_cast<int *>(s_hit + kTagEnvPerBlock * NA);  // [16]
    uint64_t *s_ep = reinterpret_cast<uint64_t *>(   // [16]  (aligned by index arithmetic: keeps the pointer an LDS pointer)
        sm + (((int)(reinterpret_cast<float *>(s_tr + kTagEnvPerBlock) - sm) + 1) & ~1));
    float *s_next = reinterpret_cast<float *>(s_ep + kTagEnvPerBlock);  // [16][row]
    const int e0 = blockIdx.x * kTagEnvPerBlock;
    const int n_here = min(kTagEnvPerBlock, c.n_env - e0);
    if (blockIdx.x == 0 && threadIdx.x == 0 && tick) *tick += tick_inc;  // sampling counter of the NEXT policy call
    for (int k = threadIdx.x; k < n_here * st; k += kTagThreads) {
        s_ap[k] = apos[(int64_t)e0 * st + k];
        s_av[k] = avel[(int64_t)e0 * st + k];
    }
    for (int k = threadIdx.x; k < n_here * lst; k += kTagThreads) s_lp[k] = lpos[(int64_t)e0 * lst + k];
    __syncthreads();
    const int el = threadIdx.x / NA, i = threadIdx.x - el * NA;  // agent lane -> (env, agent)
    const bool live = threadIdx.x < n_here * NA;
    const int e = e0 + el;
    const float *ap = s_ap + el * st, *av = s_av + el * st, *lp = s_lp + el * lst;
    const bool is_adv = i < c.n_adv;
    float npx = 0.f, npy = 0.f, nvx = 0.f, nvy = 0.f;
    if (live) tag_agent_move(c, ap, av, lp, i, act[(int64_t)e * NA + i], npx, npy, nvx, nvy);
    __syncthreads();
    if (live) {
        s_ap[el * st + 2 * i] = npx; s_ap[el * st + 2 * i + 1] = npy;
        s_av[el * st + 2 * i] = nvx; s_av[el * st + 2 * i + 1] = nvy;
    }
    __syncthreads();
    // rewards on the new positions: a good agent counts the adversaries touching it (multiples of 10: exact in f32)
    float my_rew = 0.f;
    if (live) {
        float hit;
        my_rew = tag_own_reward(c, ap, i, npx, npy, hit);
        s_hit[el * NA + i] = hit;
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
        if (is_adv) for (int g = c.n_adv; g < NA; ++g) my_rew += s_hit[el * NA + g];  // shared by the team
        const int64_t li = (int64_t)e * NA + i;
        rew[li] = my_rew;
        term[li] = 0;
        trunc[li] = (uint8_t)s_tr[el];
    }
    for (int k = threadIdx.x; k < n_here * row; k += kTagThreads) {
        const int r = k / c.obs_dim, kk = k - r * c.obs_dim, ee = r / NA;
        s_next[k] = tag_obs_elem(c, s_ap + ee * st, s_av + ee * st, s_lp + ee * lst, r - ee * NA, kk);
    }
    if (live && i == 0 && s_tr[el] && auto_reset) {
        const uint64_t ep = episode_ctr[e];
        s_ep[el] = ep;
        episode_ctr[e] = ep + 1;
    }
    __syncthreads();
    if (live && s_tr[el] && auto_reset)  // re-initialise: agent lane i draws agent i (and obstacles i, i + NA, ...)
        tag_reset_lane(c, e, seed, s_ep[el], i, s_ap + el * st, s_av + el * st, s_lp + el * lst);
    __syncthreads();
    float *g_next = obs_next + (int64_t)e0 * row;
    for (int k = threadIdx.x; k < n_here * row; k += kTagThreads) g_next[k] = s_next[k];
    if (obs_cur) {
        float *g_cur = obs_cur + (int64_t)e0 * row;
        for (int k = threadIdx.x; k < n_here * row; k += kTagThreads) {
            const int r = k / c.obs_dim, kk = k - r * c.obs_dim, ee = r / NA;
            g_cur[k] = tag_obs_elem(c, s_ap + ee * st, s_av + ee * st, s_lp + ee * lst, r - ee * NA, kk);
        }
    }
    for (int k = threadIdx.x; k < n_here * st; k += kTagThreads) {
        apos[(int64_t)e0 * st + k] = s_ap[k];
        avel[(int64_t)e0 * st + k] = s_av[k];
    }
    for (int k = threadIdx.x; k < n_here * lst; k += kTagThreads) lpos[(int64_t)e0 * lst + k] = s_lp[k];
}

}  // namespace

int tsm_mpe_tag_check_cfg(const tsm_mpe_tag_cfg *h, TagCfg *c) {
    TSM_REQUIRE(h, "simple_tag: null config");
    TSM_REQUIRE(h->n_env >= 1 && h->n_adv >= 1 && h->n_good >= 1 && h->n_adv + h->n_good <= kMaxAgents &&
                    h->n_obst >= 0 && h->n_obst <= kMaxObst && h->max_cycles >= 1,
                "simple_tag: sizes out of range (agents <= %d, obstacles <= %d)", kMaxAgents, kMaxObst);
    const int NA = h->n_adv + h->n_good;
    c->n_env = h->n_env; c->n_adv = h->n_adv; c->n_good = h->n_good; c->n_obst = h->n_obst; c->max_cycles = h->max_cycles;
    c->obs_dim = 4 + 2 * h->n_obst + 2 * (NA - 1) + 2 * h->n_good;
    c->dt = (float)h->dt; c->damping = (float)h->damping; c->contact_force = (float)h->contact_force;
    c->contact_margin = (float)h->contact_margin;
    c->adv_size = (float)h->adv_size; c->good_size = (float)h->good_size; c->obst_size = (float)h->obst_size;
    c->adv_accel = (float)h->adv_accel; c->good_accel = (float)h->good_accel;
    c->adv_speed = (float)h->adv_speed; c->good_speed = (float)h->good_speed;
    return TSM_OK;
}

TSM_EXPORT int tsm_mpe_tag_obs_dim(const tsm_mpe_tag_cfg *cfg_host) {
    TagCfg c;
    return tsm_mpe_tag_check_cfg(cfg_host, &c) == TSM_OK ? c.obs_dim : -1;
}

TSM_EXPORT int tsm_mpe_tag_reset(const tsm_mpe_tag_cfg *cfg_host, uint64_t seed, uint64_t *episode_ctr,
                                 const int64_t *env_ids, int64_t n_ids, float *agent_pos, float *agent_vel,
                                 float *landmark_pos, int32_t *steps, float *obs_out, void *stream) {
    TagCfg c;
    if (int rc = tsm_mpe_tag_check_cfg(cfg_host, &c)) return rc;
    const int64_t n = env_ids ? n_ids : c.n_env;
    if (n == 0) return TSM_OK;
    TSM_REQUIRE(episode_ctr && agent_pos && agent_vel && landmark_pos && steps && obs_out, "tsm_mpe_tag_reset: null pointer");
    hipLaunchKernelGGL(tag_reset_kernel, dim3((unsigned)ceil_div(n, 64)), dim3(64), 0, tsm_stream(stream), c, seed,
                       episode_ctr, env_ids, n, agent_pos, agent_vel, landmark_pos, steps, obs_out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int tsm_mpe_tag_step(const tsm_mpe_tag_cfg *cfg_host, uint64_t seed, uint64_t *episode_ctr,
                                const int32_t *act, float *agent_pos, float *agent_vel, float *landmark_pos,
                                int32_t *steps, float *obs_next_out, float *obs_cur_out, float *rew_out,
                                uint8_t *terminated_out, uint8_t *truncated_out, uint8_t *done_env_out, int auto_reset,
                                uint64_t *rng_tick, uint64_t rng_tick_inc, void *stream) {
    TagCfg c;
    if (int rc = tsm_mpe_tag_check_cfg(cfg_host, &c)) return rc;
    TSM_REQUIRE(episode_ctr && act && agent_pos && agent_vel && landmark_pos && steps && obs_next_out && rew_out &&
                    terminated_out && truncated_out && done_env_out,
                "tsm_mpe_tag_step: null pointer");
    const int NA = c.n_adv + c.n_good;
    const size_t shmem = sizeof(float) * (size_t)kTagEnvPerBlock * (4 * NA + (c.n_obst > 0 ? 2 * c.n_obst : 2) + NA + 1 +
                                                                     NA * c.obs_dim) + sizeof(uint64_t) * (kTagEnvPerBlock + 1);
    hipLaunchKernelGGL(tag_step_kernel, dim3((unsigned)ceil_div(c.n_env, kTagEnvPerBlock)), dim3(kTagThreads), shmem,
                       tsm_stream(stream), c, seed,
                       episode_ctr, act, agent_pos, agent_vel, landmark_pos, steps, obs_next_out, obs_cur_out, rew_out,
                       terminated_out, truncated_out, done_env_out, auto_reset, rng_tick, rng_tick_inc);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}
