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
// One thread per env: this env family is a functional row, not a tuned one (4096 envs = 4096 threads).
#include "common.h"
#include "philox.h"

namespace {

constexpr int kMaxAgents = 8, kMaxObst = 4;

struct TagCfg {
    int n_env, n_adv, n_good, n_obst, max_cycles, obs_dim;
    float dt, damping, contact_force, contact_margin;
    float adv_size, good_size, obst_size, adv_accel, good_accel, adv_speed, good_speed;
};

__device__ __forceinline__ float uni(uint32_t bits, float lo, float hi) { return lo + (hi - lo) * tsm_u01(bits); }

// positions ~ U(-1, 1)^2 for agents, U(-0.9, 0.9)^2 for obstacles; Philox counter (episode * n_env + e) * 16 + entity
__device__ void tag_reset_env(const TagCfg &c, int e, uint64_t seed, uint64_t episode, float *ap, float *av, float *lp) {
    const int NA = c.n_adv + c.n_good;
    for (int i = 0; i < NA + c.n_obst; ++i) {
        uint32_t r[4];
        tsm_philox4(seed, (episode * (uint64_t)c.n_env + (uint64_t)e) * 16ull + (uint64_t)i, r);
        if (i < NA) {
            ap[2 * i] = uni(r[0], -1.f, 1.f); ap[2 * i + 1] = uni(r[1], -1.f, 1.f);
            av[2 * i] = 0.f; av[2 * i + 1] = 0.f;
        } else {
            lp[2 * (i - NA)] = uni(r[0], -0.9f, 0.9f); lp[2 * (i - NA) + 1] = uni(r[1], -0.9f, 0.9f);
        }
    }
}

__device__ void tag_obs(const TagCfg &c, const float *ap, const float *av, const float *lp, int i, float *o) {
    const int NA = c.n_adv + c.n_good;
    int k = 0;
    o[k++] = av[2 * i]; o[k++] = av[2 * i + 1];
    o[k++] = ap[2 * i]; o[k++] = ap[2 * i + 1];
    for (int l = 0; l < c.n_obst; ++l) { o[k++] = lp[2 * l] - ap[2 * i]; o[k++] = lp[2 * l + 1] - ap[2 * i + 1]; }
    for (int j = 0; j < NA; ++j)
        if (j != i) { o[k++] = ap[2 * j] - ap[2 * i]; o[k++] = ap[2 * j + 1] - ap[2 * i + 1]; }
    for (int j = c.n_adv; j < NA; ++j)
        if (j != i) { o[k++] = av[2 * j]; o[k++] = av[2 * j + 1]; }
    while (k < c.obs_dim) o[k++] = 0.f;  // good agents: padded to the adversaries' width
}

__device__ __forceinline__ float softplus_k(float z, float k) {  // logaddexp(0, z) * k
    return (z > 0.f ? z + log1pf(expf(-z)) : log1pf(expf(z))) * k;
}

__device__ __forceinline__ float bound_pen(float x) {
    if (x < 0.9f) return 0.f;
    if (x < 1.0f) return (x - 0.9f) * 10.f;
    return fminf(expf(2.f * x - 2.f), 10.f);
}

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

__global__ void tag_step_kernel(TagCfg c, uint64_t seed, uint64_t *episode_ctr, const int32_t *__restrict__ act,
                                float *apos, float *avel, float *lpos, int32_t *steps, float *obs_next, float *obs_cur,
                                float *rew, uint8_t *term, uint8_t *trunc, uint8_t *done_env, int auto_reset,
                                uint64_t *tick, uint64_t tick_inc) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e == 0 && tick) *tick += tick_inc;  // sampling counter of the NEXT policy call (graph-replay safe)
    if (e >= c.n_env) return;
    const int NA = c.n_adv + c.n_good, NE = NA + c.n_obst;
    float px[kMaxAgents + kMaxObst], py[kMaxAgents + kMaxObst], fx[kMaxAgents], fy[kMaxAgents], sz[kMaxAgents + kMaxObst];
    float *ap = apos + (int64_t)e * NA * 2, *av = avel + (int64_t)e * NA * 2, *lp = lpos + (int64_t)e * c.n_obst * 2;
    for (int i = 0; i < NA; ++i) {
        px[i] = ap[2 * i]; py[i] = ap[2 * i + 1];
        sz[i] = i < c.n_adv ? c.adv_size : c.good_size;
        const float accel = i < c.n_adv ? c.adv_accel : c.good_accel;
        const int a = act[(int64_t)e * NA + i];
        fx[i] = (a == 1 ? -1.f : (a == 2 ? 1.f : 0.f)) * accel;
        fy[i] = (a == 3 ? -1.f : (a == 4 ? 1.f : 0.f)) * accel;
    }
    for (int l = 0; l < c.n_obst; ++l) { px[NA + l] = lp[2 * l]; py[NA + l] = lp[2 * l + 1]; sz[NA + l] = c.obst_size; }
    // soft contact forces over entity pairs a < b (agents first, then obstacles); only agents move
    for (int a = 0; a < NA; ++a)
        for (int b = a + 1; b < NE; ++b) {
            const float dx = px[a] - px[b], dy = py[a] - py[b];
            const float dmin = sz[a] + sz[b];
            const float d2 = dx * dx + dy * dy;
            const float far = dmin + 105.f * c.contact_margin;  // beyond it expf underflows: the force is exactly 0
            if (d2 > far * far) continue;
            const float dist = sqrtf(d2);
            const float pen = softplus_k(-(dist - dmin) / c.contact_margin, c.contact_margin);
            const float s = c.contact_force * pen / dist;
            fx[a] += s * dx; fy[a] += s * dy;
            if (b < NA) { fx[b] -= s * dx; fy[b] -= s * dy; }
        }
    for (int i = 0; i < NA; ++i) {
        float vx = av[2 * i] * (1.f - c.damping) + fx[i] * c.dt;
        float vy = av[2 * i + 1] * (1.f - c.damping) + fy[i] * c.dt;
        const float vmax = i < c.n_adv ? c.adv_speed : c.good_speed;
        const float sp = sqrtf(vx * vx + vy * vy);
        if (sp > vmax) { vx = vx / sp * vmax; vy = vy / sp * vmax; }
        av[2 * i] = vx; av[2 * i + 1] = vy;
        px[i] += vx * c.dt; py[i] += vy * c.dt;
        ap[2 * i] = px[i]; ap[2 * i + 1] = py[i];
    }
    // rewards on the new positions
    float adv_rew = 0.f;
    float good_rew[kMaxAgents];
    for (int g = c.n_adv; g < NA; ++g) {
        float r = 0.f;
        for (int a = 0; a < c.n_adv; ++a) {
            const float dx = px[a] - px[g], dy = py[a] - py[g];
            if (sqrtf(dx * dx + dy * dy) < sz[a] + sz[g]) { r -= 10.f; adv_rew += 10.f; }
        }
        r -= bound_pen(fabsf(px[g]));
        r -= bound_pen(fabsf(py[g]));
        good_rew[g] = r;
    }
    const int stp = steps[e] + 1;
    const bool tr = stp >= c.max_cycles;
    for (int i = 0; i < NA; ++i) {
        const int64_t o = (int64_t)e * NA + i;
        rew[o] = i < c.n_adv ? adv_rew : good_rew[i];
        term[o] = 0;
        trunc[o] = tr ? 1 : 0;
        tag_obs(c, ap, av, lp, i, obs_next + o * c.obs_dim);
    }
    done_env[e] = tr ? 1 : 0;
    if (tr && auto_reset) {
        const uint64_t ep = episode_ctr[e]++;
        tag_reset_env(c, e, seed, ep, ap, av, lp);
        steps[e] = 0;
    } else {
        steps[e] = stp;
    }
    if (obs_cur)
        for (int i = 0; i < NA; ++i) tag_obs(c, ap, av, lp, i, obs_cur + ((int64_t)e * NA + i) * c.obs_dim);
}

int check_cfg(const tsm_mpe_tag_cfg *h, TagCfg *c) {
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

}  // namespace

TSM_EXPORT int tsm_mpe_tag_obs_dim(const tsm_mpe_tag_cfg *cfg_host) {
    TagCfg c;
    return check_cfg(cfg_host, &c) == TSM_OK ? c.obs_dim : -1;
}

TSM_EXPORT int tsm_mpe_tag_reset(const tsm_mpe_tag_cfg *cfg_host, uint64_t seed, uint64_t *episode_ctr,
                                 const int64_t *env_ids, int64_t n_ids, float *agent_pos, float *agent_vel,
                                 float *landmark_pos, int32_t *steps, float *obs_out, void *stream) {
    TagCfg c;
    if (int rc = check_cfg(cfg_host, &c)) return rc;
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
    if (int rc = check_cfg(cfg_host, &c)) return rc;
    TSM_REQUIRE(episode_ctr && act && agent_pos && agent_vel && landmark_pos && steps && obs_next_out && rew_out &&
                    terminated_out && truncated_out && done_env_out,
                "tsm_mpe_tag_step: null pointer");
    hipLaunchKernelGGL(tag_step_kernel, dim3((unsigned)ceil_div(c.n_env, 64)), dim3(64), 0, tsm_stream(stream), c, seed,
                       episode_ctr, act, agent_pos, agent_vel, landmark_pos, steps, obs_next_out, obs_cur_out, rew_out,
                       terminated_out, truncated_out, done_env_out, auto_reset, rng_tick, rng_tick_inc);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}
