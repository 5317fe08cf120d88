// mpe_tag_dev.h -- device-side simple_tag world step shared by the stand-alone env kernels (mpe_tag.hip) and the
// persistent two-team rollout kernel (rollout_tag.hip).  Specification and provenance: see mpe_tag.hip (restated from
// the published PettingZoo-MPE spec; pettingzoo sources are not in the reference tree -> parity unpinned).
//
// Everything is formulated per AGENT LANE (one lane = one (env, agent) row) on an env state block that lives in LDS:
// ap/av = this env's [NA][2] positions / velocities, lp = its [n_obst][2] obstacle positions.  A world step is
//   1. tag_agent_move   (reads the OLD state of all entities, returns the lane's new pos/vel in registers)
//   --- barrier; lanes write their new pos/vel; barrier ---
//   2. tag_good_reward  (a good agent counts the adversaries touching it on the NEW positions, boundary penalty)
//   --- barrier ---
//   3. adversaries add the team's hits (10 per (good agent, adversary) pair in contact)
// Both kernels call these very functions, so their results agree bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "philox.h"

constexpr int kTagMaxAgents = 8, kTagMaxObst = 4;

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

// Element k of agent i's (zero-padded) observation -- the element-wise form of tag_obs for coalesced row writes.
__device__ __forceinline__ float tag_obs_elem(const TagCfg &c, const float *ap, const float *av, const float *lp, int i,
                                              int k) {
    const int NA = c.n_adv + c.n_good;
    if (k < 2) return av[2 * i + k];
    if (k < 4) return ap[2 * i + k - 2];
    k -= 4;
    if (k < 2 * c.n_obst) { const int l = k >> 1, x = k & 1; return lp[2 * l + x] - ap[2 * i + x]; }
    k -= 2 * c.n_obst;
    if (k < 2 * (NA - 1)) {
        int j = k >> 1;
        const int x = k & 1;
        if (j >= i) ++j;  // the other agents in increasing index
        return ap[2 * j + x] - ap[2 * i + x];
    }
    k -= 2 * (NA - 1);
    const int n_seen = c.n_good - (i >= c.n_adv ? 1 : 0);  // velocities of the good agents other than oneself
    if (k < 2 * n_seen) {
        int j = c.n_adv + (k >> 1);
        if (i >= c.n_adv && j >= i) ++j;
        return av[2 * j + (k & 1)];
    }
    return 0.f;  // padding of the good agents' rows
}

// The soft contact force entity j (agent or obstacle) exerts on agent i: false when the pair is out of range (the force is exactly
// 0 there).  Pair (lo, hi) = (min, max) of (i, j): d = p_lo - p_hi; f_lo += s d, f_hi -= s d -- returned as the term ADDED to f[i]
// (a - b == a + (-b) bit for bit).  One call per (i, j): the fused rollout spreads the calls over threads (rollout_tag.hip, phase D).
__device__ __forceinline__ bool tag_pair_force(const TagCfg &c, const float *ap, const float *lp, int i, int j, float &sx, float &sy) {
    const int NA = c.n_adv + c.n_good;
    const float my_size = i < c.n_adv ? c.adv_size : c.good_size;
    const float px = ap[2 * i], py = ap[2 * i + 1];
    const float qx = j < NA ? ap[2 * j] : lp[2 * (j - NA)], qy = j < NA ? ap[2 * j + 1] : lp[2 * (j - NA) + 1];
    const float sj = j < NA ? (j < c.n_adv ? c.adv_size : c.good_size) : c.obst_size;
    const float dx = i < j ? px - qx : qx - px, dy = i < j ? py - qy : qy - py;
    const float dmin = i < j ? my_size + sj : sj + my_size;
    const float d2 = dx * dx + dy * dy;
    const float far = dmin + 105.f * c.contact_margin;  // beyond it expf underflows: the force is exactly 0
    if (d2 > far * far) return false;
    const float dist = sqrtf(d2);
    const float pen = softplus_k(-(dist - dmin) / c.contact_margin, c.contact_margin);
    const float s = c.contact_force * pen / dist;
    const float tx = s * dx, ty = s * dy;
    sx = i < j ? tx : -tx;
    sy = i < j ? ty : -ty;
    return true;
}

__device__ __forceinline__ void tag_action_force(const TagCfg &c, int i, int a, float &fx, float &fy) {
    const float accel = i < c.n_adv ? c.adv_accel : c.good_accel;
    fx = (a == 1 ? -1.f : (a == 2 ? 1.f : 0.f)) * accel;
    fy = (a == 3 ? -1.f : (a == 4 ? 1.f : 0.f)) * accel;
}

// Damping, speed clamp and explicit Euler step of agent i under the total force (fx, fy).
__device__ __forceinline__ void tag_integrate(const TagCfg &c, int i, float px, float py, float vx, float vy, float fx, float fy,
                                              float &npx, float &npy, float &nvx, float &nvy) {
    nvx = vx * (1.f - c.damping) + fx * c.dt;
    nvy = vy * (1.f - c.damping) + fy * c.dt;
    const float vmax = i < c.n_adv ? c.adv_speed : c.good_speed;
    const float sp = sqrtf(nvx * nvx + nvy * nvy);
    if (sp > vmax) { nvx = nvx / sp * vmax; nvy = nvy / sp * vmax; }
    npx = px + nvx * c.dt;
    npy = py + nvy * c.dt;
}

// One step of agent i (action a): soft contact forces from every other entity in increasing entity index -- the order in
// which the serial pair loop (lo < hi) accumulates into f[i] -- then damping, speed clamp and integration.
__device__ __forceinline__ void tag_agent_move(const TagCfg &c, const float *ap, const float *av, const float *lp, int i,
                                               int a, float &npx, float &npy, float &nvx, float &nvy) {
    const int NE = c.n_adv + c.n_good + c.n_obst;
    float fx, fy;
    tag_action_force(c, i, a, fx, fy);
    for (int j = 0; j < NE; ++j) {  // wave-uniform trip count
        float sx, sy;
        if (j != i && tag_pair_force(c, ap, lp, i, j, sx, sy)) { fx += sx; fy += sy; }
    }
    tag_integrate(c, i, ap[2 * i], ap[2 * i + 1], av[2 * i], av[2 * i + 1], fx, fy, npx, npy, nvx, nvy);
}

// Own reward terms of agent i on the NEW positions (ap) -- a good agent: -10 per adversary touching it, minus the boundary
// penalty; `hit` = 10 * (adversaries touching it), which every adversary adds to its reward afterwards.  Adversaries: 0.
__device__ __forceinline__ float tag_own_reward(const TagCfg &c, const float *ap, int i, float npx, float npy, float &hit) {
    float my_rew = 0.f;
    hit = 0.f;
    if (i >= c.n_adv) {
        for (int a = 0; a < c.n_adv; ++a) {
            const float dx = ap[2 * a] - npx, dy = ap[2 * a + 1] - npy;
            if (sqrtf(dx * dx + dy * dy) < c.adv_size + c.good_size) { my_rew -= 10.f; hit += 10.f; }
        }
        my_rew -= bound_pen(fabsf(npx));
        my_rew -= bound_pen(fabsf(npy));
    }
    return my_rew;
}

// Re-initialisation of a finished env by its agent lanes: lane i draws agent i and obstacles i, i + NA, ...
// (Philox counter (episode * n_env + e) * 16 + entity, as tag_reset_env).
__device__ __forceinline__ void tag_reset_lane(const TagCfg &c, int e, uint64_t seed, uint64_t episode, int i, float *ap,
                                               float *av, float *lp) {
    const int NA = c.n_adv + c.n_good;
    uint32_t r4[4];
    const uint64_t base = (episode * (uint64_t)c.n_env + (uint64_t)e) * 16ull;
    tsm_philox4(seed, base + (uint64_t)i, r4);
    ap[2 * i] = uni(r4[0], -1.f, 1.f); ap[2 * i + 1] = uni(r4[1], -1.f, 1.f);
    av[2 * i] = 0.f; av[2 * i + 1] = 0.f;
    for (int l = i; l < c.n_obst; l += NA) {
        tsm_philox4(seed, base + (uint64_t)(NA + l), r4);
        lp[2 * l] = uni(r4[0], -0.9f, 0.9f); lp[2 * l + 1] = uni(r4[1], -0.9f, 0.9f);
    }
}
