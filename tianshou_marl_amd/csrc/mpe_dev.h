// mpe_dev.h -- device-side simple_spread world step shared by the stand-alone env kernels (mpe.hip) and the
// persistent rollout kernel (rollout.hip).  Specification and provenance: see mpe.hip (restated from the
// published PettingZoo-MPE spec; pettingzoo sources are not in the reference tree -> parity unpinned).
//
// Everything is formulated per AGENT LANE (one lane = one (env, agent) row) on an env state block that lives
// in LDS: ap/av/lp = this env's [N][2] positions / velocities / landmark positions.  A world step is
//   1. mpe_agent_move      (reads the OLD state of all agents, returns the lane's new pos/vel in registers)
//   --- barrier; lanes write their new pos/vel; barrier ---
//   2. mpe_landmark_min_dist (lane i covers landmark i) and mpe_local_penalty (collisions of agent i)
//   --- barrier ---
//   3. mpe_reward           (folds the landmark terms in landmark order -- same order as a serial loop)
// Serial sections on a handful of lanes are what a latency-bound step cannot afford: one wave issues a
// dependent instruction only every ~8 cycles, so the N^2 pair work is spread over N lanes per env.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "philox.h"

constexpr int kMpeMaxN = 8;

struct MpeCfg {
    int n_env, N, obs_dim, max_cycles;
    float dt, damping, contact_force, contact_margin, agent_size, landmark_size, accel, max_speed, local_ratio;
};

__device__ __forceinline__ float mpe_uni_pm1(uint32_t bits) { return tsm_u01(bits) * 2.f - 1.f; }

// Reset of agent/landmark i of env e: positions ~ U(-1, 1)^2, velocity 0.
// Philox counter = (episode * n_env + e) * 8 + i.
__device__ __forceinline__ void mpe_reset_agent(const MpeCfg &c, int e, uint64_t seed, uint64_t episode, int i,
                                                float *ap, float *av, float *lp) {
    uint32_t r[4];
    tsm_philox4(seed, (episode * (uint64_t)c.n_env + (uint64_t)e) * 8ull + (uint64_t)i, r);
    ap[2 * i] = mpe_uni_pm1(r[0]); ap[2 * i + 1] = mpe_uni_pm1(r[1]);
    lp[2 * i] = mpe_uni_pm1(r[2]); lp[2 * i + 1] = mpe_uni_pm1(r[3]);
    av[2 * i] = 0.f; av[2 * i + 1] = 0.f;
}

// Element k of agent i's observation (6N): [vel(2), pos(2), landmarks rel (2N), others rel (2(N-1)), comm 0].
__device__ __forceinline__ float mpe_obs_elem(const MpeCfg &c, const float *ap, const float *av, const float *lp,
                                              int i, int k) {
    const int N = c.N;
    if (k < 2) return av[2 * i + k];
    if (k < 4) return ap[2 * i + k - 2];
    k -= 4;
    if (k < 2 * N) { const int l = k >> 1, x = k & 1; return lp[2 * l + x] - ap[2 * i + x]; }
    k -= 2 * N;
    if (k < 2 * (N - 1)) {
        int j = k >> 1;
        const int x = k & 1;
        if (j >= i) ++j;  // others in increasing index, skipping self
        return ap[2 * j + x] - ap[2 * i + x];
    }
    return 0.f;  // communication channel of silent agents
}

// All N agent positions of one env into registers, branch-free (index clamped to N - 1): the LDS reads issue back
// to back and cost one round trip instead of one per loop iteration; the loops over them then branch on the
// wave-uniform N, so only N iterations execute.
struct MpePos { float x[kMpeMaxN], y[kMpeMaxN]; };
__device__ __forceinline__ MpePos mpe_load_pos(const MpeCfg &c, const float *ap) {
    MpePos p;
#pragma unroll
    for (int j = 0; j < kMpeMaxN; ++j) {
        const int jj = j < c.N ? j : c.N - 1;
        p.x[j] = ap[2 * jj]; p.y[j] = ap[2 * jj + 1];
    }
    return p;
}

// Soft contact force agent i receives from agent j (j != i), signed as the serial pair loop adds it: seen from the lower
// index (a, b) = (min, max): f_a += s*d, f_b -= s*d with d = p_a - p_b.  Returns false when the pair is far apart (the
// common case): z < -104, so expf(z) == 0 exactly in f32 and the force is exactly 0 -- the sqrt / exp / log1p / divide
// chain is skipped (the bound carries a margin of one contact_margin).
__device__ __forceinline__ bool mpe_pair_force(const MpeCfg &c, float px, float py, float qx, float qy, int i, int j,
                                               float &sx, float &sy) {
    const float k = c.contact_margin;
    const float far = 2.f * c.agent_size + 105.f * k;
    const float dx = i < j ? px - qx : qx - px, dy = i < j ? py - qy : qy - py;
    const float d2 = dx * dx + dy * dy;
    if (d2 > far * far) return false;
    const float dist = sqrtf(d2);
    const float z = -(dist - 2.f * c.agent_size) / k;
    const float pen = (z > 0.f ? z + log1pf(expf(-z)) : log1pf(expf(z))) * k;  // logaddexp(0, z) * k
    const float s = c.contact_force * pen / dist;
    const float tx = s * dx, ty = s * dy;
    sx = i < j ? tx : -tx;
    sy = i < j ? ty : -ty;
    return true;
}

// Damping, optional speed clamp and explicit Euler step of one agent under the total force (fx, fy).
__device__ __forceinline__ void mpe_integrate(const MpeCfg &c, float px, float py, float vx, float vy, float fx, float fy,
                                              float &npx, float &npy, float &nvx, float &nvy) {
    nvx = vx * (1.f - c.damping) + fx * c.dt;
    nvy = vy * (1.f - c.damping) + fy * c.dt;
    if (c.max_speed > 0.f) {
        const float sp = sqrtf(nvx * nvx + nvy * nvy);
        if (sp > c.max_speed) { nvx = nvx / sp * c.max_speed; nvy = nvy / sp * c.max_speed; }
    }
    npx = px + nvx * c.dt;
    npy = py + nvy * c.dt;
}

__device__ __forceinline__ float mpe_action_force(const MpeCfg &c, int act, int axis) {
    return axis == 0 ? (act == 1 ? -1.f : (act == 2 ? 1.f : 0.f)) * c.accel : (act == 3 ? -1.f : (act == 4 ? 1.f : 0.f)) * c.accel;
}

// New position / velocity of agent i after one step (action force + soft contact forces, summed over the other
// agents in increasing index; damping; optional speed clamp; explicit Euler).  rollout_rows.hip evaluates the pair forces
// on other lanes (mpe_pair_force) and folds them in the same order: the same bits.
__device__ __forceinline__ void mpe_agent_move(const MpeCfg &c, const float *ap, const float *av, int i, int act,
                                               float &npx, float &npy, float &nvx, float &nvy) {
    const MpePos p = mpe_load_pos(c, ap);
    const float px = ap[2 * i], py = ap[2 * i + 1];
    const float vx = av[2 * i], vy = av[2 * i + 1];
    float fx = mpe_action_force(c, act, 0);
    float fy = mpe_action_force(c, act, 1);
#pragma unroll
    for (int j = 0; j < kMpeMaxN; ++j) {
        if (j >= c.N) break;  // wave-uniform
        float sx, sy;
        if (j != i && mpe_pair_force(c, px, py, p.x[j], p.y[j], i, j, sx, sy)) { fx += sx; fy += sy; }
    }
    mpe_integrate(c, px, py, vx, vy, fx, fy, npx, npy, nvx, nvy);
}

// min over agents of the distance to landmark l (on the NEW positions).  Correctly rounded sqrt is monotone, so
// min_i sqrt(d2_i) == sqrt(min_i d2_i) bit for bit: one sqrt instead of N.
__device__ __forceinline__ float mpe_landmark_min_dist(const MpeCfg &c, const MpePos &p, const float *lp, int l) {
    float m2 = INFINITY;
    const float lx = lp[2 * l], ly = lp[2 * l + 1];
#pragma unroll
    for (int i = 0; i < kMpeMaxN; ++i) {
        if (i >= c.N) break;  // wave-uniform
        const float dx = p.x[i] - lx, dy = p.y[i] - ly;
        m2 = fminf(m2, dx * dx + dy * dy);
    }
    return sqrtf(m2);
}

// -1 per other agent that agent i collides with (on the NEW positions): sqrt(d2) < 2 * agent_size.  The sqrt is
// only evaluated inside a +-1 % band around the threshold; outside it the squared comparison decides identically.
__device__ __forceinline__ float mpe_local_penalty(const MpeCfg &c, const MpePos &p, const float *ap, int i) {
    float local = 0.f;
    const float thr = 2.f * c.agent_size;
    const float lo2 = (0.99f * thr) * (0.99f * thr), hi2 = (1.01f * thr) * (1.01f * thr);
    const float px = ap[2 * i], py = ap[2 * i + 1];
#pragma unroll
    for (int j = 0; j < kMpeMaxN; ++j) {
        if (j >= c.N) break;  // wave-uniform
        const float dx = px - p.x[j], dy = py - p.y[j];
        const float d2 = dx * dx + dy * dy;
        bool hit = d2 < lo2;
        if (!hit && d2 <= hi2) hit = sqrtf(d2) < thr;
        if (j != i && hit) local -= 1.f;
    }
    return local;
}

// reward of an agent from the per-landmark minima m[0..N) of its env (folded in landmark order) and its penalty
__device__ __forceinline__ float mpe_reward(const MpeCfg &c, const float *m, float local) {
    float global = 0.f;
    for (int l = 0; l < c.N; ++l) global -= m[l];
    return global * (1.f - c.local_ratio) + local * c.local_ratio;
}
// ... the same fold on minima held in registers (gathered from other lanes)
__device__ __forceinline__ float mpe_reward_regs(const MpeCfg &c, const float (&m)[kMpeMaxN], float local) {
    float global = 0.f;
#pragma unroll
    for (int l = 0; l < kMpeMaxN; ++l)
        if (l < c.N) global -= m[l];
    return global * (1.f - c.local_ratio) + local * c.local_ratio;
}
