// actor_rows_dev.h -- what the two one-launch ACTOR gradient-step kernels share (csrc/ppo_rows.hip: 32-sample tiles, all weights
// in LDS; csrc/actor_rows64.hip: 64-sample tiles, layer-2 weights in registers): the launch arguments and the loss head, so
// that both compute the PPO / policy-gradient objective of /root/reference/tianshou/algorithm/modelfree/ppo.py:183-196, 210
// (a2c.py:255-262 for loss_kind 1) with the same instructions.
#pragma once
#include "common.h"

struct TsmActorArgs {
    const float *P;          // actor parameters: w0[H][D] b0[H] w1[H][H] b1[H] w2[A][H] b2[A]
    const float *obs;        // [n][D]
    const int32_t *act;
    const float *logp_old, *adv;
    const int64_t *perm;     // sample ids of the minibatch (nullable: first_row + i)
    int64_t first_row, M;
    const float *adv_stats;  // {mean, std} of the minibatch (adv_norm)
    int D, A;
    float eps_clip, dual_clip, ent_coef;
    int adv_norm, kind;
    float *slabs;            // [grid][P]
    double *partial;         // [grid][4] = {sum clip objective, 0, sum entropy, 0}
    long long *stamps;       // diagnostics only (tsm_debug_set_stamps): phase time stamps of workgroup 0, its tiles 0..3
    int64_t *opt_step_dev;   // nullable: the device-resident optimizer step count, advanced by one per launch
};

// csrc/actor_rows64.hip
int tsm_actor_rows64_launch(const TsmActorArgs &g, int n_blocks, hipStream_t st);
int tsm_actor_rows64_init(void);

// The loss head of ONE sample on its 16 lanes (lane hj of the sample's DPP row holds logit hj, actions beyond A hold
// anything): log-softmax, entropy, the clip / dual-clip surrogate (kind 0) or the plain policy-gradient objective
// (kind 1), and d loss / d logit hj as the return value.  The exponentials of a sample run side by side and are folded in
// action order by DPP row operations -- the additions of a one-lane-per-sample loop, in its order.  `obj` / `ent`: the
// sample's clip objective and entropy (every lane of the row gets them; the caller lets the row's lane 0 accumulate).
__device__ __forceinline__ float tsm_actor_head(const TsmActorArgs &g, float logit, int hj, int lane, int a_idx, float a, float lpo,
                                                float adv_mean, float adv_std, float &obj, float &ent) {
    const int A = g.A;
    const bool on = hj < A;
    const float invM = 1.0f / (float)g.M;
    const float x = on ? logit : -INFINITY;
    const float m = row16_max(x);
    const float ex = on ? expf(x - m) : 0.f;
    float sum = 0.f;
    row_prefix_sum<0>(ex, A, sum);
    const float lse = m + logf(sum);
    const float l = on ? x - lse : 0.f;
    const float p = on ? expf(l) : 0.f;
    float h = 0.f;
    row_prefix_sub<0>(p * l, A, h);
    if (g.adv_norm) a = (a - adv_mean) / (adv_std + 1e-8f);
    const float logp = __shfl(l, (lane & 48) + a_idx, 64);
    float ratio, g_ratio;
    if (g.kind == 1) {
        ratio = 1.f; obj = logp * a; g_ratio = a;
    } else {
        ratio = expf(logp - lpo);
        const float lo = 1.0f - g.eps_clip, hi = 1.0f + g.eps_clip;
        const float rc = fminf(fmaxf(ratio, lo), hi);
        const float s1 = ratio * a, s2 = rc * a;
        const bool in_range = ratio >= lo && ratio <= hi;
        if (s1 < s2) { obj = s1; g_ratio = a; }
        else if (s1 > s2) { obj = s2; g_ratio = in_range ? a : 0.f; }
        else { obj = s1; g_ratio = 0.5f * a + (in_range ? 0.5f * a : 0.f); }
        if (g.dual_clip > 0.f && a < 0.f) {
            const float c = g.dual_clip * a;
            if (c > obj) { obj = c; g_ratio = 0.f; }
            else if (c == obj) g_ratio *= 0.5f;
        }
    }
    ent = h;
    const float g_logp = -g_ratio * ratio * invM;
    const float ec = g.ent_coef * invM;
    return on ? g_logp * ((hj == a_idx ? 1.f : 0.f) - p) + ec * p * (l + h) : 0.f;
}
