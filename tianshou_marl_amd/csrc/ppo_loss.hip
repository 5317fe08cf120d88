// ppo_loss.hip -- PPO clip loss: per-minibatch advantage statistics, fused forward + backward
// w.r.t. logits and value, deterministic scalar reduction.
//
// Replaces the loss body of PPO._update_with_batch
// (/root/reference/tianshou/algorithm/modelfree/ppo.py:182-211): advantage normalisation with torch's
// unbiased std (:184-186), ratio/clip/dual-clip (:187-196), value(-clip) loss (:198-208), entropy
// (:210), total loss (:211).  Gradients follow torch autograd conventions (minimum/maximum split ties
// 50/50, clamp passes gradient on the closed interval) -- restated in oracle/oracle.c:orc_ppo_loss.
//
// HBM traffic per sample and gradient step (A = 5, no value clip): read logits 20 + act 4 +
// logp_old 4 + adv 4 + returns 4 + value 4, write dlogits 20 + dvalue 4 = 64 B.
#include "common.h"

namespace {

// ---- advantage statistics: one 1024-thread block per minibatch, two passes in f64 ----
// The gathered values of the first pass stay in registers (up to kStatRegs per thread = 8192 rows per minibatch), so
// the second pass costs no memory round trip; longer minibatches re-gather the tail.  Per-thread summation order is
// the plain strided order in both passes.
constexpr int kStatRegs = 8;

__global__ __launch_bounds__(1024) void adv_stats_kernel(const float *__restrict__ adv,
                                                         const int64_t *__restrict__ perm,
                                                         const int64_t *__restrict__ mb_start,
                                                         float *__restrict__ stats_out) {
    __shared__ double sm[1024 / 64];
    const int64_t s0 = mb_start[blockIdx.x], s1 = mb_start[blockIdx.x + 1];
    const int64_t M = s1 - s0;
    int64_t src[kStatRegs];
    float vals[kStatRegs];
#pragma unroll
    for (int k = 0; k < kStatRegs; ++k) {  // all row ids first, then all gathers: two round trips in total
        const int64_t i = s0 + threadIdx.x + 1024 * (int64_t)k;
        src[k] = i < s1 ? (perm ? perm[i] : i) : -1;
    }
#pragma unroll
    for (int k = 0; k < kStatRegs; ++k) vals[k] = src[k] >= 0 ? adv[src[k]] : 0.f;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < kStatRegs; ++k)
        if (src[k] >= 0) acc += (double)vals[k];
    for (int64_t i = s0 + threadIdx.x + 1024 * (int64_t)kStatRegs; i < s1; i += 1024) acc += (double)adv[perm ? perm[i] : i];
    const double mean = block_sum<double, 1024>(acc, sm) / (double)M;
    acc = 0.0;
#pragma unroll
    for (int k = 0; k < kStatRegs; ++k)
        if (src[k] >= 0) {
            const double d = (double)vals[k] - mean;
            acc += d * d;
        }
    for (int64_t i = s0 + threadIdx.x + 1024 * (int64_t)kStatRegs; i < s1; i += 1024) {
        const double d = (double)adv[perm ? perm[i] : i] - mean;
        acc += d * d;
    }
    const double ss = block_sum<double, 1024>(acc, sm);
    if (threadIdx.x == 0) {
        stats_out[2 * blockIdx.x + 0] = (float)mean;
        stats_out[2 * blockIdx.x + 1] = M > 1 ? (float)sqrt(ss / (double)(M - 1)) : NAN;  // torch.std
    }
}

// ---- the same statistics for LONG minibatches (> 8192 rows): chunks of a minibatch on separate workgroups ----
// One 1024-thread workgroup per 8192-row chunk writes {sum(x - c), sum((x - c)^2)} in f64, c = the minibatch's first
// gathered value (shifted data: no cancellation between the two sums); a second launch folds the chunks of each
// minibatch in chunk order.  Chunk boundaries depend on the minibatch bounds only, so results do not depend on the
// grid.  (The one-workgroup kernel above took 324 us for a single 819 200-row minibatch.)
constexpr int64_t kStatChunk = 1024 * kStatRegs;

__global__ __launch_bounds__(1024) void adv_stats_chunk_kernel(const float *__restrict__ adv,
                                                               const int64_t *__restrict__ perm,
                                                               const int64_t *__restrict__ mb_start, int32_t n_chunk,
                                                               double *__restrict__ work) {
    __shared__ double sm[1024 / 64];
    const int64_t s0 = mb_start[blockIdx.x], s1 = mb_start[blockIdx.x + 1];
    const int64_t c0 = s0 + (int64_t)blockIdx.y * kStatChunk;
    double *out = work + ((int64_t)blockIdx.x * n_chunk + blockIdx.y) * 2;
    if (c0 >= s1) {  // uniform per workgroup
        if (threadIdx.x == 0) { out[0] = 0.0; out[1] = 0.0; }
        return;
    }
    const int64_t c1 = c0 + kStatChunk < s1 ? c0 + kStatChunk : s1;
    const float shift = adv[perm ? perm[s0] : s0];
    int64_t src[kStatRegs];
#pragma unroll
    for (int k = 0; k < kStatRegs; ++k) {
        const int64_t i = c0 + threadIdx.x + 1024 * (int64_t)k;
        src[k] = i < c1 ? (perm ? perm[i] : i) : -1;
    }
    float vals[kStatRegs];
#pragma unroll
    for (int k = 0; k < kStatRegs; ++k) vals[k] = src[k] >= 0 ? adv[src[k]] : shift;
    double a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int k = 0; k < kStatRegs; ++k) {
        const double d = (double)vals[k] - (double)shift;
        a1 += d;
        a2 += d * d;
    }
    a1 = block_sum<double, 1024>(a1, sm);
    a2 = block_sum<double, 1024>(a2, sm);
    if (threadIdx.x == 0) { out[0] = a1; out[1] = a2; }
}

__global__ __launch_bounds__(64) void adv_stats_fold_kernel(const float *__restrict__ adv, const int64_t *__restrict__ perm,
                                                            const int64_t *__restrict__ mb_start, int32_t n_chunk,
                                                            const double *__restrict__ work, float *__restrict__ stats_out) {
    const int64_t s0 = mb_start[blockIdx.x], M = mb_start[blockIdx.x + 1] - s0;
    if (threadIdx.x != 0) return;
    double a1 = 0.0, a2 = 0.0;
    for (int c = 0; c < n_chunk; ++c) {
        a1 += work[((int64_t)blockIdx.x * n_chunk + c) * 2 + 0];
        a2 += work[((int64_t)blockIdx.x * n_chunk + c) * 2 + 1];
    }
    const double shift = M > 0 ? (double)adv[perm ? perm[s0] : s0] : 0.0;
    const double mean_s = M > 0 ? a1 / (double)M : 0.0;
    stats_out[2 * blockIdx.x + 0] = (float)(shift + mean_s);
    const double ss = a2 - a1 * mean_s;  // sum (x - mean)^2
    stats_out[2 * blockIdx.x + 1] = M > 1 ? (float)sqrt((ss > 0.0 ? ss : 0.0) / (double)(M - 1)) : NAN;
}

struct Cfg {
    float eps_clip, dual_clip, vf_coef, ent_coef;
    int value_clip, adv_norm, kind, vg;
};

constexpr int kLossThreads = 256;
constexpr int kMaxA = 64;
constexpr int64_t kMaxLossBlocks = 2048;  // 8 workgroups per CU: all of them resident at once

// Every workgroup walks the same number of 256-sample tiles (no tail round of a few workgroups on an otherwise idle chip).
inline int64_t loss_blocks(int64_t M) {
    const int64_t n = ceil_div(M, kLossThreads);
    if (n <= kMaxLossBlocks) return n;
    return ceil_div(n, ceil_div(n, kMaxLossBlocks));
}

template <int A_T>
__global__ __launch_bounds__(kLossThreads) void loss_kernel(
    const float *__restrict__ logits, const float *__restrict__ value, const int32_t *__restrict__ act,
    const float *__restrict__ logp_old, const float *__restrict__ adv, const float *__restrict__ returns,
    const float *__restrict__ v_s_old, const int64_t *__restrict__ perm, int64_t first_row, int64_t M,
    int32_t A_rt, const float *__restrict__ adv_stats, Cfg cfg, float *__restrict__ dlogits,
    float *__restrict__ dvalue, double *__restrict__ partial) {
    __shared__ double sm[kLossThreads / 64];
    const int A = A_T > 0 ? A_T : A_rt;
    double t_clip = 0.0, t_vf = 0.0, t_ent = 0.0;
    // value term of one sample (ppo.py:198-208): vf = loss term, g_v = d vf / d value
    auto value_terms = [&](int64_t row, float v, float &vf, float &g_v) {
        const float ret = returns[row];
        if (cfg.value_clip) {
            const float vs = v_s_old[row];
            const float d = v - vs;
            const float dc = fminf(fmaxf(d, -cfg.eps_clip), cfg.eps_clip);
            const bool v_in = d >= -cfg.eps_clip && d <= cfg.eps_clip;
            const float vclip = vs + dc;
            const float vf1 = (ret - v) * (ret - v), vf2 = (ret - vclip) * (ret - vclip);
            const float g1 = 2.f * (v - ret), g2 = v_in ? 2.f * (vclip - ret) : 0.f;
            if (vf1 > vf2) { vf = vf1; g_v = g1; }
            else if (vf1 < vf2) { vf = vf2; g_v = g2; }
            else { vf = vf1; g_v = 0.5f * g1 + 0.5f * g2; }
        } else {
            vf = (ret - v) * (ret - v);
            g_v = 2.f * (v - ret);
        }
    };
    // loss_kind 2: the value term alone (the policy terms of these samples are produced elsewhere, csrc/ppo_rows.hip)
    auto sample_value_only = [&](int64_t i, float v) -> float {
        const int64_t row = perm ? perm[i] : first_row + i;
        float vf, g_v;
        value_terms(row, v, vf, g_v);
        t_vf += vf;
        return cfg.vf_coef * g_v * (1.0f / (float)M);
    };
    // one sample: loss terms into the running sums, d loss / d logits stored, d loss / d value returned.  `v` is the
    // sample's value (value_group > 1: the value of its joint row, shared by the row's agents)
    auto sample_full = [&](int64_t i, float v) -> float {
        const int64_t row = perm ? perm[i] : first_row + i;
        const float invM = 1.0f / (float)M;
        float lg[A_T > 0 ? A_T : kMaxA];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < A; ++j) { lg[j] = logits[i * A + j]; m = fmaxf(m, lg[j]); }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < A; ++j) s += expf(lg[j] - m);
        const float lse = m + logf(s);
        const int a_idx = act[row];
        float a = adv[row];
        if (cfg.adv_norm) a = (a - adv_stats[0]) / (adv_stats[1] + 1e-8f);
        float logp = 0.f, h = 0.f;
        float pr[A_T > 0 ? A_T : kMaxA];  // p_j = exp(log-softmax_j): needed again for the gradient (one expf per action, not two)
#pragma unroll
        for (int j = 0; j < A; ++j) {
            const float l = lg[j] - lse;
            lg[j] = l;
            pr[j] = expf(l);
            h -= pr[j] * l;
            if (j == a_idx) logp = l;
        }
        float ratio, obj, g_ratio;  // d obj / d logp = g_ratio * ratio
        if (cfg.kind == 1) {  // plain policy gradient (a2c.py:263-264, reinforce.py:375-376): obj = logp * adv
            ratio = 1.f; obj = logp * a; g_ratio = a;
        } else {
            ratio = expf(logp - logp_old[row]);
            const float lo = 1.0f - cfg.eps_clip, hi = 1.0f + cfg.eps_clip;
            const float rc = fminf(fmaxf(ratio, lo), hi);
            const float s1 = ratio * a, s2 = rc * a;
            const bool in_range = ratio >= lo && ratio <= hi;
            if (s1 < s2) { obj = s1; g_ratio = a; }
            else if (s1 > s2) { obj = s2; g_ratio = in_range ? a : 0.f; }
            else { obj = s1; g_ratio = 0.5f * a + (in_range ? 0.5f * a : 0.f); }
            if (cfg.dual_clip > 0.f && a < 0.f) {
                const float c = cfg.dual_clip * a;
                if (c > obj) { obj = c; g_ratio = 0.f; }
                else if (c == obj) g_ratio *= 0.5f;
            }
        }
        float vf, g_v;
        value_terms(row, v, vf, g_v);
        const float g_logp = -g_ratio * ratio * invM;
        const float ec = cfg.ent_coef * invM;
#pragma unroll
        for (int j = 0; j < A; ++j) {
            const float l = lg[j], p = pr[j];
            const float dlogp = (j == a_idx ? 1.f : 0.f) - p;
            const float dent = -p * (l + h);
            dlogits[i * A + j] = g_logp * dlogp - ec * dent;
        }
        t_clip += obj; t_vf += vf; t_ent += h;
        return cfg.vf_coef * g_v * invM;
    };
    // a workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...: the grid is capped (kMaxLossBlocks) so that the
    // finalize pass folds a bounded number of partial sums, in a fixed order (deterministic)
    auto sample = [&](int64_t i, float v) -> float { return cfg.kind == 2 ? sample_value_only(i, v) : sample_full(i, v); };
    const int vg = cfg.vg;
    if (vg == 1) {
        for (int64_t i = (int64_t)blockIdx.x * kLossThreads + threadIdx.x; i < M; i += (int64_t)gridDim.x * kLossThreads)
            dvalue[i] = sample(i, value[i]);
    } else if ((vg & (vg - 1)) == 0 && vg <= 64) {
        // centralized critic: the vg samples of a joint row are adjacent lanes of one wave (vg | 64, M % vg == 0, so a
        // group is never split by the loop bound); d value of the row = xor-butterfly sum over its lanes (symmetric:
        // the same bits on every lane), stored by the row's first lane
        for (int64_t i = (int64_t)blockIdx.x * kLossThreads + threadIdx.x; i < M; i += (int64_t)gridDim.x * kLossThreads) {
            float dv = sample(i, value[i / vg]);
            for (int off = 1; off < vg; off <<= 1) dv += __shfl_xor(dv, off, 64);
            if (i % vg == 0) dvalue[i / vg] = dv;
        }
    } else {  // any group size: one thread walks the samples of a joint row
        const int64_t G = M / vg;
        for (int64_t gi = (int64_t)blockIdx.x * kLossThreads + threadIdx.x; gi < G; gi += (int64_t)gridDim.x * kLossThreads) {
            const float v = value[gi];
            float dv = 0.f;
            for (int a = 0; a < vg; ++a) dv += sample(gi * vg + a, v);
            dvalue[gi] = dv;
        }
    }
    const double b_clip = block_sum<double, kLossThreads>(t_clip, sm);
    const double b_vf = block_sum<double, kLossThreads>(t_vf, sm);
    const double b_ent = block_sum<double, kLossThreads>(t_ent, sm);
    if (threadIdx.x == 0) {
        partial[4 * blockIdx.x + 0] = b_clip;
        partial[4 * blockIdx.x + 1] = b_vf;
        partial[4 * blockIdx.x + 2] = b_ent;
        partial[4 * blockIdx.x + 3] = 0.0;
    }
}

__global__ __launch_bounds__(256) void finalize_kernel(const double *__restrict__ partial, int64_t n_blocks,
                                                       int64_t M, float vf_coef, float ent_coef,
                                                       float *__restrict__ scalars) {
    __shared__ double sm[256 / 64];
    double c = 0.0, v = 0.0, e = 0.0;
    for (int64_t b = threadIdx.x; b < n_blocks; b += 256) {
        c += partial[4 * b + 0];
        v += partial[4 * b + 1];
        e += partial[4 * b + 2];
    }
    c = block_sum<double, 256>(c, sm);
    v = block_sum<double, 256>(v, sm);
    e = block_sum<double, 256>(e, sm);
    if (threadIdx.x == 0) {
        const double clip_loss = -c / (double)M, vf_loss = v / (double)M, ent_loss = e / (double)M;
        scalars[0] = (float)(clip_loss + (double)vf_coef * vf_loss - (double)ent_coef * ent_loss);
        scalars[1] = (float)clip_loss;
        scalars[2] = (float)vf_loss;
        scalars[3] = (float)ent_loss;
    }
}

Cfg to_cfg(const tsm_ppo_cfg *c) {
    Cfg k;
    k.eps_clip = (float)c->eps_clip;
    k.dual_clip = (float)c->dual_clip;
    k.vf_coef = (float)c->vf_coef;
    k.ent_coef = (float)c->ent_coef;
    k.value_clip = c->value_clip;
    k.adv_norm = c->adv_norm;
    k.kind = c->loss_kind;
    k.vg = c->value_group > 1 ? c->value_group : 1;
    return k;
}

// Data-parallel replicas: (mean, unbiased std) of a rank's part of each minibatch <-> the additive form (n, sum x,
// sum x^2) in f64 that ONE all-reduce sums over the ranks.  Every operation is rounded on its own (no fma contraction),
// so the pair is bit-reproducible from plain f64 arithmetic in any language (tests/test_gpu_parallel.py).
__global__ void adv_stats_pack_kernel(const float *__restrict__ stats, const int64_t *__restrict__ mb_start, int32_t n_mb,
                                      double *__restrict__ pack) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_mb) return;
    const double n = (double)(mb_start[k + 1] - mb_start[k]);
    const double m = (double)stats[2 * k], sd = (double)stats[2 * k + 1];
    const double nm1 = n > 1.0 ? __dsub_rn(n, 1.0) : 0.0;
    pack[3 * k] = n;
    pack[3 * k + 1] = __dmul_rn(n, m);
    pack[3 * k + 2] = __dadd_rn(__dmul_rn(nm1, __dmul_rn(sd, sd)), __dmul_rn(__dmul_rn(n, m), m));
}

__global__ void adv_stats_unpack_kernel(const double *__restrict__ pack, int32_t n_mb, float *__restrict__ stats) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_mb) return;
    const double N = pack[3 * k], S1 = pack[3 * k + 1], S2 = pack[3 * k + 2];
    const double m = __ddiv_rn(S1, N);
    const double den = N > 1.0 ? __dsub_rn(N, 1.0) : 1.0;
    double var = __ddiv_rn(__dsub_rn(S2, __dmul_rn(__dmul_rn(N, m), m)), den);
    if (!(var > 0.0)) var = 0.0;
    stats[2 * k] = (float)m;
    stats[2 * k + 1] = (float)__dsqrt_rn(var);
}

}  // namespace

TSM_EXPORT int tsm_ppo_adv_stats(const float *adv, const int64_t *perm, const int64_t *mb_start,
                                 int32_t n_mb, float *stats_out, void *stream) {
    TSM_REQUIRE(n_mb >= 0, "tsm_ppo_adv_stats: negative n_mb");
    if (n_mb == 0) return TSM_OK;
    TSM_REQUIRE(adv && mb_start && stats_out, "tsm_ppo_adv_stats: null pointer");
    hipLaunchKernelGGL(adv_stats_kernel, dim3((unsigned)n_mb), dim3(1024), 0, tsm_stream(stream), adv, perm,
                       mb_start, stats_out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int64_t tsm_ppo_adv_stats_work_elems(int32_t n_mb, int64_t max_rows) {
    if (n_mb < 0 || max_rows < 0) return -1;
    return 2 * (int64_t)n_mb * ceil_div(max_rows > 0 ? max_rows : 1, kStatChunk);
}

TSM_EXPORT int tsm_ppo_adv_stats_wide(const float *adv, const int64_t *perm, const int64_t *mb_start, int32_t n_mb,
                                      int64_t max_rows, double *work, float *stats_out, void *stream) {
    TSM_REQUIRE(n_mb >= 0 && max_rows >= 0, "tsm_ppo_adv_stats_wide: negative size");
    if (n_mb == 0) return TSM_OK;
    TSM_REQUIRE(adv && mb_start && stats_out && work, "tsm_ppo_adv_stats_wide: null pointer");
    const int64_t n_chunk = ceil_div(max_rows > 0 ? max_rows : 1, kStatChunk);
    TSM_REQUIRE(n_chunk <= 65535, "tsm_ppo_adv_stats_wide: minibatch of %lld rows is too long", (long long)max_rows);
    hipLaunchKernelGGL(adv_stats_chunk_kernel, dim3((unsigned)n_mb, (unsigned)n_chunk), dim3(1024), 0, tsm_stream(stream),
                       adv, perm, mb_start, (int32_t)n_chunk, work);
    TSM_LAUNCH_CHECK();
    hipLaunchKernelGGL(adv_stats_fold_kernel, dim3((unsigned)n_mb), dim3(64), 0, tsm_stream(stream), adv, perm, mb_start,
                       (int32_t)n_chunk, work, stats_out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int tsm_ppo_adv_stats_pack(const float *stats, const int64_t *mb_start, int32_t n_mb, double *pack_out,
                                      void *stream) {
    TSM_REQUIRE(n_mb >= 0, "tsm_ppo_adv_stats_pack: negative n_mb");
    if (n_mb == 0) return TSM_OK;
    TSM_REQUIRE(stats && mb_start && pack_out, "tsm_ppo_adv_stats_pack: null pointer");
    hipLaunchKernelGGL(adv_stats_pack_kernel, dim3((unsigned)ceil_div(n_mb, 64)), dim3(64), 0, tsm_stream(stream), stats,
                       mb_start, n_mb, pack_out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int tsm_ppo_adv_stats_unpack(const double *pack, int32_t n_mb, float *stats_out, void *stream) {
    TSM_REQUIRE(n_mb >= 0, "tsm_ppo_adv_stats_unpack: negative n_mb");
    if (n_mb == 0) return TSM_OK;
    TSM_REQUIRE(pack && stats_out, "tsm_ppo_adv_stats_unpack: null pointer");
    hipLaunchKernelGGL(adv_stats_unpack_kernel, dim3((unsigned)ceil_div(n_mb, 64)), dim3(64), 0, tsm_stream(stream), pack,
                       n_mb, stats_out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int64_t tsm_ppo_loss_partial_elems(int64_t M) { return M <= 0 ? 0 : 4 * loss_blocks(M); }

TSM_EXPORT int tsm_ppo_loss_fwd_bwd(const float *logits, const float *value, const int32_t *act,
                                    const float *logp_old, const float *adv, const float *returns,
                                    const float *v_s_old, const int64_t *perm, int64_t first_row, int64_t M,
                                    int32_t A, const float *adv_stats, const tsm_ppo_cfg *cfg_host,
                                    float *dlogits_out, float *dvalue_out, double *partial_out, void *stream) {
    TSM_REQUIRE(M >= 0 && A >= 1 && A <= kMaxA, "tsm_ppo_loss_fwd_bwd: bad sizes M=%lld A=%d", (long long)M, A);
    if (M == 0) return TSM_OK;
    TSM_REQUIRE(cfg_host, "tsm_ppo_loss_fwd_bwd: null cfg");
    const bool value_only = cfg_host->loss_kind == 2;
    TSM_REQUIRE(value && returns && dvalue_out && partial_out, "tsm_ppo_loss_fwd_bwd: null pointer");
    TSM_REQUIRE(value_only || (logits && act && logp_old && adv && dlogits_out), "tsm_ppo_loss_fwd_bwd: null pointer");
    TSM_REQUIRE(!cfg_host->value_clip || v_s_old, "tsm_ppo_loss_fwd_bwd: value_clip needs v_s_old");
    TSM_REQUIRE(value_only || !cfg_host->adv_norm || adv_stats, "tsm_ppo_loss_fwd_bwd: adv_norm needs adv_stats");
    TSM_REQUIRE(cfg_host->loss_kind >= 0 && cfg_host->loss_kind <= 2, "tsm_ppo_loss_fwd_bwd: loss_kind must be 0, 1 or 2");
    TSM_REQUIRE(cfg_host->dual_clip <= 0.0 || cfg_host->dual_clip > 1.0,
                "Dual-clip PPO parameter should greater than 1.0 but got %g", cfg_host->dual_clip);  // ppo.py:124-126
    const Cfg cfg = to_cfg(cfg_host);
    TSM_REQUIRE(M % cfg.vg == 0, "tsm_ppo_loss_fwd_bwd: M = %lld is not a multiple of value_group = %d", (long long)M, cfg.vg);
    const dim3 grid((unsigned)loss_blocks(M)), block(kLossThreads);
    hipStream_t st = tsm_stream(stream);
#define LAUNCH(AT)                                                                                         \
    hipLaunchKernelGGL((loss_kernel<AT>), grid, block, 0, st, logits, value, act, logp_old, adv, returns,  \
                       v_s_old, perm, first_row, M, A, adv_stats, cfg, dlogits_out, dvalue_out, partial_out)
    if (A == 5) LAUNCH(5);
    else LAUNCH(0);
#undef LAUNCH
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int tsm_ppo_loss_finalize(const double *partial, int64_t M, const tsm_ppo_cfg *cfg_host,
                                     float *scalars_out, void *stream) {
    TSM_REQUIRE(M >= 1 && partial && cfg_host && scalars_out, "tsm_ppo_loss_finalize: bad args");
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, tsm_stream(stream), partial,
                       loss_blocks(M), M, (float)cfg_host->vf_coef, (float)cfg_host->ent_coef,
                       scalars_out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}
