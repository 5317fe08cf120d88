// gae.hip -- fused value-mask + end-flag + delta + reverse scan + returns over (env, agent) lanes.
//
// Replaces Algorithm.compute_episodic_return / value_mask / numba `_gae`
// (/root/reference/tianshou/algorithm/algorithm_base.py:631-717,1079-1134) and the return_scaling
// arithmetic of a2c.py:132-146 for the time-major lane layout [T][n_lane].
//
// Parallelisation (gfx950): a workgroup is 64 lanes x W time-chunks (W waves).  A wave owns 64
// adjacent lanes, so every per-step load of a wave is one contiguous 256-B segment (coalesced).
// Time is cut into super-chunks of W*CH steps, walked from the end of the series; inside a
// super-chunk each wave scans its CH steps with carry 0 and publishes the affine map
// carry -> gae_at_chunk_start = B + P*carry (the linear-recurrence composition) in LDS; after one
// barrier every wave folds the maps of the later chunks to get its carry-in and replays its CH
// steps serially.  All arithmetic is f64 (the reference accumulates in f64); outputs are rounded
// to f32 once (a2c.py:149-150).
//
// HBM traffic per (lane, step): v_s 4 + v_s_next 4 + rew 4 + terminated 1 + truncated 1 read,
// adv 4 + returns 4 written = 22 B (flags_per_lane) -- the algorithmic bytes of SURVEY.md 8(d).
#include "common.h"

namespace {

// debug / tuning overrides (tools/bench_kernels.py --gae-sweep): 0 = automatic choice
int g_force_vec = 0, g_force_ch = 0, g_force_w = 0;

template <int V> __device__ __forceinline__ void ld_f(const float *__restrict__ p, float (&o)[V]) {
    if constexpr (V == 4) { const float4 q = *reinterpret_cast<const float4 *>(p); o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w; }
    else if constexpr (V == 2) { const float2 q = *reinterpret_cast<const float2 *>(p); o[0] = q.x; o[1] = q.y; }
    else o[0] = p[0];
}
template <int V> __device__ __forceinline__ void st_f(float *__restrict__ p, const float (&v)[V]) {
    if constexpr (V == 4) *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    else if constexpr (V == 2) *reinterpret_cast<float2 *>(p) = make_float2(v[0], v[1]);
    else p[0] = v[0];
}
template <int V> __device__ __forceinline__ void ld_b(const uint8_t *__restrict__ p, uint8_t (&o)[V]) {
    if constexpr (V == 4) { const uint32_t q = *reinterpret_cast<const uint32_t *>(p);
                            o[0] = q & 0xff; o[1] = (q >> 8) & 0xff; o[2] = (q >> 16) & 0xff; o[3] = q >> 24; }
    else if constexpr (V == 2) { const uint16_t q = *reinterpret_cast<const uint16_t *>(p); o[0] = q & 0xff; o[1] = q >> 8; }
    else o[0] = p[0];
}

// A thread owns VEC adjacent lanes (one 4*VEC-byte load per array and step) and CH consecutive steps of a
// super-chunk.  VEC > 1 requires !GENERIC, FLAGS_PER_LANE, L % VEC == 0 and suitably aligned pointers.
template <bool GENERIC, bool FLAGS_PER_LANE, int VEC, int CH>
__global__ __launch_bounds__(1024) void gae_lanes_kernel(
    const float *__restrict__ v_s, const float *__restrict__ v_n, const float *__restrict__ rew,
    const uint8_t *__restrict__ term, const uint8_t *__restrict__ trunc, int64_t T, int64_t L,
    int64_t lanes_per_env, const int32_t *__restrict__ env_start,
    const int32_t *__restrict__ env_len, double gamma, double gl, double v_scale_arg,
    const double *__restrict__ rms, double rms_eps, float *__restrict__ ret_out, float *__restrict__ adv_out) {
    extern __shared__ double lds[];  // [2][W][64 * VEC][2]
    const int W = blockDim.y;
    const int lx = threadIdx.x, w = threadIdx.y;
    const int64_t lane = ((int64_t)blockIdx.x * 64 + lx) * VEC;
    const bool live = lane < L;
    const int64_t env = live ? lane / lanes_per_env : 0;
    const int64_t n_env = L / lanes_per_env;
    int64_t len = T, start = 0;
    if (GENERIC && live) {
        if (env_len) len = env_len[env];
        if (env_start) start = env_start[env];
    }
    const int64_t SC = (int64_t)W * CH;
    const int64_t n_sc = (T + SC - 1) / SC;  // uniform trip count keeps the barrier structure uniform
    // return_scaling with device-resident running statistics (a2c.py:132-134): scale = sqrt(var + eps), read from HBM
    // so that a captured graph follows the statistics from update to update
    const double v_scale = rms ? sqrt(rms[1] + rms_eps) : v_scale_arg;
    const double inv_scale = 1.0 / v_scale;

    double carry_super[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) carry_super[j] = 0.0;
    for (int64_t s = n_sc - 1; s >= 0; --s) {
        const int64_t k0 = s * SC + (int64_t)w * CH;
        float a_f[CH][VEC], b_f[CH][VEC], r_f[CH][VEC];
        uint8_t te_b[CH][VEC], tr_b[CH][VEC];
        unsigned valid = 0;
        // issue every load of the chunk before the first use (memory-level parallelism)
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int64_t kk = k0 + k;
            if (live && kk < len) {
                int64_t slot = kk;
                if (GENERIC) { slot = start + kk; if (slot >= T) slot -= T; }
                const int64_t i = slot * L + lane;
                ld_f<VEC>(v_s + i, a_f[k]);
                ld_f<VEC>(v_n + i, b_f[k]);
                ld_f<VEC>(rew + i, r_f[k]);
                if (FLAGS_PER_LANE) { ld_b<VEC>(term + i, te_b[k]); ld_b<VEC>(trunc + i, tr_b[k]); }
                else { te_b[k][0] = term[slot * n_env + env]; tr_b[k][0] = trunc[slot * n_env + env]; }
                valid |= 1u << k;
            }
        }
        double delta[CH][VEC];
        unsigned keep[VEC];  // bit k set: discount = gl (no end flag)
#pragma unroll
        for (int j = 0; j < VEC; ++j) keep[j] = 0;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const bool ok = (valid >> k) & 1u;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                delta[k][j] = 0.0;
                if (ok) {
                    const uint8_t te = te_b[k][j], tr = tr_b[k][j];
                    const double vs = (double)a_f[k][j] * v_scale;
                    const double vn = te ? 0.0 : (double)b_f[k][j] * v_scale;
                    delta[k][j] = (double)r_f[k][j] + vn * gamma - vs;
                    const bool end = te | tr | (k0 + k == len - 1);
                    keep[j] |= (end ? 0u : 1u) << k;
                }
            }
        }
        // local scan with carry 0 -> affine map carry -> B + P * carry  (masked step: identity)
        double P[VEC], B[VEC];
        double *buf = lds + (size_t)(s & 1) * W * 64 * VEC * 2;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            P[j] = 1.0; B[j] = 0.0;
#pragma unroll
            for (int k = CH - 1; k >= 0; --k) {
                const bool v = (valid >> k) & 1u;
                const double d = v ? (((keep[j] >> k) & 1u) ? gl : 0.0) : 1.0;
                B[j] = delta[k][j] + d * B[j];
                P[j] *= d;
            }
            buf[(((size_t)w * 64 + lx) * VEC + j) * 2 + 0] = P[j];
            buf[(((size_t)w * 64 + lx) * VEC + j) * 2 + 1] = B[j];
        }
        __syncthreads();
        // fold the later chunks (W-1 .. w+1) onto the super-chunk carry; continue down to chunk 0 for the next carry
        double c[VEC], cs[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            double acc = carry_super[j];
            for (int ww = W - 1; ww > w; --ww) {
                const double Pw = buf[(((size_t)ww * 64 + lx) * VEC + j) * 2 + 0];
                const double Bw = buf[(((size_t)ww * 64 + lx) * VEC + j) * 2 + 1];
                acc = Bw + Pw * acc;
            }
            c[j] = acc;
            acc = B[j] + P[j] * acc;  // own chunk
            for (int ww = w - 1; ww >= 0; --ww) {
                const double Pw = buf[(((size_t)ww * 64 + lx) * VEC + j) * 2 + 0];
                const double Bw = buf[(((size_t)ww * 64 + lx) * VEC + j) * 2 + 1];
                acc = Bw + Pw * acc;
            }
            cs[j] = acc;
        }
        // replay own chunk serially from the true carry-in
        double g[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) g[j] = c[j];
#pragma unroll
        for (int k = CH - 1; k >= 0; --k) {
            if ((valid >> k) & 1u) {
                float adv_v[VEC], ret_v[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const double d = ((keep[j] >> k) & 1u) ? gl : 0.0;
                    g[j] = delta[k][j] + d * g[j];
                    adv_v[j] = (float)g[j];
                    ret_v[j] = (float)((g[j] + (double)a_f[k][j] * v_scale) * inv_scale);
                }
                const int64_t kk = k0 + k;
                int64_t slot = kk;
                if (GENERIC) { slot = start + kk; if (slot >= T) slot -= T; }
                const int64_t i = slot * L + lane;
                st_f<VEC>(adv_out + i, adv_v);
                st_f<VEC>(ret_out + i, ret_v);
            }
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) carry_super[j] = cs[j];
        // LDS is double-buffered by (s & 1): the next iteration writes the other half, and the
        // barrier of that iteration orders it against this iteration's reads of this half.
    }
}

__global__ void mc_return_kernel(const float *__restrict__ rew, int64_t T, int64_t L, double gamma,
                                 float *__restrict__ out) {
    const int64_t lane = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= L) return;
    double g = 0.0;
    for (int64_t t = T - 1; t >= 0; --t) {
        g = (double)rew[t * L + lane] + (t == T - 1 ? 0.0 : gamma * g);
        out[t * L + lane] = (float)g;
    }
}

// ---- RunningMeanStd.update on the unnormalised returns (a2c.py:144-146, utils/statistics.py:97-114) ----
// level 1: one workgroup per 8192 elements -> {sum x, sum x^2} f64 (x = returns * scale, scale from the OLD var);
// level 2: one workgroup folds the partials in index order and merges the batch moments into rms in place.
constexpr int kRmsThreads = 1024, kRmsPer = 8;
constexpr int64_t kRmsChunk = (int64_t)kRmsThreads * kRmsPer;

__global__ __launch_bounds__(kRmsThreads) void rms_partial_kernel(const float *__restrict__ x, const int64_t *__restrict__ ids,
                                                                  int64_t n, const double *__restrict__ rms, double eps,
                                                                  double *__restrict__ work) {
    __shared__ double sm[kRmsThreads / 64];
    const double scale = sqrt(rms[1] + eps);
    const int64_t base = (int64_t)blockIdx.x * kRmsChunk;
    double a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int k = 0; k < kRmsPer; ++k) {
        const int64_t i = base + threadIdx.x + (int64_t)kRmsThreads * k;
        if (i < n) {
            const double v = (double)x[ids ? ids[i] : i] * scale;
            a1 += v;
            a2 += v * v;
        }
    }
    a1 = block_sum<double, kRmsThreads>(a1, sm);
    a2 = block_sum<double, kRmsThreads>(a2, sm);
    if (threadIdx.x == 0) { work[2 * blockIdx.x] = a1; work[2 * blockIdx.x + 1] = a2; }
}

__global__ __launch_bounds__(64) void rms_merge_kernel(const double *__restrict__ work, int64_t n_part, int64_t n,
                                                       double *__restrict__ rms) {
    if (threadIdx.x != 0) return;
    double a1 = 0.0, a2 = 0.0;
    for (int64_t b = 0; b < n_part; ++b) { a1 += work[2 * b]; a2 += work[2 * b + 1]; }
    const double bm = a1 / (double)n;
    double bv = a2 / (double)n - bm * bm;  // np.var (population)
    if (bv < 0.0) bv = 0.0;
    const double mean = rms[0], var = rms[1], count = rms[2];
    const double delta = bm - mean, total = count + (double)n;
    const double m2 = var * count + bv * (double)n + delta * delta * count * (double)n / total;
    rms[0] = mean + delta * (double)n / total;
    rms[1] = m2 / total;
    rms[2] = total;
}

// ---- few lanes, long series: TIME is the parallel axis ------------------------------------------------------------------
// The MARL trainers hand PPO.learn ONE time-ordered lane of n_env * T rows (training_coordinator.py:118,154,336: a per-agent
// Batch; episode ends are flags inside the lane).  gae_lanes_kernel walks such a lane in super-chunks of W * CH = 64 steps,
// each behind its own round of loads: 200 dependent round trips for 12 800 rows (measured 682 us).  Here a workgroup owns
// one lane and its 1024 threads each own CH consecutive steps of a super-chunk of 1024 * CH steps; the chunks' affine maps
// (carry -> B + P * carry) are combined by a parallel suffix scan in LDS (10 doubling steps), every thread then replays its
// chunk from its true carry-in.  Same f64 recurrence per step; the maps are composed in a different association order than
// in gae_lanes_kernel (differences at the 1e-16 level, before the one rounding to f32).
constexpr int kScanMaxWg = 128;   // workgroups of the parallel form (all resident at once)

template <bool FLAGS_PER_LANE, int CH>
__global__ __launch_bounds__(1024) void gae_long_kernel(
    const float *__restrict__ v_s, const float *__restrict__ v_n, const float *__restrict__ rew,
    const uint8_t *__restrict__ term, const uint8_t *__restrict__ trunc, int64_t T, int64_t L, int64_t lanes_per_env,
    double gamma, double gl, double v_scale_arg, const double *__restrict__ rms, double rms_eps,
    float *__restrict__ ret_out, float *__restrict__ adv_out) {
    __shared__ double sP[2][1024], sB[2][1024];
    const int t = threadIdx.x;
    const int64_t lane = blockIdx.x;
    const int64_t env = lane / lanes_per_env, n_env = L / lanes_per_env;
    const double v_scale = rms ? sqrt(rms[1] + rms_eps) : v_scale_arg;
    const double inv_scale = 1.0 / v_scale;
    const int64_t SC = (int64_t)1024 * CH;
    const int64_t n_sc = (T + SC - 1) / SC;
    double carry_super = 0.0;
    for (int64_t s = n_sc - 1; s >= 0; --s) {
        const int64_t k0 = s * SC + (int64_t)t * CH;
        double delta[CH], vs_[CH];
        unsigned keep = 0, valid = 0;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int64_t kk = k0 + k;
            delta[k] = 0.0; vs_[k] = 0.0;
            if (kk < T) {
                const int64_t i = kk * L + lane;
                const uint8_t te = FLAGS_PER_LANE ? term[i] : term[kk * n_env + env];
                const uint8_t tr = FLAGS_PER_LANE ? trunc[i] : trunc[kk * n_env + env];
                const double vs = (double)v_s[i] * v_scale;
                const double vn = te ? 0.0 : (double)v_n[i] * v_scale;
                delta[k] = (double)rew[i] + vn * gamma - vs;
                vs_[k] = vs;
                const bool end = te | tr | (kk == T - 1);
                keep |= (end ? 0u : 1u) << k;
                valid |= 1u << k;
            }
        }
        // own chunk with carry 0 -> affine map (masked step: identity)
        double P = 1.0, B = 0.0;
#pragma unroll
        for (int k = CH - 1; k >= 0; --k) {
            const double d = ((valid >> k) & 1u) ? (((keep >> k) & 1u) ? gl : 0.0) : 1.0;
            B = delta[k] + d * B;
            P *= d;
        }
        // inclusive suffix scan over the chunks: S[t] = map of chunks t .. 1023 (the later chunk is applied first)
        int cur = 0;
        sP[0][t] = P; sB[0][t] = B;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            double p = sP[cur][t], b = sB[cur][t];
            if (t + off < 1024) {
                const double p2 = sP[cur][t + off], b2 = sB[cur][t + off];
                b = b + p * b2;
                p = p * p2;
            }
            sP[cur ^ 1][t] = p; sB[cur ^ 1][t] = b;
            cur ^= 1;
            __syncthreads();
        }
        // carry-in of this chunk = the later chunks of the super-chunk applied to the carry of the later super-chunks
        double g = carry_super;
        if (t + 1 < 1024) g = sB[cur][t + 1] + sP[cur][t + 1] * carry_super;
        const double next_super = sB[cur][0] + sP[cur][0] * carry_super;
#pragma unroll
        for (int k = CH - 1; k >= 0; --k) {
            if ((valid >> k) & 1u) {
                const double d = ((keep >> k) & 1u) ? gl : 0.0;
                g = delta[k] + d * g;
                const int64_t i = (k0 + k) * L + lane;
                adv_out[i] = (float)g;
                ret_out[i] = (float)((g + vs_[k]) * inv_scale);
            }
        }
        carry_super = next_super;
        __syncthreads();  // the LDS maps are rewritten by the next (earlier) super-chunk
    }
}

// The same scan with the super-chunks of a lane on DIFFERENT workgroups (grid n_lane x n_sc): the affine map of a whole
// super-chunk (carry -> B + P * carry) does not depend on the other super-chunks, so every workgroup forms its own, publishes it,
// waits for the maps of the LATER super-chunks of its lane and folds them in the sequential kernel's order -- the same additions
// and products, the same bits -- instead of one workgroup walking the super-chunks one after the other (12 800 rows: 27 -> 9 us).
// A workgroup only ever waits for workgroups with a SMALLER blockIdx.y (super-chunk s = n_sc - 1 - blockIdx.y: the later
// super-chunks are dispatched first), so a waiter never holds a CU that the workgroup it waits for still needs -- the form does not
// rely on all n_lane * n_sc workgroups being resident at once (the host still keeps it to <= 128 workgroups).  Hand-over through
// ONE SLOT of a device workspace (tsm_gae_set_scan_workspace; a slot per eager stream and per captured launch, so two launches in
// flight never share one): word 0 = generation g of the launch, word 1 = finished workgroups; a map is published as two
// agent-scope atomic f64 stores followed by a release store of g + 1 into its flag, read by agent-scope atomic loads (the L2s of
// different XCDs are not coherent for plain loads); the last workgroup to finish advances the generation, so graph replays work.
// A wait that runs out (never seen) poisons the carry with NaN AND sets the host-visible error word, which the host binding reads
// with the loss statistics (ops.gae_scan_failed).
struct ScanWs { unsigned gen, done; unsigned flag[kScanMaxWg]; double agg[kScanMaxWg][2]; };

template <bool FLAGS_PER_LANE>
__global__ __launch_bounds__(1024) void gae_long_par_kernel(
    const float *__restrict__ v_s, const float *__restrict__ v_n, const float *__restrict__ rew,
    const uint8_t *__restrict__ term, const uint8_t *__restrict__ trunc, int64_t T, int64_t L, int64_t lanes_per_env,
    double gamma, double gl, double v_scale_arg, const double *__restrict__ rms, double rms_eps,
    float *__restrict__ ret_out, float *__restrict__ adv_out, ScanWs *__restrict__ ws, int32_t *__restrict__ err_host) {
    constexpr int CH = 4;
    __shared__ double sP[2][1024], sB[2][1024];
    __shared__ double s_carry;
    const int t = threadIdx.x;
    const int64_t lane = blockIdx.x;
    const int n_sc = (int)gridDim.y, s = n_sc - 1 - (int)blockIdx.y;   // waited-for super-chunks (s2 > s) are dispatched first
    const int64_t env = lane / lanes_per_env, n_env = L / lanes_per_env;
    const double v_scale = rms ? sqrt(rms[1] + rms_eps) : v_scale_arg;
    const double inv_scale = 1.0 / v_scale;
    const unsigned gen = __hip_atomic_load(&ws->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int64_t k0 = (int64_t)s * 1024 * CH + (int64_t)t * CH;
    double delta[CH], vs_[CH];
    unsigned keep = 0, valid = 0;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int64_t kk = k0 + k;
        delta[k] = 0.0; vs_[k] = 0.0;
        if (kk < T) {
            const int64_t i = kk * L + lane;
            const uint8_t te = FLAGS_PER_LANE ? term[i] : term[kk * n_env + env];
            const uint8_t tr = FLAGS_PER_LANE ? trunc[i] : trunc[kk * n_env + env];
            const double vs = (double)v_s[i] * v_scale;
            const double vn = te ? 0.0 : (double)v_n[i] * v_scale;
            delta[k] = (double)rew[i] + vn * gamma - vs;
            vs_[k] = vs;
            const bool end = te | tr | (kk == T - 1);
            keep |= (end ? 0u : 1u) << k;
            valid |= 1u << k;
        }
    }
    double P = 1.0, B = 0.0;
#pragma unroll
    for (int k = CH - 1; k >= 0; --k) {
        const double d = ((valid >> k) & 1u) ? (((keep >> k) & 1u) ? gl : 0.0) : 1.0;
        B = delta[k] + d * B;
        P *= d;
    }
    int cur = 0;
    sP[0][t] = P; sB[0][t] = B;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        double p = sP[cur][t], b = sB[cur][t];
        if (t + off < 1024) {
            const double p2 = sP[cur][t + off], b2 = sB[cur][t + off];
            b = b + p * b2;
            p = p * p2;
        }
        sP[cur ^ 1][t] = p; sB[cur ^ 1][t] = b;
        cur ^= 1;
        __syncthreads();
    }
    if (t == 0) {
        const int me = (int)lane * n_sc + s;
        __hip_atomic_store(&ws->agg[me][0], sP[cur][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ws->agg[me][1], sB[cur][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ws->flag[me], gen + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        // carry into this super-chunk: the later ones folded from the last one down (the sequential kernel's recurrence)
        double carry = 0.0;
        for (int s2 = n_sc - 1; s2 > s; --s2) {
            const int o = (int)lane * n_sc + s2;
            int spins = 0;
            while (__hip_atomic_load(&ws->flag[o], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != gen + 1u) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 22)) {   // (never seen; a lost workgroup must not hang the device)
                    carry = NAN;
                    if (err_host) __hip_atomic_store(err_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
            const double p2 = __hip_atomic_load(&ws->agg[o][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double b2 = __hip_atomic_load(&ws->agg[o][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            carry = b2 + p2 * carry;
        }
        s_carry = carry;
    }
    __syncthreads();
    const double carry_super = s_carry;
    double g = carry_super;
    if (t + 1 < 1024) g = sB[cur][t + 1] + sP[cur][t + 1] * carry_super;
#pragma unroll
    for (int k = CH - 1; k >= 0; --k) {
        if ((valid >> k) & 1u) {
            const double d = ((keep >> k) & 1u) ? gl : 0.0;
            g = delta[k] + d * g;
            const int64_t i = (k0 + k) * L + lane;
            adv_out[i] = (float)g;
            ret_out[i] = (float)((g + vs_[k]) * inv_scale);
        }
    }
    if (t == 0) {   // the last workgroup to finish opens the next generation (every workgroup has read its flags by now)
        const unsigned total = gridDim.x * gridDim.y;
        if (__hip_atomic_fetch_add(&ws->done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == total - 1) {
            __hip_atomic_store(&ws->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ws->gen, gen + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int pick_waves(int64_t T, int64_t L, int vec, int ch) {
    // enough waves to fill 256 CUs x 8, but no more chunks than the series has
    int64_t blocks = ceil_div(L, 64 * vec);
    int64_t want = ceil_div(2048, blocks);
    int64_t max_by_T = ceil_div(T, ch);
    int w = 1;
    while (w < 16 && w < want && w < max_by_T) w <<= 1;
    return w;
}

template <bool G, bool F, int VEC, int CH>
void launch_gae(int W, const float *v_s, const float *v_s_next, const float *rew, const uint8_t *terminated,
                const uint8_t *truncated, int64_t T, int64_t n_lane, int64_t lanes_per_env, const int32_t *env_start,
                const int32_t *env_len, double gamma, double gl, double v_scale, const double *rms, double rms_eps,
                float *returns_out, float *adv_out, hipStream_t st) {
    dim3 block(64, W), grid((unsigned)ceil_div(n_lane, 64 * VEC));
    const size_t shmem = (size_t)2 * W * 64 * VEC * 2 * sizeof(double);
    hipLaunchKernelGGL((gae_lanes_kernel<G, F, VEC, CH>), grid, block, shmem, st, v_s, v_s_next, rew, terminated,
                       truncated, T, n_lane, lanes_per_env, env_start, env_len, gamma, gl, v_scale, rms, rms_eps, returns_out,
                       adv_out);
}

}  // namespace

// Workspaces of the parallel long-series scan: ONE per device (registered from that device), cut into slots so that two launches in
// flight never share generation / flags / maps: slots [0, kScanEagerSlots) belong to the eager streams in order of first use,
// every launch recorded into a hipGraph takes a slot of its own from the rest (a graph may be replayed on any stream, beside
// eager work).  No free slot, no workspace on the launch's device: the sequential kernel runs.
constexpr int kScanMaxDev = 16, kScanSlots = 64, kScanEagerSlots = 8;
struct ScanReg {
    ScanWs *ws = nullptr;
    int32_t *err_host = nullptr;
    hipStream_t eager_stream[kScanEagerSlots];
    int n_eager = 0, n_captured = 0;
};
static ScanReg g_scan[kScanMaxDev];

TSM_EXPORT int64_t tsm_gae_scan_workspace_bytes(void) { return (int64_t)sizeof(ScanWs) * kScanSlots; }

// Register (or, with nullptr, withdraw) the ZEROED device workspace of the parallel long-series scan for the CURRENT device.
// Without one the sequential form runs on that device.  The memory stays the caller's.
TSM_EXPORT int tsm_gae_set_scan_workspace(void *workspace, int64_t bytes) {
    TSM_REQUIRE(!workspace || bytes >= tsm_gae_scan_workspace_bytes(), "tsm_gae_set_scan_workspace: needs %lld bytes",
                (long long)tsm_gae_scan_workspace_bytes());
    int dev = 0;
    if (!workspace && hipGetDevice(&dev) != hipSuccess) {   // withdrawing is always allowed (a host without a device: nothing registered)
        (void)hipGetLastError();
        for (auto &r : g_scan) r = ScanReg{};
        return TSM_OK;
    }
    TSM_HIP(hipGetDevice(&dev));
    TSM_REQUIRE(dev >= 0 && dev < kScanMaxDev, "tsm_gae_set_scan_workspace: device %d out of range", dev);
    int32_t *keep = g_scan[dev].err_host;
    g_scan[dev] = ScanReg{};
    g_scan[dev].ws = static_cast<ScanWs *>(workspace);
    g_scan[dev].err_host = keep;
    return TSM_OK;
}

// A word of PINNED (device-visible) host memory for the current device: set to 1 by a scan whose bounded wait ran out (its carry is
// NaN then).  The host reads it where it reads its loss statistics -- no synchronisation of its own.
TSM_EXPORT int tsm_gae_set_scan_error_word(int32_t *host_pinned) {
    int dev = 0;
    TSM_HIP(hipGetDevice(&dev));
    TSM_REQUIRE(dev >= 0 && dev < kScanMaxDev, "tsm_gae_set_scan_error_word: device %d out of range", dev);
    g_scan[dev].err_host = host_pinned;
    return TSM_OK;
}

// the slot of a launch on `st` (nullptr: none -- the sequential kernel serves)
static ScanWs *scan_slot(hipStream_t st, int32_t **err_host) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kScanMaxDev || !g_scan[dev].ws) return nullptr;
    ScanReg &r = g_scan[dev];
    *err_host = r.err_host;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) return nullptr;
    if (cs != hipStreamCaptureStatusNone) {
        if (kScanEagerSlots + r.n_captured >= kScanSlots) return nullptr;
        return r.ws + kScanEagerSlots + r.n_captured++;
    }
    for (int i = 0; i < r.n_eager; ++i)
        if (r.eager_stream[i] == st) return r.ws + i;
    if (r.n_eager >= kScanEagerSlots) return nullptr;
    r.eager_stream[r.n_eager] = st;
    return r.ws + r.n_eager++;
}

extern "C" __attribute__((visibility("default"))) void tsm_debug_gae_config(int vec, int ch, int w) {
    g_force_vec = vec; g_force_ch = ch; g_force_w = w;
}

static int gae_impl(const float *v_s, const float *v_s_next, const float *rew, const uint8_t *terminated,
                    const uint8_t *truncated, int flags_per_lane, int64_t T, int64_t n_lane, int64_t lanes_per_env,
                    const int32_t *env_start, const int32_t *env_len, double gamma, double gae_lambda, double v_scale,
                    const double *rms, double rms_eps, float *returns_out, float *adv_out, void *stream) {
    TSM_REQUIRE(T >= 0 && n_lane >= 0, "tsm_gae_lanes: negative size T=%lld n_lane=%lld",
                (long long)T, (long long)n_lane);
    if (T == 0 || n_lane == 0) return TSM_OK;
    TSM_REQUIRE(v_s && v_s_next && rew && terminated && truncated && returns_out && adv_out,
                "tsm_gae_lanes: null pointer");
    TSM_REQUIRE(lanes_per_env >= 1 && n_lane % lanes_per_env == 0,
                "tsm_gae_lanes: n_lane=%lld not a multiple of lanes_per_env=%lld",
                (long long)n_lane, (long long)lanes_per_env);
    TSM_REQUIRE(rms || v_scale > 0.0, "tsm_gae_lanes: v_scale must be > 0");
    const double gl = gamma * gae_lambda;
    const bool generic = env_start || env_len;
    hipStream_t st = tsm_stream(stream);
    // Steps per thread: 4 (measured best on MI355X: 41.7 % of HBM peak at T=25 x 32768 lanes, 67 % at T=2048; 8 steps
    // per thread and the 8-/16-byte-per-lane variants VEC=2/4 lose to register pressure from the f64 scan state --
    // profiles/r01_gae_sweep.txt).  VEC stays a template parameter for that experiment; only VEC=1 is instantiated.
    // few lanes and a long series (one lane per policy in the MARL trainers' learn(batch)): time-parallel kernel
    if (!generic && n_lane <= 64 && T >= 1024 && !g_force_ch && !g_force_w) {
        const dim3 grid((unsigned)n_lane), block(1024);
#define LONG(F, C) hipLaunchKernelGGL((gae_long_kernel<F, C>), grid, block, 0, st, v_s, v_s_next, rew, terminated, truncated, T, \
                                      n_lane, lanes_per_env, gamma, gl, v_scale, rms, rms_eps, returns_out, adv_out)
        // CH = 4 steps per thread whatever T: with 16 (one super-chunk up to 16 384 steps) the f64 scan state spills 404 VGPRs at the
        // 128 registers a 1024-thread workgroup leaves per lane -- 122.6 us for the trainers' 12 800-row lane against 27.3 us as four
        // super-chunks of 4096 steps (8 steps: 100 spills, 71.4 us); the same bits in all three (tools/gae_long_time.py, round 4)
        const int64_t n_sc = ceil_div(T, 4096);
        int32_t *err_host = nullptr;
        ScanWs *ws = (n_sc > 1 && n_lane * n_sc <= kScanMaxWg) ? scan_slot(st, &err_host) : nullptr;
        if (ws) {   // super-chunks side by side (see gae_long_par_kernel)
            const dim3 grid2((unsigned)n_lane, (unsigned)n_sc);
            if (flags_per_lane)
                hipLaunchKernelGGL((gae_long_par_kernel<true>), grid2, block, 0, st, v_s, v_s_next, rew, terminated, truncated, T, n_lane,
                                   lanes_per_env, gamma, gl, v_scale, rms, rms_eps, returns_out, adv_out, ws, err_host);
            else
                hipLaunchKernelGGL((gae_long_par_kernel<false>), grid2, block, 0, st, v_s, v_s_next, rew, terminated, truncated, T, n_lane,
                                   lanes_per_env, gamma, gl, v_scale, rms, rms_eps, returns_out, adv_out, ws, err_host);
            TSM_LAUNCH_CHECK();
            return TSM_OK;
        }
        if (flags_per_lane) LONG(true, 4); else LONG(false, 4);
#undef LONG
        TSM_LAUNCH_CHECK();
        return TSM_OK;
    }
    int ch = 4;
    if (g_force_ch) ch = g_force_ch;
    (void)g_force_vec;
    int W = pick_waves(T, n_lane, 1, ch);
    if (g_force_w) W = g_force_w;
#define ARGS W, v_s, v_s_next, rew, terminated, truncated, T, n_lane, lanes_per_env, env_start, env_len, gamma, gl, \
             v_scale, rms, rms_eps, returns_out, adv_out, st
#define BY_CH(G, F)                                          \
    do {                                                     \
        if (ch == 2) launch_gae<G, F, 1, 2>(ARGS);           \
        else launch_gae<G, F, 1, 4>(ARGS);                   \
    } while (0)
    if (generic) { if (flags_per_lane) BY_CH(true, true); else BY_CH(true, false); }
    else { if (flags_per_lane) BY_CH(false, true); else BY_CH(false, false); }
#undef BY_CH
#undef ARGS
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int tsm_gae_lanes(const float *v_s, const float *v_s_next, const float *rew,
                             const uint8_t *terminated, const uint8_t *truncated,
                             int flags_per_lane, int64_t T, int64_t n_lane, int64_t lanes_per_env,
                             const int32_t *env_start, const int32_t *env_len, double gamma,
                             double gae_lambda, double v_scale, float *returns_out, float *adv_out,
                             void *stream) {
    return gae_impl(v_s, v_s_next, rew, terminated, truncated, flags_per_lane, T, n_lane, lanes_per_env, env_start,
                    env_len, gamma, gae_lambda, v_scale, nullptr, 0.0, returns_out, adv_out, stream);
}

TSM_EXPORT int tsm_gae_lanes_rms(const float *v_s, const float *v_s_next, const float *rew,
                                 const uint8_t *terminated, const uint8_t *truncated,
                                 int flags_per_lane, int64_t T, int64_t n_lane, int64_t lanes_per_env,
                                 const int32_t *env_start, const int32_t *env_len, double gamma,
                                 double gae_lambda, const double *rms, double rms_eps, float *returns_out,
                                 float *adv_out, void *stream) {
    TSM_REQUIRE(rms, "tsm_gae_lanes_rms: rms is null");
    return gae_impl(v_s, v_s_next, rew, terminated, truncated, flags_per_lane, T, n_lane, lanes_per_env, env_start,
                    env_len, gamma, gae_lambda, 1.0, rms, rms_eps, returns_out, adv_out, stream);
}

TSM_EXPORT int tsm_mc_return_to_go_lanes(const float *rew, int64_t T, int64_t n_lane, double gamma,
                                         float *out, void *stream) {
    TSM_REQUIRE(T >= 0 && n_lane >= 0, "tsm_mc_return_to_go_lanes: negative size");
    if (T == 0 || n_lane == 0) return TSM_OK;
    TSM_REQUIRE(rew && out, "tsm_mc_return_to_go_lanes: null pointer");
    hipLaunchKernelGGL(mc_return_kernel, dim3((unsigned)ceil_div(n_lane, 256)), dim3(256), 0,
                       tsm_stream(stream), rew, T, n_lane, gamma, out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int64_t tsm_rms_update_work_elems(int64_t n) { return n < 0 ? -1 : 2 * ceil_div(n > 0 ? n : 1, kRmsChunk); }

TSM_EXPORT int tsm_rms_update(const float *returns, const int64_t *ids, int64_t n, double *rms, double rms_eps,
                              double *work, void *stream) {
    TSM_REQUIRE(n >= 0, "tsm_rms_update: negative n");
    if (n == 0) return TSM_OK;
    TSM_REQUIRE(returns && rms && work, "tsm_rms_update: null pointer");
    const int64_t n_part = ceil_div(n, kRmsChunk);
    hipLaunchKernelGGL(rms_partial_kernel, dim3((unsigned)n_part), dim3(kRmsThreads), 0, tsm_stream(stream), returns, ids,
                       n, rms, rms_eps, work);
    TSM_LAUNCH_CHECK();
    hipLaunchKernelGGL(rms_merge_kernel, dim3(1), dim3(64), 0, tsm_stream(stream), work, n_part, n, rms);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}


// ---- V(obs_next) from V(obs) of the next slot ------------------------------------------------------------------------
// Rows written by a Collector are chained: obs_next of slot t IS obs of slot t + 1 unless the episode ended at t
// (collector.py:1040-1069: the next observation becomes the current one; finished envs are reset).  For T unrotated,
// equally filled slots the critic pass over obs_next (a2c.py:124) therefore repeats the pass over obs (a2c.py:123) row
// for row, except for the last slot and for rows that end an episode.  tsm_any_nonzero_u8 tells whether any episode
// ended before the last slot; tsm_value_next_select then takes V(obs_next) from the next slot's V(obs) and the last
// slot's own critic pass -- or, if an episode did end early, from the full pass that tsm_mlp_forward_cond ran.
namespace {

__global__ __launch_bounds__(1024) void any_nonzero_u8_kernel(const uint8_t *__restrict__ x, int64_t n, int32_t *__restrict__ out) {
    int any = 0;
    const int64_t n16 = ((uintptr_t)x & 15) == 0 ? n / 16 : 0;
    const uint4 *x16 = reinterpret_cast<const uint4 *>(x);
    for (int64_t i = threadIdx.x; i < n16; i += 1024) {
        const uint4 v = x16[i];
        any |= (v.x | v.y | v.z | v.w) != 0u;
    }
    for (int64_t i = n16 * 16 + threadIdx.x; i < n; i += 1024) any |= x[i] != 0;
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) *out = any ? 1 : 0;
}

__global__ void value_next_select_kernel(const float *__restrict__ v_s, const float *__restrict__ v_last,
                                         const float *__restrict__ v_full, const int32_t *__restrict__ flag, int64_t T,
                                         int64_t U, float *__restrict__ v_next) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * U) return;
    if (*flag) { v_next[i] = v_full[i]; return; }
    const int64_t t = i / U;
    v_next[i] = t + 1 < T ? v_s[i + U] : v_last[i - t * U];
}

// the same selection for ENV-major rows [E][T][U] (the MARL trainers' per-agent batches, sample_indices(0) order)
__global__ void value_next_select_em_kernel(const float *__restrict__ v_s, const float *__restrict__ v_last,
                                            const float *__restrict__ v_full, const int32_t *__restrict__ flag, int64_t E,
                                            int64_t T, int64_t U, float *__restrict__ v_next) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E * T * U) return;
    if (*flag) { v_next[i] = v_full[i]; return; }
    const int64_t u = i % U, r = i / U, t = r % T, e = r / T;
    v_next[i] = t + 1 < T ? v_s[i + U] : v_last[e * U + u];
}

// ignore_obs_next buffers (buffer_base.py:612-616): obs_next of a row is READ as obs[next(index)], and next(index) is the
// row itself at an episode end and at the newest row, else the following slot.  For T unrotated, equally filled slots:
//   V(obs_next)[t][u] = (t == T - 1 || done[t][u / lanes_per_env]) ? V(obs)[t][u] : V(obs)[t + 1][u]
__global__ void value_next_index_kernel(const float *__restrict__ v_s, const uint8_t *__restrict__ done, int64_t T,
                                        int64_t U, int64_t lanes_per_env, float *__restrict__ v_next) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * U) return;
    const int64_t t = i / U, u = i - t * U;
    const bool self = t + 1 >= T || done[t * (U / lanes_per_env) + u / lanes_per_env] != 0;
    v_next[i] = self ? v_s[i] : v_s[i + U];
}

}  // namespace

TSM_EXPORT int tsm_value_next_index(const float *v_s, const uint8_t *done, int64_t T, int64_t U, int64_t lanes_per_env,
                                    float *v_next_out, void *stream) {
    TSM_REQUIRE(T >= 1 && U >= 1 && lanes_per_env >= 1 && U % lanes_per_env == 0,
                "tsm_value_next_index: T, U >= 1 and U a multiple of lanes_per_env");
    TSM_REQUIRE(v_s && done && v_next_out, "tsm_value_next_index: null pointer");
    hipLaunchKernelGGL(value_next_index_kernel, dim3((unsigned)ceil_div(T * U, 256)), dim3(256), 0, tsm_stream(stream), v_s,
                       done, T, U, lanes_per_env, v_next_out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int tsm_value_next_select_env_major(const float *v_s, const float *v_last, const float *v_full,
                                               const int32_t *flag, int64_t E, int64_t T, int64_t U, float *v_next_out,
                                               void *stream) {
    TSM_REQUIRE(E >= 1 && T >= 1 && U >= 1, "tsm_value_next_select_env_major: E, T and U must be >= 1");
    TSM_REQUIRE(v_s && v_last && v_full && flag && v_next_out, "tsm_value_next_select_env_major: null pointer");
    hipLaunchKernelGGL(value_next_select_em_kernel, dim3((unsigned)ceil_div(E * T * U, 256)), dim3(256), 0, tsm_stream(stream),
                       v_s, v_last, v_full, flag, E, T, U, v_next_out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int tsm_any_nonzero_u8(const uint8_t *x, int64_t n, int32_t *flag_out, void *stream) {
    TSM_REQUIRE(n >= 0, "tsm_any_nonzero_u8: negative n");
    TSM_REQUIRE(flag_out && (x || n == 0), "tsm_any_nonzero_u8: null pointer");
    hipLaunchKernelGGL(any_nonzero_u8_kernel, dim3(1), dim3(1024), 0, tsm_stream(stream), x, n, flag_out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int tsm_value_next_select(const float *v_s, const float *v_last, const float *v_full, const int32_t *flag,
                                     int64_t T, int64_t U, float *v_next_out, void *stream) {
    TSM_REQUIRE(T >= 1 && U >= 1, "tsm_value_next_select: T and U must be >= 1");
    TSM_REQUIRE(v_s && v_last && v_full && flag && v_next_out, "tsm_value_next_select: null pointer");
    hipLaunchKernelGGL(value_next_select_kernel, dim3((unsigned)ceil_div(T * U, 256)), dim3(256), 0, tsm_stream(stream),
                       v_s, v_last, v_full, flag, T, U, v_next_out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}
