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

constexpr int CH = 8;  // steps per thread per super-chunk

template <bool GENERIC, bool FLAGS_PER_LANE>
__global__ __launch_bounds__(1024) void gae_lanes_kernel(
    const float *__restrict__ v_s, const float *__restrict__ v_n, const float *__restrict__ rew,
    const uint8_t *__restrict__ term, const uint8_t *__restrict__ trunc, int64_t T, int64_t L,
    int64_t lanes_per_env, const int32_t *__restrict__ env_start,
    const int32_t *__restrict__ env_len, double gamma, double gl, double v_scale,
    float *__restrict__ ret_out, float *__restrict__ adv_out) {
    extern __shared__ double lds[];  // [2][W][64][2]
    const int W = blockDim.y;
    const int lx = threadIdx.x, w = threadIdx.y;
    const int64_t lane = (int64_t)blockIdx.x * 64 + lx;
    const bool live = lane < L;
    const int64_t env = live ? lane / lanes_per_env : 0;
    const int64_t n_env = L / lanes_per_env;
    int64_t len = T, start = 0;
    if (GENERIC && live) {
        if (env_len) len = env_len[env];
        if (env_start) start = env_start[env];
    }
    // the block walks the longest lane length it owns; shorter lanes mask their tail
    int64_t max_len = T;  // uniform upper bound keeps the barrier structure uniform
    const int64_t SC = (int64_t)W * CH;
    const int64_t n_sc = (max_len + SC - 1) / SC;
    const double inv_scale = 1.0 / v_scale;

    double carry_super = 0.0;
    for (int64_t s = n_sc - 1; s >= 0; --s) {
        const int64_t k0 = s * SC + (int64_t)w * CH;
        double delta[CH];
        float vs_f[CH];
        unsigned keep = 0;  // bit k set: discount = gl (no end flag)
        unsigned valid = 0;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int64_t kk = k0 + k;
            const bool ok = live && kk < len;
            delta[k] = 0.0;
            vs_f[k] = 0.f;
            if (ok) {
                int64_t slot = kk;
                if (GENERIC) { slot = start + kk; if (slot >= T) slot -= T; }
                const int64_t i = slot * L + lane;
                const int64_t fi = FLAGS_PER_LANE ? i : slot * n_env + env;
                const float a = v_s[i], b = v_n[i], r = rew[i];
                const uint8_t te = term[fi], tr = trunc[fi];
                const double vs = (double)a * v_scale;
                const double vn = te ? 0.0 : (double)b * v_scale;
                delta[k] = (double)r + vn * gamma - vs;
                vs_f[k] = a;
                const bool end = te | tr | (kk == len - 1);
                keep |= (end ? 0u : 1u) << k;
                valid |= 1u << k;
            } else {
                keep |= 1u << k;  // masked step: identity map (delta 0, discount 1 handled below)
            }
        }
        // local scan with carry 0 -> (P, B)
        double P = 1.0, B = 0.0;
#pragma unroll
        for (int k = CH - 1; k >= 0; --k) {
            const bool v = (valid >> k) & 1u;
            const double d = v ? (((keep >> k) & 1u) ? gl : 0.0) : 1.0;
            B = delta[k] + d * B;
            P *= d;
        }
        double *buf = lds + (size_t)(s & 1) * W * 64 * 2;
        buf[((size_t)w * 64 + lx) * 2 + 0] = P;
        buf[((size_t)w * 64 + lx) * 2 + 1] = B;
        __syncthreads();
        // fold later chunks (W-1 .. w+1) onto the super-chunk carry
        double c = carry_super;
        for (int ww = W - 1; ww > w; --ww) {
            const double Pw = buf[((size_t)ww * 64 + lx) * 2 + 0];
            const double Bw = buf[((size_t)ww * 64 + lx) * 2 + 1];
            c = Bw + Pw * c;
        }
        // replay own chunk serially from the true carry-in
        double g = c;
#pragma unroll
        for (int k = CH - 1; k >= 0; --k) {
            const bool v = (valid >> k) & 1u;
            if (v) {
                const double d = ((keep >> k) & 1u) ? gl : 0.0;
                g = delta[k] + d * g;
                const int64_t kk = k0 + k;
                int64_t slot = kk;
                if (GENERIC) { slot = start + kk; if (slot >= T) slot -= T; }
                const int64_t i = slot * L + lane;
                adv_out[i] = (float)g;
                ret_out[i] = (float)((g + (double)vs_f[k] * v_scale) * inv_scale);
            }
        }
        // carry for the next (earlier) super-chunk = gae at the first step of chunk 0
        double cs = carry_super;
        for (int ww = W - 1; ww >= 0; --ww) {
            const double Pw = buf[((size_t)ww * 64 + lx) * 2 + 0];
            const double Bw = buf[((size_t)ww * 64 + lx) * 2 + 1];
            cs = Bw + Pw * cs;
        }
        carry_super = cs;
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

int pick_waves(int64_t T, int64_t L) {
    // enough waves to fill 256 CUs x 8, but no more chunks than the series has
    int64_t blocks = ceil_div(L, 64);
    int64_t want = ceil_div(2048, blocks);
    int64_t max_by_T = ceil_div(T, CH);
    int w = 1;
    while (w < 16 && w < want && w < max_by_T) w <<= 1;
    return w;
}

}  // namespace

TSM_EXPORT int tsm_gae_lanes(const float *v_s, const float *v_s_next, const float *rew,
                             const uint8_t *terminated, const uint8_t *truncated,
                             int flags_per_lane, int64_t T, int64_t n_lane, int64_t lanes_per_env,
                             const int32_t *env_start, const int32_t *env_len, double gamma,
                             double gae_lambda, double v_scale, float *returns_out, float *adv_out,
                             void *stream) {
    TSM_REQUIRE(T >= 0 && n_lane >= 0, "tsm_gae_lanes: negative size T=%lld n_lane=%lld",
                (long long)T, (long long)n_lane);
    if (T == 0 || n_lane == 0) return TSM_OK;
    TSM_REQUIRE(v_s && v_s_next && rew && terminated && truncated && returns_out && adv_out,
                "tsm_gae_lanes: null pointer");
    TSM_REQUIRE(lanes_per_env >= 1 && n_lane % lanes_per_env == 0,
                "tsm_gae_lanes: n_lane=%lld not a multiple of lanes_per_env=%lld",
                (long long)n_lane, (long long)lanes_per_env);
    TSM_REQUIRE(v_scale > 0.0, "tsm_gae_lanes: v_scale must be > 0");
    const int W = pick_waves(T, n_lane);
    dim3 block(64, W), grid((unsigned)ceil_div(n_lane, 64));
    const size_t shmem = (size_t)2 * W * 64 * 2 * sizeof(double);
    const double gl = gamma * gae_lambda;
    const bool generic = env_start || env_len;
    hipStream_t st = tsm_stream(stream);
#define LAUNCH(G, F)                                                                             \
    hipLaunchKernelGGL((gae_lanes_kernel<G, F>), grid, block, shmem, st, v_s, v_s_next, rew,     \
                       terminated, truncated, T, n_lane, lanes_per_env, env_start, env_len,      \
                       gamma, gl, v_scale, returns_out, adv_out)
    if (generic) { if (flags_per_lane) LAUNCH(true, true); else LAUNCH(true, false); }
    else { if (flags_per_lane) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
    TSM_LAUNCH_CHECK();
    return TSM_OK;
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
