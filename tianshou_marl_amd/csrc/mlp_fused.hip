// mlp_fused.hip -- actor + critic MLP (obs -> H -> H -> {A logits | 1 value}) on f32 MFMA, fused with
// the categorical head (rollout) and with the PPO loss + full backward pass (update).
//
// Replaces, for the discrete-action actor/critic of the path:
//   rollout : ProbabilisticActorPolicy.forward = actor(obs) -> Categorical(logits).sample()/mode
//             (/root/reference/tianshou/algorithm/modelfree/reinforce.py:167-192) with
//             DiscreteActor / Net / MLP (utils/net/discrete.py:67-89, common.py:172-181,343-369),
//             critic(obs) of _add_returns_and_advantages (a2c.py:121-127) and logp_old (ppo.py:157-161);
//   update  : one gradient step of PPO._update_with_batch (ppo.py:182-212) up to the parameter gradients:
//             dist = policy(mb).dist; value = critic(mb.obs); loss; loss.backward().
// Parameter vector (flat f32, torch nn.Linear layouts, actor then critic == ActorCritic.parameters()
// order, utils/net/common.py:461-474):
//   actor : W1[H][D] b1[H] W2[H][H] b2[H] W3[A][H] b3[A]     critic: W1[H][D] b1[H] W2[H][H] b2[H] W3[1][H] b3[1]
//
// gfx950 mapping: a workgroup = 4 waves owns tiles of R = 16 rows.  Weights of both nets are staged once
// per workgroup in LDS (row stride = 2 * odd dwords => conflict-free 16-row x 2-k ds_read_b32 patterns);
// every GEMM is v_mfma_f32_16x16x4_f32 (exact f32 FMA chains, 256 FLOP/clk/CU).  Wave w owns output
// column block w of each net, so L1/L2/backward run two independent accumulator chains per wave
// (actor | critic).  Weight gradients accumulate in registers across all tiles of the workgroup and are
// written once as a per-workgroup slab [P]; tsm_adam_step sums the slabs in fixed order (deterministic).
//
// HBM traffic per sample and gradient step: obs 4*D + act 4 + logp_old 4 + adv 4 + returns 4 (+ perm 8)
// = 88 B at D = 18 (SURVEY.md 8d "fully fused update step").
#include "common.h"
#include "mlp_tile.h"
#include "philox.h"

extern long long *g_tsm_stamps;  // abi.hip (diagnostics)

namespace {

// ------------------------------------------------------------------------------------------------
// rollout / inference: logits, value, optional sampling + log-prob
// ------------------------------------------------------------------------------------------------
template <int H, int DS = 0>   // DS: obs width this instantiation is compiled for (5 actions); 0 = from the arguments
__global__ __launch_bounds__(NT) void policy_forward_kernel(
    const float *__restrict__ P, const float *__restrict__ img, Dims d_arg, const float *__restrict__ obs, int64_t B,
    uint64_t seed,
    uint64_t offset, const uint64_t *__restrict__ offset_dev, int mode /*0 none, 1 sample, 2 argmax, 3 given*/,
    float *__restrict__ logits_out,
    float *__restrict__ value_out, int32_t *__restrict__ act_io, float *__restrict__ logp_out) {
    extern __shared__ float lds[];
    const Dims d = DS ? dims_const(DS, 5) : d_arg;
    const Lay<H> ly(d, false);
    if (img) stage_image<H>(lds, ly, img);
    else stage_weights<H>(lds, ly, d, P);
    if (offset_dev) offset += *offset_dev;
    const int64_t n_tiles = (B + R - 1) / R;
    // the next tile's rows are in flight (registers) while this tile runs: a workgroup of the 512-wide grid walks up to
    // n_tiles / 512 tiles and would otherwise pay one global round trip per tile in front of its forward pass
    float xr[kXRegs];
    if ((int64_t)blockIdx.x < n_tiles) prefetch_tile_x(xr, d, obs, nullptr, 0, (int64_t)blockIdx.x * R, B);
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int64_t row0 = t * R;
        __syncthreads();  // previous tile's epilogue reads / weight staging complete
        commit_tile_x(lds, ly.X, d, xr);
        if (t + gridDim.x < n_tiles) prefetch_tile_x(xr, d, obs, nullptr, 0, (t + gridDim.x) * R, B);
        __syncthreads();
        tile_forward<H>(lds, ly, d);
        // epilogue: one thread per (row, col) for the stores, one thread per row for the head
        if (logits_out) {
            for (int e = threadIdx.x; e < R * d.A; e += NT) {
                const int r = e / d.A, c = e - r * d.A;
                if (row0 + r < B) logits_out[(row0 + r) * d.A + c] = lds[ly.OUT + r * ly.ldo + c];
            }
        }
        if (threadIdx.x < R && row0 + threadIdx.x < B) {
            const int r = threadIdx.x;
            const int64_t i = row0 + r;
            const float *lg = lds + ly.OUT + r * ly.ldo;
            if (value_out) value_out[i] = lg[16];
            if (mode != 0) {
                float m = -INFINITY;
                int arg = 0;
                for (int j = 0; j < d.A; ++j) if (lg[j] > m) { m = lg[j]; arg = j; }
                float s = 0.f;
                for (int j = 0; j < d.A; ++j) s += expf(lg[j] - m);
                int a = arg;
                if (mode == 1) {
                    const float u = tsm_philox_uniform(seed, offset + (uint64_t)i) * s;
                    float c = 0.f;
                    a = d.A - 1;
                    for (int j = 0; j < d.A; ++j) { c += expf(lg[j] - m); if (u < c) { a = j; break; } }
                } else if (mode == 3) {
                    a = act_io[i];
                }
                if (mode != 3) act_io[i] = a;
                if (logp_out) logp_out[i] = lg[a] - (m + logf(s));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// update: forward + PPO loss + backward -> per-workgroup gradient slab + loss partial sums
// ------------------------------------------------------------------------------------------------
struct LossCfg {
    float eps_clip, dual_clip, vf_coef, ent_coef;
    int value_clip, adv_norm, kind;
};

template <int H>
__global__ __launch_bounds__(NT) void ppo_update_kernel(
    const float *__restrict__ P, const float *__restrict__ img, Dims d, const float *__restrict__ obs,
    const int32_t *__restrict__ act,
    const float *__restrict__ logp_old, const float *__restrict__ adv, const float *__restrict__ returns,
    const float *__restrict__ v_s_old, const int64_t *__restrict__ perm, int64_t first, int64_t M,
    const float *__restrict__ adv_stats, LossCfg cfg, float *__restrict__ slabs,
    double *__restrict__ loss_partial, int64_t *__restrict__ opt_step_dev, long long *stamps) {
#define USTAMP(k) do { if (stamps && blockIdx.x == 0 && threadIdx.x == 0) stamps[k] = (long long)wall_clock64(); } while (0)
    extern __shared__ float lds[];
    USTAMP(0);
    // diagnostics (>= 1024 slots): [64 + 2b], [65 + 2b] = start / end of workgroup b
    if (stamps && threadIdx.x == 0 && blockIdx.x < 256) stamps[64 + 2 * blockIdx.x] = (long long)wall_clock64();
    // device-resident optimizer step count (hipGraph replay): bumped here, read by the Adam kernel that follows
    if (opt_step_dev && blockIdx.x == 0 && threadIdx.x == 0) *opt_step_dev += 1;
    const Lay<H> ly(d, true);
    const POff<H> po(d.D, d.A);
    const int64_t n_tiles = (M + R - 1) / R;
    // software pipeline: the dependent gathers (perm -> row id -> obs / act / adv / ...) of the NEXT tile are in
    // flight while weights are staged / the current tile is computed
    float xr[kXRegs];
    struct RowIn { int a_idx; float adv, logp_old, ret, v_old; } rin;
    auto row_id = [&](int64_t row0) -> int64_t {
        const int64_t i = row0 + (threadIdx.x >> 4);
        return i < M ? (perm ? perm[i] : first + i) : -1;
    };
    auto prefetch_row = [&](int64_t src) {
        rin.a_idx = 0; rin.adv = 0.f; rin.logp_old = 0.f; rin.ret = 0.f; rin.v_old = 0.f;
        if (src >= 0) {
            rin.a_idx = act[src]; rin.adv = adv[src]; rin.logp_old = logp_old[src]; rin.ret = returns[src];
            if (cfg.value_clip) rin.v_old = v_s_old[src];
        }
    };
    // two memory round trips in all: (1) the row ids, (2) the id-dependent gathers together with the weight image,
    // which goes global -> LDS by DMA.  The DMA is issued LAST: vector memory returns in issue order and, with a DMA
    // outstanding, hipcc waits for everything at the next use of a load result -- the only such use before the
    // barrier is that of the row ids, which happens before the DMA is issued.
    int64_t xs[kXRegs];
    int64_t rsrc = -1;
    const bool have_tile = (int64_t)blockIdx.x < n_tiles;
    if (have_tile) {
        prefetch_tile_ids(xs, d, perm, first, (int64_t)blockIdx.x * R, M);
        rsrc = row_id((int64_t)blockIdx.x * R);
        prefetch_tile_vals(xr, xs, obs);
        prefetch_row(rsrc);
    }
    if (img) stage_image<H>(lds, ly, img);
    else stage_weights<H>(lds, ly, d, P);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const float invM = 1.0f / (float)M;
    float a_mean = 0.f, a_std = 1.f;
    if (cfg.adv_norm) { a_mean = adv_stats[0]; a_std = adv_stats[1]; }

    // register-resident gradient accumulators (MFMA C layout)
    f4 g_W3a = {0.f, 0.f, 0.f, 0.f};        // rows i (A pad 16) x cols 16w..
    f4 g_W2a[4], g_W2c[4];                  // rows 16w.. x col block jb
    f4 g_W1a[kMaxJ], g_W1c[kMaxJ];          // rows 16w.. x obs col block jb
#pragma unroll
    for (int j = 0; j < 4; ++j) { g_W2a[j] = g_W3a; g_W2c[j] = g_W3a; }
#pragma unroll
    for (int j = 0; j < kMaxJ; ++j) { g_W1a[j] = g_W3a; g_W1c[j] = g_W3a; }
    float g_b1 = 0.f, g_b2 = 0.f;           // threads 0..2H-1: bias grads of column tid (actor | critic)
    float g_W3c = 0.f;                      // threads 0..H-1
    float g_b3 = 0.f;                       // threads 0..A-1: b3a ; thread 16: b3c
    double s_clip = 0.0, s_vf = 0.0, s_ent = 0.0;  // threads 16*r (one per tile row)

    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int64_t row0 = t * R;
        __syncthreads();
        USTAMP(1);
        commit_tile_x(lds, ly.X, d, xr);
        const RowIn cur = rin;
        __syncthreads();
        USTAMP(2);
        tile_forward<H>(lds, ly, d);
        USTAMP(3);

        // ---- loss head (ppo.py:182-211): 16 lanes per row, lane j owns action j -> D3 = [dlogits(16) | dvalue] ----
        {
            const int r = threadIdx.x >> 4, j = threadIdx.x & 15;
            const int64_t i = row0 + r;
            float *d3 = lds + ly.D3 + r * ly.ldo;
            float dl = 0.f, dv = 0.f;
            if (i < M) {  // uniform over the 16 lanes of a row
                const float *lg = lds + ly.OUT + r * ly.ldo;
                const bool on = j < d.A;
                const float x = on ? lg[j] : -INFINITY;
                const float m = row16_max(x);  // (DPP rotations: same operand pairs as the xor butterflies)
                const float ex = on ? expf(x - m) : 0.f;
                const float s = row16_sum(ex);
                const float l = on ? x - (m + logf(s)) : 0.f;   // log-softmax
                const float p = ex / s;
                const float h = row16_sum(on ? -p * l : 0.f);   // entropy
                const int a_idx = cur.a_idx;
                const float logp = __shfl(l, (threadIdx.x & 48) + a_idx, 64);
                float a = cur.adv;
                if (cfg.adv_norm) a = (a - a_mean) / (a_std + 1e-8f);
                float ratio, obj, g_ratio;  // d obj / d logp = g_ratio * ratio
                if (cfg.kind == 1) {  // plain policy gradient (a2c.py:263-264, reinforce.py:375-376): obj = logp * adv
                    ratio = 1.f; obj = logp * a; g_ratio = a;
                } else {
                    ratio = expf(logp - cur.logp_old);
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
                const float v = lg[16], ret = cur.ret;
                float vf, g_v;
                if (cfg.value_clip) {
                    const float vs = cur.v_old;
                    const float dd = v - vs;
                    const float dc = fminf(fmaxf(dd, -cfg.eps_clip), cfg.eps_clip);
                    const bool v_in = dd >= -cfg.eps_clip && dd <= cfg.eps_clip;
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
                dv = cfg.vf_coef * g_v * invM;
                const float g_logp = -g_ratio * ratio * invM, ec = cfg.ent_coef * invM;
                if (on) dl = g_logp * ((j == a_idx ? 1.f : 0.f) - p) + ec * p * (l + h);
                if (j == 0) { s_clip += (double)obj; s_vf += (double)vf; s_ent += (double)h; }
            }
            d3[j] = dl;
            if (j == 0) d3[16] = dv;
        }
        __syncthreads();

        if (t + gridDim.x < n_tiles) {  // next tile's gathers fly under this tile's backward pass
            prefetch_tile_x(xr, d, obs, perm, first, (t + gridDim.x) * R, M);
            prefetch_row(row_id((t + gridDim.x) * R));
        }
        USTAMP(4);
        // ---- output-layer gradients + dh2 ----
        {
            // dW3a[i][16w + j] += sum_r dlogits[r][i] * h2a[r][16w + j]
            const float *dA = lds + ly.D3 + kq * ly.ldo + r16;            // C-type read: [k=r][i]
            const float *hB = lds + ly.H2 + kq * ly.ld2 + 16 * w + r16;   // C-type read: [k=r][j]
#pragma unroll
            for (int k0 = 0; k0 < R; k0 += 4) g_W3a = mfma(dA[k0 * ly.ldo], hB[k0 * ly.ld2], g_W3a);
            // dh2a = dlogits[R x 8] . W3a[8 x H]  (rows >= A of W3a and cols >= A of dlogits are zero)
            f4 acc = {0.f, 0.f, 0.f, 0.f};
            const float *dR = lds + ly.D3 + r16 * ly.ldo + kq;            // R-type read
            const float *wB = lds + ly.W3a + kq * ly.ldh + 16 * w + r16;  // C-type read: [k=i][j]
#pragma unroll
            for (int k0 = 0; k0 < 16; k0 += 4) acc = mfma(dR[k0], wB[k0 * ly.ldh], acc);
            const int col = 16 * w + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = kq * 4 + r;
                lds[ly.D2 + row * ly.ld2 + col] = lds[ly.H2 + row * ly.ld2 + col] > 0.f ? acc[r] : 0.f;
            }
            // critic side (rank-1): dh2c[r][j] = dv[r] * W3c[j] * (h2c > 0)
            for (int e = threadIdx.x; e < R * H; e += NT) {
                const int r = e / H, j = e - r * H;
                const float hv = lds[ly.H2 + r * ly.ld2 + H + j];
                lds[ly.D2 + r * ly.ld2 + H + j] = hv > 0.f ? lds[ly.D3 + r * ly.ldo + 16] * lds[ly.W3c + j] : 0.f;
            }
            if (threadIdx.x < H) {  // dW3c[j] += sum_r dv[r] * h2c[r][j]
                float s = 0.f;
                for (int r = 0; r < R; ++r) s = fmaf(lds[ly.D3 + r * ly.ldo + 16], lds[ly.H2 + r * ly.ld2 + H + threadIdx.x], s);
                g_W3c += s;
            }
            if (threadIdx.x < 17) {  // b3a (cols 0..15) | b3c (col 16)
                float s = 0.f;
                for (int r = 0; r < R; ++r) s += lds[ly.D3 + r * ly.ldo + threadIdx.x];
                g_b3 += s;
            }
        }
        __syncthreads();

        // ---- hidden layer 2 gradients + dh1 ----
        {
            const float *dAa = lds + ly.D2 + kq * ly.ld2 + 16 * w + r16;  // [k=r][i = 16w + r16]
            const float *dAc = dAa + H;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                const float *hBa = lds + ly.H1 + kq * ly.ld2 + 16 * jb + r16;
                const float *hBc = hBa + H;
#pragma unroll
                for (int k0 = 0; k0 < R; k0 += 4) {
                    g_W2a[jb] = mfma(dAa[k0 * ly.ld2], hBa[k0 * ly.ld2], g_W2a[jb]);
                    g_W2c[jb] = mfma(dAc[k0 * ly.ld2], hBc[k0 * ly.ld2], g_W2c[jb]);
                }
            }
            if (threadIdx.x < 2 * H) {
                float s = 0.f;
                for (int r = 0; r < R; ++r) s += lds[ly.D2 + r * ly.ld2 + threadIdx.x];
                g_b2 += s;
            }
            // dh1 = dh2 . W2   (A: R-type on D2, B: C-type on W2[k][j])
            f4 acc_a = {0.f, 0.f, 0.f, 0.f}, acc_c = {0.f, 0.f, 0.f, 0.f};
            const float *dRa = lds + ly.D2 + r16 * ly.ld2 + kq;
            const float *dRc = dRa + H;
            const float *wBa = lds + ly.W2a + kq * ly.ldh + 16 * w + r16;
            const float *wBc = lds + ly.W2c + kq * ly.ldh + 16 * w + r16;
#pragma unroll
            for (int k0 = 0; k0 < H; k0 += 4) {
                acc_a = mfma(dRa[k0], wBa[k0 * ly.ldh], acc_a);
                acc_c = mfma(dRc[k0], wBc[k0 * ly.ldh], acc_c);
            }
            const int col = 16 * w + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = kq * 4 + r;
                lds[ly.D1 + row * ly.ld2 + col] = lds[ly.H1 + row * ly.ld2 + col] > 0.f ? acc_a[r] : 0.f;
                lds[ly.D1 + row * ly.ld2 + H + col] = lds[ly.H1 + row * ly.ld2 + H + col] > 0.f ? acc_c[r] : 0.f;
            }
        }
        __syncthreads();

        // ---- layer 1 gradients ----
        {
            const float *dAa = lds + ly.D1 + kq * ly.ld2 + 16 * w + r16;
            const float *dAc = dAa + H;
#pragma unroll
            for (int jb = 0; jb < kMaxJ; ++jb) {
                if (jb < d.nJ) {
                    const float *xB = lds + ly.X + kq * d.ld1 + 16 * jb + r16;
#pragma unroll
                    for (int k0 = 0; k0 < R; k0 += 4) {
                        const float b = xB[k0 * d.ld1];
                        g_W1a[jb] = mfma(dAa[k0 * ly.ld2], b, g_W1a[jb]);
                        g_W1c[jb] = mfma(dAc[k0 * ly.ld2], b, g_W1c[jb]);
                    }
                }
            }
            if (threadIdx.x < 2 * H) {
                float s = 0.f;
                for (int r = 0; r < R; ++r) s += lds[ly.D1 + r * ly.ld2 + threadIdx.x];
                g_b1 += s;
            }
        }
    }

    USTAMP(5);
    // ---- write this workgroup's gradient slab (flat parameter layout) ----
    float *S = slabs + (int64_t)blockIdx.x * po.total;
    const int colq = r16;  // C layout: col = lane & 15, row = kq*4 + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = kq * 4 + r;
        if (row < d.A) __builtin_nontemporal_store(g_W3a[r], &S[po.aW3 + row * H + 16 * w + colq]);
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            __builtin_nontemporal_store(g_W2a[jb][r], &S[po.aW2 + (16 * w + row) * H + 16 * jb + colq]);
            __builtin_nontemporal_store(g_W2c[jb][r], &S[po.cW2 + (16 * w + row) * H + 16 * jb + colq]);
        }
#pragma unroll
        for (int jb = 0; jb < kMaxJ; ++jb) {
            const int c = 16 * jb + colq;
            if (jb < d.nJ && c < d.D) {
                __builtin_nontemporal_store(g_W1a[jb][r], &S[po.aW1 + (16 * w + row) * d.D + c]);
                __builtin_nontemporal_store(g_W1c[jb][r], &S[po.cW1 + (16 * w + row) * d.D + c]);
            }
        }
    }
    if (threadIdx.x < 2 * H) {
        const int c = threadIdx.x;
        if (c < H) { __builtin_nontemporal_store(g_b1, &S[po.ab1 + c]); __builtin_nontemporal_store(g_b2, &S[po.ab2 + c]); __builtin_nontemporal_store(g_W3c, &S[po.cW3 + c]); }
        else { __builtin_nontemporal_store(g_b1, &S[po.cb1 + c - H]); __builtin_nontemporal_store(g_b2, &S[po.cb2 + c - H]); }
    }
    if (threadIdx.x < d.A) __builtin_nontemporal_store(g_b3, &S[po.ab3 + threadIdx.x]);
    if (threadIdx.x == 16) __builtin_nontemporal_store(g_b3, &S[po.cb3]);
    // loss partial sums: row r accumulated in thread 16*r (lane j == 0 of its 16-lane group)
    {
        __shared__ double s_red[3][NT / 64];
        double c = wave_sum(s_clip), v = wave_sum(s_vf), e = wave_sum(s_ent);
        if (lane == 0) { s_red[0][w] = c; s_red[1][w] = v; s_red[2][w] = e; }
        __syncthreads();
        if (threadIdx.x == 0) {
            loss_partial[4 * blockIdx.x + 0] = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
            loss_partial[4 * blockIdx.x + 1] = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
            loss_partial[4 * blockIdx.x + 2] = s_red[2][0] + s_red[2][1] + s_red[2][2] + s_red[2][3];
            loss_partial[4 * blockIdx.x + 3] = 0.0;
        }
    }
    USTAMP(6);
    if (stamps && threadIdx.x == 0 && blockIdx.x < 256) stamps[65 + 2 * blockIdx.x] = (long long)wall_clock64();
#undef USTAMP
}

// ------------------------------------------------------------------------------------------------
// update, one NET per workgroup: grid (n_blocks, 2), blockIdx.y = 0 actor | 1 critic
// ------------------------------------------------------------------------------------------------
// The PPO / policy-gradient loss is separable: the clip (or policy-gradient) and entropy terms depend on the actor
// only, the value term on the critic only.  A workgroup therefore carries ONE net through forward + loss + backward for
// its 16-row tiles: half the MFMA chain per wave (one accumulator chain per layer instead of two interleaved ones),
// half the gradient registers and half the slab stores on the critical path, 52 KB of LDS instead of 86 KB -- and the
// actor and the critic workgroup of a tile run side by side on one CU (or on two).  Per net the arithmetic is the same
// instruction sequence as in ppo_update_kernel (same k order, same reduction trees): results are bit-identical.
template <int H>
struct LayN {  // LDS layout in floats of one net
    static constexpr int ldh = H + 2;  // 66 = 2 * 33
    static constexpr int ldo = 18;     // logits(16) | value | pad
    int W1, W2, W3, B1, B2, B3, X, H1, H2, OUT, D3, D2, D1, total;
    __host__ __device__ explicit LayN(const Dims &d) {
        int o = 0;
        W1 = o; o += H * d.ld1;
        W2 = o; o += H * ldh;
        W3 = o; o += 16 * ldh;  // actor: [16][ldh], rows >= A zero; critic: the first H floats hold W3c
        B1 = o; o += H;
        B2 = o; o += H;
        B3 = o; o += 16;
        o = (o + 3) & ~3;
        X = o; o += R * d.ld1;
        H1 = o; o += R * ldh;
        H2 = o; o += R * ldh;
        OUT = o; o += R * ldo;
        D3 = o; o += R * ldo;
        D2 = o; o += R * ldh;
        D1 = o; o += R * ldh;
        total = o;
    }
};

// weights of one net from the padded image (csrc/adam.hip keeps it current): every piece is a contiguous, 16-B aligned
// run of the image; all loads are issued before the first LDS write (one memory round trip)
template <int H>
struct NetImageRegs {   // one net's weight image in flight: global loads issued by stage_net_image_load, LDS stores by _store
    static constexpr int kIt1 = (H * (16 * kMaxJ + 2) / 4 + NT - 1) / NT, kIt2 = (H * LayN<H>::ldh / 4 + NT - 1) / NT;
    static constexpr int kIt3 = (16 * LayN<H>::ldh / 4 + NT - 1) / NT;
    f4 r1[kIt1], r2[kIt2], r3[kIt3];
    float b1, b2, b3;
};

template <int H, int NET>
__device__ __forceinline__ void stage_net_image_load(NetImageRegs<H> &q, const Lay<H> &ly, const Dims &d, const float *__restrict__ img) {
    constexpr int kIt1 = NetImageRegs<H>::kIt1, kIt2 = NetImageRegs<H>::kIt2, kIt3 = NetImageRegs<H>::kIt3;
    const f4 *s1 = reinterpret_cast<const f4 *>(img + ly.W1 + NET * H * d.ld1);
    const f4 *s2 = reinterpret_cast<const f4 *>(img + (NET ? ly.W2c : ly.W2a));
    const f4 *s3 = reinterpret_cast<const f4 *>(img + (NET ? ly.W3c : ly.W3a));
    const int n1 = H * d.ld1 / 4, n2 = H * LayN<H>::ldh / 4, n3 = NET ? H / 4 : 16 * LayN<H>::ldh / 4;
    // (unconditional loads at clamped indices -- the stores are predicated: no branch, so no wait, between two loads)
#pragma unroll
    for (int it = 0; it < kIt1; ++it) { const int i = threadIdx.x + it * NT; q.r1[it] = s1[i < n1 ? i : 0]; }
#pragma unroll
    for (int it = 0; it < kIt2; ++it) { const int i = threadIdx.x + it * NT; q.r2[it] = s2[i < n2 ? i : 0]; }
#pragma unroll
    for (int it = 0; it < kIt3; ++it) { const int i = threadIdx.x + it * NT; q.r3[it] = s3[i < n3 ? i : 0]; }
    const int tb = threadIdx.x < H ? (int)threadIdx.x : 0, t3 = threadIdx.x < 16 ? (int)threadIdx.x : 0;
    q.b1 = img[ly.B1 + NET * H + tb];
    q.b2 = img[ly.B2 + NET * H + tb];
    const float b3v = img[NET ? ly.B3c : ly.B3a + t3];
    q.b3 = NET ? (threadIdx.x == 0 ? b3v : 0.f) : b3v;
}

template <int H, int NET>
__device__ __forceinline__ void stage_net_image_store(float *lds, const NetImageRegs<H> &q, const LayN<H> &ln, const Dims &d) {
    constexpr int kIt1 = NetImageRegs<H>::kIt1, kIt2 = NetImageRegs<H>::kIt2, kIt3 = NetImageRegs<H>::kIt3;
    const int n1 = H * d.ld1 / 4, n2 = H * LayN<H>::ldh / 4, n3 = NET ? H / 4 : 16 * LayN<H>::ldh / 4;
    f4 *d1 = reinterpret_cast<f4 *>(lds + ln.W1), *d2 = reinterpret_cast<f4 *>(lds + ln.W2);
    f4 *d3 = reinterpret_cast<f4 *>(lds + ln.W3);
#pragma unroll
    for (int it = 0; it < kIt1; ++it) { const int i = threadIdx.x + it * NT; if (i < n1) d1[i] = q.r1[it]; }
#pragma unroll
    for (int it = 0; it < kIt2; ++it) { const int i = threadIdx.x + it * NT; if (i < n2) d2[i] = q.r2[it]; }
#pragma unroll
    for (int it = 0; it < kIt3; ++it) { const int i = threadIdx.x + it * NT; if (i < n3) d3[i] = q.r3[it]; }
    if (threadIdx.x < H) { lds[ln.B1 + threadIdx.x] = q.b1; lds[ln.B2 + threadIdx.x] = q.b2; }
    if (threadIdx.x < 16) lds[ln.B3 + threadIdx.x] = q.b3;
}

template <int H, int NET>
__device__ __forceinline__ void stage_net_image(float *lds, const LayN<H> &ln, const Lay<H> &ly, const Dims &d,
                                                const float *__restrict__ img) {
    constexpr int kIt1 = (H * (16 * kMaxJ + 2) / 4 + NT - 1) / NT, kIt2 = (H * LayN<H>::ldh / 4 + NT - 1) / NT;
    constexpr int kIt3 = (16 * LayN<H>::ldh / 4 + NT - 1) / NT;
    const f4 *s1 = reinterpret_cast<const f4 *>(img + ly.W1 + NET * H * d.ld1);
    const f4 *s2 = reinterpret_cast<const f4 *>(img + (NET ? ly.W2c : ly.W2a));
    const f4 *s3 = reinterpret_cast<const f4 *>(img + (NET ? ly.W3c : ly.W3a));
    const int n1 = H * d.ld1 / 4, n2 = H * LayN<H>::ldh / 4, n3 = NET ? H / 4 : 16 * LayN<H>::ldh / 4;
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    f4 r1[kIt1], r2[kIt2], r3[kIt3];
#pragma unroll
    for (int it = 0; it < kIt1; ++it) { const int i = threadIdx.x + it * NT; r1[it] = i < n1 ? s1[i] : zero; }
#pragma unroll
    for (int it = 0; it < kIt2; ++it) { const int i = threadIdx.x + it * NT; r2[it] = i < n2 ? s2[i] : zero; }
#pragma unroll
    for (int it = 0; it < kIt3; ++it) { const int i = threadIdx.x + it * NT; r3[it] = i < n3 ? s3[i] : zero; }
    float b1 = 0.f, b2 = 0.f, b3 = 0.f;
    if (threadIdx.x < H) { b1 = img[ly.B1 + NET * H + threadIdx.x]; b2 = img[ly.B2 + NET * H + threadIdx.x]; }
    if (threadIdx.x < 16) b3 = NET ? (threadIdx.x == 0 ? img[ly.B3c] : 0.f) : img[ly.B3a + threadIdx.x];
    f4 *d1 = reinterpret_cast<f4 *>(lds + ln.W1), *d2 = reinterpret_cast<f4 *>(lds + ln.W2);
    f4 *d3 = reinterpret_cast<f4 *>(lds + ln.W3);
#pragma unroll
    for (int it = 0; it < kIt1; ++it) { const int i = threadIdx.x + it * NT; if (i < n1) d1[i] = r1[it]; }
#pragma unroll
    for (int it = 0; it < kIt2; ++it) { const int i = threadIdx.x + it * NT; if (i < n2) d2[i] = r2[it]; }
#pragma unroll
    for (int it = 0; it < kIt3; ++it) { const int i = threadIdx.x + it * NT; if (i < n3) d3[i] = r3[it]; }
    if (threadIdx.x < H) { lds[ln.B1 + threadIdx.x] = b1; lds[ln.B2 + threadIdx.x] = b2; }
    if (threadIdx.x < 16) lds[ln.B3 + threadIdx.x] = b3;
}

// ... or from the flat parameter vector (first gradient step of an update: the image may be stale)
template <int H, int NET>
__device__ void stage_net_flat(float *lds, const LayN<H> &ln, const Dims &d, const float *__restrict__ P) {
    const POff<H> po(d.D, d.A);
    const int oW1 = NET ? po.cW1 : po.aW1, oW2 = NET ? po.cW2 : po.aW2, oW3 = NET ? po.cW3 : po.aW3;
    const int ob1 = NET ? po.cb1 : po.ab1, ob2 = NET ? po.cb2 : po.ab2, ob3 = NET ? po.cb3 : po.ab3;
    constexpr int U = 4;
    for (int e0 = threadIdx.x; e0 < H * d.ld1; e0 += U * NT) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * NT, row = e / d.ld1, c = e - row * d.ld1;
            v[u] = (e < H * d.ld1 && c < d.D) ? P[oW1 + row * d.D + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) if (e0 + u * NT < H * d.ld1) lds[ln.W1 + e0 + u * NT] = v[u];
    }
    for (int e0 = threadIdx.x; e0 < H * ln.ldh; e0 += U * NT) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * NT, row = e / ln.ldh, c = e - row * ln.ldh;
            v[u] = (e < H * ln.ldh && c < H) ? P[oW2 + row * H + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) if (e0 + u * NT < H * ln.ldh) lds[ln.W2 + e0 + u * NT] = v[u];
    }
    for (int e = threadIdx.x; e < 16 * ln.ldh; e += NT) {
        const int row = e / ln.ldh, c = e - row * ln.ldh;
        float v = 0.f;
        if (NET) { if (e < H) v = P[oW3 + e]; }
        else if (row < d.A && c < H) v = P[oW3 + row * H + c];
        lds[ln.W3 + e] = v;
    }
    if (threadIdx.x < H) { lds[ln.B1 + threadIdx.x] = P[ob1 + threadIdx.x]; lds[ln.B2 + threadIdx.x] = P[ob2 + threadIdx.x]; }
    if (threadIdx.x < 16) lds[ln.B3 + threadIdx.x] = threadIdx.x < (NET ? 1 : d.A) ? P[ob3 + threadIdx.x] : 0.f;
}

// slab store flavour (diagnostics, tools/sweep_slab.py): 0 = non-temporal (default: consumed once, by the next kernel),
// 1 = plain write-back, 2 = agent-scope relaxed atomic store (sc1 write-through)
template <int ST>
__device__ __forceinline__ void slab_store(float v, float *p) {
    if (ST == 0) __builtin_nontemporal_store(v, p);
    else if (ST == 1) *p = v;
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// PUB (round 5 experiment, VERDICT r4 item 1: "kernel A" of a two-kernel gradient step whose second kernel would form the weight
// gradients as a K = M product and apply Adam): forward + loss + input-gradient backward only -- no weight-gradient MFMAs, no
// bias sums, no slab; instead the tile's H1, H2, dH2, dH1 (16 x 64 each), X and d loss / d out go to `slabs` as 16-byte rows
// (19.7 KB per tile and net instead of a 22.3 KB slab).  Reachable only through tsm_debug_set_update_variant(2): it produces
// no gradient, it exists to TIME kernel A (tools/ab_kernel_a.py, profiles/r05_ab_kernel_a.txt).
template <int H>
constexpr int pub_floats(int ld1) { return 4 * R * H + R * ld1 + R * LayN<H>::ldo; }

// sum over the R tile rows of one column (row pitch `ld`), rows in increasing order -- s = 0; s += v[0]; ... -- with all R reads issued
// before the first addition (the plain loop read a few values, waited, added: 0.3-0.4 us per bias vector on wave 0, which the other
// waves then waited for at the phase's barrier)
__device__ __forceinline__ float column_sum(const float *col, int ld) {
    float v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = col[r * ld];
    __builtin_amdgcn_sched_barrier(0);
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) s += v[r];
    return s;
}

template <int H, int NET, int ST, bool PUB = false>
__device__ __forceinline__ void ppo_update_net(
    float *lds, const float *__restrict__ P, const float *__restrict__ img, const Dims &d, const float *__restrict__ obs,
    const int32_t *__restrict__ act, const float *__restrict__ logp_old, const float *__restrict__ adv,
    const float *__restrict__ returns, const float *__restrict__ v_s_old, const int64_t *__restrict__ perm, int64_t first,
    int64_t M, const float *__restrict__ adv_stats, const LossCfg &cfg, float *__restrict__ slabs,
    double *__restrict__ loss_partial, long long *stamps) {
    // diagnostics (tsm_debug_set_stamps, >= 2048 slots): [16 + 24 NET + k] = phase k of workgroup (0, NET);
    // [1024 + 2 (b + 256 NET)], [.. + 1] = start / end of workgroup (b, NET)   (tools/stamp_update.py)
#define NSTAMP(k) do { if (stamps && blockIdx.x == 0 && threadIdx.x == 0) stamps[16 + 24 * NET + (k)] = (long long)wall_clock64(); } while (0)
    NSTAMP(0);
    if (stamps && threadIdx.x == 0 && blockIdx.x < 256) stamps[1024 + 2 * (blockIdx.x + 256 * NET)] = (long long)wall_clock64();
    const LayN<H> ln(d);
    const POff<H> po(d.D, d.A);
    const int64_t n_tiles = (M + R - 1) / R;
    float xr[kXRegs];
    struct RowIn { int a_idx; float adv, logp_old, ret, v_old; } rin;
    // (row ids: an unconditional load at a clamped index + a select, see prefetch_tile_ids)
    auto row_id = [&](int64_t row0) -> int64_t {
        const int64_t i = row0 + (threadIdx.x >> 4);
        const int64_t v = perm ? perm[i < M ? i : row0] : first + i;
        return i < M ? v : -1;
    };
    auto prefetch_row = [&](int64_t src) {   // (unconditional loads at a clamped row + selects: one batch)
        const int64_t s0 = src >= 0 ? src : 0;
        const bool on = src >= 0;
        rin.a_idx = 0; rin.adv = 0.f; rin.logp_old = 0.f; rin.ret = 0.f; rin.v_old = 0.f;
        if (NET == 0) {
            const int a_ = act[s0]; const float ad_ = adv[s0], lp_ = logp_old[s0];
            rin.a_idx = on ? a_ : 0; rin.adv = on ? ad_ : 0.f; rin.logp_old = on ? lp_ : 0.f;
        } else {
            const float r_ = returns[s0];
            rin.ret = on ? r_ : 0.f;
            const float vo_ = (cfg.value_clip ? v_s_old : returns)[s0];   // (no branch around a load: its wait would be for everything)
            rin.v_old = on && cfg.value_clip ? vo_ : 0.f;
        }
    };
    // Two dependent memory round trips in front of the first tile (round 5; five before): (1) the row ids AND the weight image,
    // which depends on nothing, in one batch; (2) the id-dependent gathers.  Loads return in issue order, so the ids -- issued
    // first -- are usable while the image is still in flight.
    // The common case -- a permutation, the weight image, a first tile for every workgroup -- as ONE straight-line sequence: any
    // branch between a load and its first use makes hipcc's wait conservative (it waited for the image before it used the ids).
    if (perm && img && (int64_t)blockIdx.x < n_tiles) {
        int64_t xs[kXRegs], id0[kXRegs + 1];
        prefetch_tile_ids_load<1>(id0, d, perm, first, (int64_t)blockIdx.x * R, M);
        NetImageRegs<H> wq;
        const Lay<H> ly_img(d, true);
        stage_net_image_load<H, NET>(wq, ly_img, d, img);
        __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks the gathers below the image's LDS stores: a third round trip)
        unsigned ok;
        const int64_t rsrc = prefetch_tile_ids_done_nb(xs, ok, id0, d, (int64_t)blockIdx.x * R, M);
        prefetch_tile_vals_nb(xr, xs, ok, obs);
        prefetch_row(rsrc);
        __builtin_amdgcn_sched_barrier(0);
        stage_net_image_store<H, NET>(lds, wq, ln, d);
    } else {
        int64_t xs[kXRegs];
        if ((int64_t)blockIdx.x < n_tiles) {
            prefetch_tile_ids(xs, d, perm, first, (int64_t)blockIdx.x * R, M);
            const int64_t rsrc = row_id((int64_t)blockIdx.x * R);
            prefetch_tile_vals(xr, xs, obs);
            prefetch_row(rsrc);
        }
        if (img) { const Lay<H> ly(d, true); stage_net_image<H, NET>(lds, ln, ly, d, img); }
        else stage_net_flat<H, NET>(lds, ln, d, P);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const float invM = 1.0f / (float)M;
    float a_mean = 0.f, a_std = 1.f;
    if (NET == 0 && cfg.adv_norm) { a_mean = adv_stats[0]; a_std = adv_stats[1]; }

    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    f4 g_W3 = zero;             // actor: rows i (A pad 16) x cols 16w..
    f4 g_W2[4], g_W1[kMaxJ];    // rows 16w.. x col block jb
#pragma unroll
    for (int j = 0; j < 4; ++j) g_W2[j] = zero;
#pragma unroll
    for (int j = 0; j < kMaxJ; ++j) g_W1[j] = zero;
    float g_b1 = 0.f, g_b2 = 0.f;  // threads 0..H-1
    float g_W3c = 0.f;             // critic, threads 0..H-1
    float g_b3 = 0.f;              // actor: threads 0..A-1; critic: thread 0
    double s_a = 0.0, s_b = 0.0;   // threads 16*r: actor (clip objective, entropy) | critic (value loss, -)

    // The workgroup's gradient slab, a piece at a time: each piece is stored as soon as the LAST tile has added to it (round 5) -- the
    // output layer's after its phase, layer 2's after its phase, layer 1's at the end -- instead of all 22 KB behind the backward pass:
    // the stores fly under the remaining phases and the launch ends with a quarter of the bytes still on their way.
    float *const S = slabs + (int64_t)blockIdx.x * po.total;
    auto store_out_layer = [&]() {
        if constexpr (!PUB) {
            if (NET == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = kq * 4 + r;
                    if (row < d.A) slab_store<ST>(g_W3[r], &S[po.aW3 + row * H + 16 * w + r16]);
                }
                if (threadIdx.x < d.A) slab_store<ST>(g_b3, &S[po.ab3 + threadIdx.x]);
            } else {
                if (threadIdx.x < H) slab_store<ST>(g_W3c, &S[po.cW3 + threadIdx.x]);
                if (threadIdx.x == 0) slab_store<ST>(g_b3, &S[po.cb3]);
            }
        }
    };
    auto store_layer2 = [&]() {
        if constexpr (!PUB) {
            const int oW2 = NET ? po.cW2 : po.aW2;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) slab_store<ST>(g_W2[jb][r], &S[oW2 + (16 * w + kq * 4 + r) * H + 16 * jb + r16]);
            if (threadIdx.x < H) slab_store<ST>(g_b2, &S[(NET ? po.cb2 : po.ab2) + threadIdx.x]);
        }
    };
    auto store_layer1 = [&]() {
        if constexpr (!PUB) {
            const int oW1 = NET ? po.cW1 : po.aW1;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int jb = 0; jb < kMaxJ; ++jb) {
                    const int c = 16 * jb + r16;
                    if (jb < d.nJ && c < d.D) slab_store<ST>(g_W1[jb][r], &S[oW1 + (16 * w + kq * 4 + r) * d.D + c]);
                }
            if (threadIdx.x < H) slab_store<ST>(g_b1, &S[(NET ? po.cb1 : po.ab1) + threadIdx.x]);
        }
    };
    bool stored = false;   // (a workgroup without a tile -- grid larger than the tile count -- stores its zeros at the end)
    NSTAMP(1);
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int64_t row0 = t * R;
        const bool last_tile = t + gridDim.x >= n_tiles;
        __syncthreads();
        commit_tile_x(lds, ln.X, d, xr);
        const RowIn cur = rin;
        __syncthreads();
        NSTAMP(2);
        // ---- forward ----
        {
            f4 acc = zero;
            const float *xa = lds + ln.X + r16 * d.ld1 + kq;
            const float *wa = lds + ln.W1 + (16 * w + r16) * d.ld1 + kq;
            for (int k0 = 0; k0 < d.Kp1; k0 += 4) acc = mfma(xa[k0], wa[k0], acc);
            const int col = 16 * w + r16;
            const float b = lds[ln.B1 + col];
#pragma unroll
            for (int r = 0; r < 4; ++r) lds[ln.H1 + (kq * 4 + r) * ln.ldh + col] = fmaxf(acc[r] + b, 0.f);
        }
        __syncthreads();
        NSTAMP(3);
        {
            f4 acc = zero;
            const float *ha = lds + ln.H1 + r16 * ln.ldh + kq;
            const float *wa = lds + ln.W2 + (16 * w + r16) * ln.ldh + kq;
#pragma unroll
            for (int k0 = 0; k0 < H; k0 += 4) acc = mfma(ha[k0], wa[k0], acc);
            const int col = 16 * w + r16;
            const float b = lds[ln.B2 + col];
#pragma unroll
            for (int r = 0; r < 4; ++r) lds[ln.H2 + (kq * 4 + r) * ln.ldh + col] = fmaxf(acc[r] + b, 0.f);
        }
        __syncthreads();
        NSTAMP(4);
        if (NET == 0) {
            if (w == 0) {
                f4 acc = zero;
                const float *ha = lds + ln.H2 + r16 * ln.ldh + kq;
                const float *wa = lds + ln.W3 + r16 * ln.ldh + kq;
#pragma unroll
                for (int k0 = 0; k0 < H; k0 += 4) acc = mfma(ha[k0], wa[k0], acc);
                const float b = lds[ln.B3 + r16];
#pragma unroll
                for (int r = 0; r < 4; ++r) lds[ln.OUT + (kq * 4 + r) * ln.ldo + r16] = acc[r] + b;
            }
        } else if (threadIdx.x < R) {
            const float sv = fmaf_chain_ahead<H>(lds + ln.H2 + threadIdx.x * ln.ldh, lds + ln.W3);
            lds[ln.OUT + threadIdx.x * ln.ldo + 16] = sv + lds[ln.B3];
        }
        __syncthreads();
        NSTAMP(5);

        // ---- loss head: 16 lanes per row ----
        {
            const int r = threadIdx.x >> 4, j = threadIdx.x & 15;
            const int64_t i = row0 + r;
            float *d3 = lds + ln.D3 + r * ln.ldo;
            float dl = 0.f, dv = 0.f;
            if (i < M) {
                const float *lg = lds + ln.OUT + r * ln.ldo;
                if (NET == 0) {
                    const bool on = j < d.A;
                    const float x = on ? lg[j] : -INFINITY;
                    const float m = row16_max(x);
                    const float ex = on ? expf(x - m) : 0.f;
                    const float s = row16_sum(ex);
                    const float l = on ? x - (m + logf(s)) : 0.f;
                    const float p = ex / s;
                    const float h = row16_sum(on ? -p * l : 0.f);
                    const int a_idx = cur.a_idx;
                    const float logp = __shfl(l, (threadIdx.x & 48) + a_idx, 64);
                    float a = cur.adv;
                    if (cfg.adv_norm) a = (a - a_mean) / (a_std + 1e-8f);
                    float ratio, obj, g_ratio;
                    if (cfg.kind == 1) {
                        ratio = 1.f; obj = logp * a; g_ratio = a;
                    } else {
                        ratio = expf(logp - cur.logp_old);
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
                    const float g_logp = -g_ratio * ratio * invM, ec = cfg.ent_coef * invM;
                    if (on) dl = g_logp * ((j == a_idx ? 1.f : 0.f) - p) + ec * p * (l + h);
                    if (j == 0) { s_a += (double)obj; s_b += (double)h; }
                } else if (j == 0) {
                    const float v = lg[16], ret = cur.ret;
                    float vf, g_v;
                    if (cfg.value_clip) {
                        const float vs = cur.v_old;
                        const float dd = v - vs;
                        const float dc = fminf(fmaxf(dd, -cfg.eps_clip), cfg.eps_clip);
                        const bool v_in = dd >= -cfg.eps_clip && dd <= cfg.eps_clip;
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
                    dv = cfg.vf_coef * g_v * invM;
                    s_a += (double)vf;
                }
            }
            if (NET == 0) d3[j] = dl;
            else if (j == 0) d3[16] = dv;
        }
        __syncthreads();
        NSTAMP(6);

        if (t + gridDim.x < n_tiles) {  // next tile's gathers fly under this tile's backward pass
            prefetch_tile_x(xr, d, obs, perm, first, (t + gridDim.x) * R, M);
            prefetch_row(row_id((t + gridDim.x) * R));
        }
        // ---- output-layer gradients + dh2 ----
        if (NET == 0) {
            const float *dA = lds + ln.D3 + kq * ln.ldo + r16;
            const float *hB = lds + ln.H2 + kq * ln.ldh + 16 * w + r16;
            if constexpr (!PUB) {
#pragma unroll
                for (int k0 = 0; k0 < R; k0 += 4) g_W3 = mfma(dA[k0 * ln.ldo], hB[k0 * ln.ldh], g_W3);
            }
            f4 acc = zero;
            const float *dR = lds + ln.D3 + r16 * ln.ldo + kq;
            const float *wB = lds + ln.W3 + kq * ln.ldh + 16 * w + r16;
#pragma unroll
            for (int k0 = 0; k0 < 16; k0 += 4) acc = mfma(dR[k0], wB[k0 * ln.ldh], acc);
            const int col = 16 * w + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = kq * 4 + r;
                lds[ln.D2 + row * ln.ldh + col] = lds[ln.H2 + row * ln.ldh + col] > 0.f ? acc[r] : 0.f;
            }
            if (!PUB && threadIdx.x < 16) g_b3 += column_sum(lds + ln.D3 + threadIdx.x, ln.ldo);
        } else {
            // (round 5: every LDS read of the phase is issued before the first use.  Written as three loops -- dH2 elements, the W3
            //  column chain, the b3 sum -- each iteration read, waited and computed: 1.2-1.3 us of the critic workgroup, which is the
            //  slower of the two nets, tools/stamp_update.py.  Thread (w, j = tid & 63) owns dH2 rows w, w + 4, w + 8, w + 12 of column j;
            //  wave 0 the W3 chain of column j over all 16 rows, in row order as before; thread 0 the b3 sum.)
            static_assert(NT == 4 * H && R == 16, "one column per lane, four rows per wave");
            const int j = threadIdx.x & (H - 1);
            float d4[4], h4[4], dv[R], hcol[R];
#pragma unroll
            for (int k = 0; k < 4; ++k) { d4[k] = lds[ln.D3 + (4 * k + w) * ln.ldo + 16]; h4[k] = lds[ln.H2 + (4 * k + w) * ln.ldh + j]; }
            const float w3j = lds[ln.W3 + j];
            if (!PUB && w == 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) { dv[r] = lds[ln.D3 + r * ln.ldo + 16]; hcol[r] = lds[ln.H2 + r * ln.ldh + j]; }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 4; ++k) lds[ln.D2 + (4 * k + w) * ln.ldh + j] = h4[k] > 0.f ? d4[k] * w3j : 0.f;
            if (!PUB && w == 0) {
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < R; ++r) s = fmaf(dv[r], hcol[r], s);
                g_W3c += s;
                if (threadIdx.x == 0) {
                    float sb = 0.f;
#pragma unroll
                    for (int r = 0; r < R; ++r) sb += dv[r];
                    g_b3 += sb;
                }
            }
        }
        if (last_tile) { store_out_layer(); stored = true; }
        __syncthreads();
        NSTAMP(7);
        // ---- hidden layer 2 gradients + dh1 ----
        {
            const float *dA = lds + ln.D2 + kq * ln.ldh + 16 * w + r16;
            if constexpr (!PUB) {
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    const float *hB = lds + ln.H1 + kq * ln.ldh + 16 * jb + r16;
#pragma unroll
                    for (int k0 = 0; k0 < R; k0 += 4) g_W2[jb] = mfma(dA[k0 * ln.ldh], hB[k0 * ln.ldh], g_W2[jb]);
                }
                if (threadIdx.x < H) g_b2 += column_sum(lds + ln.D2 + threadIdx.x, ln.ldh);
                if (last_tile) store_layer2();   // (before the dH1 chain: the stores fly under its 16 dependent MFMAs)
            }
            f4 acc = zero;
            const float *dR = lds + ln.D2 + r16 * ln.ldh + kq;
            const float *wB = lds + ln.W2 + kq * ln.ldh + 16 * w + r16;
#pragma unroll
            for (int k0 = 0; k0 < H; k0 += 4) acc = mfma(dR[k0], wB[k0 * ln.ldh], acc);
            const int col = 16 * w + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = kq * 4 + r;
                lds[ln.D1 + row * ln.ldh + col] = lds[ln.H1 + row * ln.ldh + col] > 0.f ? acc[r] : 0.f;
            }
        }
        __syncthreads();
        NSTAMP(8);
        if constexpr (PUB) {   // ---- publish the tile instead of layer-1 gradients (rows as 16-byte pieces: 8-byte aligned in LDS) ----
            float *pub = slabs + ((int64_t)t * 2 + NET) * pub_floats<H>(d.ld1);
            const int pr = threadIdx.x >> 4, pc = (threadIdx.x & 15) * 4;
            const int src[4] = {ln.H1, ln.H2, ln.D2, ln.D1};
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float2 a0 = *reinterpret_cast<const float2 *>(lds + src[m] + pr * ln.ldh + pc);
                const float2 a1 = *reinterpret_cast<const float2 *>(lds + src[m] + pr * ln.ldh + pc + 2);
                __builtin_nontemporal_store(f4{a0.x, a0.y, a1.x, a1.y}, reinterpret_cast<f4 *>(pub + m * R * H + pr * H + pc));
            }
            for (int e = threadIdx.x; e < R * d.ld1; e += NT) __builtin_nontemporal_store(lds[ln.X + e], pub + 4 * R * H + e);
            for (int e = threadIdx.x; e < R * ln.ldo; e += NT) __builtin_nontemporal_store(lds[ln.D3 + e], pub + 4 * R * H + R * d.ld1 + e);
        } else
        // ---- layer 1 gradients ----
        {
            const float *dA = lds + ln.D1 + kq * ln.ldh + 16 * w + r16;
#pragma unroll
            for (int jb = 0; jb < kMaxJ; ++jb) {
                if (jb < d.nJ) {
                    const float *xB = lds + ln.X + kq * d.ld1 + 16 * jb + r16;
#pragma unroll
                    for (int k0 = 0; k0 < R; k0 += 4) g_W1[jb] = mfma(dA[k0 * ln.ldh], xB[k0 * d.ld1], g_W1[jb]);
                }
            }
            if (threadIdx.x < H) g_b1 += column_sum(lds + ln.D1 + threadIdx.x, ln.ldh);
        }
    }

    NSTAMP(9);
    // ---- what is left of this net's half of the workgroup's gradient slab ----
    if (!stored) { store_out_layer(); store_layer2(); }
    store_layer1();
    NSTAMP(10);
    // loss partial sums: actor -> clip objective [0], entropy [2]; critic -> value loss [1]
    {
        __shared__ double s_red[2][NT / 64];
        const double a = wave_sum(s_a), b = wave_sum(s_b);
        if (lane == 0) { s_red[0][w] = a; s_red[1][w] = b; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double ta = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
            const double tb = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
            if (NET == 0) {
                loss_partial[4 * blockIdx.x + 0] = ta;
                loss_partial[4 * blockIdx.x + 2] = tb;
                loss_partial[4 * blockIdx.x + 3] = 0.0;
            } else {
                loss_partial[4 * blockIdx.x + 1] = ta;
            }
            NSTAMP(11);
            if (stamps && blockIdx.x < 256) stamps[1025 + 2 * (blockIdx.x + 256 * NET)] = (long long)wall_clock64();
        }
    }
#undef NSTAMP
}

// DS: 0 = dimensions from the launch arguments; else the observation width this instantiation is compiled for (5 actions): the
// headline job's 18 -- its index arithmetic (divisions by ld1 and D, layout offsets) then costs constants instead of ~35-instruction
// integer divisions and scalar registers.
template <int H, int ST, int DS, bool PUB = false>
__global__ __launch_bounds__(NT, 2) void ppo_update_split_kernel(   // (2: waves per SIMD at least -- both nets' workgroups of a tile share a CU)

    const float *__restrict__ P, const float *__restrict__ img, Dims d_arg, const float *__restrict__ obs,
    const int32_t *__restrict__ act, const float *__restrict__ logp_old, const float *__restrict__ adv,
    const float *__restrict__ returns, const float *__restrict__ v_s_old, const int64_t *__restrict__ perm, int64_t first,
    int64_t M, const float *__restrict__ adv_stats, LossCfg cfg, float *__restrict__ slabs,
    double *__restrict__ loss_partial, int64_t *__restrict__ opt_step_dev, long long *stamps) {
    extern __shared__ float lds[];
    const Dims d = DS ? dims_const(DS, 5) : d_arg;
    // device-resident optimizer step count (hipGraph replay): bumped here, read by the Adam kernel that follows
    if (opt_step_dev && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *opt_step_dev += 1;
    if (blockIdx.y == 0)
        ppo_update_net<H, 0, ST, PUB>(lds, P, img, d, obs, act, logp_old, adv, returns, v_s_old, perm, first, M, adv_stats, cfg, slabs,
                                      loss_partial, stamps);
    else
        ppo_update_net<H, 1, ST, PUB>(lds, P, img, d, obs, act, logp_old, adv, returns, v_s_old, perm, first, M, adv_stats, cfg, slabs,
                                      loss_partial, stamps);
}

__global__ __launch_bounds__(256) void update_finalize_kernel(const double *__restrict__ partial, int n_blocks,
                                                              int64_t M, float vf_coef, float ent_coef,
                                                              float *__restrict__ scalars) {
    __shared__ double sm[256 / 64];
    double c = 0.0, v = 0.0, e = 0.0;
    for (int b = threadIdx.x; b < n_blocks; b += 256) {
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

// one workgroup per gradient step: scalars[k] from partial[k][0..n_blocks[k])
__global__ __launch_bounds__(256) void update_finalize_many_kernel(const double *__restrict__ partial,
                                                                   int64_t stride_elems,
                                                                   const int32_t *__restrict__ n_blocks,
                                                                   const int64_t *__restrict__ M, float vf_coef,
                                                                   float ent_coef, float *__restrict__ scalars) {
    __shared__ double sm[256 / 64];
    const int k = blockIdx.x;
    const double *p = partial + (int64_t)k * stride_elems;
    double c = 0.0, v = 0.0, e = 0.0;
    for (int b = threadIdx.x; b < n_blocks[k]; b += 256) {
        c += p[4 * b + 0];
        v += p[4 * b + 1];
        e += p[4 * b + 2];
    }
    c = block_sum<double, 256>(c, sm);
    v = block_sum<double, 256>(v, sm);
    e = block_sum<double, 256>(e, sm);
    if (threadIdx.x == 0) {
        const double m = (double)M[k];
        const double clip_loss = -c / m, vf_loss = v / m, ent_loss = e / m;
        scalars[4 * k + 0] = (float)(clip_loss + (double)vf_coef * vf_loss - (double)ent_coef * ent_loss);
        scalars[4 * k + 1] = (float)clip_loss;
        scalars[4 * k + 2] = (float)vf_loss;
        scalars[4 * k + 3] = (float)ent_loss;
    }
}

}  // namespace

TSM_EXPORT int64_t tsm_policy_param_count(int32_t obs_dim, int32_t hidden, int32_t n_act) {
    if (hidden != 64) return -1;
    return POff<64>(obs_dim, n_act).total;
}

TSM_EXPORT int64_t tsm_policy_image_elems(int32_t obs_dim, int32_t hidden, int32_t n_act) {
    Dims d;
    if (make_dims(obs_dim, hidden, n_act, &d)) return -1;
    const Lay<64> ly(d, false);
    const int64_t padded = (int64_t)image_f4_padded(ly.X) * 4;
    // the staging copy writes `padded` floats into LDS: must stay inside the smallest kernel layout
    return padded <= ly.total ? padded : -1;
}

// map_out_host[i] = offset of flat parameter i inside the padded image (tsm_policy_param_count entries)
TSM_EXPORT int tsm_policy_image_map(int32_t obs_dim, int32_t hidden, int32_t n_act, int32_t *map_out_host) {
    Dims d;
    if (int rc = make_dims(obs_dim, hidden, n_act, &d)) return rc;
    TSM_REQUIRE(map_out_host, "tsm_policy_image_map: null pointer");
    constexpr int H = 64;
    const Lay<H> ly(d, false);
    const POff<H> po(d.D, d.A);
    int32_t *m = map_out_host;
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < d.D; ++c) {
            m[po.aW1 + r * d.D + c] = ly.W1 + r * d.ld1 + c;
            m[po.cW1 + r * d.D + c] = ly.W1 + (H + r) * d.ld1 + c;
        }
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < H; ++c) {
            m[po.aW2 + r * H + c] = ly.W2a + r * ly.ldh + c;
            m[po.cW2 + r * H + c] = ly.W2c + r * ly.ldh + c;
        }
    for (int r = 0; r < d.A; ++r)
        for (int c = 0; c < H; ++c) m[po.aW3 + r * H + c] = ly.W3a + r * ly.ldh + c;
    for (int c = 0; c < H; ++c) {
        m[po.cW3 + c] = ly.W3c + c;
        m[po.ab1 + c] = ly.B1 + c;
        m[po.cb1 + c] = ly.B1 + H + c;
        m[po.ab2 + c] = ly.B2 + c;
        m[po.cb2 + c] = ly.B2 + H + c;
    }
    for (int c = 0; c < d.A; ++c) m[po.ab3 + c] = ly.B3a + c;
    m[po.cb3] = ly.B3c;
    return TSM_OK;
}

TSM_EXPORT int tsm_policy_forward(const float *params, const float *param_image, int32_t obs_dim, int32_t hidden,
                                  int32_t n_act,
                                  const float *obs, int64_t B, int mode, uint64_t seed, uint64_t offset,
                                  const uint64_t *offset_dev, float *logits_out, float *value_out, int32_t *act_io, float *logp_out,
                                  void *stream) {
    Dims d;
    if (int rc = make_dims(obs_dim, hidden, n_act, &d)) return rc;
    TSM_REQUIRE(B >= 0 && mode >= 0 && mode <= 3, "tsm_policy_forward: bad B / mode");
    if (B == 0) return TSM_OK;
    TSM_REQUIRE(params && obs, "tsm_policy_forward: null pointer");
    TSM_REQUIRE(mode == 0 || act_io, "tsm_policy_forward: mode %d needs act_io", mode);
    const Lay<64> ly(d, false);
    const size_t shmem = (size_t)ly.total * sizeof(float);
    const int64_t n_tiles = ceil_div(B, R);
    // one resident round of workgroups (two per CU by their LDS): measured 12 800 rows 19.8 -> 13.5 us, 76 800 rows 42.4 -> 35.3 us
    // against the former cap of 1024 (round 4, with a grid-cap probe since removed; caps of 256 / 384 / 768 are slower at every size)
    unsigned grid = (unsigned)(n_tiles < 512 ? n_tiles : 512);
    static bool attr_set = false;
    if (!attr_set) {
        TSM_HIP(tsm_allow_max_lds(reinterpret_cast<const void *>(policy_forward_kernel<64>)));
        TSM_HIP(tsm_allow_max_lds(reinterpret_cast<const void *>(policy_forward_kernel<64, 18>)));
        TSM_HIP(tsm_allow_max_lds(reinterpret_cast<const void *>(policy_forward_kernel<64, 16>)));
        attr_set = true;
    }
    const bool spec = d.A == 5 && !tsm_opt(TSM_OPT_GENERIC);
    auto kern = (spec && d.D == 18) ? policy_forward_kernel<64, 18> : (spec && d.D == 16) ? policy_forward_kernel<64, 16> : policy_forward_kernel<64>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), shmem, tsm_stream(stream), params,
                       param_image, d, obs,
                       B, seed, offset, offset_dev, mode, logits_out, value_out, act_io, logp_out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

// diagnostics (not in the public header): 0 = one net per workgroup (default), 1 = both nets in one workgroup
static int g_update_variant = 0;
static int g_slab_store = 0;
extern "C" __attribute__((visibility("default"))) void tsm_debug_set_slab_store(int v) { g_slab_store = v; }
extern "C" __attribute__((visibility("default"))) void tsm_debug_set_update_variant(int v) { g_update_variant = v; }
extern "C" int tsm_debug_update_variant_get(void) { return g_update_variant; }
extern "C" int tsm_debug_slab_store_get(void) { return g_slab_store; }

TSM_EXPORT int tsm_ppo_update_grid(int64_t M, int32_t max_blocks) {
    // One-net workgroups need 52 KB of LDS: three fit on a CU (768 on the chip).  The rule comes from sweeps of the
    // gradient step (update + Adam, us) over the slab count n_blocks -- every slab is 45 KB written here and read back
    // by Adam, so fewer slabs pay for a second tile per workgroup once the chip is full:
    //   tiles  128: 14.8 @128 | 18.3 @64        256: 19.8 @256 | 19.2 @128 | 25.2 @86 | 27.0 @64 (tools/sweep_slab.py)
    //          320: 22.0 @160 | 22.2 @320 | 24.9 @107                       384: 22.4 @192 | 25.1 @384
    //          512: 24.8 @256 | 25.5 @384     1024: 34.3 @384 | 35.7 @256 | 44.7 @512     4096: 88 @384 | 103 @256 | 112 @512
    // 256 tiles (the headline minibatch of 4096 rows): 128 pairs of 2 tiles halve the slab traffic (5.7 MB written + 5.7 MB
    // read back per gradient step instead of 11.4 + 11.4) at the same isolated step time, but inside the update graph the
    // job is not faster (0.419 vs 0.418 ms per update): the fused kernel gets 1.1 us longer, Adam 1.1 us shorter.  Kept at
    // one tile per pair (profiles/r02_sweep_slab.txt).
    const int64_t n_tiles = ceil_div(M > 0 ? M : 1, R);
    int64_t g;
    if (n_tiles <= 256) g = n_tiles;
    else if (n_tiles < 1024) g = (n_tiles + 1) / 2 < 256 ? (n_tiles + 1) / 2 : 256;
    else g = 384;
    if (max_blocks > 0 && g > max_blocks) g = max_blocks;
    return (int)g;
}

TSM_EXPORT int tsm_ppo_update_fused(const float *params, const float *param_image, int32_t obs_dim, int32_t hidden,
                                    int32_t n_act,
                                    const float *obs, const int32_t *act, const float *logp_old, const float *adv,
                                    const float *returns, const float *v_s_old, const int64_t *perm,
                                    int64_t first_row, int64_t M, const float *adv_stats,
                                    const tsm_ppo_cfg *cfg_host, int32_t n_blocks, float *grad_slabs_out,
                                    double *loss_partial_out, float *scalars_out, int64_t *opt_step_dev,
                                    void *stream) {
    Dims d;
    if (int rc = make_dims(obs_dim, hidden, n_act, &d)) return rc;
    TSM_REQUIRE(M >= 1, "tsm_ppo_update_fused: M must be >= 1");
    TSM_REQUIRE(cfg_host && params && obs && act && logp_old && adv && returns && grad_slabs_out && loss_partial_out,
                "tsm_ppo_update_fused: null pointer");
    TSM_REQUIRE(!cfg_host->value_clip || v_s_old, "tsm_ppo_update_fused: value_clip needs v_s_old");
    TSM_REQUIRE(!cfg_host->adv_norm || adv_stats, "tsm_ppo_update_fused: adv_norm needs adv_stats");
    TSM_REQUIRE(cfg_host->loss_kind == 0 || cfg_host->loss_kind == 1, "tsm_ppo_update_fused: loss_kind must be 0 or 1");
    TSM_REQUIRE(cfg_host->dual_clip <= 0.0 || cfg_host->dual_clip > 1.0,
                "Dual-clip PPO parameter should greater than 1.0 but got %g", cfg_host->dual_clip);
    TSM_REQUIRE(n_blocks >= 1 && n_blocks <= ceil_div(M, R), "tsm_ppo_update_fused: n_blocks=%d out of range",
                n_blocks);
    LossCfg cfg;
    cfg.eps_clip = (float)cfg_host->eps_clip;
    cfg.dual_clip = (float)cfg_host->dual_clip;
    cfg.vf_coef = (float)cfg_host->vf_coef;
    cfg.ent_coef = (float)cfg_host->ent_coef;
    cfg.value_clip = cfg_host->value_clip;
    cfg.adv_norm = cfg_host->adv_norm;
    cfg.kind = cfg_host->loss_kind;
    const Lay<64> ly(d, true);
    const size_t shmem = (size_t)ly.total * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        // the kernel also has a few hundred bytes of static LDS: leave headroom below the 160 KiB of a CU
        TSM_HIP(tsm_allow_max_lds(reinterpret_cast<const void *>(ppo_update_kernel<64>)));
        attr_set = true;
    }
    hipStream_t st = tsm_stream(stream);
    if (g_update_variant == 2) {   // (experiment: kernel A alone, see ppo_update_net<..., PUB>; no gradient comes out of it)
        TSM_REQUIRE(d.D == 18 && d.A == 5 && (int64_t)n_blocks == ceil_div(M, R) && param_image,
                    "update variant 2 (kernel A timing) serves the headline shape only: obs 18, 5 actions, one tile per workgroup, image");
        const LayN<64> ln(d);
        hipLaunchKernelGGL((ppo_update_split_kernel<64, 0, 18, true>), dim3((unsigned)n_blocks, 2), dim3(NT),
                           (size_t)ln.total * sizeof(float), st, params, param_image, d, obs, act, logp_old, adv, returns,
                           v_s_old, perm, first_row, M, adv_stats, cfg, grad_slabs_out, loss_partial_out, opt_step_dev, g_tsm_stamps);
    } else if (g_update_variant == 0) {
        // one net per workgroup (grid.y = actor | critic): see ppo_update_split_kernel
        const LayN<64> ln(d);
        const bool spec = d.A == 5 && !tsm_opt(TSM_OPT_GENERIC);   // ("generic_kernels": the generic instantiation)
        auto kern = g_slab_store == 1 ? ppo_update_split_kernel<64, 1, 0>
                    : g_slab_store == 2 ? ppo_update_split_kernel<64, 2, 0>
                    : (spec && d.D == 18) ? ppo_update_split_kernel<64, 0, 18>     // BASELINE configs[1] (simple_spread, N = 3)
                    : (spec && d.D == 16) ? ppo_update_split_kernel<64, 0, 16>     // configs[4] (simple_tag 3 v 1)
                    : ppo_update_split_kernel<64, 0, 0>;
        hipLaunchKernelGGL(kern, dim3((unsigned)n_blocks, 2), dim3(NT),
                           (size_t)ln.total * sizeof(float), st, params, param_image, d, obs, act, logp_old, adv, returns,
                           v_s_old, perm, first_row, M, adv_stats, cfg, grad_slabs_out, loss_partial_out, opt_step_dev, g_tsm_stamps);
    } else {  // both nets in one workgroup (update variant 1: an A/B reference, tools/ab_update_variant.py)
        hipLaunchKernelGGL((ppo_update_kernel<64>), dim3((unsigned)n_blocks), dim3(NT), shmem, st, params, param_image, d,
                           obs, act,
                           logp_old, adv, returns, v_s_old, perm, first_row, M, adv_stats, cfg, grad_slabs_out,
                           loss_partial_out, opt_step_dev, g_tsm_stamps);
    }
    TSM_LAUNCH_CHECK();
    if (scalars_out) {
        hipLaunchKernelGGL(update_finalize_kernel, dim3(1), dim3(256), 0, st, loss_partial_out, n_blocks, M,
                           cfg.vf_coef, cfg.ent_coef, scalars_out);
        TSM_LAUNCH_CHECK();
    }
    return TSM_OK;
}

// Loss statistics of many gradient steps in ONE launch (the per-step finalize of tsm_ppo_update_fused is
// skipped by passing scalars_out = NULL there): step k reads partial + k*stride_elems, n_blocks_dev[k]
// workgroup rows, M_dev[k] samples; writes scalars_out[k][4] = {loss, clip, vf, ent} (ppo.py:213-216).
TSM_EXPORT int tsm_ppo_finalize_many(const double *partial, int64_t stride_elems, const int32_t *n_blocks_dev,
                                     const int64_t *M_dev, int32_t n_steps, const tsm_ppo_cfg *cfg_host,
                                     float *scalars_out, void *stream) {
    TSM_REQUIRE(n_steps >= 0, "tsm_ppo_finalize_many: negative n_steps");
    if (n_steps == 0) return TSM_OK;
    TSM_REQUIRE(partial && n_blocks_dev && M_dev && cfg_host && scalars_out && stride_elems >= 4,
                "tsm_ppo_finalize_many: bad args");
    hipLaunchKernelGGL(update_finalize_many_kernel, dim3((unsigned)n_steps), dim3(256), 0, tsm_stream(stream), partial,
                       stride_elems, n_blocks_dev, M_dev, (float)cfg_host->vf_coef, (float)cfg_host->ent_coef,
                       scalars_out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}
