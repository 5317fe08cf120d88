// rollout.hip -- persistent rollout kernel: the whole Collector loop of one collect(n_step) call in ONE launch.
//
// Fuses, per vector step and without leaving the chip (/root/reference paths):
//   policy forward + Categorical sample/log-prob + critic value   modelfree/reinforce.py:167-192, a2c.py:121-127,
//                                                                 ppo.py:157-161 (logp_old, v_s, v_s_ are what
//                                                                 _preprocess_batch would recompute later with
//                                                                 the SAME frozen parameters)
//   env.step of every env + reset of finished envs                 data/collector.py:875,971 ; env/venvs.py:237-322
//   buffer.add: index algebra + AoS->SoA payload scatter           data/buffer/manager.py:131-193
// Environments are independent and the policy is frozen during collection, so a workgroup owns EPB = 16 / N
// whole environments (<= 16 agent rows = one MFMA row tile) for all T steps: weights are staged in LDS once,
// env state lives in LDS, and only the buffer rows travel to HBM (coalesced: consecutive envs of a workgroup
// are adjacent in the time-major store).  Results are bit-identical to the unfused sequence
// tsm_policy_forward -> tsm_mpe_spread_step -> tsm_vrb_add (tests/test_gpu_pipeline.py).
//
// Extra output v_next = V(obs_next) (needed by GAE, a2c.py:124): for rows whose episode continues it is the
// critic value of the next step's forward; for finished episodes the terminal observation gets its own
// forward pass before the env is re-initialised; the last step is bootstrapped after the loop.
#include "common.h"
#include "mlp_tile.h"
#include "mpe_dev.h"
#include "philox.h"
#include "vrb_dev.h"

int tsm_mpe_check_cfg(const tsm_mpe_cfg *h, MpeCfg *c);  // mpe.hip
extern long long *g_tsm_stamps;                               // abi.hip (diagnostics)

namespace {

struct RolloutArgs {
    // policy
    const float *P, *img;
    Dims d;
    uint64_t pol_seed, offset;
    const uint64_t *offset_dev;
    int mode;  // 1 sample, 2 argmax
    // env
    MpeCfg c;
    uint64_t env_seed;
    uint64_t *episode_ctr;
    float *apos, *avel, *lpos;
    int32_t *steps;
    int auto_reset;
    float *obs_cur_out;  // [n_env][N][D] next policy input after the rollout
    // buffer
    void *vrb_state;
    int64_t S;
    uint8_t *done_store;
    float *obs_store, *obs_next_store, *rew_store, *logp_store, *vs_store, *vnext_store;
    int32_t *act_store;
    uint8_t *term_store, *trunc_store;
    // per-step outputs [n_steps][n_env]...
    int64_t *ptr_out, *ep_len_out, *ep_idx_out;
    double *ep_rew_out;
    int n_steps;
    // compact record of the episodes finished during this rollout (nullable): i64 words
    //   [n_env] count | [n_env][max_ep] (step << 32 | length) | [n_env][max_ep][N] f64 return
    int64_t *ep_rec;
    int max_ep;
    uint64_t offset_inc;
    uint64_t *offset_dev_rw;
    uint32_t *done_ctr;
    long long *stamps;  // diagnostic build only (tsm_debug_set_stamps): phase time stamps of workgroup 0
};


#define XSTAMP(k) do { if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0 && t < 4) a.stamps[900 + t * 8 + (k)] = (long long)wall_clock64(); } while (0)
#define STAMP(k) do { if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0 && t < 4) a.stamps[t * 8 + (k)] = (long long)wall_clock64(); } while (0)

// With NT2 = 512 the workgroup has EIGHT waves: in the forward pass waves 0-3 carry the actor and waves 4-7 the critic (one MFMA chain
// per wave and layer); the element-wise loops (observation build, payload scatter) spread over all 512 threads;
// everything else runs on threads 0..255.  Measured: forward 2.2 -> 2.1 us, scatter 0.7 -> 0.5 us per vector step.  The
// forward gains little because the two chains of a SIMD share its matrix pipe wherever they sit (dropping the critic
// chains altogether, as an experiment, gave 1.3 us): the tile forward is MFMA-issue bound at 16 rows per CU.
// NT2 = 256 is the four-wave form (both chains on every wave, tile_forward): its workgroups need half the registers, so
// two of them share a CU when their LDS fits twice -- the better choice once there are more workgroups than CUs.
template <int H, int NT2>
__global__ __launch_bounds__(NT2) void rollout_kernel(RolloutArgs a) {
    extern __shared__ float lds[];
    const Dims d = a.d;
    const MpeCfg c = a.c;
    const Lay<H> ly(d, false);
    const int N = c.N, D = d.D, st = 2 * N;
    const int EPB = R / N;                    // envs per workgroup
    const int e0 = blockIdx.x * EPB;
    const int n_here = min(EPB, c.n_env - e0);
    const int rows_here = n_here * N;         // live tile rows (<= 16): row r = (env el, agent i), r = el*N + i
    const int64_t B = c.n_env;
    // extra LDS behind the forward layout
    float *XN0 = lds + ly.total;              // [R][ld1] second observation tile (obs_next rows)
    float *s_ap = XN0 + R * d.ld1;             // [EPB][N][2]
    float *s_av = s_ap + R * 2;
    float *s_lp = s_av + R * 2;
    float *s_rew = s_lp + R * 2;              // [R]
    float *s_logp = s_rew + R;                // [R]
    float *s_val = s_logp + R;                // [R]
    float *s_m = s_val + R;                   // [R] landmark minima
    int *s_act = reinterpret_cast<int *>(s_m + R);          // [R]
    int *s_steps = s_act + R;                               // [EPB]
    int *s_done = s_steps + R;                              // [EPB] done flag of the step just taken
    int64_t *s_row = reinterpret_cast<int64_t *>(                    // [EPB] slot*B + env of the step just added
        (reinterpret_cast<uintptr_t>(s_done + R) + 7) & ~(uintptr_t)7);
    uint64_t *s_ep = reinterpret_cast<uint64_t *>(s_row + R);        // [EPB] episode counter of finished envs
    // s_done / s_row keep the values of step t until phase D of step t + 1 overwrites them, so phase C of step t + 1
    // reads them as "the previous step" (pending V(obs_next) stores) without a copy

    // diagnostics (tsm_debug_set_stamps, >= 1024 slots): [64 + 2b], [65 + 2b] = start / end of workgroup b;
    // [640 + t] = start of step t in workgroup 0
    if (a.stamps && threadIdx.x == 0 && blockIdx.x < 256) a.stamps[64 + 2 * blockIdx.x] = (long long)wall_clock64();
    const bool main_t = threadIdx.x < NT;  // waves 0-3
    if (main_t) {
        if (a.img) stage_image<H>(lds, ly, a.img);
        else stage_weights<H>(lds, ly, d, a.P);
    }
    for (int i = threadIdx.x; i < R * d.ld1; i += NT2) { lds[ly.X + i] = 0.f; XN0[i] = 0.f; }
    const VrbState vs = vrb_view(a.vrb_state, B, N);
    // agent lane r < rows_here (wave 0) <-> (env el, agent i); env lane 64 + q (wave 1) owns env q's bookkeeping, so
    // the buffer index algebra runs beside the agent lanes' physics instead of after it
    const int r = threadIdx.x, el = r / N, ai = r - el * N;
    const bool lane_live = r < rows_here;
    const int e = e0 + el;
    const int bel = (int)threadIdx.x - 64;  // env lane: local env index
    const bool env_lane = bel >= 0 && bel < n_here;
    const int be = e0 + bel;
    // sub-buffer bookkeeping of "my" env lives in registers for the whole rollout: the per-step index algebra
    // then has no dependent global loads, only fire-and-forget stores
    int64_t v_ins = 0, v_size = 0, v_eplen = 0, v_epstart = 0, v_last = 0;
    int n_fin = 0;  // episodes this env finished during the rollout
    double v_epret[kMpeMaxN];
#pragma unroll
    for (int k = 0; k < kMpeMaxN; ++k) v_epret[k] = 0.0;
    if (env_lane) {
        v_ins = vs.ins[be]; v_size = vs.size[be]; v_eplen = vs.ep_len[be]; v_epstart = vs.ep_start[be];
        v_last = vs.last_index[be];
#pragma unroll
        for (int k = 0; k < kMpeMaxN; ++k) if (k < N) v_epret[k] = vs.ep_return[(int64_t)be * N + k];
        s_steps[bel] = a.steps[be];
        s_done[bel] = 1;  // "no pending v_next" before the first step
        s_row[bel] = 0;
    }
    for (int i = threadIdx.x; i < n_here * st; i += NT2) {
        s_ap[i] = a.apos[(int64_t)e0 * st + i];
        s_av[i] = a.avel[(int64_t)e0 * st + i];
        s_lp[i] = a.lpos[(int64_t)e0 * st + i];
    }
    const uint64_t off0 = a.offset + (a.offset_dev ? *a.offset_dev : 0ull);
    __syncthreads();
    // two observation tiles, swapped every step: obs_next of step t IS obs of step t + 1 (rows of re-initialised
    // envs are rebuilt in F), so the observation function runs once per step, not twice
    Lay<H> lyf = ly;
    int xcur = ly.X, xnxt = ly.total;
    // A. observation rows from the LDS-resident state: one (row, element) per thread
    for (int i = threadIdx.x; i < rows_here * D; i += NT2) {
        const int rr = i / D, k = i - rr * D, ee = rr / N;
        lds[xcur + rr * d.ld1 + k] = mpe_obs_elem(c, s_ap + ee * st, s_av + ee * st, s_lp + ee * st, rr - ee * N, k);
    }
    __syncthreads();

    for (int t = 0; t <= a.n_steps; ++t) {
        const bool last = t == a.n_steps;  // extra pass: bootstrap value of the final observation only
        float *XN = lds + xnxt;
        STAMP(0);
        if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0 && t < 64) a.stamps[640 + t] = (long long)wall_clock64();
        STAMP(1);
        // B. actor + critic forward of the 16-row tile
        lyf.X = xcur;
        if constexpr (NT2 == 2 * NT) tile_forward_split<H>(lds, lyf, d);
        else tile_forward<H>(lds, lyf, d);
        STAMP(2);
        // C. head: 16 lanes per row compute exp(logit - max) in parallel; lane 0 of the row then folds them in
        //    action order (same arithmetic order as tsm_policy_forward => identical samples and log-probs)
        if (main_t) {
            const int hr = threadIdx.x >> 4, j = threadIdx.x & 15;
            const float *lg = lds + ly.OUT + hr * ly.ldo;
            const bool on = j < d.A;
            const float x = on ? lg[j] : -INFINITY;
            const float m = row16_max(x);
            const float ex = on ? expf(x - m) : 0.f;
            const int gbase = threadIdx.x & 48;  // first lane of this row's 16-lane group inside the wave
            float ssum = 0.f;
            row_prefix_sum<0>(ex, d.A, ssum);
            if (j == 0 && hr < rows_here) {
                const float val = lg[16];
                const int hel = hr / N;
                if (!s_done[hel] && a.vnext_store)  // V(obs_next) of the previous step == V(obs) of this one
                    a.vnext_store[s_row[hel] * N + (hr - hel * N)] = val;
                if (!last) s_val[hr] = val;
            }
            if (!last) {
                int act = 0;
                if (a.mode == 1) {
                    const uint64_t gi = (uint64_t)e0 * N + hr;  // global row index env*N + agent
                    const float u = tsm_philox_uniform(a.pol_seed, off0 + (uint64_t)t * B * N + gi) * ssum;
                    float cs = 0.f;
                    act = d.A - 1;
                    bool found = false;
                    row_cdf_pick<0>(ex, d.A, u, cs, act, found);
                } else {  // dist.mode: first index attaining the maximum
                    const unsigned long long eq = __ballot(on && x == m);
                    act = __ffsll((long long)((eq >> gbase) & 0xFFFFull)) - 1;
                }
                const float la = __shfl(x, gbase + act, 64);
                if (j == 0 && hr < rows_here) {
                    s_act[hr] = act;
                    s_logp[hr] = la - (m + logf(ssum));
                }
            }
        }
        if (last) {
            // the observation of the next collect() call
            if (a.obs_cur_out)
                for (int i = threadIdx.x; i < rows_here * D; i += NT2) {
                    const int rr = i / D, k = i - rr * D;
                    a.obs_cur_out[((int64_t)e0 * N + rr) * D + k] = lds[xcur + rr * d.ld1 + k];
                }
            break;
        }
        __syncthreads();
        STAMP(3);
        // D. env step, one lane per agent (mpe_dev.h): move -> barrier -> publish -> barrier -> reward terms.
        //    Beside the move, the env lanes do the buffer index algebra on their register-resident sub-buffer state
        //    (buffer_base.py:373-410 + manager.py:170-177; same arithmetic as vrb_add_row in vrb_dev.h).
        float npx = 0.f, npy = 0.f, nvx = 0.f, nvy = 0.f;
        if (lane_live) mpe_agent_move(c, s_ap + el * st, s_av + el * st, ai, s_act[r], npx, npy, nvx, nvy);
        bool tr = false, rec = false;
        int64_t o = 0;
        if (env_lane) {
            const int stp = s_steps[bel] + 1;
            tr = stp >= c.max_cycles;
            s_steps[bel] = stp;
            o = (int64_t)t * B + be;
            const int64_t cur = v_ins;
            int64_t sz = v_size + 1; if (sz > a.S) sz = a.S;
            int64_t nxt = cur + 1; if (nxt >= a.S) nxt -= a.S;
            const int64_t elen = v_eplen + 1;
            if (v_epstart > sz) atomicExch((unsigned long long *)vs.error_flag, 1ull);
            rec = tr && a.ep_rec && n_fin < a.max_ep;
            if (rec) a.ep_rec[B + (int64_t)be * a.max_ep + n_fin] = ((int64_t)t << 32) | elen;
            a.ep_len_out[o] = tr ? elen : 0;
            a.ptr_out[o] = cur + (int64_t)be * a.S;
            a.ep_idx_out[o] = v_epstart + (int64_t)be * a.S;
            v_ins = nxt; v_size = sz; v_eplen = tr ? 0 : elen; v_epstart = tr ? nxt : v_epstart;
            v_last = cur + (int64_t)be * a.S;
            a.done_store[cur * B + be] = tr ? 1 : 0;
            s_row[bel] = cur * B + be;
            s_done[bel] = tr ? 1 : 0;
        }
        __syncthreads();
        XSTAMP(0);
        if (lane_live) {
            s_ap[el * st + 2 * ai] = npx; s_ap[el * st + 2 * ai + 1] = npy;
            s_av[el * st + 2 * ai] = nvx; s_av[el * st + 2 * ai + 1] = nvy;
        }
        __syncthreads();
        XSTAMP(1);
        float local = 0.f;
        if (lane_live) {
            const MpePos pos = mpe_load_pos(c, s_ap + el * st);
            s_m[r] = mpe_landmark_min_dist(c, pos, s_lp + el * st, ai);
            local = mpe_local_penalty(c, pos, s_ap + el * st, ai);
        }
        XSTAMP(2);
        // obs_next rows (terminal observation for finished episodes) on waves 1-7 while wave 0 works on the reward terms
        if (threadIdx.x >= 64)
            for (int i = threadIdx.x - 64; i < rows_here * D; i += NT2 - 64) {
                const int rr = i / D, k = i - rr * D, ee = rr / N;
                XN[rr * d.ld1 + k] =
                    mpe_obs_elem(c, s_ap + ee * st, s_av + ee * st, s_lp + ee * st, rr - ee * N, k);
            }
        __syncthreads();
        XSTAMP(3);
        if (lane_live) s_rew[r] = mpe_reward(c, s_m + el * N, local);
        __syncthreads();
        XSTAMP(4);
        if (env_lane) {  // episode returns (needs the rewards); runs beside the payload scatter below
            double *rec_rew = rec ? reinterpret_cast<double *>(a.ep_rec + B + (int64_t)B * a.max_ep) +
                                        ((int64_t)be * a.max_ep + n_fin) * N : nullptr;
#pragma unroll
            for (int k = 0; k < kMpeMaxN; ++k) {
                if (k < N) {
                    const double acc = v_epret[k] + (double)s_rew[bel * N + k];
                    a.ep_rew_out[o * N + k] = tr ? acc : 0.0;
                    if (rec) rec_rew[k] = acc;
                    v_epret[k] = tr ? 0.0 : acc;
                }
            }
            n_fin += tr ? 1 : 0;
        }
        STAMP(4);
        // E. payload scatter into the time-major SoA store (rows of consecutive envs are adjacent)
        for (int i = threadIdx.x; i < rows_here * D; i += NT2) {
            const int rr = i / D, k = i - rr * D, ee = rr / N;
            const int64_t dst = (s_row[ee] * N + (rr - ee * N)) * D + k;
            a.obs_store[dst] = lds[xcur + rr * d.ld1 + k];
            if (a.obs_next_store) a.obs_next_store[dst] = XN[rr * d.ld1 + k];
        }
        if (lane_live) {
            const int64_t dst = s_row[el] * N + ai;
            a.act_store[dst] = s_act[r];
            a.rew_store[dst] = s_rew[r];
            a.term_store[dst] = 0;
            a.trunc_store[dst] = (uint8_t)s_done[el];
            if (a.logp_store) a.logp_store[dst] = s_logp[r];
            if (a.vs_store) a.vs_store[dst] = s_val[r];
        }
        STAMP(5);
        // F. finished episodes: critic value of the terminal observation, then re-initialise the env
        int any_done = lane_live ? s_done[el] : 0;
        any_done = __syncthreads_or(any_done);
        if (any_done) {
            if (a.vnext_store) {
                lyf.X = xnxt;
                if constexpr (NT2 == 2 * NT) tile_forward_split<H>(lds, lyf, d);
                else tile_forward<H>(lds, lyf, d);
                if (lane_live && s_done[el]) a.vnext_store[s_row[el] * N + ai] = lds[ly.OUT + r * ly.ldo + 16];
            }
            if (a.auto_reset) {
                if (env_lane && s_done[bel]) {
                    const uint64_t ep = a.episode_ctr[be];
                    s_ep[bel] = ep;
                    a.episode_ctr[be] = ep + 1;
                    s_steps[bel] = 0;
                }
                __syncthreads();
                if (lane_live && s_done[el])
                    mpe_reset_agent(c, e, a.env_seed, s_ep[el], ai, s_ap + el * st, s_av + el * st, s_lp + el * st);
                __syncthreads();
                // first observation of the new episodes
                for (int i = threadIdx.x; i < rows_here * D; i += NT2) {
                    const int rr = i / D, k = i - rr * D, ee = rr / N;
                    if (s_done[ee])
                        XN[rr * d.ld1 + k] =
                            mpe_obs_elem(c, s_ap + ee * st, s_av + ee * st, s_lp + ee * st, rr - ee * N, k);
                }
                __syncthreads();  // the next forward reads these rows at once
            }
        }
        // no barrier at the end of a step: nothing written above is read before the first barrier inside the next
        // step's forward pass, and that barrier also separates this step's LDS reads from the next step's writes
        { const int tmp = xcur; xcur = xnxt; xnxt = tmp; }
        STAMP(6);
    }
    // env state + sub-buffer bookkeeping back to HBM
    __syncthreads();
    if (a.stamps && threadIdx.x == 0 && blockIdx.x < 256) a.stamps[65 + 2 * blockIdx.x] = (long long)wall_clock64();
    for (int i = threadIdx.x; i < n_here * st; i += NT2) {
        a.apos[(int64_t)e0 * st + i] = s_ap[i];
        a.avel[(int64_t)e0 * st + i] = s_av[i];
        a.lpos[(int64_t)e0 * st + i] = s_lp[i];
    }
    if (env_lane) {
        a.steps[be] = s_steps[bel];
        vs.ins[be] = v_ins; vs.size[be] = v_size; vs.ep_len[be] = v_eplen; vs.ep_start[be] = v_epstart;
        vs.last_index[be] = v_last; vs.lengths[be] = v_size;
        if (a.ep_rec) a.ep_rec[be] = n_fin;  // may exceed max_ep: the host treats that as an overflow
#pragma unroll
        for (int k = 0; k < kMpeMaxN; ++k) if (k < N) vs.ep_return[(int64_t)be * N + k] = v_epret[k];
    }
    // the last workgroup to get here advances the sampling counter: every workgroup read it (off0) before finishing
    if (a.done_ctr && threadIdx.x == 0) {
        if (atomicAdd(a.done_ctr, 1u) == gridDim.x - 1) {
            *a.offset_dev_rw += a.offset_inc;
            *a.done_ctr = 0u;
        }
    }
}

__global__ void u64_add_kernel(uint64_t *p, uint64_t inc) { *p += inc; }

}  // namespace

TSM_EXPORT int tsm_u64_add(uint64_t *counter, uint64_t inc, void *stream) {
    TSM_REQUIRE(counter, "tsm_u64_add: null pointer");
    hipLaunchKernelGGL(u64_add_kernel, dim3(1), dim3(1), 0, tsm_stream(stream), counter, inc);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int tsm_rollout_spread(const tsm_rollout_desc *desc_host, void *stream) {
    TSM_REQUIRE(desc_host, "tsm_rollout_spread: null descriptor");
    const tsm_rollout_desc &h = *desc_host;
    RolloutArgs a;
    if (int rc = make_dims(h.obs_dim, h.hidden, h.n_act, &a.d)) return rc;
    if (int rc = tsm_mpe_check_cfg(&h.env, &a.c)) return rc;
    TSM_REQUIRE(a.c.obs_dim == h.obs_dim, "tsm_rollout_spread: obs_dim %d != 6 * n_agent", h.obs_dim);
    TSM_REQUIRE(h.n_act == 5, "tsm_rollout_spread: simple_spread has 5 discrete actions");
    TSM_REQUIRE(a.c.N <= R, "tsm_rollout_spread: n_agent exceeds the row tile");
    TSM_REQUIRE(h.n_steps >= 1 && h.sub_size >= 1, "tsm_rollout_spread: bad n_steps / sub_size");
    TSM_REQUIRE(h.mode == 1 || h.mode == 2, "tsm_rollout_spread: mode must be 1 (sample) or 2 (argmax)");
    TSM_REQUIRE((h.params || h.param_image) && h.episode_ctr && h.agent_pos && h.agent_vel && h.landmark_pos &&
                    h.steps && h.vrb_state && h.done_store && h.obs_store && h.act_store && h.rew_store &&
                    h.term_store && h.trunc_store && h.ptr_out && h.ep_rew_out && h.ep_len_out && h.ep_idx_out,
                "tsm_rollout_spread: null pointer");
    a.P = h.params; a.img = h.param_image;
    a.pol_seed = h.policy_seed; a.offset = h.offset; a.offset_dev = h.offset_dev; a.mode = h.mode;
    a.env_seed = h.env_seed; a.episode_ctr = h.episode_ctr;
    a.apos = h.agent_pos; a.avel = h.agent_vel; a.lpos = h.landmark_pos; a.steps = h.steps;
    a.auto_reset = h.auto_reset; a.obs_cur_out = h.obs_cur_out;
    a.vrb_state = h.vrb_state; a.S = h.sub_size; a.done_store = h.done_store;
    a.obs_store = h.obs_store; a.obs_next_store = h.obs_next_store; a.rew_store = h.rew_store;
    a.logp_store = h.logp_store; a.vs_store = h.vs_store; a.vnext_store = h.vnext_store;
    a.act_store = h.act_store; a.term_store = h.term_store; a.trunc_store = h.trunc_store;
    a.ptr_out = h.ptr_out; a.ep_len_out = h.ep_len_out; a.ep_idx_out = h.ep_idx_out; a.ep_rew_out = h.ep_rew_out;
    a.n_steps = h.n_steps;
    TSM_REQUIRE(!h.ep_rec || h.max_ep >= 1, "tsm_rollout_spread: ep_rec needs max_ep >= 1");
    a.ep_rec = h.ep_rec; a.max_ep = h.max_ep;
    TSM_REQUIRE(!h.done_ctr || h.offset_dev, "tsm_rollout_spread: done_ctr needs offset_dev");
    a.offset_inc = h.offset_inc; a.done_ctr = h.done_ctr;
    a.offset_dev_rw = const_cast<uint64_t *>(reinterpret_cast<const uint64_t *>(h.offset_dev));
    a.stamps = g_tsm_stamps;
    const Lay<64> ly(a.d, false);
    const size_t extra = (size_t)R * a.d.ld1 + 3 * R * 2 + 4 * R + 4 * R + 3 * 2 * R + 8;
    const size_t shmem = ((size_t)ly.total + extra) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        TSM_HIP(tsm_allow_max_lds(reinterpret_cast<const void *>(rollout_kernel<64, NT>)));
        TSM_HIP(tsm_allow_max_lds(reinterpret_cast<const void *>(rollout_kernel<64, 2 * NT>)));
        attr_set = true;
    }
    const int EPB = R / a.c.N;
    const unsigned n_wg = (unsigned)ceil_div(a.c.n_env, EPB);
    // Eight waves when a CU holds one workgroup anyway (no more workgroups than CUs, or an LDS footprint above half a
    // CU); four waves otherwise, so that two workgroups fit the register file of a CU.  Measured (us per 25-step
    // collect, 4 / 8 waves): 1024 envs x 3 agents 242 / 234; 4096 x 3: 505 / 775; 4096 x 8 (91 KB of LDS): 7100 / 2250.
    const bool eight = n_wg <= 256 || shmem > 80 * 1024;
    if (eight) hipLaunchKernelGGL((rollout_kernel<64, 2 * NT>), dim3(n_wg), dim3(2 * NT), shmem, tsm_stream(stream), a);
    else hipLaunchKernelGGL((rollout_kernel<64, NT>), dim3(n_wg), dim3(NT), shmem, tsm_stream(stream), a);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

