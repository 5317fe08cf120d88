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
#include "wave_mlp.h"

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
// DS / NS: 0 = obs width / agent count from the launch arguments, else the values this instantiation is compiled for (BASELINE
// configs[1]: 18 / 3): the per-step element loops divide by D, N and ld1 -- ~35-instruction integer divisions when those are run-time
// values, a multiply and a shift when they are constants.
template <int H, int NT2, int DS = 0, int NS = 0>
__global__ __launch_bounds__(NT2) void rollout_kernel(RolloutArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const Dims d = DS ? dims_const(DS, 5) : a.d;
    MpeCfg c_ = a.c;
    if (NS) { c_.N = NS; c_.obs_dim = DS; }
    const MpeCfg c = c_;
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
    int *s_done = s_steps + R;                              // [2][EPB] done flag of the step just taken (generation t & 1)
    // (8-byte alignment by INDEX arithmetic off the 16-byte aligned LDS base: rounding the address through uintptr_t made s_row /
    //  s_ep generic pointers, every access to them a FLAT instruction, and every flat load waits for vmcnt(0) -- i.e. for all of the
    //  step's outstanding global stores: round 5, 24 flat operations in this file)
    int64_t *s_row = reinterpret_cast<int64_t *>(                    // [2][EPB] slot*B + env of the step just added (generation t & 1)
        lds + (((int)(reinterpret_cast<float *>(s_done + 2 * R) - lds) + 1) & ~1));
    uint64_t *s_ep = reinterpret_cast<uint64_t *>(s_row + 2 * R);    // [EPB] episode counter of finished envs
    float *s_cx = lds + (((int)(reinterpret_cast<float *>(s_ep + R) - lds) + 3) & ~3);   // [R][8] pair forces of (row, other agent), phase D; 16-byte rows
    float *s_cy = s_cx + R * 8;
    int *s_cv = reinterpret_cast<int *>(s_cy + R * 8);               // [R][8] "the pair is in range"
    float *s_u = reinterpret_cast<float *>(s_cv + R * 8);            // [2][R] the sampling uniforms of step t (generation t & 1)
    int *s_any = reinterpret_cast<int *>(s_u + 2 * R);               // [2] "an episode of this tile ends at step t"
    // s_done / s_row in two generations: step t's index algebra (wave 4, beside the head of phase C) writes generation t & 1 while
    // the head reads generation (t - 1) & 1 as "the previous step" (pending V(obs_next) stores)

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
    // agent lane r < rows_here (wave 0) <-> (env el, agent i); env lane 256 + q (wave 4) owns env q's bookkeeping: the buffer
    // index algebra of a step runs beside its head (phase C: waves 0-3), off the step's dependent chain
    const int r = threadIdx.x, el = r / N, ai = r - el * N;
    const bool lane_live = r < rows_here;
    const int e = e0 + el;
    // (the four-wave form has no waves 4 and 5: its env lanes sit on wave 1, its uniform / payload lanes on wave 2)
    constexpr int kEnvBase = NT2 >= 512 ? 256 : 64, kUBase = NT2 >= 512 ? 320 : 128;
    const int bel = (int)threadIdx.x - kEnvBase;  // env lane: local env index
    const bool env_lane = bel >= 0 && bel < n_here;
    const int be = e0 + bel;
    // DEFER (eight-wave form): the reward terms of a step that ends no episode -- landmark minima, collision penalty, reward, the
    // episode returns and the rew_store row -- are not needed by the next step's forward pass: wave 6, idle during the head
    // (phase C), computes them ONE STEP LATE from the positions the env step left in LDS (they move again only in the next
    // step's phase D, behind two barriers).  A step that ends an episode keeps them in place, before the reset (phase F).
    // Same functions on the same inputs: the same bits.  The episode-return accumulators live on wave 6's lanes then.
    constexpr bool DEFER = NT2 >= 512;
    constexpr int kRewBase = 384;
    const int wl = (int)threadIdx.x - kRewBase;           // reward lane: agent row wl; for wl < n_here also env wl's returns
    const bool rew_lane = DEFER && wl >= 0 && wl < rows_here;
    // (the episode returns of a step follow on wave 7, one phase later still: beside the next step's pair forces)
    constexpr int kRetBase = 448;
    const int rl = (int)threadIdx.x - kRetBase;
    const bool ret_lane = DEFER ? (rl >= 0 && rl < n_here) : env_lane;
    const int rq = DEFER ? rl : bel;                      // the return lane's local env
    bool pending = false;                                 // (uniform) the previous step's reward terms are still to do
    // sub-buffer bookkeeping of "my" env lives in registers for the whole rollout: the per-step index algebra
    // then has no dependent global loads, only fire-and-forget stores
    int64_t v_ins = 0, v_size = 0, v_eplen = 0, v_epstart = 0, v_last = 0;
    int n_fin = 0;    // episodes this env finished during the rollout (env lanes)
    int n_fin_r = 0;  // the same count on the return lanes
    double v_epret[kMpeMaxN];
#pragma unroll
    for (int k = 0; k < kMpeMaxN; ++k) v_epret[k] = 0.0;
    if (ret_lane) {
#pragma unroll
        for (int k = 0; k < kMpeMaxN; ++k) if (k < N) v_epret[k] = vs.ep_return[(int64_t)(e0 + rq) * N + k];
    }
    // episode returns of step tt on the return lanes from the rewards in s_rew: trq = the env's episode ends at that step
    auto step_returns = [&](int tt, bool trq) {
        const int64_t bq = e0 + rq;
        const bool recq = trq && a.ep_rec && n_fin_r < a.max_ep;
        const int64_t oq = (int64_t)tt * B + bq;
        double *rec_rew = recq ? reinterpret_cast<double *>(a.ep_rec + B + (int64_t)B * a.max_ep) +
                                     ((int64_t)bq * a.max_ep + n_fin_r) * N : nullptr;
#pragma unroll
        for (int k = 0; k < kMpeMaxN; ++k) {
            if (k < N) {
                const double acc = v_epret[k] + (double)s_rew[rq * N + k];
                a.ep_rew_out[oq * N + k] = trq ? acc : 0.0;
                if (recq) rec_rew[k] = acc;
                v_epret[k] = trq ? 0.0 : acc;
            }
        }
        n_fin_r += trq ? 1 : 0;
    };
    if (env_lane) {
        v_ins = vs.ins[be]; v_size = vs.size[be]; v_eplen = vs.ep_len[be]; v_epstart = vs.ep_start[be];
        v_last = vs.last_index[be];
        s_steps[bel] = a.steps[be];
        s_done[bel] = 1; s_done[R + bel] = 1;  // "no pending v_next" before the first step
        s_row[bel] = 0; s_row[R + bel] = 0;
    }
    for (int i = threadIdx.x; i < n_here * st; i += NT2) {
        s_ap[i] = a.apos[(int64_t)e0 * st + i];
        s_av[i] = a.avel[(int64_t)e0 * st + i];
        s_lp[i] = a.lpos[(int64_t)e0 * st + i];
    }
    const uint64_t off0 = a.offset + (a.offset_dev ? *a.offset_dev : 0ull);
    // The sampling uniform of (step, row) depends on nothing the step computes: ten Philox rounds (~0.25 us of dependent
    // instructions) that the head used to run between the logits and the sample.  Wave 5 draws step t + 1's beside step t's head.
    const int ur = (int)threadIdx.x - kUBase;   // uniform lane: row
    const bool u_lane = a.mode == 1 && ur >= 0 && ur < rows_here;
    if (u_lane) s_u[ur] = tsm_philox_uniform(a.pol_seed, off0 + (uint64_t)e0 * N + (uint64_t)ur);
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
        const int g = (t & 1) * R;   // this step's generation of s_row / s_done; g ^ R: the previous step's
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
        bool tr = false, rec = false;
        int64_t o = 0;
        if (env_lane && !last) {   // the buffer index algebra of this step (beside the head; D .. F read generation t & 1)
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
            // (the record carries CollectStats.lens = len(episode_batch): the episode's rows IN THE BUFFER, collector.py:203,990-993 --
            //  after a reset_buffer(keep_statistics=True) an episode counts its rows since the reset; ep_len_out stays add()'s ep_len)
            if (rec) a.ep_rec[B + (int64_t)be * a.max_ep + n_fin] = ((int64_t)t << 32) | ((cur >= v_epstart ? cur - v_epstart : cur - v_epstart + a.S) + 1);
            n_fin += tr ? 1 : 0;
            a.ep_len_out[o] = tr ? elen : 0;
            a.ptr_out[o] = cur + (int64_t)be * a.S;
            a.ep_idx_out[o] = v_epstart + (int64_t)be * a.S;
            v_ins = nxt; v_size = sz; v_eplen = tr ? 0 : elen; v_epstart = tr ? nxt : v_epstart;
            v_last = cur + (int64_t)be * a.S;
            a.done_store[cur * B + be] = tr ? 1 : 0;
            s_row[g + bel] = cur * B + be;
            s_done[g + bel] = tr ? 1 : 0;
            const unsigned long long ends = __ballot(tr);   // (the env lanes share wave 4)
            if (bel == 0) s_any[t & 1] = ends != 0ull;
        }
        if (u_lane && !last) s_u[((t + 1) & 1) * R + ur] = tsm_philox_uniform(a.pol_seed, off0 + (uint64_t)(t + 1) * B * N + (uint64_t)e0 * N + (uint64_t)ur);
        if (DEFER && pending) {
            // wave 6, beside the head: the reward terms of step t - 1, which ended no episode (generation g ^ R)
            if (wl >= 0 && wl < 64) {
                // (all 64 lanes run it: the minima travel between lanes by ds_bpermute -- one round trip, where an LDS write -> read pair
                //  costs two; idle lanes work on row 0 and store nothing)
                const int wr = wl < rows_here ? wl : 0;
                const int wel = wr / N, wai = wr - wel * N;
                const int64_t rowq = s_row[(g ^ R) + wel];      // (read ahead of the chain below)
                const MpePos pos = mpe_load_pos(c, s_ap + wel * st);
                const float md = mpe_landmark_min_dist(c, pos, s_lp + wel * st, wai);
                const float local = mpe_local_penalty(c, pos, s_ap + wel * st, wai);
                float mv[kMpeMaxN];
#pragma unroll
                for (int l = 0; l < kMpeMaxN; ++l) mv[l] = l < N ? __shfl(md, wel * N + l, 64) : 0.f;
                const float rw = mpe_reward_regs(c, mv, local);
                if (wl < rows_here) { s_rew[wl] = rw; a.rew_store[rowq * N + wai] = rw; }
            }
            pending = false;
        }
        // pair force task p = (agent row p >> 3, other agent p & 7) -> s_cx / s_cy / s_cv[p]
        auto pair_task = [&](int p) {
            const int rp = p >> 3, jp = p & 7;
            float sx = 0.f, sy = 0.f;
            int ok = 0;
            if (rp < rows_here && jp < N) {
                const int ep = rp / N, ip = rp - ep * N;
                const float *ap = s_ap + ep * st;
                if (jp != ip) ok = mpe_pair_force(c, ap[2 * ip], ap[2 * ip + 1], ap[2 * jp], ap[2 * jp + 1], ip, jp, sx, sy) ? 1 : 0;
            }
            s_cx[p] = sx; s_cy[p] = sy; s_cv[p] = ok;
        };
        if (DEFER && !last) {
            // The pair forces of THIS step depend on the positions alone, not on the actions the head is sampling: wave 7 evaluates them
            // beside the head (up to four agents: 16 rows x 4 partners on its 64 lanes; more: waves 7 and 5 share the 128 tasks), and
            // phase D starts at the fold.  (They ran as a phase of their own behind the head: 0.6-0.76 us of a 3.9 us step.)
            if (N <= 4) { if (rl >= 0 && rl < 64) pair_task(((rl >> 2) << 3) | (rl & 3)); }
            else if (rl >= 0 && rl < 64) pair_task(rl);
            else if (ur >= 0 && ur < 64) pair_task(64 + ur);
        }
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
                if (!s_done[(g ^ R) + hel] && a.vnext_store)  // V(obs_next) of the previous step == V(obs) of this one
                    a.vnext_store[s_row[(g ^ R) + hel] * N + (hr - hel * N)] = val;
                if (!last) s_val[hr] = val;
            }
            if (!last) {
                int act = 0;
                if (a.mode == 1) {
                    // (the uniform of global row e0 * N + hr at step t: tsm_philox_uniform(pol_seed, off0 + t * B * N + row), drawn ahead)
                    const float u = s_u[(t & 1) * R + (hr < rows_here ? hr : 0)] * ssum;
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
        __syncthreads();
        STAMP(3);
        // the episode returns of step t - 1 from s_rew (written by that step's reward phase, or by wave 6 just above): wave 7, beside the pair forces
        if (DEFER && t > 0 && ret_lane) {
            step_returns(t - 1, s_done[(g ^ R) + rq] != 0);
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
        // D. env step (mpe_dev.h): pair forces -> barrier -> fold, integrate, publish -> barrier -> reward terms.
        //    (The env lanes' buffer index algebra on their register-resident sub-buffer state -- buffer_base.py:373-410 +
        //    manager.py:170-177; same arithmetic as vrb_add_row in vrb_dev.h -- runs beside the head, above.)
        // (round 5: the pair forces as one (agent row, other agent) task per thread of waves 0-1, folded by the agent lanes in partner
        //  order -- mpe_agent_move's sums, rollout_wave64_kernel's scheme.  One lane per agent ran its N - 1 pair evaluations -- sqrt,
        //  exp, log1p, a division each when the pair is in range -- one after the other: 1.3-1.7 us of a 5.6 us step,
        //  profiles/r05_stamp_rollout.txt.)
        if constexpr (!DEFER) {
            if (threadIdx.x < R * 8) pair_task(threadIdx.x);
            XSTAMP(5);   // (wave 0 is through its own part: the pair forces)
            __syncthreads();
        }
        XSTAMP(0);
        if (lane_live) {   // fold, integrate, publish: a lane reads and writes its own agent's position / velocity only
            const float px = s_ap[el * st + 2 * ai], py = s_ap[el * st + 2 * ai + 1];
            const float vx = s_av[el * st + 2 * ai], vy = s_av[el * st + 2 * ai + 1];
            float fx = mpe_action_force(c, s_act[r], 0);
            float fy = mpe_action_force(c, s_act[r], 1);
            typedef int i4 __attribute__((ext_vector_type(4)));
            const i4 v0 = *reinterpret_cast<const i4 *>(s_cv + 8 * r), v1 = *reinterpret_cast<const i4 *>(s_cv + 8 * r + 4);
            const f4 x0 = *reinterpret_cast<const f4 *>(s_cx + 8 * r), x1 = *reinterpret_cast<const f4 *>(s_cx + 8 * r + 4);
            const f4 y0 = *reinterpret_cast<const f4 *>(s_cy + 8 * r), y1 = *reinterpret_cast<const f4 *>(s_cy + 8 * r + 4);
#pragma unroll
            for (int j = 0; j < kMpeMaxN; ++j) {
                const int ok = j < 4 ? v0[j & 3] : v1[j & 3];
                if (j < N && ok) { fx += j < 4 ? x0[j & 3] : x1[j & 3]; fy += j < 4 ? y0[j & 3] : y1[j & 3]; }
            }
            float npx, npy, nvx, nvy;
            mpe_integrate(c, px, py, vx, vy, fx, fy, npx, npy, nvx, nvy);
            s_ap[el * st + 2 * ai] = npx; s_ap[el * st + 2 * ai + 1] = npy;
            s_av[el * st + 2 * ai] = nvx; s_av[el * st + 2 * ai + 1] = nvy;
        }
        __syncthreads();
        XSTAMP(1);
        // (the env lanes left the flag beside the index algebra: a workgroup-wide OR here cost two barriers per step, 0.36 us)
        const int any_done = s_any[t & 1];
        const bool now = !DEFER || any_done;   // (uniform) this step's reward terms in place: an episode ends, the envs are about to be reset
        if (now) {
            float local = 0.f;
            if (lane_live) {
                const MpePos pos = mpe_load_pos(c, s_ap + el * st);
                s_m[r] = mpe_landmark_min_dist(c, pos, s_lp + el * st, ai);
                local = mpe_local_penalty(c, pos, s_ap + el * st, ai);
            }
            XSTAMP(2);
            // (four-wave form) obs_next rows (terminal observation for finished episodes) on waves 1-3 while wave 0 works on the reward terms
            if (!DEFER && threadIdx.x >= 64)
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
            if (!DEFER && ret_lane) {  // episode returns (needs the rewards); runs beside the payload scatter below
                step_returns(t, s_done[g + rq] != 0);
            }
        }
        STAMP(4);
        // E. payload scatter into the time-major SoA store (rows of consecutive envs are adjacent).  A step that ends no episode has
        //    no reward phase: the thread that stores an obs_next element builds it here (no barrier between the build and the scatter)
        for (int i = threadIdx.x; i < rows_here * D; i += NT2) {
            const int rr = i / D, k = i - rr * D, ee = rr / N;
            float v;
            if constexpr (DEFER) {
                v = mpe_obs_elem(c, s_ap + ee * st, s_av + ee * st, s_lp + ee * st, rr - ee * N, k);
                XN[rr * d.ld1 + k] = v;
            } else {
                v = XN[rr * d.ld1 + k];
            }
            const int64_t dst = (s_row[g + ee] * N + (rr - ee * N)) * D + k;
            a.obs_store[dst] = lds[xcur + rr * d.ld1 + k];
            if (a.obs_next_store) a.obs_next_store[dst] = v;
        }
        if (ur >= 0 && ur < rows_here) {   // (the rows' payload on wave 5: wave 0 has its observation elements to store)
            const int pel = ur / N, pai = ur - pel * N;
            const int64_t dst = s_row[g + pel] * N + pai;
            a.act_store[dst] = s_act[ur];
            if (now) a.rew_store[dst] = s_rew[ur];
            a.term_store[dst] = 0;
            a.trunc_store[dst] = (uint8_t)s_done[g + pel];
            if (a.logp_store) a.logp_store[dst] = s_logp[ur];
            if (a.vs_store) a.vs_store[dst] = s_val[ur];
        }
        if constexpr (DEFER) __syncthreads();   // the next forward (or F's) reads the obs_next rows at once
        pending = !now;
        STAMP(5);
        // F. finished episodes: critic value of the terminal observation, then re-initialise the env
        if (any_done) {
            if (a.vnext_store) {
                lyf.X = xnxt;
                if constexpr (NT2 == 2 * NT) tile_forward_split<H>(lds, lyf, d);
                else tile_forward<H>(lds, lyf, d);
                if (lane_live && s_done[g + el]) a.vnext_store[s_row[g + el] * N + ai] = lds[ly.OUT + r * ly.ldo + 16];
            }
            if (a.auto_reset) {
                if (env_lane && s_done[g + bel]) {
                    const uint64_t ep = a.episode_ctr[be];
                    s_ep[bel] = ep;
                    a.episode_ctr[be] = ep + 1;
                    s_steps[bel] = 0;
                }
                __syncthreads();
                if (lane_live && s_done[g + el])
                    mpe_reset_agent(c, e, a.env_seed, s_ep[el], ai, s_ap + el * st, s_av + el * st, s_lp + el * st);
                __syncthreads();
                // first observation of the new episodes
                for (int i = threadIdx.x; i < rows_here * D; i += NT2) {
                    const int rr = i / D, k = i - rr * D, ee = rr / N;
                    if (s_done[g + ee])
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
    }
    if (ret_lane) {
#pragma unroll
        for (int k = 0; k < kMpeMaxN; ++k) if (k < N) vs.ep_return[(int64_t)(e0 + rq) * N + k] = v_epret[k];
    }
    // the last workgroup to get here advances the sampling counter: every workgroup read it (off0) before finishing
    if (a.done_ctr && threadIdx.x == 0) {
        if (atomicAdd(a.done_ctr, 1u) == gridDim.x - 1) {
            *a.offset_dev_rw += a.offset_inc;
            *a.done_ctr = 0u;
        }
    }
}

// ---- wave-autonomous form ------------------------------------------------------------------------------------------------
// rollout_kernel above runs one 16-row tile per workgroup on eight waves: every phase of a vector step (forward layer by layer,
// heads, env step, scatter) is a workgroup-wide phase behind a barrier, ~10 barriers and 8.8 us per step (6.8 us once compiled for the
// job's dimensions, 4.4-4.8 us since round 5 took the pair forces, the index algebra and the sampling uniforms off its dependent
// chain) at BASELINE configs[1]
// (1024 envs x 3 agents: 205 workgroups), with the matrix pipe busy 7 % of the time.  Here ONE WAVE owns the 16 / N environments
// (<= 16 agent rows) for all T steps and runs a step from the observation to the buffer rows by itself, with no barrier and no
// cross-wave hand-over: both nets' layers as transposed products on the wave's samples (wave_mlp.h: weights = A operand read
// from the very LDS image the tile form stages, samples = B operand in registers), the critic's value as the tile form's 64-long
// fmaf chain on one lane per row, heads one lane per row (the arithmetic of categorical.hip, as in rollout_rows.hip), pair
// forces as 16 x 8 tasks over the 64 lanes folded in partner order.  Bit-identical to the tile form and to the unfused launch
// sequence (tests/test_gpu_pipeline.py, tests/test_gpu_rollout.py).  A workgroup is 1, 2 or 4 such waves sharing one weight image.
struct Rw64Lay {  // offsets (floats) inside a wave's private LDS block
    static constexpr int kLdc = 68;  // critic layer-2 row: 64 + 4 (16-byte aligned rows, conflict-free 16-byte reads)
    int AP, AV, LP, LG, H2C, VAL, CX, CY, CV, MIN, REW, LOGP, ACT, STEPS, DONE, ROW, EP, size;
    __host__ __device__ Rw64Lay() {
        int q = 0;
        AP = q; q += 2 * R;
        AV = q; q += 2 * R;
        LP = q; q += 2 * R;
        LG = q; q += R * 16;          // logits [16 rows][16]
        H2C = q; q += R * kLdc;       // critic layer-2 activations [16 rows][64]
        VAL = q; q += R;
        CX = q; q += R * 8;
        CY = q; q += R * 8;
        CV = q; q += R * 8;           // int
        MIN = q; q += R;
        REW = q; q += R;
        LOGP = q; q += R;
        ACT = q; q += R;              // int
        STEPS = q; q += R;            // int [envs of the wave]
        DONE = q; q += R;             // int
        ROW = q; q += 2 * R;          // int64
        EP = q; q += 2 * R;           // uint64
        size = (q + 3) & ~3;
    }
};

template <int NJ, int DS = 0, int NS = 0>   // DS / NS: as in rollout_kernel (compiled for obs width 18 / 3 agents)
__global__ __launch_bounds__(NT) void rollout_wave64_kernel(RolloutArgs a) {
    constexpr int H = 64, MB = H / 16, KB1 = 4 * NJ, KBH = H / 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const Dims d = DS ? dims_const(DS, 5) : a.d;
    MpeCfg c_ = a.c;
    if (NS) { c_.N = NS; c_.obs_dim = DS; }
    const MpeCfg c = c_;
    const Lay<H> ly(d, false);
    const Rw64Lay wl;
    constexpr int ld1 = 16 * NJ + 2;   // == d.ld1
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n_waves = blockDim.x >> 6, c16 = lane & 15, kq = lane >> 4;
    const int N = c.N, D = d.D, A = d.A, st = 2 * N;
    const int EPW = R / N;                                   // whole environments per wave
    const int e0 = (blockIdx.x * n_waves + w) * EPW;         // first env of this wave
    const int n_here = max(0, min(EPW, c.n_env - e0));
    const int rows_here = n_here * N;
    const int64_t B = c.n_env;
    // ---- the weight image (exactly Lay<64>'s [0, X): W1 actor | critic, W2a, W2c, W3a, W3c, biases) ----
    const int img_floats = image_f4_padded(ly.X) * 4;        // the DMA copies whole chunks (zeros behind ly.X)
    if (a.img) {
        const int n4p = img_floats / 4, wave_base = w * 64;
        for (int e = 0; e < n4p; e += blockDim.x)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.img + (size_t)(e + tid) * 4),
                                             (__attribute__((address_space(3))) void *)(lds + (size_t)(e + wave_base) * 4), 16, 0, 0);
    } else {
        for (int v = tid; v < NT; v += blockDim.x) stage_weights<H>(lds, ly, d, a.P, v);   // (the work of virtual thread v of NT)
    }
    const int wvo = img_floats + w * wl.size;
    float *wv = lds + wvo;
    float *s_ap = wv + wl.AP, *s_av = wv + wl.AV, *s_lp = wv + wl.LP, *s_lg = wv + wl.LG, *s_h2c = wv + wl.H2C, *s_val = wv + wl.VAL,
          *s_cx = wv + wl.CX, *s_cy = wv + wl.CY, *s_m = wv + wl.MIN, *s_rew = wv + wl.REW, *s_logp = wv + wl.LOGP;
    int *s_cv = reinterpret_cast<int *>(wv + wl.CV), *s_act = reinterpret_cast<int *>(wv + wl.ACT),
        *s_steps = reinterpret_cast<int *>(wv + wl.STEPS), *s_done = reinterpret_cast<int *>(wv + wl.DONE);
    int64_t *s_row = reinterpret_cast<int64_t *>(wv + wl.ROW);
    uint64_t *s_ep = reinterpret_cast<uint64_t *>(wv + wl.EP);

    const VrbState vs = vrb_view(a.vrb_state, B, N);
    // agent lane r = lane < rows_here <-> (env el, agent ai) of the wave; env lane 16 + bel owns env bel's bookkeeping
    const int r = lane, el = r / N, ai = r - el * N;
    const bool lane_live = r < rows_here;
    const int e = e0 + el;
    const int bel = lane - 16;
    const bool env_lane = bel >= 0 && bel < n_here;
    const int be = e0 + bel;
    int64_t v_ins = 0, v_size = 0, v_eplen = 0, v_epstart = 0, v_last = 0;
    int n_fin = 0;
    double v_epret[kMpeMaxN];
#pragma unroll
    for (int k = 0; k < kMpeMaxN; ++k) v_epret[k] = 0.0;
    if (env_lane) {
        v_ins = vs.ins[be]; v_size = vs.size[be]; v_eplen = vs.ep_len[be]; v_epstart = vs.ep_start[be];
        v_last = vs.last_index[be];
#pragma unroll
        for (int k = 0; k < kMpeMaxN; ++k) if (k < N) v_epret[k] = vs.ep_return[(int64_t)be * N + k];
    }
    float st_ap = 0.f, st_av = 0.f, st_lp = 0.f;   // (the wave's env state: loaded here, stored to LDS behind the barrier)
    if (lane < n_here * st) {
        st_ap = a.apos[(int64_t)e0 * st + lane]; st_av = a.avel[(int64_t)e0 * st + lane]; st_lp = a.lpos[(int64_t)e0 * st + lane];
    }
    const int steps0 = env_lane ? a.steps[be] : 0;
    const uint64_t off0 = a.offset + (a.offset_dev ? *a.offset_dev : 0ull);
    __syncthreads();   // weights staged (the DMA may have written zeros past the image: private blocks are initialised below);
                       // every wave has read the sampling counter.  The only workgroup barrier.
    if (n_here == 0) return;  // a wave without environments (never wave 0, whose thread 0 updates the counter at the end)
    if (env_lane) { s_steps[bel] = steps0; s_done[bel] = 1; s_row[bel] = 0; }   // done = 1: "no pending V(obs_next)"
    if (lane < n_here * st) { s_ap[lane] = st_ap; s_av[lane] = st_av; s_lp[lane] = st_lp; }
    if (lane < 16) s_lg[16 * 16 - 1 - lane] = 0.f;   // (the zero cell of the observation fragments lives in the last logits row pad)

    // ---- observation fragments: element k = 4 kb + kq of sample c16 (one LDS value or the difference of two; the two offsets
    //      are worked out once; a zero cell stands in where there is no subtrahend, for k >= D and for samples >= rows_here) ----
    const int zoff = wvo + wl.LG + 16 * 15 + 15;   // row 15, column 15 of the logits block: rows_here <= 15 whenever N does not
                                                   // divide 16 ... and for N | 16 column 15 >= A is never written: always 0.f
    uint32_t el_ab[KB1];
    const int sn_el = c16 / N, sn_ai = c16 - sn_el * N;
    const bool sn_live = c16 < rows_here;
#pragma unroll
    for (int kb = 0; kb < KB1; ++kb) {
        const int k = 4 * kb + kq;
        int oa = zoff, ob = zoff;
        if (sn_live && k < D) {
            const int base = sn_el * st, i_ = sn_ai;
            if (k < 2) oa = wvo + wl.AV + base + 2 * i_ + k;
            else if (k < 4) oa = wvo + wl.AP + base + 2 * i_ + (k - 2);
            else {
                int kk = k - 4;
                if (kk < 2 * N) { oa = wvo + wl.LP + base + kk; ob = wvo + wl.AP + base + 2 * i_ + (kk & 1); }
                else {
                    kk -= 2 * N;
                    if (kk < 2 * (N - 1)) {
                        int jj = kk >> 1;
                        const int x = kk & 1;
                        if (jj >= i_) ++jj;  // others in increasing index, skipping self
                        oa = wvo + wl.AP + base + 2 * jj + x; ob = wvo + wl.AP + base + 2 * i_ + x;
                    }
                }
            }
        }
        el_ab[kb] = ((uint32_t)oa << 16) | (uint32_t)ob;
    }
    const int ND = N * D;
    const int obs_col = sn_ai * D + kq;
    auto obs_frags = [&](float (&x)[KB1]) {
        float xa[KB1], xs[KB1];
#pragma unroll
        for (int kb = 0; kb < KB1; ++kb) { xa[kb] = lds[el_ab[kb] >> 16]; xs[kb] = lds[el_ab[kb] & 0xFFFFu]; }
#pragma unroll
        for (int kb = 0; kb < KB1; ++kb) x[kb] = xa[kb] - xs[kb];
    };
    const int kb_full = D >> 2, k_rem = D & 3;
    auto obs_rows_out = [&](float *dst, const float (&x)[KB1]) {
#pragma unroll
        for (int kb = 0; kb < KB1; ++kb) {
            if (kb < kb_full) dst[4 * kb] = x[kb];
            else if (kb == kb_full && kq < k_rem) dst[4 * kb] = x[kb];
        }
    };
    int pair_ei[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int rp = (lane + 64 * q) >> 3, ep = rp / N;
        pair_ei[q] = rp < rows_here ? (ep << 3) | (rp - ep * N) : -1;
    }
    const int kb1_n = d.Kp1 >> 2;   // k-steps of layer 1 (the tile form's k extent: D rounded up to 4)
    const float *w1a = lds + ly.W1 + c16 * ld1 + kq, *w1c = w1a + H * ld1;
    const float *w2a = lds + ly.W2a + c16 * ly.ldh + kq, *w2c = lds + ly.W2c + c16 * ly.ldh + kq, *w3a = lds + ly.W3a + c16 * ly.ldh + kq;
    // V(row) of the wave's samples from the observation fragments: layers 1, 2 as products, layer 3 as mlp_tile.h's fmaf chain
    // (s = fmaf(h[j], w3[j], s), j = 0 .. 63, + b3) on lane `row`; returned on lanes < 16
    auto critic_value = [&](const float (&x)[KB1]) -> float {
        wf4 acc[MB];
        float hb[4 * MB];
        wave_layer<MB, KB1>(w1c, ld1, x, kb1_n, acc);
        wave_bias_relu<MB, true>(lds + ly.B1 + H + 4 * kq, acc);
        wave_to_frags<MB>(acc, hb);
        wave_layer<MB, KBH>(w2c, ly.ldh, hb, KBH, acc);
        wave_bias_relu<MB, true>(lds + ly.B2 + H + 4 * kq, acc);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) *reinterpret_cast<wf4 *>(s_h2c + c16 * Rw64Lay::kLdc + 16 * mb + 4 * kq) = acc[mb];
        float s = 0.f;
        if (lane < R) {
            wf4 hv[H / 4], wq[H / 4];
#pragma unroll
            for (int q = 0; q < H / 4; ++q) {
                hv[q] = *reinterpret_cast<const wf4 *>(s_h2c + lane * Rw64Lay::kLdc + 4 * q);
                wq[q] = *reinterpret_cast<const wf4 *>(lds + ly.W3c + 4 * q);
            }
#pragma unroll
            for (int j = 0; j < H; ++j) s = fmaf(hv[j >> 2][j & 3], wq[j >> 2][j & 3], s);
            s += lds[ly.B3c];
        }
        return s;
    };

    for (int t = 0; t <= a.n_steps; ++t) {
        const bool last = t == a.n_steps;  // extra pass: bootstrap value of the final observation only
        // the previous step's row slot / done flag of this lane's env (V(obs_next) of step t - 1 == V(obs) of step t)
        int64_t p_row = 0;
        int p_done = 1;
        if (lane_live) { p_row = s_row[el]; p_done = s_done[el]; }
        // ---- A. buffer index algebra of this step on the env lanes ----
        bool tr = false, rec = false;
        int64_t o = 0;
        if (env_lane && !last) {
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
            // (the record carries CollectStats.lens = len(episode_batch): the episode's rows IN THE BUFFER, collector.py:203,990-993 --
            //  after a reset_buffer(keep_statistics=True) an episode counts its rows since the reset; ep_len_out stays add()'s ep_len)
            if (rec) a.ep_rec[B + (int64_t)be * a.max_ep + n_fin] = ((int64_t)t << 32) | ((cur >= v_epstart ? cur - v_epstart : cur - v_epstart + a.S) + 1);
            a.ep_len_out[o] = tr ? elen : 0;
            a.ptr_out[o] = cur + (int64_t)be * a.S;
            a.ep_idx_out[o] = v_epstart + (int64_t)be * a.S;
            v_ins = nxt; v_size = sz; v_eplen = tr ? 0 : elen; v_epstart = tr ? nxt : v_epstart;
            v_last = cur + (int64_t)be * a.S;
            a.done_store[cur * B + be] = tr ? 1 : 0;
            s_row[bel] = cur * B + be;
            s_done[bel] = tr ? 1 : 0;
        }
        // ---- B. observation fragments (also the buffer's obs rows), critic value, actor logits ----
        float xb[KB1];
        obs_frags(xb);
        if (last) {
            if (a.obs_cur_out && sn_live) obs_rows_out(a.obs_cur_out + ((int64_t)e0 * N + c16) * D + kq, xb);
        } else if (sn_live) {
            obs_rows_out(a.obs_store + s_row[sn_el] * ND + obs_col, xb);
        }
        const float val = critic_value(xb);
        if (lane_live && !p_done && a.vnext_store) a.vnext_store[p_row * N + ai] = val;
        if (last) break;
        {
            wf4 acc[MB];
            float hb[4 * MB];
            wave_layer<MB, KB1>(w1a, ld1, xb, kb1_n, acc);
            wave_bias_relu<MB, true>(lds + ly.B1 + 4 * kq, acc);
            wave_to_frags<MB>(acc, hb);
            wave_layer<MB, KBH>(w2a, ly.ldh, hb, KBH, acc);
            wave_bias_relu<MB, true>(lds + ly.B2 + 4 * kq, acc);
            wave_to_frags<MB>(acc, hb);
            wf4 lg = wf4{0.f, 0.f, 0.f, 0.f};   // logits (A padded to 16): lane holds actions 4 kq + i of sample c16
#pragma unroll
            for (int kb = 0; kb < KBH; ++kb) lg = wave_mfma4(w3a[4 * kb], hb[kb], lg);
            const wf4 b = *reinterpret_cast<const wf4 *>(lds + ly.B3a + 4 * kq);
            // (row 15 / column 15 holds the zero cell: never a live logit -- A <= 8 here; the write below keeps it 0 + 0)
            if (!(c16 == 15 && kq == 3))
                *reinterpret_cast<wf4 *>(s_lg + c16 * 16 + 4 * kq) = wf4{lg[0] + b[0], lg[1] + b[1], lg[2] + b[2], lg[3] + b[3]};
        }
        // ---- C. heads: one lane per row, the arithmetic of categorical.hip ----
        if (lane_live) {
            float lgv[8], ex[8];
            {
                const wf4 l0 = *reinterpret_cast<const wf4 *>(s_lg + lane * 16), l1 = *reinterpret_cast<const wf4 *>(s_lg + lane * 16 + 4);
                lgv[0] = l0[0]; lgv[1] = l0[1]; lgv[2] = l0[2]; lgv[3] = l0[3]; lgv[4] = l1[0]; lgv[5] = l1[1]; lgv[6] = l1[2]; lgv[7] = l1[3];
            }
            float m = -INFINITY;
            int arg = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) if (j < A && lgv[j] > m) { m = lgv[j]; arg = j; }
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { ex[j] = 0.f; if (j < A) { ex[j] = expf(lgv[j] - m); s += ex[j]; } }
            const float lse = m + logf(s);
            int act = arg;
            if (a.mode == 1) {
                const uint64_t gi = (uint64_t)e0 * N + (uint64_t)lane;  // global row env * N + agent
                const float u = tsm_philox_uniform(a.pol_seed, off0 + (uint64_t)t * B * N + gi) * s;
                float cs = 0.f;
                bool found = false;
                act = A - 1;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < A) {
                        cs += ex[j];
                        if (!found && u < cs) { act = j; found = true; }
                    }
            }
            float la = lgv[0];
#pragma unroll
            for (int j = 1; j < 8; ++j) la = act == j ? lgv[j] : la;
            s_act[lane] = act;
            s_logp[lane] = la - lse;
            s_val[lane] = val;
        }
        // ---- D. env step (mpe_dev.h): 16 x 8 pair tasks over the 64 lanes, folded by the agent lanes in partner order ----
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int p = lane + 64 * q, jp = p & 7;
            float sx = 0.f, sy = 0.f;
            int ok = 0;
            if (pair_ei[q] >= 0 && jp < N) {
                const int ip = pair_ei[q] & 7;
                const float *ap = s_ap + (pair_ei[q] >> 3) * st;
                if (jp != ip) ok = mpe_pair_force(c, ap[2 * ip], ap[2 * ip + 1], ap[2 * jp], ap[2 * jp + 1], ip, jp, sx, sy) ? 1 : 0;
            }
            s_cx[p] = sx; s_cy[p] = sy; s_cv[p] = ok;
        }
        if (lane_live) {
            const float px = s_ap[el * st + 2 * ai], py = s_ap[el * st + 2 * ai + 1];
            const float vx = s_av[el * st + 2 * ai], vy = s_av[el * st + 2 * ai + 1];
            float fx = mpe_action_force(c, s_act[r], 0);
            float fy = mpe_action_force(c, s_act[r], 1);
            {
                typedef int i4 __attribute__((ext_vector_type(4)));
                const i4 v0 = *reinterpret_cast<const i4 *>(s_cv + 8 * r), v1 = *reinterpret_cast<const i4 *>(s_cv + 8 * r + 4);
                const wf4 x0 = *reinterpret_cast<const wf4 *>(s_cx + 8 * r), x1 = *reinterpret_cast<const wf4 *>(s_cx + 8 * r + 4);
                const wf4 y0 = *reinterpret_cast<const wf4 *>(s_cy + 8 * r), y1 = *reinterpret_cast<const wf4 *>(s_cy + 8 * r + 4);
#pragma unroll
                for (int j = 0; j < kMpeMaxN; ++j) {
                    const int ok = j < 4 ? v0[j & 3] : v1[j & 3];
                    if (j < N && ok) { fx += j < 4 ? x0[j & 3] : x1[j & 3]; fy += j < 4 ? y0[j & 3] : y1[j & 3]; }
                }
            }
            float npx, npy, nvx, nvy;
            mpe_integrate(c, px, py, vx, vy, fx, fy, npx, npy, nvx, nvy);
            s_ap[el * st + 2 * ai] = npx; s_ap[el * st + 2 * ai + 1] = npy;
            s_av[el * st + 2 * ai] = nvx; s_av[el * st + 2 * ai + 1] = nvy;
        }
        float local = 0.f;
        if (lane_live) {
            const MpePos pos = mpe_load_pos(c, s_ap + el * st);
            s_m[r] = mpe_landmark_min_dist(c, pos, s_lp + el * st, ai);
            local = mpe_local_penalty(c, pos, s_ap + el * st, ai);
        }
        // obs_next fragments (the terminal observation for finished episodes): into the buffer, and kept for F
        float xn[KB1];
        const unsigned long long done_mask = __ballot(lane_live && s_done[el]);
        if (a.obs_next_store || (done_mask && a.vnext_store)) obs_frags(xn);
        if (a.obs_next_store && sn_live) obs_rows_out(a.obs_next_store + s_row[sn_el] * ND + obs_col, xn);
        if (lane_live) {
            float mv[kMpeMaxN];
#pragma unroll
            for (int l = 0; l < kMpeMaxN; ++l) mv[l] = s_m[el * N + (l < N ? l : N - 1)];
            float global = 0.f;
#pragma unroll
            for (int l = 0; l < kMpeMaxN; ++l) if (l < N) global -= mv[l];
            s_rew[r] = global * (1.f - c.local_ratio) + local * c.local_ratio;
        }
        if (env_lane) {  // episode returns
            double *rec_rew = rec ? reinterpret_cast<double *>(a.ep_rec + B + (int64_t)B * a.max_ep) +
                                        ((int64_t)be * a.max_ep + n_fin) * N : nullptr;
#pragma unroll
            for (int k = 0; k < kMpeMaxN; ++k) {
                if (k < N) {
                    const double ac = v_epret[k] + (double)s_rew[bel * N + k];
                    a.ep_rew_out[o * N + k] = tr ? ac : 0.0;
                    if (rec) rec_rew[k] = ac;
                    v_epret[k] = tr ? 0.0 : ac;
                }
            }
            n_fin += tr ? 1 : 0;
        }
        // ---- E. payload ----
        if (lane_live) {
            const int64_t dst = s_row[el] * N + ai;
            a.act_store[dst] = s_act[r];
            a.rew_store[dst] = s_rew[r];
            a.term_store[dst] = 0;
            a.trunc_store[dst] = (uint8_t)s_done[el];
            if (a.logp_store) a.logp_store[dst] = s_logp[r];
            if (a.vs_store) a.vs_store[dst] = s_val[r];
        }
        // ---- F. finished episodes: critic value of the terminal observation, then re-initialise the env ----
        if (done_mask) {   // (wave-uniform)
            if (a.vnext_store) {
                const float vt = critic_value(xn);
                if (lane_live && s_done[el]) a.vnext_store[s_row[el] * N + ai] = vt;
            }
            if (a.auto_reset) {
                if (env_lane && s_done[bel]) {
                    const uint64_t ep = a.episode_ctr[be];
                    s_ep[bel] = ep;
                    a.episode_ctr[be] = ep + 1;
                    s_steps[bel] = 0;
                }
                if (lane_live && s_done[el])
                    mpe_reset_agent(c, e, a.env_seed, s_ep[el], ai, s_ap + el * st, s_av + el * st, s_lp + el * st);
            }
        }
    }
    // env state + sub-buffer bookkeeping back to HBM
    if (lane < n_here * st) {
        a.apos[(int64_t)e0 * st + lane] = s_ap[lane];
        a.avel[(int64_t)e0 * st + lane] = s_av[lane];
        a.lpos[(int64_t)e0 * st + lane] = s_lp[lane];
    }
    if (env_lane) {
        a.steps[be] = s_steps[bel];
        vs.ins[be] = v_ins; vs.size[be] = v_size; vs.ep_len[be] = v_eplen; vs.ep_start[be] = v_epstart;
        vs.last_index[be] = v_last; vs.lengths[be] = v_size;
        if (a.ep_rec) a.ep_rec[be] = n_fin;  // may exceed max_ep: the host treats that as an overflow
#pragma unroll
        for (int k = 0; k < kMpeMaxN; ++k) if (k < N) vs.ep_return[(int64_t)be * N + k] = v_epret[k];
    }
    if (a.done_ctr && tid == 0) {  // the last workgroup advances the sampling counter (every wave has read it: the barrier above)
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
    {   // Which form (option "rollout_form": 0 by rule, 1 tile, 2 wave).  Rule: the tile form while its workgroups (one 16-row tile
        // each) fit the CUs in one round, the wave-autonomous form above that.  Measured (us per 25-step collect + reset_buffer,
        // tile / wave, tools/time_rollout64.py, kernels compiled for obs width 18 / 3 agents where they apply): 512 envs x 3 agents
        // 276 / 316; 1024 x 3 299 / 336; 1536 x 3 422 / 354; 2048 x 3 408 / 373; 4096 x 3 681 / 490; 1024 x 8 697 / 506;
        // 4096 x 8 2423 / 989 -- one wave running a step by itself is slower than eight cooperating ones, but it does not queue
        // behind a second round of workgroups.
        const int form = tsm_opt(TSM_OPT_ROLLOUT_FORM);
        const int64_t total_waves = ceil_div(a.c.n_env, R / a.c.N);
        // (observation widths above 48 -- nine or ten agents -- stay on the tile form: the wave form's four-block instantiation
        //  spilled ten vector registers, tools/resource_usage.py, and was dropped in round 5)
        if (a.d.nJ <= 3 && (form == 2 || (form == 0 && total_waves > 256))) {
            int n_waves = 1;
            while (n_waves < 4 && ceil_div(total_waves, n_waves) > 256) n_waves *= 2;
            const Rw64Lay wl;
            const size_t shmem_w = ((size_t)image_f4_padded(ly.X) * 4 + (size_t)n_waves * wl.size) * sizeof(float);
            TSM_REQUIRE(shmem_w <= kTsmMaxLds, "tsm_rollout_spread: LDS layout of %zu bytes does not fit", shmem_w);
            const unsigned n_wg_w = (unsigned)ceil_div(total_waves, n_waves);
            static bool attr_w[4] = {false, false, false, false};
#define LAUNCHW(NJ)                                                                                                    \
    do {                                                                                                               \
        if (!attr_w[NJ - 1]) {                                                                                         \
            TSM_HIP(tsm_allow_max_lds(reinterpret_cast<const void *>(rollout_wave64_kernel<NJ>)));                    \
            attr_w[NJ - 1] = true;                                                                                     \
        }                                                                                                              \
        hipLaunchKernelGGL((rollout_wave64_kernel<NJ>), dim3(n_wg_w), dim3(64 * n_waves), shmem_w, tsm_stream(stream), a); \
    } while (0)
            if (a.d.D == 18 && a.d.A == 5 && a.c.N == 3 && !tsm_opt(TSM_OPT_GENERIC)) {
                static bool attr_18 = false;
                if (!attr_18) {
                    TSM_HIP(tsm_allow_max_lds(reinterpret_cast<const void *>(rollout_wave64_kernel<2, 18, 3>)));
                    attr_18 = true;
                }
                hipLaunchKernelGGL((rollout_wave64_kernel<2, 18, 3>), dim3(n_wg_w), dim3(64 * n_waves), shmem_w, tsm_stream(stream), a);
            } else
            switch (a.d.nJ) {
                case 1: LAUNCHW(1); break;
                case 2: LAUNCHW(2); break;
                default: LAUNCHW(3); break;
            }
#undef LAUNCHW
            TSM_LAUNCH_CHECK();
            return TSM_OK;
        }
    }
    const size_t extra = (size_t)R * a.d.ld1 + 3 * R * 2 + 4 * R + 4 * R + 3 * 2 * R + 8 + 3 * R * 8 + 4 + 3 * R + 2 * R + 4;
    const size_t shmem = ((size_t)ly.total + extra) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        TSM_HIP(tsm_allow_max_lds(reinterpret_cast<const void *>(rollout_kernel<64, NT>)));
        TSM_HIP(tsm_allow_max_lds(reinterpret_cast<const void *>(rollout_kernel<64, 2 * NT>)));
        TSM_HIP(tsm_allow_max_lds(reinterpret_cast<const void *>(rollout_kernel<64, 2 * NT, 18, 3>)));
        attr_set = true;
    }
    const int EPB = R / a.c.N;
    const unsigned n_wg = (unsigned)ceil_div(a.c.n_env, EPB);
    // Eight waves when a CU holds one workgroup anyway (no more workgroups than CUs, or an LDS footprint above half a
    // CU); four waves otherwise, so that two workgroups fit the register file of a CU.  Measured (us per 25-step
    // collect, 4 / 8 waves): 1024 envs x 3 agents 242 / 234; 4096 x 3: 505 / 775; 4096 x 8 (91 KB of LDS): 7100 / 2250.
    const bool eight = n_wg <= 256 || shmem > 80 * 1024;
    const bool c18 = a.d.D == 18 && a.d.A == 5 && a.c.N == 3 && !tsm_opt(TSM_OPT_GENERIC);   // ("generic_kernels": the generic form)
    if (eight && c18) hipLaunchKernelGGL((rollout_kernel<64, 2 * NT, 18, 3>), dim3(n_wg), dim3(2 * NT), shmem, tsm_stream(stream), a);
    else if (eight) hipLaunchKernelGGL((rollout_kernel<64, 2 * NT>), dim3(n_wg), dim3(2 * NT), shmem, tsm_stream(stream), a);
    else hipLaunchKernelGGL((rollout_kernel<64, NT>), dim3(n_wg), dim3(NT), shmem, tsm_stream(stream), a);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

