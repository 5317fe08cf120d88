// rollout_tag.hip -- persistent rollout kernel for the two-team env (simple_tag): the whole Collector loop of one
// collect(n_step) call under GROUPED policies (one policy per team) in ONE launch.
//
// BASELINE configs[4] (/root/reference paths): FlexibleMultiAgentPolicyManager(mode="grouped") forwards every team's
// agents through that team's policy (algorithm/multiagent/flexible_policy.py:96-98, marl.py:137-190), the vector env
// steps every world (env/venvs.py:237-322 over enhanced_pettingzoo_env.py:175-222) and the buffer manager adds the joint
// rows (data/buffer/manager.py:131-193) -- data/collector.py:854-1069 per vector step.  Fused exactly as rollout.hip
// fuses the shared-policy loop, with two differences:
//   * TWO parameter sets live in LDS (team 0 = adversaries = agent columns [0, n_adv), team 1 = good agents); both
//     nets run on the workgroup's 16-row tile and every row takes the outputs of its own team's net.  A tile row's
//     result does not depend on the other rows of the tile, so logits / values are bit-identical to tsm_policy_forward
//     on the team's own rows;
//   * the sampling counter of (step t, agent column a, env e) is  offset[team] + *offset_dev + t n_env NA + a n_env + e:
//     agent-major inside a step, as MultiAgentPolicy.act_device hands the columns of a team to its policy (one shared
//     policy sees the [n_env][NA] rows in one call instead: env_major_counter, e NA + a).
// A workgroup owns EPB = 16 / NA whole worlds for all T steps: weights staged once, env state in LDS, only the buffer
// rows travel to HBM.  Results are bit-identical to the unfused sequence (per team: tsm_policy_forward) ->
// tsm_mpe_tag_step -> tsm_vrb_add (tests/test_gpu_tag.py).  V(obs_next) is not produced (the league / self-play trainers
// learn from per-agent batches and run their own critic passes: training_coordinator.py:160-179).
#include "common.h"
#include "mlp_tile.h"
#include "mpe_tag_dev.h"
#include "philox.h"
#include "vrb_dev.h"

int tsm_mpe_tag_check_cfg(const tsm_mpe_tag_cfg *h, TagCfg *c);  // mpe_tag.hip

extern long long *g_tsm_stamps;  // abi.hip (diagnostics)

namespace {

struct TagRolloutArgs {
    // policies: [0] adversaries, [1] good agents
    const float *P[2];
    uint64_t pol_seed[2], offset[2];
    int mode[2];  // 1 sample, 2 argmax
    int env_major;  // sampling counter inside a step: 0 column * n_env + env (grouped), 1 env * NA + column (one shared policy)
    Dims d;
    const uint64_t *offset_dev;
    // env
    TagCfg c;
    uint64_t env_seed;
    uint64_t *episode_ctr;
    float *apos, *avel, *lpos;
    int32_t *steps;
    int auto_reset;
    float *obs_cur_out;  // [n_env][NA][D] next policy input after the rollout
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
    // compact record of the episodes finished during this rollout (nullable), layout as in rollout.hip
    int64_t *ep_rec;
    int max_ep;
    uint64_t offset_inc;
    uint64_t *offset_dev_rw;
    uint32_t *done_ctr;
    long long *stamps;  // diagnostics only (tsm_debug_set_stamps, tools/stamp_rollout_tag.py): phase time stamps of workgroup 0
};


#define TSTAMP(k) do { if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0 && t < 4) a.stamps[t * 8 + (k)] = (long long)wall_clock64(); } while (0)

constexpr int NT2 = 2 * NT;  // eight waves: actor chains on waves 0-3, critic chains on waves 4-7 (tile_forward_split)

template <int H>
__global__ __launch_bounds__(NT2) void rollout_tag_kernel(TagRolloutArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const Dims d = a.d;
    const TagCfg c = a.c;
    const Lay<H> ly(d, false);                    // team 0: weights + the shared activation buffers
    const int wB = ly.total;                      // team 1: weights only, the image layout of [0, ly.X) behind team 0
    const int w_size = (ly.X + 3) & ~3;
    Lay<H> lyB = ly;
    lyB.W1 += wB; lyB.W2a += wB; lyB.W2c += wB; lyB.W3a += wB; lyB.W3c += wB;
    lyB.B1 += wB; lyB.B2 += wB; lyB.B3a += wB; lyB.B3c += wB;
    lyB.OUT = wB + w_size;                        // [R][ldo] team 1's logits | value
    const int xn0 = lyB.OUT + R * ly.ldo;         // [R][ld1] second observation tile
    const int NA = c.n_adv + c.n_good, D = d.D, st = 2 * NA, lst = 2 * c.n_obst;
    const int EPB = R / NA;                       // worlds per workgroup
    const int e0 = blockIdx.x * EPB;
    const int n_here = min(EPB, c.n_env - e0);
    const int rows_here = n_here * NA;            // live tile rows (<= 16): row r = (env el, agent i), r = el*NA + i
    const int64_t B = c.n_env;
    float *s_ap = lds + xn0 + R * d.ld1;          // [EPB][NA][2]
    float *s_av = s_ap + R * 2;
    float *s_lp = s_av + R * 2;                   // [EPB][n_obst][2]  (EPB * n_obst <= 8 * 4)
    float *s_rew = s_lp + 2 * 8 * kTagMaxObst + R;  // [R]  (R floats in front of it are unused: the team's hit terms travel by shuffle)
    float *s_logp = s_rew + R;                    // [R]
    float *s_val = s_logp + R;                    // [R]
    int *s_act = reinterpret_cast<int *>(s_val + R);        // [R]
    int *s_steps = s_act + R;                               // [EPB]
    int *s_done = s_steps + R;                              // [2][EPB] done flag of the step just taken (generation t & 1)
    // (aligned by index arithmetic off the 16-byte aligned base: an address rounded through uintptr_t is a generic pointer, its
    //  accesses FLAT instructions that wait for every outstanding global store -- see csrc/rollout.hip)
    int64_t *s_row = reinterpret_cast<int64_t *>(                    // [EPB] slot*B + env of the step just added
        lds + (((int)(reinterpret_cast<float *>(s_done + 2 * R) - lds) + 1) & ~1));   // (two generations, see csrc/rollout.hip)
    uint64_t *s_ep = reinterpret_cast<uint64_t *>(s_row + 2 * R);    // [EPB] episode counter of finished envs
    constexpr int kPS = 16;                                          // pair slots per row (entities: <= 8 agents + 4 obstacles)
    float *s_cx = lds + (((int)(reinterpret_cast<float *>(s_ep + R) - lds) + 3) & ~3);   // [R][kPS] pair forces, phase D; 16-byte rows
    float *s_cy = s_cx + R * kPS;
    int *s_cv = reinterpret_cast<int *>(s_cy + R * kPS);             // [R][kPS] "the pair is in range"
    float *s_u = reinterpret_cast<float *>(s_cv + R * kPS);          // [2][R] the sampling uniforms of step t (generation t & 1)
    int *s_any = reinterpret_cast<int *>(s_u + 2 * R);               // [2] "an episode of this tile ends at step t"

    if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[40] = (long long)wall_clock64();
    // waves 0-3 stage team 0's weights, waves 4-7 team 1's (batches of 4 loads per thread: ~1 us per workgroup)
    if (threadIdx.x < NT) stage_weights<H>(lds, ly, d, a.P[0]);
    else stage_weights<H>(lds + wB, ly, d, a.P[1], (int)threadIdx.x - NT);
    for (int i = threadIdx.x; i < R * d.ld1; i += NT2) { lds[ly.X + i] = 0.f; lds[xn0 + i] = 0.f; }
    const VrbState vs = vrb_view(a.vrb_state, B, NA);
    // agent lane r < rows_here (wave 0) <-> (env el, agent ai); env lane 256 + q (wave 4) owns env q's bookkeeping; beside the head on
    // waves 0-3 run the index algebra (wave 4), the next step's uniforms (wave 5) and this step's pair forces (waves 6, 7)
    const int r = threadIdx.x, el = r / NA, ai = r - el * NA;
    const bool lane_live = r < rows_here;
    const int e = e0 + el;
    const int bel = (int)threadIdx.x - 256;  // env lane: local env index
    const bool env_lane = bel >= 0 && bel < n_here;
    const int be = e0 + bel;
    int64_t v_ins = 0, v_size = 0, v_eplen = 0, v_epstart = 0, v_last = 0;
    int n_fin = 0;  // episodes this env finished during the rollout
    double v_epret[kTagMaxAgents];
#pragma unroll
    for (int k = 0; k < kTagMaxAgents; ++k) v_epret[k] = 0.0;
    if (env_lane) {
        v_ins = vs.ins[be]; v_size = vs.size[be]; v_eplen = vs.ep_len[be]; v_epstart = vs.ep_start[be];
        v_last = vs.last_index[be];
#pragma unroll
        for (int k = 0; k < kTagMaxAgents; ++k) if (k < NA) v_epret[k] = vs.ep_return[(int64_t)be * NA + k];
        s_steps[bel] = a.steps[be];
        s_done[bel] = 0; s_done[R + bel] = 0;
        s_row[bel] = 0; s_row[R + bel] = 0;
    }
    for (int i = threadIdx.x; i < n_here * st; i += NT2) {
        s_ap[i] = a.apos[(int64_t)e0 * st + i];
        s_av[i] = a.avel[(int64_t)e0 * st + i];
    }
    for (int i = threadIdx.x; i < n_here * lst; i += NT2) s_lp[i] = a.lpos[(int64_t)e0 * lst + i];
    // the pair force tasks (agent row, other entity) of a step, NE per row, compacted over the 128 lanes of waves 6 and 7 (two passes
    // cover 16 rows x 12 entities): packed (env << 8) | (agent << 4) | entity, -1 = no task
    const int NE = NA + c.n_obst;
    int pair_id[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int p = (int)threadIdx.x - 384 + 128 * q;
        const int rp = p / NE, ep = rp / NA;
        pair_id[q] = (threadIdx.x >= 384 && p < rows_here * NE) ? (ep << 8) | ((rp - ep * NA) << 4) | (p - rp * NE) : -1;
    }
    const uint64_t off0 = a.offset_dev ? *a.offset_dev : 0ull;
    // the sampling uniform of (step, row) drawn one step ahead by wave 5, beside the head (csrc/rollout.hip has the story)
    const int ur = (int)threadIdx.x - 320, u_el = ur / NA, u_ai = ur - u_el * NA;
    const int u_team = u_ai >= c.n_adv ? 1 : 0;
    const bool u_lane = ur >= 0 && ur < rows_here && a.mode[u_team] == 1;
    const uint64_t u_gi = a.env_major ? (uint64_t)(e0 + u_el) * NA + u_ai : (uint64_t)u_ai * B + (uint64_t)(e0 + u_el);
    if (u_lane) s_u[ur] = tsm_philox_uniform(a.pol_seed[u_team], a.offset[u_team] + off0 + u_gi);
    __syncthreads();
    // two observation tiles, swapped every step: obs_next of step t IS obs of step t + 1 (rows of re-initialised
    // envs are rebuilt in F), so the observation function runs once per step, not twice
    Lay<H> lyf = ly, lyg = lyB;
    int xcur = ly.X, xnxt = xn0;
    // A. observation rows from the LDS-resident state: one (row, element) per thread
    for (int i = threadIdx.x; i < rows_here * D; i += NT2) {
        const int rr = i / D, k = i - rr * D, ee = rr / NA;
        lds[xcur + rr * d.ld1 + k] = tag_obs_elem(c, s_ap + ee * st, s_av + ee * st, s_lp + ee * lst, rr - ee * NA, k);
    }
    __syncthreads();

    if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[41] = (long long)wall_clock64();
    for (int t = 0; t < a.n_steps; ++t) {
        const int g = (t & 1) * R;   // this step's generation of s_row / s_done; g ^ R: the previous step's
        float *XN = lds + xnxt;
        TSTAMP(0);
        // B. both teams' actor + critic forward of the 16-row tile (same X, separate OUT)
        lyf.X = xcur; lyg.X = xcur;
        // (round 4: both teams' nets side by side -- waves 0-3 team 0, waves 4-7 team 1, two chains per wave, three barriers instead
        //  of six -- was built, bit-identical, and changed nothing: collect 0.276 ms either way; the two passes stay)
        tile_forward_split<H>(lds, lyf, d);
        tile_forward_split<H>(lds, lyg, d);
        TSTAMP(1);
        bool tr = false, rec = false;
        int64_t o = 0;
        if (env_lane) {   // the buffer index algebra of this step: wave 4, beside the head on waves 0-3 (D .. F read generation t & 1)
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
            s_row[g + bel] = cur * B + be;
            s_done[g + bel] = tr ? 1 : 0;
            const unsigned long long ends = __ballot(tr);   // (the env lanes share wave 4)
            if (bel == 0) s_any[t & 1] = ends != 0ull;
        }
        if (u_lane && t + 1 < a.n_steps)
            s_u[((t + 1) & 1) * R + ur] = tsm_philox_uniform(a.pol_seed[u_team], a.offset[u_team] + off0 + (uint64_t)(t + 1) * B * NA + u_gi);
        // The pair forces of THIS step depend on the positions alone, not on the actions the head is sampling: waves 6 and 7 evaluate
        // them beside the head, one (agent row, other entity) task per lane, and phase D starts at the fold (csrc/rollout.hip has the
        // measurement; they ran as a phase of their own behind the head's barrier).
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (pair_id[q] >= 0) {
                const int ep = pair_id[q] >> 8, ip = (pair_id[q] >> 4) & 15, jp = pair_id[q] & 15;
                float sx = 0.f, sy = 0.f;
                int ok = 0;
                if (jp != ip) ok = tag_pair_force(c, s_ap + ep * st, s_lp + ep * lst, ip, jp, sx, sy) ? 1 : 0;
                const int slot = kPS * (ep * NA + ip) + jp;
                s_cx[slot] = sx; s_cy[slot] = sy; s_cv[slot] = ok;
            }
        // C. head: 16 lanes per row compute exp(logit - max) in parallel; lane 0 of the row then folds them in
        //    action order (same arithmetic order as tsm_policy_forward => identical samples and log-probs)
        if (threadIdx.x < NT) {
            const int hr = threadIdx.x >> 4, j = threadIdx.x & 15;
            const int hel = hr / NA, hai = hr - hel * NA;
            const int team = hai >= c.n_adv ? 1 : 0;
            const float *lg = lds + (team ? lyB.OUT : ly.OUT) + hr * ly.ldo;
            const bool on = j < d.A;
            const float x = on ? lg[j] : -INFINITY;
            const float m = row16_max(x);
            const float ex = on ? expf(x - m) : 0.f;
            const int gbase = threadIdx.x & 48;  // first lane of this row's 16-lane group inside the wave
            float ssum = 0.f;
            row_prefix_sum<0>(ex, d.A, ssum);
            int act = 0;
            if (a.mode[team] == 1) {
                // (tsm_philox_uniform(pol_seed[team], offset[team] + off0 + t * B * NA + gi), gi = (e0 + hel) * NA + hai for one policy
                //  call over [n_env][NA] rows, hai * B + e0 + hel for one call per column: drawn ahead, s_u)
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
                s_val[hr] = lg[16];
                s_act[hr] = act;
                s_logp[hr] = la - (m + logf(ssum));
                // V(obs_next) of the previous step's row == V(obs) of this one where the episode goes on (s_row / s_done still hold
                // the previous step: D rewrites them behind the barrier below)
                if (a.vnext_store && t > 0 && !s_done[(g ^ R) + hel]) a.vnext_store[s_row[(g ^ R) + hel] * NA + hai] = lg[16];
            }
        }
        __syncthreads();
        TSTAMP(2);
        // D. env step (mpe_tag_dev.h): fold of the pair forces (evaluated beside the head, above: one (agent row, other entity) task
        //    per lane instead of one lane per agent walking its 5 partners -- sqrt, exp, log1p, a division each when the pair is in
        //    range), integrate, publish -> barrier -> reward terms.
        //    (The env lanes' buffer index algebra on their register-resident sub-buffer state -- buffer_base.py:373-410 +
        //    manager.py:170-177; same arithmetic as vrb_add_row in vrb_dev.h -- runs beside the head too.)
        float npx = 0.f, npy = 0.f;
        if (lane_live) {   // fold, integrate, publish: a lane reads and writes its own agent's position / velocity only
            float fx, fy, nvx, nvy;
            tag_action_force(c, ai, s_act[r], fx, fy);
#pragma unroll
            for (int q = 0; q < kPS / 4; ++q) {
                typedef int i4 __attribute__((ext_vector_type(4)));
                const i4 v = *reinterpret_cast<const i4 *>(s_cv + kPS * r + 4 * q);
                const f4 x = *reinterpret_cast<const f4 *>(s_cx + kPS * r + 4 * q), y = *reinterpret_cast<const f4 *>(s_cy + kPS * r + 4 * q);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (4 * q + u < NE && v[u]) { fx += x[u]; fy += y[u]; }
            }
            tag_integrate(c, ai, s_ap[el * st + 2 * ai], s_ap[el * st + 2 * ai + 1], s_av[el * st + 2 * ai], s_av[el * st + 2 * ai + 1], fx, fy,
                          npx, npy, nvx, nvy);
            s_ap[el * st + 2 * ai] = npx; s_ap[el * st + 2 * ai + 1] = npy;
            s_av[el * st + 2 * ai] = nvx; s_av[el * st + 2 * ai + 1] = nvy;
        }
        __syncthreads();
        // reward terms on wave 0 (the agent lanes); obs_next rows (terminal observation for finished episodes) on waves 1-7 beside it
        float my_rew = 0.f;
        if (threadIdx.x < 64) {
            float hit = 0.f;
            if (lane_live) my_rew = tag_own_reward(c, s_ap + el * st, ai, npx, npy, hit);
            // an adversary adds the hits of its env's good agents in agent order (shared by the team): the agent lanes of a tile are
            // lanes of this one wave, so the terms come by shuffle (they went through LDS and a barrier)
            for (int gg = c.n_adv; gg < NA; ++gg) {
                const float h = __shfl(hit, el * NA + gg, 64);
                if (ai < c.n_adv) my_rew += h;
            }
            if (lane_live) s_rew[r] = my_rew;
        } else {
            for (int i = threadIdx.x - 64; i < rows_here * D; i += NT2 - 64) {
                const int rr = i / D, k = i - rr * D, ee = rr / NA;
                XN[rr * d.ld1 + k] = tag_obs_elem(c, s_ap + ee * st, s_av + ee * st, s_lp + ee * lst, rr - ee * NA, k);
            }
        }
        __syncthreads();
        if (env_lane) {  // episode returns (needs the rewards); runs beside the payload scatter below
            double *rec_rew = rec ? reinterpret_cast<double *>(a.ep_rec + B + (int64_t)B * a.max_ep) +
                                        ((int64_t)be * a.max_ep + n_fin) * NA : nullptr;
#pragma unroll
            for (int k = 0; k < kTagMaxAgents; ++k) {
                if (k < NA) {
                    const double acc = v_epret[k] + (double)s_rew[bel * NA + k];
                    a.ep_rew_out[o * NA + k] = tr ? acc : 0.0;
                    if (rec) rec_rew[k] = acc;
                    v_epret[k] = tr ? 0.0 : acc;
                }
            }
            n_fin += tr ? 1 : 0;
        }
        TSTAMP(3);
        // E. payload scatter into the time-major SoA store (rows of consecutive envs are adjacent)
        for (int i = threadIdx.x; i < rows_here * D; i += NT2) {
            const int rr = i / D, k = i - rr * D, ee = rr / NA;
            const int64_t dst = (s_row[g + ee] * NA + (rr - ee * NA)) * D + k;
            a.obs_store[dst] = lds[xcur + rr * d.ld1 + k];
            if (a.obs_next_store) a.obs_next_store[dst] = XN[rr * d.ld1 + k];
        }
        if (lane_live) {
            const int64_t dst = s_row[g + el] * NA + ai;
            a.act_store[dst] = s_act[r];
            a.rew_store[dst] = s_rew[r];
            a.term_store[dst] = 0;
            a.trunc_store[dst] = (uint8_t)s_done[g + el];
            if (a.logp_store) a.logp_store[dst] = s_logp[r];
            if (a.vs_store) a.vs_store[dst] = s_val[r];
        }
        TSTAMP(4);
        // F. finished episodes: the critic value of the terminal observation (both teams' nets on the obs_next tile, every row takes
        //    its own team's), then re-initialise the env, first observation of the new episode
        if (a.auto_reset || a.vnext_store) {
            const int any_done = s_any[t & 1];   // (left by the env lanes beside the index algebra: no workgroup-wide OR)
            if (any_done && a.vnext_store) {
                lyf.X = xnxt; lyg.X = xnxt;
                tile_forward_split<H>(lds, lyf, d);
                tile_forward_split<H>(lds, lyg, d);
                if (lane_live && s_done[g + el])
                    a.vnext_store[s_row[g + el] * NA + ai] = lds[(ai >= c.n_adv ? lyB.OUT : ly.OUT) + r * ly.ldo + 16];
            }
            if (any_done && a.auto_reset) {
                if (env_lane && s_done[g + bel]) {
                    const uint64_t ep = a.episode_ctr[be];
                    s_ep[bel] = ep;
                    a.episode_ctr[be] = ep + 1;
                    s_steps[bel] = 0;
                }
                __syncthreads();
                if (lane_live && s_done[g + el])
                    tag_reset_lane(c, e, a.env_seed, s_ep[el], ai, s_ap + el * st, s_av + el * st, s_lp + el * lst);
                __syncthreads();
                for (int i = threadIdx.x; i < rows_here * D; i += NT2) {
                    const int rr = i / D, k = i - rr * D, ee = rr / NA;
                    if (s_done[g + ee])
                        XN[rr * d.ld1 + k] =
                            tag_obs_elem(c, s_ap + ee * st, s_av + ee * st, s_lp + ee * lst, rr - ee * NA, k);
                }
            }
        }
        __syncthreads();  // the next forward (or the epilogue) reads the tile at once; s_* of this step are free again
        TSTAMP(5);
        { const int tmp = xcur; xcur = xnxt; xnxt = tmp; }
    }
    if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[42] = (long long)wall_clock64();
    // V(obs_next) of the final step's rows whose episode goes on: one more pass over the observation the next collect starts from
    if (a.vnext_store) {
        const int gl = ((a.n_steps - 1) & 1) * R;   // the final step's generation
        int open_rows = lane_live ? !s_done[gl + el] : 0;
        open_rows = __syncthreads_or(open_rows);
        if (open_rows) {
            lyf.X = xcur; lyg.X = xcur;
            tile_forward_split<H>(lds, lyf, d);
            tile_forward_split<H>(lds, lyg, d);
            if (lane_live && !s_done[gl + el])
                a.vnext_store[s_row[gl + el] * NA + ai] = lds[(ai >= c.n_adv ? lyB.OUT : ly.OUT) + r * ly.ldo + 16];
        }
    }
    // the observation of the next collect() call
    if (a.obs_cur_out)
        for (int i = threadIdx.x; i < rows_here * D; i += NT2) {
            const int rr = i / D, k = i - rr * D;
            a.obs_cur_out[((int64_t)e0 * NA + rr) * D + k] = lds[xcur + rr * d.ld1 + k];
        }
    // env state + sub-buffer bookkeeping back to HBM
    for (int i = threadIdx.x; i < n_here * st; i += NT2) {
        a.apos[(int64_t)e0 * st + i] = s_ap[i];
        a.avel[(int64_t)e0 * st + i] = s_av[i];
    }
    for (int i = threadIdx.x; i < n_here * lst; i += NT2) a.lpos[(int64_t)e0 * lst + i] = s_lp[i];
    if (env_lane) {
        a.steps[be] = s_steps[bel];
        vs.ins[be] = v_ins; vs.size[be] = v_size; vs.ep_len[be] = v_eplen; vs.ep_start[be] = v_epstart;
        vs.last_index[be] = v_last; vs.lengths[be] = v_size;
        if (a.ep_rec) a.ep_rec[be] = n_fin;  // may exceed max_ep: the host treats that as an overflow
#pragma unroll
        for (int k = 0; k < kTagMaxAgents; ++k) if (k < NA) vs.ep_return[(int64_t)be * NA + k] = v_epret[k];
    }
    // the last workgroup to get here advances the sampling counter: every workgroup read it (off0) before finishing
    if (a.done_ctr && threadIdx.x == 0) {
        if (atomicAdd(a.done_ctr, 1u) == gridDim.x - 1) {
            *a.offset_dev_rw += a.offset_inc;
            *a.done_ctr = 0u;
        }
    }
}

size_t tag_rollout_lds_floats(const Dims &d) {
    const Lay<64> ly(d, false);
    const size_t w_size = (size_t)((ly.X + 3) & ~3);
    return (size_t)ly.total + w_size + (size_t)R * ly.ldo + (size_t)R * d.ld1 + 2 * R * 2 + 2 * 8 * kTagMaxObst + 4 * R +
           3 * R + 2 * 2 * R + 8 + 3 * R * 16 + 4 + 3 * R + 2 * R + 4;
}

}  // namespace

TSM_EXPORT int tsm_rollout_tag(const tsm_rollout_tag_desc *desc_host, void *stream) {
    TSM_REQUIRE(desc_host, "tsm_rollout_tag: null descriptor");
    const tsm_rollout_tag_desc &h = *desc_host;
    TagRolloutArgs a;
    if (int rc = make_dims(h.obs_dim, h.hidden, h.n_act, &a.d)) return rc;
    if (int rc = tsm_mpe_tag_check_cfg(&h.env, &a.c)) return rc;
    const int NA = a.c.n_adv + a.c.n_good;
    TSM_REQUIRE(a.c.obs_dim == h.obs_dim, "tsm_rollout_tag: obs_dim %d != the env's padded width %d", h.obs_dim, a.c.obs_dim);
    TSM_REQUIRE(h.n_act == 5, "tsm_rollout_tag: simple_tag has 5 discrete actions");
    TSM_REQUIRE(NA <= R && R / NA <= 8, "tsm_rollout_tag: %d agents do not fit the row tile", NA);
    TSM_REQUIRE(h.n_steps >= 1 && h.sub_size >= 1, "tsm_rollout_tag: bad n_steps / sub_size");
    for (int k = 0; k < 2; ++k) {
        TSM_REQUIRE(h.mode[k] == 1 || h.mode[k] == 2, "tsm_rollout_tag: mode must be 1 (sample) or 2 (argmax)");
        TSM_REQUIRE(h.params[k], "tsm_rollout_tag: null parameters");
        a.P[k] = h.params[k]; a.pol_seed[k] = h.policy_seed[k]; a.offset[k] = h.offset[k]; a.mode[k] = h.mode[k];
    }
    TSM_REQUIRE(h.episode_ctr && h.agent_pos && h.agent_vel && h.landmark_pos && h.steps && h.vrb_state && h.done_store &&
                    h.obs_store && h.act_store && h.rew_store && h.term_store && h.trunc_store && h.ptr_out &&
                    h.ep_rew_out && h.ep_len_out && h.ep_idx_out,
                "tsm_rollout_tag: null pointer");
    a.offset_dev = h.offset_dev;
    a.env_major = h.env_major_counter ? 1 : 0;
    a.env_seed = h.env_seed; a.episode_ctr = h.episode_ctr;
    a.apos = h.agent_pos; a.avel = h.agent_vel; a.lpos = h.landmark_pos; a.steps = h.steps;
    a.auto_reset = h.auto_reset; a.obs_cur_out = h.obs_cur_out;
    a.vrb_state = h.vrb_state; a.S = h.sub_size; a.done_store = h.done_store;
    a.obs_store = h.obs_store; a.obs_next_store = h.obs_next_store; a.rew_store = h.rew_store;
    a.logp_store = h.logp_store; a.vs_store = h.vs_store; a.vnext_store = h.vnext_store;
    a.act_store = h.act_store; a.term_store = h.term_store; a.trunc_store = h.trunc_store;
    a.ptr_out = h.ptr_out; a.ep_len_out = h.ep_len_out; a.ep_idx_out = h.ep_idx_out; a.ep_rew_out = h.ep_rew_out;
    a.n_steps = h.n_steps;
    TSM_REQUIRE(!h.ep_rec || h.max_ep >= 1, "tsm_rollout_tag: ep_rec needs max_ep >= 1");
    a.ep_rec = h.ep_rec; a.max_ep = h.max_ep;
    TSM_REQUIRE(!h.done_ctr || h.offset_dev, "tsm_rollout_tag: done_ctr needs offset_dev");
    a.offset_inc = h.offset_inc; a.done_ctr = h.done_ctr;
    a.offset_dev_rw = const_cast<uint64_t *>(reinterpret_cast<const uint64_t *>(h.offset_dev));
    a.stamps = g_tsm_stamps;
    const size_t shmem = tag_rollout_lds_floats(a.d) * sizeof(float);
    TSM_REQUIRE(shmem <= kTsmMaxLds, "tsm_rollout_tag: %zu bytes of LDS needed", shmem);
    static bool attr_set = false;
    if (!attr_set) {
        TSM_HIP(tsm_allow_max_lds(reinterpret_cast<const void *>(rollout_tag_kernel<64>)));
        attr_set = true;
    }
    const int EPB = R / NA;
    const unsigned n_wg = (unsigned)ceil_div(a.c.n_env, EPB);
    hipLaunchKernelGGL((rollout_tag_kernel<64>), dim3(n_wg), dim3(NT2), shmem, tsm_stream(stream), a);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}
