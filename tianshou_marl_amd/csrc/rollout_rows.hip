// rollout_rows.hip -- persistent rollout for a 128-wide ACTOR on batched simple_spread: the whole Collector loop of one
// collect(n_step) call in ONE launch (BASELINE configs[2]: N = 8 agents, actor 48-128-128-5, 4096 envs).
//
// Same job as rollout.hip (policy forward + Categorical sample / log-prob, env.step of every env + reset of finished
// envs, buffer.add index algebra + payload scatter: /root/reference/tianshou/data/collector.py:854-1069,
// modelfree/reinforce.py:167-192, env/venvs.py:237-322, data/buffer/manager.py:131-193), for policies whose nets do not
// fit the 64-wide actor+critic tile of rollout.hip.  Only the ACTOR runs inside the rollout: a centralized critic's
// first layer alone (384 x 128 f32 = 192 KB) exceeds a CU's LDS, and the update needs the critic values of all rows in
// one batch anyway (a2c.py:121-127), so V(obs) / V(obs_next) are computed there by two large GEMM passes instead of 2 x 25
// small ones here.
//
// gfx950 mapping.  A workgroup (512 threads, one per CU) owns EPB = 128 / N whole environments for all T steps: the
// actor's weights are staged in LDS once (~100 KB), the env state (positions, velocities, landmarks) lives in LDS, and
// only buffer rows travel to HBM.  Per vector step the workgroup's rows are processed as tiles of 32 samples through the
// three layers on v_mfma_f32_16x16x4_f32 with the k order of csrc/dense.hip, so logits -- hence sampled actions and
// log-probs (Philox counter = offset + step * n_env * N + env * N + agent, arithmetic of categorical.hip) -- are
// bit-identical to the unfused sequence tsm_mlp_forward -> tsm_categorical_sample -> tsm_mpe_spread_step -> tsm_vrb_add.
#include "common.h"
#include "mpe_dev.h"
#include "philox.h"
#include "vrb_dev.h"
#include "wave_mlp.h"

int tsm_mpe_check_cfg(const tsm_mpe_cfg *h, MpeCfg *c);  // mpe.hip
extern long long *g_tsm_stamps;                               // abi.hip (diagnostics, tools/stamp_rollout_rows.py)

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kH = 128, kTile = 32, kThreads = 512, kLdh = kH + 2, kLdo = 8, kRowsWg = 128;  // kLdo: logits row (A <= 8)

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct RrLay {  // LDS layout in floats
    int nJ, ld1, W1, W2, W3, B1, B2, B3, X, H1, H2, LG, AP, AV, LP, REW, LOGP, MIN, ACT, STEPS, DONE, ROW, EP, total;
    __host__ __device__ explicit RrLay(int D) {
        nJ = (D + 15) / 16;
        ld1 = 16 * nJ + 2;
        int o = 0;
        W1 = o; o += kH * ld1;
        W2 = o; o += kH * kLdh;
        W3 = o; o += 16 * kLdh;
        B1 = o; o += kH;
        B2 = o; o += kH;
        B3 = o; o += 16;
        X = o; o += kTile * ld1;
        H1 = o; o += kTile * kLdh;
        H2 = o; o += kTile * kLdh;
        LG = o; o += kRowsWg * kLdo;  // logits of ALL rows of the step: the heads run once, after the last tile
        AP = o; o += kRowsWg * 2;
        AV = o; o += kRowsWg * 2;
        LP = o; o += kRowsWg * 2;
        REW = o; o += kRowsWg;
        LOGP = o; o += kRowsWg;
        MIN = o; o += kRowsWg;
        ACT = o; o += kRowsWg;      // int
        STEPS = o; o += kRowsWg;    // int  [EPB]
        DONE = o; o += kRowsWg;     // int  [EPB]
        o = (o + 1) & ~1;
        ROW = o; o += 2 * kRowsWg;  // int64 [EPB]
        EP = o; o += 2 * kRowsWg;   // uint64 [EPB]
        total = o;
    }
};

struct RrArgs {
    const float *P;  // actor parameters w0[H][D] b0[H] w1[H][H] b1[H] w2[A][H] b2[A]
    int D, A, mode;
    uint64_t pol_seed, offset;
    const uint64_t *offset_dev;
    MpeCfg c;
    uint64_t env_seed;
    uint64_t *episode_ctr;
    float *apos, *avel, *lpos;
    int32_t *steps;
    int auto_reset;
    float *obs_cur_out;
    void *vrb_state;
    int64_t S;
    uint8_t *done_store;
    float *obs_store, *obs_next_store, *rew_store, *logp_store;
    int32_t *act_store;
    uint8_t *term_store, *trunc_store;
    int64_t *ptr_out, *ep_len_out, *ep_idx_out;
    double *ep_rew_out;
    int n_steps;
    int64_t *ep_rec;
    int max_ep;
    uint64_t offset_inc;
    uint64_t *offset_dev_rw;
    uint32_t *done_ctr;
    long long *stamps;  // diagnostics only (tsm_debug_set_stamps): phase time stamps of workgroup 0, steps 0..3
};

// slot t * 32 + k of the stamp buffer: k = 0 step start, 1 index algebra, 2 obs(0), 3 + 3 i .. 5 + 3 i layers 1 / 2 / 3 of
// tile i, 15 heads, 16 move, 17 publish, 18 reward terms + obs_next, 19 reward, 20 stores, 21 step end
#define RSTAMP(k) do { if (a.stamps && blockIdx.x == 0 && tid == 0 && t < 4) a.stamps[t * 32 + (k)] = (long long)wall_clock64(); } while (0)

template <int NJ>
__global__ __launch_bounds__(kThreads) void rollout_rows_kernel(RrArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const MpeCfg c = a.c;
    const RrLay ly(a.D);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c16 = lane & 15, kq = lane >> 4;
    const int N = c.N, D = a.D, A = a.A, ld1 = ly.ld1, st = 2 * N;
    const int EPB = kRowsWg / N;
    const int e0 = blockIdx.x * EPB;
    const int n_here = min(EPB, c.n_env - e0);
    const int rows_here = n_here * N;
    const int n_tiles = (rows_here + kTile - 1) / kTile;
    const int64_t B = c.n_env;
    float *s_ap = lds + ly.AP, *s_av = lds + ly.AV, *s_lp = lds + ly.LP, *s_rew = lds + ly.REW, *s_logp = lds + ly.LOGP,
          *s_m = lds + ly.MIN;
    int *s_act = reinterpret_cast<int *>(lds + ly.ACT), *s_steps = reinterpret_cast<int *>(lds + ly.STEPS),
        *s_done = reinterpret_cast<int *>(lds + ly.DONE);
    int64_t *s_row = reinterpret_cast<int64_t *>(lds + ly.ROW);
    uint64_t *s_ep = reinterpret_cast<uint64_t *>(lds + ly.EP);

    // ---- weights into LDS once (zero pads: W1 columns >= D, W3 rows >= A) ----
    const int oB1 = kH * D, oW2 = oB1 + kH, oB2 = oW2 + kH * kH, oW3 = oB2 + kH, oB3 = oW3 + A * kH;
    tsm_stage_padded<kThreads>(lds + ly.W1, a.P, kH * ld1, ld1, kH, D, D);            // (batched loads: common.h)
    tsm_stage_padded<kThreads>(lds + ly.W2, a.P + oW2, kH * kLdh, kLdh, kH, kH, kH);
    tsm_stage_padded<kThreads>(lds + ly.W3, a.P + oW3, 16 * kLdh, kLdh, A, kH, kH);
    if (tid < kH) { lds[ly.B1 + tid] = a.P[oB1 + tid]; lds[ly.B2 + tid] = a.P[oB2 + tid]; }
    if (tid < 16) lds[ly.B3 + tid] = tid < A ? a.P[oB3 + tid] : 0.f;
    for (int i = tid; i < kTile * ld1; i += kThreads) lds[ly.X + i] = 0.f;

    const VrbState vs = vrb_view(a.vrb_state, B, N);
    // agent lane r (threads 0..127) <-> (env el, agent ai); env lane (threads 128..128+EPB) owns env bel's bookkeeping
    const int r = tid, el = r / N, ai = r - el * N;
    const bool lane_live = r < rows_here;
    const int e = e0 + el;
    const int bel = tid - kRowsWg;
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
        s_steps[bel] = a.steps[be];
        s_done[bel] = 0;
        s_row[bel] = 0;
    }
    for (int i = tid; i < n_here * st; i += kThreads) {
        s_ap[i] = a.apos[(int64_t)e0 * st + i];
        s_av[i] = a.avel[(int64_t)e0 * st + i];
        s_lp[i] = a.lpos[(int64_t)e0 * st + i];
    }
    const uint64_t off0 = a.offset + (a.offset_dev ? *a.offset_dev : 0ull);

    // ---- observation elements owned by this thread: element e = tid + 512 j of the workgroup's [rows][D] block ----
    // mpe_obs_elem(row, k) is one LDS value or the difference of two (landmark / other agent minus own position); the two
    // LDS offsets are worked out once here (a zero cell stands in for "no subtrahend": x - 0.f == x bit for bit), so that
    // an element costs two LDS reads and a subtraction wherever it is needed -- the buffer's obs row and the layer-1
    // tile at the start of a step, the obs_next row after the env step -- instead of the index arithmetic of
    // mpe_obs_elem (measured: 1.7-2.4 us per 32-row tile, 3 times per step).
    constexpr int kEl = 4 * NJ;
    const int zoff = ly.B3 + 15;  // always 0.f (A <= 8 < 16)
    uint32_t el_ab[kEl], el_rk[kEl];
#pragma unroll
    for (int q = 0; q < kEl; ++q) {
        const int e_ = tid + kThreads * q;
        int oa = zoff, ob = zoff;
        uint32_t rk = 0xFFFFFFFFu;
        if (e_ < rows_here * D) {
            const int rr = e_ / D, k = e_ - rr * D, ee = rr / N, i_ = rr - ee * N, base = ee * st;
            if (k < 2) oa = ly.AV + base + 2 * i_ + k;
            else if (k < 4) oa = ly.AP + base + 2 * i_ + (k - 2);
            else {
                int kk = k - 4;
                if (kk < 2 * N) { oa = ly.LP + base + kk; ob = ly.AP + base + 2 * i_ + (kk & 1); }
                else {
                    kk -= 2 * N;
                    if (kk < 2 * (N - 1)) {
                        int jj = kk >> 1;
                        const int x = kk & 1;
                        if (jj >= i_) ++jj;  // others in increasing index, skipping self
                        oa = ly.AP + base + 2 * jj + x; ob = ly.AP + base + 2 * i_ + x;
                    }
                }
            }
            rk = ((uint32_t)ee << 24) | ((uint32_t)rr << 16) | ((uint32_t)i_ << 8) | (uint32_t)k;
        }
        el_ab[q] = ((uint32_t)oa << 16) | (uint32_t)ob;
        el_rk[q] = rk;
    }
    auto obs_val = [&](int q) { return lds[el_ab[q] >> 16] - lds[el_ab[q] & 0xFFFFu]; };
    // rows of tile `tile` into the layer-1 tile X (rows beyond rows_here keep stale values: their outputs are never read)
    auto fill_x = [&](int tile) {
#pragma unroll
        for (int q = 0; q < kEl; ++q) {
            const uint32_t rk = el_rk[q];
            const int rr = (rk >> 16) & 0xFF;
            if (rk != 0xFFFFFFFFu && (rr >> 5) == tile) lds[ly.X + (rr & 31) * ld1 + (int)(rk & 0xFF)] = obs_val(q);
        }
    };
    // the pair (agent lane, other agent) tasks of the env step: task p = tid + 512 q -> agent lane p >> 3, other p & 7
    int pair_ei[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int rp = (tid + kThreads * q) >> 3, ep = rp / N;
        pair_ei[q] = rp < rows_here ? (ep << 3) | (rp - ep * N) : -1;
    }
    float *s_cx = lds + ly.H1, *s_cy = s_cx + 8 * kRowsWg;           // [128][8] pair forces (H1 is idle during the env step)
    int *s_cv = reinterpret_cast<int *>(s_cy + 8 * kRowsWg);         // [128][8] 1 = the pair is in contact range
    const int ND = N * D;
    __syncthreads();

    for (int t = 0; t < a.n_steps; ++t) {
        // ---- A. buffer index algebra of this step on the env lanes (buffer_base.py:373-410 + manager.py:170-177; same
        //         arithmetic as vrb_add_row): the slot of the row is known before the payload exists ----
        RSTAMP(0);
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
        __syncthreads();
        RSTAMP(1);
        // ---- B. the buffer's obs rows from the LDS-resident state; per 32-row tile the actor forward; then the
        //         Categorical heads of all rows.  The state does not change before the env step, so tile i + 1's rows go
        //         into X as soon as layer 1 of tile i has read it, and the logits layer of tile i (waves 0-1) runs beside
        //         layer 1 of tile i + 1: two barriers per tile.
#pragma unroll
        for (int q = 0; q < kEl; ++q) {
            const uint32_t rk = el_rk[q];
            if (rk != 0xFFFFFFFFu) {
                const float v = obs_val(q);
                a.obs_store[s_row[rk >> 24] * ND + (int)((rk >> 8) & 0xFF) * D + (int)(rk & 0xFF)] = v;
                const int rr = (rk >> 16) & 0xFF;
                if (rr < kTile) lds[ly.X + rr * ld1 + (int)(rk & 0xFF)] = v;
            }
        }
        __syncthreads();
        RSTAMP(2);
        for (int tile = 0; tile < n_tiles; ++tile) {
            const int r0 = tile * kTile;
            const int col = 16 * w + c16;
            {   // H1 = relu(X W1^T + b1)
                f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
                const float *pa = lds + ly.X + c16 * ld1 + kq;
                const float *pb = lds + ly.W1 + col * ld1 + kq;
#pragma unroll
                for (int k0 = 0; k0 < 16 * NJ; k0 += 4) {
                    const float bv = pb[k0];
                    acc[0] = mfma4(pa[k0], bv, acc[0]);
                    acc[1] = mfma4(pa[16 * ld1 + k0], bv, acc[1]);
                }
                const float bb = lds[ly.B1 + col];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float v = acc[mt][q] + bb;
                        lds[ly.H1 + (mt * 16 + kq * 4 + q) * kLdh + col] = v > 0.f ? v : 0.f;
                    }
            }
            __syncthreads();
            RSTAMP(3 + 3 * tile);
            if (tile + 1 < n_tiles) fill_x(tile + 1);
            {   // H2 = relu(H1 W2^T + b2)
                f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
                const float *pa = lds + ly.H1 + c16 * kLdh + kq;
                const float *pb = lds + ly.W2 + col * kLdh + kq;
#pragma unroll
                for (int k0 = 0; k0 < kH; k0 += 4) {
                    const float bv = pb[k0];
                    acc[0] = mfma4(pa[k0], bv, acc[0]);
                    acc[1] = mfma4(pa[16 * kLdh + k0], bv, acc[1]);
                }
                const float bb = lds[ly.B2 + col];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float v = acc[mt][q] + bb;
                        lds[ly.H2 + (mt * 16 + kq * 4 + q) * kLdh + col] = v > 0.f ? v : 0.f;
                    }
            }
            __syncthreads();
            RSTAMP(4 + 3 * tile);
            if (w < 2) {  // logits (A padded to 16): waves 0 / 1 take the two row halves
                f4 acc = f4{0.f, 0.f, 0.f, 0.f};
                const float *pa = lds + ly.H2 + (16 * w + c16) * kLdh + kq;
                const float *pb = lds + ly.W3 + c16 * kLdh + kq;
#pragma unroll
                for (int k0 = 0; k0 < kH; k0 += 4) acc = mfma4(pa[k0], pb[k0], acc);
                const float bb = lds[ly.B3 + c16];
                if (c16 < kLdo)
#pragma unroll
                    for (int q = 0; q < 4; ++q) lds[ly.LG + (r0 + 16 * w + kq * 4 + q) * kLdo + c16] = acc[q] + bb;
            }
            // no barrier here: the next tile's layer 1 reads X (written before the last barrier) and writes H1 (its
            // readers passed that barrier); its layer 2 overwrites H2 only behind the barrier that follows layer 1, which
            // waves 0-1 reach after the logits above
            RSTAMP(5 + 3 * tile);
        }
        __syncthreads();  // LG complete
        if (tid < rows_here) {  // heads: one lane per row, the arithmetic of categorical.hip
            const float *lg = lds + ly.LG + tid * kLdo;
            float m = -INFINITY;
            int arg = 0;
            for (int j = 0; j < A; ++j) { const float v = lg[j]; if (v > m) { m = v; arg = j; } }
            float s = 0.f;
            for (int j = 0; j < A; ++j) s += expf(lg[j] - m);
            const float lse = m + logf(s);
            int act = arg;
            if (a.mode == 1) {
                const uint64_t gi = (uint64_t)e0 * N + (uint64_t)tid;  // global row env * N + agent
                const float u = tsm_philox_uniform(a.pol_seed, off0 + (uint64_t)t * B * N + gi) * s;
                float cs = 0.f;
                act = A - 1;
                for (int j = 0; j < A; ++j) {
                    cs += expf(lg[j] - m);
                    if (u < cs) { act = j; break; }
                }
            }
            s_act[tid] = act;
            s_logp[tid] = lg[act] - lse;
        }
        // ---- C. env step (mpe_dev.h).  The soft contact forces -- sqrt / exp / log1p per pair in range, ~6 us when every
        //         agent lane walks its N - 1 partners -- are evaluated as 128 x 8 pair tasks over all 512 threads and
        //         folded by the agent lanes in partner order: the same additions in the same order as mpe_agent_move ----
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int p = tid + kThreads * q, jp = p & 7;
            float sx = 0.f, sy = 0.f;
            int ok = 0;
            if (pair_ei[q] >= 0 && jp < N) {
                const int ip = pair_ei[q] & 7;
                const float *ap = s_ap + (pair_ei[q] >> 3) * st;
                if (jp != ip) ok = mpe_pair_force(c, ap[2 * ip], ap[2 * ip + 1], ap[2 * jp], ap[2 * jp + 1], ip, jp, sx, sy) ? 1 : 0;
            }
            s_cx[p] = sx; s_cy[p] = sy; s_cv[p] = ok;
        }
        __syncthreads();
        RSTAMP(15);
        RSTAMP(16);
        if (lane_live) {
            const float px = s_ap[el * st + 2 * ai], py = s_ap[el * st + 2 * ai + 1];
            const float vx = s_av[el * st + 2 * ai], vy = s_av[el * st + 2 * ai + 1];
            float fx = mpe_action_force(c, s_act[r], 0);
            float fy = mpe_action_force(c, s_act[r], 1);
#pragma unroll
            for (int j = 0; j < kMpeMaxN; ++j) {
                if (j >= N) break;
                if (s_cv[8 * r + j]) { fx += s_cx[8 * r + j]; fy += s_cy[8 * r + j]; }
            }
            float npx, npy, nvx, nvy;
            mpe_integrate(c, px, py, vx, vy, fx, fy, npx, npy, nvx, nvy);
            // publish at once: the pair tasks (the only readers of other lanes' positions) are behind the barrier above
            s_ap[el * st + 2 * ai] = npx; s_ap[el * st + 2 * ai + 1] = npy;
            s_av[el * st + 2 * ai] = nvx; s_av[el * st + 2 * ai + 1] = nvy;
        }
        __syncthreads();
        RSTAMP(17);
        float local = 0.f;
        if (lane_live) {
            const MpePos pos = mpe_load_pos(c, s_ap + el * st);
            s_m[r] = mpe_landmark_min_dist(c, pos, s_lp + el * st, ai);
            local = mpe_local_penalty(c, pos, s_ap + el * st, ai);
        }
        // obs_next rows (the terminal observation for finished episodes) straight into the buffer
        if (a.obs_next_store) {
#pragma unroll
            for (int q = 0; q < kEl; ++q) {
                const uint32_t rk = el_rk[q];
                if (rk != 0xFFFFFFFFu)
                    a.obs_next_store[s_row[rk >> 24] * ND + (int)((rk >> 8) & 0xFF) * D + (int)(rk & 0xFF)] = obs_val(q);
            }
        }
        __syncthreads();
        RSTAMP(18);
        if (lane_live) s_rew[r] = mpe_reward(c, s_m + el * N, local);
        __syncthreads();
        RSTAMP(19);
        if (env_lane) {  // episode returns
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
        if (lane_live) {
            const int64_t dst = s_row[el] * N + ai;
            a.act_store[dst] = s_act[r];
            a.rew_store[dst] = s_rew[r];
            a.term_store[dst] = 0;
            a.trunc_store[dst] = (uint8_t)s_done[el];
            if (a.logp_store) a.logp_store[dst] = s_logp[r];
        }
        RSTAMP(20);
        // ---- D. finished episodes: re-initialise the env (the next step's observation rows see the new state) ----
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
        }
        __syncthreads();
        RSTAMP(21);
    }
    // the observation of the next collect() call, env state and sub-buffer bookkeeping back to HBM
    if (a.obs_cur_out) {
#pragma unroll
        for (int q = 0; q < kEl; ++q) {
            const uint32_t rk = el_rk[q];
            if (rk != 0xFFFFFFFFu)
                a.obs_cur_out[((int64_t)e0 * N + (int)((rk >> 16) & 0xFF)) * D + (int)(rk & 0xFF)] = obs_val(q);
        }
    }
    for (int i = tid; i < n_here * st; i += kThreads) {
        a.apos[(int64_t)e0 * st + i] = s_ap[i];
        a.avel[(int64_t)e0 * st + i] = s_av[i];
        a.lpos[(int64_t)e0 * st + i] = s_lp[i];
    }
    if (env_lane) {
        a.steps[be] = s_steps[bel];
        vs.ins[be] = v_ins; vs.size[be] = v_size; vs.ep_len[be] = v_eplen; vs.ep_start[be] = v_epstart;
        vs.last_index[be] = v_last; vs.lengths[be] = v_size;
        if (a.ep_rec) a.ep_rec[be] = n_fin;
#pragma unroll
        for (int k = 0; k < kMpeMaxN; ++k) if (k < N) vs.ep_return[(int64_t)be * N + k] = v_epret[k];
    }
    if (a.done_ctr && tid == 0) {  // the last workgroup advances the sampling counter (every one has read it)
        if (atomicAdd(a.done_ctr, 1u) == gridDim.x - 1) {
            *a.offset_dev_rw += a.offset_inc;
            *a.done_ctr = 0u;
        }
    }
}


// ---- wave-autonomous form ------------------------------------------------------------------------------------------------
// The tile form above is a chain of workgroup-wide phases with 2 + 2 x tiles + 7 barriers per step: every phase waits for its
// slowest wave, the single-wave phases (logits on waves 0-1, heads on two waves, bookkeeping on one) leave the others idle, and
// a step costs 25.2 us at configs[2].  Here a WAVE owns 16 / N whole environments (<= 16 agent rows) for all T steps and runs
// every phase of a step by itself: no workgroup barrier inside the step loop: 20.3 us per step (collect 631 -> 507 us).
// What this form does NOT buy (measured in round 4 with timing probes since removed from the kernel, profiles/r04_time_rollout_rows.txt, + tools/probes/mfma_valu_overlap.hip): one wave's VALU phase
// does not run under the f32 MFMAs of the other wave of its SIMD -- on gfx950 an f32 MFMA and another wave's VALU instructions
// take turns (times add: 10.9 us of MFMAs + 8.5 us of everything else per step, with or without a phase offset between the two
// waves, with or without s_setprio), so the step's floor is the sum of the two instruction streams, not their maximum.
//   * The layers run on the wave's 16 samples as transposed products with register-resident activations (wave_mlp.h).
//   * Env state, logits, pair forces and the per-env bookkeeping live in a 3 KB LDS block private to the wave; LDS accesses of
//     one wave execute in program order, so a value written by one lane is seen by the later read of another lane.
constexpr int kWaves = kThreads / 64, kRowsWave = 16;

struct RwLay {  // LDS layout in floats: weights as in RrLay, then one private block per wave
    int nJ, ld1, W1, W2, W3, B1, B2, B3, WV, wv;
    int AP, AV, LP, LG, CX, CY, CV, MIN, REW, LOGP, ACT, STEPS, DONE, ROW, EP;  // offsets inside a wave's block
    int total;
    __host__ __device__ explicit RwLay(int D) {
        nJ = (D + 15) / 16;
        ld1 = 16 * nJ + 2;
        int o = 0;
        W1 = o; o += kH * ld1;
        W2 = o; o += kH * kLdh;
        W3 = o; o += 16 * kLdh;
        B1 = o; o += kH;
        B2 = o; o += kH;
        B3 = o; o += 16;
        o = (o + 3) & ~3;
        WV = o;
        int q = 0;
        AP = q; q += 2 * kRowsWave;
        AV = q; q += 2 * kRowsWave;
        LP = q; q += 2 * kRowsWave;
        LG = q; q += kRowsWave * kLdo;
        CX = q; q += kRowsWave * 8;
        CY = q; q += kRowsWave * 8;
        CV = q; q += kRowsWave * 8;   // int
        MIN = q; q += kRowsWave;
        REW = q; q += kRowsWave;
        LOGP = q; q += kRowsWave;
        ACT = q; q += kRowsWave;      // int
        STEPS = q; q += kRowsWave;    // int [envs of the wave]
        DONE = q; q += kRowsWave;     // int
        ROW = q; q += 2 * kRowsWave;  // int64
        EP = q; q += 2 * kRowsWave;   // uint64
        wv = (q + 3) & ~3;
        total = WV + kWaves * wv;
    }
};

#define WSTAMP(k) do { if (a.stamps && blockIdx.x == 0 && tid == 0 && t < 4) a.stamps[t * 32 + (k)] = (long long)wall_clock64(); } while (0)

// NS: 0 = agent count from the launch arguments, else the count this instantiation is compiled for (BASELINE configs[2]: 8, obs
// width 48): the `j < N` predicates of the unrolled partner / landmark loops and the divisions by N and D become constants -- this
// kernel's floor is its instruction count (DESIGN.md section 4 "Issue slots and clocks").
template <int NJ, int NS = 0>
__global__ __launch_bounds__(kThreads) void rollout_wave_kernel(RrArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    MpeCfg c_ = a.c;
    if (NS) { c_.N = NS; c_.obs_dim = 6 * NS; }
    const MpeCfg c = c_;
    const RwLay ly(a.D);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c16 = lane & 15, kq = lane >> 4;
    constexpr int ld1 = 16 * NJ + 2;                     // (== ly.ld1: compile-time, so fragment offsets are immediates)
    const int N = c.N, D = NS ? 6 * NS : a.D, st = 2 * N;
    constexpr int A = 5;   // (tsm_rollout_spread_actor requires the 5 actions of simple_spread)
    const int EPW = kRowsWave / N;                       // whole environments per wave
    const int e0 = (blockIdx.x * kWaves + w) * EPW;      // first env of this wave
    const int n_here = max(0, min(EPW, c.n_env - e0));
    const int rows_here = n_here * N;
    const int64_t B = c.n_env;
    float *wv = lds + ly.WV + w * ly.wv;
    float *s_ap = wv + ly.AP, *s_av = wv + ly.AV, *s_lp = wv + ly.LP, *s_lg = wv + ly.LG, *s_cx = wv + ly.CX, *s_cy = wv + ly.CY,
          *s_m = wv + ly.MIN, *s_rew = wv + ly.REW, *s_logp = wv + ly.LOGP;
    int *s_cv = reinterpret_cast<int *>(wv + ly.CV), *s_act = reinterpret_cast<int *>(wv + ly.ACT),
        *s_steps = reinterpret_cast<int *>(wv + ly.STEPS), *s_done = reinterpret_cast<int *>(wv + ly.DONE);
    int64_t *s_row = reinterpret_cast<int64_t *>(wv + ly.ROW);
    uint64_t *s_ep = reinterpret_cast<uint64_t *>(wv + ly.EP);

    // ---- weights into LDS once (zero pads: W1 columns >= D, W3 rows >= A) ----
    const int oB1 = kH * D, oW2 = oB1 + kH, oB2 = oW2 + kH * kH, oW3 = oB2 + kH, oB3 = oW3 + A * kH;
    tsm_stage_padded<kThreads>(lds + ly.W1, a.P, kH * ld1, ld1, kH, D, D);
    tsm_stage_padded<kThreads>(lds + ly.W2, a.P + oW2, kH * kLdh, kLdh, kH, kH, kH);
    tsm_stage_padded<kThreads>(lds + ly.W3, a.P + oW3, 16 * kLdh, kLdh, A, kH, kH);
    if (tid < kH) { lds[ly.B1 + tid] = a.P[oB1 + tid]; lds[ly.B2 + tid] = a.P[oB2 + tid]; }
    if (tid < 16) lds[ly.B3 + tid] = tid < A ? a.P[oB3 + tid] : 0.f;

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
        s_steps[bel] = a.steps[be];
        s_done[bel] = 0;
        s_row[bel] = 0;
    }
    if (lane < n_here * st) {   // (n_here * st <= 32)
        s_ap[lane] = a.apos[(int64_t)e0 * st + lane];
        s_av[lane] = a.avel[(int64_t)e0 * st + lane];
        s_lp[lane] = a.lpos[(int64_t)e0 * st + lane];
    }
    const uint64_t off0 = a.offset + (a.offset_dev ? *a.offset_dev : 0ull);

    // ---- the observation elements this lane holds as B fragments: element k = 4 kb + kq of sample c16 (as in the tile form:
    //      one LDS value or the difference of two, the two offsets worked out once; a zero cell stands in where there is no
    //      subtrahend and for the pads k >= D / samples >= rows_here) ----
    constexpr int KB1 = 4 * NJ;
    const int zoff = ly.B3 + 15;  // always 0.f (A <= 8 < 16)
    uint32_t el_ab[KB1];
    const int sn_el = c16 / N, sn_ai = c16 - sn_el * N;   // (env, agent) of the lane's sample
    const bool sn_live = c16 < rows_here;
    const int wvo = ly.WV + w * ly.wv;
#pragma unroll
    for (int kb = 0; kb < KB1; ++kb) {
        const int k = 4 * kb + kq;
        int oa = zoff, ob = zoff;
        if (sn_live && k < D) {
            const int base = sn_el * st, i_ = sn_ai;
            if (k < 2) oa = wvo + ly.AV + base + 2 * i_ + k;
            else if (k < 4) oa = wvo + ly.AP + base + 2 * i_ + (k - 2);
            else {
                int kk = k - 4;
                if (kk < 2 * N) { oa = wvo + ly.LP + base + kk; ob = wvo + ly.AP + base + 2 * i_ + (kk & 1); }
                else {
                    kk -= 2 * N;
                    if (kk < 2 * (N - 1)) {
                        int jj = kk >> 1;
                        const int x = kk & 1;
                        if (jj >= i_) ++jj;  // others in increasing index, skipping self
                        oa = wvo + ly.AP + base + 2 * jj + x; ob = wvo + ly.AP + base + 2 * i_ + x;
                    }
                }
            }
        }
        el_ab[kb] = ((uint32_t)oa << 16) | (uint32_t)ob;
    }
    const int ND = N * D;
    const int obs_col = sn_ai * D + kq;   // + 4 kb: the element's place inside its env's row block
    // all 2 x KB1 LDS reads in flight, then the subtractions (written as one loop the reads are waited for pair by pair)
    auto obs_frags = [&](float (&x)[KB1]) {
        float xa[KB1], xs[KB1];
#pragma unroll
        for (int kb = 0; kb < KB1; ++kb) { xa[kb] = lds[el_ab[kb] >> 16]; xs[kb] = lds[el_ab[kb] & 0xFFFFu]; }
#pragma unroll
        for (int kb = 0; kb < KB1; ++kb) x[kb] = xa[kb] - xs[kb];
    };
    // elements k = 4 kb + kq < D of the lane's sample to row memory: whole k-steps under a wave-uniform test, the one
    // partial k-step of an odd agent count (D = 6 N, D & 3 == 2) under a lane test
    const int kb_full = D >> 2, k_rem = D & 3;
    auto obs_rows_out = [&](float *dst, const float (&x)[KB1]) {
#pragma unroll
        for (int kb = 0; kb < KB1; ++kb) {
            if (kb < kb_full) dst[4 * kb] = x[kb];
            else if (kb == kb_full && kq < k_rem) dst[4 * kb] = x[kb];
        }
    };
    // the pair (agent row, other agent) tasks of the env step: task p = lane + 64 q -> agent row p >> 3, other p & 7
    int pair_ei[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int rp = (lane + 64 * q) >> 3, ep = rp / N;
        pair_ei[q] = rp < rows_here ? (ep << 3) | (rp - ep * N) : -1;
    }
    const float *w1f = lds + ly.W1 + c16 * ld1 + kq, *w2f = lds + ly.W2 + c16 * kLdh + kq, *w3f = lds + ly.W3 + c16 * kLdh + kq;
    __syncthreads();   // weights staged; every wave has read the sampling counter.  The only workgroup barrier.
    if (n_here == 0) return;  // a wave without environments (never wave 0, whose thread 0 updates the counter below)
    for (int t = 0; t < a.n_steps; ++t) {
        // ---- A. buffer index algebra of this step on the env lanes (as in the tile form) ----
        WSTAMP(0);
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
        // ---- B. observation fragments (also the buffer's obs rows), the actor forward, the Categorical heads ----
        float xb[KB1];
        obs_frags(xb);
        if (sn_live) obs_rows_out(a.obs_store + s_row[sn_el] * ND + obs_col, xb);
        WSTAMP(1);
        f4 acc[8];
        float hb[32];
        wave_layer<8, KB1>(w1f, ld1, xb, KB1, acc);
        wave_bias_relu<8, false>(lds + ly.B1 + 4 * kq, acc);
        wave_to_frags<8>(acc, hb);
        WSTAMP(2);
        wave_layer<8, 32>(w2f, kLdh, hb, 32, acc);
        wave_bias_relu<8, false>(lds + ly.B2 + 4 * kq, acc);
        wave_to_frags<8>(acc, hb);
        WSTAMP(3);
        {   // logits (A padded to 16): lane holds actions 4 kq + i of sample c16
            f4 lg = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 32; ++kb) lg = mfma4(w3f[4 * kb], hb[kb], lg);
            if (kq < kLdo / 4) {
                const f4 b = *reinterpret_cast<const f4 *>(lds + ly.B3 + 4 * kq);
                *reinterpret_cast<f4 *>(s_lg + c16 * kLdo + 4 * kq) = f4{lg[0] + b[0], lg[1] + b[1], lg[2] + b[2], lg[3] + b[3]};
            }
        }
        WSTAMP(4);
        if (lane_live) {  // heads: one lane per row, the arithmetic of categorical.hip (all logits in registers first; the
                          // sampler re-uses the exponentials of the normaliser: the same inputs, the same values)
            float lg[kLdo], ex[kLdo];
            {
                const f4 l0 = *reinterpret_cast<const f4 *>(s_lg + lane * kLdo), l1 = *reinterpret_cast<const f4 *>(s_lg + lane * kLdo + 4);
                lg[0] = l0[0]; lg[1] = l0[1]; lg[2] = l0[2]; lg[3] = l0[3]; lg[4] = l1[0]; lg[5] = l1[1]; lg[6] = l1[2]; lg[7] = l1[3];
            }
            float m = -INFINITY;
            int arg = 0;
#pragma unroll
            for (int j = 0; j < kLdo; ++j) if (j < A && lg[j] > m) { m = lg[j]; arg = j; }
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < kLdo; ++j) { ex[j] = 0.f; if (j < A) { ex[j] = expf(lg[j] - m); s += ex[j]; } }
            const float lse = m + logf(s);
            int act = arg;
            if (a.mode == 1) {
                const uint64_t gi = (uint64_t)e0 * N + (uint64_t)lane;  // global row env * N + agent
                const float u = tsm_philox_uniform(a.pol_seed, off0 + (uint64_t)t * B * N + gi) * s;
                float cs = 0.f;
                bool found = false;
                act = A - 1;
#pragma unroll
                for (int j = 0; j < kLdo; ++j)
                    if (j < A) {
                        cs += ex[j];
                        if (!found && u < cs) { act = j; found = true; }
                    }
            }
            float la = lg[0];
#pragma unroll
            for (int j = 1; j < kLdo; ++j) la = act == j ? lg[j] : la;
            s_act[lane] = act;
            s_logp[lane] = la - lse;
        }
        WSTAMP(5);
        // ---- C. env step (mpe_dev.h): 16 x 8 pair tasks over the 64 lanes, folded by the agent lanes in partner order ----
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
        WSTAMP(6);
        if (lane_live) {
            const float px = s_ap[el * st + 2 * ai], py = s_ap[el * st + 2 * ai + 1];
            const float vx = s_av[el * st + 2 * ai], vy = s_av[el * st + 2 * ai + 1];
            float fx = mpe_action_force(c, s_act[r], 0);
            float fy = mpe_action_force(c, s_act[r], 1);
            {   // the row's 8 pair terms in six 16-byte reads, folded in partner order
                typedef int i4 __attribute__((ext_vector_type(4)));
                const i4 v0 = *reinterpret_cast<const i4 *>(s_cv + 8 * r), v1 = *reinterpret_cast<const i4 *>(s_cv + 8 * r + 4);
                const f4 x0 = *reinterpret_cast<const f4 *>(s_cx + 8 * r), x1 = *reinterpret_cast<const f4 *>(s_cx + 8 * r + 4);
                const f4 y0 = *reinterpret_cast<const f4 *>(s_cy + 8 * r), y1 = *reinterpret_cast<const f4 *>(s_cy + 8 * r + 4);
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
        WSTAMP(7);
        float local = 0.f;
        if (lane_live) {
            const MpePos pos = mpe_load_pos(c, s_ap + el * st);
            s_m[r] = mpe_landmark_min_dist(c, pos, s_lp + el * st, ai);
            local = mpe_local_penalty(c, pos, s_ap + el * st, ai);
        }
        // obs_next rows (the terminal observation for finished episodes) straight into the buffer
        if (a.obs_next_store) {
            float xn[KB1];
            obs_frags(xn);
            if (sn_live) obs_rows_out(a.obs_next_store + s_row[sn_el] * ND + obs_col, xn);
        }
        if (lane_live) {   // mpe_reward's fold in landmark order, the N minima read back to back
            float mv[kMpeMaxN];
#pragma unroll
            for (int l = 0; l < kMpeMaxN; ++l) mv[l] = s_m[el * N + (l < N ? l : N - 1)];
            float global = 0.f;
#pragma unroll
            for (int l = 0; l < kMpeMaxN; ++l) if (l < N) global -= mv[l];
            s_rew[r] = global * (1.f - c.local_ratio) + local * c.local_ratio;
        }
        WSTAMP(8);
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
        if (lane_live) {
            const int64_t dst = s_row[el] * N + ai;
            a.act_store[dst] = s_act[r];
            a.rew_store[dst] = s_rew[r];
            a.term_store[dst] = 0;
            a.trunc_store[dst] = (uint8_t)s_done[el];
            if (a.logp_store) a.logp_store[dst] = s_logp[r];
        }
        // ---- D. finished episodes: re-initialise the env (the next step's observation fragments see the new state) ----
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
        WSTAMP(9);
    }
    // the observation of the next collect() call, env state and sub-buffer bookkeeping back to HBM
    if (a.obs_cur_out) {
        float xn[KB1];
        obs_frags(xn);
        if (sn_live) obs_rows_out(a.obs_cur_out + ((int64_t)e0 * N + c16) * D + kq, xn);
    }
    if (lane < n_here * st) {
        a.apos[(int64_t)e0 * st + lane] = s_ap[lane];
        a.avel[(int64_t)e0 * st + lane] = s_av[lane];
        a.lpos[(int64_t)e0 * st + lane] = s_lp[lane];
    }
    if (env_lane) {
        a.steps[be] = s_steps[bel];
        vs.ins[be] = v_ins; vs.size[be] = v_size; vs.ep_len[be] = v_eplen; vs.ep_start[be] = v_epstart;
        vs.last_index[be] = v_last; vs.lengths[be] = v_size;
        if (a.ep_rec) a.ep_rec[be] = n_fin;
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

}  // namespace

// Same descriptor as tsm_rollout_spread; `params` = the ACTOR's parameters (hidden == 128), vs_store / vnext_store are
// not written (the critic does not run inside this rollout).
TSM_EXPORT int tsm_rollout_spread_actor(const tsm_rollout_desc *desc_host, void *stream) {
    TSM_REQUIRE(desc_host, "tsm_rollout_spread_actor: null descriptor");
    const tsm_rollout_desc &h = *desc_host;
    RrArgs a{};
    if (int rc = tsm_mpe_check_cfg(&h.env, &a.c)) return rc;
    TSM_REQUIRE(h.hidden == kH, "tsm_rollout_spread_actor: hidden must be 128 (got %d)", h.hidden);
    // (up to eight agents: the four-block instantiations for observation widths 49..64 spilled 10-59 vector registers,
    //  tools/resource_usage.py, and were dropped in round 5; the host collects such jobs through the three-launch loop)
    TSM_REQUIRE(a.c.obs_dim == h.obs_dim && h.obs_dim <= 48, "tsm_rollout_spread_actor: obs_dim %d != 6 * n_agent (<= 48)", h.obs_dim);
    TSM_REQUIRE(h.n_act == 5, "tsm_rollout_spread_actor: simple_spread has 5 discrete actions");
    TSM_REQUIRE(a.c.N >= 1 && a.c.N <= kMpeMaxN, "tsm_rollout_spread_actor: n_agent out of range");
    TSM_REQUIRE(h.n_steps >= 1 && h.sub_size >= 1, "tsm_rollout_spread_actor: bad n_steps / sub_size");
    TSM_REQUIRE(h.mode == 1 || h.mode == 2, "tsm_rollout_spread_actor: mode must be 1 (sample) or 2 (argmax)");
    TSM_REQUIRE(h.params && h.episode_ctr && h.agent_pos && h.agent_vel && h.landmark_pos && h.steps && h.vrb_state &&
                    h.done_store && h.obs_store && h.act_store && h.rew_store && h.term_store && h.trunc_store && h.ptr_out &&
                    h.ep_rew_out && h.ep_len_out && h.ep_idx_out,
                "tsm_rollout_spread_actor: null pointer");
    TSM_REQUIRE(!h.ep_rec || h.max_ep >= 1, "tsm_rollout_spread_actor: ep_rec needs max_ep >= 1");
    TSM_REQUIRE(!h.done_ctr || h.offset_dev, "tsm_rollout_spread_actor: done_ctr needs offset_dev");
    a.P = h.params; a.D = h.obs_dim; a.A = h.n_act; a.mode = h.mode;
    a.pol_seed = h.policy_seed; a.offset = h.offset; a.offset_dev = h.offset_dev;
    a.env_seed = h.env_seed; a.episode_ctr = h.episode_ctr;
    a.apos = h.agent_pos; a.avel = h.agent_vel; a.lpos = h.landmark_pos; a.steps = h.steps;
    a.auto_reset = h.auto_reset; a.obs_cur_out = h.obs_cur_out;
    a.vrb_state = h.vrb_state; a.S = h.sub_size; a.done_store = h.done_store;
    a.obs_store = h.obs_store; a.obs_next_store = h.obs_next_store; a.rew_store = h.rew_store; a.logp_store = h.logp_store;
    a.act_store = h.act_store; a.term_store = h.term_store; a.trunc_store = h.trunc_store;
    a.ptr_out = h.ptr_out; a.ep_len_out = h.ep_len_out; a.ep_idx_out = h.ep_idx_out; a.ep_rew_out = h.ep_rew_out;
    a.n_steps = h.n_steps; a.ep_rec = h.ep_rec; a.max_ep = h.max_ep;
    a.offset_inc = h.offset_inc; a.done_ctr = h.done_ctr;
    a.offset_dev_rw = const_cast<uint64_t *>(reinterpret_cast<const uint64_t *>(h.offset_dev));
    a.stamps = g_tsm_stamps;
    hipStream_t st = tsm_stream(stream);
    // the wave-autonomous form (default); "rollout_form" = 1 selects the tile form where it has a spill-free instantiation (widths
    // up to 32: its three-block form spilled ten registers)
    if (tsm_opt(TSM_OPT_ROLLOUT_FORM) != 1 || RrLay(h.obs_dim).nJ > 2) {
        const RwLay ly(h.obs_dim);
        const size_t shmem = (size_t)ly.total * sizeof(float);
        TSM_REQUIRE(shmem <= kTsmMaxLds, "tsm_rollout_spread_actor: LDS layout of %zu bytes does not fit", shmem);
        const unsigned n_wg = (unsigned)ceil_div(a.c.n_env, kWaves * (kRowsWave / a.c.N));
        static bool attr_w[4] = {false, false, false, false};
#define LAUNCHW(NJ)                                                                                                    \
    do {                                                                                                               \
        if (!attr_w[NJ - 1]) {                                                                                         \
            TSM_HIP(tsm_allow_max_lds(reinterpret_cast<const void *>(rollout_wave_kernel<NJ>)));                      \
            attr_w[NJ - 1] = true;                                                                                     \
        }                                                                                                              \
        hipLaunchKernelGGL((rollout_wave_kernel<NJ>), dim3(n_wg), dim3(kThreads), shmem, st, a);                       \
    } while (0)
        if (a.c.N == 8 && h.obs_dim == 48 && !tsm_opt(TSM_OPT_GENERIC)) {   // BASELINE configs[2] ("generic_kernels": the generic form)
            static bool attr_8 = false;
            if (!attr_8) {
                TSM_HIP(tsm_allow_max_lds(reinterpret_cast<const void *>(rollout_wave_kernel<3, 8>)));
                attr_8 = true;
            }
            hipLaunchKernelGGL((rollout_wave_kernel<3, 8>), dim3(n_wg), dim3(kThreads), shmem, st, a);
        } else {
            switch (ly.nJ) {
                case 1: LAUNCHW(1); break;
                case 2: LAUNCHW(2); break;
                default: LAUNCHW(3); break;
            }
        }
#undef LAUNCHW
        TSM_LAUNCH_CHECK();
        return TSM_OK;
    }
    const RrLay ly(h.obs_dim);
    const size_t shmem = (size_t)ly.total * sizeof(float);
    TSM_REQUIRE(shmem <= kTsmMaxLds, "tsm_rollout_spread_actor: LDS layout of %zu bytes does not fit", shmem);
    const int EPB = kRowsWg / a.c.N;
    const unsigned n_wg = (unsigned)ceil_div(a.c.n_env, EPB);
    static bool attr_set[4] = {false, false, false, false};
#define LAUNCH(NJ)                                                                                                     \
    do {                                                                                                               \
        if (!attr_set[NJ - 1]) {                                                                                       \
            TSM_HIP(tsm_allow_max_lds(reinterpret_cast<const void *>(rollout_rows_kernel<NJ>)));                      \
            attr_set[NJ - 1] = true;                                                                                   \
        }                                                                                                              \
        hipLaunchKernelGGL((rollout_rows_kernel<NJ>), dim3(n_wg), dim3(kThreads), shmem, st, a);                       \
    } while (0)
    switch (ly.nJ) {
        case 1: LAUNCH(1); break;
        default: LAUNCH(2); break;
    }
#undef LAUNCH
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}
