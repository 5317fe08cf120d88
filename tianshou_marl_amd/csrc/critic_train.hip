// critic_train.hip -- one gradient step of a 128-wide critic (in -> 128 -> 128 -> n_out) in TWO launches: (A) forward, loss,
// backward down to dH1 and every weight gradient except the first layer's; (B) dW1 = dH1^T X as a split-K pass.  Two losses:
//   LOSS 0  the value term of the PPO loss on joint rows -- one V(row) against the returns of the row's N agents
//           (/root/reference/tianshou/algorithm/modelfree/ppo.py:198-208; tsm_critic_rows_grad_ppo)
//   LOSS 1  the TD loss of CTDEPolicy.learn: values = critic(global_obs).mean(1), td_target = rew + gamma values_next
//           (1 - terminated), critic_loss = mse(values, td_target)   (tianshou/algorithm/multiagent/ctde.py:149-172;
//           tsm_critic_rows_grad_td), on rows whose successor IS the next row of the same env (chained rows).
//
// gfx950 mapping (second generation of csrc/ppo_rows.hip's critic kernel; what changed and why):
//   * Kernel A is csrc/critic_rows.hip's forward pipeline with the backward pass behind it.  Wave w owns output columns
//     [16 w, 16 w + 16); its B-operand fragments of W1 (K1 / 4 registers per lane) are loaded ONCE and stay in registers for
//     every tile of the launch; the tile's observation block sits in LDS with XOR-swizzled 16-B chunks (conflict-free
//     ds_read_b128) and the NEXT tile's rows are fetched into registers meanwhile: layer 1 has no staging copies and no
//     barriers (the first-generation kernel: 12 K-slices x (global -> registers -> LDS -> barrier), twice per tile).
//   * The layer-1 weight gradient is NOT accumulated per workgroup any more.  A rank-32 update written as a full
//     128 x K1 slab per workgroup was 196 KB x 256 workgroups = 50 MB per step at K1 = 384 (114 MB of HBM traffic per launch
//     by the PMC counters against 13 MB of algorithmic bytes), and its 96 accumulator registers were what kept the weights
//     out of registers.  Kernel A publishes dH1 (128 floats per row); kernel B (csrc/critic_dw1.hip) computes
//     dW1 = dH1^T X over row chunks and writes a few dozen partial slabs (split-K).
//   * dW2 (8 accumulator tiles per wave), dW3 and the biases stay in registers across all tiles of the workgroup and are
//     written once, as the workgroup's slab of the REST of the parameters (b1 | W2 | b2 | W3 | b3).
//   * Layer 3 (n_out <= 16 outputs, padded to one 16-column MFMA tile) and its backward are MFMA like the actor's logits.
//   * LOSS 1 needs V of a row's SUCCESSOR: a tile owns 31 consecutive env-major rows and computes the 32nd as a halo, so
//     the target of every owned row comes out of the same forward pass (3 % redundant rows instead of a second pass over
//     all of them); the last row of an env's block takes V(obs_next) from `v_last` (tsm_critic_rows_forward on those rows).
#include "critic_rows_dev.h"

extern long long *g_tsm_stamps;  // abi.hip (diagnostics, tools/stamp_critic_train.py)

namespace {

constexpr int kLdo = 18;

struct TrainLay {  // LDS layout in floats
    int ldx, W2, W3, X, H1, H2, B1, B2, B3, Q, V, DV, RID, RIDC, RED, PV, total;
    __host__ __device__ explicit TrainLay(int KJ) {
        ldx = ((16 * KJ + 63) / 64) * 64;
        int o = 0;
        W2 = o; o += kH * kLdh;
        W3 = o; o += 16 * kLdh;
        X = o; o += kRows * ldx;
        H1 = o; o += kRows * kLdh;
        H2 = o; o += kRows * kLdh;
        B1 = o; o += kH;
        B2 = o; o += kH;
        B3 = o; o += 16;
        Q = o; o += kRows * kLdo;       // layer-3 outputs, then d loss / d outputs in place
        V = o; o += kRows;              // row values (LOSS 1)
        DV = o; o += kRows * 16;        // LOSS 0: d loss / d value of a row's agents (N <= 16); final reduction scratch
        RID = o; o += 4 * kRows;        // row ids of the tile being FETCHED (int64), two buffers by tile parity
        RIDC = o; o += 2 * kRows;       // row ids of the tile being COMPUTED (int64)
        PV = Q;                         // LOSS 0: the eight waves' partial sums of V(row) [8][32] (layer 3 folded into layer 2);
                                        // read by the loss head BEFORE the barrier behind which it writes d loss / d value to Q
        RED = DV;
        total = o;
    }
    __device__ void launder() {         // (critic_rows_dev.h: opaque_s)
        W2 = opaque_s(W2); W3 = opaque_s(W3); X = opaque_s(X); H1 = opaque_s(H1); H2 = opaque_s(H2); B1 = opaque_s(B1);
        B2 = opaque_s(B2); B3 = opaque_s(B3); Q = opaque_s(Q); V = opaque_s(V); DV = opaque_s(DV); RID = opaque_s(RID);
        RIDC = opaque_s(RIDC); PV = Q; RED = DV;
    }
};

struct TrainArgs {
    const float *P;          // critic parameters: w0[H][K1] b0[H] w1[H][H] b1[H] w2[n_out][H] b2[n_out]
    const float *w1_img;     // nullable: w0 in fragment order [8 waves][KJ][64 lanes][4] (tsm_critic_rows_w1_image)
    const float *obs;        // rows [n][K1]
    const int64_t *rows;     // row ids of the minibatch (nullable)
    int64_t first_row, Mr;   // Mr: rows in the minibatch
    int64_t tm_T, tm_E;      // rows == NULL and tm_T > 0: minibatch row i is store row (i % tm_T) * tm_E + i / tm_T (env-major
                             // order over a time-major store); else first_row + i
    int K1, n_out;
    // LOSS 0: PPO value term
    const float *returns, *v_s_old;   // per SAMPLE (lane id = row * N + agent)
    int N;
    float eps_clip, vf_coef;
    int value_clip;
    // LOSS 1: TD
    const float *rew;        // scalar of store row r: rew[r * sc_stride + sc_off] (one agent's column of the joint rows)
    const uint8_t *term;
    int64_t sc_stride, sc_off;
    const float *v_last;     // [Mr / tm_T]: V(obs_next) of the last row of every env block
    const float *v_full;     // nullable [Mr]: V(obs_next) of every row (env-major), used INSTEAD of the next row's value when
    const int32_t *use_full; //   *use_full != 0 (an episode ended before the last slot: rows there are not chained)
    float gamma;
    float *dh1;              // [Mr][128]: d loss / d (layer-1 pre-activation) of minibatch row i (kernel B's A operand)
    float *h1_out, *dh2_out; // nullable pair [Mr][128]: PUBLISH layer-1 activations and d loss / d (layer-2 pre-activation) of
                             // row i -- dW2 = dH2^T H1 is then kernel B's too (split-K) and this launch keeps no W2 gradient
    float *slabs;            // [grid][P - 128 K1]: b1 | W2 | b2 | W3 | b3 gradients of the workgroup
    double *partial;         // [grid][4]: LOSS 0 {0, sum vf, 0, 0}; LOSS 1 {sum (td - v), sum (v - td)^2, 0, 0}
    long long *stamps;
};

#define TSTAMP(k) do { if (g.stamps && blockIdx.x == 0 && tid == 0 && it < 3) g.stamps[300 + 16 * it + (k)] = (long long)wall_clock64(); } while (0)

// Store row of minibatch row i (clamped to a valid row; masked by the caller).  LOSS 1 walks the env-major view of a
// time-major store: computed, no load (32-bit arithmetic: Mr < 2^31).  LOSS 0: an id list or a contiguous range -- written
// as a uniform BRANCH, not a select, so that the loaded id is not consumed (and waited for) where it is requested.
template <int LOSS>
__device__ __forceinline__ int64_t row_of(const TrainArgs &g, int64_t i) {
    const int64_t ic = i < g.Mr ? i : g.Mr - 1;  // (Mr >= 1)
    if constexpr (LOSS == 1) {
        const int t_ = (int)g.tm_T, ii = (int)ic;
        return (int64_t)(ii % t_) * g.tm_E + ii / t_;
    } else {
        int64_t r;
        if (g.rows) r = g.rows[ic];
        else r = g.first_row + ic;
        return r;
    }
}

template <int KJ, bool VEC, int LOSS, bool PUB>
__global__ __launch_bounds__(kThreads) void critic_rows_train_kernel(TrainArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    TrainLay ly(KJ);
    constexpr int ldx = ((16 * KJ + 63) / 64) * 64;
    ly.launder();
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c16 = lane & 15, kq = lane >> 4;
    const int K1 = g.K1, n_out = g.n_out, N = g.N;
    constexpr int OWN = LOSS == 1 ? kRows - 1 : kRows;  // rows a tile owns (LOSS 1: the 32nd is the halo)
    const int64_t n_tiles = (g.Mr + OWN - 1) / OWN;
    const int oB1 = kH * K1, oW2 = oB1 + kH, oB2 = oW2 + kH * kH, oW3 = oB2 + kH, oB3 = oW3 + n_out * kH;
    const int col = 16 * w + c16;
    // H1 / dH2 published for kernel B's dW2 (LOSS 0 only).  A template parameter: as a run-time branch its two stores cost the
    // default form 31 spilled registers at K1 = 384 (25.0 -> 26.9 us per launch).
    constexpr bool pub = PUB;
    if (g.stamps && blockIdx.x == 0 && tid == 0) g.stamps[298] = (long long)wall_clock64();

    // row-id pipeline of threads < 32: ids of row tid of the tiles t (being computed) .. t + 3.  Requested FIRST: id -> row is
    // a dependent round trip, and the weight loads below cover it
    const int64_t gs = gridDim.x;
    int64_t tile = blockIdx.x;
    int64_t id0 = 0, id1 = 0, id2 = 0, id3 = 0;   // (LOSS 0 only: LOSS 1 computes its row ids where it needs them)
    if (tid < kRows) {
        id0 = row_of<LOSS>(g, tile * OWN + tid);
        id1 = row_of<LOSS>(g, (tile + gs) * OWN + tid);
        if constexpr (LOSS == 0) {
            id2 = row_of<LOSS>(g, (tile + 2 * gs) * OWN + tid);
            id3 = row_of<LOSS>(g, (tile + 3 * gs) * OWN + tid);
        }
    }
    // ---- this wave's fragment of W1, once, into registers: from the fragment image when there is one (1 KB of consecutive
    // bytes per wave instruction; the gather from the row-major matrix touches sixteen 1.5 KB rows per 16-lane group and
    // costs 3.8 us more at K1 = 384: profiles/r04_critic_prologue_experiment.txt) ----
    f4 w1f[KJ];
    if (g.w1_img) {
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            const float4 q = reinterpret_cast<const float4 *>(g.w1_img)[(size_t)(w * KJ + j) * 64 + lane];
            w1f[j] = f4{q.x, q.y, q.z, q.w};
        }
    } else {
#pragma unroll
        for (int j = 0; j < KJ; ++j) w1f[j] = load_w1_frag<VEC>(g.P + (size_t)col * K1, 16 * j + 4 * kq, K1);
    }
    float4 w2q[8];            // W2 requested here, stored to LDS after the first tile's rows have been requested too (requesting
    w2_load(w2q, g.P + oW2);  // it BEHIND the rows was measured 1 us slower: the LDS stores then wait for the rows as well)
    // W3 (rows >= n_out zero) and the biases as one BATCH of loads in flight (a load -> LDS-store loop pays a memory round trip
    // per iteration: five of them here)
    constexpr int kN3 = (16 * kLdh + kThreads - 1) / kThreads;
    float w3q[kN3], bq1 = 0.f, bq2 = 0.f, bq3 = 0.f;
#pragma unroll
    for (int u = 0; u < kN3; ++u) {
        const int e = tid + u * kThreads, r = e / kLdh, c = e - r * kLdh;
        const bool ok = e < 16 * kLdh && r < n_out && c < kH;
        const float v = g.P[ok ? oW3 + r * kH + c : 0];   // (clamped, always-valid address + select)
        w3q[u] = ok ? v : 0.f;
    }
    if (tid < kH) { bq1 = g.P[oB1 + tid]; bq2 = g.P[oB2 + tid]; }
    if (tid < 16) bq3 = tid < n_out ? g.P[oB3 + tid] : 0.f;

    // ---- staging of a tile (as csrc/critic_rows.hip): thread -> chunks q = tid + 512 u of the 32 x (4 KJ) chunk grid ----
    constexpr int CPR = 4 * KJ;
    constexpr int NX = (kRows * CPR + kThreads - 1) / kThreads;
    f4 xr[NX];
    int64_t *rid = reinterpret_cast<int64_t *>(lds + ly.RID), *ridc = reinterpret_cast<int64_t *>(lds + ly.RIDC);
    auto fetch_tile = [&](int64_t tile_, int par) {  // data of tile_ whose row ids are in RID buffer `par`
        const int tq = opaque_v(tid);                // (chunk coordinates recomputed per call, not kept across the tile)
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int q = tq + kThreads * u, r = (q / CPR) & (kRows - 1), c = q % CPR;
            const int64_t i = tile_ * OWN + r;
            const bool ok = q < kRows * CPR && i < g.Mr && 4 * c < K1;
            const int64_t row = rid[par * kRows + r];
            if constexpr (VEC) {
                const int cc = 4 * c < K1 ? 4 * c : K1 - 4;
                const float4 v = *reinterpret_cast<const float4 *>(g.obs + row * K1 + cc);
                xr[u] = ok ? f4{v.x, v.y, v.z, v.w} : f4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = 4 * c + e;
                    const float v = g.obs[row * K1 + (k < K1 ? k : K1 - 1)];
                    xr[u][e] = (ok && k < K1) ? v : 0.f;
                }
            }
        }
    };
    auto commit_tile = [&]() {
        const int tq = opaque_v(tid);
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int q = tq + kThreads * u, r = (q / CPR) & (kRows - 1), c = q % CPR;
            if (q < kRows * CPR) *reinterpret_cast<f4 *>(lds + ly.X + xs_off(r, c, ldx)) = xr[u];
        }
    };

    // ---- persistent gradient accumulators (everything but dW1: kernel B) ----
    f4 gW2[8], gW3;
#pragma unroll
    for (int i = 0; i < 8; ++i) gW2[i] = f4{0.f, 0.f, 0.f, 0.f};
    gW3 = f4{0.f, 0.f, 0.f, 0.f};
    float gW3s = 0.f;          // LOSS 0 (one output): dW3[col] over this lane's rows (folded over kq at the end)
    // db2 / db3 = column sums of dH2 / dQ over the tile's rows as per-thread PARTIAL sums, folded after the last tile (a serial
    // loop of two waves / 16 threads over the 32 rows held the other waves at the next barrier): thread (c = tid & 127,
    // q = tid >> 7) sums rows [8 q, 8 q + 8) of dH2 column c; thread (c = tid & 15, q = tid >> 4 < 16) rows 2 q, 2 q + 1 of dQ column c
    float gB2 = 0.f, gB3 = 0.f;
    float gB1 = 0.f;           // db1[col], this lane's rows (folded over kq at the end)
    double t_a = 0.0, t_b = 0.0;  // loss statistics (LOSS 0: t_b = sum vf; LOSS 1: t_a = sum adv, t_b = sum sq)

    if (tid < kRows) rid[tid] = id0;
    __syncthreads();
    fetch_tile(tile, 0);      // the first tile's rows are in flight behind the weights ...
    // ---- resident weights: W2, W3 (rows >= n_out zero), biases ----
    w2_store(lds + ly.W2, w2q);
#pragma unroll
    for (int u = 0; u < kN3; ++u) {
        const int e = tid + u * kThreads;
        if (e < 16 * kLdh) lds[ly.W3 + e] = w3q[u];
    }
    if (tid < kH) { lds[ly.B1 + tid] = bq1; lds[ly.B2 + tid] = bq2; }
    if (tid < 16) lds[ly.B3 + tid] = bq3;
    commit_tile();
    __syncthreads();          // every thread has read RID
    if (tid < kRows) rid[tid] = id1;
    __syncthreads();
    // (LOSS 0, uniform: a workgroup with one tile -- the PPO minibatch -- skips the next-tile traffic.  LOSS 1 keeps the
    // unconditional form: its workgroups always have many tiles, and behind a branch the loads are no longer scheduled into the
    // MFMA phase in front of them: +1.1 us per tile, measured)
    if (LOSS != 0 || tile + gs < n_tiles) fetch_tile(tile + gs, 0);
    __syncthreads();          // (RID buffer 0 is rewritten at the top of the loop)

    int it = 0;
    for (; tile < n_tiles; tile += gs, ++it) {
        TSTAMP(0);
        if (tid < kRows) {    // published by barrier (A)
            if constexpr (LOSS == 0) {
                ridc[tid] = id0;
                rid[(it & 1) * kRows + tid] = id2;
                id0 = id1; id1 = id2; id2 = id3;
            } else {
                ridc[tid] = row_of<LOSS>(g, tile * OWN + tid);
                rid[(it & 1) * kRows + tid] = row_of<LOSS>(g, (tile + 2 * gs) * OWN + tid);
            }
        }
        // ---- P1: H1 = relu(X W1^T + b1); W1 from registers, X by ds_read_b128 ----
        {
            f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
            const float *xa[4] = {lds + ly.X + xa_base(c16, kq, ldx, 0), lds + ly.X + xa_base(c16, kq, ldx, 1),
                                  lds + ly.X + xa_base(c16, kq, ldx, 2), lds + ly.X + xa_base(c16, kq, ldx, 3)};
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                const f4 a0 = *reinterpret_cast<const f4 *>(xa[j & 3] + 64 * (j >> 2));
                const f4 a1 = *reinterpret_cast<const f4 *>(xa[j & 3] + 64 * (j >> 2) + 16 * ldx);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[0] = mfma4(a0[i], w1f[j][i], acc[0]);
                    acc[1] = mfma4(a1[i], w1f[j][i], acc[1]);
                }
            }
            const float bb = lds[ly.B1 + col];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = mt * 16 + kq * 4 + r;
                    const float h = fmaxf(acc[mt][r] + bb, 0.f);
                    lds[ly.H1 + row * kLdh + col] = h;
                    if constexpr (pub) {   // (uniform) kernel B's B operand of dW2: row-major [Mr][128], 64-B segments per row and wave
                        const int64_t i = tile * OWN + row;
                        if (row < OWN && i < g.Mr) g.h1_out[i * kH + col] = h;
                    }
                }
        }
        __syncthreads();  // (A) H1 complete; every wave is done with X; RID / RIDC published
        TSTAMP(1);
        if (LOSS != 0 || tile + gs < n_tiles) commit_tile();   // the next tile's rows: requested at the end of the previous tile, they flew during L1
        if constexpr (LOSS == 0) {
            if (tid < kRows) id3 = row_of<LOSS>(g, (tile + 4 * gs) * OWN + tid);  // (three tiles ahead)
        }
        // loss inputs of this tile: row id -> scalars are dependent global loads; they fly during layers 2 and 3
        float pf_a = 0.f, pf_b = 0.f;
        if constexpr (LOSS == 0) {
            if (tid < kRows * N) {
                const int r_ = tid / N;
                if (tile * OWN + r_ < g.Mr) {
                    const int64_t sidx = ridc[r_] * N + (tid - r_ * N);
                    pf_a = g.returns[sidx];
                    if (g.value_clip) pf_b = g.v_s_old[sidx];
                }
            }
        } else {
            if (tid < OWN && tile * OWN + tid < g.Mr) {
                const int64_t sidx = ridc[tid] * g.sc_stride + g.sc_off;
                pf_a = g.rew[sidx];
                pf_b = g.term[sidx] ? 0.f : 1.f;
            }
        }
        // ---- P2: H2 = relu(H1 W2^T + b2) ----
        // LOSS 0 (n_out == 1): layer 3 is a 128-long dot product per row -- folded into this epilogue exactly as
        // csrc/critic_rows.hip does it (the same additions in the same order: V here is bit-identical to tsm_critic_rows_forward's),
        // H2 stays in REGISTERS across the loss head and only dH2 goes to LDS: no layer-3 MFMA phases, three barriers less
        float h2r[2][4];
        {
            f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
            const float *pa = lds + ly.H1 + c16 * kLdh + kq;
            const float *pb = lds + ly.W2 + col * kLdh + kq;
#pragma unroll 8
            for (int k0 = 0; k0 < kH; k0 += 4) {
                const float bv = pb[k0];
                acc[0] = mfma4(pa[k0], bv, acc[0]);
                acc[1] = mfma4(pa[16 * kLdh + k0], bv, acc[1]);
            }
            const float bb = lds[ly.B2 + col];
            if constexpr (LOSS == 0) {
                const float w3c = lds[ly.W3 + col];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        h2r[mt][r] = fmaxf(acc[mt][r] + bb, 0.f);
                        const float sv = row16_sum(h2r[mt][r] * w3c);  // over this wave's 16 columns
                        if (c16 == 0) lds[ly.PV + w * kRows + mt * 16 + kq * 4 + r] = sv;
                    }
            } else {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) lds[ly.H2 + (mt * 16 + kq * 4 + r) * kLdh + col] = fmaxf(acc[mt][r] + bb, 0.f);
            }
        }
        __syncthreads();
        TSTAMP(2);
        // ---- P3 (LOSS 1): Q = H2 W3^T + b3 (n_out padded to 16): waves 0 / 1 take the two row halves ----
        if constexpr (LOSS != 0) {
            if (w < 2) {
                f4 acc = f4{0.f, 0.f, 0.f, 0.f};
                const float *a = lds + ly.H2 + (16 * w + c16) * kLdh + kq;
                const float *b = lds + ly.W3 + c16 * kLdh + kq;
#pragma unroll 8
                for (int k0 = 0; k0 < kH; k0 += 4) acc = mfma4(a[k0], b[k0], acc);
                const float bb = lds[ly.B3 + c16];
#pragma unroll
                for (int r = 0; r < 4; ++r) lds[ly.Q + (16 * w + kq * 4 + r) * kLdo + c16] = acc[r] + bb;
            }
            __syncthreads();
        }
        TSTAMP(3);
        // ---- P4: loss head: Q -> d loss / d Q in place ----
        if constexpr (LOSS == 0) {
            // value term for the N agents of every row (ppo.py:198-208) against the row's one V = Q[row][0]
            if (tid < kRows * N) {
                const int r = tid / N, a = tid - r * N;
                const int64_t i = tile * OWN + r;
                float dv = 0.f;
                if (i < g.Mr) {
                    float v = lds[ly.PV + r];
#pragma unroll
                    for (int ww = 1; ww < 8; ++ww) v += lds[ly.PV + ww * kRows + r];
                    v += lds[ly.B3];
                    const float ret = pf_a;
                    float vf, g_v;
                    if (g.value_clip) {
                        const float vs = pf_b;
                        const float d = v - vs;
                        const float dc = fminf(fmaxf(d, -g.eps_clip), g.eps_clip);
                        const bool v_in = d >= -g.eps_clip && d <= g.eps_clip;
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
                    dv = g.vf_coef * g_v * (1.0f / (float)(g.Mr * N));
                    t_b += vf;
                }
                lds[ly.DV + r * 16 + a] = dv;
            }
            __syncthreads();
            if (tid < kRows) {  // d loss / d value of the row = sum over its agents, in agent order
                float sacc = 0.f;
                for (int a = 0; a < N; ++a) sacc += lds[ly.DV + tid * 16 + a];
                lds[ly.Q + tid * kLdo] = sacc;
                gB3 += sacc;                     // db3 = sum of the rows' d loss / d value (folded over the 32 threads at the end)
            }
        } else {
            // values = Q.mean(1) (ctde.py:154-157: sequential f32 sum / n); td target from the NEXT row's value
            if (tid < kRows) {
                float s = 0.f;
                for (int j = 0; j < n_out; ++j) s += lds[ly.Q + tid * kLdo + j];
                lds[ly.V + tid] = n_out == 1 ? s : s / (float)n_out;
            }
            __syncthreads();
            if (tid < kRows) {
                const int64_t i = tile * OWN + tid;
                float dq = 0.f;
                if (tid < OWN && i < g.Mr) {
                    const float v = lds[ly.V + tid];
                    const bool last = (i + 1) % g.tm_T == 0;   // last row of its env's block
                    const float vn = (g.use_full && *g.use_full) ? g.v_full[i]
                                     : (last ? g.v_last[i / g.tm_T] : lds[ly.V + tid + 1]);
                    const float td = pf_a + g.gamma * vn * pf_b;
                    const float diff = v - td;
                    dq = 2.f * diff / ((float)g.Mr * (float)n_out);
                    t_a += (double)(td - v);
                    t_b += (double)diff * (double)diff;
                }
                for (int j = 0; j < 16; ++j) lds[ly.Q + tid * kLdo + j] = j < n_out ? dq : 0.f;
            }
        }
        __syncthreads();
        TSTAMP(4);
        // ---- P5: dW3 += dQ^T H2 ; db3 ; dH2 = (dQ W3) * relu'(H2) ----
        if constexpr (LOSS == 0) {
            // one output: dH2[row][col] = dq[row] w3[col] where H2 > 0 and dW3[col] += dq[row] H2[row][col], from the
            // registers that still hold this lane's eight H2 values
            const float w3c = lds[ly.W3 + col];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = mt * 16 + kq * 4 + r;
                    const float dq = lds[ly.Q + row * kLdo];
                    gW3s += dq * h2r[mt][r];
                    const float d2v = h2r[mt][r] > 0.f ? dq * w3c : 0.f;
                    lds[ly.H2 + row * kLdh + col] = d2v;
                    if constexpr (pub) {   // kernel B's A operand of dW2
                        const int64_t i = tile * OWN + row;
                        if (row < OWN && i < g.Mr) g.dh2_out[i * kH + col] = d2v;
                    }
                }
        } else {
            {
                const float *a = lds + ly.Q + kq * kLdo + c16;             // A[i = out][k = row]
                const float *b = lds + ly.H2 + kq * kLdh + col;            // B[k = row][j = hidden col]
#pragma unroll
                for (int r0 = 0; r0 < kRows; r0 += 4) gW3 = mfma4(a[r0 * kLdo], b[r0 * kLdh], gW3);
            }
            if (tid < 256) {
                const float *q = lds + ly.Q + 2 * (tid >> 4) * kLdo + (tid & 15);
                gB3 += q[0] + q[kLdo];
            }
            f4 d2[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
            {
                const float *a = lds + ly.Q + c16 * kLdo + kq;             // A[i = row][k = out]
                const float *b = lds + ly.W3 + kq * kLdh + col;            // B[k = out][j = hidden col]
#pragma unroll
                for (int k0 = 0; k0 < 16; k0 += 4) {
                    const float bv = b[k0 * kLdh];
                    d2[0] = mfma4(a[k0], bv, d2[0]);
                    d2[1] = mfma4(a[16 * kLdo + k0], bv, d2[1]);
                }
            }
            __syncthreads();  // every wave has read H2 for dW3
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float *p = lds + ly.H2 + (mt * 16 + kq * 4 + r) * kLdh + col;
                    *p = *p > 0.f ? d2[mt][r] : 0.f;
                }
        }
        __syncthreads();
        TSTAMP(5);
        // ---- P6: dW2 += dH2^T H1 ; db2 ; dH1 = (dH2 W2) * relu'(H1) -> global (kernel B), db1 ----
        if constexpr (!pub) {   // (published H1 / dH2: dW2 is formed by kernel B over row chunks instead of as a rank-32 slab per tile)
            const float *a = lds + ly.H2 + kq * kLdh + col;            // A[i = out o][k = row]
            const float *b = lds + ly.H1 + kq * kLdh + c16;            // B[k = row][j = in col]
#pragma unroll 2
            for (int r0 = 0; r0 < kRows; r0 += 4) {
                const float av = a[r0 * kLdh];
#pragma unroll
                for (int ti = 0; ti < 8; ++ti) gW2[ti] = mfma4(av, b[r0 * kLdh + 16 * ti], gW2[ti]);
            }
            // (round 5) the last tile has added to dW2: its 64 KB of the slab go out now, under the dH1 products, instead of behind
            // the loop where all 256 workgroups store at once (csrc/actor_rows64.hip: the same move was worth 5 us of a 103 us launch)
            if (tile + gs >= n_tiles) {
                int lo = (16 * w + kq * 4) * kH + c16;
                asm volatile("" : "+v"(lo));   // (computed here, not hoisted out of the loop: registers)
                float *sl_ = g.slabs + (size_t)blockIdx.x * (size_t)(oB3 + n_out - oB1) + (oW2 - oB1) + lo;
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int ti = 0; ti < 8; ++ti) __builtin_nontemporal_store(gW2[ti][r], sl_ + r * kH + 16 * ti);
            }
        }
        {
            const float *q = lds + ly.H2 + 8 * (tid >> 7) * kLdh + (tid & 127);
            gB2 += ((q[0] + q[kLdh]) + (q[2 * kLdh] + q[3 * kLdh])) + ((q[4 * kLdh] + q[5 * kLdh]) + (q[6 * kLdh] + q[7 * kLdh]));
        }
        {
            f4 d1[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
            const float *a = lds + ly.H2 + c16 * kLdh + kq;            // A[i = row][k = o]
            const float *b = lds + ly.W2 + kq * kLdh + col;            // B[k = o][j = in col]
#pragma unroll 8
            for (int k0 = 0; k0 < kH; k0 += 4) {
                const float bv = b[k0 * kLdh];
                d1[0] = mfma4(a[k0], bv, d1[0]);
                d1[1] = mfma4(a[16 * kLdh + k0], bv, d1[1]);
            }
            // lane (c16, kq) holds dH1[row = mt 16 + kq 4 + r][col]: masked by relu'(H1), straight to global memory
            // (row-major [Mr][128]: 16 consecutive columns per row and wave = 64-B segments), and into db1
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = mt * 16 + kq * 4 + r;
                    const float v = lds[ly.H1 + row * kLdh + col] > 0.f ? d1[mt][r] : 0.f;
                    gB1 += v;
                    const int64_t i = tile * OWN + row;
                    if (row < OWN && i < g.Mr) g.dh1[i * kH + col] = v;
                }
        }
        __syncthreads();  // H1 / H2 / Q are free for the next tile
        // the rows of the tile after the next: in flight during the next tile's layer 1, committed behind its barrier (A)
        // (requested HERE and not a phase earlier: 24 registers less across the backward phases)
        // (RID buffer it & 1: written at the top of this iteration, published by its barrier (A); the next iteration writes
        // the other buffer, so these reads need no barrier behind them)
        if (LOSS != 0 || tile + 2 * gs < n_tiles) fetch_tile(tile + 2 * gs, it & 1);
        TSTAMP(6);
    }

    // ---- the workgroup's slab of b1 | W2 | b2 | W3 | b3: written once, streamed ----
    const int nrest = oB3 + n_out - oB1;
    float *slab = g.slabs + (size_t)blockIdx.x * (size_t)nrest;
    const int sW2 = oW2 - oB1, sB2 = oB2 - oB1, sW3 = oW3 - oB1, sB3 = oB3 - oB1;  // offsets inside the slab (b1 first)
    gB1 += __shfl_xor(gB1, 16, 64);   // fold the four row groups (kq) of a column
    gB1 += __shfl_xor(gB1, 32, 64);
    if (kq == 0) __builtin_nontemporal_store(gB1, slab + col);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = 16 * w + kq * 4 + r;
        if constexpr (!pub) {
            if ((int64_t)blockIdx.x >= n_tiles) {   // (a workgroup with tiles stored dW2 behind its last tile's products; one without: zeros)
#pragma unroll
                for (int ti = 0; ti < 8; ++ti) __builtin_nontemporal_store(gW2[ti][r], slab + sW2 + o * kH + 16 * ti + c16);
            }
        }
        if constexpr (LOSS != 0) {
            const int a = kq * 4 + r;
            if (a < n_out) __builtin_nontemporal_store(gW3[r], slab + sW3 + a * kH + 16 * w + c16);
        }
    }
    if constexpr (LOSS == 0) {
        gW3s += __shfl_xor(gW3s, 16, 64);   // fold the four row groups (kq) of a column, as db1
        gW3s += __shfl_xor(gW3s, 32, 64);
        if (kq == 0) __builtin_nontemporal_store(gW3s, slab + sW3 + col);
    }
    {   // fold the bias partials (fixed order) through the idle H1 region
        float *sc = lds + ly.H1;   // [4][128] db2 | [16][16] db3
        sc[tid] = gB2;
        if (tid < 256) sc[512 + tid] = gB3;
        __syncthreads();
        if (tid < 128) __builtin_nontemporal_store((sc[tid] + sc[128 + tid]) + (sc[256 + tid] + sc[384 + tid]), slab + sB2 + tid);
        else if (tid >= 256 && tid < 256 + n_out) {
            float t = 0.f;
            if constexpr (LOSS == 0) {   // (one output: threads 0..31 hold the rows' partial sums)
                for (int k = 0; k < kRows; ++k) t += sc[512 + k];
            } else {
                for (int k = 0; k < 16; ++k) t += sc[512 + 16 * k + (tid - 256)];
            }
            __builtin_nontemporal_store(t, slab + sB3 + tid - 256);
        }
    }
    {   // loss statistics: wave sums, then the waves in order
        double *red = reinterpret_cast<double *>(lds + ly.RED);
        const double a = wave_sum(t_a), b = wave_sum(t_b);
        __syncthreads();
        if (lane == 0) { red[w] = a; red[8 + w] = b; }
        __syncthreads();
        if (tid == 0) {
            double aa = 0.0, bb = 0.0;
            for (int k = 0; k < kThreads / 64; ++k) { aa += red[k]; bb += red[8 + k]; }
            g.partial[4 * blockIdx.x + 0] = LOSS == 0 ? 0.0 : aa;
            g.partial[4 * blockIdx.x + 1] = bb;
            g.partial[4 * blockIdx.x + 2] = 0.0;
            g.partial[4 * blockIdx.x + 3] = 0.0;
            if (g.stamps && blockIdx.x == 0) g.stamps[299] = (long long)wall_clock64();
        }
    }
}

template <int KJ, bool VEC, int LOSS, bool PUB = false>
int launch_train_v(const TrainArgs &g, int grid, hipStream_t st) {
    const TrainLay ly(KJ);
    const size_t shmem = (size_t)ly.total * sizeof(float);
    TSM_REQUIRE(shmem <= kMaxLds, "critic gradient step: LDS layout of %zu bytes does not fit", shmem);
    static bool attr_set = false;  // (set before any capture: tsm_critic_rows_init)
    if (!attr_set) {
        TSM_HIP(tsm_allow_max_lds(reinterpret_cast<const void *>(critic_rows_train_kernel<KJ, VEC, LOSS, PUB>)));
        attr_set = true;
    }
    if (grid > 0) {
        hipLaunchKernelGGL((critic_rows_train_kernel<KJ, VEC, LOSS, PUB>), dim3((unsigned)grid), dim3(kThreads), shmem, st, g);
        TSM_LAUNCH_CHECK();
    }
    return TSM_OK;
}

template <int KJ>
int launch_train(const TrainArgs &g, int loss, int grid, hipStream_t st) {
    if (g.K1 == 0) {  // tsm_critic_rows_init: the attributes of every form
        int rc = launch_train_v<KJ, true, 0>(g, 0, st);
        if (rc == TSM_OK) rc = launch_train_v<KJ, true, 1>(g, 0, st);
        if (rc == TSM_OK) rc = launch_train_v<KJ, true, 0, true>(g, 0, st);
        if constexpr (KJ <= 4) {
            if (rc == TSM_OK) rc = launch_train_v<KJ, false, 0>(g, 0, st);
            if (rc == TSM_OK) rc = launch_train_v<KJ, false, 1>(g, 0, st);
            if (rc == TSM_OK) rc = launch_train_v<KJ, false, 0, true>(g, 0, st);
        }
        return rc;
    }
    const bool vec = (g.K1 & 3) == 0;
    if (g.h1_out) TSM_REQUIRE(loss == 0, "critic gradient step: H1 / dH2 are published by the PPO form only");
    if constexpr (KJ <= 4) {
        if (!vec && g.h1_out) return launch_train_v<KJ, false, 0, true>(g, grid, st);
        if (!vec) return loss == 0 ? launch_train_v<KJ, false, 0>(g, grid, st) : launch_train_v<KJ, false, 1>(g, grid, st);
    } else {
        TSM_REQUIRE(vec, "critic gradient step: input widths above 64 must be multiples of 4 (got %d)", g.K1);
    }
    if (g.h1_out) return launch_train_v<KJ, true, 0, true>(g, grid, st);
    return loss == 0 ? launch_train_v<KJ, true, 0>(g, grid, st) : launch_train_v<KJ, true, 1>(g, grid, st);
}

int dispatch_train(int kj, const TrainArgs &g, int loss, int grid, hipStream_t st) {
    switch (kj) {
        case 1: return launch_train<1>(g, loss, grid, st);
        case 2: return launch_train<2>(g, loss, grid, st);
        case 3: return launch_train<3>(g, loss, grid, st);
        case 4: return launch_train<4>(g, loss, grid, st);
        case 6: return launch_train<6>(g, loss, grid, st);
        case 8: return launch_train<8>(g, loss, grid, st);
        case 12: return launch_train<12>(g, loss, grid, st);
        case 16: return launch_train<16>(g, loss, grid, st);
        case 24: return launch_train<24>(g, loss, grid, st);
        default: break;
    }
    tsm_set_error("critic gradient step: unsupported input width");
    return TSM_ERR_INVALID;
}

}  // namespace

// one-time function attributes of the gradient-step instantiations serving in_dim (called by tsm_critic_rows_init)
int tsm_critic_dw1_init();  // critic_dw1.hip
int tsm_critic_train_init(int32_t in_dim) {
    TrainArgs g{};
    const int rc = dispatch_train(pick_kj(in_dim), g, 0, 0, nullptr);
    return rc != TSM_OK ? rc : tsm_critic_dw1_init();
}

TSM_EXPORT int64_t tsm_critic_rows_param_count(int32_t in_dim, int32_t hidden, int32_t n_out) {
    if (hidden != kH || in_dim < 1 || n_out < 1 || n_out > 16) return -1;
    return (int64_t)hidden * in_dim + hidden + (int64_t)hidden * hidden + hidden + (int64_t)n_out * hidden + n_out;
}

// workgroups of kernel A (= slabs of the b1 | W2 | b2 | W3 | b3 part) for Mr rows; td: tiles own 31 rows
TSM_EXPORT int tsm_critic_rows_grad_grid(int64_t Mr, int32_t td) {
    if (Mr <= 0) return 0;
    const int64_t tiles = ceil_div(Mr, td ? kRows - 1 : kRows);
    const int cu = n_cu_dev();
    return (int)(tiles < cu ? tiles : cu);
}

// ---- W1 in fragment order (see TrainArgs::w1_img) ----
namespace {
__global__ void w1_image_kernel(const float *__restrict__ w1, int K1, int KJ, float *__restrict__ img) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // image element [w][j][lane][i]
    if (e >= (int64_t)8 * KJ * 64 * 4) return;
    const int i = (int)(e & 3), lane = (int)((e >> 2) & 63), wj = (int)(e >> 8), j = wj % KJ, w = wj / KJ;
    const int o = 16 * w + (lane & 15), k = 16 * j + 4 * (lane >> 4) + i;
    img[e] = k < K1 ? w1[(size_t)o * K1 + k] : 0.f;
}
}  // namespace

TSM_EXPORT int tsm_critic_rows_w1_image_kj(int32_t in_dim) { return pick_kj(in_dim); }

TSM_EXPORT int64_t tsm_critic_rows_w1_image_elems(int32_t in_dim) {
    const int kj = pick_kj(in_dim);
    return kj ? (int64_t)kH * 16 * kj : -1;
}

TSM_EXPORT int tsm_critic_rows_w1_image(const float *w1, int32_t in_dim, float *image_out, void *stream) {
    const int kj = pick_kj(in_dim);
    TSM_REQUIRE(kj != 0 && w1 && image_out, "tsm_critic_rows_w1_image: in_dim = %d unsupported or null pointer", in_dim);
    const int64_t n = (int64_t)kH * 16 * kj;
    hipLaunchKernelGGL(w1_image_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, tsm_stream(stream), w1, in_dim, kj, image_out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int tsm_critic_rows_grad_ppo(const float *critic_params, const float *w1_image, int32_t in_dim, int32_t hidden, int32_t n_agent,
                                        const float *obs_rows, const float *returns, const float *v_s_old,
                                        const int64_t *rows, int64_t first_row, int64_t Mr, const tsm_ppo_cfg *cfg,
                                        int32_t n_blocks, float *dh1_out, float *h1_out, float *dh2_out, float *rest_slabs_out,
                                        double *loss_partial_out, void *stream) {
    TSM_REQUIRE(hidden == kH && pick_kj(in_dim) != 0 && ((in_dim & 3) == 0 || in_dim <= 64) && n_agent >= 1 && n_agent <= 16,
                "tsm_critic_rows_grad_ppo supports hidden == 128, in_dim <= 384 (a multiple of 4 above 64), n_agent <= 16 "
                "(got %d / %d / %d)", hidden, in_dim, n_agent);
    TSM_REQUIRE(Mr >= 1 && cfg, "tsm_critic_rows_grad_ppo: empty minibatch or null cfg");
    TSM_REQUIRE(critic_params && obs_rows && returns && dh1_out && rest_slabs_out && loss_partial_out,
                "tsm_critic_rows_grad_ppo: null pointer");
    TSM_REQUIRE(!cfg->value_clip || v_s_old, "tsm_critic_rows_grad_ppo: value_clip needs v_s_old");
    TSM_REQUIRE(n_blocks >= 1 && n_blocks <= ceil_div(Mr, kRows), "tsm_critic_rows_grad_ppo: n_blocks = %d out of range", n_blocks);
    TrainArgs g{};
    g.P = critic_params; g.w1_img = w1_image; g.obs = obs_rows; g.rows = rows; g.first_row = first_row; g.Mr = Mr; g.K1 = in_dim; g.n_out = 1;
    TSM_REQUIRE((h1_out == nullptr) == (dh2_out == nullptr), "tsm_critic_rows_grad_ppo: h1_out and dh2_out come as a pair");
    g.h1_out = h1_out; g.dh2_out = dh2_out;
    g.returns = returns; g.v_s_old = v_s_old; g.N = n_agent;
    g.eps_clip = (float)cfg->eps_clip; g.vf_coef = (float)cfg->vf_coef; g.value_clip = cfg->value_clip;
    g.dh1 = dh1_out; g.slabs = rest_slabs_out; g.partial = loss_partial_out; g.stamps = g_tsm_stamps;
    return dispatch_train(pick_kj(in_dim), g, 0, n_blocks, tsm_stream(stream));
}

TSM_EXPORT int tsm_critic_rows_grad_td(const float *critic_params, const float *w1_image, int32_t in_dim, int32_t hidden, int32_t n_out,
                                       const float *joint_rows, int64_t T, int64_t E, const float *rew,
                                       const uint8_t *terminated, int64_t scalar_stride, int64_t scalar_offset,
                                       const float *v_last, const float *v_next_full, const int32_t *use_full,
                                       double gamma, int32_t n_blocks, float *dh1_out, float *rest_slabs_out,
                                       double *loss_partial_out, void *stream) {
    TSM_REQUIRE(hidden == kH && pick_kj(in_dim) != 0 && ((in_dim & 3) == 0 || in_dim <= 64) && n_out >= 1 && n_out <= 16,
                "tsm_critic_rows_grad_td supports hidden == 128, in_dim <= 384 (a multiple of 4 above 64), n_out <= 16 "
                "(got %d / %d / %d)", hidden, in_dim, n_out);
    TSM_REQUIRE(T >= 1 && E >= 1 && scalar_stride >= 1 && scalar_offset >= 0 && scalar_offset < scalar_stride,
                "tsm_critic_rows_grad_td: bad sizes (T %lld, E %lld)", (long long)T, (long long)E);
    TSM_REQUIRE(critic_params && joint_rows && rew && terminated && v_last && dh1_out && rest_slabs_out && loss_partial_out,
                "tsm_critic_rows_grad_td: null pointer");
    TSM_REQUIRE(!use_full || v_next_full, "tsm_critic_rows_grad_td: use_full needs v_next_full");
    const int64_t B = T * E;
    TSM_REQUIRE(n_blocks >= 1 && n_blocks <= ceil_div(B, kRows - 1), "tsm_critic_rows_grad_td: n_blocks = %d out of range", n_blocks);
    TrainArgs g{};
    g.P = critic_params; g.w1_img = w1_image; g.obs = joint_rows; g.rows = nullptr; g.first_row = 0; g.Mr = B; g.tm_T = T; g.tm_E = E;
    g.K1 = in_dim; g.n_out = n_out; g.N = 1;
    g.rew = rew; g.term = terminated; g.sc_stride = scalar_stride; g.sc_off = scalar_offset; g.v_last = v_last;
    g.v_full = v_next_full; g.use_full = use_full;
    g.gamma = (float)gamma;
    g.dh1 = dh1_out; g.slabs = rest_slabs_out; g.partial = loss_partial_out; g.stamps = g_tsm_stamps;
    return dispatch_train(pick_kj(in_dim), g, 1, n_blocks, tsm_stream(stream));
}
