// critic_train.hip -- one gradient step of a 128-wide critic (in -> 128 -> 128 -> n_out) in ONE launch: forward, loss,
// the whole backward pass, for the two losses the hot path has:
//   LOSS 0  the value term of the PPO loss on joint rows -- one V(row) against the returns of the row's N agents
//           (/root/reference/tianshou/algorithm/modelfree/ppo.py:198-208; tsm_ppo_critic_rows_update)
//   LOSS 1  the TD loss of CTDEPolicy.learn: values = critic(global_obs).mean(1), td_target = rew + gamma values_next
//           (1 - terminated), critic_loss = mse(values, td_target)   (tianshou/algorithm/multiagent/ctde.py:149-172;
//           tsm_ctde_critic_rows_update), on rows whose successor IS the next row of the same env (chained rows).
//
// gfx950 mapping (the second generation of csrc/ppo_rows.hip's critic kernel; what changed and why):
//   * The first layer's weights never pass through LDS.  Wave w owns output columns [16 w, 16 w + 16) and loads its
//     B-operand fragments W1[col][16 j + 4 kq .. + 3] straight from global memory (L2-resident: every workgroup reads the
//     same 192 KB) in double-buffered batches of 4 k-groups while the previous batch is multiplied: layer 1 has no
//     staging copies and no barriers (the old kernel: 12 K-slices x (global -> registers -> LDS -> barrier)).
//   * The tile's observation block (32 rows x K1, 48 KB at K1 = 384) stays in LDS for the whole tile, 16-B chunks
//     XOR-swizzled by the row (critic_rows_dev.h: conflict-free ds_read_b128 as the A operand of layer 1): the layer-1
//     weight gradient dW1 += dH1^T X reads it again from LDS instead of streaming the rows a second time from L2.
//   * dW1 (K1 / 16 accumulator tiles per wave), dW2 (8) and dW3 (1) live in registers across all tiles of the workgroup
//     and are written once, as the workgroup's gradient slab (deterministic: static tile assignment, slabs summed in
//     order by tsm_adam_step_segs).
//   * Layer 3 (n_out <= 16 outputs, padded to one 16-column MFMA tile) and its backward are MFMA like the actor's logits.
//   * LOSS 1 needs V of a row's SUCCESSOR: a tile owns 31 consecutive env-major rows and computes the 32nd as a halo, so
//     the target of every owned row comes out of the same forward pass (3 % redundant rows instead of a second pass over
//     all of them); the last row of an env's block takes V(obs_next) from `v_last` (a small forward pass of its own,
//     tsm_critic_rows_forward).
#include "critic_rows_dev.h"

extern long long *g_tsm_stamps;  // abi.hip (diagnostics, tools/stamp_critic_train.py)

namespace {

constexpr int kLdo = 18;

struct TrainLay {  // LDS layout in floats
    int ldx, W2, W3, X, H1, H2, B1, B2, B3, Q, V, DV, RID, RED, total;
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
        V = o; o += 2 * kRows;          // row values / loss scratch
        DV = o; o += kRows * 16;        // LOSS 0: d loss / d value of a row's agents (N <= 16)
        RID = o; o += 2 * kRows;        // row ids of the tile (int64)
        RED = o; o += 32;               // final reduction scratch (doubles)
        total = o;
    }
    __device__ void launder() {         // (critic_rows_dev.h: opaque_s)
        W2 = opaque_s(W2); W3 = opaque_s(W3); X = opaque_s(X); H1 = opaque_s(H1); H2 = opaque_s(H2); B1 = opaque_s(B1);
        B2 = opaque_s(B2); B3 = opaque_s(B3); Q = opaque_s(Q); V = opaque_s(V); DV = opaque_s(DV); RID = opaque_s(RID);
        RED = opaque_s(RED);
    }
};

struct TrainArgs {
    const float *P;          // critic parameters: w0[H][K1] b0[H] w1[H][H] b1[H] w2[n_out][H] b2[n_out]
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
    float gamma;
    float *slabs;            // [grid][P]
    double *partial;         // [grid][4]: LOSS 0 {0, sum vf, 0, 0}; LOSS 1 {sum (td - v), sum (v - td)^2, 0, 0}
    long long *stamps;
};

#define TSTAMP(k) do { if (g.stamps && blockIdx.x == 0 && tid == 0 && it == 0) g.stamps[300 + (k)] = (long long)wall_clock64(); } while (0)

template <int KJ, bool VEC, int LOSS>
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
    const float *w1row = g.P + (size_t)col * K1;

    // ---- resident weights: W2, W3 (rows >= n_out zero), biases; X pads zero ----
    stage_w2_rows(lds + ly.W2, g.P + oW2);
    for (int e = tid; e < 16 * kLdh; e += kThreads) {
        const int r = e / kLdh, c = e - r * kLdh;
        lds[ly.W3 + e] = (r < n_out && c < kH) ? g.P[oW3 + r * kH + c] : 0.f;
    }
    if (tid < kH) { lds[ly.B1 + tid] = g.P[oB1 + tid]; lds[ly.B2 + tid] = g.P[oB2 + tid]; }
    if (tid < 16) lds[ly.B3 + tid] = tid < n_out ? g.P[oB3 + tid] : 0.f;
    for (int e = tid; e < kRows * ldx; e += kThreads) lds[ly.X + e] = 0.f;

    // ---- persistent gradient accumulators ----
    f4 gW1[KJ], gW2[8], gW3;
#pragma unroll
    for (int i = 0; i < KJ; ++i) gW1[i] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) gW2[i] = f4{0.f, 0.f, 0.f, 0.f};
    gW3 = f4{0.f, 0.f, 0.f, 0.f};
    float gB = 0.f;            // threads 0..127: db1; 128..255: db2; 256..271: db3
    double t_a = 0.0, t_b = 0.0;  // loss statistics (LOSS 0: t_b = sum vf; LOSS 1: t_a = sum adv, t_b = sum sq)

    int64_t *rid = reinterpret_cast<int64_t *>(lds + ly.RID);
    constexpr int CPR = 4 * KJ;                                       // chunks per (padded) row
    constexpr int NX = (kRows * CPR + kThreads - 1) / kThreads;       // chunks per thread
    constexpr int JB = KJ < 4 ? KJ : (KJ > 16 ? 2 : 4);              // k-groups per W1 batch (registers are scarce at K1 = 384)
    constexpr int NB = (KJ + JB - 1) / JB;
    __syncthreads();

    int it = 0;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, ++it) {
        TSTAMP(0);
        // ---- P0: row ids -> LDS; first W1 batch in flight; observation rows -> swizzled tile ----
        if (tid < kRows) {
            const int64_t i = tile * OWN + tid;
            const int64_t ic = i < g.Mr ? i : g.Mr - 1;
            rid[tid] = g.rows ? g.rows[ic] : (g.tm_T > 0 ? (ic % g.tm_T) * g.tm_E + ic / g.tm_T : g.first_row + ic);
        }
        f4 wcur[JB], wnext[JB];
#pragma unroll
        for (int q = 0; q < JB; ++q) wcur[q] = load_w1_frag<VEC>(w1row, 16 * q + 4 * kq, K1);
        __syncthreads();
        // (passes of two chunks per thread: the gradient accumulators leave few registers for loads in flight)
#pragma unroll 1
        for (int u0 = 0; u0 < NX; u0 += 2) {
            f4 xr[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = tid + kThreads * (u0 + u), r = (q / CPR) & (kRows - 1), c = q % CPR;
                const int64_t i = tile * OWN + r;
                const bool ok = q < kRows * CPR && i < g.Mr && 4 * c < K1;
                const int64_t row = rid[r];
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
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = tid + kThreads * (u0 + u), r = (q / CPR) & (kRows - 1), c = q % CPR;
                if (q < kRows * CPR) *reinterpret_cast<f4 *>(lds + ly.X + xs_off(r, c, ldx)) = xr[u];
            }
        }
        // loss inputs of this tile (dependent on the row ids): issued here, they fly during the forward pass
        float pf_a = 0.f, pf_b = 0.f;
        if constexpr (LOSS == 0) {
            if (tid < kRows * N) {
                const int r_ = tid / N;
                if (tile * OWN + r_ < g.Mr) {
                    const int64_t sidx = rid[r_] * N + (tid - r_ * N);
                    pf_a = g.returns[sidx];
                    if (g.value_clip) pf_b = g.v_s_old[sidx];
                }
            }
        } else {
            if (tid < OWN && tile * OWN + tid < g.Mr) {
                const int64_t sidx = rid[tid] * g.sc_stride + g.sc_off;
                pf_a = g.rew[sidx];
                pf_b = g.term[sidx] ? 0.f : 1.f;
            }
        }
        __syncthreads();
        TSTAMP(1);
        // ---- P1: H1 = relu(X W1^T + b1): W1 fragments streamed from global memory, X by ds_read_b128 ----
        {
            f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
            const float *xa[4] = {lds + ly.X + xa_base(c16, kq, ldx, 0), lds + ly.X + xa_base(c16, kq, ldx, 1),
                                  lds + ly.X + xa_base(c16, kq, ldx, 2), lds + ly.X + xa_base(c16, kq, ldx, 3)};
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (b + 1 < NB) {
#pragma unroll
                    for (int q = 0; q < JB; ++q) {
                        const int j = (b + 1) * JB + q;
                        wnext[q] = j < KJ ? load_w1_frag<VEC>(w1row, 16 * j + 4 * kq, K1) : f4{0.f, 0.f, 0.f, 0.f};
                    }
                }
#pragma unroll
                for (int q = 0; q < JB; ++q) {
                    const int j = b * JB + q;
                    if (j < KJ) {
                        const f4 a0 = *reinterpret_cast<const f4 *>(xa[j & 3] + 64 * (j >> 2));
                        const f4 a1 = *reinterpret_cast<const f4 *>(xa[j & 3] + 64 * (j >> 2) + 16 * ldx);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            acc[0] = mfma4(a0[i], wcur[q][i], acc[0]);
                            acc[1] = mfma4(a1[i], wcur[q][i], acc[1]);
                        }
                    }
                }
                if (b + 1 < NB) {
#pragma unroll
                    for (int q = 0; q < JB; ++q) wcur[q] = wnext[q];
                }
            }
            const float bb = lds[ly.B1 + col];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) lds[ly.H1 + (mt * 16 + kq * 4 + r) * kLdh + col] = fmaxf(acc[mt][r] + bb, 0.f);
        }
        __syncthreads();
        TSTAMP(2);
        // ---- P2: H2 = relu(H1 W2^T + b2) ----
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
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) lds[ly.H2 + (mt * 16 + kq * 4 + r) * kLdh + col] = fmaxf(acc[mt][r] + bb, 0.f);
        }
        __syncthreads();
        TSTAMP(3);
        // ---- P3: Q = H2 W3^T + b3 (n_out padded to 16): waves 0 / 1 take the two row halves ----
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
        TSTAMP(4);
        // ---- P4: loss head: Q -> d loss / d Q in place ----
        if constexpr (LOSS == 0) {
            // value term for the N agents of every row (ppo.py:198-208) against the row's one V = Q[row][0]
            if (tid < kRows * N) {
                const int r = tid / N, a = tid - r * N;
                const int64_t i = tile * OWN + r;
                float dv = 0.f;
                if (i < g.Mr) {
                    const float v = lds[ly.Q + r * kLdo], ret = pf_a;
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
                lds[ly.Q + tid * kLdo] = sacc;   // (columns 1..15 hold exact zeros: W3 / b3 pads)
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
                    const float vn = last ? g.v_last[i / g.tm_T] : lds[ly.V + tid + 1];
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
        TSTAMP(5);
        // ---- P5: dW3 += dQ^T H2 ; db3 ; dH2 = (dQ W3) * relu'(H2) ----
        {
            const float *a = lds + ly.Q + kq * kLdo + c16;             // A[i = out][k = row]
            const float *b = lds + ly.H2 + kq * kLdh + col;            // B[k = row][j = hidden col]
#pragma unroll
            for (int r0 = 0; r0 < kRows; r0 += 4) gW3 = mfma4(a[r0 * kLdo], b[r0 * kLdh], gW3);
        }
        if (tid >= 256 && tid < 272) {
            float s = 0.f;
#pragma unroll 8
            for (int r = 0; r < kRows; ++r) s += lds[ly.Q + r * kLdo + (tid - 256)];
            gB += s;
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
        __syncthreads();
        TSTAMP(6);
        // ---- P6: dW2 += dH2^T H1 ; db2 ; dH1 = (dH2 W2) * relu'(H1) ----
        {
            const float *a = lds + ly.H2 + kq * kLdh + col;            // A[i = out o][k = row]
            const float *b = lds + ly.H1 + kq * kLdh + c16;            // B[k = row][j = in col]
#pragma unroll 2
            for (int r0 = 0; r0 < kRows; r0 += 4) {
                const float av = a[r0 * kLdh];
#pragma unroll
                for (int ti = 0; ti < 8; ++ti) gW2[ti] = mfma4(av, b[r0 * kLdh + 16 * ti], gW2[ti]);
            }
        }
        if (tid >= 128 && tid < 256) {
            float s = 0.f;
#pragma unroll 8
            for (int r = 0; r < kRows; ++r) s += lds[ly.H2 + r * kLdh + (tid - 128)];
            gB += s;
        }
        f4 d1[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
        {
            const float *a = lds + ly.H2 + c16 * kLdh + kq;            // A[i = row][k = o]
            const float *b = lds + ly.W2 + kq * kLdh + col;            // B[k = o][j = in col]
#pragma unroll 8
            for (int k0 = 0; k0 < kH; k0 += 4) {
                const float bv = b[k0 * kLdh];
                d1[0] = mfma4(a[k0], bv, d1[0]);
                d1[1] = mfma4(a[16 * kLdh + k0], bv, d1[1]);
            }
        }
        __syncthreads();  // every wave has read H1 for dW2
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float *p = lds + ly.H1 + (mt * 16 + kq * 4 + r) * kLdh + col;
                *p = *p > 0.f ? d1[mt][r] : 0.f;
            }
        __syncthreads();
        TSTAMP(7);
        // ---- P7: dW1 += dH1^T X ; db1 (the observation tile is still in LDS) ----
        {
            const float *a = lds + ly.H1 + kq * kLdh + col;            // A[i = out o][k = row]
            const float *xb = lds + ly.X + xb_base(c16, kq, ldx);
#pragma unroll 1
            for (int r0 = 0; r0 < kRows; r0 += 4) {
                const float av = a[r0 * kLdh];
#pragma unroll
                for (int ti = 0; ti < KJ; ++ti)                       // B[k = row r0 + kq][j = obs col 16 ti + c16]
                    gW1[ti] = mfma4(av, xb[r0 * ldx + 64 * (ti >> 2) + 16 * ((ti & 3) ^ ((r0 >> 2) & 3))], gW1[ti]);
            }
        }
        if (tid < 128) {
            float s = 0.f;
#pragma unroll 8
            for (int r = 0; r < kRows; ++r) s += lds[ly.H1 + r * kLdh + tid];
            gB += s;
        }
        __syncthreads();  // X / H1 / H2 / Q are free for the next tile
        TSTAMP(8);
    }

    // ---- the workgroup's gradient slab: written once, streamed (consumed once, by the reduction kernel) ----
    float *slab = g.slabs + (size_t)blockIdx.x * (size_t)(oB3 + n_out);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = 16 * w + kq * 4 + r;
#pragma unroll
        for (int ti = 0; ti < KJ; ++ti) {
            const int k = 16 * ti + c16;
            if (k < K1) __builtin_nontemporal_store(gW1[ti][r], slab + (size_t)o * K1 + k);
        }
#pragma unroll
        for (int ti = 0; ti < 8; ++ti) __builtin_nontemporal_store(gW2[ti][r], slab + oW2 + o * kH + 16 * ti + c16);
        const int a = kq * 4 + r;
        if (a < n_out) __builtin_nontemporal_store(gW3[r], slab + oW3 + a * kH + 16 * w + c16);
    }
    if (tid < 128) __builtin_nontemporal_store(gB, slab + oB1 + tid);
    else if (tid < 256) __builtin_nontemporal_store(gB, slab + oB2 + tid - 128);
    else if (tid < 256 + n_out) __builtin_nontemporal_store(gB, slab + oB3 + tid - 256);
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
        }
    }
}

template <int KJ, bool VEC, int LOSS>
int launch_train_v(const TrainArgs &g, int grid, hipStream_t st) {
    const TrainLay ly(KJ);
    const size_t shmem = (size_t)ly.total * sizeof(float);
    TSM_REQUIRE(shmem <= kMaxLds, "critic gradient step: LDS layout of %zu bytes does not fit", shmem);
    static bool attr_set = false;  // (set before any capture: tsm_critic_rows_init)
    if (!attr_set) {
        TSM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(critic_rows_train_kernel<KJ, VEC, LOSS>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
        attr_set = true;
    }
    if (grid > 0) {
        hipLaunchKernelGGL((critic_rows_train_kernel<KJ, VEC, LOSS>), dim3((unsigned)grid), dim3(kThreads), shmem, st, g);
        TSM_LAUNCH_CHECK();
    }
    return TSM_OK;
}

template <int KJ>
int launch_train(const TrainArgs &g, int loss, int grid, hipStream_t st) {
    if (g.K1 == 0) {  // tsm_critic_rows_init: the attributes of every form
        int rc = launch_train_v<KJ, true, 0>(g, 0, st);
        if (rc == TSM_OK) rc = launch_train_v<KJ, true, 1>(g, 0, st);
        if constexpr (KJ <= 4) {
            if (rc == TSM_OK) rc = launch_train_v<KJ, false, 0>(g, 0, st);
            if (rc == TSM_OK) rc = launch_train_v<KJ, false, 1>(g, 0, st);
        }
        return rc;
    }
    const bool vec = (g.K1 & 3) == 0;
    if constexpr (KJ <= 4) {
        if (!vec) return loss == 0 ? launch_train_v<KJ, false, 0>(g, grid, st) : launch_train_v<KJ, false, 1>(g, grid, st);
    } else {
        TSM_REQUIRE(vec, "critic gradient step: input widths above 64 must be multiples of 4 (got %d)", g.K1);
    }
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
int tsm_critic_train_init(int32_t in_dim) {
    TrainArgs g{};
    return dispatch_train(pick_kj(in_dim), g, 0, 0, nullptr);
}

// the PPO value term on the second-generation kernel (called by tsm_ppo_critic_rows_update, csrc/ppo_rows.hip)
int tsm_critic_train_ppo(const float *critic_params, int32_t in_dim, int32_t n_agent, const float *obs_rows,
                         const float *returns, const float *v_s_old, const int64_t *rows, int64_t first_row, int64_t Mr,
                         const tsm_ppo_cfg *cfg, int32_t n_blocks, float *grad_slabs_out, double *loss_partial_out,
                         void *stream) {
    TrainArgs g{};
    g.P = critic_params; g.obs = obs_rows; g.rows = rows; g.first_row = first_row; g.Mr = Mr; g.K1 = in_dim; g.n_out = 1;
    g.returns = returns; g.v_s_old = v_s_old; g.N = n_agent;
    g.eps_clip = (float)cfg->eps_clip; g.vf_coef = (float)cfg->vf_coef; g.value_clip = cfg->value_clip;
    g.slabs = grad_slabs_out; g.partial = loss_partial_out; g.stamps = g_tsm_stamps;
    return dispatch_train(pick_kj(in_dim), g, 0, n_blocks, tsm_stream(stream));
}

TSM_EXPORT int64_t tsm_ctde_critic_rows_param_count(int32_t in_dim, int32_t hidden, int32_t n_out) {
    if (hidden != kH || in_dim < 1 || n_out < 1 || n_out > 16) return -1;
    return (int64_t)hidden * in_dim + hidden + (int64_t)hidden * hidden + hidden + (int64_t)n_out * hidden + n_out;
}

// workgroups (= gradient slabs) for B rows: tiles own 31 rows (the 32nd is the successor halo), one workgroup per CU
TSM_EXPORT int tsm_ctde_critic_rows_grid(int64_t B) {
    if (B <= 0) return 0;
    const int64_t tiles = ceil_div(B, kRows - 1);
    const int cu = n_cu_dev();
    return (int)(tiles < cu ? tiles : cu);
}

TSM_EXPORT int tsm_ctde_critic_rows_update(const float *critic_params, int32_t in_dim, int32_t hidden, int32_t n_out,
                                           const float *joint_rows, int64_t T, int64_t E, const float *rew,
                                           const uint8_t *terminated, int64_t scalar_stride, int64_t scalar_offset,
                                           const float *v_last, double gamma, int32_t n_blocks, float *grad_slabs_out,
                                           double *loss_partial_out, void *stream) {
    TSM_REQUIRE(hidden == kH && pick_kj(in_dim) != 0 && n_out >= 1 && n_out <= 16,
                "tsm_ctde_critic_rows_update supports hidden == 128, in_dim <= 384, n_out <= 16 (got %d / %d / %d)", hidden,
                in_dim, n_out);
    TSM_REQUIRE(T >= 1 && E >= 1 && scalar_stride >= 1 && scalar_offset >= 0 && scalar_offset < scalar_stride,
                "tsm_ctde_critic_rows_update: bad sizes (T %lld, E %lld)", (long long)T, (long long)E);
    TSM_REQUIRE(critic_params && joint_rows && rew && terminated && v_last && grad_slabs_out && loss_partial_out,
                "tsm_ctde_critic_rows_update: null pointer");
    const int64_t B = T * E;
    TSM_REQUIRE(n_blocks >= 1 && n_blocks <= ceil_div(B, kRows - 1), "tsm_ctde_critic_rows_update: n_blocks = %d out of range", n_blocks);
    TrainArgs g{};
    g.P = critic_params; g.obs = joint_rows; g.rows = nullptr; g.first_row = 0; g.Mr = B; g.tm_T = T; g.tm_E = E;
    g.K1 = in_dim; g.n_out = n_out; g.N = 1;
    g.rew = rew; g.term = terminated; g.sc_stride = scalar_stride; g.sc_off = scalar_offset; g.v_last = v_last;
    g.gamma = (float)gamma;
    g.slabs = grad_slabs_out; g.partial = loss_partial_out; g.stamps = g_tsm_stamps;
    return dispatch_train(pick_kj(in_dim), g, 1, n_blocks, tsm_stream(stream));
}
