// ppo_rows.hip -- one PPO gradient step of a 128-wide ACTOR (obs -> 128 -> 128 -> A logits) in a single launch:
// forward, Categorical log-prob / entropy, clip objective, and the whole backward pass, for the north star's roofline
// configuration (simple_spread N = 8: actor 48-128-128-5 next to a centralized critic, BASELINE configs[2]).
//
// Replaces, for the actor half of `PPO._update_with_batch` (/root/reference/tianshou/algorithm/modelfree/ppo.py:182-212):
//   dist = policy(minibatch).dist; advantage normalisation; ratio / clip / dual-clip surrogate; dist.entropy();
//   the actor's share of loss.backward().
// The PPO loss is separable -- the clip and entropy terms need the actor only, the value term the critic only (DESIGN.md
// section 4) -- so the critic of this configuration runs beside it on the dense GEMMs (csrc/dense.hip) with
// `tsm_ppo_loss_fwd_bwd(loss_kind = 2)` for the value term, and the two gradient halves meet in the Adam step.
//
// gfx950 mapping.  One persistent workgroup (512 threads = 8 waves, 2 per SIMD) per CU walks 32-sample tiles of the
// minibatch (tile t, t + grid, ...).  The actor's weights stay in LDS for the lifetime of the workgroup (W1 | W2 | W3:
// ~100 KB of the CU's 160 KB), a tile's activations next to them; every layer product is v_mfma_f32_16x16x4_f32 (exact
// f32, DESIGN.md "Why f32 MFMA").  Wave w owns output columns [16 w, 16 w + 16) of a layer and both 16-row halves of the
// tile, so a weight fragment read from LDS serves two MFMAs.  Weight gradients never leave registers between tiles:
// wave w accumulates the 16 x 128 block dW2[16 w ..][:] (8 accumulator tiles), dW1[16 w ..][:] and one tile of dW3
// across all tiles of its workgroup and writes them ONCE, as the workgroup's gradient slab (deterministic: tiles are
// assigned statically, slabs are summed in order by tsm_reduce_slabs / tsm_adam_step).
// Algorithmic HBM traffic per sample and step: obs 4 D + act 4 + logp_old 4 + adv 4 + id 8; activations
// never touch HBM (the dense path writes and re-reads 4 x 512 B per sample).
#include "actor_rows_dev.h"
#include <stdlib.h>

extern long long *g_tsm_stamps;  // abi.hip (diagnostics, tools/stamp_actor_rows.py)

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kH = 128;        // hidden width
constexpr int kRows = 32;      // samples per tile (two 16-row MFMA tiles)
constexpr int kThreads = 512;  // 8 waves
constexpr int kLdh = kH + 2;   // 130 = 2 x odd: conflict-free [lane & 15][lane >> 4] operand reads
constexpr int kLdo = 18;

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// W2 [128][128] -> LDS rows of kLdh floats: 16-B global loads (8 per thread, all in flight), 8-B LDS stores
__device__ __forceinline__ void stage_w2(float *dst, const float *__restrict__ src, bool aligned) {
    if (aligned) {
        float4 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = reinterpret_cast<const float4 *>(src)[threadIdx.x + u * kThreads];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e4 = threadIdx.x + u * kThreads, r = e4 >> 5, c = (e4 & 31) * 4;
            float *p = dst + r * kLdh + c;
            *reinterpret_cast<float2 *>(p) = make_float2(q[u].x, q[u].y);
            *reinterpret_cast<float2 *>(p + 2) = make_float2(q[u].z, q[u].w);
        }
    } else {
        for (int e = threadIdx.x; e < kH * kH; e += kThreads) dst[(e >> 7) * kLdh + (e & 127)] = src[e];
    }
    for (int e = threadIdx.x; e < kH * 2; e += kThreads) dst[(e >> 1) * kLdh + kH + (e & 1)] = 0.f;  // the two pad columns
}

struct RowsLay {  // LDS layout in floats
    int nJ, ld1, W1, W2, W3, B1, B2, B3, X, H1, H2, LG, total;
    __host__ __device__ explicit RowsLay(int D) {
        nJ = (D + 15) / 16;
        ld1 = 16 * nJ + 2;
        int o = 0;
        W1 = o; o += kH * ld1;
        W2 = o; o += kH * kLdh;
        W3 = o; o += 16 * kLdh;
        B1 = o; o += kH;
        B2 = o; o += kH;
        B3 = o; o += 16;
        X = o; o += kRows * ld1;
        H1 = o; o += kRows * kLdh;
        H2 = o; o += kRows * kLdh;
        LG = o; o += kRows * kLdo;
        total = o;
    }
};

using ActorArgs = TsmActorArgs;  // actor_rows_dev.h

#define ASTAMP(k) do { if (g.stamps && blockIdx.x == 0 && tid == 0 && it < 4) g.stamps[it * 16 + (k)] = (long long)wall_clock64(); } while (0)

template <int NJ>
__global__ __launch_bounds__(kThreads) void ppo_actor_rows_kernel(ActorArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const RowsLay ly(g.D);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c16 = lane & 15, kq = lane >> 4;
    const int D = g.D, A = g.A, ld1 = ly.ld1;
    const int64_t n_tiles = (g.M + kRows - 1) / kRows;
    // the gradient step this launch opens: the Adam launch behind it reads the advanced count (tsm_adam_step(step_dev))
    if (g.opt_step_dev && blockIdx.x == 0 && tid == 0) *g.opt_step_dev += 1;

    // ---- stage the weights once (zero pads: W1 columns >= D, W3 rows >= A) ----
    const int oW1 = 0, oB1 = kH * D, oW2 = oB1 + kH, oB2 = oW2 + kH * kH, oW3 = oB2 + kH, oB3 = oW3 + A * kH;
    tsm_stage_padded<kThreads>(lds + ly.W1, g.P + oW1, kH * ld1, ld1, kH, D, D);
    stage_w2(lds + ly.W2, g.P + oW2, (oW2 & 3) == 0);
    tsm_stage_padded<kThreads>(lds + ly.W3, g.P + oW3, 16 * kLdh, kLdh, A, kH, kH);
    if (tid < kH) { lds[ly.B1 + tid] = g.P[oB1 + tid]; lds[ly.B2 + tid] = g.P[oB2 + tid]; }
    if (tid < 16) lds[ly.B3 + tid] = tid < A ? g.P[oB3 + tid] : 0.f;

    // ---- persistent gradient accumulators ----
    f4 gW2[8], gW1[NJ], gW3;
#pragma unroll
    for (int i = 0; i < 8; ++i) gW2[i] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NJ; ++i) gW1[i] = f4{0.f, 0.f, 0.f, 0.f};
    gW3 = f4{0.f, 0.f, 0.f, 0.f};
    float gB = 0.f;               // threads 0..127: db1[tid]; 128..255: db2[tid - 128]; 256..271: db3[tid - 256]
    double t_clip = 0.0, t_ent = 0.0;  // sample leaders (lane 0 of every 16-lane group)

    // X tile of a tile: 32 samples x (4 nJ) float4 pieces; thread -> (sample, piece); prefetched one tile ahead
    const int n_piece = 4 * NJ;   // float4 pieces per padded row (16 nJ floats)
    const int xr = tid / n_piece, xp = tid - xr * n_piece;
    const bool x_thread = tid < kRows * n_piece;
    float xv[4] = {0.f, 0.f, 0.f, 0.f};
    // Sample ids go through the permutation: id -> row is two dependent global round trips.  The ids are therefore
    // fetched TWO tiles ahead and the rows / loss-head inputs one tile ahead, so that neither round trip is ever waited
    // for inside a tile (measured: the id wait in front of the row loads cost 1.0 us of a 14 us tile).
    // x threads: (sample xr, 16-B piece xp) of the X tile; head lanes: sample tid >> 4 (its 16 lanes hold the 16 logits)
    const int hs = tid >> 4, hj = tid & 15;
    int64_t idx_x = -1, idx_h = -1;   // ids of the tile whose rows are fetched next
    auto fetch_ids = [&](int64_t tile_) {
        idx_x = -1; idx_h = -1;
        if (tile_ >= n_tiles) return;
        const int64_t ix = tile_ * kRows + xr, ih = tile_ * kRows + hs;
        if (x_thread && ix < g.M) idx_x = g.perm ? g.perm[ix] : g.first_row + ix;
        if (ih < g.M) idx_h = g.perm ? g.perm[ih] : g.first_row + ih;
    };
    int h_act = 0, n_act = 0;         // loss-head inputs of the tile in progress / of the next tile (in flight)
    float h_adv = 0.f, h_lpo = 0.f, n_adv = 0.f, n_lpo = 0.f;
    auto fetch_rows = [&]() {          // rows of the ids in idx_x / idx_h
        xv[0] = xv[1] = xv[2] = xv[3] = 0.f;
        if (idx_x >= 0) {
            const float *src = g.obs + idx_x * D + 4 * xp;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * xp + j < D) xv[j] = src[j];
        }
        n_act = 0; n_adv = 0.f; n_lpo = 0.f;
        if (idx_h >= 0) {
            n_act = g.act[idx_h];
            n_adv = g.adv ? g.adv[idx_h] : 1.f;
            if (g.kind != 1) n_lpo = g.logp_old[idx_h];
        }
    };
    const float adv_mean = g.adv_norm ? g.adv_stats[0] : 0.f, adv_std = g.adv_norm ? g.adv_stats[1] : 1.f;
    int64_t tile = blockIdx.x;
    fetch_ids(tile);
    fetch_rows();
    fetch_ids(tile + gridDim.x);
    __syncthreads();  // weights staged

    for (int it = 0; tile < n_tiles; tile += gridDim.x, ++it) {
        ASTAMP(0);
        // ---- P0: commit the prefetched X tile (the previous tile's readers are behind the loop-end barrier) ----
        if (x_thread) {
#pragma unroll
            for (int j = 0; j < 4; ++j) lds[ly.X + xr * ld1 + 4 * xp + j] = xv[j];
        }
        h_act = n_act; h_adv = n_adv; h_lpo = n_lpo;
        __syncthreads();
        fetch_rows();                              // next tile's rows (its ids arrived during the previous tile) ...
        fetch_ids(tile + 2 * (int64_t)gridDim.x);  // ... and the ids of the tile after it: both fly during the whole tile

        ASTAMP(1);
        const int col = 16 * w + c16;
        // ---- P1: H1 = relu(X W1^T + b1) ----
        {
            f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
            const float *a = lds + ly.X + c16 * ld1 + kq;
            const float *b = lds + ly.W1 + col * ld1 + kq;
#pragma unroll
            for (int k0 = 0; k0 < 16 * NJ; k0 += 4) {
                const float bv = b[k0];
                acc[0] = mfma4(a[k0], bv, acc[0]);
                acc[1] = mfma4(a[16 * ld1 + k0], bv, acc[1]);
            }
            const float bb = lds[ly.B1 + col];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) lds[ly.H1 + (mt * 16 + kq * 4 + r) * kLdh + col] = fmaxf(acc[mt][r] + bb, 0.f);
        }
        __syncthreads();
        ASTAMP(2);
        // ---- P2: H2 = relu(H1 W2^T + b2) ----
        {
            f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
            const float *a = lds + ly.H1 + c16 * kLdh + kq;
            const float *b = lds + ly.W2 + col * kLdh + kq;
#pragma unroll
            for (int k0 = 0; k0 < kH; k0 += 4) {
                const float bv = b[k0];
                acc[0] = mfma4(a[k0], bv, acc[0]);
                acc[1] = mfma4(a[16 * kLdh + k0], bv, acc[1]);
            }
            const float bb = lds[ly.B2 + col];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) lds[ly.H2 + (mt * 16 + kq * 4 + r) * kLdh + col] = fmaxf(acc[mt][r] + bb, 0.f);
        }
        __syncthreads();
        ASTAMP(3);
        // ---- P3: logits = H2 W3^T + b3 (A padded to 16): waves 0 / 1 take the two row halves ----
        if (w < 2) {
            f4 acc = f4{0.f, 0.f, 0.f, 0.f};
            const float *a = lds + ly.H2 + (16 * w + c16) * kLdh + kq;
            const float *b = lds + ly.W3 + c16 * kLdh + kq;
#pragma unroll
            for (int k0 = 0; k0 < kH; k0 += 4) acc = mfma4(a[k0], b[k0], acc);
            const float bb = lds[ly.B3 + c16];
#pragma unroll
            for (int r = 0; r < 4; ++r) lds[ly.LG + (16 * w + kq * 4 + r) * kLdo + c16] = acc[r] + bb;
        }
        __syncthreads();
        ASTAMP(4);
        // ---- P4: loss head (ppo.py:183-196, 210); logits -> d loss / d logits in place.  16 lanes per sample (lane j holds
        //         action j) on all eight waves: the exponentials of a sample run side by side and are folded in action
        //         order by DPP row operations -- the additions of the one-lane-per-sample loop, in its order ----
        {
            float *lg = lds + ly.LG + hs * kLdo;
            const int64_t i = tile * kRows + hs;
            float outv = 0.f;
            if (i < g.M) {  // uniform over the sample's 16 lanes
                float obj, ent;
                outv = tsm_actor_head(g, lg[hj], hj, lane, h_act, h_adv, h_lpo, adv_mean, adv_std, obj, ent);
                if (hj == 0) { t_clip += obj; t_ent += ent; }
            }
            lg[hj] = outv;  // rows beyond M and actions beyond A: zero
        }
        __syncthreads();
        ASTAMP(5);
        // ---- P5: dW3 += dLG^T H2 ; db3 ; dH2 = (dLG W3) * relu'(H2) ----
        {
            const float *a = lds + ly.LG + kq * kLdo + c16;            // A[i = a][k = row]
            const float *b = lds + ly.H2 + kq * kLdh + col;            // B[k = row][j = hidden col]
#pragma unroll
            for (int r0 = 0; r0 < kRows; r0 += 4) gW3 = mfma4(a[r0 * kLdo], b[r0 * kLdh], gW3);
        }
        if (tid >= 256 && tid < 272) {
            float s = 0.f;
            for (int r = 0; r < kRows; ++r) s += lds[ly.LG + r * kLdo + (tid - 256)];
            gB += s;
        }
        f4 d2[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
        {
            const float *a = lds + ly.LG + c16 * kLdo + kq;            // A[i = row][k = a]
            const float *b = lds + ly.W3 + kq * kLdh + col;            // B[k = a][j = hidden col]
#pragma unroll
            for (int k0 = 0; k0 < 16; k0 += 4) {
                const float bv = b[k0 * kLdh];
                d2[0] = mfma4(a[k0], bv, d2[0]);
                d2[1] = mfma4(a[16 * kLdo + k0], bv, d2[1]);
            }
        }
        __syncthreads();  // every wave has read H2 for dW3
        ASTAMP(6);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float *p = lds + ly.H2 + (mt * 16 + kq * 4 + r) * kLdh + col;
                *p = *p > 0.f ? d2[mt][r] : 0.f;
            }
        __syncthreads();
        ASTAMP(7);
        // ---- P6: dW2 += dH2^T H1 ; db2 ; dH1 = (dH2 W2) * relu'(H1) ----
        {
            const float *a = lds + ly.H2 + kq * kLdh + col;            // A[i = out o][k = row]
            const float *b = lds + ly.H1 + kq * kLdh + c16;            // B[k = row][j = in col]
#pragma unroll
            for (int r0 = 0; r0 < kRows; r0 += 4) {
                const float av = a[r0 * kLdh];
#pragma unroll
                for (int ti = 0; ti < 8; ++ti) gW2[ti] = mfma4(av, b[r0 * kLdh + 16 * ti], gW2[ti]);
            }
        }
        if (tid >= 128 && tid < 256) {
            float s = 0.f;
            for (int r = 0; r < kRows; ++r) s += lds[ly.H2 + r * kLdh + (tid - 128)];
            gB += s;
        }
        f4 d1[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
        {
            const float *a = lds + ly.H2 + c16 * kLdh + kq;            // A[i = row][k = o]
            const float *b = lds + ly.W2 + kq * kLdh + col;            // B[k = o][j = in col]
#pragma unroll
            for (int k0 = 0; k0 < kH; k0 += 4) {
                const float bv = b[k0 * kLdh];
                d1[0] = mfma4(a[k0], bv, d1[0]);
                d1[1] = mfma4(a[16 * kLdh + k0], bv, d1[1]);
            }
        }
        __syncthreads();  // every wave has read H1 for dW2
        ASTAMP(8);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float *p = lds + ly.H1 + (mt * 16 + kq * 4 + r) * kLdh + col;
                *p = *p > 0.f ? d1[mt][r] : 0.f;
            }
        __syncthreads();
        ASTAMP(9);
        // ---- P7: dW1 += dH1^T X ; db1 ----
        {
            const float *a = lds + ly.H1 + kq * kLdh + col;            // A[i = out o][k = row]
            const float *b = lds + ly.X + kq * ld1 + c16;              // B[k = row][j = obs col]
#pragma unroll
            for (int r0 = 0; r0 < kRows; r0 += 4) {
                const float av = a[r0 * kLdh];
#pragma unroll
                for (int ti = 0; ti < NJ; ++ti) gW1[ti] = mfma4(av, b[r0 * ld1 + 16 * ti], gW1[ti]);
            }
        }
        if (tid < 128) {
            float s = 0.f;
            for (int r = 0; r < kRows; ++r) s += lds[ly.H1 + r * kLdh + tid];
            gB += s;
        }
        __syncthreads();  // X / H1 / H2 / LG are free for the next tile
        ASTAMP(10);
    }

    // ---- the workgroup's gradient slab: written once, streamed (consumed once, by the reduction kernel) ----
    float *slab = g.slabs + (size_t)blockIdx.x * (size_t)(oB3 + A);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = 16 * w + kq * 4 + r;
#pragma unroll
        for (int ti = 0; ti < NJ; ++ti) {
            const int i = 16 * ti + c16;
            if (i < D) __builtin_nontemporal_store(gW1[ti][r], slab + oW1 + o * D + i);
        }
#pragma unroll
        for (int ti = 0; ti < 8; ++ti) __builtin_nontemporal_store(gW2[ti][r], slab + oW2 + o * kH + 16 * ti + c16);
        const int a = kq * 4 + r;
        if (a < A) __builtin_nontemporal_store(gW3[r], slab + oW3 + a * kH + 16 * w + c16);
    }
    if (tid < 128) __builtin_nontemporal_store(gB, slab + oB1 + tid);
    else if (tid < 256) __builtin_nontemporal_store(gB, slab + oB2 + tid - 128);
    else if (tid < 256 + A) __builtin_nontemporal_store(gB, slab + oB3 + tid - 256);
    {   // loss statistics: the sample leaders (lane 0 of each 16-lane group) hold the terms; fixed order: wave sums, then waves
        __shared__ double s_stat[2][kThreads / 64];
        const double c = wave_sum(t_clip), e = wave_sum(t_ent);
        if (lane == 0) { s_stat[0][w] = c; s_stat[1][w] = e; }
        __syncthreads();
        if (tid == 0) {
            double cc = 0.0, ee = 0.0;
            for (int k = 0; k < kThreads / 64; ++k) { cc += s_stat[0][k]; ee += s_stat[1][k]; }
            g.partial[4 * blockIdx.x + 0] = cc;
            g.partial[4 * blockIdx.x + 1] = 0.0;
            g.partial[4 * blockIdx.x + 2] = ee;
            g.partial[4 * blockIdx.x + 3] = 0.0;
        }
    }
}

// ================================================================================================================
// The critic of the same configuration in one launch: value = MLP(joint row) with K1 = N * D inputs (centralized critic,
// 384 for simple_spread N = 8; K1 = D for a local critic), value term of the PPO loss (ppo.py:198-208) for the N agents of
// the row, and the critic's backward pass.
//
// The first layer's weights (128 x 384 f32 = 192 KB) exceed a CU's LDS, so layer 1 streams K-slices of 32 columns of W1
// and of the 32-row observation block through a double buffer (global -> registers -> LDS, one barrier per slice) while the
// layer's accumulators stay in registers; W2 stays resident like the actor's.  The weight gradient of layer 1 streams the
// observation slices a second time (they are L2-hot) against dH1 kept in LDS: wave w accumulates dW1[16 w ..][32 s .. 32 s + 32)
// for every slice s in registers (2 tiles x NS slices), across all blocks of its workgroup, and writes them once.
// ================================================================================================================
constexpr int kKs = 32, kLds = kKs + 2;

struct CritLay {
    int W2, W3, B1, B2, H1, H2, WS, XS, V, DV, total;
    __host__ __device__ CritLay() {
        int o = 0;
        W2 = o; o += kH * kLdh;
        W3 = o; o += kH;
        B1 = o; o += kH;
        B2 = o; o += kH;
        H1 = o; o += kRows * kLdh;
        H2 = o; o += kRows * kLdh;
        WS = o; o += 2 * kH * kLds;      // W1 slice, double buffered
        XS = o; o += 2 * kRows * kLds;   // observation slice, double buffered
        V = o; o += kRows;
        DV = o; o += kRows * 16;         // per-sample d loss / d value of a row's agents (N <= 16)
        total = o;
    }
};

struct CriticArgs {
    const float *P;          // critic parameters: w0[H][K1] b0[H] w1[H][H] b1[H] w2[1][H] b2[1]
    const float *obs;        // joint rows [n_rows][K1]
    const float *returns, *v_s_old;   // per SAMPLE (lane id = row * N + agent)
    const int64_t *rows;     // joint-row ids of the minibatch (nullable: first_row + i)
    int64_t first_row, Mr;   // rows in the minibatch
    int K1, N;
    float eps_clip, vf_coef;
    int value_clip;
    float *slabs;            // [grid][P]
    double *partial;         // [grid][4] = {0, sum vf, 0, 0}
    long long *stamps;       // diagnostics only (tsm_debug_set_stamps): phase time stamps of workgroup 0
};

#define CSTAMP(k) do { if (g.stamps && blockIdx.x == 0 && tid == 0) g.stamps[200 + (k)] = (long long)wall_clock64(); } while (0)

template <int NS>
__global__ __launch_bounds__(kThreads) void ppo_critic_rows_kernel(CriticArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const CritLay ly;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c16 = lane & 15, kq = lane >> 4;
    const int K1 = g.K1, N = g.N;
    const int64_t n_blocks = (g.Mr + kRows - 1) / kRows;
    const int oB1 = kH * K1, oW2 = oB1 + kH, oB2 = oW2 + kH * kH, oW3 = oB2 + kH, oB3 = oW3 + kH;
    stage_w2(lds + ly.W2, g.P + oW2, (oW2 & 3) == 0);
    if (tid < kH) { lds[ly.W3 + tid] = g.P[oW3 + tid]; lds[ly.B1 + tid] = g.P[oB1 + tid]; lds[ly.B2 + tid] = g.P[oB2 + tid]; }
    const float b3 = g.P[oB3];
    CSTAMP(0);

    // dW2 lives in registers for the whole launch; dW1 (2 NS tiles per wave: 96 registers at K1 = 384) only during a
    // block's layer-1 weight-gradient phase -- it is folded into the workgroup's slab (plain store for the first block,
    // read-add-store by the same lanes afterwards: fixed order) so that the other phases keep their registers
    f4 gW2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) gW2[i] = f4{0.f, 0.f, 0.f, 0.f};
    float *slab = g.slabs + (size_t)blockIdx.x * (size_t)(oB3 + 1);
    float gB = 0.f;   // threads 0..127: db1; 128..255: db2; 256..383: dW3[tid - 256]; 384: db3
    double t_vf = 0.0;

    // slice staging: W1 slice = 128 rows x 8 float4 (two per thread), observation slice = 32 rows x 8 float4 (threads < 256)
    const int wr0 = tid >> 3, wp = tid & 7;          // W1 rows wr0 and wr0 + 64, float4 piece wp
    const int xr = tid >> 3;                          // observation row (tid < 256)
    // three register sets: slice s + 3 is requested while slice s is multiplied, so a slice has ~3 compute phases to
    // arrive (one phase is ~0.4 us of MFMA work, an L2 round trip 1-2 us)
    f4 rw[3][2], rx[3];
    int64_t x_row = -1;
    auto fetch_slice = [&](int s, int set, bool with_w) {
        const int k = kKs * s + 4 * wp;
        const bool whole = k + 3 < K1 && (K1 & 3) == 0;  // 16-B aligned, inside the row
        if (with_w) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float *src = g.P + (size_t)(wr0 + 64 * h) * K1 + k;
                if (whole) {
                    const float4 q = *reinterpret_cast<const float4 *>(src);
                    rw[set][h] = f4{q.x, q.y, q.z, q.w};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) rw[set][h][j] = k + j < K1 ? src[j] : 0.f;
                }
            }
        }
        if (tid < 256) {
            if (x_row >= 0 && whole) {
                const float4 q = *reinterpret_cast<const float4 *>(g.obs + x_row * K1 + k);
                rx[set] = f4{q.x, q.y, q.z, q.w};
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) rx[set][j] = (x_row >= 0 && k + j < K1) ? g.obs[x_row * K1 + k + j] : 0.f;
            }
        }
    };
    auto commit_slice = [&](int buf, int set, bool with_w) {
        if (with_w) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float *dst = lds + ly.WS + buf * kH * kLds + (wr0 + 64 * h) * kLds + 4 * wp;  // 8-B aligned (34 = 2 x 17)
                *reinterpret_cast<float2 *>(dst) = make_float2(rw[set][h][0], rw[set][h][1]);
                *reinterpret_cast<float2 *>(dst + 2) = make_float2(rw[set][h][2], rw[set][h][3]);
            }
        }
        if (tid < 256) {
            float *dst = lds + ly.XS + buf * kRows * kLds + xr * kLds + 4 * wp;
            *reinterpret_cast<float2 *>(dst) = make_float2(rx[set][0], rx[set][1]);
            *reinterpret_cast<float2 *>(dst + 2) = make_float2(rx[set][2], rx[set][3]);
        }
    };
    __syncthreads();

    for (int64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const int col = 16 * w + c16;
        {
            const int64_t i = blk * kRows + xr;
            x_row = (tid < 256 && i < g.Mr) ? (g.rows ? g.rows[i] : g.first_row + i) : -1;
        }
        // value-loss inputs of (row tid / N, agent tid % N): row id -> returns / v_s_old are two dependent global round
        // trips; issued here, they fly during the whole forward pass
        float pf_ret = 0.f, pf_vs = 0.f;
        if (tid < kRows * N) {
            const int r_ = tid / N;
            const int64_t i_ = blk * kRows + r_;
            if (i_ < g.Mr) {
                const int64_t sidx_ = (g.rows ? g.rows[i_] : g.first_row + i_) * N + (tid - r_ * N);
                pf_ret = g.returns[sidx_];
                if (g.value_clip) pf_vs = g.v_s_old[sidx_];
            }
        }
        CSTAMP(1);
        // ---- L1: H1 = relu(X W1^T + b1), K-slices through the double buffer ----
        {
            f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
            fetch_slice(0, 0, true);
            if (NS > 1) fetch_slice(1, 1, true);
            if (NS > 2) fetch_slice(2, 2, true);
            commit_slice(0, 0, true);
            __syncthreads();
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (s + 3 < NS) fetch_slice(s + 3, s % 3, true);  // (set s % 3 went to LDS before slice s was multiplied)
                const float *pa = lds + ly.XS + (s & 1) * kRows * kLds + c16 * kLds + kq;
                const float *pb = lds + ly.WS + (s & 1) * kH * kLds + col * kLds + kq;
#pragma unroll
                for (int k0 = 0; k0 < kKs; k0 += 4) {
                    const float bv = pb[k0];
                    acc[0] = mfma4(pa[k0], bv, acc[0]);
                    acc[1] = mfma4(pa[16 * kLds + k0], bv, acc[1]);
                }
                if (s + 1 < NS) commit_slice((s + 1) & 1, (s + 1) % 3, true);
                __syncthreads();
            }
            const float bb = lds[ly.B1 + col];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[mt][r] + bb;
                    lds[ly.H1 + (mt * 16 + kq * 4 + r) * kLdh + col] = v > 0.f ? v : 0.f;
                }
        }
        __syncthreads();
        CSTAMP(2);
        // ---- L2 ----
        {
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
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[mt][r] + bb;
                    lds[ly.H2 + (mt * 16 + kq * 4 + r) * kLdh + col] = v > 0.f ? v : 0.f;
                }
        }
        __syncthreads();
        CSTAMP(3);
        // ---- L3: value of every row (one lane per row, k-ordered fma chain) ----
        if (tid < kRows) {
            const float *h = lds + ly.H2 + tid * kLdh;
            float sacc = 0.f;
            for (int j = 0; j < kH; ++j) sacc = fmaf(h[j], lds[ly.W3 + j], sacc);
            lds[ly.V + tid] = sacc + b3;
        }
        __syncthreads();
        CSTAMP(4);
        // ---- value term for the N agents of every row (ppo.py:198-208) ----
        if (tid < kRows * N) {
            const int r = tid / N, a = tid - r * N;
            const int64_t i = blk * kRows + r;
            float dv = 0.f;
            if (i < g.Mr) {
                const float v = lds[ly.V + r], ret = pf_ret;
                float vf, g_v;
                if (g.value_clip) {
                    const float vs = pf_vs;
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
                t_vf += vf;
            }
            lds[ly.DV + r * 16 + a] = dv;
        }
        __syncthreads();
        if (tid < kRows) {  // d loss / d value of the row = sum over its agents, in agent order
            float sacc = 0.f;
            for (int a = 0; a < N; ++a) sacc += lds[ly.DV + tid * 16 + a];
            lds[ly.V + tid] = sacc;
        }
        __syncthreads();
        CSTAMP(5);
        // ---- backward: dW3, db3 (VALU), dH2 = dv (x) w3 * relu'(H2) in place ----
        if (tid >= 256 && tid < 256 + kH) {
            const int j = tid - 256;
            float sacc = 0.f;
            for (int r = 0; r < kRows; ++r) sacc = fmaf(lds[ly.V + r], lds[ly.H2 + r * kLdh + j], sacc);
            gB += sacc;
        } else if (tid == 384) {
            float sacc = 0.f;
            for (int r = 0; r < kRows; ++r) sacc += lds[ly.V + r];
            gB += sacc;
        }
        __syncthreads();
        for (int e = tid; e < kRows * kH; e += kThreads) {
            const int r = e >> 7, j = e & (kH - 1);
            float *p = lds + ly.H2 + r * kLdh + j;
            *p = *p > 0.f ? lds[ly.V + r] * lds[ly.W3 + j] : 0.f;
        }
        __syncthreads();
        CSTAMP(6);
        // ---- dW2 += dH2^T H1 ; db2 ; dH1 = (dH2 W2) * relu'(H1) ----
        {
            const float *pa = lds + ly.H2 + kq * kLdh + col;
            const float *pb = lds + ly.H1 + kq * kLdh + c16;
#pragma unroll
            for (int r0 = 0; r0 < kRows; r0 += 4) {
                const float av = pa[r0 * kLdh];
#pragma unroll
                for (int ti = 0; ti < 8; ++ti) gW2[ti] = mfma4(av, pb[r0 * kLdh + 16 * ti], gW2[ti]);
            }
        }
        if (tid >= 128 && tid < 256) {
            float sacc = 0.f;
            for (int r = 0; r < kRows; ++r) sacc += lds[ly.H2 + r * kLdh + (tid - 128)];
            gB += sacc;
        }
        f4 d1[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
        {
            const float *pa = lds + ly.H2 + c16 * kLdh + kq;
            const float *pb = lds + ly.W2 + kq * kLdh + col;
#pragma unroll
            for (int k0 = 0; k0 < kH; k0 += 4) {
                const float bv = pb[k0 * kLdh];
                d1[0] = mfma4(pa[k0], bv, d1[0]);
                d1[1] = mfma4(pa[16 * kLdh + k0], bv, d1[1]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float *p = lds + ly.H1 + (mt * 16 + kq * 4 + r) * kLdh + col;
                *p = *p > 0.f ? d1[mt][r] : 0.f;
            }
        __syncthreads();
        if (tid < 128) {
            float sacc = 0.f;
            for (int r = 0; r < kRows; ++r) sacc += lds[ly.H1 + r * kLdh + tid];
            gB += sacc;
        }
        CSTAMP(7);
        // ---- dW1[:, slice s] += dH1^T X[:, slice s]: the observation slices once more (L2-hot) ----
        // slices in groups of GS: a group's 2 GS accumulator tiles live in registers, then go to the slab
        constexpr int GS = NS > 4 ? 4 : NS;
        const bool first_blk = blk == (int64_t)blockIdx.x;
        fetch_slice(0, 0, false);
        if (NS > 1) fetch_slice(1, 1, false);
        if (NS > 2) fetch_slice(2, 2, false);
        commit_slice(0, 0, false);
        __syncthreads();
#pragma unroll
        for (int g0 = 0; g0 < NS; g0 += GS) {
            f4 gW1[2 * GS];
#pragma unroll
            for (int i = 0; i < 2 * GS; ++i) gW1[i] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < GS; ++q) {
                const int s = g0 + q;
                if (s < NS) {
                    if (s + 3 < NS) fetch_slice(s + 3, s % 3, false);
                    const float *pa = lds + ly.H1 + kq * kLdh + col;                            // A[i = out o][k = row]
                    const float *pb = lds + ly.XS + (s & 1) * kRows * kLds + kq * kLds + c16;   // B[k = row][j = slice col]
#pragma unroll
                    for (int r0 = 0; r0 < kRows; r0 += 4) {
                        const float av = pa[r0 * kLdh];
                        gW1[2 * q] = mfma4(av, pb[r0 * kLds], gW1[2 * q]);
                        gW1[2 * q + 1] = mfma4(av, pb[r0 * kLds + 16], gW1[2 * q + 1]);
                    }
                    if (s + 1 < NS) commit_slice((s + 1) & 1, (s + 1) % 3, false);
                    __syncthreads();
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 16 * w + kq * 4 + r;
#pragma unroll
                for (int t2 = 0; t2 < 2 * GS; ++t2) {
                    const int k = kKs * g0 + 16 * t2 + c16;
                    if (k < K1) {
                        float *dst = slab + (size_t)o * K1 + k;
                        *dst = first_blk ? gW1[t2][r] : *dst + gW1[t2][r];
                    }
                }
            }
        }
    }

    CSTAMP(8);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = 16 * w + kq * 4 + r;
#pragma unroll
        for (int ti = 0; ti < 8; ++ti) __builtin_nontemporal_store(gW2[ti][r], slab + oW2 + o * kH + 16 * ti + c16);
    }
    if (tid < 128) __builtin_nontemporal_store(gB, slab + oB1 + tid);
    else if (tid < 256) __builtin_nontemporal_store(gB, slab + oB2 + tid - 128);
    else if (tid < 384) __builtin_nontemporal_store(gB, slab + oW3 + tid - 256);
    else if (tid == 384) __builtin_nontemporal_store(gB, slab + oB3);
    __shared__ double sm[kThreads / 64];
    const double tot = block_sum<double, kThreads>(t_vf, sm);
    CSTAMP(9);
    if (tid == 0) {
        g.partial[4 * blockIdx.x + 0] = 0.0;
        g.partial[4 * blockIdx.x + 1] = tot;
        g.partial[4 * blockIdx.x + 2] = 0.0;
        g.partial[4 * blockIdx.x + 3] = 0.0;
    }
}

int rows_supported(int32_t D, int32_t H, int32_t A) { return H == kH && D >= 1 && D <= 64 && A >= 1 && A <= 16; }

int n_cu() {
    static int cached = 0;
    if (!cached) {
        hipDeviceProp_t p;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cached = p.multiProcessorCount;
        if (cached <= 0) cached = 256;
    }
    return cached;
}

}  // namespace

TSM_EXPORT int tsm_ppo_actor_rows_supported(int32_t obs_dim, int32_t hidden, int32_t n_act) {
    return rows_supported(obs_dim, hidden, n_act);
}

TSM_EXPORT int64_t tsm_ppo_actor_rows_param_count(int32_t obs_dim, int32_t hidden, int32_t n_act) {
    if (!rows_supported(obs_dim, hidden, n_act)) return -1;
    return (int64_t)hidden * obs_dim + hidden + (int64_t)hidden * hidden + hidden + (int64_t)n_act * hidden + n_act;
}

// Which actor kernel serves a minibatch of M samples: 64-sample tiles with W2 in registers (csrc/actor_rows64.hip) once
// every CU gets at least one of them, the 32-sample tiles of this file below that (twice the tiles to spread over the CUs).
// Option "actor_tile" = 32 | 64 forces one (tsm_kernel_option_set; default from TSM_ACTOR_TILE): tests run the reference
// fixtures through both, tools time them A/B.
static bool actor_tile64(int64_t M) {
    if (const int forced = tsm_opt(TSM_OPT_ACTOR_TILE)) return forced == 64;
    return ceil_div(M, 64) >= n_cu();
}

// workgroups (= gradient slabs) for a minibatch of M samples: one per CU, never more than there are tiles
TSM_EXPORT int tsm_ppo_actor_rows_grid(int64_t M) {
    if (M <= 0) return 0;
    const int64_t tiles = ceil_div(M, actor_tile64(M) ? 64 : kRows);
    const int cu = n_cu();
    return (int)(tiles < cu ? tiles : cu);
}

// Raise the dynamic-LDS limit of EVERY instantiation of the two kernels (once per process: one process drives one GPU).  Hosts
// that capture the launches into a hipGraph call this first, outside the capture (GenericPPO does when it picks these
// kernels); the launch entry points call it too, so a plain caller needs nothing.
TSM_EXPORT int tsm_ppo_rows_init(void) {
    static bool done = false;
    if (done) return TSM_OK;
#define ALLOW(k) TSM_HIP(tsm_allow_max_lds(reinterpret_cast<const void *>(k)))
    ALLOW(ppo_actor_rows_kernel<1>); ALLOW(ppo_actor_rows_kernel<2>); ALLOW(ppo_actor_rows_kernel<3>); ALLOW(ppo_actor_rows_kernel<4>);
    ALLOW(ppo_critic_rows_kernel<1>); ALLOW(ppo_critic_rows_kernel<2>); ALLOW(ppo_critic_rows_kernel<3>);
#undef ALLOW
    if (const int rc = tsm_actor_rows64_init(); rc != TSM_OK) return rc;
    done = true;
    return TSM_OK;
}

TSM_EXPORT int tsm_ppo_actor_rows_update(const float *actor_params, int32_t obs_dim, int32_t hidden, int32_t n_act,
                                         const float *obs, const int32_t *act, const float *logp_old, const float *adv,
                                         const int64_t *perm, int64_t first_row, int64_t M, const float *adv_stats,
                                         const tsm_ppo_cfg *cfg, int32_t n_blocks, float *grad_slabs_out,
                                         double *loss_partial_out, int64_t *opt_step_dev, void *stream) {
    TSM_REQUIRE(rows_supported(obs_dim, hidden, n_act),
                "tsm_ppo_actor_rows_update supports hidden == 128, obs_dim <= 64, n_act <= 16 (got %d / %d / %d)", hidden,
                obs_dim, n_act);
    TSM_REQUIRE(M >= 1 && cfg, "tsm_ppo_actor_rows_update: empty minibatch or null cfg");
    TSM_REQUIRE(first_row >= 0 && first_row + M < ((int64_t)1 << 31), "tsm_ppo_actor_rows_update: row ids must stay below 2^31");
    TSM_REQUIRE(actor_params && obs && act && grad_slabs_out && loss_partial_out, "tsm_ppo_actor_rows_update: null pointer");
    TSM_REQUIRE(cfg->loss_kind == 1 || (logp_old && adv), "tsm_ppo_actor_rows_update: the clip objective needs logp_old and adv");
    TSM_REQUIRE(adv || !cfg->adv_norm, "tsm_ppo_actor_rows_update: adv_norm without adv");
    TSM_REQUIRE(!cfg->adv_norm || adv_stats, "tsm_ppo_actor_rows_update: adv_norm needs adv_stats");
    TSM_REQUIRE(cfg->loss_kind == 0 || cfg->loss_kind == 1, "tsm_ppo_actor_rows_update: loss_kind must be 0 or 1");
    TSM_REQUIRE(cfg->dual_clip <= 0.0 || cfg->dual_clip > 1.0,
                "Dual-clip PPO parameter should greater than 1.0 but got %g", cfg->dual_clip);
    const bool t64 = actor_tile64(M);
    TSM_REQUIRE(n_blocks >= 1 && n_blocks <= ceil_div(M, t64 ? 64 : kRows), "tsm_ppo_actor_rows_update: n_blocks = %d out of range",
                n_blocks);
    ActorArgs g{};
    g.P = actor_params; g.obs = obs; g.act = act; g.logp_old = logp_old; g.adv = adv; g.perm = perm;
    g.first_row = first_row; g.M = M; g.adv_stats = adv_stats; g.D = obs_dim; g.A = n_act;
    g.eps_clip = (float)cfg->eps_clip; g.dual_clip = (float)cfg->dual_clip; g.ent_coef = (float)cfg->ent_coef;
    g.adv_norm = cfg->adv_norm; g.kind = cfg->loss_kind;
    g.slabs = grad_slabs_out; g.partial = loss_partial_out;
    g.stamps = g_tsm_stamps;
    g.opt_step_dev = opt_step_dev;
    if (t64) return tsm_actor_rows64_launch(g, n_blocks, tsm_stream(stream));
    const RowsLay ly(obs_dim);
    const size_t shmem = (size_t)ly.total * sizeof(float);
    TSM_REQUIRE(shmem <= kTsmMaxLds, "tsm_ppo_actor_rows_update: LDS layout of %zu bytes does not fit", shmem);
    if (const int rc = tsm_ppo_rows_init(); rc != TSM_OK) return rc;
    hipStream_t st = tsm_stream(stream);
#define LAUNCH(NJ)                                                                                                     \
    do {                                                                                                               \
        hipLaunchKernelGGL((ppo_actor_rows_kernel<NJ>), dim3((unsigned)n_blocks), dim3(kThreads), shmem, st, g);       \
    } while (0)
    switch (ly.nJ) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 3: LAUNCH(3); break;
        default: LAUNCH(4); break;
    }
#undef LAUNCH
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

// ---- critic ----
TSM_EXPORT int tsm_ppo_critic_rows_supported(int32_t in_dim, int32_t hidden, int32_t n_agent) {
    // Up to three 32-column slices of the input (round 5): the instantiations for 4 / 6 / 8 / 12 slices spilled 7-14 vector registers
    // (tools/resource_usage.py) and nothing selects them any more -- wider critics take the two-launch step of csrc/critic_train.hip
    // (every width that is a multiple of 4, up to 384), odd widths above 96 the dense kernels.
    return hidden == kH && in_dim >= 1 && in_dim <= 3 * kKs && n_agent >= 1 && n_agent <= 16;
}

TSM_EXPORT int64_t tsm_ppo_critic_rows_param_count(int32_t in_dim, int32_t hidden) {
    if (hidden != kH || in_dim < 1) return -1;
    return (int64_t)hidden * in_dim + hidden + (int64_t)hidden * hidden + hidden + hidden + 1;
}

// workgroups (= gradient slabs) for Mr joint rows: blocks of 32 rows, one workgroup per CU (a block is a latency-bound
// chain of phases: measured 68 us with 256 workgroups x 1 block against 123 us with 128 x 2 at 8192 rows, although every
// workgroup writes a 264 KB slab for the 384-wide critic)
TSM_EXPORT int tsm_ppo_critic_rows_grid(int64_t Mr) {
    if (Mr <= 0) return 0;
    const int64_t blocks = ceil_div(Mr, kRows);
    const int cap = n_cu();
    return (int)(blocks < cap ? blocks : cap);
}

TSM_EXPORT int tsm_ppo_critic_rows_update(const float *critic_params, int32_t in_dim, int32_t hidden, int32_t n_agent,
                                          const float *obs_rows, const float *returns, const float *v_s_old,
                                          const int64_t *rows, int64_t first_row, int64_t Mr, const tsm_ppo_cfg *cfg,
                                          int32_t n_blocks, float *grad_slabs_out, double *loss_partial_out, void *stream) {
    TSM_REQUIRE(tsm_ppo_critic_rows_supported(in_dim, hidden, n_agent),
                "tsm_ppo_critic_rows_update supports hidden == 128, in_dim <= 384, n_agent <= 16 (got %d / %d / %d)", hidden,
                in_dim, n_agent);
    TSM_REQUIRE(Mr >= 1 && cfg, "tsm_ppo_critic_rows_update: empty minibatch or null cfg");
    TSM_REQUIRE(critic_params && obs_rows && returns && grad_slabs_out && loss_partial_out,
                "tsm_ppo_critic_rows_update: null pointer");
    TSM_REQUIRE(!cfg->value_clip || v_s_old, "tsm_ppo_critic_rows_update: value_clip needs v_s_old");
    TSM_REQUIRE(n_blocks >= 1 && n_blocks <= ceil_div(Mr, kRows), "tsm_ppo_critic_rows_update: n_blocks = %d out of range", n_blocks);
    CriticArgs g{};
    g.P = critic_params; g.obs = obs_rows; g.returns = returns; g.v_s_old = v_s_old; g.rows = rows;
    g.first_row = first_row; g.Mr = Mr; g.K1 = in_dim; g.N = n_agent;
    g.eps_clip = (float)cfg->eps_clip; g.vf_coef = (float)cfg->vf_coef; g.value_clip = cfg->value_clip;
    g.slabs = grad_slabs_out; g.partial = loss_partial_out;
    g.stamps = g_tsm_stamps;
    const CritLay ly;
    const size_t shmem = (size_t)ly.total * sizeof(float);
    const int ns = (in_dim + kKs - 1) / kKs;
    if (const int rc = tsm_ppo_rows_init(); rc != TSM_OK) return rc;
    hipStream_t st = tsm_stream(stream);
#define LAUNCHC(NS)                                                                                                    \
    case NS:                                                                                                           \
        hipLaunchKernelGGL((ppo_critic_rows_kernel<NS>), dim3((unsigned)n_blocks), dim3(kThreads), shmem, st, g);      \
        break;
    switch (ns) {
        LAUNCHC(1) LAUNCHC(2) LAUNCHC(3)
        default:
            TSM_REQUIRE(false, "tsm_ppo_critic_rows_update: in_dim = %d needs %d slices of 32 (instantiated: 1, 2, 3)", in_dim, ns);
    }
#undef LAUNCHC
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}
