// actor_rows64.hip -- the one-launch ACTOR gradient step (obs -> 128 -> 128 -> A logits: forward, Categorical log-prob /
// entropy, clip objective, whole backward pass) on 64-sample tiles with the layer-2 weights in REGISTERS.
//
// Replaces, like csrc/ppo_rows.hip's `ppo_actor_rows_kernel` (whose arguments, slab layout and loss head it shares,
// actor_rows_dev.h), the actor half of `PPO._update_with_batch`
// (/root/reference/tianshou/algorithm/modelfree/ppo.py:182-212) and, with loss_kind 1 and adv = NULL, the actor term of
// `CTDEPolicy.learn` (multiagent/ctde.py:174-185).  It serves minibatches with at least one 64-sample tile per CU
// (tsm_ppo_actor_rows_grid decides); smaller ones keep the 32-sample kernel, which has twice the tiles to spread.
//
// Why a second kernel.  The 32-sample kernel's phase stamps (profiles/r02_stamp_actor_rows.txt) put its two big phases at
// the matrix pipe's issue rate (layer 2: 2.24 us, dW2 + dH1: 4.28 us of a 12.5 us tile) and ~3.8 us of a tile in what is NOT
// MFMA work: ten barrier-separated phases each pay a pipeline fill, the logits run on two waves, commits / head / masked
// write-backs are latency.  Those costs are per TILE, not per sample: a 64-sample tile halves them per sample.  W2 (66 KB in
// LDS) is what stands in the way of the larger activation buffers, so it moves into registers: wave w owns output columns
// [16 w, 16 w + 16) of layer 2 and input columns [16 w, 16 w + 16) of its transpose, i.e. 32 + 32 B-operand fragments
// (v_mfma_f32_16x16x4_f32: lane (c16, kq) holds B[k = 4 ks + kq][n = c16]), staged once per launch through the LDS region
// the activations use afterwards.  A weight fragment then serves FOUR MFMAs (four 16-row blocks) without an LDS read.
// Same k order per output as the 32-sample kernel for layers 1 and 2 and every gradient; the logits are summed as two
// 64-long halves on all eight waves (instead of one 128-long chain on two).
#include "actor_rows_dev.h"

extern long long *g_tsm_stamps;  // abi.hip (diagnostics, tools/stamp_actor_rows.py)

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kH = 128;        // hidden width
constexpr int kRows = 64;      // samples per tile (four 16-row MFMA tiles)
constexpr int kRT = 4;
constexpr int kThreads = 512;  // 8 waves
constexpr int kLdh = kH + 2;   // 130 = 2 x odd: conflict-free [lane & 15][lane >> 4] operand reads
constexpr int kLdo = 18;
constexpr int kPF = 1;         // steps (of <= 4 MFMAs) an MFMA stream's LDS operands are read ahead, mfma_stream

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// LDS offsets above 64 KB as opaque scalars (critic_rows_dev.h has the story: one address VGPR per access otherwise)
__device__ __forceinline__ int opaque_s(int x) {
    asm volatile("" : "+s"(x));
    return x;
}

// An MFMA stream whose LDS operands are fetched PF steps ahead of the MFMAs that use them (round 5).  Written as the plain loop
// "read the step's operands, issue its MFMAs", hipcc emits ds_read -> s_waitcnt lgkmcnt(0) -> MFMAs per step: the wave sits out one
// LDS latency (100+ cycles under load) per 2-8 MFMAs (64-256 cycles), and the partner wave of the SIMD, in the same phase of
// the same loop, cannot cover all of it -- the big phases ran at 0.77-0.81 of the matrix pipe's rate.  ld(s, r) reads step s's NR
// operand values, mm(s, r) issues its MFMAs; both are unrolled, the ring of PF + 1 register sets is renaming, not copying.
// The order of the MFMAs, and with it every sum, is the plain loop's.
template <int N, int PF, int NR, typename L, typename C>
__device__ __forceinline__ void mfma_stream(L ld, C mm) {
    float r[PF + 1][NR];
#pragma unroll
    for (int s = 0; s < PF && s < N; ++s) ld(s, r[s % (PF + 1)]);
#pragma unroll
    for (int s = 0; s < N; ++s) {
        if (s + PF < N) ld(s + PF, r[(s + PF) % (PF + 1)]);
        __builtin_amdgcn_sched_barrier(0);   // (without it the scheduler sinks the reads back to their first use)
        mm(s, r[s % (PF + 1)]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

struct Lay64 {  // LDS layout in floats
    int nJ = 0, ld1 = 0, W1 = 0, W3 = 0, B1 = 0, B2 = 0, B3 = 0, U = 0, X = 0, H1 = 0, H2 = 0, LG = 0, HI = 0, total = 0;
    __host__ __device__ constexpr explicit Lay64(int D) {
        nJ = (D + 15) / 16;
        ld1 = 16 * nJ + 2;
        int o = 0;
        W1 = o; o += kH * ld1;
        W3 = o; o += 16 * kLdh;
        B1 = o; o += kH;
        B2 = o; o += kH;
        B3 = o; o += 16;
        U = o;                       // W2 [128][kLdh] while the fragments are loaded; the tile's activations afterwards
        X = o; o += 2 * kRows * ld1;     // two buffers: the next tile's rows land in the other one while this tile is computed
        H1 = o; o += kRows * kLdh;
        H2 = o; o += kRows * kLdh;
        LG = o; o += 2 * kRows * kLdo;   // the two k-halves of the logits; d loss / d logits in the first half afterwards
        HI = o; o += 2 * kRows * 4;      // loss-head inputs [buffer][sample]{action, advantage, old log-prob, -}
        const int w2_end = U + kH * kLdh;
        total = o > w2_end ? o : w2_end;
    }
};

#define ASTAMP(k) do { if (g.stamps && blockIdx.x == 0 && tid == 0 && it < 4) g.stamps[it * 16 + (k)] = (long long)wall_clock64(); } while (0)

template <int NJ>
__global__ __launch_bounds__(kThreads) void actor_rows64_kernel(TsmActorArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // the layout is a function of NJ alone (ld1 = 16 NJ + 2): compile-time, so every LDS offset is an immediate instead of one of
    // ~12 scalar registers (the kernel spilled 43 SGPRs to VGPR lanes: 124 v_readlane per tile) and of a v_add per access
    constexpr Lay64 ly(16 * NJ);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c16 = lane & 15, kq = lane >> 4;
    const int D = g.D, A = g.A;
    constexpr int ld1 = ly.ld1;
    const int64_t n_tiles = (g.M + kRows - 1) / kRows;
    if (g.opt_step_dev && blockIdx.x == 0 && tid == 0) *g.opt_step_dev += 1;  // the gradient step this launch opens
    if (g.stamps && blockIdx.x == 0 && tid == 0) g.stamps[200] = (long long)wall_clock64();
    if (g.stamps && tid == 0 && blockIdx.x < 512) g.stamps[1024 + 2 * blockIdx.x] = (long long)wall_clock64();   // (>= 2048 slots) start / end of every workgroup
    const int col = 16 * w + c16;

    // ---- stage the weights once (zero pads: W1 columns >= D, W3 rows >= A); W2 passes through LDS into registers ----
    const int oW1 = 0, oB1 = kH * D, oW2 = oB1 + kH, oB2 = oW2 + kH * kH, oW3 = oB2 + kH, oB3 = oW3 + A * kH;
    // (W1 and W3 as BATCHES of loads, all in flight before the first LDS store: written as plain load -> store loops they ran
    // one memory round trip per iteration, 13 + 5 of them -- most of an 8.4 us prologue, tools/stamp_actor_rows.py)
    constexpr int kLd1 = 16 * NJ + 2, kN1 = (kH * kLd1 + kThreads - 1) / kThreads, kN3 = (16 * kLdh + kThreads - 1) / kThreads;
    float w1q[kN1], w3q[kN3];
#pragma unroll
    for (int u = 0; u < kN1; ++u) {
        const int e = tid + u * kThreads, r = e / kLd1, c = e - r * kLd1;
        const bool ok = e < kH * kLd1 && c < D;
        const float v = g.P[ok ? oW1 + r * D + c : 0];   // (clamped, always-valid address + select: no divergent branch)
        w1q[u] = ok ? v : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kN3; ++u) {
        const int e = tid + u * kThreads, r = e / kLdh, c = e - r * kLdh;
        const bool ok = e < 16 * kLdh && r < A && c < kH;
        const float v = g.P[ok ? oW3 + r * kH + c : 0];
        w3q[u] = ok ? v : 0.f;
    }
    {
        float *dst = lds + ly.U;
        const float *src = g.P + oW2;
        if ((oW2 & 3) == 0) {
            float4 q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = reinterpret_cast<const float4 *>(src)[tid + u * kThreads];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e4 = tid + u * kThreads, r = e4 >> 5, c = (e4 & 31) * 4;
                float *p = dst + r * kLdh + c;
                *reinterpret_cast<float2 *>(p) = make_float2(q[u].x, q[u].y);
                *reinterpret_cast<float2 *>(p + 2) = make_float2(q[u].z, q[u].w);
            }
        } else {
            for (int e = tid; e < kH * kH; e += kThreads) dst[(e >> 7) * kLdh + (e & 127)] = src[e];
        }
    }
#pragma unroll
    for (int u = 0; u < kN1; ++u) {
        const int e = tid + u * kThreads;
        if (e < kH * kLd1) lds[ly.W1 + e] = w1q[u];
    }
#pragma unroll
    for (int u = 0; u < kN3; ++u) {
        const int e = tid + u * kThreads;
        if (e < 16 * kLdh) lds[ly.W3 + e] = w3q[u];
    }
    if (tid < kH) { lds[ly.B1 + tid] = g.P[oB1 + tid]; lds[ly.B2 + tid] = g.P[oB2 + tid]; }
    if (tid < 16) lds[ly.B3 + tid] = tid < A ? g.P[oB3 + tid] : 0.f;
    __syncthreads();
    float w2f[32], w2b[32];   // layer 2 forward: B[k = in][n = out col] = W2[col][k];  backward: B[k = out][n = in col] = W2[k][col]
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
        w2f[ks] = lds[ly.U + col * kLdh + 4 * ks + kq];
        w2b[ks] = lds[ly.U + (4 * ks + kq) * kLdh + col];
    }

    // ---- persistent gradient accumulators ----
    f4 gW2[8], gW1[NJ], gW3;
#pragma unroll
    for (int i = 0; i < 8; ++i) gW2[i] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NJ; ++i) gW1[i] = f4{0.f, 0.f, 0.f, 0.f};
    gW3 = f4{0.f, 0.f, 0.f, 0.f};
    // bias gradients = column sums of dH1 / dH2 / dLG over the tile's rows, as per-thread PARTIAL sums (a serial loop of a few
    // threads over 64 rows kept seven waves waiting at the next barrier: 1.4 us per tile): thread (c = tid & 127, q = tid >> 7)
    // sums rows [16 q, 16 q + 16) of column c; for the 16 logit columns thread (c = tid & 15, q = tid >> 4 < 16) rows [4 q, 4 q + 4).
    // The partials are folded once, after the last tile.
    float gB1 = 0.f, gB2 = 0.f, gB3 = 0.f;
    const int bc = tid & 127, bq = tid >> 7;
    // loss statistics of the sample leaders (lane 0 of every 16-lane group), in LDS between the tiles (four registers the loop
    // does not have; only P4 touches them)
    __shared__ double s_tstat[kThreads / 16][2];
    if ((tid & 15) == 0) { s_tstat[tid >> 4][0] = 0.0; s_tstat[tid >> 4][1] = 0.0; }
    // relu'(H1), relu'(H2) of the 16 elements this lane writes (bit mt * 4 + r): the backward pass masks with them instead of
    // reading each element back (16 dependent LDS read -> write pairs per phase, 0.55-0.6 us per tile each, profiles/r04_stamp_actor_rows.txt)
    unsigned live1 = 0, live2 = 0;

    // X tile: 64 samples x (4 NJ) 16-B pieces = two (sample, piece) slots per thread; loss-head inputs: samples hs, hs + 32
    // of the tile on the 16 lanes of group hs.  Ids are fetched TWO tiles ahead, rows / head inputs one tile ahead (id -> row
    // is two dependent global round trips; neither is waited for inside a tile).
    constexpr int n_piece = 4 * NJ;
    const int hs = tid >> 4, hj = tid & 15;
    int xr[2], xp[2];
    bool xon[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int e = tid + p * kThreads;
        xr[p] = e / n_piece; xp[p] = e - xr[p] * n_piece;
        xon[p] = e < kRows * n_piece;
    }
    float xv[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // Row ids as 32-bit integers (the host checks first_row + M and every permutation entry against 2^31: tsm_ppo_actor_rows_update):
    // a row address is then ONE v_mad_u64_u32 off the base pointer instead of a 64 x 32-bit multiply chain, and a whole-width row
    // piece (D == 16 NJ: 48 at BASELINE configs[2]) ONE 16-byte load instead of four predicated 4-byte loads -- the id / row fetches
    // were ~300 of the ~820 non-MFMA vector instructions of a tile (round 5; f32 MFMAs and VALU instructions take turns on a SIMD)
    int idx_x[2] = {-1, -1}, idx_h[2] = {-1, -1};
    const bool full_rows = D == 16 * NJ;
    auto fetch_ids = [&](int64_t tile_) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            idx_x[p] = -1; idx_h[p] = -1;
            if (tile_ >= n_tiles) continue;
            const int64_t ix = tile_ * kRows + xr[p], ih = tile_ * kRows + hs + 32 * p;
            if (xon[p] && ix < g.M) idx_x[p] = (int)(g.perm ? g.perm[ix] : g.first_row + ix);
            if (ih < g.M) idx_h[p] = (int)(g.perm ? g.perm[ih] : g.first_row + ih);
        }
    };
    int n_act[2] = {0, 0};
    float n_adv[2] = {0.f, 0.f}, n_lpo[2] = {0.f, 0.f};
    auto fetch_rows = [&]() {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            xv[p][0] = xv[p][1] = xv[p][2] = xv[p][3] = 0.f;
            if (idx_x[p] >= 0) {
                const float *src = g.obs + ((uint64_t)(uint32_t)idx_x[p] * (uint32_t)D + (uint32_t)(4 * xp[p]));
                if (full_rows) {   // (uniform: rows are a whole number of 16-byte pieces, 16-byte aligned off a 256-byte aligned base)
                    const float4 q = *reinterpret_cast<const float4 *>(src);
                    xv[p][0] = q.x; xv[p][1] = q.y; xv[p][2] = q.z; xv[p][3] = q.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (4 * xp[p] + j < D) xv[p][j] = src[j];
                }
            }
            n_act[p] = 0; n_adv[p] = 0.f; n_lpo[p] = 0.f;
            if (idx_h[p] >= 0) {
                const uint32_t ih = (uint32_t)idx_h[p];
                n_act[p] = g.act[ih];
                n_adv[p] = g.adv ? g.adv[ih] : 1.f;
                if (g.kind != 1) n_lpo[p] = g.logp_old[ih];
            }
        }
    };
    const float adv_mean = g.adv_norm ? g.adv_stats[0] : 0.f, adv_std = g.adv_norm ? g.adv_stats[1] : 1.f;
    // The fetched rows and head inputs go to LDS buffer `buf` (round 5: they were held in 20 registers for a whole tile and
    // committed in a phase of their own, P0, 0.4-1.0 us and a barrier per tile): fetched between the forward and the backward pass of
    // tile t, written before its P6, read from P1 of tile t + 1 on -- behind the loop-end barrier.
    auto commit_rows = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (xon[p]) {
                float *dst = lds + ly.X + buf * (kRows * ld1) + xr[p] * ld1 + 4 * xp[p];   // (8-byte aligned: ld1 is even)
                *reinterpret_cast<float2 *>(dst) = make_float2(xv[p][0], xv[p][1]);
                *reinterpret_cast<float2 *>(dst + 2) = make_float2(xv[p][2], xv[p][3]);
            }
            if (hj == 0) {
                float *hi = lds + ly.HI + (buf * kRows + hs + 32 * p) * 4;
                *reinterpret_cast<float2 *>(hi) = make_float2(__int_as_float(n_act[p]), n_adv[p]);
                hi[2] = n_lpo[p];
            }
        }
    };
    int64_t tile = blockIdx.x;
    fetch_ids(tile);
    fetch_rows();
    fetch_ids(tile + gridDim.x);
    __syncthreads();  // every wave holds its W2 fragments: region U is free for the activations
    commit_rows(0);
    __syncthreads();

    if (g.stamps && blockIdx.x == 0 && tid == 0) { g.stamps[201] = (long long)wall_clock64(); g.stamps[204] = (long long)__builtin_amdgcn_s_memtime(); }
    for (int it = 0; tile < n_tiles; tile += gridDim.x, ++it) {
        // (made opaque INSIDE the loop: as loop invariants the phase base addresses built from them were hoisted and held ~30
        // VGPRs for the whole loop -- the kernel sits at the 256-register cap of two waves per SIMD)
        const int oX = opaque_s(ly.X), oH1 = opaque_s(ly.H1), oH2 = opaque_s(ly.H2), oLG = opaque_s(ly.LG), oW1l = opaque_s(ly.W1),
                  oW3l = opaque_s(ly.W3);
        const int cur = it & 1;
        const int oXc = oX + cur * (kRows * ld1);   // this tile's rows
        ASTAMP(0);
        ASTAMP(1);
        // ---- P1: H1 = relu(X W1^T + b1) ----
        {
            f4 acc[kRT];
#pragma unroll
            for (int mt = 0; mt < kRT; ++mt) acc[mt] = f4{0.f, 0.f, 0.f, 0.f};
            const float *a = lds + oXc + c16 * ld1 + kq;
            const float *b = lds + oW1l + col * ld1 + kq;
            mfma_stream<4 * NJ, kPF, kRT + 1>(
                [&](int s, float (&x)[kRT + 1]) {
                    x[kRT] = b[4 * s];
#pragma unroll
                    for (int mt = 0; mt < kRT; ++mt) x[mt] = a[mt * 16 * ld1 + 4 * s];
                },
                [&](int, const float (&x)[kRT + 1]) {
#pragma unroll
                    for (int mt = 0; mt < kRT; ++mt) acc[mt] = mfma4(x[mt], x[kRT], acc[mt]);
                });
            const float bb = lds[ly.B1 + col];
            live1 = 0;
#pragma unroll
            for (int mt = 0; mt < kRT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float h = fmaxf(acc[mt][r] + bb, 0.f);
                    live1 |= (unsigned)(h > 0.f) << (mt * 4 + r);
                    lds[oH1 + (mt * 16 + kq * 4 + r) * kLdh + col] = h;
                }
            asm volatile("" : "+v"(live1));   // (one register of bits, not sixteen compare results held as lane masks until the backward pass)
        }
        __syncthreads();
        ASTAMP(2);
        // ---- P2: H2 = relu(H1 W2^T + b2), W2 fragments from registers ----
        {
            f4 acc[kRT];
#pragma unroll
            for (int mt = 0; mt < kRT; ++mt) acc[mt] = f4{0.f, 0.f, 0.f, 0.f};
            const float *a = lds + oH1 + c16 * kLdh + kq;
            mfma_stream<32, kPF, kRT>(
                [&](int ks, float (&x)[kRT]) {
#pragma unroll
                    for (int mt = 0; mt < kRT; ++mt) x[mt] = a[mt * 16 * kLdh + 4 * ks];
                },
                [&](int ks, const float (&x)[kRT]) {
#pragma unroll
                    for (int mt = 0; mt < kRT; ++mt) acc[mt] = mfma4(x[mt], w2f[ks], acc[mt]);
                });
            ASTAMP(11);
            const float bb = lds[ly.B2 + col];
            live2 = 0;
#pragma unroll
            for (int mt = 0; mt < kRT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float h = fmaxf(acc[mt][r] + bb, 0.f);
                    live2 |= (unsigned)(h > 0.f) << (mt * 4 + r);
                    lds[oH2 + (mt * 16 + kq * 4 + r) * kLdh + col] = h;
                }
            asm volatile("" : "+v"(live2));
            ASTAMP(12);
        }
        __syncthreads();
        // the next tile's rows and head inputs (its ids arrived during the previous tile): issued where no 16-register accumulator set
        // is live, in flight during the logits, the loss head and dW3 / dH2
        fetch_rows();
        ASTAMP(3);
        // ---- P3: logits = H2 W3^T + b3 (A padded to 16) on all eight waves: wave w takes row block w & 3 and the k half
        //         w >> 2; the head adds the two halves and the bias ----
        {
            const int mt = w & 3, kh = w >> 2;
            f4 acc = f4{0.f, 0.f, 0.f, 0.f};
            const float *a = lds + oH2 + (16 * mt + c16) * kLdh + kq + 64 * kh;
            const float *b = lds + oW3l + c16 * kLdh + kq + 64 * kh;
            mfma_stream<8, kPF, 4>(
                [&](int s, float (&x)[4]) { x[0] = a[8 * s]; x[1] = b[8 * s]; x[2] = a[8 * s + 4]; x[3] = b[8 * s + 4]; },
                [&](int, const float (&x)[4]) { acc = mfma4(x[0], x[1], acc); acc = mfma4(x[2], x[3], acc); });
#pragma unroll
            for (int r = 0; r < 4; ++r) lds[oLG + (kh * kRows + 16 * mt + kq * 4 + r) * kLdo + c16] = acc[r];
        }
        __syncthreads();
        ASTAMP(4);
        // ---- P4: loss head (ppo.py:183-196, 210); logits -> d loss / d logits (first LG half).  16 lanes per sample ----
        {   // (both samples of a lane group side by side: the two exp / log chains overlap; rows beyond M run on zeros and are masked)
            float logit[2], outv[2], obj[2], ent[2];
            bool live[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int s_ = hs + 32 * p;
                const float *lg = lds + oLG + s_ * kLdo;
                live[p] = tile * kRows + s_ < g.M;  // uniform over the sample's 16 lanes
                logit[p] = (lg[hj] + lg[kRows * kLdo + hj]) + lds[ly.B3 + hj];
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const float *hi = lds + ly.HI + (cur * kRows + hs + 32 * p) * 4;   // (one address per 16-lane group: a broadcast read)
                outv[p] = tsm_actor_head(g, live[p] ? logit[p] : 0.f, hj, lane, __float_as_int(hi[0]), hi[1], hi[2], adv_mean, adv_std,
                                         obj[p], ent[p]);
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                if (live[p] && hj == 0) { s_tstat[hs][0] += obj[p]; s_tstat[hs][1] += ent[p]; }
                lds[oLG + (hs + 32 * p) * kLdo + hj] = live[p] ? outv[p] : 0.f;  // rows beyond M and actions beyond A: zero
            }
        }
        __syncthreads();
        ASTAMP(5);
        // ---- P5: dW3 += dLG^T H2 ; db3 ; dH2 = (dLG W3) * relu'(H2) ----
        {
            const float *a = lds + oLG + kq * kLdo + c16;            // A[i = a][k = row]
            const float *b = lds + oH2 + kq * kLdh + col;            // B[k = row][j = hidden col]
            mfma_stream<kRows / 8, kPF, 4>(
                [&](int s, float (&x)[4]) {
                    x[0] = a[8 * s * kLdo]; x[1] = b[8 * s * kLdh]; x[2] = a[(8 * s + 4) * kLdo]; x[3] = b[(8 * s + 4) * kLdh];
                },
                [&](int, const float (&x)[4]) { gW3 = mfma4(x[0], x[1], gW3); gW3 = mfma4(x[2], x[3], gW3); });
        }
        if (tid < 256) {
            const float *q = lds + oLG + 4 * (tid >> 4) * kLdo + (tid & 15);
            gB3 += (q[0] + q[kLdo]) + (q[2 * kLdo] + q[3 * kLdo]);
        }
        {
            f4 d2[kRT];
#pragma unroll
            for (int mt = 0; mt < kRT; ++mt) d2[mt] = f4{0.f, 0.f, 0.f, 0.f};
            const float *a = lds + oLG + c16 * kLdo + kq;            // A[i = row][k = a]
            const float *b = lds + oW3l + kq * kLdh + col;           // B[k = a][j = hidden col]
            mfma_stream<4, kPF, kRT + 1>(
                [&](int s, float (&x)[kRT + 1]) {
                    x[kRT] = b[4 * s * kLdh];
#pragma unroll
                    for (int mt = 0; mt < kRT; ++mt) x[mt] = a[mt * 16 * kLdo + 4 * s];
                },
                [&](int, const float (&x)[kRT + 1]) {
#pragma unroll
                    for (int mt = 0; mt < kRT; ++mt) d2[mt] = mfma4(x[mt], x[kRT], d2[mt]);
                });
            __syncthreads();  // every wave has read H2 for dW3
            ASTAMP(6);
#pragma unroll
            for (int mt = 0; mt < kRT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)   // relu'(H2) from the bits this lane kept when it wrote the same elements of H2
                    lds[oH2 + (mt * 16 + kq * 4 + r) * kLdh + col] = (live2 >> (mt * 4 + r) & 1u) ? d2[mt][r] : 0.f;
        }
        commit_rows(cur ^ 1);
        fetch_ids(tile + 2 * (int64_t)gridDim.x);   // the ids of the tile after the next: in flight until the next tile's P3
        __syncthreads();
        ASTAMP(7);
        // ---- P6: dW2 += dH2^T H1 ; db2 ; dH1 = (dH2 W2) * relu'(H1), W2 fragments from registers ----
        {
            const float *a = lds + oH2 + kq * kLdh + col;            // A[i = out o][k = row]
            const float *b = lds + oH1 + kq * kLdh + c16;            // B[k = row][j = in col]
            mfma_stream<kRows / 4, 1, 9>(
                [&](int s, float (&x)[9]) {
                    x[8] = a[4 * s * kLdh];
#pragma unroll
                    for (int ti = 0; ti < 8; ++ti) x[ti] = b[4 * s * kLdh + 16 * ti];
                },
                [&](int, const float (&x)[9]) {
#pragma unroll
                    for (int ti = 0; ti < 8; ++ti) gW2[ti] = mfma4(x[8], x[ti], gW2[ti]);
                });
        }
        ASTAMP(13);
        {
            const float *q = lds + oH2 + 16 * bq * kLdh + bc;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r += 2) { s0 += q[r * kLdh]; s1 += q[(r + 1) * kLdh]; }
            gB2 += s0 + s1;
        }
        {
            ASTAMP(14);
            f4 d1[kRT];
#pragma unroll
            for (int mt = 0; mt < kRT; ++mt) d1[mt] = f4{0.f, 0.f, 0.f, 0.f};
            const float *a = lds + oH2 + c16 * kLdh + kq;            // A[i = row][k = o]
            mfma_stream<32, kPF, kRT>(
                [&](int ks, float (&x)[kRT]) {
#pragma unroll
                    for (int mt = 0; mt < kRT; ++mt) x[mt] = a[mt * 16 * kLdh + 4 * ks];
                },
                [&](int ks, const float (&x)[kRT]) {
#pragma unroll
                    for (int mt = 0; mt < kRT; ++mt) d1[mt] = mfma4(x[mt], w2b[ks], d1[mt]);
                });
            ASTAMP(15);
            __syncthreads();  // every wave has read H1 for dW2
            ASTAMP(8);
#pragma unroll
            for (int mt = 0; mt < kRT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    lds[oH1 + (mt * 16 + kq * 4 + r) * kLdh + col] = (live1 >> (mt * 4 + r) & 1u) ? d1[mt][r] : 0.f;
        }
        __syncthreads();
        ASTAMP(9);
        // (round 5) the last tile has added to dW2: its 64 KB of the slab go out now, under layer 1's gradients, instead of with
        // everything else behind the loop -- all 256 workgroups store at once there.  (Here, not right behind the dW2 stream: the dH1
        // accumulators are dead.)
        if (tile + gridDim.x >= n_tiles) {
            int lo = (16 * w + kq * 4) * kH + c16;
            asm volatile("" : "+v"(lo));   // (computed here: as a loop invariant the address pair is hoisted and spills two registers)
            float *slab_ = g.slabs + (size_t)blockIdx.x * (size_t)(oB3 + A) + oW2 + lo;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ti = 0; ti < 8; ++ti) __builtin_nontemporal_store(gW2[ti][r], slab_ + r * kH + 16 * ti);
        }
        // ---- P7: dW1 += dH1^T X ; db1 ----
        {
            const float *a = lds + oH1 + kq * kLdh + col;            // A[i = out o][k = row]
            const float *b = lds + oXc + kq * ld1 + c16;             // B[k = row][j = obs col]
            mfma_stream<kRows / 4, kPF, NJ + 1>(
                [&](int s, float (&x)[NJ + 1]) {
                    x[NJ] = a[4 * s * kLdh];
#pragma unroll
                    for (int ti = 0; ti < NJ; ++ti) x[ti] = b[4 * s * ld1 + 16 * ti];
                },
                [&](int, const float (&x)[NJ + 1]) {
#pragma unroll
                    for (int ti = 0; ti < NJ; ++ti) gW1[ti] = mfma4(x[NJ], x[ti], gW1[ti]);
                });
        }
        {
            const float *q = lds + oH1 + 16 * bq * kLdh + bc;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r += 2) { s0 += q[r * kLdh]; s1 += q[(r + 1) * kLdh]; }
            gB1 += s0 + s1;
        }
        __syncthreads();  // X / H1 / H2 / LG are free for the next tile
        ASTAMP(10);
    }

    // ---- the workgroup's gradient slab: written once, streamed (consumed once, by the reduction kernel) ----
    if (g.stamps && blockIdx.x == 0 && tid == 0) { g.stamps[202] = (long long)wall_clock64(); g.stamps[205] = (long long)__builtin_amdgcn_s_memtime(); }
    float *slab = g.slabs + (size_t)blockIdx.x * (size_t)(oB3 + A);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = 16 * w + kq * 4 + r;
#pragma unroll
        for (int ti = 0; ti < NJ; ++ti) {
            const int i = 16 * ti + c16;
            if (i < D) __builtin_nontemporal_store(gW1[ti][r], slab + oW1 + o * D + i);
        }
        if ((int64_t)blockIdx.x >= n_tiles) {   // (a workgroup with tiles stored dW2 behind its last tile's P6; one without stores its zeros)
#pragma unroll
            for (int ti = 0; ti < 8; ++ti) __builtin_nontemporal_store(gW2[ti][r], slab + oW2 + o * kH + 16 * ti + c16);
        }
        const int a = kq * 4 + r;
        if (a < A) __builtin_nontemporal_store(gW3[r], slab + oW3 + a * kH + 16 * w + c16);
    }
    {   // fold the bias partials (fixed order) through the now idle activation region
        float *sc = lds + ly.X;   // [4][128] db1 | [4][128] db2 | [16][16] db3
        sc[bq * 128 + bc] = gB1;
        sc[512 + bq * 128 + bc] = gB2;
        if (tid < 256) sc[1024 + tid] = gB3;
        __syncthreads();
        if (tid < 128) __builtin_nontemporal_store((sc[tid] + sc[128 + tid]) + (sc[256 + tid] + sc[384 + tid]), slab + oB1 + tid);
        else if (tid < 256) {
            const float *q = sc + 512 + (tid - 128);
            __builtin_nontemporal_store((q[0] + q[128]) + (q[256] + q[384]), slab + oB2 + tid - 128);
        } else if (tid < 256 + A) {
            float t = 0.f;
            for (int k = 0; k < 16; ++k) t += sc[1024 + 16 * k + (tid - 256)];
            __builtin_nontemporal_store(t, slab + oB3 + tid - 256);
        }
    }
    {   // loss statistics: the sample leaders (lane 0 of each 16-lane group) hold the terms; fixed order: wave sums, then waves
        __shared__ double s_stat[2][kThreads / 64];
        const double t_clip = hj == 0 ? s_tstat[hs][0] : 0.0, t_ent = hj == 0 ? s_tstat[hs][1] : 0.0;   // (written by this thread)
        const double c = wave_sum(t_clip), e = wave_sum(t_ent);
        int t_ = threadIdx.x;
        asm volatile("" : "+v"(t_));   // (lane and wave index re-derived here: kept from the prologue they cost the kernel's 257th register)
        if ((t_ & 63) == 0) { s_stat[0][t_ >> 6] = c; s_stat[1][t_ >> 6] = e; }
        __syncthreads();
        if (tid == 0) {
            double cc = 0.0, ee = 0.0;
            for (int k = 0; k < kThreads / 64; ++k) { cc += s_stat[0][k]; ee += s_stat[1][k]; }
            g.partial[4 * blockIdx.x + 0] = cc;
            g.partial[4 * blockIdx.x + 1] = 0.0;
            g.partial[4 * blockIdx.x + 2] = ee;
            g.partial[4 * blockIdx.x + 3] = 0.0;
            if (g.stamps && blockIdx.x == 0) g.stamps[203] = (long long)wall_clock64();
            if (g.stamps && blockIdx.x < 512) g.stamps[1025 + 2 * blockIdx.x] = (long long)wall_clock64();
        }
    }
}

}  // namespace

int tsm_actor_rows64_init(void) {
    static bool done = false;
    if (done) return TSM_OK;
    TSM_HIP(tsm_allow_max_lds(reinterpret_cast<const void *>(actor_rows64_kernel<1>)));
    TSM_HIP(tsm_allow_max_lds(reinterpret_cast<const void *>(actor_rows64_kernel<2>)));
    TSM_HIP(tsm_allow_max_lds(reinterpret_cast<const void *>(actor_rows64_kernel<3>)));
    TSM_HIP(tsm_allow_max_lds(reinterpret_cast<const void *>(actor_rows64_kernel<4>)));
    done = true;
    return TSM_OK;
}

int tsm_actor_rows64_launch(const TsmActorArgs &g, int n_blocks, hipStream_t st) {
    const Lay64 ly(g.D);
    const size_t shmem = (size_t)ly.total * sizeof(float);
    TSM_REQUIRE(shmem <= kTsmMaxLds, "actor gradient step (64-sample tiles): LDS layout of %zu bytes does not fit", shmem);
    TSM_REQUIRE(n_blocks >= 1 && n_blocks <= ceil_div(g.M, kRows), "actor gradient step (64-sample tiles): n_blocks = %d out of range",
                n_blocks);
    if (const int rc = tsm_actor_rows64_init(); rc != TSM_OK) return rc;
    switch (ly.nJ) {
        case 1: hipLaunchKernelGGL((actor_rows64_kernel<1>), dim3((unsigned)n_blocks), dim3(kThreads), shmem, st, g); break;
        case 2: hipLaunchKernelGGL((actor_rows64_kernel<2>), dim3((unsigned)n_blocks), dim3(kThreads), shmem, st, g); break;
        case 3: hipLaunchKernelGGL((actor_rows64_kernel<3>), dim3((unsigned)n_blocks), dim3(kThreads), shmem, st, g); break;
        default: hipLaunchKernelGGL((actor_rows64_kernel<4>), dim3((unsigned)n_blocks), dim3(kThreads), shmem, st, g); break;
    }
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}
