// critic_rows.hip -- V(row) of a 128-wide critic (in -> 128 -> 128 -> 1) for MANY rows in one launch, activations never
// leaving the CU: the critic passes of A2C's preprocessing,
//     v_s  = critic(batch.obs);  v_s_ = critic(batch.obs_next)       /root/reference/tianshou/algorithm/modelfree/a2c.py:121-127
// over the whole buffer (102 400 joint rows of 384 floats at BASELINE configs[2]), and the critic half of a rollout step.
// It replaces three tsm_mlp_forward GEMM launches whose 128-wide activations went through HBM twice.
//
// gfx950 mapping.  One persistent 512-thread workgroup (8 waves, 2 per SIMD) per CU walks 32-row tiles.
//   * Wave w owns output columns [16 w, 16 w + 16) of both hidden layers.  Its fragment of the first layer's weights
//     (16 columns x K1) is what `v_mfma_f32_16x16x4_f32` wants as the B operand -- one float per lane and k -- so it
//     is loaded ONCE, straight from global memory into registers (K1 / 4 registers per lane: 96 at K1 = 384), and
//     stays there for every tile of the launch: the 192 KB first-layer matrix, which does not fit a CU's LDS, never
//     passes through LDS at all and layer 1 runs without a single barrier or staging copy.
//   * Lane (c16, kq) keeps W1[col][16 j + 4 kq + i], i = 0..3, as one 16-B load; the A operand of the same k comes
//     out of the tile's observation block in LDS with one ds_read_b128 per 16-row half.  MFMA step (j, i) therefore
//     sums k = 16 j + i, 16 j + 4 + i, 16 j + 8 + i, 16 j + 12 + i -- a fixed permutation of the k order, the same for
//     every row and every launch (values differ from the k-ordered GEMM by f32 rounding only; every V this engine
//     compares with another V comes from this kernel).
//   * The observation tile (32 x K1) lives in LDS with its 16-B chunks XOR-swizzled by the row (chunk ^ (row & 15)
//     inside groups of 16 chunks; row pitch a multiple of 64 floats): ds_read_b128 of 16 rows x 4 k-groups and the
//     ds_write_b128 of the staging copy are both bank-conflict free (checked exhaustively in tools/lds_banks.py).  The next
//     tile's rows are fetched into registers while this tile is multiplied.
//   * W2 (64 KB) is LDS-resident for the whole launch; layer 3 (128 -> 1) is folded into layer 2's epilogue: every
//     lane multiplies its relu(h2) by w3[col], 16-lane DPP row sums, then the 8 waves' partial sums in wave order.
//     Two barriers per tile.
// Algorithmic HBM traffic: 4 K1 B read + 4 B written per row.
#include "critic_rows_dev.h"
#include <stdlib.h>

namespace {

struct FwdLay {  // LDS layout in floats
    int ldx, W2, X, H1, B1, B2, W3, RED, RID, total;
    __host__ __device__ explicit FwdLay(int KJ) {
        ldx = ((16 * KJ + 63) / 64) * 64;
        int o = 0;
        W2 = o; o += kH * kLdh;
        X = o; o += kRows * ldx;
        H1 = o; o += kRows * kLdh;
        B1 = o; o += kH;
        B2 = o; o += kH;
        W3 = o; o += kH;
        RED = o; o += 8 * kRows;
        RID = o; o += 2 * kRows;  // row ids of the tile being fetched (int64)
        total = o;
    }
    __device__ void launder() {   // (critic_rows_dev.h: opaque_s)
        W2 = opaque_s(W2); X = opaque_s(X); H1 = opaque_s(H1); B1 = opaque_s(B1); B2 = opaque_s(B2); W3 = opaque_s(W3);
        RED = opaque_s(RED); RID = opaque_s(RID);
    }
};

struct FwdArgs {
    const float *P;        // critic parameters: w0[H][K1] b0[H] w1[H][H] b1[H] w2[n_out][H] b2[n_out]
    int n_out;             // outputs; the value of a row is their mean (CTDEPolicy.learn, ctde.py:154-157), n_out = 1: itself
    const float *obs;      // rows [n][K1]
    const int64_t *rows;   // row ids (nullable: first_row + i)
    int64_t first_row, Mr;
    int K1;
    float *out;            // [Mr]
    const int32_t *run_if; // nullable device flag: 0 -> the launch is a no-op
};

// VEC: K1 % 4 == 0 (rows and W1 rows are whole 16-B chunks).  All loads go to clamped, always-valid addresses and
// are zeroed by a select afterwards: no divergent branches, every load of a batch in flight at once.
template <int KJ, bool VEC>
__global__ __launch_bounds__(kThreads) void critic_rows_forward_kernel(FwdArgs g) {
    if (g.run_if && *g.run_if == 0) return;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    FwdLay ly(KJ);
    constexpr int ldx = ((16 * KJ + 63) / 64) * 64;
    ly.launder();
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c16 = lane & 15, kq = lane >> 4;
    const int K1 = g.K1;
    const int64_t n_tiles = (g.Mr + kRows - 1) / kRows;
    const int oB1 = kH * K1, oW2 = oB1 + kH, oB2 = oW2 + kH * kH, oW3 = oB2 + kH, oB3 = oW3 + g.n_out * kH;
    const int col = 16 * w + c16;

    // ---- this wave's fragment of W1, once, into registers ----
    f4 w1f[KJ];
#pragma unroll
    for (int j = 0; j < KJ; ++j) w1f[j] = load_w1_frag<VEC>(g.P + (size_t)col * K1, 16 * j + 4 * kq, K1);
    // ---- W2, biases, w3 into LDS; X pads zeroed (staging only ever rewrites the real chunks) ----
    stage_w2_rows(lds + ly.W2, g.P + oW2);
    // the mean over the outputs of a linear layer is a linear layer with the mean weights: w3[col] = mean_j W3[j][col]
    const int n_out = g.n_out;
    if (tid < kH) {
        lds[ly.B1 + tid] = g.P[oB1 + tid];
        lds[ly.B2 + tid] = g.P[oB2 + tid];
        float s3 = 0.f;
        for (int j = 0; j < n_out; ++j) s3 += g.P[oW3 + j * kH + tid];
        lds[ly.W3 + tid] = n_out == 1 ? s3 : s3 / (float)n_out;
    }
    for (int e = tid; e < kRows * ldx; e += kThreads) lds[ly.X + e] = 0.f;
    float b3 = 0.f;
    for (int j = 0; j < n_out; ++j) b3 += g.P[oB3 + j];
    if (n_out > 1) b3 /= (float)n_out;

    // ---- staging of a tile: thread -> chunks q = tid + 512 u of the 32 x (4 KJ) chunk grid ----
    constexpr int CPR = 4 * KJ;                                       // chunks per (padded) row
    constexpr int NX = (kRows * CPR + kThreads - 1) / kThreads;       // chunks per thread
    f4 xr[NX];
    // Row ids go through LDS (RID): thread r < 32 fetches the id of row r of the tile AFTER the one whose data is
    // being fetched, so that id -> data is never two dependent global round trips inside one phase.
    int64_t *rid = reinterpret_cast<int64_t *>(lds + ly.RID);
    auto load_id = [&](int64_t tile_) -> int64_t {   // threads < 32: id of row tid of tile_ (clamped to a valid row)
        const int64_t i = tile_ * kRows + (tid & (kRows - 1));
        const int64_t ic = i < g.Mr ? i : g.Mr - 1;  // (Mr >= 1)
        return g.rows ? g.rows[ic] : g.first_row + ic;
    };
    auto fetch_tile = [&](int64_t tile_) {           // data of tile_ whose row ids are in RID
        const int tq = opaque_v(tid);                // (chunk coordinates recomputed per call, not kept across the tile)
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int q = tq + kThreads * u, r = (q / CPR) & (kRows - 1), c = q % CPR;
            const int64_t i = tile_ * kRows + r;
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
    };
    auto commit_tile = [&]() {
        const int tq = opaque_v(tid);
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int q = tq + kThreads * u, r = q / CPR, c = q - r * CPR;
            if (q < kRows * CPR) *reinterpret_cast<f4 *>(lds + ly.X + xs_off(r, c, ldx)) = xr[u];
        }
    };

    int64_t tile = blockIdx.x;
    const int64_t gs = gridDim.x;
    int64_t my_id = 0;
    if (tid < kRows) rid[tid] = load_id(tile);
    __syncthreads();  // ids of the first tile; pads zeroed before the first commit
    fetch_tile(tile);
    if (tid < kRows) my_id = load_id(tile + gs);
    commit_tile();
    __syncthreads();  // (every thread has read RID)
    if (tid < kRows) { rid[tid] = my_id; my_id = load_id(tile + 2 * gs); }
    __syncthreads();
    fetch_tile(tile + gs);
    __syncthreads();  // (RID is rewritten at the top of the loop)

    for (; tile < n_tiles; tile += gs) {
        // ids of tile + 2 gs (loaded during the previous tile) -> RID, published by barrier (A); the ids of tile + 3 gs fly
        if (tid < kRows) { rid[tid] = my_id; my_id = load_id(tile + 3 * gs); }
        // ---- L1: H1 = relu(X W1^T + b1); W1 from registers, X by ds_read_b128 ----
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
                for (int r = 0; r < 4; ++r) lds[ly.H1 + (mt * 16 + kq * 4 + r) * kLdh + col] = fmaxf(acc[mt][r] + bb, 0.f);
        }
        __syncthreads();  // (A) H1 complete; every wave is done with X
        commit_tile();                          // the next tile's rows (in registers since the previous tile) ...
        fetch_tile(tile + 2 * gs);              // ... and the loads of the one after it fly during layer 2
        // ---- L2 + L3: v = sum_col relu(H1 W2^T + b2)[col] * w3[col] ----
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
            const float bb = lds[ly.B2 + col], w3c = lds[ly.W3 + col];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float s = row16_sum(fmaxf(acc[mt][r] + bb, 0.f) * w3c);  // over this wave's 16 columns
                    if (c16 == 0) lds[ly.RED + w * kRows + mt * 16 + kq * 4 + r] = s;
                }
        }
        __syncthreads();  // (B) partial sums of the 8 waves; X of the next tile committed
        if (tid < kRows) {
            float v = lds[ly.RED + tid];
#pragma unroll
            for (int ww = 1; ww < 8; ++ww) v += lds[ly.RED + ww * kRows + tid];
            const int64_t i = tile * kRows + tid;
            if (i < g.Mr) g.out[i] = v + b3;
        }
    }
}

template <int KJ, bool VEC>
int launch_forward_v(const FwdArgs &g, int grid, hipStream_t st) {
    const FwdLay ly(KJ);
    const size_t shmem = (size_t)ly.total * sizeof(float);
    TSM_REQUIRE(shmem <= kMaxLds, "tsm_critic_rows_forward: LDS layout of %zu bytes does not fit", shmem);
    static bool attr_set = false;  // (set before any capture: tsm_critic_rows_init)
    if (!attr_set) {
        TSM_HIP(tsm_allow_max_lds(reinterpret_cast<const void *>(critic_rows_forward_kernel<KJ, VEC>)));
        attr_set = true;
    }
    if (grid > 0) {
        hipLaunchKernelGGL((critic_rows_forward_kernel<KJ, VEC>), dim3((unsigned)grid), dim3(kThreads), shmem, st, g);
        TSM_LAUNCH_CHECK();
    }
    return TSM_OK;
}

template <int KJ>
int launch_forward(const FwdArgs &g, int grid, hipStream_t st) {
    // (K1 == 0 only from tsm_critic_rows_init, which sets the attributes of both forms)
    if (g.K1 == 0) {
        const int rc = launch_forward_v<KJ, true>(g, 0, st);
        return rc != TSM_OK ? rc : launch_forward_v<KJ, false>(g, 0, st);
    }
    return (g.K1 & 3) == 0 ? launch_forward_v<KJ, true>(g, grid, st) : launch_forward_v<KJ, false>(g, grid, st);
}

// ================================================================================================================
// EXPERIMENTAL, opt-in (TSM_SPLIT_BF16=1; never the default path): layer 1 of the same forward on the bf16 matrix pipe with
// f32-level accuracy.  gfx950 issues v_mfma_f32_16x16x32_bf16 (16 384 flop) every 16 cycles per SIMD, v_mfma_f32_16x16x4_f32
// (2 048 flop) every 32: 16x the rate.  An f32 value is the sum of three bf16 values, x = x0 + x1 + x2 (x0 = bf16(x),
// x1 = bf16(x - x0), x2 = bf16(x - x0 - x1): 24 mantissa bits), so a product is the sum of nine bf16 x bf16 products, each
// EXACT in the f32 accumulator; the six largest (x0 w0, x0 w1, x1 w0, x1 w1, x0 w2, x2 w0) leave a relative error of
// ~2^-22 per product, the size of f32 rounding itself.  Six 16-cycle MFMAs per 32 k instead of eight 32-cycle ones: 2.67x.
// The observation tile sits in LDS as three bf16 planes (same XOR swizzle, a plane row = 16 KJ2 dwords), split once when the
// tile is committed; W1's fragments are split once per launch into registers (3 x 4 VGPRs per 32 k).  Layers 2 / 3 as above.
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split3(float x, __bf16 &h, __bf16 &m, __bf16 &l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

struct FwdLayB {  // LDS layout in floats (a bf16 plane row: 16 KJ2 floats = 32 KJ2 bf16)
    int ldx, W2, X, H1, B1, B2, W3, RED, RID, total;
    __host__ __device__ explicit FwdLayB(int KJ2) {
        ldx = ((16 * KJ2 + 63) / 64) * 64;
        int o = 0;
        W2 = o; o += kH * kLdh;
        X = o; o += 3 * kRows * ldx;      // planes x0 | x1 | x2
        H1 = o; o += kRows * kLdh;
        B1 = o; o += kH;
        B2 = o; o += kH;
        W3 = o; o += kH;
        RED = o; o += 8 * kRows;
        RID = o; o += 2 * kRows;
        total = o;
    }
    __device__ void launder() {
        W2 = opaque_s(W2); X = opaque_s(X); H1 = opaque_s(H1); B1 = opaque_s(B1); B2 = opaque_s(B2); W3 = opaque_s(W3);
        RED = opaque_s(RED); RID = opaque_s(RID);
    }
};

template <int KJ2>   // K1 <= 32 KJ2, K1 % 4 == 0
__global__ __launch_bounds__(kThreads) void critic_rows_forward_bf16x6_kernel(FwdArgs g) {
    if (g.run_if && *g.run_if == 0) return;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    FwdLayB ly(KJ2);
    constexpr int ldx = ((16 * KJ2 + 63) / 64) * 64;
    constexpr int plane = kRows * ldx;
    ly.launder();
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c16 = lane & 15, kq = lane >> 4;
    const int K1 = g.K1;
    const int64_t n_tiles = (g.Mr + kRows - 1) / kRows;
    const int oB1 = kH * K1, oW2 = oB1 + kH, oB2 = oW2 + kH * kH, oW3 = oB2 + kH, oB3 = oW3 + g.n_out * kH;
    const int col = 16 * w + c16;

    // ---- this wave's fragment of W1: lane (c16, kq) keeps W1[col][32 J + 8 kq + i], i < 8, as three bf16 x 8 ----
    bf8 w0[KJ2], w1[KJ2], w2[KJ2];
#pragma unroll
    for (int J = 0; J < KJ2; ++J) {
        const f4 lo4 = load_w1_frag<true>(g.P + (size_t)col * K1, 32 * J + 8 * kq, K1);
        const f4 hi4 = load_w1_frag<true>(g.P + (size_t)col * K1, 32 * J + 8 * kq + 4, K1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __bf16 h, m, l;
            split3(lo4[i], h, m, l); w0[J][i] = h; w1[J][i] = m; w2[J][i] = l;
            split3(hi4[i], h, m, l); w0[J][4 + i] = h; w1[J][4 + i] = m; w2[J][4 + i] = l;
        }
    }
    stage_w2_rows(lds + ly.W2, g.P + oW2);
    const int n_out = g.n_out;
    if (tid < kH) {
        lds[ly.B1 + tid] = g.P[oB1 + tid];
        lds[ly.B2 + tid] = g.P[oB2 + tid];
        float s3 = 0.f;
        for (int j = 0; j < n_out; ++j) s3 += g.P[oW3 + j * kH + tid];
        lds[ly.W3 + tid] = n_out == 1 ? s3 : s3 / (float)n_out;
    }
    for (int e = tid; e < 3 * plane; e += kThreads) lds[ly.X + e] = 0.f;
    float b3 = 0.f;
    for (int j = 0; j < n_out; ++j) b3 += g.P[oB3 + j];
    if (n_out > 1) b3 /= (float)n_out;

    // ---- staging: thread -> f32 chunks (4 floats) q = tid + 512 u of the 32 x (8 KJ2) chunk grid ----
    constexpr int CPR = 8 * KJ2;
    constexpr int NX = (kRows * CPR + kThreads - 1) / kThreads;
    f4 xr[NX];
    int64_t *rid = reinterpret_cast<int64_t *>(lds + ly.RID);
    auto load_id = [&](int64_t tile_) -> int64_t {
        const int64_t i = tile_ * kRows + (tid & (kRows - 1));
        const int64_t ic = i < g.Mr ? i : g.Mr - 1;
        return g.rows ? g.rows[ic] : g.first_row + ic;
    };
    auto fetch_tile = [&](int64_t tile_) {
        const int tq = opaque_v(tid);
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int q = tq + kThreads * u, r = (q / CPR) & (kRows - 1), c = q % CPR;
            const int64_t i = tile_ * kRows + r;
            const bool ok = q < kRows * CPR && i < g.Mr && 4 * c < K1;
            const int64_t row = rid[r];
            const int cc = 4 * c < K1 ? 4 * c : K1 - 4;
            const float4 v = *reinterpret_cast<const float4 *>(g.obs + row * K1 + cc);
            xr[u] = ok ? f4{v.x, v.y, v.z, v.w} : f4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto commit_tile = [&]() {   // split into the three planes: f32 chunk c -> half (c & 1) of bf16 chunk c >> 1
        const int tq = opaque_v(tid);
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int q = tq + kThreads * u, r = q / CPR, c = q - r * CPR;
            if (q < kRows * CPR) {
                bf2 p[3][2];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    __bf16 h, m, l;
                    split3(xr[u][e], h, m, l);
                    p[0][e >> 1][e & 1] = h; p[1][e >> 1][e & 1] = m; p[2][e >> 1][e & 1] = l;
                }
                float *dst = lds + ly.X + xs_off(r, c >> 1, ldx) + 2 * (c & 1);
#pragma unroll
                for (int s_ = 0; s_ < 3; ++s_) {
                    *reinterpret_cast<bf2 *>(dst + s_ * plane) = p[s_][0];
                    *reinterpret_cast<bf2 *>(dst + s_ * plane + 1) = p[s_][1];
                }
            }
        }
    };

    int64_t tile = blockIdx.x;
    const int64_t gs = gridDim.x;
    int64_t my_id = 0;
    if (tid < kRows) rid[tid] = load_id(tile);
    __syncthreads();
    fetch_tile(tile);
    if (tid < kRows) my_id = load_id(tile + gs);
    commit_tile();
    __syncthreads();
    if (tid < kRows) { rid[tid] = my_id; my_id = load_id(tile + 2 * gs); }
    __syncthreads();
    fetch_tile(tile + gs);
    __syncthreads();

    for (; tile < n_tiles; tile += gs) {
        if (tid < kRows) { rid[tid] = my_id; my_id = load_id(tile + 3 * gs); }
        // ---- L1 on the bf16 pipe: six products per 32 k and row half, smallest terms first ----
        {
            f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
            const float *xa[4] = {lds + ly.X + xa_base(c16, kq, ldx, 0), lds + ly.X + xa_base(c16, kq, ldx, 1),
                                  lds + ly.X + xa_base(c16, kq, ldx, 2), lds + ly.X + xa_base(c16, kq, ldx, 3)};
#pragma unroll
            for (int J = 0; J < KJ2; ++J) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float *pa = xa[J & 3] + 64 * (J >> 2) + 16 * h * ldx;
                    const bf8 a0 = *reinterpret_cast<const bf8 *>(pa);
                    const bf8 a1 = *reinterpret_cast<const bf8 *>(pa + plane);
                    const bf8 a2 = *reinterpret_cast<const bf8 *>(pa + 2 * plane);
                    f4 c = acc[h];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, w0[J], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, w2[J], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, w1[J], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, w0[J], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, w1[J], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, w0[J], c, 0, 0, 0);
                    acc[h] = c;
                }
            }
            const float bb = lds[ly.B1 + col];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) lds[ly.H1 + (mt * 16 + kq * 4 + r) * kLdh + col] = fmaxf(acc[mt][r] + bb, 0.f);
        }
        __syncthreads();
        commit_tile();
        fetch_tile(tile + 2 * gs);
        // ---- L2 + L3 (f32 matrix pipe, as in critic_rows_forward_kernel) ----
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
            const float bb = lds[ly.B2 + col], w3c = lds[ly.W3 + col];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float s = row16_sum(fmaxf(acc[mt][r] + bb, 0.f) * w3c);
                    if (c16 == 0) lds[ly.RED + w * kRows + mt * 16 + kq * 4 + r] = s;
                }
        }
        __syncthreads();
        if (tid < kRows) {
            float v = lds[ly.RED + tid];
#pragma unroll
            for (int ww = 1; ww < 8; ++ww) v += lds[ly.RED + ww * kRows + tid];
            const int64_t i = tile * kRows + tid;
            if (i < g.Mr) g.out[i] = v + b3;
        }
    }
}

template <int KJ2>
int launch_forward_bf16(const FwdArgs &g, int grid, hipStream_t st) {
    const FwdLayB ly(KJ2);
    const size_t shmem = (size_t)ly.total * sizeof(float);
    TSM_REQUIRE(shmem <= kMaxLds, "tsm_critic_rows_forward (bf16 x 6): LDS layout of %zu bytes does not fit", shmem);
    static bool attr_set = false;
    if (!attr_set) {
        TSM_HIP(tsm_allow_max_lds(reinterpret_cast<const void *>(critic_rows_forward_bf16x6_kernel<KJ2>)));
        attr_set = true;
    }
    if (grid > 0) {
        hipLaunchKernelGGL((critic_rows_forward_bf16x6_kernel<KJ2>), dim3((unsigned)grid), dim3(kThreads), shmem, st, g);
        TSM_LAUNCH_CHECK();
    }
    return TSM_OK;
}

// experimental path on?  (option "split_bf16": tsm_kernel_option_set, default from TSM_SPLIT_BF16; only widths with an
// instantiation: K1 % 4 == 0, K1 <= 384)
int split_bf16_kj2(int K1) {
    if (!tsm_opt(TSM_OPT_SPLIT_BF16) || (K1 & 3)) return 0;
    static const int inst[] = {2, 3, 6, 12};
    for (int k : inst)
        if (32 * k >= K1) return k;
    return 0;
}

int dispatch_forward_bf16(int kj2, const FwdArgs &g, int grid, hipStream_t st) {
    switch (kj2) {
        case 2: return launch_forward_bf16<2>(g, grid, st);
        case 3: return launch_forward_bf16<3>(g, grid, st);
        case 6: return launch_forward_bf16<6>(g, grid, st);
        default: return launch_forward_bf16<12>(g, grid, st);
    }
}

int dispatch_forward(int kj, const FwdArgs &g, int grid, hipStream_t st) {
    switch (kj) {
        case 1: return launch_forward<1>(g, grid, st);
        case 2: return launch_forward<2>(g, grid, st);
        case 3: return launch_forward<3>(g, grid, st);
        case 4: return launch_forward<4>(g, grid, st);
        case 6: return launch_forward<6>(g, grid, st);
        case 8: return launch_forward<8>(g, grid, st);
        case 12: return launch_forward<12>(g, grid, st);
        case 16: return launch_forward<16>(g, grid, st);
        case 24: return launch_forward<24>(g, grid, st);
        default: break;
    }
    tsm_set_error("tsm_critic_rows_forward: unsupported input width");
    return TSM_ERR_INVALID;
}

}  // namespace

TSM_EXPORT int tsm_critic_rows_forward_supported(int32_t in_dim, int32_t hidden) {
    return hidden == kH && in_dim >= 1 && pick_kj(in_dim) != 0;
}

// One-time function attributes (dynamic LDS size) for the instantiation that serves `in_dim`: call it outside any
// stream capture (the first launch would otherwise set them inside one).
int tsm_critic_train_init(int32_t in_dim);  // critic_train.hip

TSM_EXPORT int tsm_critic_rows_init(int32_t in_dim, int32_t hidden) {
    TSM_REQUIRE(tsm_critic_rows_forward_supported(in_dim, hidden), "tsm_critic_rows_init: unsupported critic %d -> %d", in_dim, hidden);
    FwdArgs g{};
    int rc = dispatch_forward(pick_kj(in_dim), g, 0, nullptr);
    if (rc == TSM_OK && split_bf16_kj2(in_dim)) rc = dispatch_forward_bf16(split_bf16_kj2(in_dim), g, 0, nullptr);
    return rc != TSM_OK ? rc : tsm_critic_train_init(in_dim);
}

TSM_EXPORT int tsm_critic_rows_forward(const float *critic_params, int32_t in_dim, int32_t hidden, int32_t n_out,
                                       const float *obs_rows, const int64_t *rows, int64_t first_row, int64_t Mr,
                                       const int32_t *run_if, float *values_out, void *stream) {
    TSM_REQUIRE(tsm_critic_rows_forward_supported(in_dim, hidden),
                "tsm_critic_rows_forward supports hidden == 128, in_dim <= 384 (got %d / %d)", hidden, in_dim);
    TSM_REQUIRE(Mr >= 0 && n_out >= 1 && n_out <= 16, "tsm_critic_rows_forward: bad sizes (Mr %lld, n_out %d)", (long long)Mr, n_out);
    if (Mr == 0) return TSM_OK;
    TSM_REQUIRE(critic_params && obs_rows && values_out, "tsm_critic_rows_forward: null pointer");
    FwdArgs g{};
    g.P = critic_params; g.obs = obs_rows; g.rows = rows; g.first_row = first_row; g.Mr = Mr; g.K1 = in_dim;
    g.out = values_out; g.run_if = run_if; g.n_out = n_out;
    const int64_t tiles = ceil_div(Mr, kRows);
    const int cu = n_cu_dev();
    if (const int kj2 = split_bf16_kj2(in_dim))   // experimental, opt-in (TSM_SPLIT_BF16=1)
        return dispatch_forward_bf16(kj2, g, (int)(tiles < cu ? tiles : cu), tsm_stream(stream));
    return dispatch_forward(pick_kj(in_dim), g, (int)(tiles < cu ? tiles : cu), tsm_stream(stream));
}
