// critic_dw1.hip -- the first layer's weight gradient of the one-launch critic step as a split-K pass:
//     dW1[o][k] = sum_i dH1[i][o] * X[row(i)][k]          (o < 128, k < K1, i over the minibatch rows)
// i.e. `w0.grad` of `loss.backward()` for the critic of PPO / CTDEPolicy.learn (modelfree/ppo.py:209-212,
// multiagent/ctde.py:188-190), given dH1 = d loss / d (layer-1 pre-activation) published by csrc/critic_train.hip.
//
// Why a pass of its own: accumulated inside the per-tile kernel, every workgroup wrote a full 128 x K1 slab for a rank-32
// update (196 KB x 256 workgroups = 50 MB per step at K1 = 384, read back by the optimizer).  Here workgroup (cb, rc) owns
// the 128 x 48 (or 128 x 96: dw1_nt) block of columns and sums over the row chunk rc in registers (3 / 6 accumulator tiles per wave:
// wave w owns outputs [16 w, 16 w + 16)), so the reduction over ROWS happens on chip and only n_chunk partial slabs (32 at
// 8192 rows, K1 = 384) go to memory.  Sub-chunks of 64 rows are staged through LDS (dH1 [64][128], X [64][48]; row pitches =
// 16 mod 32 floats: the [k = row][lane = column] operand reads are bank-conflict free), the next sub-chunk's rows are in
// flight in registers meanwhile, their ids one sub-chunk further ahead.
// Round 4 (an experiment build of tools/critic_pair_time.py, 8192 rows of 384, launch alone back to back): 4 x 96 columns x 64 chunks 13.4 us; 8 x 48
// columns x 32 chunks 13.6 us; the same x 64 chunks (two workgroups per CU) 12.9 us; 4 x 96 x 128 chunks 13.5 us -- the launch
// is a latency chain (ids -> rows -> LDS -> 5.6 us of MFMA -> slab) that no split shortens, so the form with the FEWEST slab
// bytes is the one kept: half the writes here and half the optimizer's reads of them (12.6 -> 6.3 MB each).  32-row sub-chunks
// (33 KB of LDS: four workgroups per CU, more room for the side reductions below) were measured too: 17.2 vs 14.6 us alone,
// 22.5 vs 20.3 us in situ -- twice the barriers per row cost more than the extra residency gives.
#include "critic_rows_dev.h"
#include "adam_dev.h"

namespace {

constexpr int kSub = 64;       // rows per sub-chunk
constexpr int kLdA = 144;      // dH1 rows in LDS (128 + 16): pitch = 16 mod 32
// NT = MFMA column tiles per workgroup: 6 (96 columns) or 3 (48 columns); X rows in LDS: 112 / 80 floats (pitch = 16 mod 32)

struct Dw1Args {
    const float *dh1;        // [Mr][128]
    const float *obs;        // rows [n][K1]
    const int64_t *rows;     // nullable
    int64_t first_row, Mr, tm_T, tm_E;   // row addressing as in critic_train.hip
    int K1;
    int64_t RC;              // rows per chunk (a multiple of 64)
    int ncb, n_chunk;        // column blocks, row chunks
    float *slabs;            // [n_chunk][128 K1]
    // second product of the same launch (nullable): dW2[o][k] = sum_i dH2[i][o] H1[i][k] over the same row chunks, from the
    // activations kernel A published in MINIBATCH order (no row-id indirection) -- column blocks ncb1 .. ncb - 1
    const float *dh2, *h1;   // [Mr][128] each
    float *slabs2;           // [n_chunk][128 x 128]
    int ncb1;                // column blocks of the first product
    // side reductions (see below): workgroups >= n_main sum gradient slabs of OTHER parameter segments into one row each
    struct Side { const float *slabs; int64_t n, stride; int32_t n_slab, first_blk; float *out; } side[2];
    int n_side, n_main;
};

// Side reductions.  The launch is a latency chain (row ids -> rows -> LDS -> 5.6 us of MFMA -> slab) that leaves most of the
// memory system idle, while the optimizer launch behind it is bandwidth-bound on reading every kernel's gradient slabs (54 MB
// at the C3 step: 14 us).  Workgroups behind the main grid therefore sum slab sets that are already complete -- the actor
// step's and the critic's small-gradient slabs -- down to ONE row each while the main workgroups wait on their loads: the
// optimizer then reads 6 MB instead of 48.  128 parameters per workgroup as two independent halves of 256 threads; per
// parameter the additions are slab_sum_block's in the same order, and a one-row "slab set" passes through the optimizer's own
// sum unchanged (x + 0 + 0 + 0), so the gradient has the same bits wherever the sum is formed.
__device__ __forceinline__ void dw1_side_reduce(const Dw1Args &g, int b, float *lds) {
    const int k = (g.n_side > 1 && b >= g.side[1].first_blk) ? 1 : 0;
    const Dw1Args::Side sr = g.side[k];
    // 64 parameters per workgroup on its first four waves (the other four have nothing to load): a workgroup's share of HBM bandwidth
    // is ~17 KB / us when every CU streams, so 64 KB per workgroup keeps two to three of them inside the main workgroups' 13 us
    // (128 parameters per workgroup: 20.1 us for the launch; this form: measured below)
    const int tid = threadIdx.x, half = 0, lane = tid & 63, sl = (tid >> 6) & 3;
    const bool active = tid < 256;   // (waves 4..7 only meet the barrier)
    const int64_t i = active ? (int64_t)(b - sr.first_blk) * kCols + lane : sr.n;
    // (slab_lane_sum's additions, sequentially over slabs sl, sl + 4, ..., with 32 loads in flight instead of 16: these
    // workgroups are few and latency-bound, the optimizer's own loop is bandwidth-bound)
    float acc = 0.f;
    if (i < sr.n) {
        const float *__restrict__ base = sr.slabs + i;
        int s = sl;
#pragma unroll 1
        for (; s + 124 < sr.n_slab; s += 128) {
            float t[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) t[u] = base[(int64_t)(s + 4 * u) * sr.stride];
#pragma unroll
            for (int u = 0; u < 32; ++u) acc += t[u];
        }
        for (; s + 28 < sr.n_slab; s += 32) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = base[(int64_t)(s + 4 * u) * sr.stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += t[u];
        }
        for (; s < sr.n_slab; s += 4) acc += base[(int64_t)s * sr.stride];
    }
    float *sm = lds + half * 256;
    if (active) sm[sl * 64 + lane] = acc;
    __syncthreads();
    if (active && sl == 0 && i < sr.n) sr.out[i] = sm[lane] + sm[64 + lane] + sm[128 + lane] + sm[192 + lane];
}

__device__ __forceinline__ int64_t dw1_row_of(const Dw1Args &g, int64_t i) {
    const int64_t ic = i < g.Mr ? i : g.Mr - 1;
    if (g.rows) return g.rows[ic];
    if (g.tm_T > 0) {   // env-major view of a time-major store (32-bit arithmetic, Mr < 2^31: a 64-bit division is ~100 instructions)
        const int t_ = (int)g.tm_T, ii = (int)ic;
        return (int64_t)(ii % t_) * g.tm_E + ii / t_;
    }
    return g.first_row + ic;
}

template <bool VEC, int NT>
__global__ __launch_bounds__(kThreads) void critic_dw1_kernel(Dw1Args g) {
    constexpr int kCB = 16 * NT, kLdB = NT == 6 ? 112 : 80, CPRB = 4 * NT;   // CPRB: 16-B chunks per X row of the block
    constexpr int NB = (kSub * CPRB + kThreads - 1) / kThreads;     // X chunks per thread (3 at NT = 6, 2 at NT = 3)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *sA = lds, *sB = lds + kSub * kLdA;
    int64_t *rid = reinterpret_cast<int64_t *>(lds + kSub * (kLdA + 112));   // (one LDS size for both forms: 112 = the wider X pitch)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c16 = lane & 15, kq = lane >> 4;
    // XCD-aware block -> (column block, row chunk) map: workgroups are dealt round-robin over the 8 XCDs (each with an L2 of its
    // own), and the ncb workgroups of one row chunk read the SAME dH1 rows -- give them ids that differ by multiples of 8, so
    // that they share an L2 and the chunk leaves memory once instead of ncb times (speed / traffic only: any placement is correct)
    if ((int)blockIdx.x >= g.n_main) {   // (uniform per workgroup)
        dw1_side_reduce(g, (int)blockIdx.x - g.n_main, lds);
        return;
    }
    const int ncb = g.ncb, bid = blockIdx.x;
    const int rc_lo = bid & 7, t_ = bid >> 3, cb = t_ % ncb, rc = (t_ / ncb) * 8 + rc_lo;
    if (rc >= g.n_chunk) return;   // (the grid is padded to whole groups of 8 chunks)
    const bool second = cb >= g.ncb1;                       // (uniform) this workgroup belongs to the dW2 product
    const float *__restrict__ Asrc = second ? g.dh2 : g.dh1;
    const float *__restrict__ Bsrc = second ? g.h1 : g.obs;
    const int K1 = second ? kH : g.K1, c0 = kCB * (second ? cb - g.ncb1 : cb);   // K1: row length of the B operand / output
    const int64_t i_lo = (int64_t)rc * g.RC, i_hi = i_lo + g.RC < g.Mr ? i_lo + g.RC : g.Mr;
    const int n_sub = (int)((i_hi - i_lo + kSub - 1) / kSub);

    f4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};

    // staging: dH1 sub-chunk = 64 rows x 32 chunks of 16 B (4 per thread), X block = 64 rows x 4 NT chunks (NB per thread)
    f4 ra[4], rb[NB];
    auto fetch = [&](int s) {
        const int64_t i0 = i_lo + (int64_t)s * kSub;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = tid + kThreads * u, r = q >> 5, c = q & 31;
            const int64_t i = i0 + r;
            const int64_t ic = i < i_hi ? i : i_hi - 1;
            const float4 v = *reinterpret_cast<const float4 *>(Asrc + ic * kH + 4 * c);
            ra[u] = i < i_hi ? f4{v.x, v.y, v.z, v.w} : f4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int q = tid + kThreads * u, r = (q / CPRB) & (kSub - 1), c = q % CPRB;
            const int64_t i = i0 + r;
            const int k = c0 + 4 * c;
            const bool ok = q < kSub * CPRB && i < i_hi && k < K1;
            const int64_t row = rid[r];
            if constexpr (VEC) {
                const int kc = k < K1 ? k : K1 - 4;
                const float4 v = *reinterpret_cast<const float4 *>(Bsrc + row * K1 + kc);
                rb[u] = ok ? f4{v.x, v.y, v.z, v.w} : f4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = Bsrc[row * K1 + (k + e < K1 ? k + e : K1 - 1)];
                    rb[u][e] = (ok && k + e < K1) ? v : 0.f;
                }
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = tid + kThreads * u, r = q >> 5, c = q & 31;
            *reinterpret_cast<f4 *>(sA + r * kLdA + 4 * c) = ra[u];
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int q = tid + kThreads * u, r = q / CPRB, c = q - r * CPRB;
            if (q < kSub * CPRB) *reinterpret_cast<f4 *>(sB + r * kLdB + 4 * c) = rb[u];
        }
    };

    // B-operand row of minibatch row i: the observation row it names (first product), or i itself (H1 is in minibatch order)
    auto row_id = [&](int64_t i) -> int64_t {
        const int64_t ic = i < i_hi ? i : i_hi - 1;
        return second ? ic : dw1_row_of(g, ic);
    };
    int64_t my_id = 0;
    if (tid < kSub) rid[tid] = row_id(i_lo + tid);
    __syncthreads();
    fetch(0);
    if (tid < kSub) {
        my_id = row_id(i_lo + kSub + tid);
    }
    for (int s = 0; s < n_sub; ++s) {
        commit();             // sub-chunk s: registers -> LDS (the previous one's readers are behind the loop-end barrier)
        __syncthreads();      // ... and every thread has read RID for it
        if (tid < kSub) {
            rid[tid] = my_id;  // ids of sub-chunk s + 1; those of s + 2 fly during this one
            my_id = row_id(i_lo + (int64_t)(s + 2) * kSub + tid);
        }
        __syncthreads();
        if (s + 1 < n_sub) fetch(s + 1);
        const float *pa = sA + kq * kLdA + 16 * w + c16;   // A[i = o][k = row]
        const float *pb = sB + kq * kLdB + c16;            // B[k = row][j = column]
#pragma unroll 4
        for (int r0 = 0; r0 < kSub; r0 += 4) {
            const float av = pa[r0 * kLdA];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma4(av, pb[r0 * kLdB + 16 * t], acc[t]);
        }
        __syncthreads();
    }
    float *slab = (second ? g.slabs2 : g.slabs) + (size_t)rc * (size_t)kH * K1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = 16 * w + kq * 4 + r;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int k = c0 + 16 * t + c16;
            if (k < K1) __builtin_nontemporal_store(acc[t][r], slab + (size_t)o * K1 + k);
        }
    }
}

}  // namespace

constexpr size_t kDw1Lds = (size_t)kSub * (kLdA + 112) * sizeof(float) + kSub * sizeof(int64_t);  // 66 048 B

// one-time function attributes (dynamic LDS above 64 KB): outside any stream capture (tsm_critic_rows_init)
int tsm_critic_dw1_init() {
    static bool done = false;
    if (!done) {
#define DW1_ATTR(V, T) TSM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(critic_dw1_kernel<V, T>), \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDw1Lds))
        DW1_ATTR(true, 6); DW1_ATTR(false, 6); DW1_ATTR(true, 3); DW1_ATTR(false, 3);
#undef DW1_ATTR
        done = true;
    }
    return TSM_OK;
}

// rows per chunk / number of chunks (= partial slabs of dW1) for Mr rows: a full chip of workgroups, chunks of whole
// 64-row sub-chunks
// Column tiles per workgroup.  Few rows per CU (the PPO minibatch: 8192 rows over 256 CUs): the launch is a latency chain
// whatever the split, so 48-column blocks -- half the slabs.  Many rows (CTDEPolicy.learn: 102 400): the MFMA loop is what
// counts, and 96-column blocks reuse every dH1 operand read for six products instead of three (117.8 vs 143.4 us).
static int dw1_nt(int64_t Mr, int32_t in_dim) {
    const int64_t chunks6 = n_cu_dev() / ceil_div(in_dim, 96) > 0 ? n_cu_dev() / ceil_div(in_dim, 96) : 1;
    return Mr / chunks6 >= 512 ? 6 : 3;
}

static void dw1_plan(int64_t Mr, int32_t in_dim, int64_t *RC, int *n_chunk) {
    const int ncb = (int)ceil_div(in_dim, 16 * dw1_nt(Mr, in_dim));
    int64_t want = (int64_t)n_cu_dev() / ncb;
    if (want < 1) want = 1;
    const int64_t subs = ceil_div(Mr, kSub);
    if (want > subs) want = subs;
    *RC = ceil_div(subs, want) * kSub;
    *n_chunk = (int)ceil_div(Mr, *RC);
}

TSM_EXPORT int tsm_critic_rows_dw1_chunks(int64_t Mr, int32_t in_dim) {
    if (Mr <= 0 || in_dim < 1) return 0;
    int64_t RC;
    int n;
    dw1_plan(Mr, in_dim, &RC, &n);
    return n;
}

TSM_EXPORT int tsm_critic_rows_dw1(const float *dh1, const float *obs_rows, int32_t in_dim, const int64_t *rows,
                                   int64_t first_row, int64_t tm_T, int64_t tm_E, int64_t Mr, int32_t n_chunks,
                                   float *w1_slabs_out, const float *dh2, const float *h1, float *w2_slabs_out,
                                   const tsm_slab_reduce *side_host, int32_t n_side, void *stream) {
    TSM_REQUIRE(in_dim >= 1 && in_dim <= 384 && ((in_dim & 3) == 0 || in_dim <= 64) && Mr >= 1,
                "tsm_critic_rows_dw1: in_dim = %d (<= 384, a multiple of 4 above 64), Mr = %lld", in_dim, (long long)Mr);
    TSM_REQUIRE(tm_T <= 0 || Mr < (1ll << 31), "tsm_critic_rows_dw1: the time-major row mapping is 32-bit (Mr < 2^31)");
    TSM_REQUIRE(dh1 && obs_rows && w1_slabs_out, "tsm_critic_rows_dw1: null pointer");
    TSM_REQUIRE(tm_T == 0 || (tm_T > 0 && tm_E > 0 && tm_T * tm_E == Mr), "tsm_critic_rows_dw1: T x E must equal Mr");
    {
        const int rc_ = tsm_critic_dw1_init();
        if (rc_ != TSM_OK) return rc_;
    }
    int64_t RC;
    int n;
    dw1_plan(Mr, in_dim, &RC, &n);
    TSM_REQUIRE(n_chunks == n, "tsm_critic_rows_dw1: n_chunks = %d, tsm_critic_rows_dw1_chunks says %d", n_chunks, n);
    Dw1Args g{};
    g.dh1 = dh1; g.obs = obs_rows; g.rows = rows; g.first_row = first_row; g.Mr = Mr; g.tm_T = tm_T; g.tm_E = tm_E;
    g.K1 = in_dim; g.RC = RC; g.slabs = w1_slabs_out;
    const int nt = dw1_nt(Mr, in_dim);
    TSM_REQUIRE((dh2 == nullptr) == (h1 == nullptr) && (dh2 == nullptr) == (w2_slabs_out == nullptr),
                "tsm_critic_rows_dw1: dh2, h1 and w2_slabs_out come together");
    g.ncb1 = (int)ceil_div(in_dim, 16 * nt);
    g.ncb = g.ncb1 + (dh2 ? (int)ceil_div(kH, 16 * nt) : 0); g.n_chunk = n;
    g.dh2 = dh2; g.h1 = h1; g.slabs2 = w2_slabs_out;
    g.n_main = (int)(g.ncb * ceil_div(n, 8) * 8);
    TSM_REQUIRE(n_side >= 0 && n_side <= 2 && (n_side == 0 || side_host), "tsm_critic_rows_dw1: at most two side reductions");
    int side_blocks = 0;
    for (int k = 0; k < n_side; ++k) {
        const tsm_slab_reduce &r = side_host[k];
        TSM_REQUIRE(r.slabs && r.out && r.n >= 1 && r.n_slab >= 1 && r.stride >= r.n, "tsm_critic_rows_dw1: bad side reduction %d", k);
        g.side[k].slabs = r.slabs; g.side[k].n = r.n; g.side[k].stride = r.stride; g.side[k].n_slab = r.n_slab; g.side[k].out = r.out;
        g.side[k].first_blk = side_blocks;
        side_blocks += (int)ceil_div(r.n, kCols);
    }
    g.n_side = n_side;
    const dim3 grid((unsigned)(g.n_main + side_blocks));
    const bool vec = (in_dim & 3) == 0;
#define DW1_LAUNCH(V, T) hipLaunchKernelGGL((critic_dw1_kernel<V, T>), grid, dim3(kThreads), kDw1Lds, tsm_stream(stream), g)
    if (nt == 6) { if (vec) DW1_LAUNCH(true, 6); else DW1_LAUNCH(false, 6); }
    else { if (vec) DW1_LAUNCH(true, 3); else DW1_LAUNCH(false, 3); }
#undef DW1_LAUNCH
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}
