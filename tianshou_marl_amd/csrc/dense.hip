// dense.hip -- fully-connected networks of arbitrary width: forward, backward and weight-gradient slabs.
//
// Replaces, for networks that do not fit the fused 64-wide kernels of mlp_fused.hip:
//   nn.Linear / F.relu / tanh stacks of the actor and critic modules
//     DecentralizedActor.forward      /root/reference/tianshou/algorithm/multiagent/ctde.py:366-379
//     CentralizedCritic.forward       ctde.py:402-414
//     MLP (Net preprocess + head)     tianshou/utils/net/common.py:67-160, discrete.py:28-170
//   and `loss.backward()` through them (ctde.py:188-194, ppo.py:208-210).
//
// One tiled GEMM kernel on f32-input MFMA (v_mfma_f32_16x16x4_f32: exact f32 products and accumulation, so the
// 1e-5 parity bar of the north star holds; bf16 inputs would not).  Workgroup = 256 threads = 4 waves in a 2x2
// arrangement over a 64x64 output tile, each wave 32x32 = 2x2 MFMA accumulators; the K loop stages 64x16 tiles
// of both operands through LDS (row stride 18 words = 2 x odd: conflict-free on the 32-bank LDS for the MFMA operand read pattern
// lane -> [lane & 15][lane >> 4], and 16-B aligned for vector stores), double buffered.
// Three instantiations cover a layer:
//   forward   Y[B,O]  = act(X[B,K] . W[O,K]^T + b)                   A row-major, B row-major([N,K])
//   dgrad     dX[B,K] = (dZ[B,O] . W[O,K]) * act'(X)                 A row-major, B "k-major" ([K,N])
//   wgrad     dW[O,K] = dZ^T . X,  db[O] = column sums of dZ          A k-major, B k-major + a virtual ones column,
//             split over the batch into `n_split` slabs (deterministic; tsm_adam_step sums the slabs)
// Parameter vector layout = torch `parameters()` order: w0[d1,d0], b0[d1], w1[d2,d1], b1[d2], ...
#include "common.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int BM = 64, BN = 64, BK = 16, LDT = BK + 2, NT = 256;  // a k tile = BK / 16 sub-tiles of 16, staged side by side
// (BK = 32 measured slower on MI355X: 29.6 % vs 34.3 % of the f32-MFMA peak on the 384-128-128-8 critic forward)
constexpr int KS = BK / 16;

enum { EPI_FWD = 0, EPI_DGRAD = 1, EPI_WGRAD = 2 };

struct GemmArgs {
    const float *A; int64_t lda;   // row-major: A[m*lda + k]; k-major: A[k*lda + m]
    const float *B; int64_t ldb;   // row-major: B[n*ldb + k]; k-major: B[k*ldb + n]
    int64_t M, N, K;               // N includes the virtual ones column for wgrad
    int64_t k_per_split;
    float *C; int64_t ldc;         // fwd / dgrad output
    const float *bias;             // fwd
    const float *X; int64_t ldx;   // dgrad: layer input (activation output of the previous layer)
    int act;                       // 0 none, 1 relu, 2 tanh (fwd: applied; dgrad: derivative w.r.t. X)
    int64_t slab_stride, w_off, b_off;  // wgrad
    const int32_t *run_if;         // nullable device flag: the launch is a no-op when *run_if == 0 (tsm_mlp_forward_cond)
};

__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return tanhf(v);
    return v;
}
// derivative expressed through the activation OUTPUT y (relu: y > 0; tanh: 1 - y^2)
__device__ __forceinline__ float act_bwd(float y, int act) {
    if (act == 1) return y > 0.f ? 1.f : 0.f;
    if (act == 2) return 1.f - y * y;
    return 1.f;
}

// tile[r][k] <- src[(r0 + r) * ld + k0 + k]   (contiguous along k).  `fast` is workgroup-uniform: the whole 64x16 piece
// is in bounds and every row start is 16-B aligned -> one unconditional 16-B load per thread (the compiler can then
// issue it early and wait for it late); edge pieces take the guarded path.
__device__ __forceinline__ void load_rowmajor(const float *__restrict__ src, int64_t ld, int64_t r0, int64_t R,
                                              int64_t k0, int64_t kend, float (&v)[4], int tid, bool fast) {
    const int r = tid >> 2, k4 = (tid & 3) * 4;
    const int64_t row = r0 + r, k = k0 + k4;
    if (fast) {
        const float4 q = *reinterpret_cast<const float4 *>(src + row * ld + k);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        return;
    }
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (row >= R) return;
    const float *p = src + row * ld + k;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (k + i < kend) v[i] = p[i];
}
__device__ __forceinline__ void store_rowmajor(float *tile, const float (&v)[4], int tid, int koff) {
    const int r = tid >> 2, k4 = (tid & 3) * 4 + koff;
    // row stride 18 words (2 x odd): conflict-free operand reads on the 32-bank LDS; rows are 8-B aligned only
    *reinterpret_cast<float2 *>(tile + r * LDT + k4) = make_float2(v[0], v[1]);
    *reinterpret_cast<float2 *>(tile + r * LDT + k4 + 2) = make_float2(v[2], v[3]);
}

// tile[r][k] <- src[(k0 + k) * ld + r0 + r]   (contiguous along r); `ones_r` = index of a virtual all-ones row
__device__ __forceinline__ void load_kmajor(const float *__restrict__ src, int64_t ld, int64_t r0, int64_t R,
                                            int64_t k0, int64_t kend, int64_t ones_r, float (&v)[4], int tid, bool fast) {
    const int k = tid >> 4, r4 = (tid & 15) * 4;
    const int64_t kk = k0 + k, row = r0 + r4;
    if (fast) {
        const float4 q = *reinterpret_cast<const float4 *>(src + kk * ld + row);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        return;
    }
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (kk >= kend) return;
    const float *p = src + kk * ld + row;
    const int64_t Rdata = ones_r >= 0 ? ones_r : R;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (row + i < Rdata) v[i] = p[i];
        else if (row + i == ones_r) v[i] = 1.f;
    }
}
__device__ __forceinline__ void store_kmajor(float *tile, const float (&v)[4], int tid, int koff) {
    const int k = (tid >> 4) + koff, r4 = (tid & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[(r4 + i) * LDT + k] = v[i];
}

// G > 1: intra-workgroup split-K.  G groups of 4 waves each walk 1/G of the workgroup's k range with their own LDS
// tiles; their accumulators are then folded through LDS in group order (deterministic) by group 0, which runs the
// epilogue.  Used for weight gradients of small layers, where (output tiles x slabs) alone cannot fill the chip and the
// k loop over the batch is latency-bound.
template <bool A_KMAJOR, bool B_KMAJOR, int EPI, int G>
__global__ __launch_bounds__(NT * G) void gemm_kernel(GemmArgs g) {
    constexpr int kTile = (BM + BN) * LDT;  // one A tile + one B tile
    __shared__ __attribute__((aligned(16))) float lds_all[G][2 * kTile];
    if (g.run_if && *g.run_if == 0) return;  // workgroup-uniform
    const int grp = threadIdx.x / NT;
    float *const lds_g = lds_all[grp];  // tile pair `buf` of this group: A at lds_g + buf * kTile, B behind it
    const int tid = threadIdx.x % NT, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
    int64_t kbeg = (int64_t)blockIdx.z * g.k_per_split;
    int64_t kend = kbeg + g.k_per_split < g.K ? kbeg + g.k_per_split : g.K;
    int64_t n_iter = 0;  // identical for every group: the barriers inside the k loop are workgroup-wide
    if (G > 1) {
        const int64_t span = kend > kbeg ? kend - kbeg : 0;
        const int64_t per = (((span + G - 1) / G + BK - 1) / BK) * BK;  // whole tiles per group
        n_iter = per / BK;
        const int64_t b2 = kbeg + (int64_t)grp * per;
        const int64_t e2 = b2 + per < kend ? b2 + per : kend;
        kbeg = b2 < kend ? b2 : kend;
        kend = e2 > kbeg ? e2 : kbeg;
    } else {
        n_iter = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
    }
    const int64_t ones_n = EPI == EPI_WGRAD ? g.N - 1 : -1;

    f4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    // workgroup-uniform "no edge, aligned" predicates of the two operands (the k extent is checked per tile)
    const auto al16 = [](const float *p, int64_t ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 3) == 0; };
    const bool a_in = al16(g.A, g.lda) && m0 + BM <= g.M;
    const bool b_in = al16(g.B, g.ldb) && n0 + BN <= (ones_n >= 0 ? ones_n : g.N);
    float va[KS][4], vb[KS][4];
    auto fetch = [&](int64_t k0) {
#pragma unroll
        for (int h = 0; h < KS; ++h) {
            const bool k_in = k0 + 16 * (h + 1) <= kend;
            if (A_KMAJOR) load_kmajor(g.A, g.lda, m0, g.M, k0 + 16 * h, kend, -1, va[h], tid, a_in && k_in);
            else load_rowmajor(g.A, g.lda, m0, g.M, k0 + 16 * h, kend, va[h], tid, a_in && k_in);
            if (B_KMAJOR) load_kmajor(g.B, g.ldb, n0, g.N, k0 + 16 * h, kend, ones_n, vb[h], tid, b_in && k_in);
            else load_rowmajor(g.B, g.ldb, n0, g.N, k0 + 16 * h, kend, vb[h], tid, b_in && k_in);
        }
    };
    auto commit = [&](int buf) {
        float *at = lds_g + buf * kTile, *bt = at + BM * LDT;
#pragma unroll
        for (int h = 0; h < KS; ++h) {
            if (A_KMAJOR) store_kmajor(at, va[h], tid, 16 * h); else store_rowmajor(at, va[h], tid, 16 * h);
            if (B_KMAJOR) store_kmajor(bt, vb[h], tid, 16 * h); else store_rowmajor(bt, vb[h], tid, 16 * h);
        }
    };

    int buf = 0;
    if (n_iter > 0) {
        fetch(kbeg);  // (a group whose share of the range is empty stages zeros)
        commit(0);
    }
    __syncthreads();
    for (int64_t it = 0; it < n_iter; ++it) {
        const int64_t k0 = kbeg + it * BK;
        const bool more = it + 1 < n_iter;
        if (more) fetch(k0 + BK);  // global loads of the next tile fly while this one is multiplied
        const float *a_t = lds_g + buf * kTile + (wm * 32 + (lane & 15)) * LDT + (lane >> 4);
        const float *b_t = lds_g + buf * kTile + BM * LDT + (wn * 32 + (lane & 15)) * LDT + (lane >> 4);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            const float a0 = a_t[kk], a1 = a_t[16 * LDT + kk];
            const float b0 = b_t[kk], b1 = b_t[16 * LDT + kk];
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) commit(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    if (G > 1) {  // fold the groups' accumulators in group order; the staging tiles are free after the last barrier
        float *mine = lds_all[grp] + tid * 16;
        if (grp > 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mine[(i * 2 + j) * 4 + r] = acc[i][j][r];
        }
        __syncthreads();
        if (grp > 0) return;
        for (int q = 1; q < G; ++q) {
            const float *o = lds_all[q] + tid * 16;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += o[(i * 2 + j) * 4 + r];
        }
    }
    // C fragment: register r of lane l holds C[(l >> 4) * 4 + r][l & 15]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t col = n0 + wn * 32 + j * 16 + (lane & 15);
            if (col >= g.N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
                if (row >= g.M) continue;
                float v = acc[i][j][r];
                if (EPI == EPI_FWD) {
                    g.C[row * g.ldc + col] = act_fwd(v + g.bias[col], g.act);
                } else if (EPI == EPI_DGRAD) {
                    if (g.act) v *= act_bwd(g.X[row * g.ldx + col], g.act);
                    g.C[row * g.ldc + col] = v;
                } else {
                    float *dst = g.C + (int64_t)blockIdx.z * g.slab_stride;
                    if (col == g.N - 1) dst[g.b_off + row] = v;                      // db[o]
                    else dst[g.w_off + row * (g.N - 1) + col] = v;                   // dW[o][k]
                }
            }
        }
}

int check_desc(const tsm_mlp_desc *d, const char *who) {
    TSM_REQUIRE(d, "%s: null descriptor", who);
    TSM_REQUIRE(d->n_layers >= 1 && d->n_layers <= TSM_MLP_MAX_LAYERS, "%s: n_layers must be in [1, %d]", who,
                TSM_MLP_MAX_LAYERS);
    TSM_REQUIRE(d->act >= 0 && d->act <= 2, "%s: act must be 0 (none), 1 (relu) or 2 (tanh)", who);
    for (int i = 0; i <= d->n_layers; ++i)
        TSM_REQUIRE(d->dims[i] >= 1 && d->dims[i] <= (1 << 20), "%s: dims[%d] = %d out of range", who, i, d->dims[i]);
    return TSM_OK;
}

}  // namespace

TSM_EXPORT int64_t tsm_mlp_param_count(const tsm_mlp_desc *d) {
    if (check_desc(d, "tsm_mlp_param_count") != TSM_OK) return -1;
    int64_t n = 0;
    for (int i = 0; i < d->n_layers; ++i) n += (int64_t)d->dims[i + 1] * d->dims[i] + d->dims[i + 1];
    return n;
}

TSM_EXPORT int64_t tsm_mlp_act_elems(const tsm_mlp_desc *d, int64_t B) {
    if (check_desc(d, "tsm_mlp_act_elems") != TSM_OK || B < 0) return -1;
    int64_t n = 0;
    for (int i = 1; i <= d->n_layers; ++i) n += B * d->dims[i];
    return n;
}

static int mlp_forward_impl(const tsm_mlp_desc *d, const float *params, const float *x, int64_t B, float *acts,
                            const int32_t *run_if, void *stream, const char *who) {
    if (int rc = check_desc(d, who)) return rc;
    TSM_REQUIRE(B >= 0, "%s: negative batch", who);
    if (B == 0) return TSM_OK;
    TSM_REQUIRE(params && x && acts, "%s: null pointer", who);
    const float *in = x;
    float *out = acts;
    const float *p = params;
    for (int l = 0; l < d->n_layers; ++l) {
        const int64_t K = d->dims[l], O = d->dims[l + 1];
        GemmArgs g{};
        g.A = in; g.lda = K; g.B = p; g.ldb = K; g.M = B; g.N = O; g.K = K; g.k_per_split = K;
        g.C = out; g.ldc = O; g.bias = p + O * K; g.act = l + 1 < d->n_layers ? d->act : 0;
        g.run_if = run_if;
        dim3 grid((unsigned)ceil_div(O, BN), (unsigned)ceil_div(B, BM), 1);
        TSM_REQUIRE(grid.y <= 65535, "%s: batch %lld too large for one launch (max %d rows)", who, (long long)B, 65535 * BM);
        hipLaunchKernelGGL((gemm_kernel<false, false, EPI_FWD, 1>), grid, dim3(NT), 0, tsm_stream(stream), g);
        TSM_LAUNCH_CHECK();
        p += O * K + O;
        in = out;
        out += B * O;
    }
    return TSM_OK;
}

TSM_EXPORT int tsm_mlp_forward(const tsm_mlp_desc *d, const float *params, const float *x, int64_t B, float *acts,
                               void *stream) {
    return mlp_forward_impl(d, params, x, B, acts, nullptr, stream, "tsm_mlp_forward");
}

TSM_EXPORT int tsm_mlp_forward_cond(const tsm_mlp_desc *d, const float *params, const float *x, int64_t B, float *acts,
                                    const int32_t *run_if_nonzero, void *stream) {
    TSM_REQUIRE(run_if_nonzero, "tsm_mlp_forward_cond: null flag");
    return mlp_forward_impl(d, params, x, B, acts, run_if_nonzero, stream, "tsm_mlp_forward_cond");
}

TSM_EXPORT int tsm_mlp_backward(const tsm_mlp_desc *d, const float *params, const float *x, int64_t B,
                                const float *acts, const float *d_out, float *d_acts, int32_t n_split,
                                float *slabs, int64_t slab_stride, void *stream) {
    if (int rc = check_desc(d, "tsm_mlp_backward")) return rc;
    TSM_REQUIRE(B >= 1, "tsm_mlp_backward: batch must be >= 1");
    TSM_REQUIRE(n_split >= 1 && n_split <= 65535, "tsm_mlp_backward: n_split out of range");
    TSM_REQUIRE(params && x && acts && d_out && slabs, "tsm_mlp_backward: null pointer");
    TSM_REQUIRE(d->n_layers == 1 || d_acts, "tsm_mlp_backward: d_acts workspace required for n_layers > 1");
    TSM_REQUIRE(ceil_div(B, BM) <= 65535, "tsm_mlp_backward: batch too large for one launch");
    const int L = d->n_layers;
    const int64_t n_param = tsm_mlp_param_count(d);
    TSM_REQUIRE(slab_stride == 0 || slab_stride >= n_param, "tsm_mlp_backward: slab_stride smaller than the parameter count");
    if (slab_stride == 0) slab_stride = n_param;
    // offsets of each layer's parameters / activation block
    int64_t p_off[TSM_MLP_MAX_LAYERS], a_off[TSM_MLP_MAX_LAYERS + 1];
    {
        int64_t po = 0, ao = 0;
        for (int l = 0; l < L; ++l) {
            p_off[l] = po;
            a_off[l] = ao;  // block holding the OUTPUT of layer l
            po += (int64_t)d->dims[l + 1] * d->dims[l] + d->dims[l + 1];
            ao += B * d->dims[l + 1];
        }
    }
    // k range of the batch handled by one slab: a multiple of BK so that tiles never straddle two slabs
    int64_t k_per = ceil_div(ceil_div(B, n_split), BK) * BK;
    const float *dz = d_out;  // gradient w.r.t. the pre-activation output of layer l (last layer is linear)
    for (int l = L - 1; l >= 0; --l) {
        const int64_t K = d->dims[l], O = d->dims[l + 1];
        const float *in = l == 0 ? x : acts + a_off[l - 1];
        const float *W = params + p_off[l];
        {   // wgrad + bias grad into every slab
            GemmArgs g{};
            g.A = dz; g.lda = O; g.B = in; g.ldb = K; g.M = O; g.N = K + 1; g.K = B; g.k_per_split = k_per;
            g.C = slabs; g.slab_stride = slab_stride; g.w_off = p_off[l]; g.b_off = p_off[l] + O * K;
            dim3 grid((unsigned)ceil_div(K + 1, BN), (unsigned)ceil_div(O, BM), (unsigned)n_split);
            // few output tiles (small layers): 4 wave-groups per workgroup split the batch range once more
            if ((int64_t)grid.x * grid.y * grid.z < 1024)
                hipLaunchKernelGGL((gemm_kernel<true, true, EPI_WGRAD, 4>), grid, dim3(NT * 4), 0, tsm_stream(stream), g);
            else
                hipLaunchKernelGGL((gemm_kernel<true, true, EPI_WGRAD, 1>), grid, dim3(NT), 0, tsm_stream(stream), g);
            TSM_LAUNCH_CHECK();
        }
        if (l > 0) {  // dgrad, multiplied by the derivative of the previous layer's activation
            float *dx = d_acts + a_off[l - 1];
            GemmArgs g{};
            g.A = dz; g.lda = O; g.B = W; g.ldb = K; g.M = B; g.N = K; g.K = O; g.k_per_split = O;
            g.C = dx; g.ldc = K; g.X = in; g.ldx = K; g.act = d->act;
            dim3 grid((unsigned)ceil_div(K, BN), (unsigned)ceil_div(B, BM), 1);
            hipLaunchKernelGGL((gemm_kernel<false, true, EPI_DGRAD, 1>), grid, dim3(NT), 0, tsm_stream(stream), g);
            TSM_LAUNCH_CHECK();
            dz = dx;
        }
    }
    return TSM_OK;
}
