// mlp_tile.h -- f32-MFMA tile engine of the fused actor + critic MLP (shared by mlp_fused.hip and rollout.hip).
// See mlp_fused.hip for the network definition, parameter layout and the gfx950 mapping.
#pragma once
#include "common.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int R = 16;      // rows per tile
constexpr int NT = 256;    // threads per workgroup (4 waves)
constexpr int kMaxJ = 4;   // obs_dim <= 64

struct Dims {
    int D, A;       // obs dim, n actions
    int Kp1;        // D rounded up to 4 (L1 k extent)
    int nJ;         // ceil(D / 16)
    int ld1;        // 16*nJ + 2
};

// Dims of a network whose observation width and action count are known at compile time (kernel specialisations for BASELINE
// configs[1]: obs_dim 18, 5 actions): every derived offset, loop bound and division by ld1 / D becomes a constant.
__host__ __device__ constexpr Dims dims_const(int D, int A) {
    Dims d{};
    d.D = D; d.A = A; d.Kp1 = (D + 3) / 4 * 4; d.nJ = (D + 15) / 16; d.ld1 = 16 * ((D + 15) / 16) + 2;
    return d;
}

template <int H>
struct Lay {  // LDS layout in floats
    static constexpr int ldh = H + 2;       // 66: 2*33
    static constexpr int ld2 = 2 * H + 2;   // 130: 2*65
    static constexpr int ldo = 18;          // logits(16) | value | pad
    int W1, W2a, W2c, W3a, W3c, B1, B2, B3a, B3c, X, H1, H2, OUT, D3, D2, D1, total;
    __host__ __device__ Lay(const Dims &d, bool bwd) {
        int o = 0;
        W1 = o; o += 2 * H * d.ld1;
        W2a = o; o += H * ldh;
        W2c = o; o += H * ldh;
        W3a = o; o += 16 * ldh;
        W3c = o; o += H;
        B1 = o; o += 2 * H;
        B2 = o; o += 2 * H;
        B3a = o; o += 16;
        B3c = o; o += 2;
        o = (o + 3) & ~3;  // weights image [0, X) is a whole number of 16-B words
        X = o; o += R * d.ld1;
        H1 = o; o += R * ld2;
        H2 = o; o += R * ld2;
        OUT = o; o += R * ldo;
        D3 = o; o += bwd ? R * ldo : 0;
        D2 = o; o += bwd ? R * ld2 : 0;
        D1 = o; o += bwd ? R * ld2 : 0;
        total = o;
    }
};

__device__ __forceinline__ f4 mfma(float a, float b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// s = 0; s = fmaf(h[j], w[j], s) for j = 0 .. N - 1 in this order (the critic's output layer on one lane per row) with the operands read
// from LDS eight steps at a time, a group ahead of their use.  The plain loop read four values, waited, used them: 0.84-0.88 us for a
// 64-long chain against 0.52 us this way (tools/stamp_update.py).
template <int N>
__device__ __forceinline__ float fmaf_chain_ahead(const float *h, const float *w) {
    constexpr int Q = 8;
    static_assert(N % (2 * Q) == 0, "pairs of whole groups");
    float h0[Q], w0[Q], h1[Q], w1[Q], s = 0.f;
    int off = 0;   // always 0, but opaque: ties each group's reads to the chain (see `pin` below)
    // (Ordering by data, not by __builtin_amdgcn_sched_barrier: the reads have no dependence on the barrier intrinsic, instruction
    //  selection hoists all 128 of them to the top of the block and the machine scheduler may then not move them back across the
    //  barriers -- 128 live registers, measured: 149 -> 256 + spills.  The empty asm reads and writes BOTH the running sum and the
    //  offset the next group's addresses use: it cannot move above the fmas before it, nor the next group's reads above it.)
    auto pin = [&]() { asm volatile("" : "+v"(s), "+v"(off)); };
#pragma unroll
    for (int p = 0; p < Q; ++p) { h0[p] = h[p]; w0[p] = w[p]; }
#pragma unroll
    for (int j0 = 0; j0 < N; j0 += 2 * Q) {
        pin();
#pragma unroll
        for (int p = 0; p < Q; ++p) { h1[p] = h[off + j0 + Q + p]; w1[p] = w[off + j0 + Q + p]; }
#pragma unroll
        for (int p = 0; p < Q; ++p) s = fmaf(h0[p], w0[p], s);
        pin();
        if (j0 + 2 * Q < N) {
#pragma unroll
            for (int p = 0; p < Q; ++p) { h0[p] = h[off + j0 + 2 * Q + p]; w0[p] = w[off + j0 + 2 * Q + p]; }
        }
#pragma unroll
        for (int p = 0; p < Q; ++p) s = fmaf(h1[p], w1[p], s);
    }
    return s;
}

// parameter offsets in the flat vector
template <int H>
struct POff {
    int aW1, ab1, aW2, ab2, aW3, ab3, cW1, cb1, cW2, cb2, cW3, cb3, total;
    __host__ __device__ POff(int D, int A) {
        int o = 0;
        aW1 = o; o += H * D; ab1 = o; o += H; aW2 = o; o += H * H; ab2 = o; o += H; aW3 = o; o += A * H; ab3 = o; o += A;
        cW1 = o; o += H * D; cb1 = o; o += H; cW2 = o; o += H * H; cb2 = o; o += H; cW3 = o; o += H; cb3 = o; o += 1;
        total = o;
    }
};

template <int H>
__device__ void stage_weights(float *lds, const Lay<H> &ly, const Dims &d, const float *__restrict__ P, int tid_in = -1) {
    const POff<H> po(d.D, d.A);
    const int tid = tid_in < 0 ? (int)threadIdx.x : tid_in;  // NT threads take part (rollout_tag.hip: waves 4-7 pass tid - NT)
    // loads are issued in batches of U per thread before any LDS write, so a batch costs one memory round trip (U = 4 until round 4:
    // 13 round trips, 7 us of the tag rollout's prologue and ~4 us of every policy_forward launch that reads the flat parameters)
    constexpr int U = 12;
    for (int e0 = tid; e0 < 2 * H * d.ld1; e0 += U * NT) {  // W1 actor|critic, zero padded
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * NT;
            const int row = e / d.ld1, c = e - row * d.ld1;
            v[u] = 0.f;
            if (e < 2 * H * d.ld1 && c < d.D) v[u] = row < H ? P[po.aW1 + row * d.D + c] : P[po.cW1 + (row - H) * d.D + c];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (e0 + u * NT < 2 * H * d.ld1) lds[ly.W1 + e0 + u * NT] = v[u];
    }
    for (int e0 = tid; e0 < H * ly.ldh; e0 += U * NT) {
        float va[U], vc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * NT;
            const int row = e / ly.ldh, c = e - row * ly.ldh;
            const bool on = e < H * ly.ldh && c < H;
            va[u] = on ? P[po.aW2 + row * H + c] : 0.f;
            vc[u] = on ? P[po.cW2 + row * H + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * NT;
            if (e < H * ly.ldh) { lds[ly.W2a + e] = va[u]; lds[ly.W2c + e] = vc[u]; }
        }
    }
    {   // W3 (actor, rows >= A zero) and the five H-long vectors: all loads of a thread first, then the LDS writes
        constexpr int N3 = (16 * Lay<H>::ldh + NT - 1) / NT;
        float v3[N3], vv[5];
#pragma unroll
        for (int u = 0; u < N3; ++u) {
            const int e = tid + u * NT, row = e / ly.ldh, c = e - row * ly.ldh;
            v3[u] = (e < 16 * ly.ldh && row < d.A && c < H) ? P[po.aW3 + row * H + c] : 0.f;
        }
        const int e5 = tid < H ? tid : 0;
        vv[0] = P[po.cW3 + e5]; vv[1] = P[po.ab1 + e5]; vv[2] = P[po.cb1 + e5]; vv[3] = P[po.ab2 + e5]; vv[4] = P[po.cb2 + e5];
#pragma unroll
        for (int u = 0; u < N3; ++u) {
            const int e = tid + u * NT;
            if (e < 16 * ly.ldh) lds[ly.W3a + e] = v3[u];
        }
        if (tid < H) {
            lds[ly.W3c + tid] = vv[0]; lds[ly.B1 + tid] = vv[1]; lds[ly.B1 + H + tid] = vv[2]; lds[ly.B2 + tid] = vv[3];
            lds[ly.B2 + H + tid] = vv[4];
        }
    }
    if (tid < 16) lds[ly.B3a + tid] = tid < d.A ? P[po.ab3 + tid] : 0.f;
    if (tid == 0) lds[ly.B3c] = P[po.cb3];
}

constexpr int kStageChunk = 4 * 256;  // float4 per full round of the staging loop (4 per thread x NT threads)
__host__ __device__ inline int image_f4_padded(int x_floats) {
    return ((x_floats / 4) + kStageChunk - 1) / kStageChunk * kStageChunk;
}

// Staging from the padded parameter image kept up to date by tsm_adam_step (img[0 .. ly.X) has exactly the
// LDS layout W1..B3c incl. zero pads): straight 16-B copies, ~12 independent loads per thread.
template <int H>
__device__ __forceinline__ void stage_image(float *lds, const Lay<H> &ly, const float *__restrict__ img) {
    // LDS-DMA (global_load_lds_dwordx4): the image has exactly the LDS layout, so each wave instruction copies one
    // contiguous 1-KiB piece (wave-uniform LDS base + lane * 16 B) with no VGPR round trip, and all ~12 pieces of a
    // wave are in flight at once -- one memory round trip for the whole 48 KB instead of three.  The image is padded
    // with zeros to a whole number of kStageChunk float4, so there is no tail predicate; the copy may run a few KB
    // past ly.X into the activation buffers, which only ever receive zeros or are rewritten after the next barrier
    // (the barrier drains the DMA: hipcc emits vmcnt(0) in front of it).
    const int n4p = image_f4_padded(ly.X);
    const int wave_base = (threadIdx.x >> 6) * 64;
    for (int e = 0; e < n4p; e += NT) {
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(img + (size_t)(e + threadIdx.x) * 4),
            (__attribute__((address_space(3))) void *)(lds + (size_t)(e + wave_base) * 4), 16, 0, 0);
    }
}

// forward of one 16-row tile already staged in lds[ly.X]; leaves H1, H2, OUT (logits | value) in LDS
template <int H>
__device__ __forceinline__ void tile_forward(float *lds, const Lay<H> &ly, const Dims &d) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    // ---- L1: [R x D] x W1^T -> H1 (actor cols 0..H-1 | critic cols H..2H-1) ----
    {
        f4 acc_a = {0.f, 0.f, 0.f, 0.f}, acc_c = {0.f, 0.f, 0.f, 0.f};
        const float *xa = lds + ly.X + r16 * d.ld1 + kq;
        const float *wa = lds + ly.W1 + (16 * w + r16) * d.ld1 + kq;
        const float *wc = lds + ly.W1 + (H + 16 * w + r16) * d.ld1 + kq;
        for (int k0 = 0; k0 < d.Kp1; k0 += 4) {
            const float a = xa[k0];
            acc_a = mfma(a, wa[k0], acc_a);
            acc_c = mfma(a, wc[k0], acc_c);
        }
        const int col = 16 * w + r16;
        const float ba = lds[ly.B1 + col], bc = lds[ly.B1 + H + col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = kq * 4 + r;
            lds[ly.H1 + row * ly.ld2 + col] = fmaxf(acc_a[r] + ba, 0.f);
            lds[ly.H1 + row * ly.ld2 + H + col] = fmaxf(acc_c[r] + bc, 0.f);
        }
    }
    __syncthreads();
    // ---- L2 ----
    {
        f4 acc_a = {0.f, 0.f, 0.f, 0.f}, acc_c = {0.f, 0.f, 0.f, 0.f};
        const float *ha = lds + ly.H1 + r16 * ly.ld2 + kq;
        const float *hc = ha + H;
        const float *wa = lds + ly.W2a + (16 * w + r16) * ly.ldh + kq;
        const float *wc = lds + ly.W2c + (16 * w + r16) * ly.ldh + kq;
#pragma unroll
        for (int k0 = 0; k0 < H; k0 += 4) {
            acc_a = mfma(ha[k0], wa[k0], acc_a);
            acc_c = mfma(hc[k0], wc[k0], acc_c);
        }
        const int col = 16 * w + r16;
        const float ba = lds[ly.B2 + col], bc = lds[ly.B2 + H + col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = kq * 4 + r;
            lds[ly.H2 + row * ly.ld2 + col] = fmaxf(acc_a[r] + ba, 0.f);
            lds[ly.H2 + row * ly.ld2 + H + col] = fmaxf(acc_c[r] + bc, 0.f);
        }
    }
    __syncthreads();
    // ---- L3: wave 0 -> logits (MFMA, A padded to 16); wave 1 -> value (VALU dot) ----
    if (w == 0) {
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        const float *ha = lds + ly.H2 + r16 * ly.ld2 + kq;
        const float *wa = lds + ly.W3a + r16 * ly.ldh + kq;
#pragma unroll
        for (int k0 = 0; k0 < H; k0 += 4) acc = mfma(ha[k0], wa[k0], acc);
        const float b = lds[ly.B3a + r16];
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[ly.OUT + (kq * 4 + r) * ly.ldo + r16] = acc[r] + b;
    } else if (w == 1 && lane < R) {
        const float *hc = lds + ly.H2 + lane * ly.ld2 + H;
        float s = 0.f;
        for (int j = 0; j < H; ++j) s = fmaf(hc[j], lds[ly.W3c + j], s);
        lds[ly.OUT + lane * ly.ldo + 16] = s + lds[ly.B3c];
    }
    __syncthreads();
}

// The same forward for a workgroup of EIGHT waves (rollout.hip): waves 0-3 carry the actor, waves 4-7 the critic, one
// accumulator chain per wave and layer instead of two interleaved ones.  Same chains, same k order as tile_forward:
// bit-identical outputs.  All 512 threads must call it (three barriers inside).
template <int H>
__device__ __forceinline__ void tile_forward_split(float *lds, const Lay<H> &ly, const Dims &d) {
    const int lane = threadIdx.x & 63, w8 = threadIdx.x >> 6, net = w8 >> 2, w = w8 & 3;
    const int r16 = lane & 15, kq = lane >> 4;
    const int col = 16 * w + r16;
    {   // L1
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        const float *xa = lds + ly.X + r16 * d.ld1 + kq;
        const float *wp = lds + ly.W1 + (net * H + col) * d.ld1 + kq;
        for (int k0 = 0; k0 < d.Kp1; k0 += 4) acc = mfma(xa[k0], wp[k0], acc);
        const float b = lds[ly.B1 + net * H + col];
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[ly.H1 + (kq * 4 + r) * ly.ld2 + net * H + col] = fmaxf(acc[r] + b, 0.f);
    }
    __syncthreads();
    {   // L2
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        const float *ha = lds + ly.H1 + r16 * ly.ld2 + net * H + kq;
        const float *wp = lds + (net ? ly.W2c : ly.W2a) + col * ly.ldh + kq;
#pragma unroll
        for (int k0 = 0; k0 < H; k0 += 4) acc = mfma(ha[k0], wp[k0], acc);
        const float b = lds[ly.B2 + net * H + col];
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[ly.H2 + (kq * 4 + r) * ly.ld2 + net * H + col] = fmaxf(acc[r] + b, 0.f);
    }
    __syncthreads();
    if (w8 == 0) {  // logits (MFMA, A padded to 16)
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        const float *ha = lds + ly.H2 + r16 * ly.ld2 + kq;
        const float *wa = lds + ly.W3a + r16 * ly.ldh + kq;
#pragma unroll
        for (int k0 = 0; k0 < H; k0 += 4) acc = mfma(ha[k0], wa[k0], acc);
        const float b = lds[ly.B3a + r16];
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[ly.OUT + (kq * 4 + r) * ly.ldo + r16] = acc[r] + b;
    } else if (w8 == 5 && lane < R) {  // value (VALU dot) -- on wave 5: wave 4 shares its SIMD with wave 0, and the f32 FMAs of one
        // wave of a SIMD take turns with the MFMAs of the other (profiles/r05_probe_mfma_f32_issue.txt); tag rollout 185.4 -> 182.4 us
        // (fmaf_chain_ahead here changed nothing in the rollouts: their forward 1.60 -> 1.64 us, round 5)
        const float *hc = lds + ly.H2 + lane * ly.ld2 + H;
        float s = 0.f;
        for (int j = 0; j < H; ++j) s = fmaf(hc[j], lds[ly.W3c + j], s);
        lds[ly.OUT + lane * ly.ldo + 16] = s + lds[ly.B3c];
    }
    __syncthreads();
}

constexpr int kXRegs = (R * (16 * kMaxJ + 2) + NT - 1) / NT;  // X tile elements per thread (worst case)

// Gather one X tile into registers in two phases so that other loads can be issued in between:
//   phase A  row ids (perm)   -> xs[]      phase B  obs values (dependent on the row ids) -> xr[]
// (Round 5: the row-id loads are UNCONDITIONAL loads at a clamped index, all issued before the first use.  Written with the load
//  inside `if (c < D && i < M)` every iteration was its own basic block and hipcc waited for each id before issuing the next: three
//  to five dependent memory round trips in front of the gathers of every launch of the update kernels -- the ISA showed
//  load / s_waitcnt vmcnt(0) / load / s_waitcnt vmcnt(0) ...; row0 < M holds for every tile that is launched.)
// phase A as two halves, so that a caller can put independent loads between the id loads and their first use:
//   _load : id[it] = row id of X element threadIdx.x + it * NT (garbage where the element does not exist); id[kXRegs] = row id of
//           tile row threadIdx.x >> 4 (the row whose loss-head inputs this thread fetches)
//   _done : xs[it] = source offset of the element or -1; returns the row id of tile row threadIdx.x >> 4 or -1
template <int HAS_PERM = -1>   // -1: decided at run time (a branch: the waits behind its join are conservative); 0 / 1: the caller knows
__device__ __forceinline__ void prefetch_tile_ids_load(int64_t (&id)[kXRegs + 1], const Dims &d, const int64_t *__restrict__ perm,
                                                       int64_t first, int64_t row0, int64_t M) {
    const int64_t ir = row0 + (threadIdx.x >> 4);
    if (HAS_PERM == 1 || (HAS_PERM == -1 && perm)) {
#pragma unroll
        for (int it = 0; it < kXRegs; ++it) {
            const int e = threadIdx.x + it * NT, r = e / d.ld1;
            const int64_t i = row0 + r;
            id[it] = perm[(e < R * d.ld1 && i < M) ? i : row0];
        }
        id[kXRegs] = perm[ir < M ? ir : row0];
    } else {
#pragma unroll
        for (int it = 0; it < kXRegs; ++it) id[it] = first + row0 + (threadIdx.x + it * NT) / d.ld1;
        id[kXRegs] = first + ir;
    }
}
__device__ __forceinline__ int64_t prefetch_tile_ids_done(int64_t (&xs)[kXRegs], const int64_t (&id)[kXRegs + 1], const Dims &d,
                                                          int64_t row0, int64_t M) {
#pragma unroll
    for (int it = 0; it < kXRegs; ++it) {
        const int e = threadIdx.x + it * NT, r = e / d.ld1, c = e - r * d.ld1;
        xs[it] = (e < R * d.ld1 && c < d.D && row0 + r < M) ? id[it] * d.D + c : -1;
    }
    return row0 + (threadIdx.x >> 4) < M ? id[kXRegs] : -1;
}
// The same without a branch anywhere: every id was loaded at a clamped (valid) row, so id * D + min(c, D - 1) is a readable element
// whatever the lane; `ok` keeps which of them the tile really holds.  (With the -1 form above hipcc turns the select around the
// 64-bit multiply into a branch and SINKS the id's load into it, behind whatever was issued in between.)
__device__ __forceinline__ int64_t prefetch_tile_ids_done_nb(int64_t (&xs)[kXRegs], unsigned &ok, const int64_t (&id)[kXRegs + 1],
                                                             const Dims &d, int64_t row0, int64_t M) {
    ok = 0;
#pragma unroll
    for (int it = 0; it < kXRegs; ++it) {
        const int e = threadIdx.x + it * NT, r = e / d.ld1, c = e - r * d.ld1;
        xs[it] = id[it] * d.D + (c < d.D ? c : 0);
        ok |= (unsigned)(e < R * d.ld1 && c < d.D && row0 + r < M) << it;
    }
    return row0 + (threadIdx.x >> 4) < M ? id[kXRegs] : -1;
}
__device__ __forceinline__ void prefetch_tile_vals_nb(float (&xr)[kXRegs], const int64_t (&xs)[kXRegs], unsigned ok,
                                                      const float *__restrict__ obs) {
#pragma unroll
    for (int it = 0; it < kXRegs; ++it) {
        const float v = obs[xs[it]];
        xr[it] = (ok >> it & 1u) ? v : 0.f;
    }
}
__device__ __forceinline__ void prefetch_tile_ids(int64_t (&xs)[kXRegs], const Dims &d,
                                                  const int64_t *__restrict__ perm, int64_t first, int64_t row0,
                                                  int64_t M) {
    int64_t id[kXRegs + 1];
    prefetch_tile_ids_load(id, d, perm, first, row0, M);
    (void)prefetch_tile_ids_done(xs, id, d, row0, M);
}

__device__ __forceinline__ void prefetch_tile_vals(float (&xr)[kXRegs], const int64_t (&xs)[kXRegs],
                                                   const float *__restrict__ obs) {
#pragma unroll
    for (int it = 0; it < kXRegs; ++it) {   // (unconditional loads at a clamped offset + a select: one batch, no branches)
        const float v = obs[xs[it] >= 0 ? xs[it] : 0];
        xr[it] = xs[it] >= 0 ? v : 0.f;
    }
}

__device__ __forceinline__ void prefetch_tile_x(float (&xr)[kXRegs], const Dims &d, const float *__restrict__ obs,
                                                const int64_t *__restrict__ perm, int64_t first, int64_t row0,
                                                int64_t M) {
    int64_t xs[kXRegs];
    prefetch_tile_ids(xs, d, perm, first, row0, M);
    prefetch_tile_vals(xr, xs, obs);
}

// ... and commit it to LDS once the previous tile no longer needs the buffer.
__device__ __forceinline__ void commit_tile_x(float *lds, int X, const Dims &d, const float (&xr)[kXRegs]) {
#pragma unroll
    for (int it = 0; it < kXRegs; ++it) {
        const int e = threadIdx.x + it * NT;
        if (e < R * d.ld1) lds[X + e] = xr[it];
    }
}

__device__ __forceinline__ void load_tile_x(float *lds, int X, const Dims &d, const float *__restrict__ obs,
                                            const int64_t *__restrict__ perm, int64_t first, int64_t row0,
                                            int64_t M) {
    float xr[kXRegs];
    prefetch_tile_x(xr, d, obs, perm, first, row0, M);
    commit_tile_x(lds, X, d, xr);
}

static int make_dims(int32_t obs_dim, int32_t hidden, int32_t n_act, Dims *d) {
    TSM_REQUIRE(hidden == 64, "fused MLP supports hidden == 64 (got %d)", hidden);
    TSM_REQUIRE(obs_dim >= 1 && obs_dim <= 16 * kMaxJ, "fused MLP supports 1 <= obs_dim <= %d (got %d)", 16 * kMaxJ,
                obs_dim);
    TSM_REQUIRE(n_act >= 1 && n_act <= 16, "fused MLP supports 1 <= n_act <= 16 (got %d)", n_act);
    d->D = obs_dim;
    d->A = n_act;
    d->Kp1 = (obs_dim + 3) / 4 * 4;
    d->nJ = (obs_dim + 15) / 16;
    d->ld1 = 16 * d->nJ + 2;
    return TSM_OK;
}


}  // namespace
