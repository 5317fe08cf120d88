// critic_rows_dev.h -- pieces shared by the one-launch critic kernels (critic_rows.hip: forward; critic_train.hip: gradient step)
#pragma once
#include "common.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kH = 128;        // hidden width
constexpr int kRows = 32;      // rows per tile (two 16-row MFMA tiles)
constexpr int kThreads = 512;  // 8 waves
constexpr int kLdh = kH + 2;   // 130 = 2 x odd: conflict-free [lane & 15][lane >> 4] operand reads

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// LDS region offsets beyond 64 KB do not fit the 16-bit offset field of a DS instruction.  Left as compile-time constants,
// the compiler folds each region offset with every access's own constant and then keeps ONE address register PER ACCESS
// (hundreds of them, hoisted out of the tile loop: 230 VGPRs for the smallest instantiation).  Passing the offsets through an
// empty asm makes them opaque scalar values: region base + lane part is one add, the per-access constants stay immediates.
__device__ __forceinline__ int opaque_s(int x) {
    asm volatile("" : "+s"(x));
    return x;
}

// The same for a per-lane value: what is computed from the result is recomputed where it is used instead of being hoisted
// out of the tile loop and kept (or spilled) across every phase -- registers are the scarce resource of these kernels.
__device__ __forceinline__ int opaque_v(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

// swizzled position (in floats) of 16-B chunk c of tile row r
__device__ __forceinline__ int xs_off(int r, int c, int ldx) { return r * ldx + (((c & ~15) | ((c & 15) ^ (r & 15))) << 2); }

// The same positions written so that the compiler sees "per-lane base + compile-time constant" (it does not find the
// bit-separable structure of the XOR by itself and otherwise keeps one address register per access):
//   A operand of layer 1, lane (c16, kq), row half h, k-group j: chunk 4 j + kq of row 16 h + c16 lives at
//       xa_base(c16, kq, ldx, j & 3) + 64 (j >> 2) + 16 h ldx
//   B operand of the weight gradient, lane (c16, kq), rows r0 + kq (r0 a multiple of 4), column 16 ti + c16:
//       xb_base(c16, kq, ldx) + r0 ldx + 64 (ti >> 2) + 16 ((ti & 3) ^ ((r0 >> 2) & 3))
__device__ __forceinline__ int xa_base(int c16, int kq, int ldx, int m) {
    return c16 * ldx + ((((m ^ (c16 >> 2)) << 2) | (kq ^ (c16 & 3))) << 2);
}
__device__ __forceinline__ int xb_base(int c16, int kq, int ldx) { return kq * ldx + (((c16 >> 2) ^ kq) << 2) + (c16 & 3); }

// per-lane load of W1[col][16 j + 4 kq .. + 3] (the B-operand fragment of k-group j) from a row `src` of K1 floats:
// clamped, always-valid address + select, so that a batch of them is in flight at once
template <bool VEC>
__device__ __forceinline__ f4 load_w1_frag(const float *__restrict__ src, int k, int K1) {
    if constexpr (VEC) {
        const int kc = k < K1 ? k : K1 - 4;
        const float4 q = *reinterpret_cast<const float4 *>(src + kc);
        return k < K1 ? f4{q.x, q.y, q.z, q.w} : f4{0.f, 0.f, 0.f, 0.f};
    } else {
        f4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = src[k + i < K1 ? k + i : K1 - 1];
            r[i] = k + i < K1 ? v : 0.f;
        }
        return r;
    }
}

// W2 [128][128] -> LDS rows of kLdh floats: 8 16-B loads per thread, all in flight (global memory takes them at any
// 4-B alignment), 8-B LDS stores (kLdh is even); pad columns zeroed.  Two halves, so that a caller can put other loads
// between the request and the use.
__device__ __forceinline__ void w2_load(float4 (&q)[8], const float *__restrict__ src) {
#pragma unroll
    for (int u = 0; u < 8; ++u) q[u] = reinterpret_cast<const float4 *>(src)[threadIdx.x + u * kThreads];
}
__device__ __forceinline__ void w2_store(float *dst, const float4 (&q)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e4 = threadIdx.x + u * kThreads, r = e4 >> 5, c = (e4 & 31) * 4;
        float *p = dst + r * kLdh + c;
        *reinterpret_cast<float2 *>(p) = make_float2(q[u].x, q[u].y);
        *reinterpret_cast<float2 *>(p + 2) = make_float2(q[u].z, q[u].w);
    }
    for (int e = threadIdx.x; e < kH * 2; e += kThreads) dst[(e >> 1) * kLdh + kH + (e & 1)] = 0.f;
}
__device__ __forceinline__ void stage_w2_rows(float *dst, const float *__restrict__ src) {
    float4 q[8];
    w2_load(q, src);
    w2_store(dst, q);
}

int n_cu_dev() {
    static int cached = 0;
    if (!cached) {
        hipDeviceProp_t p;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cached = p.multiProcessorCount;
        if (cached <= 0) cached = 256;
    }
    return cached;
}

constexpr size_t kMaxLds = kTsmMaxLds;

// smallest instantiated KJ (k-groups of 16) covering in_dim; 0 = unsupported
int pick_kj(int in_dim) {
    static const int inst[] = {1, 2, 3, 4, 6, 8, 12, 16, 24};
    for (int kj : inst)
        if (16 * kj >= in_dim) return kj;
    return 0;
}

}  // namespace
