// common.h -- shared helpers for the gfx950 kernels behind include/tsmarl.h
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/tsmarl.h"

#define TSM_EXPORT extern "C" __attribute__((visibility("default")))

// thread-local last-error message (abi.hip)
void tsm_set_error(const char *fmt, ...);

#define TSM_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            tsm_set_error(__VA_ARGS__);   \
            return TSM_ERR_INVALID;       \
        }                                 \
    } while (0)

#define TSM_HIP(call)                                                                  \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            tsm_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),       \
                          __FILE__, __LINE__);                                         \
            return TSM_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

#define TSM_LAUNCH_CHECK()                                                             \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess) {                                                        \
            tsm_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_),   \
                          __FILE__, __LINE__);                                         \
            return TSM_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

// kernel selection options (abi.hip: environment default, tsm_kernel_option_set override)
enum { TSM_OPT_ACTOR_TILE = 0, TSM_OPT_SPLIT_BF16 = 1, TSM_OPT_GENERIC = 2, TSM_OPT_ROLLOUT_FORM = 3, TSM_OPT_COUNT };
int tsm_opt(int id);

static inline hipStream_t tsm_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;  // gfx950 wavefront

// Dynamic LDS of a workgroup on gfx950: 160 KB.  ONE constant for the host-side layout checks and for the value handed to
// hipFuncAttributeMaxDynamicSharedMemorySize (a layout that passes the check can always be launched).
constexpr size_t kTsmMaxLds = 160 * 1024;

// Raise a kernel's dynamic-LDS limit to all the LDS a workgroup can have beside the kernel's own static __shared__
// variables (the runtime refuses static + dynamic > 160 KB with "invalid argument").  A layout above that room still fails
// loudly: the launch returns an error that TSM_LAUNCH_CHECK reports.
static inline hipError_t tsm_allow_max_lds(const void *kernel) {
    hipFuncAttributes a{};
    hipError_t e = hipFuncGetAttributes(&a, kernel);
    if (e != hipSuccess) return e;
    const size_t room = a.sharedSizeBytes < kTsmMaxLds ? kTsmMaxLds - a.sharedSizeBytes : 0;
    return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)room);
}

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Stage a row-major global matrix (rows_ok x cols_ok, row pitch src_ld) into an LDS array of n_dst floats laid out in rows of
// `ld` floats; everything outside the matrix is zeroed.  EIGHT loads per thread are in flight before the first LDS store: the
// plain `for (e = tid; ...) lds[e] = src[...]` form of this loop pays one memory round trip per iteration (the compiler cannot
// hoist loads over LDS stores it cannot prove disjoint) -- 13 iterations for a 128 x 48 layer, most of an 8 us kernel prologue.
// All threads of the NT-thread workgroup call it; no barrier inside.
template <int NT>
__device__ __forceinline__ void tsm_stage_padded(float *dst, const float *__restrict__ src, int n_dst, int ld, int rows_ok,
                                                 int cols_ok, int src_ld) {
    for (int e0 = 0; e0 < n_dst; e0 += 8 * NT) {
        float q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + (int)threadIdx.x + u * NT, r = e / ld, c = e - r * ld;
            const bool ok = e < n_dst && r < rows_ok && c < cols_ok;
            const float v = src[ok ? r * src_ld + c : 0];   // clamped, always-valid address + select: no divergent branch
            q[u] = ok ? v : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + (int)threadIdx.x + u * NT;
            if (e < n_dst) dst[e] = q[u];
        }
    }
}

// ---- wave / block reductions (64-wide) ----
// 16-lane rows of a wave are DPP rows: row-wide rotate / broadcast are plain VALU operand modifiers (no LDS crossbar
// round trip as for ds_bpermute).  dpp_ctrl: row_ror:n = 0x120 + n, row_newbcast:n = 0x150 + n (gfx90a+).
template <int CTRL>
__device__ __forceinline__ float row_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// All-reduce over the 16 lanes of a row by rotations 8, 4, 2, 1.  After the first step the values have period 8, so
// every rotation pairs the same operands as the xor butterfly __shfl_xor(v, 8 / 4 / 2 / 1): bit-identical results.
__device__ __forceinline__ float row16_sum(float v) {
    v += row_dpp<0x128>(v); v += row_dpp<0x124>(v); v += row_dpp<0x122>(v); v += row_dpp<0x121>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, row_dpp<0x128>(v)); v = fmaxf(v, row_dpp<0x124>(v));
    v = fmaxf(v, row_dpp<0x122>(v)); v = fmaxf(v, row_dpp<0x121>(v));
    return v;
}

// Ordered folds over the 16 lanes of a row (lane j holds term j; A <= 16 terms, wave-uniform): the same additions in the
// same order as a serial loop over j, so results are bit-identical to the one-lane form.
//   row_prefix_sum: s += t[0] + t[1] + ... + t[A - 1]        row_prefix_sub: s = ((s - t[0]) - t[1]) - ...
template <int J>
__device__ __forceinline__ void row_prefix_sum(float t, int A, float &s) {
    if constexpr (J < 16) {
        if (J < A) {
            s += row_dpp<0x150 + J>(t);
            row_prefix_sum<J + 1>(t, A, s);
        }
    }
}
template <int J>
__device__ __forceinline__ void row_prefix_sub(float t, int A, float &s) {
    if constexpr (J < 16) {
        if (J < A) {
            s -= row_dpp<0x150 + J>(t);
            row_prefix_sub<J + 1>(t, A, s);
        }
    }
}
// inverse-CDF pick: first j with u < t[0] + ... + t[j] (same running sum as row_prefix_sum)
template <int J>
__device__ __forceinline__ void row_cdf_pick(float t, int A, float u, float &cs, int &act, bool &found) {
    if constexpr (J < 16) {
        if (J < A) {
            cs += row_dpp<0x150 + J>(t);
            if (!found && u < cs) { act = J; found = true; }
            row_cdf_pick<J + 1>(t, A, u, cs, act, found);
        }
    }
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// sum over a block of NT threads (NT multiple of 64, <= 1024); result valid in every thread
template <typename T, int NT>
__device__ __forceinline__ T block_sum(T v, T *smem /* NT/64 entries */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem[w] = v;
    __syncthreads();
    T r = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) r += smem[i];
    return r;
}

// inclusive prefix sum across the 64 lanes of a wave
template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        T n = __shfl_up(v, off, 64);
        if (lane >= off) v += n;
    }
    return v;
}

// block-wide exclusive scan for a 1-D block of NT threads (NT multiple of 64, <= 1024).
// Returns this thread's exclusive prefix; *total receives the block sum (valid in all threads).
template <typename T, int NT>
__device__ __forceinline__ T block_exclusive_scan(T v, T *smem /* NT/64 + 1 entries */, T *total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const T inc = wave_inclusive_scan(v);
    __syncthreads();
    if (lane == 63) smem[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        T run = 0;
        for (int i = 0; i < NT / 64; ++i) { T t = smem[i]; smem[i] = run; run += t; }
        smem[NT / 64] = run;
    }
    __syncthreads();
    *total = smem[NT / 64];
    return smem[w] + inc - v;
}
