// Probe (round 5): what does a gradient all-reduce INSIDE one persistent launch cost on MI355X, against the
// two kernel boundaries + slab round trip + Adam launch it would replace in the headline gradient step (~13 us of 19.5)?
//
// n_wg persistent workgroups (one per CU, 512 threads).  Every step each workgroup produces a vector of E floats (its gradient
// contribution: the C-layout slab of csrc/mlp_fused.hip, 14 848 floats with padding) after `work_us` of stand-in compute, and
// every workgroup must end the step holding the SUM over all workgroups (so that each can apply Adam redundantly to weights it
// keeps in LDS).  Hierarchical exchange, placement-independent (every handed-off byte stored `sc1` and drained, every load of it
// an `sc1` load: MI355X_MICROARCH.md "Valid forms", row 1; R2 granules for hop 2), fixed summation order (deterministic):
//   groups c = b % 8 (blocks b, b + 8, ... usually share an XCD: a speed bonus only), rank j = b / 8
//   hop 1  reduce-scatter inside the group: every member publishes its slab (16-B sc1 stores), arrival counter of the group,
//          member j sums slice j (E / 32 floats) over the group's slabs in rank order
//   hop 2  across groups: the slice partial as 8-byte {stamp, value} granules; member j of EVERY group reads slice j of all
//          eight groups (data-tagged: no flag, no fence) and sums them in group order -> slice j of the total
//   hop 3  all-gather inside the group: slice j of the total into the group's copy, arrival counter, everyone reads all E floats
// Double-buffered by step parity (slab reuse is ordered by the counters; granules and the gathered copy need the other parity).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/xcd_exchange.hip -o tools/probes/xcd_exchange
//   tools/probes/xcd_exchange [n_wg=256] [steps=200] [work_us=5.7] [E=14848]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int kThreads = 512, kGroups = 8, kSlices = 32, kSpinMax = 1 << 22;

struct Ws {
    unsigned *cnt1, *cnt3;   // [kGroups][32] (one 128-B line per group)
    unsigned *err;           // bounded spin ran out
    float *slab;             // [n_wg][E]
    u64 *part;               // [2][kGroups][E] granules
    float *gsum;             // [2][kGroups][E]
    float *out;              // [n_wg][E] accumulated totals (verification)
    long long *stamps;       // [steps + 1] wall clock of workgroup 0
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f4 ld16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, byte_off, 0, 16);
}
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// one lane waits until *ctr >= target (relaxed agent-scope = sc1 loads); false on give-up
__device__ __forceinline__ bool wait_ge(unsigned *ctr, unsigned target, unsigned *err) {
    for (int spins = 0; spins < kSpinMax; ++spins) {
        if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}

__device__ __forceinline__ float contrib(int b, int e, int n) { return (float)((b * 131 + e * 7 + n * 3) % 13); }

__global__ __launch_bounds__(kThreads) void exchange_kernel(Ws w, int E, int n_steps, int work_ticks, int mode) {
    extern __shared__ float lds[];   // [4][Es] fold of hop 1 | [kGroups][Es] of hop 2
    const int b = blockIdx.x, n_wg = gridDim.x, t = threadIdx.x;
    const int c = b % kGroups, j = b / kGroups;
    const int K = (n_wg - c + kGroups - 1) / kGroups;        // members of my group
    auto kg = [&](int g) { return g < n_wg ? (n_wg - g + kGroups - 1) / kGroups : 0; };   // members of group g
    const int Es = E / kSlices, Es4 = Es / 4;               // floats / float4 per slice
    const auto r_slab = rsrc(w.slab), r_gsum = rsrc(w.gsum);
    const int nq = (E / 4 + kThreads - 1) / kThreads;        // float4 per thread of a whole vector (<= 8)
    f4 total[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) total[i] = f4{0.f, 0.f, 0.f, 0.f};
    __shared__ int s_dead;
    if (t == 0) s_dead = 0;
    __syncthreads();
    for (int n = 1; n <= n_steps; ++n) {
        if (b == 0 && t == 0) w.stamps[n - 1] = wall_clock64();
        // ---- stand-in for forward + loss + backward of the workgroup's tile ----
        if (t == 0) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < work_ticks) __builtin_amdgcn_s_sleep(1); }
        __syncthreads();
        if (mode == 1) continue;   // (compute only: the loop's own floor)
        const int par = n & 1;
        // ---- hop 1: publish the slab, arrive, reduce my slices over the group ----
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = t + kThreads * i;
            if (i < nq && q < E / 4) {
                const f4 v = f4{contrib(b, 4 * q, n), contrib(b, 4 * q + 1, n), contrib(b, 4 * q + 2, n), contrib(b, 4 * q + 3, n)};
                st16_sc1(r_slab, ((unsigned)b * E + 4 * q) * 4u, v);
            }
        }
        drain();
        __syncthreads();
        if (t == 0) {
            __hip_atomic_fetch_add(w.cnt1 + c * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!s_dead && !wait_ge(w.cnt1 + c * 32, (unsigned)(K * n), w.err)) s_dead = 1;
        }
        __syncthreads();
        for (int sl = j; sl < kSlices; sl += K) {
            const int col = t & 127, l4 = t >> 7;
            f4 acc = f4{0.f, 0.f, 0.f, 0.f};
            if (col < Es4) {
                f4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int m = l4 + 4 * u;
                    v[u] = f4{0.f, 0.f, 0.f, 0.f};
                    if (m < K) v[u] = ld16_sc1(r_slab, ((unsigned)(c + kGroups * m) * E + sl * Es + 4 * col) * 4u);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
                *reinterpret_cast<f4 *>(lds + (l4 * Es4 + col) * 4) = acc;
            }
            __syncthreads();
            // ---- hop 2: the slice partial as granules; sweep the eight groups' granules of this slice ----
            if (t < Es4) {
                const f4 p = ((*reinterpret_cast<f4 *>(lds + (0 * Es4 + t) * 4) + *reinterpret_cast<f4 *>(lds + (1 * Es4 + t) * 4)) +
                              *reinterpret_cast<f4 *>(lds + (2 * Es4 + t) * 4)) + *reinterpret_cast<f4 *>(lds + (3 * Es4 + t) * 4);
                u64 *g = w.part + ((size_t)(par * kGroups + c) * E + sl * Es + 4 * t);
                const float pe[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    __hip_atomic_store(g + u, ((u64)(unsigned)n << 32) | (u64)__float_as_uint(pe[u]), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();   // (the fold buffer is reused below)
            {
                const int g2 = t >> 6, lane = t & 63;   // wave g2 sweeps group g2's granules of the slice
                const u64 *g = w.part + ((size_t)(par * kGroups + g2) * E + sl * Es);
                constexpr int NG = 8;                   // granules per lane (Es <= 512)
                unsigned val[NG];
                bool dead = s_dead != 0;
                if (kg(g2) > 0) {
                    for (int spins = 0;; ++spins) {
                        bool ok = true;
#pragma unroll
                        for (int k = 0; k < NG; ++k) {
                            const int e = lane + 64 * k;
                            if (e < Es) {
                                const u64 x = __hip_atomic_load(const_cast<u64 *>(g) + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                val[k] = (unsigned)x;
                                ok &= (unsigned)(x >> 32) == (unsigned)n;
                            }
                        }
                        if (__all(ok) || dead) break;
                        if (spins > kSpinMax / 64) { __hip_atomic_store(w.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); dead = true; }
                        if ((spins & 15) == 15 && __hip_atomic_load(w.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) dead = true;
                    }
                }
#pragma unroll
                for (int k = 0; k < NG; ++k) {
                    const int e = lane + 64 * k;
                    if (e < Es) lds[g2 * Es + e] = kg(g2) > 0 ? __builtin_bit_cast(float, val[k]) : 0.f;
                }
            }
            __syncthreads();
            // ---- hop 3 (store side): slice of the total into my group's gathered copy ----
            if (t < Es4) {
                f4 s = *reinterpret_cast<f4 *>(lds + 0 * Es + 4 * t);
#pragma unroll
                for (int g2 = 1; g2 < kGroups; ++g2) s += *reinterpret_cast<f4 *>(lds + g2 * Es + 4 * t);
                st16_sc1(r_gsum, ((unsigned)(par * kGroups + c) * E + sl * Es + 4 * t) * 4u, s);
            }
            __syncthreads();
        }
        drain();
        __syncthreads();
        if (t == 0) {
            __hip_atomic_fetch_add(w.cnt3 + c * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!s_dead && !wait_ge(w.cnt3 + c * 32, (unsigned)(K * n), w.err)) s_dead = 1;
        }
        __syncthreads();
        // ---- hop 3 (load side): the whole total; stand-in for Adam on every parameter ----
        {
            f4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int q = t + kThreads * i;
                v[i] = f4{0.f, 0.f, 0.f, 0.f};
                if (i < nq && q < E / 4) v[i] = ld16_sc1(r_gsum, ((unsigned)(par * kGroups + c) * E + 4 * q) * 4u);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) total[i] += v[i];
        }
    }
    if (b == 0 && t == 0) w.stamps[n_steps] = wall_clock64();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = t + kThreads * i;
        if (i < nq && q < E / 4) *reinterpret_cast<f4 *>(w.out + (size_t)b * E + 4 * q) = total[i];
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

int main(int argc, char **argv) {
    const int n_wg = argc > 1 ? atoi(argv[1]) : 256, steps = argc > 2 ? atoi(argv[2]) : 200;
    const double work_us = argc > 3 ? atof(argv[3]) : 5.7;
    const int E = argc > 4 ? atoi(argv[4]) : 14848;
    if (n_wg < 1 || n_wg > 256 || E % (kSlices * 4) || E / kSlices > 512 || E / 4 > 8 * kThreads) { printf("bad arguments\n"); return 2; }
    Ws w;
    unsigned *ctl;
    const size_t ctl_bytes = (2 * kGroups * 32 + 32) * sizeof(unsigned);
    CK(hipMalloc(&ctl, ctl_bytes));
    w.cnt1 = ctl; w.cnt3 = ctl + kGroups * 32; w.err = ctl + 2 * kGroups * 32;
    CK(hipMalloc(&w.slab, (size_t)n_wg * E * 4));
    CK(hipMalloc(&w.part, (size_t)2 * kGroups * E * 8));
    CK(hipMalloc(&w.gsum, (size_t)2 * kGroups * E * 4));
    CK(hipMalloc(&w.out, (size_t)n_wg * E * 4));
    CK(hipMalloc(&w.stamps, (steps + 1) * 8));
    const size_t shmem = (size_t)kGroups * (E / kSlices) * 4;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e30f;
        std::vector<long long> stamps(steps + 1);
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemsetAsync(ctl, 0, ctl_bytes, st));
            CK(hipMemsetAsync(w.part, 0, (size_t)2 * kGroups * E * 8, st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(exchange_kernel, dim3(n_wg), dim3(kThreads), shmem, st, w, E, steps, (int)(work_us * 100.0), mode);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) { best = ms; CK(hipMemcpy(stamps.data(), w.stamps, (steps + 1) * 8, hipMemcpyDeviceToHost)); }
        }
        std::vector<double> d(steps);
        for (int i = 0; i < steps; ++i) d[i] = (stamps[i + 1] - stamps[i]) / 100.0;
        std::vector<double> s = d;
        std::sort(s.begin(), s.end());
        printf("%s: n_wg %d, E %d floats (%.1f KB), stand-in compute %.1f us: launch %.1f us for %d steps = %.2f us per step "
               "(workgroup 0 stamps: median %.2f, p10 %.2f, p90 %.2f, first %.2f)\n",
               mode ? "compute only   " : "compute + exchange", n_wg, E, E * 4 / 1024.0, work_us, best * 1e3, steps, best * 1e3 / steps,
               s[steps / 2], s[steps / 10], s[steps * 9 / 10], d[0]);
        if (mode == 0) {
            unsigned err = 0;
            CK(hipMemcpy(&err, w.err, 4, hipMemcpyDeviceToHost));
            std::vector<float> out((size_t)n_wg * E);
            CK(hipMemcpy(out.data(), w.out, out.size() * 4, hipMemcpyDeviceToHost));
            std::vector<double> want(E, 0.0);
            for (int n = 1; n <= steps; ++n)
                for (int b = 0; b < n_wg; ++b)
                    for (int e = 0; e < E; ++e) want[e] += (double)((b * 131 + e * 7 + n * 3) % 13);
            size_t bad = 0, by_u[4] = {0, 0, 0, 0}, by_c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int shown = 0;
            for (int b = 0; b < n_wg; ++b)
                for (int e = 0; e < E; ++e)
                    if (out[(size_t)b * E + e] != (float)want[e]) {
                        ++bad; ++by_u[e & 3]; ++by_c[b & 7];
                        if (shown++ < 6) printf("  mismatch: block %d element %d: got %.1f, want %.1f\n", b, e, out[(size_t)b * E + e], want[e]);
                    }
            if (bad) printf("  mismatches by e %% 4: %zu %zu %zu %zu; by block %% 8: %zu %zu %zu %zu %zu %zu %zu %zu\n", by_u[0], by_u[1], by_u[2],
                            by_u[3], by_c[0], by_c[1], by_c[2], by_c[3], by_c[4], by_c[5], by_c[6], by_c[7]);
            printf("  verification: error word %u, %zu of %zu words differ from the exact sums\n", err, bad, out.size());
            if (err || bad) return 1;
        }
    }
    return 0;
}
