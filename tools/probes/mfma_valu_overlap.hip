// NOTE (round 5): the MFMA lines of this probe contain 64 v_accvgpr copies per 64 MFMAs -- the role of a wave is chosen INSIDE the
// loop, so the accumulators cross a branch every repetition.  Its "all 8 waves: MFMA only" 1.98 us (37 cycles per MFMA) is therefore
// NOT the pipe's issue rate: tools/probes/mfma_f32_issue.hip measures 32.2 cycles at 1, 2 and 4 waves per SIMD and repeats the
// overlap question in clean loops (0.856 + 0.360 -> 1.200 us: the conclusion stands).
// (also: the same question for a bf16 MFMA, v_mfma_f32_16x16x16_bf16)
// Probe: do the MFMAs of one wave and the VALU instructions of ANOTHER wave of the same SIMD overlap on gfx950?
// One workgroup of 512 threads per CU (two waves per SIMD).  Modes: every wave runs `mf` blocks of 64 MFMAs
// (v_mfma_f32_16x16x4_f32, 8 independent accumulators) and / or `va` blocks of 64 x 8 dependent-free v_fma_f32.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_overlap.hip -o tools/probes/mfma_valu_overlap
// (Built with -mllvm -amdgpu-mfma-vgpr-form=0 the compiler itself keeps the accumulators in AccVGPRs: the same times, 1.151 / 1.969 us
//  per repetition for one / two waves per SIMD -- the register class of the accumulator does not change the pipe's rate.)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma_block(f4 (&acc)[8], float a, float b) {
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 7], 0, 0, 0);
}
// integer VALU work: 512 dependent-free v_xor / v_add pairs
__device__ __forceinline__ void ialu_block(unsigned (&u)[8], unsigned s) {
#pragma unroll
    for (int i = 0; i < 256; ++i) u[i & 7] = (u[i & 7] ^ s) + 0x9E3779B9u;
}
// the same f32 MFMAs with the accumulators in AccVGPRs (inline asm, "a" constraint)
__device__ __forceinline__ void mfma_block_acc(f4 (&acc)[8], float a, float b) {
#pragma unroll
    for (int i = 0; i < 64; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i & 7]) : "v"(a), "v"(b));
}
typedef short s4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma_block_bf16(f4 (&acc)[8], float a, float b) {   // v_mfma_f32_16x16x16_bf16
    const s4 av = {(short)__builtin_bit_cast(int, a), 1, 2, 3}, bv = {(short)__builtin_bit_cast(int, b), 3, 2, 1};
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, bv, acc[i & 7], 0, 0, 0);
}
__device__ __forceinline__ void valu_block(float (&v)[8], float s) {
#pragma unroll
    for (int i = 0; i < 512; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], s, 1.0f);
}
// role of a wave: bit 0 = MFMA work, bit 1 = VALU work; mode picks the roles of waves 0-3 / 4-7
__global__ __launch_bounds__(512) void k(float *out, int reps, int role_lo, int role_hi, long long *clk) {
    const int w = threadIdx.x >> 6;
    const int role = w < 4 ? role_lo : role_hi;
    f4 acc[8];
    float v[8];
    unsigned u[8];
    for (int i = 0; i < 8; ++i) u[i] = threadIdx.x * 7 + i;
    for (int i = 0; i < 8; ++i) { acc[i] = f4{0.f, 0.f, 0.f, 0.f}; v[i] = (float)threadIdx.x; }
    const float a = out[threadIdx.x], b = out[512 + threadIdx.x];
    __syncthreads();
    const long long c0 = (long long)__builtin_amdgcn_s_memtime();   // shader-clock ticks (guide: in-kernel clock = d memtime / d memrealtime x 100 MHz)
    const long long t0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
        if (role & 1) mfma_block(acc, a, b);
        if (role & 4) mfma_block_bf16(acc, a, b);
        if (role & 8) mfma_block_acc(acc, a, b);
        if (role & 16) ialu_block(u, __builtin_bit_cast(unsigned, a) | 5u);
        if (role & 2) valu_block(v, a);
    }
    __syncthreads();
    const long long t1 = wall_clock64();
    const long long c1 = (long long)__builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i] + (float)u[i];
    out[1024 + blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = c1 - c0; }
}
int main() {
    float *d; long long *c, h;
    if (hipMalloc(&d, (1024 + 256 * 512) * 4) != hipSuccess || hipMalloc(&c, 16) != hipSuccess) return 2;
    if (hipMemset(d, 0, (1024 + 256 * 512) * 4) != hipSuccess) return 2;
    const int reps = 200;
    struct { const char *name; int lo, hi; } modes[] = {
        {"all 8 waves: MFMA only", 1, 1}, {"all 8 waves: VALU only", 2, 2}, {"all 8 waves: MFMA then VALU (each rep)", 3, 3},
        {"waves 0-3 MFMA, waves 4-7 VALU", 1, 2}, {"waves 0-3 MFMA only (4-7 idle)", 1, 0}, {"waves 0-3 VALU only (4-7 idle)", 2, 0},
        {"waves 0-3 MFMA+VALU, 4-7 idle", 3, 0},
        {"waves 0-3 bf16 MFMA only (4-7 idle)", 4, 0}, {"all 8 waves: bf16 MFMA only", 4, 4}, {"waves 0-3 bf16 MFMA, waves 4-7 VALU", 4, 2},
        {"waves 0-3 bf16 MFMA, waves 4-7 f32 MFMA", 4, 1},
        {"waves 0-3 MFMA (AccVGPR accumulators) only", 8, 0}, {"waves 0-3 MFMA (AccVGPR), waves 4-7 VALU", 8, 2},
        {"all 8 waves: MFMA (AccVGPR) then VALU", 10, 10},
        {"waves 0-3 integer VALU only (4-7 idle)", 16, 0}, {"waves 0-3 MFMA, waves 4-7 integer VALU", 1, 16},
        {"waves 0-3 bf16 MFMA, waves 4-7 integer VALU", 4, 16}};
    for (int it = 0; it < 30; ++it) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, reps, 3, 3, c);   // clocks up
    for (auto &m : modes) {
        long long best = 1ll << 60;
        for (int it = 0; it < 6; ++it) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, reps, m.lo, m.hi, c);
            if (hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
            if (h < best) best = h;
        }
        printf("%-46s %7.3f us per rep (min of 6 launches)\n", m.name, best / 100.0 / reps);
    }
    // The clock itself (round 5; MI355X_MICROARCH.md "DVFS give-back" item 6): shader-clock ticks (s_memtime) over 100 MHz ticks
    // (s_memrealtime) around the loop, after >= 2 s of back-to-back launches of the same mode on all 256 CUs.
    struct { const char *name; int lo, hi; } cm[] = {{"all 8 waves: MFMA only", 1, 1}, {"waves 0-3 MFMA only (4-7 idle)", 1, 0},
                                                     {"all 8 waves: MFMA then VALU (each rep)", 3, 3}, {"all 8 waves: VALU only", 2, 2}};
    for (auto &m : cm) {
        const int big = 20000;
        double mhz[8], us_rep[8];
        for (int it = 0; it < 60; ++it) {   // ~40 ms per launch
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, big, m.lo, m.hi, c);
            if (it >= 52) {
                long long hh[2];
                if (hipMemcpy(hh, c, 16, hipMemcpyDeviceToHost) != hipSuccess) return 2;
                mhz[it - 52] = (double)hh[1] / (double)hh[0] * 100.0;
                us_rep[it - 52] = hh[0] / 100.0 / big;
            }
        }
        double a = 0, b = 0;
        for (int i = 0; i < 8; ++i) { a += mhz[i]; b += us_rep[i]; }
        printf("clock under sustained load: %-40s %7.1f MHz (s_memtime / s_memrealtime), %7.3f us per rep (mean of the last 8 of 60 launches of %d reps)\n",
               m.name, a / 8, b / 8, big);
    }
    return 0;
}
