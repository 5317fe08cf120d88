// Probe: cycles per f32 MFMA on one SIMD of gfx950 by instruction shape and waves per SIMD.
//   v_mfma_f32_16x16x4_f32 (2048 flop, 32 cycles per SIMD nominal) vs v_mfma_f32_32x32x2_f32 (4096 flop, 64 cycles nominal),
//   1 / 2 / 4 waves per SIMD (256 / 512 / 1024 threads per CU), independent accumulators, operands in registers.
// Question behind it: profiles/r05_probe_mfma_valu_overlap.txt measured 37 cycles per 16x16x4 at two waves per SIMD
// (0.86 of nominal) and 43 at one; is the bubble per INSTRUCTION (then the 32x32x2 shape halves it per flop) and does a third /
// fourth wave hide it?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f32_issue.hip -o /tmp/mfma_f32_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int SHAPE, int NACC>
__global__ __launch_bounds__(1024) void k(float *out, int reps, long long *clk) {
    const float a = out[threadIdx.x], b = out[1024 + threadIdx.x];
    float s = 0.f;
    __syncthreads();
    const long long c0 = (long long)__builtin_amdgcn_s_memtime();   // shader-clock ticks; wall_clock64 = s_memrealtime, 100 MHz
    const long long t0 = wall_clock64();
    if (SHAPE == 0) {
        f4 acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < reps; ++r) {
#pragma unroll
            for (int i = 0; i < 64; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i % NACC], 0, 0, 0);
        }
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        f16v acc[NACC];
        for (int i = 0; i < NACC; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int r = 0; r < reps; ++r) {
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i % NACC], 0, 0, 0);
        }
        for (int i = 0; i < NACC; ++i)
            for (int j = 0; j < 16; ++j) s += acc[i][j];
    }
    __syncthreads();
    const long long t1 = wall_clock64();
    const long long c1 = (long long)__builtin_amdgcn_s_memtime();
    out[2048 + blockIdx.x * 1024 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = c1 - c0; }
}

// Do the f32 MFMAs of one wave and the f32 FMAs of the OTHER wave of the same SIMD overlap?  512 threads: waves 0-3 (one per SIMD) run
// `mf` (64 MFMAs per repetition, 2048 cycles at the nominal rate), waves 4-7 run `va` (384 independent v_fma_f32 per repetition, 1536
// cycles of VALU issue).  Each role has its own loop (the round-4 probe chose the role INSIDE one loop and paid 64 accumulator
// copies per repetition for it: its "0.86 / 0.74 of the nominal rate" was that, profiles/r05_probe_mfma_f32_issue.txt).
__global__ __launch_bounds__(512) void k2(float *out, int reps, int role_lo, int role_hi, long long *clk) {
    const int w = threadIdx.x >> 6, role = w < 4 ? role_lo : role_hi;
    const float a = out[threadIdx.x], b = out[1024 + threadIdx.x];
    float s = 0.f;
    __syncthreads();
    const long long t0 = wall_clock64();
    if (role == 1) {
        f4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < reps; ++r) {
#pragma unroll
            for (int i = 0; i < 64; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 7], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if (role == 2) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = a + (float)i;
        for (int r = 0; r < reps; ++r) {
#pragma unroll
            for (int i = 0; i < 384; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], b, 1.0f);
        }
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    __syncthreads();
    const long long t1 = wall_clock64();
    out[2048 + blockIdx.x * 1024 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

template <int SHAPE, int NACC>
static int run(const char *name, float *d, long long *c) {
    const int reps = 400;
    for (int threads : {256, 512, 1024}) {
        long long best = 1ll << 60, h;
        for (int it = 0; it < 8; ++it) {
            hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(256), dim3(threads), 0, 0, d, reps, c);
            if (hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
            if (h < best) best = h;
        }
        // per repetition each wave issues 64 x 2048 flop = 32 x 4096 flop; the SIMD holds threads / 256 waves
        const double us = best / 100.0 / reps, waves = threads / 256.0;
        const double n_instr = (SHAPE == 0 ? 64 : 32) * waves;
        const double cyc = us * 2400.0 / n_instr, nominal = SHAPE == 0 ? 32.0 : 64.0;
        printf("%-34s %d accumulators, %.0f wave(s) per SIMD: %7.3f us per repetition = %6.2f cycles per MFMA at 2.4 GHz (nominal %.0f: %.3f)\n",
               name, NACC, waves, us, cyc, nominal, nominal / cyc);
    }
    return 0;
}

int main() {
    float *d;
    long long *c;
    if (hipMalloc(&d, (2048 + 256 * 1024) * 4) != hipSuccess || hipMalloc(&c, 16) != hipSuccess) return 2;
    if (hipMemset(d, 0, (2048 + 256 * 1024) * 4) != hipSuccess) return 2;
    for (int it = 0; it < 30; ++it) hipLaunchKernelGGL((k<0, 8>), dim3(256), dim3(512), 0, 0, d, 400, c);   // clocks up
    int rc = 0;
    rc |= run<0, 8>("v_mfma_f32_16x16x4_f32", d, c);
    rc |= run<0, 4>("v_mfma_f32_16x16x4_f32", d, c);
    rc |= run<0, 2>("v_mfma_f32_16x16x4_f32", d, c);
    rc |= run<0, 1>("v_mfma_f32_16x16x4_f32", d, c);
    rc |= run<1, 4>("v_mfma_f32_32x32x2_f32", d, c);
    rc |= run<1, 2>("v_mfma_f32_32x32x2_f32", d, c);
    rc |= run<1, 1>("v_mfma_f32_32x32x2_f32", d, c);
    {
        struct { const char *name; int lo, hi; } modes[] = {{"waves 0-3 MFMA, waves 4-7 idle", 1, 0}, {"waves 0-3 idle, waves 4-7 FMA", 0, 2},
                                                           {"waves 0-3 MFMA, waves 4-7 FMA (same SIMDs)", 1, 2}, {"all 8 waves MFMA", 1, 1},
                                                           {"all 8 waves FMA", 2, 2}};
        for (auto &m : modes) {
            long long best = 1ll << 60, h;
            for (int it = 0; it < 8; ++it) {
                hipLaunchKernelGGL(k2, dim3(256), dim3(512), 0, 0, d, 400, m.lo, m.hi, c);
                if (hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
                if (h < best) best = h;
            }
            printf("overlap: %-44s %7.3f us per repetition (64 MFMAs = 0.853 us, 384 FMAs = 0.640 us of issue at 2.4 GHz)\n", m.name, best / 100.0 / 400);
        }
    }
    // The same instruction stream on RANDOM operands (the runs above multiply zeros), sustained: 60 launches of ~30 ms on all 256 CUs,
    // the last 8 reported with the shader clock (s_memtime ticks per 100 MHz s_memrealtime tick).
    {
        static float h[2048];
        unsigned x = 12345u;
        for (int i = 0; i < 2048; ++i) { x = x * 1664525u + 1013904223u; h[i] = (float)(int)(x >> 8) / 8388608.0f - 1.0f; }
        for (int zero = 1; zero >= 0; --zero) {
            if (zero) { if (hipMemset(d, 0, 2048 * 4) != hipSuccess) return 2; }
            else if (hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice) != hipSuccess) return 2;
            for (int threads : {256, 512}) {
                const int big = 20000;
                double us = 0, mhz = 0;
                for (int it = 0; it < 60; ++it) {
                    hipLaunchKernelGGL((k<0, 8>), dim3(256), dim3(threads), 0, 0, d, big, c);
                    if (it >= 52) {
                        long long hh[2];
                        if (hipMemcpy(hh, c, 16, hipMemcpyDeviceToHost) != hipSuccess) return 2;
                        us += hh[0] / 100.0 / big / 8;
                        mhz += (double)hh[1] / (double)hh[0] * 100.0 / 8;
                    }
                }
                const double n_instr = 64.0 * threads / 256.0;
                printf("sustained, %s operands, %d wave(s) per SIMD: %7.3f us per repetition = %6.2f cycles per MFMA at 2.4 GHz nominal (%.3f of the"
                       " nominal rate); s_memtime / s_memrealtime = %.1f MHz\n",
                       zero ? "ZERO  " : "RANDOM", threads / 256, us, us * 2400.0 / n_instr, 32.0 / (us * 2400.0 / n_instr), mhz);
            }
        }
    }
    return rc;
}
