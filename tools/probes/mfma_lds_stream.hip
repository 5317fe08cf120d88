// Probe: what rate does an f32 MFMA stream reach on gfx950 when its operands come from LDS the way csrc/actor_rows64.hip reads them?
// One workgroup of 512 threads per CU (two waves per SIMD), a 64 x 130 float activation tile H1 and H2 in LDS, per repetition the
// weight-gradient stream of the kernel's P6 (dW2 += dH2^T H1: 16 steps of 1 + 8 reads -> 8 MFMAs on 8 accumulators) or its
// input-gradient stream (dH1 = dH2 W2: 32 steps of 4 reads -> 4 MFMAs, the B operand in a register).
//   mode 0  operands in registers (no LDS): the matrix pipe's own rate
//   mode 1  dW2 stream, rows 4 s + kq per step (the kernel's): lanes 0-15 read row r, lanes 16-31 row r + 1 -> banks 2 apart, 2-way conflicts
//   mode 2  dW2 stream, rows 32 (s >> 3) + 8 kq + (s & 7): lanes 16-31 are 8 rows = 1040 floats on -> 16 banks apart, conflict-free
//   mode 3  dH1 stream ([row = c16][k = kq] reads: conflict-free at a stride of 130)
//   modes 4, 5, 6 = 1, 2, 3 with the step's reads issued one step ahead of its MFMAs
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_lds_stream.hip -o tools/probes/mfma_lds_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kLd = 130;
__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int reps, long long *clk) {
    __shared__ float H1[64 * kLd], H2[64 * kLd];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c16 = lane & 15, kq = lane >> 4, col = 16 * w + c16;
    for (int e = tid; e < 64 * kLd; e += 512) { H1[e] = out[e & 1023]; H2[e] = out[(e + 7) & 1023]; }
    __syncthreads();
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    float wreg[32];
    for (int i = 0; i < 32; ++i) wreg[i] = out[(tid + i) & 1023];
    const long long t0 = wall_clock64();
    constexpr int PF = MODE >= 4 ? 1 : 0, M = MODE >= 4 ? MODE - 3 : MODE;
    for (int r = 0; r < reps; ++r) {
        if (M == 0) {
#pragma unroll
            for (int i = 0; i < 128; ++i) acc[i & 7] = mfma4(wreg[i & 31], wreg[(i + 5) & 31], acc[i & 7]);
        } else if (M == 1 || M == 2) {
            auto row = [&](int s) { return M == 1 ? 4 * s : 32 * (s >> 3) + (s & 7); };
            const int lane_row = M == 1 ? kq : 8 * kq;
            const float *a = H2 + lane_row * kLd + col, *b = H1 + lane_row * kLd + c16;
            float x[2][9];
            auto ld = [&](int s, float (&y)[9]) {
                y[8] = a[row(s) * kLd];
#pragma unroll
                for (int ti = 0; ti < 8; ++ti) y[ti] = b[row(s) * kLd + 16 * ti];
            };
            if (PF) ld(0, x[0]);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (PF) { if (s + 1 < 16) ld(s + 1, x[(s + 1) & 1]); } else ld(s, x[s & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ti = 0; ti < 8; ++ti) acc[ti] = mfma4(x[s & 1][8], x[s & 1][ti], acc[ti]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            const float *a = H2 + c16 * kLd + kq;
            float x[2][4];
            auto ld = [&](int s, float (&y)[4]) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) y[mt] = a[mt * 16 * kLd + 4 * s];
            };
            if (PF) ld(0, x[0]);
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                if (PF) { if (s + 1 < 32) ld(s + 1, x[(s + 1) & 1]); } else ld(s, x[s & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = mfma4(x[s & 1][mt], wreg[s], acc[mt]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __syncthreads();
    const long long t1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[1024 + blockIdx.x * 512 + tid] = s;
    if (blockIdx.x == 0 && tid == 0) clk[0] = t1 - t0;
}

template <int MODE>
static int run(const char *name, float *d, long long *c) {
    const int reps = 400;
    long long best = 1ll << 60, h;
    for (int it = 0; it < 8; ++it) {
        hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, d, reps, c);
        if (hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        if (h < best) best = h;
    }
    const double us = best / 100.0 / reps, cyc = us * 2400.0 / 256.0;   // 128 MFMAs per wave and repetition, two waves per SIMD
    printf("%-72s %7.3f us per 128 MFMAs per wave = %6.2f cycles per MFMA (nominal 32: %.3f)\n", name, us, cyc, 32.0 / cyc);
    return 0;
}

int main() {
    float *d;
    long long *c;
    if (hipMalloc(&d, (1024 + 256 * 512) * 4) != hipSuccess || hipMalloc(&c, 16) != hipSuccess) return 2;
    if (hipMemset(d, 0, (1024 + 256 * 512) * 4) != hipSuccess) return 2;
    for (int it = 0; it < 30; ++it) hipLaunchKernelGGL((k<0>), dim3(256), dim3(512), 0, 0, d, 400, c);
    int rc = 0;
    rc |= run<0>("operands in registers", d, c);
    rc |= run<1>("dW2 stream, rows 4 s + kq (2-way bank conflicts), reads at use", d, c);
    rc |= run<2>("dW2 stream, rows 8 kq + s (conflict-free), reads at use", d, c);
    rc |= run<3>("dH1 stream ([c16][kq] reads, conflict-free), reads at use", d, c);
    rc |= run<4>("dW2 stream, rows 4 s + kq (2-way bank conflicts), reads one step ahead", d, c);
    rc |= run<5>("dW2 stream, rows 8 kq + s (conflict-free), reads one step ahead", d, c);
    rc |= run<6>("dH1 stream ([c16][kq] reads, conflict-free), reads one step ahead", d, c);
    return rc;
}
