// Probe: lane semantics of v_permlane32_swap / v_permlane16_swap on gfx950, and the 4 x 4 (register, lane group) transpose
// built from them (rollout_rows.hip: C-layout accumulators -> B-operand fragments without an LDS round trip).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/permlane_swap.hip -o /tmp/permlane_swap && /tmp/permlane_swap
#include <hip/hip_runtime.h>
#include <stdio.h>
// (inline asm: with this compiler the builtin's second result aliases the first -- both stores read the same register)
__device__ __forceinline__ void swap32(float &a, float &b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap16(float &a, float &b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__global__ void k(float *out) {
    const int l = threadIdx.x;
    float a = (float)l, b = 100.f + l;
    swap32(a, b);
    out[l] = a; out[64 + l] = b;
    a = (float)l; b = 100.f + l;
    swap16(a, b);
    out[128 + l] = a; out[192 + l] = b;
    // 4 x 4 transpose: r[i] in lane group g holds 10 * g + i  ->  afterwards r[j] in group g should hold 10 * j + g
    float r[4];
    for (int i = 0; i < 4; ++i) r[i] = 10.f * (l >> 4) + i;
    swap32(r[0], r[2]); swap32(r[1], r[3]); swap16(r[0], r[1]); swap16(r[2], r[3]);
    for (int i = 0; i < 4; ++i) out[256 + 64 * i + l] = r[i];
}
int main() {
    float *d, h[512];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 2;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    const char *names[] = {"swap32 a", "swap32 b", "swap16 a", "swap16 b", "T r0", "T r1", "T r2", "T r3"};
    for (int q = 0; q < 8; ++q) {
        printf("%-9s", names[q]);
        for (int l = 0; l < 64; l += 4) printf(" %5.0f", h[64 * q + l]);   // every 4th lane
        printf("\n");
    }
    int bad = 0;
    for (int i = 0; i < 4; ++i) for (int l = 0; l < 64; ++l) bad += h[256 + 64 * i + l] != 10.f * i + (l >> 4);
    printf("4x4 transpose: %s\n", bad ? "WRONG" : "ok");
    return bad != 0;
}
