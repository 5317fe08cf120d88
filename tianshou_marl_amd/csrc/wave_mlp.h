// wave_mlp.h -- one MLP layer on the 16 samples a WAVE owns, for the wave-autonomous rollout kernels (rollout_rows.hip:
// rollout_wave_kernel, rollout.hip: rollout_wave64_kernel).
//
// Products are formed TRANSPOSED: the weights are the A operand of v_mfma_f32_16x16x4_f32 (m = output unit; fragments read from
// LDS, where the matrix lies as [unit][k] with a row pitch of K + 2 floats -- the layout the tile kernels read it in as their B
// operand), the wave's samples are the B operand (n = sample = lane & 15, k = 4 kb + (lane >> 4)) held in REGISTERS.  Each output
// element is the chain c + sum_k a_k b_k over the same k order as in the tile kernels (the roles of the two factors are
// exchanged, the products are not): bit-identical activations (tests/test_gpu_rollout.py, tests/test_gpu_pipeline.py).
// The accumulator layout (lane: sample = lane & 15, units 16 mb + 4 (lane >> 4) + i) becomes the next layer's B fragments (unit
// 4 kb + (lane >> 4)) by a 4 x 4 transpose between register index and lane group: v_permlane32_swap + v_permlane16_swap
// (gfx950), four instructions per 16 units, no LDS round trip (tools/probes/permlane_swap.hip).
#pragma once
#include <hip/hip_runtime.h>

typedef float wf4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ wf4 wave_mfma4(float a, float b, wf4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// 4 x 4 transpose between the register index and the lane group (lane >> 4): afterwards r[j] of group g holds what r[g] of
// group j held.  (Inline asm: with this compiler the second result of __builtin_amdgcn_permlane*_swap aliases the first.)
__device__ __forceinline__ void lane_group_transpose(wf4 &r) {
    float r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\ts_nop 1\n\t"
                 "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1"
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
    r = wf4{r0, r1, r2, r3};
}

// out[unit][sample] = sum_k W[unit][k] x[k][sample] for MB blocks of 16 units; xb[kb] = the B fragment of k-step kb; only the
// first kb_n (wave-uniform, <= KB) k-steps are issued
template <int MB, int KB>
__device__ __forceinline__ void wave_layer(const float *__restrict__ wfrag /* W + (lane & 15) * ld + (lane >> 4) */, int ld,
                                           const float (&xb)[KB], int kb_n, wf4 (&acc)[MB]) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = wf4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        if (kb < kb_n) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[mb] = wave_mfma4(wfrag[16 * mb * ld + 4 * kb], xb[kb], acc[mb]);
        }
    }
}

// bias + ReLU in the accumulator layout (FMAX: fmaxf(v, 0) as mlp_tile.h writes it, else v > 0 ? v : 0 as rollout_rows.hip does)
template <int MB, bool FMAX>
__device__ __forceinline__ void wave_bias_relu(const float *__restrict__ bias /* b + 4 * (lane >> 4) */, wf4 (&acc)[MB]) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const wf4 b = *reinterpret_cast<const wf4 *>(bias + 16 * mb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = acc[mb][i] + b[i];
            acc[mb][i] = FMAX ? fmaxf(v, 0.f) : (v > 0.f ? v : 0.f);
        }
    }
}

// ... then the accumulators of 16 units become four B fragments of the next layer
template <int MB>
__device__ __forceinline__ void wave_to_frags(wf4 (&acc)[MB], float (&hb)[4 * MB]) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        lane_group_transpose(acc[mb]);
#pragma unroll
        for (int j = 0; j < 4; ++j) hb[4 * mb + j] = acc[mb][j];
    }
}
