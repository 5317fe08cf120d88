// adam.hip -- gradient-slab reduction + clip_grad_norm_ + Adam on one flat f32 parameter vector.
//
// Replaces Algorithm.Optimizer.step (/root/reference/tianshou/algorithm/algorithm_base.py:485-498):
// zero_grad -> backward -> nn.utils.clip_grad_norm_(params, max_grad_norm) -> torch.optim.Adam.step
// (AdamOptimizerFactory, tianshou/algorithm/optim.py:91-111; torch single-tensor Adam, amsgrad off):
//   g   = sum_slab grad_slab                      (fixed summation tree => deterministic)
//   g  *= min(1, max_norm / (||g||_2 + 1e-6))     (only when max_grad_norm > 0)
//   g  += weight_decay * p
//   m   = b1 m + (1-b1) g ;  v = b2 v + (1-b2) g^2
//   p  -= lr / (1-b1^t) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
//
// (Round 4, measured: 16-B slab loads -- four consecutive parameters per lane -- were built and dropped.  Slab rows start at
// odd float offsets (23 429 / 17 025 parameters per slab), and a dwordx4 load that is not 16-B aligned runs at a quarter of
// the rate: 26.1 us against 14.1 us for the C3 step's 54 MB of slabs; with strides padded to multiples of four it was
// 12.6 against 13.6 us -- the launch moves 4.3 TB/s either way, so what shortens it is fewer slab bytes, not wider loads.)
//
// gfx950 mapping: the slab reduction is latency-bound (n ~ 11 k parameters, up to 256 slabs), so a
// workgroup owns 64 parameters x 4 slab lanes: every wave reads 256-B coalesced rows of 64 parameters and
// the 4 waves walk disjoint slab subsets with 8 loads in flight each; partial sums meet in LDS.
// Optionally the kernel also refreshes a padded "LDS image" copy of the parameters (img[map[i]] = p[i]) that
// the fused MLP kernels stage with straight 16-B copies (csrc/mlp_fused.hip).
// 64 slab loads in flight per lane in this file's kernels (adam_dev.h): the headline optimizer launch is a latency chain of 4 x 16
// dependent-free loads per lane otherwise -- 6.05 -> 5.56 us in situ at 256 slabs of 45 KB; the same additions in the same order
#define TSM_ADAM_DEEP 1
#include "adam_dev.h"

namespace {

// pass 1 of the clipping path: g = sum of slabs -> work[0..n), per-block sum of squares -> work[n + blk]
__global__ __launch_bounds__(256) void reduce_norm_kernel(const float *__restrict__ slabs, int32_t n_slab, int64_t n,
                                                          float *__restrict__ work) {
    __shared__ float sm[256];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * kCols + lane;
    const float g = slab_sum_block(slabs, n_slab, n, i, sm, n);
    if (sl == 0) {
        if (i < n) work[i] = g;
        float q = i < n ? g * g : 0.f;
        q = wave_sum(q);
        if (lane == 0) work[n + blockIdx.x] = q;
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ slabs,
                                                   int32_t n_slab, int64_t n, float *__restrict__ m,
                                                   float *__restrict__ v, double lr_host,
                                                   const double *__restrict__ lr_dev, double beta1d, double beta2d,
                                                   int64_t step_host, const int64_t *__restrict__ step_dev,
                                                   float eps, float weight_decay, float max_norm,
                                                   const float *__restrict__ work, float *__restrict__ img,
                                                   const int32_t *__restrict__ img_map) {
    __shared__ float sm[256];
    __shared__ float s_coef;
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * kCols + lane;
    float g;
    if (max_norm > 0.f) {  // clipping path: slabs were already summed by reduce_norm_kernel
        const int nblk = (int)((n + kCols - 1) / kCols);
        float q = 0.f;
        for (int b = threadIdx.x; b < nblk; b += 256) q += work[n + b];
        q = wave_sum(q);
        if (lane == 0) sm[sl] = q;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float tot = sm[0] + sm[1] + sm[2] + sm[3];
            const float c = max_norm / (sqrtf(tot) + 1e-6f);
            s_coef = c < 1.f ? c : 1.f;
        }
        __syncthreads();
        g = i < n ? work[i] * s_coef : 0.f;
    } else {
        g = slab_sum_block(slabs, n_slab, n, i, sm, n);
    }
    if (sl != 0 || i >= n) return;
    const float pn = adam_apply(p, m, v, i, g, lr_host, lr_dev, beta1d, beta2d, step_host, step_dev, eps, weight_decay);
    if (img) img[img_map[i]] = pn;
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float *__restrict__ slabs, int32_t n_slab,
                                                           int64_t n, float scale, float *__restrict__ out) {
    __shared__ float sm[256];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * kCols + lane;
    const float g = slab_sum_block(slabs, n_slab, n, i, sm, n);
    if (sl == 0 && i < n) out[i] = g * scale;
}

__global__ void scatter_image_kernel(const float *__restrict__ p, int64_t n, const int32_t *__restrict__ map,
                                     float *__restrict__ img) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) img[map[i]] = p[i];
}


// ---- segmented forms: the flat vector is covered by up to TSM_MAX_SLAB_SEGS segments, each with its own slab array
// (actor slabs from one kernel, critic slabs from another: csrc/ppo_rows.hip) -- one launch for the whole vector ----
struct SegArgs {
    tsm_slab_seg seg[TSM_MAX_SLAB_SEGS];
    int32_t first_blk[TSM_MAX_SLAB_SEGS + 1];  // block range of every segment
    int32_t n_seg;
};

// Position of W1[o][k] (row-major [128][K1] segment, element il = o K1 + k) in the FRAGMENT IMAGE the one-launch critic
// kernels stage with coalesced 16-B loads (csrc/critic_train.hip): [wave = o / 16][k-group j = k / 16][lane = 16 (k / 4 % 4)
// + o % 16][k % 4] -- wave w's j-th load instruction reads 1 KB of consecutive bytes.
__device__ __forceinline__ int64_t frag_image_index(int64_t il, int K1, int KJ) {
    const int o = (int)(il / K1), k = (int)(il - (int64_t)o * K1);
    return ((((int64_t)(o >> 4) * KJ + (k >> 4)) * 64 + ((k >> 2) & 3) * 16 + (o & 15)) << 2) + (k & 3);
}

__device__ __forceinline__ int seg_of_block(const SegArgs &a, int blk) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < TSM_MAX_SLAB_SEGS; ++j)
        if (j < a.n_seg && blk >= a.first_blk[j]) k = j;
    return k;
}

// g[offset + i] = scale * sum of the segment's slabs; with `norm_work` also the per-block sum of squares (clip path)
__global__ __launch_bounds__(256) void reduce_segs_kernel(SegArgs a, int64_t n_total, float scale, float *__restrict__ out,
                                                          float *__restrict__ blk_sq) {
    __shared__ float sm[256];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int k = seg_of_block(a, blockIdx.x);
    const tsm_slab_seg sg = a.seg[k];
    const int64_t i = (int64_t)(blockIdx.x - a.first_blk[k]) * kCols + lane;
    const float g = slab_sum_block(sg.slabs, sg.n_slab, sg.n, i, sm, sg.stride) * scale * (sg.scale_dev ? *sg.scale_dev : 1.f);
    if (sl == 0) {
        if (i < sg.n) out[sg.offset + i] = g;
        if (blk_sq) {
            float q = i < sg.n ? g * g : 0.f;
            q = wave_sum(q);
            if (lane == 0) blk_sq[blockIdx.x] = q;
        }
    }
}

__global__ __launch_bounds__(256) void adam_segs_kernel(float *__restrict__ p, SegArgs a, int64_t n_total, float *__restrict__ m,
                                                        float *__restrict__ v, double lr_host,
                                                        const double *__restrict__ lr_dev, double beta1d, double beta2d,
                                                        int64_t step_host, const int64_t *__restrict__ step_dev,
                                                        float eps, float weight_decay, float max_norm,
                                                        const float *__restrict__ work) {
    __shared__ float sm[256];
    __shared__ float s_coef;
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int k = seg_of_block(a, blockIdx.x);
    const tsm_slab_seg sg = a.seg[k];
    const int64_t il = (int64_t)(blockIdx.x - a.first_blk[k]) * kCols + lane;
    const int64_t i = sg.offset + il;
    float g;
    if (max_norm > 0.f) {  // clipping path: reduce_segs_kernel left the summed gradient in work[0..n) and the blocks' squares behind it
        const int nblk = a.first_blk[a.n_seg];
        float q = 0.f;
        for (int b = threadIdx.x; b < nblk; b += 256) q += work[n_total + b];
        q = wave_sum(q);
        if (lane == 0) sm[sl] = q;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float tot = sm[0] + sm[1] + sm[2] + sm[3];
            const float c = max_norm / (sqrtf(tot) + 1e-6f);
            s_coef = c < 1.f ? c : 1.f;
        }
        __syncthreads();
        g = il < sg.n ? work[i] * s_coef : 0.f;
    } else {
        g = slab_sum_block(sg.slabs, sg.n_slab, sg.n, il, sm, sg.stride);
        if (sg.scale_dev) g *= *sg.scale_dev;
    }
    if (sl != 0 || il >= sg.n) return;
    const float pn = adam_apply(p, m, v, i, g, lr_host, lr_dev, beta1d, beta2d, step_host, step_dev, eps, weight_decay);
    if (sg.frag_image) sg.frag_image[frag_image_index(il, sg.frag_k1, sg.frag_kj)] = pn;
}

int fill_segs(SegArgs &a, const tsm_slab_seg *segs, int32_t n_seg, int64_t n_total) {
    if (!segs || n_seg < 1 || n_seg > TSM_MAX_SLAB_SEGS) return -1;
    a.n_seg = n_seg;
    int64_t covered = 0, blk = 0;
    for (int k = 0; k < n_seg; ++k) {
        const tsm_slab_seg &s = segs[k];
        // segments tile [0, n_total) in order: every parameter gets exactly one gradient
        if (!s.slabs || s.n < 1 || s.n_slab < 1 || s.stride < s.n || s.offset != covered) return -1;
        // a fragment image is that of a [128][frag_k1] row-major weight matrix in k-groups of 16
        if (s.frag_image && (s.frag_k1 < 1 || s.n != (int64_t)128 * s.frag_k1 || 16 * s.frag_kj < s.frag_k1)) return -1;
        a.seg[k] = s;
        a.first_blk[k] = (int32_t)blk;
        blk += ceil_div(s.n, kCols);
        covered += s.n;
    }
    for (int k = n_seg; k <= TSM_MAX_SLAB_SEGS; ++k) a.first_blk[k] = (int32_t)blk;
    return covered == n_total ? (int)blk : -1;
}

}  // namespace

// Sum per-workgroup gradient slabs into one flat gradient (fixed tree => deterministic), times `scale`.
// Used in front of the RCCL all-reduce of the data-parallel path (one flat buffer per gradient step).
TSM_EXPORT int tsm_reduce_slabs(const float *grad_slabs, int32_t n_slab, int64_t n, double scale, float *out,
                                void *stream) {
    TSM_REQUIRE(n >= 0 && n_slab >= 1, "tsm_reduce_slabs: bad sizes");
    if (n == 0) return TSM_OK;
    TSM_REQUIRE(grad_slabs && out, "tsm_reduce_slabs: null pointer");
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)ceil_div(n, kCols)), dim3(256), 0, tsm_stream(stream),
                       grad_slabs, n_slab, n, (float)scale, out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

// The same for a flat vector whose parts were differentiated by different kernels: segment k covers parameters
// [offset, offset + n) and owns n_slab slabs, `stride` floats apart.  One launch.
TSM_EXPORT int tsm_reduce_slabs_segs(const tsm_slab_seg *segs, int32_t n_seg, int64_t n, double scale, float *out,
                                     void *stream) {
    SegArgs a{};
    const int blocks = fill_segs(a, segs, n_seg, n);
    TSM_REQUIRE(blocks > 0 && out, "tsm_reduce_slabs_segs: the segments must tile [0, n) in order (1..%d of them), non-null",
                TSM_MAX_SLAB_SEGS);
    hipLaunchKernelGGL(reduce_segs_kernel, dim3((unsigned)blocks), dim3(256), 0, tsm_stream(stream), a, n, (float)scale,
                       out, (float *)nullptr);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

// tsm_adam_step over segmented slabs (one launch for actor + critic; with max_grad_norm > 0 one reduction launch in
// front of it and ONE norm over the whole vector, as clip_grad_norm_ over ActorCritic.parameters() has it).
// work: tsm_adam_work_elems(n) floats when clipping.
TSM_EXPORT int tsm_adam_step_segs(float *param, const tsm_slab_seg *segs, int32_t n_seg, int64_t n, float *exp_avg,
                                  float *exp_avg_sq, int64_t step, const int64_t *step_dev, double lr, const double *lr_dev,
                                  double beta1, double beta2, double eps, double weight_decay, double max_grad_norm,
                                  float *work, void *stream) {
    SegArgs a{};
    const int blocks = fill_segs(a, segs, n_seg, n);
    TSM_REQUIRE(blocks > 0, "tsm_adam_step_segs: the segments must tile [0, n) in order (1..%d of them), non-null",
                TSM_MAX_SLAB_SEGS);
    TSM_REQUIRE(step >= 1 || step_dev, "tsm_adam_step_segs: step = %lld", (long long)step);
    TSM_REQUIRE(param && exp_avg && exp_avg_sq, "tsm_adam_step_segs: null pointer");
    TSM_REQUIRE(max_grad_norm <= 0.0 || work, "tsm_adam_step_segs: clipping needs work[tsm_adam_work_elems(n)]");
    hipStream_t st = tsm_stream(stream);
    if (max_grad_norm > 0.0) {
        hipLaunchKernelGGL(reduce_segs_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, n, 1.0f, work, work + n);
        TSM_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(adam_segs_kernel, dim3((unsigned)blocks), dim3(256), 0, st, param, a, n, exp_avg, exp_avg_sq, lr,
                       lr_dev, beta1, beta2, step, step_dev, (float)eps, (float)weight_decay, (float)max_grad_norm, work);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

// (room for the per-block squares of the segmented form too: every segment rounds its block count up)
TSM_EXPORT int64_t tsm_adam_work_elems(int64_t n) { return n < 0 ? -1 : n + ceil_div(n > 0 ? n : 1, kCols) + TSM_MAX_SLAB_SEGS; }

TSM_EXPORT int tsm_adam_step(float *param, const float *grad_slabs, int32_t n_slab, int64_t n, float *exp_avg,
                             float *exp_avg_sq, int64_t step, const int64_t *step_dev, double lr, const double *lr_dev,
                             double beta1, double beta2, double eps, double weight_decay, double max_grad_norm, float *work,
                             float *param_image, const int32_t *image_map, void *stream) {
    TSM_REQUIRE(n >= 0 && n_slab >= 1 && (step >= 1 || step_dev), "tsm_adam_step: bad sizes n=%lld n_slab=%d step=%lld",
                (long long)n, n_slab, (long long)step);
    if (n == 0) return TSM_OK;
    TSM_REQUIRE(param && grad_slabs && exp_avg && exp_avg_sq, "tsm_adam_step: null pointer");
    TSM_REQUIRE(max_grad_norm <= 0.0 || work, "tsm_adam_step: clipping needs work[tsm_adam_work_elems(n)]");
    TSM_REQUIRE(!param_image || image_map, "tsm_adam_step: param_image needs image_map");
    hipStream_t st = tsm_stream(stream);
    const dim3 grid((unsigned)ceil_div(n, kCols));
    if (max_grad_norm > 0.0) {
        hipLaunchKernelGGL(reduce_norm_kernel, grid, dim3(256), 0, st, grad_slabs, n_slab, n, work);
        TSM_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(adam_kernel, grid, dim3(256), 0, st, param, grad_slabs, n_slab, n, exp_avg, exp_avg_sq, lr,
                       lr_dev, beta1, beta2, step, step_dev, (float)eps, (float)weight_decay, (float)max_grad_norm, work,
                       param_image, image_map);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

// img[map[i]] = param[i] (initial fill / after load_state_dict); pads of `img` must be zero already.
TSM_EXPORT int tsm_scatter_image(const float *param, int64_t n, const int32_t *image_map, float *param_image,
                                 void *stream) {
    TSM_REQUIRE(n >= 0, "tsm_scatter_image: negative n");
    if (n == 0) return TSM_OK;
    TSM_REQUIRE(param && image_map && param_image, "tsm_scatter_image: null pointer");
    hipLaunchKernelGGL(scatter_image_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, tsm_stream(stream), param,
                       n, image_map, param_image);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}
