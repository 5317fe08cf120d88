// dispatch.hip -- per-agent row dispatch for AEC-style batches (rows tagged with an agent id).
//
// Replaces `np.nonzero(batch.obs.agent_id == agent_id)[0]` and `holder.act[agent_index] = act`
// (/root/reference/tianshou/algorithm/multiagent/marl.py:148,170-180,233) -- a stable partition of
// row numbers by agent.  Bit-exact with numpy (ascending row order inside each agent).
//
// Three launches: per-block histogram -> single-block exclusive scan over (agent, block) ->
// stable scatter using wave ballots (rank of a row among equal-agent rows of lower lane id).
#include "common.h"

namespace {

constexpr int kRowsPerBlk = 1024;

__global__ __launch_bounds__(1024) void hist_kernel(const int32_t *__restrict__ agent_id, int64_t B,
                                                    int32_t n_agent, int64_t n_blocks,
                                                    int64_t *__restrict__ scratch) {
    extern __shared__ int s_cnt[];
    for (int a = threadIdx.x; a < n_agent; a += blockDim.x) s_cnt[a] = 0;
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * kRowsPerBlk + threadIdx.x;
    if (r < B) {
        const int32_t a = agent_id[r];
        if (a >= 0 && a < n_agent) atomicAdd(&s_cnt[a], 1);
    }
    __syncthreads();
    for (int a = threadIdx.x; a < n_agent; a += blockDim.x)
        scratch[(int64_t)a * n_blocks + blockIdx.x] = s_cnt[a];
}

__global__ __launch_bounds__(1024) void scan_kernel(int64_t *__restrict__ scratch, int64_t n,
                                                    int64_t n_blocks, int32_t n_agent,
                                                    int64_t *__restrict__ offsets_out) {
    __shared__ int64_t sm[1024 / 64 + 1];
    int64_t base = 0;
    for (int64_t c = 0; c < n; c += 1024) {
        const int64_t i = c + threadIdx.x;
        const int64_t v = i < n ? scratch[i] : 0;
        int64_t total;
        const int64_t ex = block_exclusive_scan<int64_t, 1024>(v, sm, &total);
        if (i < n) {
            scratch[i] = base + ex;
            if (i % n_blocks == 0) offsets_out[i / n_blocks] = base + ex;
        }
        base += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) { scratch[n] = base; offsets_out[n_agent] = base; }
}

__global__ __launch_bounds__(1024) void scatter_kernel(const int32_t *__restrict__ agent_id, int64_t B,
                                                       int32_t n_agent, int64_t n_blocks,
                                                       const int64_t *__restrict__ scratch,
                                                       int64_t *__restrict__ index_out) {
    extern __shared__ int s_wave_cnt[];  // [16 waves][n_agent]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * kRowsPerBlk + threadIdx.x;
    int32_t a = -1;
    if (r < B) { a = agent_id[r]; if (a < 0 || a >= n_agent) a = -1; }
    int rank_in_wave = 0;
    // one ballot per distinct agent present in the wave (wave-uniform loop)
    unsigned long long todo = __ballot(a >= 0);
    for (int aa = lane; aa < n_agent; aa += 64) s_wave_cnt[w * n_agent + aa] = 0;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int32_t cur = __shfl(a, leader, 64);
        const unsigned long long m = __ballot(a == cur);
        if (a == cur) rank_in_wave = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == leader) s_wave_cnt[w * n_agent + cur] = __popcll(m);
        todo &= ~m;
    }
    __syncthreads();
    if (a >= 0) {
        int before = 0;
        for (int ww = 0; ww < w; ++ww) before += s_wave_cnt[ww * n_agent + a];
        index_out[scratch[(int64_t)a * n_blocks + blockIdx.x] + before + rank_in_wave] = r;
    }
}

template <bool SCATTER>
__global__ void rows_kernel(const char *__restrict__ src, const int64_t *__restrict__ index, int64_t n,
                            int64_t row_bytes, char *__restrict__ dst, int word) {
    const int64_t wpr = row_bytes / word;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * wpr) return;
    const int64_t r = i / wpr, o = i - r * wpr;
    const int64_t sr = SCATTER ? r : index[r], dr = SCATTER ? index[r] : r;
    if (word == 4)
        reinterpret_cast<uint32_t *>(dst + dr * row_bytes)[o] = reinterpret_cast<const uint32_t *>(src + sr * row_bytes)[o];
    else
        dst[dr * row_bytes + o] = src[sr * row_bytes + o];
}

}  // namespace

TSM_EXPORT int tsm_agent_index(const int32_t *agent_id, int64_t B, int32_t n_agent, int64_t *index_out,
                               int64_t *offsets_out, int64_t *scratch, void *stream) {
    TSM_REQUIRE(B >= 0 && n_agent >= 1 && n_agent <= 1024, "tsm_agent_index: bad sizes B=%lld n_agent=%d",
                (long long)B, n_agent);
    TSM_REQUIRE(offsets_out && scratch, "tsm_agent_index: null pointer");
    hipStream_t st = tsm_stream(stream);
    const int64_t n_blocks = B > 0 ? ceil_div(B, kRowsPerBlk) : 1;
    if (B == 0) {
        TSM_HIP(hipMemsetAsync(offsets_out, 0, sizeof(int64_t) * (n_agent + 1), st));
        return TSM_OK;
    }
    TSM_REQUIRE(agent_id && index_out, "tsm_agent_index: null pointer");
    hipLaunchKernelGGL(hist_kernel, dim3((unsigned)n_blocks), dim3(1024), n_agent * sizeof(int), st, agent_id, B,
                       n_agent, n_blocks, scratch);
    TSM_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, scratch, (int64_t)n_agent * n_blocks, n_blocks,
                       n_agent, offsets_out);
    TSM_LAUNCH_CHECK();
    hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)n_blocks), dim3(1024), 16 * n_agent * sizeof(int), st,
                       agent_id, B, n_agent, n_blocks, scratch, index_out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

static int rows_common(bool scatter, const void *src, const int64_t *index, int64_t n, int64_t row_bytes,
                       void *dst, void *stream) {
    TSM_REQUIRE(n >= 0 && row_bytes > 0, "tsm_%s_rows: bad sizes", scatter ? "scatter" : "gather");
    if (n == 0) return TSM_OK;
    TSM_REQUIRE(src && index && dst, "tsm_%s_rows: null pointer", scatter ? "scatter" : "gather");
    const int word = (row_bytes % 4 == 0 && ((uintptr_t)src | (uintptr_t)dst) % 4 == 0) ? 4 : 1;
    const dim3 grid((unsigned)ceil_div(n * (row_bytes / word), 256));
    if (scatter)
        hipLaunchKernelGGL((rows_kernel<true>), grid, dim3(256), 0, tsm_stream(stream), (const char *)src, index, n,
                           row_bytes, (char *)dst, word);
    else
        hipLaunchKernelGGL((rows_kernel<false>), grid, dim3(256), 0, tsm_stream(stream), (const char *)src, index, n,
                           row_bytes, (char *)dst, word);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int tsm_scatter_rows(const void *src, const int64_t *index, int64_t n, int64_t row_bytes, void *dst,
                                void *stream) {
    return rows_common(true, src, index, n, row_bytes, dst, stream);
}
TSM_EXPORT int tsm_gather_rows(const void *src, const int64_t *index, int64_t n, int64_t row_bytes, void *dst,
                               void *stream) {
    return rows_common(false, src, index, n, row_bytes, dst, stream);
}
