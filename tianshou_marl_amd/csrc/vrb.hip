// vrb.hip -- device-resident VectorReplayBuffer: index algebra + AoS->SoA payload scatter.
//
// Replaces ReplayBufferManager.add / ReplayBuffer._update_state_pre_add / sample_indices(0) /
// unfinished_index / _prev_index / _next_index / reset
// (/root/reference/tianshou/data/buffer/manager.py:70-229,306-358; buffer_base.py:292-410,495-531;
//  vecbuf.py:33-37).  Integer results are bit-exact with the reference (tests/test_gpu_vrb.py).
//
// State (one device allocation, caller-owned):
//   i64 insertion_idx[B] size[B] ep_len[B] ep_start_idx[B] last_index[B] lengths[B]
//   f64 ep_return[B][D]
//   i64 error_flag
// Payload storage is time-major SoA: field[(slot * B + env) * row_bytes ...]; a vector step that
// appends one row to every sub-buffer is therefore a single contiguous, fully coalesced copy.
#include "common.h"
#include "vrb_dev.h"

namespace {

__global__ void vrb_reset_kernel(void *state, int64_t B, int64_t S, int64_t D, int keep_stats,
                                 int init) {
    VrbState s = vrb_view(state, B, D);
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e == 0 && init) *s.error_flag = 0;
    if (e >= B) return;
    s.last_index[e] = e * S;  // manager.py:72
    s.lengths[e] = 0;         // manager.py:73
    s.ins[e] = 0;             // buffer_base.py:296
    s.size[e] = 0;
    s.ep_start[e] = 0;
    if (!keep_stats) {        // buffer_base.py:297-298
        s.ep_len[e] = 0;
        for (int64_t k = 0; k < D; ++k) s.ep_return[e * D + k] = 0.0;
    }
}

constexpr int kMaxFields = 12;
struct FieldTable {
    tsm_field f[kMaxFields];
    int n;
};

constexpr int kRowsPerBlock = 8;

// One launch: (1) per-row index algebra (buffer_base.py:373-410 + manager.py:170-177),
// (2) payload scatter of the same rows into the time-major SoA store.
__global__ __launch_bounds__(256) void vrb_add_kernel(
    void *state, int64_t B, int64_t S, int64_t D, const int64_t *__restrict__ ids, int64_t R,
    const float *__restrict__ rew, const uint8_t *__restrict__ done, uint8_t *__restrict__ done_store,
    FieldTable ft, int64_t *__restrict__ ptr_out, double *__restrict__ ep_rew_out,
    int64_t *__restrict__ ep_len_out, int64_t *__restrict__ ep_idx_out) {
    __shared__ int64_t s_dst_row[kRowsPerBlock];  // slot * B + env
    VrbState s = vrb_view(state, B, D);
    const int64_t r0 = (int64_t)blockIdx.x * kRowsPerBlock;
    const int nrow = (int)min((int64_t)kRowsPerBlock, R - r0);
    if (threadIdx.x < nrow) {
        const int64_t r = r0 + threadIdx.x;
        const int64_t e = ids ? ids[r] : r;
        const int64_t cur = vrb_add_row(s, B, S, D, e, rew + r * D, done[r] != 0, done_store, ptr_out + r,
                                        ep_rew_out + r * D, ep_len_out + r, ep_idx_out + r);
        s_dst_row[threadIdx.x] = cur * B + e;
    }
    __syncthreads();
    for (int fi = 0; fi < ft.n; ++fi) {
        const tsm_field f = ft.f[fi];
        const char *src = reinterpret_cast<const char *>(f.src) + r0 * f.row_bytes;
        char *dst = reinterpret_cast<char *>(f.dst);
        const bool w16 = (f.row_bytes % 16 == 0) && ((((uintptr_t)f.src) | ((uintptr_t)f.dst)) % 16 == 0);
        const bool w4 = (f.row_bytes % 4 == 0) && ((((uintptr_t)f.src) | ((uintptr_t)f.dst)) % 4 == 0);
        if (w16) {
            const int64_t wpr = f.row_bytes / 16;
            for (int64_t i = threadIdx.x; i < wpr * nrow; i += blockDim.x) {
                const int64_t rr = i / wpr, o = i - rr * wpr;
                reinterpret_cast<uint4 *>(dst + s_dst_row[rr] * f.row_bytes)[o] =
                    reinterpret_cast<const uint4 *>(src + rr * f.row_bytes)[o];
            }
        } else if (w4) {
            const int64_t wpr = f.row_bytes / 4;
            for (int64_t i = threadIdx.x; i < wpr * nrow; i += blockDim.x) {
                const int64_t rr = i / wpr, o = i - rr * wpr;
                reinterpret_cast<uint32_t *>(dst + s_dst_row[rr] * f.row_bytes)[o] =
                    reinterpret_cast<const uint32_t *>(src + rr * f.row_bytes)[o];
            }
        } else {
            for (int64_t i = threadIdx.x; i < f.row_bytes * nrow; i += blockDim.x) {
                const int64_t rr = i / f.row_bytes, o = i - rr * f.row_bytes;
                dst[s_dst_row[rr] * f.row_bytes + o] = src[rr * f.row_bytes + o];
            }
        }
    }
}

// exclusive scan of per-env sizes -> scratch[B+1]; n_out = total.  One 1024-thread block.
__global__ __launch_bounds__(1024) void vrb_size_scan_kernel(const void *state, int64_t B, int64_t D,
                                                             int64_t *scratch, int64_t *n_out) {
    __shared__ int64_t sm[1024 / 64 + 1];
    VrbState s = vrb_view(const_cast<void *>(state), B, D);
    int64_t base = 0;
    for (int64_t c = 0; c < B; c += 1024) {
        const int64_t e = c + threadIdx.x;
        const int64_t v = e < B ? s.size[e] : 0;
        int64_t total;
        const int64_t ex = block_exclusive_scan<int64_t, 1024>(v, sm, &total);
        if (e < B) scratch[e] = base + ex;
        base += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) { scratch[B] = base; *n_out = base; }
}

// buffer_base.py:511-514 per sub-buffer (+ offset, manager.py:224-229)
__global__ void vrb_sample_fill_kernel(const void *state, int64_t B, int64_t S, int64_t D,
                                       const int64_t *__restrict__ scratch, int64_t *__restrict__ out) {
    VrbState s = vrb_view(const_cast<void *>(state), B, D);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * S) return;
    const int64_t e = i / S, j = i - e * S;
    const int64_t sz = s.size[e], ins = s.ins[e];
    if (j >= sz) return;
    const int64_t head = sz - ins;  // len of arange(ins, size)
    const int64_t slot = j < head ? ins + j : j - head;
    out[scratch[e] + j] = slot + e * S;
}

// buffer_base.py:309-312 per sub-buffer, manager.py:85-91; compaction keeps env order
__global__ __launch_bounds__(1024) void vrb_unfinished_kernel(const void *state, int64_t B, int64_t S,
                                                              int64_t D, const uint8_t *__restrict__ done_store,
                                                              int64_t *__restrict__ out, int64_t *n_out) {
    __shared__ int64_t sm[1024 / 64 + 1];
    VrbState s = vrb_view(const_cast<void *>(state), B, D);
    int64_t base = 0;
    for (int64_t c = 0; c < B; c += 1024) {
        const int64_t e = c + threadIdx.x;
        int64_t flag = 0, last = 0;
        if (e < B) {
            const int64_t sz = s.size[e];
            if (sz > 0) {
                last = (s.ins[e] - 1) % sz;
                if (last < 0) last += sz;
                flag = done_store[last * B + e] ? 0 : 1;
            }
        }
        int64_t total;
        const int64_t ex = block_exclusive_scan<int64_t, 1024>(flag, sm, &total);
        if (flag) out[base + ex] = last + e * S;
        base += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_out = base;
}

__device__ __forceinline__ int64_t pymod(int64_t a, int64_t m) {
    int64_t r = a % m;
    return r < 0 ? r + m : r;
}

// manager.py:306-331 (prev) / :334-358 (next)
template <bool NEXT>
__global__ void vrb_prevnext_kernel(const void *state, int64_t B, int64_t S, int64_t D,
                                    const uint8_t *__restrict__ done_store,
                                    const int64_t *__restrict__ index, int64_t n,
                                    int64_t *__restrict__ out) {
    VrbState s = vrb_view(const_cast<void *>(state), B, D);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t idx = pymod(index[i], B * S);
    const int64_t e = idx / S, start = e * S;
    int64_t cur_len = s.lengths[e];
    if (cur_len < 1) cur_len = 1;
    if (NEXT) {
        const int64_t sub = idx - start;
        const int64_t end_flag = (done_store[sub * B + e] ? 1 : 0) | (idx == s.last_index[e] ? 1 : 0);
        out[i] = pymod(sub + 1 - end_flag, cur_len) + start;
    } else {
        const int64_t sub = pymod(idx - start - 1, cur_len);
        const int64_t end_flag = (done_store[sub * B + e] ? 1 : 0) | (sub + start == s.last_index[e] ? 1 : 0);
        out[i] = pymod(sub + end_flag, cur_len) + start;
    }
}

__global__ void vrb_gather_kernel(const char *__restrict__ store, int64_t B, int64_t S, int64_t row_bytes,
                                  const int64_t *__restrict__ index, int64_t n, char *__restrict__ out,
                                  int word) {
    const int64_t wpr = row_bytes / word;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * wpr) return;
    const int64_t r = i / wpr, o = i - r * wpr;
    const int64_t idx = pymod(index[r], B * S);
    const int64_t e = idx / S, slot = idx - e * S;
    const char *src = store + (slot * B + e) * row_bytes;
    if (word == 4) reinterpret_cast<uint32_t *>(out + r * row_bytes)[o] = reinterpret_cast<const uint32_t *>(src)[o];
    else out[r * row_bytes + o] = src[o];
}

}  // namespace

TSM_EXPORT int64_t tsm_vrb_state_bytes(int64_t buffer_num, int64_t rew_dim) {
    if (buffer_num < 0 || rew_dim < 1) return -1;
    return (6 * buffer_num + buffer_num * rew_dim + 1) * 8;
}

static int vrb_check_dims(int64_t B, int64_t S, int64_t D) {
    TSM_REQUIRE(B >= 1 && S >= 1 && D >= 1, "vrb: bad dims buffer_num=%lld sub_size=%lld rew_dim=%lld",
                (long long)B, (long long)S, (long long)D);
    return TSM_OK;
}

TSM_EXPORT int tsm_vrb_init(void *state, int64_t B, int64_t S, int64_t D, void *stream) {
    if (int rc = vrb_check_dims(B, S, D)) return rc;
    TSM_REQUIRE(state, "tsm_vrb_init: null state");
    hipLaunchKernelGGL(vrb_reset_kernel, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0, tsm_stream(stream),
                       state, B, S, D, 0, 1);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int tsm_vrb_reset(void *state, int64_t B, int64_t S, int64_t D, int keep_statistics,
                             void *stream) {
    if (int rc = vrb_check_dims(B, S, D)) return rc;
    TSM_REQUIRE(state, "tsm_vrb_reset: null state");
    hipLaunchKernelGGL(vrb_reset_kernel, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0, tsm_stream(stream),
                       state, B, S, D, keep_statistics, 0);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int tsm_vrb_add(void *state, int64_t B, int64_t S, int64_t D, const int64_t *buffer_ids,
                           int64_t R, const float *rew, const uint8_t *done, uint8_t *done_store,
                           const tsm_field *fields_host, int n_fields, int64_t *ptr_out,
                           double *ep_rew_out, int64_t *ep_len_out, int64_t *ep_idx_out, void *stream) {
    if (int rc = vrb_check_dims(B, S, D)) return rc;
    TSM_REQUIRE(R >= 0, "tsm_vrb_add: negative R");
    if (R == 0) return TSM_OK;
    TSM_REQUIRE(buffer_ids || R <= B, "tsm_vrb_add: R=%lld rows but only %lld sub-buffers", (long long)R,
                (long long)B);
    TSM_REQUIRE(state && rew && done && done_store && ptr_out && ep_rew_out && ep_len_out && ep_idx_out,
                "tsm_vrb_add: null pointer");
    TSM_REQUIRE(n_fields >= 0 && n_fields <= kMaxFields, "tsm_vrb_add: n_fields=%d exceeds %d", n_fields,
                kMaxFields);
    FieldTable ft;
    ft.n = n_fields;
    for (int i = 0; i < n_fields; ++i) {
        TSM_REQUIRE(fields_host[i].src && fields_host[i].dst && fields_host[i].row_bytes > 0,
                    "tsm_vrb_add: field %d invalid", i);
        ft.f[i] = fields_host[i];
    }
    hipLaunchKernelGGL(vrb_add_kernel, dim3((unsigned)ceil_div(R, kRowsPerBlock)), dim3(256), 0,
                       tsm_stream(stream), state, B, S, D, buffer_ids, R, rew, done, done_store, ft, ptr_out,
                       ep_rew_out, ep_len_out, ep_idx_out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int tsm_vrb_check(void *state, int64_t B, int64_t D, void *stream) {
    TSM_REQUIRE(state && B >= 1 && D >= 1, "tsm_vrb_check: bad args");
    int64_t flag = 0;
    VrbState s = vrb_view(state, B, D);
    TSM_HIP(hipMemcpyAsync(&flag, s.error_flag, sizeof(flag), hipMemcpyDeviceToHost, tsm_stream(stream)));
    TSM_HIP(hipStreamSynchronize(tsm_stream(stream)));
    if (flag) {
        tsm_set_error("MalformedBufferError: episode start index outside the available samples "
                      "(buffer_base.py:380-386)");
        return TSM_ERR_MALFORMED_BUFFER;
    }
    return TSM_OK;
}

TSM_EXPORT int tsm_vrb_sample_indices_all(const void *state, int64_t B, int64_t S, int64_t *out,
                                          int64_t *n_out, int64_t *scratch, void *stream) {
    TSM_REQUIRE(state && out && n_out && scratch && B >= 1 && S >= 1, "tsm_vrb_sample_indices_all: bad args");
    // rew_dim does not matter for the fields read here (they precede ep_return)
    hipLaunchKernelGGL(vrb_size_scan_kernel, dim3(1), dim3(1024), 0, tsm_stream(stream), state, B, (int64_t)1,
                       scratch, n_out);
    TSM_LAUNCH_CHECK();
    hipLaunchKernelGGL(vrb_sample_fill_kernel, dim3((unsigned)ceil_div(B * S, 256)), dim3(256), 0,
                       tsm_stream(stream), state, B, S, (int64_t)1, scratch, out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int tsm_vrb_unfinished_index(const void *state, int64_t B, int64_t S, const uint8_t *done_store,
                                        int64_t *out, int64_t *n_out, void *stream) {
    TSM_REQUIRE(state && done_store && out && n_out && B >= 1 && S >= 1, "tsm_vrb_unfinished_index: bad args");
    hipLaunchKernelGGL(vrb_unfinished_kernel, dim3(1), dim3(1024), 0, tsm_stream(stream), state, B, S,
                       (int64_t)1, done_store, out, n_out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int tsm_vrb_prev(const void *state, int64_t B, int64_t S, const uint8_t *done_store,
                            const int64_t *index, int64_t n, int64_t *out, void *stream) {
    TSM_REQUIRE(n >= 0, "tsm_vrb_prev: negative n");
    if (n == 0) return TSM_OK;
    TSM_REQUIRE(state && done_store && index && out && B >= 1 && S >= 1, "tsm_vrb_prev: bad args");
    hipLaunchKernelGGL((vrb_prevnext_kernel<false>), dim3((unsigned)ceil_div(n, 256)), dim3(256), 0,
                       tsm_stream(stream), state, B, S, (int64_t)1, done_store, index, n, out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int tsm_vrb_next(const void *state, int64_t B, int64_t S, const uint8_t *done_store,
                            const int64_t *index, int64_t n, int64_t *out, void *stream) {
    TSM_REQUIRE(n >= 0, "tsm_vrb_next: negative n");
    if (n == 0) return TSM_OK;
    TSM_REQUIRE(state && done_store && index && out && B >= 1 && S >= 1, "tsm_vrb_next: bad args");
    hipLaunchKernelGGL((vrb_prevnext_kernel<true>), dim3((unsigned)ceil_div(n, 256)), dim3(256), 0,
                       tsm_stream(stream), state, B, S, (int64_t)1, done_store, index, n, out);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

TSM_EXPORT int tsm_vrb_gather(const void *store, int64_t B, int64_t S, int64_t row_bytes,
                              const int64_t *index, int64_t n, void *out, void *stream) {
    TSM_REQUIRE(n >= 0 && row_bytes > 0, "tsm_vrb_gather: bad sizes");
    if (n == 0) return TSM_OK;
    TSM_REQUIRE(store && index && out && B >= 1 && S >= 1, "tsm_vrb_gather: bad args");
    const int word = (row_bytes % 4 == 0 && ((uintptr_t)store | (uintptr_t)out) % 4 == 0) ? 4 : 1;
    hipLaunchKernelGGL(vrb_gather_kernel, dim3((unsigned)ceil_div(n * (row_bytes / word), 256)), dim3(256), 0,
                       tsm_stream(stream), reinterpret_cast<const char *>(store), B, S, row_bytes, index, n,
                       reinterpret_cast<char *>(out), word);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}
