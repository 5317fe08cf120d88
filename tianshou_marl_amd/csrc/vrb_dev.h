// vrb_dev.h -- device-side VectorReplayBuffer index algebra shared by vrb.hip and the persistent rollout
// kernel.  Restates ReplayBuffer._update_state_pre_add + ReplayBufferManager.add
// (/root/reference/tianshou/data/buffer/buffer_base.py:355-410, manager.py:159-177); bit-exact with the
// reference (tests/test_gpu_kernels.py::test_vrb_trace_bit_exact).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct VrbState {
    int64_t *ins, *size, *ep_len, *ep_start, *last_index, *lengths;
    double *ep_return;
    int64_t *error_flag;
};

// state = i64 insertion_idx[B] size[B] ep_len[B] ep_start_idx[B] last_index[B] lengths[B] | f64 ep_return[B][D] | i64 flag
__host__ __device__ inline VrbState vrb_view(void *state, int64_t B, int64_t D) {
    VrbState s;
    int64_t *p = reinterpret_cast<int64_t *>(state);
    s.ins = p;
    s.size = p + B;
    s.ep_len = p + 2 * B;
    s.ep_start = p + 3 * B;
    s.last_index = p + 4 * B;
    s.lengths = p + 5 * B;
    s.ep_return = reinterpret_cast<double *>(p + 6 * B);
    s.error_flag = p + 6 * B + B * D;
    return s;
}

// Index bookkeeping for ONE row added to sub-buffer `e`.  Returns the slot the payload goes to.
// Outputs: *ptr_out (flat reference index), ep_rew_out[D], *ep_len_out, *ep_idx_out  (manager.py:193).
__device__ inline int64_t vrb_add_row(const VrbState &s, int64_t B, int64_t S, int64_t D, int64_t e,
                                      const float *rew_row, bool d, uint8_t *done_store, int64_t *ptr_out,
                                      double *ep_rew_out, int64_t *ep_len_out, int64_t *ep_idx_out) {
    const int64_t cur = s.ins[e];                        // buffer_base.py:373
    int64_t sz = s.size[e] + 1; if (sz > S) sz = S;       // :374
    int64_t nxt = cur + 1; if (nxt >= S) nxt -= S;        // :375
    const int64_t elen = s.ep_len[e] + 1;                 // :378
    const int64_t estart = s.ep_start[e];
    if (estart > sz) atomicExch((unsigned long long *)s.error_flag, 1ull);  // :380-386
    for (int64_t k = 0; k < D; ++k) {
        const double acc = s.ep_return[e * D + k] + (double)rew_row[k];      // :377
        ep_rew_out[k] = d ? acc : 0.0;                    // :389-402
        s.ep_return[e * D + k] = d ? 0.0 : acc;           // :409
    }
    *ep_len_out = d ? elen : 0;
    const int64_t off = e * S;
    *ptr_out = cur + off;                                 // manager.py:170
    *ep_idx_out = estart + off;                           // manager.py:171
    s.ins[e] = nxt;
    s.size[e] = sz;
    s.ep_len[e] = d ? 0 : elen;
    s.ep_start[e] = d ? nxt : estart;                     // :409
    s.last_index[e] = cur + off;                          // manager.py:176
    s.lengths[e] = sz;                                    // manager.py:177
    done_store[cur * B + e] = d ? 1 : 0;
    return cur;
}
