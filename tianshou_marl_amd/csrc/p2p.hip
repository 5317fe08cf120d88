// p2p.hip -- latency-grade all-reduce of a SMALL vector over peer-mapped memory ("one-shot /
// direct all-reduce ... over the 7 xGMI links" for the 45 KB shared-policy gradient, which a ring collective prices at
// several link latencies per step, 18 times per update).
//
// The reference has no distributed path at all (only single-process nn.DataParallel wrappers,
// /root/reference/tianshou/utils/net/common.py:477-519); this replaces the one collective the env-sharded update has:
// `torch.distributed.all_reduce(flat_grad)` in front of Optimizer.step (algorithm_base.py:485-498) for replicas.
//
// One-shot, write-to-peers:
//   * every rank owns an INBOX in fine-grained device memory, exported with hipIpcGetMemHandle and mapped by every peer
//     (on a node all 8 GPUs are one xGMI hop apart): inbox[parity][sender][max_floats] + flags[parity][sender][slice].
//   * one launch per all-reduce, S workgroups.  Workgroup s of rank r (1) stores slice s of its vector into
//     inbox[p][r][slice s] of EVERY rank (its own included), (2) drains its stores and release-fences at system scope,
//     (3) stamps flags[p][r][s] = sequence number on every rank, (4) polls its OWN flags[p][*][s] until every sender's stamp
//     has arrived (bounded spin), (5) sums slice s over the senders in RANK ORDER out of its own inbox and writes the result
//     back into the vector.  No grid-wide barrier: a slice only ever waits for the same slice of the other ranks.
//   * every rank adds the same numbers in the same order: the result is bit-identical on all ranks (replicas stay
//     bit-identical), and equal to (((x_0 + x_1) + x_2) + ...).
//   * parity = call count & 1.  A sender can be at most one call ahead of a receiver that it has not heard from (its call
//     k + 1 needs the receiver's stamps of call k + 1, which the receiver issues only after it finished call k), so two
//     inbox halves suffice; stamps are 64-bit and never reset.
// Failure behaviour: a peer that never stamps (crashed rank) ends the spin after ~2 s with the handle's error word set;
// the host reads it at its next synchronisation point and raises -- no hang.
#include "adam_dev.h"
#include <string.h>

namespace {

constexpr int kP2PMaxWorld = 16;
constexpr int kP2PSlices = 32;        // workgroups per all-reduce (= slices of the vector)
constexpr int kP2PMaxSlices = 256;    // flag slots per (parity, sender): the fused step below slices finer (one per 64+ parameters)
constexpr int kP2PThreads = 256;
constexpr uint64_t kSpinLimit = 40000000ull;   // polls (~50 ns each): a couple of seconds

struct P2PHandle {
    int rank, world;
    int64_t max_floats;
    size_t bytes;
    void *local;                       // this rank's inbox allocation (fine-grained)
    void *peer[kP2PMaxWorld];          // every rank's inbox as mapped into this process (peer[rank] == local)
    bool opened[kP2PMaxWorld];
    uint64_t *seq_dev;                 // device: {stamp of the NEXT call (starts at 1), workgroups finished in the running call}
    int *err_dev;                      // device int: set by a timed-out spin
};

__host__ __device__ inline size_t p2p_data_floats(int world, int64_t max_floats) { return (size_t)2 * world * max_floats; }
__host__ __device__ inline size_t p2p_flag_offset_bytes(int world, int64_t max_floats) {
    return (p2p_data_floats(world, max_floats) * sizeof(float) + 255) / 256 * 256;
}

struct P2PArgs {
    float *data;
    int64_t n;
    int rank, world;
    int64_t max_floats;
    uint64_t *seq_dev;                 // [0] stamp of this call, read by every workgroup at its start and advanced by the
                                       // last one to finish ([1] counts them): launch arguments stay constant, so a captured
                                       // launch can be replayed
    char *peer[kP2PMaxWorld];
    int *err;
};

__global__ __launch_bounds__(kP2PThreads) void p2p_all_reduce_kernel(P2PArgs a) {
    const int s = blockIdx.x, tid = threadIdx.x;
    const uint64_t seq = __hip_atomic_load(a.seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int par = (int)(seq & 1);
    const int64_t per = (a.n + kP2PSlices - 1) / kP2PSlices;
    const int64_t lo = (int64_t)s * per, hi = lo + per < a.n ? lo + per : a.n;
    const size_t flag_off = p2p_flag_offset_bytes(a.world, a.max_floats);
    // (1) my slice into everybody's inbox
    for (int r = 0; r < a.world; ++r) {
        float *dst = reinterpret_cast<float *>(a.peer[r]) + ((size_t)par * a.world + a.rank) * a.max_floats;
        for (int64_t i = lo + tid; i < hi; i += kP2PThreads) dst[i] = a.data[i];
    }
    // (2) all stores of this workgroup are out and visible system-wide before the stamps
    __threadfence_system();
    __syncthreads();
    // (3) stamp: flags[par][rank][s] on every rank
    if (tid < a.world) {
        uint64_t *fl = reinterpret_cast<uint64_t *>(a.peer[tid] + flag_off) + ((size_t)par * a.world + a.rank) * kP2PMaxSlices + s;
        __hip_atomic_store(fl, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // (4) wait for every sender's stamp of this slice (my own flags, written by the peers)
    if (tid < a.world) {
        const uint64_t *fl = reinterpret_cast<const uint64_t *>(a.peer[a.rank] + flag_off) + ((size_t)par * a.world + tid) * kP2PMaxSlices + s;
        uint64_t spins = 0;
        while (__hip_atomic_load(fl, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > kSpinLimit) {  // a peer is gone: flag the error, leave (the host raises at its next sync)
                atomicExch(a.err, 1);
                break;
            }
        }
    }
    __syncthreads();
    __threadfence_system();
    // (5) sum over the senders in rank order
    const float *in = reinterpret_cast<const float *>(a.peer[a.rank]) + (size_t)par * a.world * a.max_floats;
    for (int64_t i = lo + tid; i < hi; i += kP2PThreads) {
        float acc = in[i];
        for (int r = 1; r < a.world; ++r) acc += in[(size_t)r * a.max_floats + i];
        a.data[i] = acc;
    }
    // the last workgroup to get here advances the stamp for the next call (every workgroup read it at its start)
    __syncthreads();
    if (tid == 0) {
        const uint64_t done = __hip_atomic_fetch_add(a.seq_dev + 1, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (uint64_t)gridDim.x - 1) {
            __hip_atomic_store(a.seq_dev + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.seq_dev, seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- slab reduction + all-reduce + Adam in ONE launch ----------------------------------------------------------------------
// The data-parallel gradient step was three launches (tsm_reduce_slabs -> all-reduce -> tsm_adam_step).  None of the three needs
// more than its own SLICE of the parameter vector, so one kernel can do all of it per parameter -- if the exchange itself has no
// workgroup-wide step.  The flag protocol above has two system-scope fences per workgroup (an L2 write-back + invalidate each):
// with one workgroup per 64 parameters they serialise per XCD, and a first fused version built on it took ~30 us per call.  This
// kernel therefore exchanges LL-style ("low latency", as RCCL's LL protocol): every element travels as ONE naturally aligned
// 8-byte store {float bits, 32-bit stamp of the call} into a second inbox region [parity][sender][max_floats] of 8-byte words;
// the receiving lane polls ITS OWN elements until the stamp matches.  An 8-byte store is a single transaction, so data and stamp
// arrive together: no fence, no flag array, no barrier -- the thread that summed parameter i's slabs sends it, waits for the
// peers' element i, sums in rank order and applies Adam to i.  Same instructions as the three-launch form (adam_dev.h):
// bit-identical.  (No gradient-norm clip: the global norm is a grid-wide dependency; such steps keep the three launches.)
struct P2PAdamArgs {
    P2PArgs c;                    // data unused
    float *p, *m, *v;
    const float *slabs;
    int32_t n_slab;
    double lr_host, beta1, beta2;
    const double *lr_dev;
    int64_t step_host;
    const int64_t *step_dev;
    float eps, weight_decay, scale;
    float *img;
    const int32_t *img_map;
    size_t ll_off;                // byte offset of the LL region inside an inbox allocation
};

__global__ __launch_bounds__(256) void p2p_adam_kernel(P2PAdamArgs g) {
    __shared__ float sm[256];
    const P2PArgs &a = g.c;
    const int tid = threadIdx.x, lane = tid & 63, sl = tid >> 6;
    const uint64_t seq = __hip_atomic_load(a.seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int par = (int)(seq & 1);
    const uint64_t tag = (seq & 0xFFFFFFFFull) << 32;
    const int64_t i = (int64_t)blockIdx.x * kCols + lane;
    const float gsum = slab_sum_block(g.slabs, g.n_slab, a.n, i, sm, a.n);
    if (sl == 0 && i < a.n) {
        const float val = gsum * g.scale;
        const uint64_t word = tag | (uint64_t)__float_as_uint(val);
        for (int r = 0; r < a.world; ++r)
            if (r != a.rank) {
                uint64_t *dst = reinterpret_cast<uint64_t *>(a.peer[r] + g.ll_off) + ((size_t)par * a.world + a.rank) * a.max_floats + i;
                __hip_atomic_store(dst, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        const uint64_t *in = reinterpret_cast<const uint64_t *>(a.peer[a.rank] + g.ll_off) + (size_t)par * a.world * a.max_floats + i;
        float acc = 0.f;
        for (int r = 0; r < a.world; ++r) {
            float vr = val;
            if (r != a.rank) {
                uint64_t wv, spins = 0;
                while (((wv = __hip_atomic_load(in + (size_t)r * a.max_floats, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) &
                        0xFFFFFFFF00000000ull) != tag) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > kSpinLimit) {  // a peer is gone: flag the error, go on with what there is
                        atomicExch(a.err, 1);
                        break;
                    }
                }
                vr = __uint_as_float((uint32_t)wv);
            }
            acc = r == 0 ? vr : acc + vr;
        }
        const float pn = adam_apply(g.p, g.m, g.v, i, acc, g.lr_host, g.lr_dev, g.beta1, g.beta2, g.step_host, g.step_dev, g.eps,
                                    g.weight_decay);
        if (g.img) g.img[g.img_map[i]] = pn;
    }
    __syncthreads();
    if (tid == 0) {   // the last workgroup to get here advances the stamp for the next call (every workgroup read it at its start)
        const uint64_t done = __hip_atomic_fetch_add(a.seq_dev + 1, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (uint64_t)gridDim.x - 1) {
            __hip_atomic_store(a.seq_dev + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.seq_dev, seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__host__ __device__ inline size_t p2p_ll_offset_bytes(int world, int64_t max_floats) {
    return (p2p_flag_offset_bytes(world, max_floats) + (size_t)2 * world * kP2PMaxSlices * sizeof(uint64_t) + 255) / 256 * 256;
}

}  // namespace

TSM_EXPORT int64_t tsm_p2p_ipc_handle_bytes(void) { return (int64_t)sizeof(hipIpcMemHandle_t); }

// Allocates this rank's inbox (fine-grained device memory on the current device) and returns an opaque handle.
TSM_EXPORT int tsm_p2p_create(int32_t rank, int32_t world, int64_t max_floats, void **handle_out) {
    TSM_REQUIRE(world >= 1 && world <= kP2PMaxWorld && rank >= 0 && rank < world && max_floats >= 1 && handle_out,
                "tsm_p2p_create: rank %d / world %d (<= %d), max_floats %lld", rank, world, kP2PMaxWorld, (long long)max_floats);
    P2PHandle *h = new P2PHandle();
    h->rank = rank; h->world = world; h->max_floats = max_floats;
    h->bytes = p2p_ll_offset_bytes(world, max_floats) + (size_t)2 * world * max_floats * sizeof(uint64_t);   // + the LL region
    for (int i = 0; i < kP2PMaxWorld; ++i) { h->peer[i] = nullptr; h->opened[i] = false; }
    hipError_t e = hipExtMallocWithFlags(&h->local, h->bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        tsm_set_error("tsm_p2p_create: hipExtMallocWithFlags(fine-grained, %zu bytes) failed: %s", h->bytes, hipGetErrorString(e));
        delete h;
        return TSM_ERR_HIP;
    }
    TSM_HIP(hipMemset(h->local, 0, h->bytes));
    TSM_HIP(hipMalloc(reinterpret_cast<void **>(&h->err_dev), sizeof(int)));
    TSM_HIP(hipMemset(h->err_dev, 0, sizeof(int)));
    TSM_HIP(hipMalloc(reinterpret_cast<void **>(&h->seq_dev), 2 * sizeof(uint64_t)));
    const uint64_t init[2] = {1ull, 0ull};
    TSM_HIP(hipMemcpy(h->seq_dev, init, sizeof(init), hipMemcpyHostToDevice));
    TSM_HIP(hipDeviceSynchronize());
    h->peer[rank] = h->local;
    *handle_out = h;
    return TSM_OK;
}

TSM_EXPORT int tsm_p2p_export(void *handle, void *ipc_handle_out) {
    TSM_REQUIRE(handle && ipc_handle_out, "tsm_p2p_export: null pointer");
    P2PHandle *h = static_cast<P2PHandle *>(handle);
    TSM_HIP(hipIpcGetMemHandle(static_cast<hipIpcMemHandle_t *>(ipc_handle_out), h->local));
    return TSM_OK;
}

TSM_EXPORT int tsm_p2p_import(void *handle, int32_t peer, const void *ipc_handle) {
    TSM_REQUIRE(handle && ipc_handle, "tsm_p2p_import: null pointer");
    P2PHandle *h = static_cast<P2PHandle *>(handle);
    TSM_REQUIRE(peer >= 0 && peer < h->world && peer != h->rank, "tsm_p2p_import: peer %d out of range", peer);
    TSM_REQUIRE(!h->opened[peer], "tsm_p2p_import: peer %d imported twice", peer);
    hipIpcMemHandle_t m;
    memcpy(&m, ipc_handle, sizeof(m));
    TSM_HIP(hipIpcOpenMemHandle(&h->peer[peer], m, hipIpcMemLazyEnablePeerAccess));
    h->opened[peer] = true;
    return TSM_OK;
}

// In-place sum of data[0 .. n) over the ranks (n <= max_floats, the same n on every rank); asynchronous on `stream` and
// capturable into a hipGraph (the call's stamp lives in device memory, so a captured launch replays correctly).
TSM_EXPORT int tsm_p2p_all_reduce(void *handle, float *data, int64_t n, void *stream) {
    TSM_REQUIRE(handle && data, "tsm_p2p_all_reduce: null pointer");
    P2PHandle *h = static_cast<P2PHandle *>(handle);
    TSM_REQUIRE(n >= 1 && n <= h->max_floats, "tsm_p2p_all_reduce: n = %lld exceeds the inbox (%lld floats)", (long long)n,
                (long long)h->max_floats);
    for (int r = 0; r < h->world; ++r) TSM_REQUIRE(h->peer[r], "tsm_p2p_all_reduce: peer %d was never imported", r);
    P2PArgs a{};
    a.data = data; a.n = n; a.rank = h->rank; a.world = h->world; a.max_floats = h->max_floats; a.seq_dev = h->seq_dev;
    for (int r = 0; r < h->world; ++r) a.peer[r] = static_cast<char *>(h->peer[r]);
    a.err = h->err_dev;
    hipLaunchKernelGGL(p2p_all_reduce_kernel, dim3(kP2PSlices), dim3(kP2PThreads), 0, tsm_stream(stream), a);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

// tsm_reduce_slabs (x 1 / world) + all-reduce + tsm_adam_step (no gradient-norm clip) in ONE launch: every lane does all three
// for its own parameter (p2p_adam_kernel, LL exchange).  Arguments as tsm_adam_step; n <= max_floats.
TSM_EXPORT int tsm_p2p_adam_step(void *handle, float *param, const float *grad_slabs, int32_t n_slab, int64_t n, float *exp_avg,
                                 float *exp_avg_sq, int64_t step, const int64_t *step_dev, double lr, const double *lr_dev,
                                 double beta1, double beta2, double eps, double weight_decay, float *param_image,
                                 const int32_t *image_map, void *stream) {
    TSM_REQUIRE(handle && param && grad_slabs && exp_avg && exp_avg_sq, "tsm_p2p_adam_step: null pointer");
    P2PHandle *h = static_cast<P2PHandle *>(handle);
    TSM_REQUIRE(n >= 1 && n <= h->max_floats && n_slab >= 1 && (step >= 1 || step_dev),
                "tsm_p2p_adam_step: bad sizes n=%lld (inbox %lld floats) n_slab=%d step=%lld", (long long)n, (long long)h->max_floats,
                n_slab, (long long)step);
    TSM_REQUIRE(!param_image || image_map, "tsm_p2p_adam_step: param_image needs image_map");
    for (int r = 0; r < h->world; ++r) TSM_REQUIRE(h->peer[r], "tsm_p2p_adam_step: peer %d was never imported", r);
    P2PAdamArgs g{};
    g.c.data = nullptr; g.c.n = n; g.c.rank = h->rank; g.c.world = h->world; g.c.max_floats = h->max_floats; g.c.seq_dev = h->seq_dev;
    for (int r = 0; r < h->world; ++r) g.c.peer[r] = static_cast<char *>(h->peer[r]);
    g.c.err = h->err_dev;
    g.p = param; g.m = exp_avg; g.v = exp_avg_sq; g.slabs = grad_slabs; g.n_slab = n_slab;
    g.lr_host = lr; g.lr_dev = lr_dev; g.beta1 = beta1; g.beta2 = beta2; g.step_host = step; g.step_dev = step_dev;
    g.eps = (float)eps; g.weight_decay = (float)weight_decay; g.scale = 1.0f / (float)h->world;
    g.img = param_image; g.img_map = image_map;
    g.ll_off = p2p_ll_offset_bytes(h->world, h->max_floats);
    hipLaunchKernelGGL(p2p_adam_kernel, dim3((unsigned)ceil_div(n, kCols)), dim3(256), 0, tsm_stream(stream), g);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

// 1 if a spin timed out since the handle was created (synchronises the device).
TSM_EXPORT int tsm_p2p_failed(void *handle) {
    if (!handle) return 1;
    P2PHandle *h = static_cast<P2PHandle *>(handle);
    int v = 0;
    if (hipMemcpy(&v, h->err_dev, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    return v != 0;
}

TSM_EXPORT int tsm_p2p_destroy(void *handle) {
    if (!handle) return TSM_OK;
    P2PHandle *h = static_cast<P2PHandle *>(handle);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < h->world; ++r)
        if (r != h->rank && h->opened[r]) (void)hipIpcCloseMemHandle(h->peer[r]);
    (void)hipFree(h->local);
    (void)hipFree(h->err_dev);
    (void)hipFree(h->seq_dev);
    delete h;
    return TSM_OK;
}
