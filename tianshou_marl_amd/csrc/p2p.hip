// p2p.hip -- latency-grade all-reduce of a SMALL vector over peer-mapped memory ("one-shot /
// direct all-reduce ... over the 7 xGMI links" for the 45 KB shared-policy gradient, which a ring collective prices at
// several link latencies per step, 18 times per update).
//
// The reference has no distributed path at all (only single-process nn.DataParallel wrappers,
// /root/reference/tianshou/utils/net/common.py:477-519); this replaces the one collective the env-sharded update has:
// `torch.distributed.all_reduce(flat_grad)` in front of Optimizer.step (algorithm_base.py:485-498) for replicas.
//
// One-shot, write-to-peers, LL-style ("low latency", as RCCL's LL protocol):
//   * every rank owns an INBOX in fine-grained device memory, exported with hipIpcGetMemHandle and mapped by every peer
//     (on a node all 8 GPUs are one xGMI hop apart): inbox[parity][sender][max_floats] of 8-BYTE words.
//   * every element travels as ONE naturally aligned 8-byte store {float bits, low 32 bits of the call's sequence number}: a
//     single transaction, so data and stamp arrive together -- no fence, no flag array, no barrier.  One launch per all-reduce;
//     the lane that owns element i stores it into every peer's inbox, polls ITS OWN inbox for the peers' element i until the
//     stamp matches (bounded spin), sums the senders in RANK ORDER and writes the result back.
//   * every rank adds the same numbers in the same order: the result is bit-identical on all ranks (replicas stay
//     bit-identical), and equal to (((x_0 + x_1) + x_2) + ...).
//   * parity = call count & 1.  A rank starts call k + 1 only after its call k has finished, i.e. after it received every
//     peer's call-k elements -- which the peers sent after finishing THEIR call k - 1: when call k + 1 overwrites the inbox half
//     of call k - 1, every receiver is done with it.  Two halves suffice; a stale word carries an older stamp and never matches.
//   * the sequence number lives in device memory (read by every workgroup at its start, advanced by the last one to finish), so
//     launch arguments are constant and a captured launch replays inside a hipGraph.
// An earlier version signalled per 1/32 slice with flags behind system-scope fences (an L2 write-back + invalidate each): ~10 us
// per call with nobody to wait for, and ~30 us once a fused step used one workgroup per 64 parameters (the fences serialise per
// XCD).  The LL form has neither.
// Failure behaviour (fail-stop, the last good parameters survive): a peer that never sends (crashed rank) ends the spin after
// ~2 s with the handle's error word set.  The lane that timed out does NOT use the partial sum: the fused step skips Adam for
// that element (parameter and moments untouched), the plain all-reduce leaves the element as it was; every LATER launch on the
// handle sees the error word at its start and does nothing at all (so a captured update does not spin 2 s per step).  The host
// reads the word at its next synchronisation point (tsm_p2p_failed) and raises -- no hang, no replica stepped on a partial sum.
// First use: tsm_p2p_handshake exchanges one stamped word per peer with the same bounded spin BEFORE anything is captured;
// the ranks then agree (over the process group) to use this path or the backend's own collective for the rest of the process.
#define TSM_ADAM_DEEP 1   // as in csrc/adam.hip: 64 slab loads in flight per lane (same order of additions)
#include "adam_dev.h"
#include <string.h>

namespace {

constexpr int kP2PMaxWorld = 16;
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
    float *hs_dev;                     // device: the handshake's one element
    uint64_t spin_limit;               // polls per missing word (tsm_p2p_set_timeout; default ~2 s)
};

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
    uint64_t spin_limit;               // polls before a missing word counts as a dead peer
};

// element i of sender `from`, call parity `par`, inside an inbox
__device__ __forceinline__ uint64_t *ll_slot(char *inbox, int par, int world, int from, int64_t max_floats, int64_t i) {
    return reinterpret_cast<uint64_t *>(inbox) + ((size_t)par * world + from) * max_floats + i;
}

// send `val` as element i of this call to every peer, receive the peers' element i; returns the rank-ordered sum in `sum`
// and false if a peer's word did not arrive within the spin limit (`sum` is then not to be used)
__device__ __forceinline__ bool ll_exchange(const P2PArgs &a, int par, uint64_t tag, int64_t i, float val, float &sum) {
    const uint64_t word = tag | (uint64_t)__float_as_uint(val);
    for (int r = 0; r < a.world; ++r)
        if (r != a.rank) __hip_atomic_store(ll_slot(a.peer[r], par, a.world, a.rank, a.max_floats, i), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    float acc = 0.f;
    bool ok = true;
    for (int r = 0; r < a.world && ok; ++r) {
        float vr = val;
        if (r != a.rank) {
            const uint64_t *src = ll_slot(a.peer[a.rank], par, a.world, r, a.max_floats, i);
            uint64_t wv, spins = 0;
            while (((wv = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) & 0xFFFFFFFF00000000ull) != tag) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > a.spin_limit) {  // a peer is gone: flag it, use nothing of this exchange (the host raises at its next sync)
                    atomicExch(a.err, 1);
                    ok = false;
                    break;
                }
            }
            vr = __uint_as_float((uint32_t)wv);
        }
        acc = r == 0 ? vr : acc + vr;
    }
    sum = acc;
    return ok;
}

// has an earlier launch on this handle lost a peer?  (then this one does nothing: see "Failure behaviour")
__device__ __forceinline__ bool ll_dead(const P2PArgs &a) { return __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; }

// the last workgroup to get here advances the stamp for the next call (every workgroup read it at its start)
__device__ __forceinline__ void ll_finish(const P2PArgs &a, uint64_t seq) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint64_t done = __hip_atomic_fetch_add(a.seq_dev + 1, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (uint64_t)gridDim.x - 1) {
            __hip_atomic_store(a.seq_dev + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.seq_dev, seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ __launch_bounds__(kP2PThreads) void p2p_all_reduce_kernel(P2PArgs a) {
    const uint64_t seq = __hip_atomic_load(a.seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int par = (int)(seq & 1);
    const uint64_t tag = (seq & 0xFFFFFFFFull) << 32;
    const int64_t i = (int64_t)blockIdx.x * kP2PThreads + threadIdx.x;
    if (i < a.n && !ll_dead(a)) {
        float sum;
        if (ll_exchange(a, par, tag, i, a.data[i], sum)) a.data[i] = sum;
    }
    ll_finish(a, seq);
}

// ---- slab reduction + all-reduce + Adam in ONE launch ----------------------------------------------------------------------
// The data-parallel gradient step was three launches (tsm_reduce_slabs -> all-reduce -> tsm_adam_step).  None of the three needs
// more than its own element of the parameter vector, and the LL exchange has no workgroup-wide step: the lane that summed
// parameter i's slabs (x 1 / world) sends it, waits for the peers' element i, sums in rank order and applies Adam to i.  Same
// instructions as the three-launch form (adam_dev.h): bit-identical.  (No gradient-norm clip: the global norm is a grid-wide
// dependency; such steps keep the three launches.)
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
};

__global__ __launch_bounds__(256) void p2p_adam_kernel(P2PAdamArgs g) {
    __shared__ float sm[256];
    const P2PArgs &a = g.c;
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const uint64_t seq = __hip_atomic_load(a.seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int par = (int)(seq & 1);
    const uint64_t tag = (seq & 0xFFFFFFFFull) << 32;
    const int64_t i = (int64_t)blockIdx.x * kCols + lane;
    const float gsum = slab_sum_block(g.slabs, g.n_slab, a.n, i, sm, a.n);
    if (sl == 0 && i < a.n && !ll_dead(a)) {
        float acc;
        if (ll_exchange(a, par, tag, i, gsum * g.scale, acc)) {  // (a timed-out element keeps its last good parameter and moments)
            const float pn = adam_apply(g.p, g.m, g.v, i, acc, g.lr_host, g.lr_dev, g.beta1, g.beta2, g.step_host, g.step_dev, g.eps,
                                        g.weight_decay);
            if (g.img) g.img[g.img_map[i]] = pn;
        }
    }
    ll_finish(a, seq);
}

}  // namespace

TSM_EXPORT int64_t tsm_p2p_ipc_handle_bytes(void) { return (int64_t)sizeof(hipIpcMemHandle_t); }

// Allocates this rank's inbox (fine-grained device memory on the current device) and returns an opaque handle.
TSM_EXPORT int tsm_p2p_create(int32_t rank, int32_t world, int64_t max_floats, void **handle_out) {
    TSM_REQUIRE(world >= 1 && world <= kP2PMaxWorld && rank >= 0 && rank < world && max_floats >= 1 && handle_out,
                "tsm_p2p_create: rank %d / world %d (<= %d), max_floats %lld", rank, world, kP2PMaxWorld, (long long)max_floats);
    P2PHandle *h = new P2PHandle();
    h->rank = rank; h->world = world; h->max_floats = max_floats;
    h->bytes = (size_t)2 * world * max_floats * sizeof(uint64_t);   // inbox[parity][sender][max_floats] of 8-byte words
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
    TSM_HIP(hipMalloc(reinterpret_cast<void **>(&h->hs_dev), sizeof(float)));
    h->spin_limit = kSpinLimit;
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
    a.err = h->err_dev; a.spin_limit = h->spin_limit;
    hipLaunchKernelGGL(p2p_all_reduce_kernel, dim3((unsigned)ceil_div(n, kP2PThreads)), dim3(kP2PThreads), 0, tsm_stream(stream), a);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}

// Polls per missing word ~ seconds x 2e7 (a poll is a system-scope load + s_sleep(1), ~50 ns).  Default ~2 s.
TSM_EXPORT int tsm_p2p_set_timeout(void *handle, double seconds) {
    TSM_REQUIRE(handle && seconds > 0.0, "tsm_p2p_set_timeout: null handle or non-positive time");
    static_cast<P2PHandle *>(handle)->spin_limit = (uint64_t)(seconds * 2.0e7) + 1;
    return TSM_OK;
}

// First-use handshake: one all-reduce of ONE element holding rank + 1 -- every peer's inbox is written once and every peer's
// word is awaited once, with the handle's bounded spin.  Synchronises `stream`.  *ok_out = 1 iff no spin timed out AND the sum
// is world (world + 1) / 2.  Call it on every rank at the same point, outside any stream capture; on 0 the handle is spent
// (its error word is set): destroy it and use the process group's collective instead.
TSM_EXPORT int tsm_p2p_handshake(void *handle, int32_t *ok_out, void *stream) {
    TSM_REQUIRE(handle && ok_out, "tsm_p2p_handshake: null pointer");
    P2PHandle *h = static_cast<P2PHandle *>(handle);
    const float mine = (float)(h->rank + 1);
    TSM_HIP(hipMemcpyAsync(h->hs_dev, &mine, sizeof(float), hipMemcpyHostToDevice, tsm_stream(stream)));
    TSM_HIP(hipStreamSynchronize(tsm_stream(stream)));   // (`mine` is a stack variable)
    if (const int rc = tsm_p2p_all_reduce(handle, h->hs_dev, 1, stream); rc != TSM_OK) return rc;
    float got = 0.f;
    TSM_HIP(hipMemcpyAsync(&got, h->hs_dev, sizeof(float), hipMemcpyDeviceToHost, tsm_stream(stream)));
    TSM_HIP(hipStreamSynchronize(tsm_stream(stream)));
    *ok_out = !tsm_p2p_failed(handle) && got == (float)(h->world * (h->world + 1) / 2);
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
    g.c.err = h->err_dev; g.c.spin_limit = h->spin_limit;
    g.p = param; g.m = exp_avg; g.v = exp_avg_sq; g.slabs = grad_slabs; g.n_slab = n_slab;
    g.lr_host = lr; g.lr_dev = lr_dev; g.beta1 = beta1; g.beta2 = beta2; g.step_host = step; g.step_dev = step_dev;
    g.eps = (float)eps; g.weight_decay = (float)weight_decay; g.scale = 1.0f / (float)h->world;
    g.img = param_image; g.img_map = image_map;
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

// The same word without a host synchronisation: an asynchronous copy into PINNED host memory behind everything queued on `stream`
// so far.  The host reads *host_pinned_out after it has waited for that point of the stream (the event it records for its loss
// statistics): every update checks the handle at a synchronisation it already pays for.
TSM_EXPORT int tsm_p2p_error_async(void *handle, int32_t *host_pinned_out, void *stream) {
    TSM_REQUIRE(handle && host_pinned_out, "tsm_p2p_error_async: null pointer");
    P2PHandle *h = static_cast<P2PHandle *>(handle);
    TSM_HIP(hipMemcpyAsync(host_pinned_out, h->err_dev, sizeof(int32_t), hipMemcpyDeviceToHost, tsm_stream(stream)));
    return TSM_OK;
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
    (void)hipFree(h->hs_dev);
    delete h;
    return TSM_OK;
}
