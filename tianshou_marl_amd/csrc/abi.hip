// abi.hip -- version, error reporting, device info and the small memory/stream helpers of
// include/tsmarl.h.  No reference counterpart (the reference is pure Python; these exist so that a
// non-PyTorch host can drive the C-ABI).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void tsm_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

TSM_EXPORT int tsm_abi_version(void) { return TSM_ABI_VERSION; }
TSM_EXPORT const char *tsm_last_error(void) { return g_err; }

TSM_EXPORT int tsm_device_info(int *n_cu, int *wave_size, int64_t *hbm_bytes, char *name_out) {
    int dev = 0;
    TSM_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    TSM_HIP(hipGetDeviceProperties(&p, dev));
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (wave_size) *wave_size = p.warpSize;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    if (name_out) {
        strncpy(name_out, p.gcnArchName, 63);
        name_out[63] = 0;
    }
    return TSM_OK;
}

TSM_EXPORT int tsm_mem_alloc(void **dptr, int64_t bytes) {
    TSM_REQUIRE(dptr && bytes >= 0, "tsm_mem_alloc: bad args");
    TSM_HIP(hipMalloc(dptr, (size_t)(bytes > 0 ? bytes : 1)));
    return TSM_OK;
}
TSM_EXPORT int tsm_mem_free(void *dptr) {
    if (dptr) TSM_HIP(hipFree(dptr));
    return TSM_OK;
}
TSM_EXPORT int tsm_mem_h2d(void *dst, const void *src_host, int64_t bytes, void *stream) {
    TSM_REQUIRE(bytes >= 0 && (bytes == 0 || (dst && src_host)), "tsm_mem_h2d: bad args");
    if (bytes) TSM_HIP(hipMemcpyAsync(dst, src_host, (size_t)bytes, hipMemcpyHostToDevice, tsm_stream(stream)));
    return TSM_OK;
}
TSM_EXPORT int tsm_mem_d2h(void *dst_host, const void *src, int64_t bytes, void *stream) {
    TSM_REQUIRE(bytes >= 0 && (bytes == 0 || (dst_host && src)), "tsm_mem_d2h: bad args");
    if (bytes) TSM_HIP(hipMemcpyAsync(dst_host, src, (size_t)bytes, hipMemcpyDeviceToHost, tsm_stream(stream)));
    return TSM_OK;
}
TSM_EXPORT int tsm_mem_set(void *dst, int value, int64_t bytes, void *stream) {
    TSM_REQUIRE(bytes >= 0 && (bytes == 0 || dst), "tsm_mem_set: bad args");
    if (bytes) TSM_HIP(hipMemsetAsync(dst, value, (size_t)bytes, tsm_stream(stream)));
    return TSM_OK;
}
TSM_EXPORT int tsm_stream_sync(void *stream) {
    TSM_HIP(hipStreamSynchronize(tsm_stream(stream)));
    return TSM_OK;
}
// Take `stream` out of hipGraph capture mode if it is in it (a capture that failed half-way leaves it there, and every
// later synchronising call on the device then fails); the partial graph is dropped.  Returns 1 if a capture was ended.
TSM_EXPORT int tsm_stream_abort_capture(void *stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(tsm_stream(stream), &st) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (st == hipStreamCaptureStatusNone) return 0;
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture(tsm_stream(stream), &g);  // an invalidated capture reports an error but still ends
    (void)hipGetLastError();
    if (g) (void)hipGraphDestroy(g);
    return 1;
}

// ---- kernel selection options ----------------------------------------------------------------------------------------
// Where one entry point has two kernels behind it, the choice is a rule over the problem size.  An option overrides the rule
// for this process: its default comes from the environment (read once, at first use), tsm_kernel_option_set() replaces it at
// any time -- so one test process can run the same reference fixture through every kernel of an entry point.
//   "actor_tile"  0: by minibatch size (tsm_ppo_actor_rows_grid) | 32 | 64      env TSM_ACTOR_TILE
//   "split_bf16"  0: f32 matrix pipe | 1: layer 1 of the critic forward on the bf16 pipe with three-way split operands
//                 (experimental, never the default)                              env TSM_SPLIT_BF16
//   "generic_kernels" 0: the instantiations compiled for BASELINE's dimensions where they apply (obs 18 / 5 actions / 3 agents,
//                 obs 16 / 5 actions, obs 48 / 8 agents) | 1: the generic forms everywhere (the same bits: tests/ compare the two;
//                 also the A/B switch of the timing tools)                          env TSM_GENERIC_KERNELS
//   "rollout_form" the persistent rollouts (tsm_rollout_spread, tsm_rollout_spread_actor): 0 by rule (wave-autonomous form where
//                 all its waves are resident at once) | 1: tile form | 2: wave-autonomous form
//                                                                                   env TSM_ROLLOUT_FORM
namespace {
struct KernelOption { const char *name, *env; int value; bool resolved; };
KernelOption g_opts[TSM_OPT_COUNT] = {{"actor_tile", "TSM_ACTOR_TILE", 0, false}, {"split_bf16", "TSM_SPLIT_BF16", 0, false},
                                      {"generic_kernels", "TSM_GENERIC_KERNELS", 0, false}, {"rollout_form", "TSM_ROLLOUT_FORM", 0, false}};
bool opt_valid(int id, int v) {
    if (id == TSM_OPT_ROLLOUT_FORM) return v >= 0 && v <= 2;
    return id == TSM_OPT_ACTOR_TILE ? (v == 0 || v == 32 || v == 64) : (v == 0 || v == 1);
}
}  // namespace

int tsm_opt(int id) {
    KernelOption &o = g_opts[id];
    if (!o.resolved) {
        const char *e = getenv(o.env);
        const int v = e ? atoi(e) : 0;
        o.value = opt_valid(id, v) ? v : 0;
        o.resolved = true;
    }
    return o.value;
}

static int opt_id(const char *name) {
    for (int i = 0; name && i < TSM_OPT_COUNT; ++i)
        if (!strcmp(name, g_opts[i].name)) return i;
    return -1;
}

TSM_EXPORT int tsm_kernel_option_get(const char *name, int32_t *value_out) {
    const int id = opt_id(name);
    TSM_REQUIRE(id >= 0 && value_out, "tsm_kernel_option_get: unknown option '%s'", name ? name : "(null)");
    *value_out = tsm_opt(id);
    return TSM_OK;
}

TSM_EXPORT int tsm_kernel_option_set(const char *name, int32_t value) {
    const int id = opt_id(name);
    TSM_REQUIRE(id >= 0, "tsm_kernel_option_set: unknown option '%s'", name ? name : "(null)");
    TSM_REQUIRE(opt_valid(id, value), "tsm_kernel_option_set: %d is not a value of '%s'", value, name);
    g_opts[id].value = value;
    g_opts[id].resolved = true;
    return TSM_OK;
}

// diagnostic hook (not part of the public ABI header): device buffer of i64 phase time stamps written by
// workgroup 0 of the rollout / update kernels when set (tools/stamp_*.py)
long long *g_tsm_stamps = nullptr;
extern "C" __attribute__((visibility("default"))) void tsm_debug_set_stamps(long long *p) { g_tsm_stamps = p; }
// Everything process-wide that changes WHICH kernel an entry point launches or arms its phase stamps, for hosts that must prove
// a measurement ran the default configuration (bench.py prints it and refuses non-defaults): out[0] = update variant
// (tsm_debug_set_update_variant), out[1] = slab store flavour (tsm_debug_set_slab_store), out[2] = stamps armed, out[3] = 0.
extern "C" int tsm_debug_update_variant_get(void);
extern "C" int tsm_debug_slab_store_get(void);
extern "C" __attribute__((visibility("default"))) void tsm_debug_get_state(int32_t *out4) {
    out4[0] = tsm_debug_update_variant_get();
    out4[1] = tsm_debug_slab_store_get();
    out4[2] = g_tsm_stamps != nullptr;
    out4[3] = 0;
}
