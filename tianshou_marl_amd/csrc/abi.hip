// abi.hip -- version, error reporting, device info and the small memory/stream helpers of
// include/tsmarl.h.  No reference counterpart (the reference is pure Python; these exist so that a
// non-PyTorch host can drive the C-ABI).
#include <stdarg.h>
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

// diagnostic hook (not part of the public ABI header): device buffer of i64 phase time stamps written by
// workgroup 0 of the rollout / update kernels when set (tools/stamp_*.py)
long long *g_tsm_stamps = nullptr;
extern "C" __attribute__((visibility("default"))) void tsm_debug_set_stamps(long long *p) { g_tsm_stamps = p; }
