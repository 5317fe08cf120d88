// gather_fields.hip -- several row-gathers with element conversion in ONE launch.
//
// What it replaces.  The MARL trainers hand every policy a per-agent Batch (training_coordinator.py:118,154,336); reading it out
// of the time-major device store as env-major rows (the reference's flat index order, manager.py:131-193) took one torch copy per
// field plus a dtype conversion for the actions and the two flag fields -- 9 launches per agent --, and `learn()` then moved the six
// fields into its static graph buffers with 6 more (int64 actions -> int32, bool flags -> u8): 30 launches of ~5 us per tag step,
// a fifth of it (profiles/r04_tag_step_timeline.txt).  Here a field is described once (source, row mapping, width, the two element
// types) and up to TSM_MAX_GATHER_FIELDS of them travel in one launch.  No reference counterpart (the reference indexes numpy arrays).
#include "common.h"

namespace {

struct GatherArgs {
    tsm_gather_field f[TSM_MAX_GATHER_FIELDS];
};

template <typename T>
__device__ __forceinline__ double load_as(const void *p, int64_t i) { return (double)reinterpret_cast<const T *>(p)[i]; }

// (values are small integers, 0 / 1 flags or f32: exact through an f32 -> f32 move, exact as integers below 2^53 otherwise)
__global__ __launch_bounds__(256) void gather_fields_kernel(GatherArgs a) {
    const tsm_gather_field f = a.f[blockIdx.y];
    const int64_t total = f.n_rows * f.width;
    const unsigned width = (unsigned)f.width, T32 = (unsigned)f.T;   // (the host checks total < 2^31: 32-bit divisions, not ~100-instruction 64-bit ones)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const unsigned iu = (unsigned)i, ru = iu / width, c = iu - ru * width;
        int64_t sr = ru;
        if (T32 > 0) { const unsigned e = ru / T32, t = ru - e * T32; sr = (int64_t)t * f.E + e; }   // env-major row <- time-major slot
        const int64_t s = sr * f.src_row_stride + f.src_offset + c;
        if (f.src_kind == TSM_KIND_F32 && f.dst_kind == TSM_KIND_F32) {
            reinterpret_cast<float *>(f.dst)[i] = reinterpret_cast<const float *>(f.src)[s];
            continue;
        }
        int64_t v;
        switch (f.src_kind) {
            case TSM_KIND_I32: v = reinterpret_cast<const int32_t *>(f.src)[s]; break;
            case TSM_KIND_I64: v = reinterpret_cast<const int64_t *>(f.src)[s]; break;
            default: v = reinterpret_cast<const uint8_t *>(f.src)[s]; break;
        }
        switch (f.dst_kind) {
            case TSM_KIND_I32: reinterpret_cast<int32_t *>(f.dst)[i] = (int32_t)v; break;
            case TSM_KIND_I64: reinterpret_cast<int64_t *>(f.dst)[i] = v; break;
            case TSM_KIND_F32: reinterpret_cast<float *>(f.dst)[i] = (float)v; break;
            default: reinterpret_cast<uint8_t *>(f.dst)[i] = (uint8_t)(v != 0); break;
        }
    }
}

}  // namespace

TSM_EXPORT int tsm_gather_fields(const tsm_gather_field *fields_host, int32_t n_fields, void *stream) {
    TSM_REQUIRE(fields_host && n_fields >= 1 && n_fields <= TSM_MAX_GATHER_FIELDS, "tsm_gather_fields: 1..%d fields", TSM_MAX_GATHER_FIELDS);
    GatherArgs a{};
    int64_t most = 0;
    for (int k = 0; k < n_fields; ++k) {
        const tsm_gather_field &f = fields_host[k];
        TSM_REQUIRE(f.n_rows >= 0 && f.width >= 1 && f.src_row_stride >= 0 && f.src_offset >= 0, "tsm_gather_fields: bad sizes in field %d", k);
        TSM_REQUIRE(f.n_rows == 0 || (f.src && f.dst), "tsm_gather_fields: null pointer in field %d", k);
        TSM_REQUIRE(f.src_kind >= 0 && f.src_kind <= TSM_KIND_U8 && f.dst_kind >= 0 && f.dst_kind <= TSM_KIND_U8,
                    "tsm_gather_fields: unknown element kind in field %d", k);
        TSM_REQUIRE(f.src_kind != TSM_KIND_F32 || f.dst_kind == TSM_KIND_F32, "tsm_gather_fields: f32 sources convert to f32 only (field %d)", k);
        TSM_REQUIRE(f.T == 0 || (f.T > 0 && f.E > 0 && f.n_rows == f.T * f.E), "tsm_gather_fields: n_rows != T * E in field %d", k);
        TSM_REQUIRE(f.n_rows * f.width < (1ll << 31), "tsm_gather_fields: field %d has 2^31 elements or more", k);
        a.f[k] = f;
        if (f.n_rows * f.width > most) most = f.n_rows * f.width;
    }
    if (most == 0) return TSM_OK;
    int64_t gx = ceil_div(most, 256 * 4);   // ~4 elements per thread of the largest field
    if (gx > 2048) gx = 2048;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(gather_fields_kernel, dim3((unsigned)gx, (unsigned)n_fields), dim3(256), 0, tsm_stream(stream), a);
    TSM_LAUNCH_CHECK();
    return TSM_OK;
}
