// ctx.hip -- context, device memory, image batches, pooled pyramid buffers of libov2hip.so
#include "ov2_internal.h"

#include <cstdlib>

ov2_status ov2_set_err(ov2_ctx *ctx, ov2_status s, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) {
        std::lock_guard<std::mutex> g(ctx->mu);
        ctx->err = buf;
    }
    return s;
}

extern "C" const char *ov2_status_string(ov2_status s)
{
    switch (s) {
    case OV2_OK: return "ok";
    case OV2_ERR_INVALID: return "invalid argument";
    case OV2_ERR_HIP: return "HIP runtime error";
    case OV2_ERR_NOMEM: return "out of memory";
    case OV2_ERR_NODEVICE: return "no gfx950 device";
    case OV2_ERR_UNSUPPORTED: return "unsupported";
    default: return "unknown";
    }
}

extern "C" const char *ov2_last_error(const ov2_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

extern "C" ov2_status ov2_ctx_create(int device, ov2_ctx **out) { return ov2_ctx_create_ex(device, 0, out); }

extern "C" ov2_status ov2_ctx_create_ex(int device, int high_priority, ov2_ctx **out)
{
    if (!out) return OV2_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return OV2_ERR_NODEVICE;  // never a CPU fallback
    if (device < 0 || device >= ndev) return OV2_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return OV2_ERR_HIP;
    ov2_ctx *c = new ov2_ctx();
    c->device = device;
    c->scratch_dev = nullptr;
    c->scratch_bytes = 0;
    c->tmp_img = nullptr;
    c->ktime_on = false;
    c->klt_lanes = 0;
    c->ba_arena = nullptr;
    c->ba_arena_cap = 0;
    c->ba_host = nullptr;
    c->ba_host_cap = 0;
    c->ba_copy_stream = nullptr;
    c->klt_counts[0] = c->klt_counts[1] = nullptr;
    c->klt_counts_words = 0;
    c->klt_counts_cur = 0;
    c->klt_counts_dirty = true;
    c->ba_copy_ev[0] = c->ba_copy_ev[1] = nullptr;
    c->ba_arena2 = nullptr;
    c->ba_arena2_cap = 0;
    c->stage_host = c->stage_dev = nullptr;
    c->stage_cap = 0;
    c->stage_ev = nullptr;
    {   // the tracking kernels' yield / resume (OV2_KLT_YIELD=after,groups switches it on; default 0,0 = off)
        c->klt_epoch = 0; c->klt_ybuf = nullptr; c->klt_ybuf_bytes = 0;
        c->klt_yield_after = 0; c->klt_yield_groups = 0; c->klt_yield_pickup = 0;   // off: every variant measured slower than waiting for the stragglers (klt.hip, klt_rec)
        const char *e = getenv("OV2_KLT_YIELD");
        int a = 0, g = 0;
        if (e && sscanf(e, "%d,%d", &a, &g) == 2 && a >= 0 && a <= 100 && g >= 0 && g <= 20) { c->klt_yield_after = g > 0 ? a : 0; c->klt_yield_groups = g; }
        const char *pk = getenv("OV2_KLT_PICKUP");
        if (pk && atoi(pk) >= 0) c->klt_yield_pickup = atoi(pk);
    }
    c->stage_ev_pending = false;
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    // experiment hook: OV2_CU_SPLIT=k keeps the first k compute units of the device for the high-priority contexts
    // (the local-BA worker) and the remaining ones for the others, instead of sharing all of them by priority
    const char *split_env = getenv("OV2_CU_SPLIT");
    const int split = split_env ? atoi(split_env) : 0;
    hipError_t e1, e2;
    if (split > 0) {
        hipDeviceProp_t prop;
        (void)hipGetDeviceProperties(&prop, device);
        const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
        std::vector<uint32_t> mask(words, 0u);
        for (int i = 0; i < ncu; ++i) {
            const bool mine = high_priority ? (i < split) : (i >= split);
            if (mine) mask[i / 32] |= 1u << (i % 32);
        }
        e1 = hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)words, mask.data());
        e2 = hipExtStreamCreateWithCUMask(&c->stream_pyr, (uint32_t)words, mask.data());
    } else {
        e1 = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, high_priority ? prio_greatest : prio_least);
        // experiment hook: OV2_SINGLE_STREAM=1 builds the pyramids on the main stream (no overlap with the tracking)
        const char *one = getenv("OV2_SINGLE_STREAM");
        if (one && atoi(one) > 0) { c->stream_pyr = c->stream; e2 = hipSuccess; }
        else e2 = hipStreamCreateWithPriority(&c->stream_pyr, hipStreamNonBlocking, high_priority ? prio_greatest : prio_least);
    }
    if (e1 != hipSuccess || e2 != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&c->stage_ev, hipEventDisableTiming) != hipSuccess) {
        delete c;
        return OV2_ERR_HIP;
    }
    *out = c;
    return OV2_OK;
}

extern "C" void ov2_ctx_destroy(ov2_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->stream_pyr);
    for (ov2_pyr_buf *b : c->pool) {
        (void)hipEventDestroy(b->ready_ev);
        (void)hipEventDestroy(b->free_ev);
        (void)hipEventDestroy(b->free_ev2);
        (void)hipFree(b->base);
        if (b->lut) (void)hipFree(b->lut);
        delete b;
    }
    for (ov2_map *m : c->maps) ov2_map_orphan(m);
    if (c->tmp_img) ov2_images_destroy(c->tmp_img);
    for (auto &r : c->ktime_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto e : c->ktime_free) (void)hipEventDestroy(e);
    if (c->scratch_dev) (void)hipFree(c->scratch_dev);
    if (c->ba_arena) (void)hipFree(c->ba_arena);
    if (c->ba_host) (void)hipHostFree(c->ba_host);
    for (int i = 0; i < 2; ++i) if (c->klt_counts[i]) (void)hipFree(c->klt_counts[i]);
    if (c->ba_copy_stream) { (void)hipStreamSynchronize(c->ba_copy_stream); (void)hipStreamDestroy(c->ba_copy_stream); }
    for (int i = 0; i < 2; ++i) if (c->ba_copy_ev[i]) (void)hipEventDestroy(c->ba_copy_ev[i]);
    if (c->ba_arena2) (void)hipFree(c->ba_arena2);
    if (c->klt_ybuf) (void)hipFree(c->klt_ybuf);
    if (c->stage_host) (void)hipHostFree(c->stage_host);
    if (c->stage_dev) (void)hipFree(c->stage_dev);
    (void)hipEventDestroy(c->ev0);
    (void)hipEventDestroy(c->ev1);
    if (c->stage_ev) (void)hipEventDestroy(c->stage_ev);
    (void)hipStreamDestroy(c->stream);
    if (c->stream_pyr != c->stream) (void)hipStreamDestroy(c->stream_pyr);
    delete c;
}

extern "C" ov2_status ov2_ctx_synchronize(ov2_ctx *c)
{
    if (!c) return OV2_ERR_INVALID;
    OV2_HIP(c, hipSetDevice(c->device));   // HIP's current device is per thread
    OV2_HIP(c, hipStreamSynchronize(c->stream_pyr));
    OV2_HIP(c, hipStreamSynchronize(c->stream));
    return OV2_OK;
}

extern "C" ov2_status ov2_timer_start(ov2_ctx *c)
{
    if (!c) return OV2_ERR_INVALID;
    OV2_HIP(c, hipSetDevice(c->device));   // HIP's current device is per thread
    OV2_HIP(c, hipEventRecord(c->ev0, c->stream));
    return OV2_OK;
}

extern "C" ov2_status ov2_timer_stop(ov2_ctx *c, float *ms)
{
    if (!c || !ms) return OV2_ERR_INVALID;
    OV2_HIP(c, hipSetDevice(c->device));   // HIP's current device is per thread
    OV2_HIP(c, hipEventRecord(c->ev1, c->stream));
    OV2_HIP(c, hipEventSynchronize(c->ev1));
    OV2_HIP(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return OV2_OK;
}

extern "C" ov2_status ov2_dev_alloc(ov2_ctx *c, size_t bytes, void **dptr)
{
    if (!c || !dptr) return OV2_ERR_INVALID;
    OV2_HIP(c, hipSetDevice(c->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) return ov2_set_err(c, OV2_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return OV2_OK;
}

extern "C" ov2_status ov2_dev_free(ov2_ctx *c, void *dptr)
{
    if (!c) return OV2_ERR_INVALID;
    OV2_HIP(c, hipSetDevice(c->device));   // HIP's current device is per thread
    if (dptr) OV2_HIP(c, hipFree(dptr));
    return OV2_OK;
}

extern "C" ov2_status ov2_memcpy_h2d(ov2_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c || (bytes && (!dst || !src))) return OV2_ERR_INVALID;
    OV2_HIP(c, hipSetDevice(c->device));   // HIP's current device is per thread
    OV2_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    OV2_HIP(c, hipStreamSynchronize(c->stream));
    return OV2_OK;
}

extern "C" ov2_status ov2_memcpy_d2h(ov2_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c || (bytes && (!dst || !src))) return OV2_ERR_INVALID;
    OV2_HIP(c, hipSetDevice(c->device));   // HIP's current device is per thread
    OV2_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    OV2_HIP(c, hipStreamSynchronize(c->stream));
    return OV2_OK;
}

extern "C" ov2_status ov2_memcpy_d2d(ov2_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c || (bytes && (!dst || !src))) return OV2_ERR_INVALID;
    OV2_HIP(c, hipSetDevice(c->device));   // HIP's current device is per thread
    OV2_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return OV2_OK;
}

ov2_status ov2_scratch(ov2_ctx *c, size_t bytes, void **out)
{
    if (bytes > c->scratch_bytes) {
        OV2_HIP(c, hipStreamSynchronize(c->stream));
        if (c->scratch_dev) OV2_HIP(c, hipFree(c->scratch_dev));
        c->scratch_dev = nullptr;
        c->scratch_bytes = 0;
        size_t want = bytes + bytes / 2 + 4096;
        hipError_t e = hipMalloc(&c->scratch_dev, want);
        if (e != hipSuccess) return ov2_set_err(c, OV2_ERR_NOMEM, "scratch hipMalloc(%zu)", want);
        c->scratch_bytes = want;
    }
    *out = c->scratch_dev;
    return OV2_OK;
}

ov2_status ov2_staging(ov2_ctx *c, size_t bytes, void **host, void **dev)
{
    if (c->stage_ev_pending) {   // an asynchronous call left an upload out of the pinned block in flight
        OV2_HIP(c, hipEventSynchronize(c->stage_ev));
        c->stage_ev_pending = false;
    }
    if (bytes > c->stage_cap) {
        OV2_HIP(c, hipStreamSynchronize(c->stream));
        if (c->stage_host) OV2_HIP(c, hipHostFree(c->stage_host));
        if (c->stage_dev) OV2_HIP(c, hipFree(c->stage_dev));
        c->stage_host = c->stage_dev = nullptr;
        c->stage_cap = 0;
        const size_t want = bytes + bytes / 2 + 4096;
        if (hipHostMalloc(&c->stage_host, want, hipHostMallocDefault) != hipSuccess || hipMalloc(&c->stage_dev, want) != hipSuccess)
            return ov2_set_err(c, OV2_ERR_NOMEM, "staging allocation of %zu bytes", want);
        c->stage_cap = want;
    }
    *host = c->stage_host;
    *dev = c->stage_dev;
    return OV2_OK;
}

// ---- images ---------------------------------------------------------------------------------------

extern "C" ov2_status ov2_images_create(ov2_ctx *c, int batch, int w, int h, ov2_images **out)
{
    if (!c || !out || batch <= 0 || w <= 0 || h <= 0) return OV2_ERR_INVALID;
    *out = nullptr;
    OV2_HIP(c, hipSetDevice(c->device));
    ov2_images *im = new ov2_images();
    im->ctx = c;
    im->batch = batch;
    im->w = w;
    im->h = h;
    im->stride = ov2_round_up(w, 64);
    im->bstride = (size_t)im->stride * h;
    hipError_t e = hipMalloc((void **)&im->base, im->bstride * batch);
    if (e != hipSuccess) {
        delete im;
        return ov2_set_err(c, OV2_ERR_NOMEM, "images hipMalloc: %s", hipGetErrorString(e));
    }
    *out = im;
    return OV2_OK;
}

extern "C" ov2_status ov2_images_upload(ov2_ctx *c, ov2_images *im, int b, const uint8_t *host, int stride)
{
    if (!c || !im || !host || b < 0 || b >= im->batch || stride < im->w) return OV2_ERR_INVALID;
    OV2_HIP(c, hipSetDevice(c->device));   // HIP's current device is per thread
    OV2_HIP(c, hipMemcpy2DAsync(im->base + im->bstride * b, im->stride, host, stride, im->w, im->h,
                                hipMemcpyHostToDevice, c->stream));
    OV2_HIP(c, hipStreamSynchronize(c->stream));
    return OV2_OK;
}

extern "C" void ov2_images_destroy(ov2_images *im)
{
    if (!im) return;
    (void)hipFree(im->base);
    delete im;
}

// ---- per-kernel event timing ------------------------------------------------------------------------

const char *ov2_kernel_names[OV2_K_MAX] = {"clahe_lut_kernel", "level0_kernel", "level_kernel", "klt_fb_kernel",
                                           "klt_stage1_kernel", "klt_stage2_kernel",
                                           "ba_eval_kernel", "ba_colnorm_kernel", "ba_scale_kernels", "ba_lmdiag_kernel",
                                           "ba_sinit_kernel", "ba_schur_kernel", "ba_chol_kernel", "ba_backsub_kernel",
                                           "ba_plus_kernel", "ba_flag_kernel", "ba_reduce_kernel", "ba_misc_kernels",
                                           nullptr, nullptr, "detect_cell_kernels", "detect_mask_kernel", "subpix_kernel",
                                           "pnp_kernel", "klt_compact_kernel",
                                           "detect_list_kernels", "map_setup_kernels", "tri_kernel", "stereo_sad_kernel",
                                           "stereo_gate_kernel", "brief_kernel", "match_kernels", "pose_graph_kernel"};

static hipEvent_t ktime_event(ov2_ctx *c)
{
    if (!c->ktime_free.empty()) {
        hipEvent_t e = c->ktime_free.back();
        c->ktime_free.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void ov2_ktime_begin(ov2_ctx *c, int id, hipStream_t st)
{
    ov2_ktime_rec r;
    r.id = id;
    r.e0 = ktime_event(c);
    r.e1 = ktime_event(c);
    (void)hipEventRecord(r.e0, st);
    c->ktime_recs.push_back(r);
}

void ov2_ktime_end(ov2_ctx *c, hipStream_t st) { (void)hipEventRecord(c->ktime_recs.back().e1, st); }

extern "C" ov2_status ov2_ktime_enable(ov2_ctx *c, int on)
{
    if (!c) return OV2_ERR_INVALID;
    c->ktime_on = on != 0;
    return OV2_OK;
}

extern "C" ov2_status ov2_ktime_report(ov2_ctx *c, int max_kernels, const char **names, double *total_ms,
                                       long long *launches, int *n_out)
{
    if (!c || !n_out || max_kernels < 0) return OV2_ERR_INVALID;
    OV2_HIP(c, hipStreamSynchronize(c->stream_pyr));
    OV2_HIP(c, hipStreamSynchronize(c->stream));
    double tot[OV2_K_MAX] = {0};
    long long cnt[OV2_K_MAX] = {0};
    for (auto &r : c->ktime_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess && r.id >= 0 && r.id < OV2_K_MAX) {
            tot[r.id] += ms;
            cnt[r.id] += 1;
        }
        c->ktime_free.push_back(r.e0);
        c->ktime_free.push_back(r.e1);
    }
    c->ktime_recs.clear();
    int n = 0;
    for (int i = 0; i < OV2_K_MAX && n < max_kernels; ++i) {
        if (!cnt[i]) continue;
        if (names) names[n] = ov2_kernel_names[i] ? ov2_kernel_names[i] : "?";
        if (total_ms) total_ms[n] = tot[i];
        if (launches) launches[n] = cnt[i];
        ++n;
    }
    *n_out = n;
    return OV2_OK;
}

// ---- diagnostics ------------------------------------------------------------------------------------------------
// PMC calibration of the KLT staging pattern (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own
// access pattern"): every lane of a wave loads 16 bytes from a DIFFERENT row (rows `stride` bytes apart, as the KLT
// kernels fetch one window row per lane), column by column, until the whole buffer has been read exactly once.
__global__ __launch_bounds__(64) void dbg_rowload16_kernel(const uint4 *__restrict__ buf, size_t stride16, size_t nrows,
                                                           unsigned *__restrict__ out)
{
    const size_t r = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (r >= nrows) return;
    const uint4 *row = buf + r * stride16;
    unsigned acc = 0;
    for (size_t c = 0; c < stride16; ++c) {
        const uint4 v = row[c];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    out[r] = acc;
}

extern "C" ov2_status ov2_dbg_rowload16(ov2_ctx *c, const void *d_buf, size_t stride_bytes, size_t nrows, uint32_t *d_out)
{
    if (!c || !d_buf || !d_out || stride_bytes % 16 || !nrows) return OV2_ERR_INVALID;
    OV2_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(dbg_rowload16_kernel, dim3((unsigned)((nrows + 63) / 64)), dim3(64), 0, c->stream, (const uint4 *)d_buf,
                       stride_bytes / 16, nrows, d_out);
    OV2_HIP(c, hipGetLastError());
    return OV2_OK;
}
