// ov2_internal.h -- shared state of libov2hip.so (not installed; the public ABI is include/ov2slam_hip.h)
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ov2slam_hip.h"

#define OV2_MAX_LEVELS 8
// left margin of every padded plane, in pixels: >= win (<= 11) and a multiple of 16 so that the interior
// of a row starts 16-byte (u8 plane) / 64-byte (s16x2 plane) aligned for vector loads and stores.
#define OV2_LM 16

struct ov2_level_desc {
    int w, h;             // unpadded size
    int istride;          // bytes per row of the u8 plane (multiple of 64)
    int gstride;          // int16x2 elements per row of the gradient plane (multiple of 16)
    int rows;             // allocated rows = h + 2*pad (+ slack)
    size_t img_off;       // byte offset of image plane of batch 0 inside the allocation
    size_t grad_off;      // byte offset of gradient plane of batch 0
    size_t img_bstride;   // bytes between consecutive batch entries (image plane)
    size_t grad_bstride;  // bytes between consecutive batch entries (gradient plane)
};

// what kernels see (passed by value)
struct ov2_pyr_view {
    int nlevels, pad, batch;
    unsigned char *base;
    ov2_level_desc lv[OV2_MAX_LEVELS];
};

struct ov2_pyr_buf {  // pooled allocation; geometry key = (w,h,pad,max_level,batch)
    int w, h, pad, max_level, batch;
    size_t bytes;
    unsigned char *base;
    unsigned char *lut;  // CLAHE LUTs: batch * tiles * 256 (allocated lazily, max tiles 64x64)
    hipEvent_t ready_ev; // recorded on the ctx pyramid stream when the build is enqueued; consumers wait on it
    hipEvent_t free_ev;  // recorded on the ctx main stream when the last handle is released; the next build waits on it
    bool has_free_ev;
    hipEvent_t free_ev2; // recorded on ANOTHER context's main stream by ov2_pyr_release_from (the mapper's readers); the next build waits on it too
    bool has_free_ev2;
    // the int16 (Ix, Iy) planes are written on demand (ov2_pyr_need_grad): the 9 x 9 tracking path derives the Scharr
    // values inside its kernel and never reads them.  Both fields are guarded by the owning ctx's mutex.
    bool grad_built;
    hipEvent_t grad_ev;  // recorded behind the kernels that wrote the gradient planes
    ov2_pyr_view view;
};

enum ov2_kernel_id {
    OV2_K_CLAHE_LUT = 0, OV2_K_LEVEL0, OV2_K_LEVEL, OV2_K_KLT_FB, OV2_K_KLT_STAGE1, OV2_K_KLT_STAGE2,
    OV2_K_BA_FIRST,  // BA kernels register from here (ba.hip): 12 ids
    OV2_K_DETECT = 20,
    OV2_K_MAP = 26,  // map mirror scans (map.hip)
    OV2_K_MAX = 48
};
extern const char *ov2_kernel_names[OV2_K_MAX];

struct ov2_ktime_rec {
    int id;
    hipEvent_t e0, e1;
};

struct ov2_ctx {
    int device;
    hipStream_t stream;                  // main stream: KLT, detectors, BA
    hipStream_t stream_pyr;              // pyramid builds run here so that frame t+1's pyramid overlaps frame t's KLT
    hipEvent_t ev0, ev1;
    std::mutex mu;                       // guards pool + err
    std::vector<ov2_pyr_buf *> pool;     // free pyramid buffers
    std::vector<struct ov2_map *> maps;  // live map mirrors (their device memory is released with the ctx)
    std::string err;
    // scratch for host-pointer entry points (grown on demand)
    void *scratch_dev;
    size_t scratch_bytes;
    ov2_images *tmp_img;                 // staging image for ov2_pyramid_build(host img)
    void *ba_arena;                      // device arena of ov2_ba_solve, grown on demand, kept across solves
    size_t ba_arena_cap;
    void *stage_host, *stage_dev;        // pinned staging block + its device twin for the host-pointer entry points
    size_t stage_cap;
    hipEvent_t stage_ev;                 // recorded behind an asynchronous upload out of the pinned staging block;
    bool stage_ev_pending;               //   ov2_staging waits for it before handing the block out again
    void *ba_arena2;                     // second device block: cell / pair structure of the Schur complement (sized after the build learns the counts)
    size_t ba_arena2_cap;
    void *ba_host;                       // pinned host mirror of the uploaded head of the arena (same offsets)
    size_t ba_host_cap;
    unsigned *klt_counts[2];             // tallies + list lengths of the two-stage KLT, double-buffered: the last kernel of a call
    size_t klt_counts_words;             //   zeroes the OTHER buffer for the next call (no memset in front of every frame)
    int klt_counts_cur;
    bool klt_counts_dirty;               // a call did not run to its end: both buffers are zeroed again
    hipStream_t ba_copy_stream;          // second half of a batch upload (measurements) travels here while the program build sorts
    hipEvent_t ba_copy_ev[2];            // [0] head uploaded (main stream), [1] measurements uploaded (copy stream)
    void *klt_ybuf; size_t klt_ybuf_bytes;   // straggler records of the tracking kernels (written only by them)
    int klt_epoch;                       // serial number of the two-stage tracking calls (publication word of the straggler records)
    int klt_yield_after, klt_yield_groups, klt_yield_pickup;   // ov2_klt_set_yield (three-lane tracking kernels: klt_rec in klt.hip)
    int klt_lanes;                       // ov2_klt_set_lanes: 0 = by call size, 3 / 8 / 16 = forced lanes per keypoint
    // optional per-kernel hipEvent timing (bench.py roofline leg); off by default
    bool ktime_on;
    std::vector<ov2_ktime_rec> ktime_recs;   // recorded (kernel id, event pair) since the last report
    std::vector<hipEvent_t> ktime_free;      // recycled events
};

struct ov2_images {
    ov2_ctx *ctx;
    int batch, w, h;
    int stride;        // bytes per row (multiple of 64)
    size_t bstride;    // bytes per image
    unsigned char *base;
};

struct ov2_pyr {
    std::atomic<int> refs;
    ov2_ctx *ctx;
    ov2_pyr_buf *buf;
};

ov2_status ov2_set_err(ov2_ctx *ctx, ov2_status s, const char *fmt, ...);
ov2_status ov2_scratch(ov2_ctx *ctx, size_t bytes, void **out);
// pinned host block of `bytes` and a device block of the same size (grown on demand, kept by the ctx)
ov2_status ov2_staging(ov2_ctx *ctx, size_t bytes, void **host, void **dev);
// ov2_ctx_destroy: free the device side of a map that outlives its ctx (the handle stays valid for ov2_map_destroy)
void ov2_map_orphan(struct ov2_map *m);

#define OV2_HIP(ctx, call)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return ov2_set_err((ctx), OV2_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                               __FILE__, __LINE__);                                                 \
    } while (0)

// bracket a launch with events when kernel timing is enabled
void ov2_ktime_begin(ov2_ctx *c, int id, hipStream_t st);
void ov2_ktime_end(ov2_ctx *c, hipStream_t st);
// make the ctx main stream wait for the build of `p` (enqueued on the pyramid stream)
ov2_status ov2_pyr_wait_ready(ov2_ctx *c, const ov2_pyr *p);
// ... and, for consumers of the gradient planes, for those planes as well (writes them on c's main stream if no one has yet)
ov2_status ov2_pyr_need_grad(ov2_ctx *c, const ov2_pyr *p);
// klt.hip: two-stage forward-backward tracking (see ov2_klt_tracking_frame_dev); rule33 = 0 for stereo matching
ov2_status ov2_klt_two_stage_dev(ov2_ctx *c, const ov2_pyr *prev, const ov2_pyr *cur, int win, int nlevels_full, int max_iter,
                                 float eps, float err_th, float fb_th, int n, const float *d_kps, const float *d_prior,
                                 const uint8_t *d_has_prior, const int32_t *d_img_idx, float *d_out_xy,
                                 uint8_t *d_out_status, int32_t *d_p3p_req, uint32_t *d_iters, int rule33);
#define OV2_LAUNCH_ON(ctx, id, st, ...)     \
    do {                                    \
        if ((ctx)->ktime_on) ov2_ktime_begin((ctx), (id), (st)); \
        hipLaunchKernelGGL(__VA_ARGS__);    \
        if ((ctx)->ktime_on) ov2_ktime_end((ctx), (st)); \
    } while (0)
#define OV2_LAUNCH(ctx, id, ...) OV2_LAUNCH_ON(ctx, id, (ctx)->stream, __VA_ARGS__)

static inline int ov2_round_up(int v, int m) { return (v + m - 1) / m * m; }
