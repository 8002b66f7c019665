// klt.hip -- forward-backward pyramidal Lucas-Kanade, one keypoint per 64-lane wavefront (gfx950).
//
// Replaces (reference, /root/reference): FeatureTracker::fbKltTracking src/feature_tracker.cpp:35-137
// (= 2x cv::calcOpticalFlowPyrLK + the status / err / inBorder / forward-backward gates) and the two-stage
// batching of VisualFrontEnd::kltTracking src/visual_front_end.cpp:132-275.
// Arithmetic: OpenCV LKTrackerInvoker semantics as restated in oracle/ov2_oracle_fe.c -- 14-bit bilinear
// weights (v_rndne == cvRound), CV_DESCALE fixed point, EXACT integer sums for A11/A12/A22/b1/b2 converted to
// fp32 once (order independent, so the wave reduction is bit-identical to the oracle's scalar loop), fp32
// 2x2 solve with contraction off, fp64 for the two comparisons OpenCV does in double.
//
// Mapping: the (win x win <= 121) window pixels are spread over the 64 lanes (2 px per lane); the template
// (I, Ix, Iy as int16 values) lives in registers for the whole level; every LK iteration is
//   gather 2x4 u8 of J (L1/L2 resident, padded planes => no bounds logic)  ->  2 int32 MACs per lane
//   ->  DPP row reduction (quad_perm, row_half_mirror, row_mirror) + 4 v_readlane + scalar int64 add
//   ->  wave-uniform fp32 update.
// The whole pyramid loop, the gates and the backward pass run inside one launch; no host round trip.
#include "ov2_internal.h"

namespace {

struct klt_params {
    int win, nlevels, max_iter;
    double eps2;     // criteria.epsilon^2 (double, as in calcOpticalFlowPyrLK)
    float err_th;    // nklt_err
    double fb_th;    // fmax_fbklt_dist promoted to double for the cv::norm comparison
    float min_eig_thr;
};

#define W_BITS 14
__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

template <int CTRL>
__device__ __forceinline__ int dpp_add(int v)
{
    // v + (v moved by the DPP pattern); all lanes active, bound_ctrl irrelevant for these patterns
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}

// exact wave sum of one int32 per lane whose 16-lane partial sums fit int32; result as int64, wave-uniform
__device__ __forceinline__ long long wave_sum_i64(int v)
{
    v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);  // row_half_mirror
    v = dpp_add<0x140>(v);  // row_mirror  -> every lane holds its 16-lane row total
    const long long r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16),
                    r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ void lk_weights(float a, float b, int &w00, int &w01, int &w10, int &w11)
{
    w00 = (int)__builtin_rintf((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
    w01 = (int)__builtin_rintf(a * (1.f - b) * (float)(1 << W_BITS));
    w10 = (int)__builtin_rintf((1.f - a) * b * (float)(1 << W_BITS));
    w11 = (1 << W_BITS) - w00 - w01 - w10;
}

__device__ __forceinline__ int bilin_u8(const unsigned char *p, int stride, int w00, int w01, int w10, int w11)
{
    // two unaligned 16-bit loads: (x,y),(x+1,y) and (x,y+1),(x+1,y+1)
    unsigned short r0, r1;
    __builtin_memcpy(&r0, p, 2);
    __builtin_memcpy(&r1, p + stride, 2);
    return (r0 & 255) * w00 + (r0 >> 8) * w01 + (r1 & 255) * w10 + (r1 >> 8) * w11;
}

// One LKTrackerInvoker pass for the wave's keypoint on one level.  All scalar state is wave-uniform.
// returns the number of iterations executed.
__device__ __forceinline__ int lk_level(const unsigned char *__restrict__ Iimg, const short *__restrict__ Igrad,
                                        const unsigned char *__restrict__ Jimg, const ov2_level_desc &LI,
                                        const ov2_level_desc &LJ, int pad, int level, int max_level, float kx,
                                        float ky, float &nx_io, float &ny_io, int &status, float &err,
                                        const klt_params &P, int lane)
{
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    const int win = P.win;
    const float half = (float)(win - 1) * 0.5f;
    const float lscale = 1.f / (float)(1 << level);
    float px = kx * lscale, py = ky * lscale;
    float nx, ny;
    if (level == max_level) { nx = nx_io * lscale; ny = ny_io * lscale; }
    else { nx = nx_io * 2.f; ny = ny_io * 2.f; }
    nx_io = nx; ny_io = ny;

    px -= half; py -= half;
    const int ipx = (int)floorf(px), ipy = (int)floorf(py);
    if (ipx < -win || ipx >= LI.w || ipy < -win || ipy >= LI.h) {
        if (level == 0) { status = 0; err = 0.f; }
        return 0;
    }
    int w00, w01, w10, w11;
    lk_weights(px - (float)ipx, py - (float)ipy, w00, w01, w10, w11);

    // window pixel(s) of this lane
    const int npx = win * win;
    const int q0 = lane, q1 = lane + 64;
    const bool v0 = q0 < npx, v1 = q1 < npx;
    const int wy0 = q0 / win, wx0 = q0 - wy0 * win;
    const int wy1 = q1 / win, wx1 = q1 - wy1 * win;

    int Iv0 = 0, Ix0 = 0, Iy0 = 0, Iv1 = 0, Ix1 = 0, Iy1 = 0;
    {
        const int sI = LI.istride, sG = LI.gstride;
        if (v0) {
            const size_t r = (size_t)(wy0 + ipy + pad);
            const int c = OV2_LM + ipx + wx0;
            Iv0 = descale(bilin_u8(Iimg + r * sI + c, sI, w00, w01, w10, w11), W_BITS - 5);
            const int *g = reinterpret_cast<const int *>(Igrad) + r * sG + c;  // (Ix,Iy) packed in one dword
            const int g00 = g[0], g01 = g[1], g10 = g[sG], g11 = g[sG + 1];
            Ix0 = descale((short)(g00 & 0xffff) * w00 + (short)(g01 & 0xffff) * w01 + (short)(g10 & 0xffff) * w10 +
                          (short)(g11 & 0xffff) * w11, W_BITS);
            Iy0 = descale((g00 >> 16) * w00 + (g01 >> 16) * w01 + (g10 >> 16) * w10 + (g11 >> 16) * w11, W_BITS);
        }
        if (v1) {
            const size_t r = (size_t)(wy1 + ipy + pad);
            const int c = OV2_LM + ipx + wx1;
            Iv1 = descale(bilin_u8(Iimg + r * sI + c, sI, w00, w01, w10, w11), W_BITS - 5);
            const int *g = reinterpret_cast<const int *>(Igrad) + r * sG + c;
            const int g00 = g[0], g01 = g[1], g10 = g[sG], g11 = g[sG + 1];
            Ix1 = descale((short)(g00 & 0xffff) * w00 + (short)(g01 & 0xffff) * w01 + (short)(g10 & 0xffff) * w10 +
                          (short)(g11 & 0xffff) * w11, W_BITS);
            Iy1 = descale((g00 >> 16) * w00 + (g01 >> 16) * w01 + (g10 >> 16) * w10 + (g11 >> 16) * w11, W_BITS);
        }
    }
    // |Ix|,|Iy| <= 4080 for u8 images: 2 products per lane and 16-lane partial sums fit int32 exactly
    const long long sA11 = wave_sum_i64(Ix0 * Ix0 + Ix1 * Ix1);
    const long long sA12 = wave_sum_i64(Ix0 * Iy0 + Ix1 * Iy1);
    const long long sA22 = wave_sum_i64(Iy0 * Iy0 + Iy1 * Iy1);
    const float A11 = (float)(double)sA11 * FLT_SCALE;
    const float A12 = (float)(double)sA12 * FLT_SCALE;
    const float A22 = (float)(double)sA22 * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    const float min_eig = __fdiv_rn(A22 + A11 - __fsqrt_rn((A11 - A22) * (A11 - A22) + 4.f * A12 * A12),
                                    (float)(2 * win * win));
    err = min_eig;  // OPTFLOW_LK_GET_MIN_EIGENVALS
    if (min_eig < P.min_eig_thr || D < 1.1920929e-07f /* FLT_EPSILON */) {
        if (level == 0) status = 0;
        return 0;
    }
    D = __fdiv_rn(1.f, D);

    nx -= half; ny -= half;
    float pdx = 0.f, pdy = 0.f;
    const int sJ = LJ.istride;
    int j;
    for (j = 0; j < P.max_iter; ++j) {
        const int inx = (int)floorf(nx), iny = (int)floorf(ny);
        if (inx < -win || inx >= LJ.w || iny < -win || iny >= LJ.h) {
            if (level == 0) status = 0;
            break;
        }
        lk_weights(nx - (float)inx, ny - (float)iny, w00, w01, w10, w11);
        int pb1 = 0, pb2 = 0;
        if (v0) {
            const int diff = descale(bilin_u8(Jimg + (size_t)(wy0 + iny + pad) * sJ + OV2_LM + inx + wx0, sJ, w00, w01,
                                              w10, w11), W_BITS - 5) - Iv0;
            pb1 = diff * Ix0; pb2 = diff * Iy0;
        }
        if (v1) {
            const int diff = descale(bilin_u8(Jimg + (size_t)(wy1 + iny + pad) * sJ + OV2_LM + inx + wx1, sJ, w00, w01,
                                              w10, w11), W_BITS - 5) - Iv1;
            pb1 += diff * Ix1; pb2 += diff * Iy1;
        }
        // |diff| <= 8160, |Ix| <= 4080: 2 products/lane <= 6.7e7, 16-lane partials <= 1.07e9 fit int32
        const float b1 = (float)(double)wave_sum_i64(pb1) * FLT_SCALE;
        const float b2 = (float)(double)wave_sum_i64(pb2) * FLT_SCALE;
        const float dx = (A12 * b2 - A22 * b1) * D;
        const float dy = (A12 * b1 - A11 * b2) * D;
        nx += dx; ny += dy;
        nx_io = nx + half; ny_io = ny + half;
        if ((double)dx * dx + (double)dy * dy <= P.eps2) { ++j; break; }
        // std::abs(float) < 0.01 (double literal)  <=>  <= 0.01f
        if (j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) {
            nx_io -= dx * 0.5f; ny_io -= dy * 0.5f;
            ++j; break;
        }
        pdx = dx; pdy = dy;
    }
    return j;
}

struct plane_ptrs {
    const unsigned char *img;
    const short *grad;
};

__device__ __forceinline__ plane_ptrs level_planes(const ov2_pyr_view &v, int l, int b)
{
    plane_ptrs p;
    p.img = v.base + v.lv[l].img_off + v.lv[l].img_bstride * b;
    p.grad = reinterpret_cast<const short *>(v.base + v.lv[l].grad_off + v.lv[l].grad_bstride * b);
    return p;
}

// FeatureTracker::fbKltTracking for the wave's keypoint. returns status (0/1); fx,fy = forward result.
__device__ __forceinline__ int fb_track_one(const ov2_pyr_view &pv, const ov2_pyr_view &cv, int b, float kx, float ky,
                                            float &fx, float &fy, const klt_params &P, int nlevels, int lane,
                                            unsigned &iters)
{
    int status = 1;
    float err = 0.f;
    unsigned it = 0;
    for (int l = nlevels; l >= 0; --l) {
        const plane_ptrs I = level_planes(pv, l, b), J = level_planes(cv, l, b);
        it += lk_level(I.img, I.grad, J.img, pv.lv[l], cv.lv[l], pv.pad, l, nlevels, kx, ky, fx, fy, status, err, P, lane);
        it += 1u << 16;  // one level pass (template fetch) -- high half of the work word
    }
    // gates of src/feature_tracker.cpp:79-101
    const int W0 = cv.lv[0].w, H0 = cv.lv[0].h;
    int ok = status && !(err > P.err_th) &&
             (1.f <= fx && fx < (float)W0 - 1.f && 1.f <= fy && fy < (float)H0 - 1.f);
    if (ok) {
        // backward pass cur -> prev on level 0 from the original keypoint (src/feature_tracker.cpp:113)
        int st2 = 1;
        float e2 = 0.f, bx = kx, by = ky;
        const plane_ptrs I = level_planes(cv, 0, b), J = level_planes(pv, 0, b);
        it += lk_level(I.img, I.grad, J.img, cv.lv[0], pv.lv[0], cv.pad, 0, 0, fx, fy, bx, by, st2, e2, P, lane);
        it += 1u << 16;
        if (!st2) ok = 0;
        else {
            const float dx = kx - bx, dy = ky - by;
            const double nrm = __dsqrt_rn((double)dx * dx + (double)dy * dy);  // cv::norm(Point2f) is double
            if (nrm > P.fb_th) ok = 0;
        }
    }
    iters = it;
    return ok;
}

__global__ __launch_bounds__(64) void klt_fb_kernel(ov2_pyr_view pv, ov2_pyr_view cv, klt_params P, int n,
                                                    const float2 *__restrict__ kps, float2 *__restrict__ priors,
                                                    unsigned char *__restrict__ status,
                                                    const int *__restrict__ img_idx, unsigned *__restrict__ iters)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    const int b = img_idx ? img_idx[i] : 0;
    const float2 kp = kps[i];
    float2 pr = priors[i];
    unsigned it = 0;
    const int ok = fb_track_one(pv, cv, b, kp.x, kp.y, pr.x, pr.y, P, P.nlevels, lane, it);
    if (lane == 0) {
        priors[i] = pr;
        status[i] = (unsigned char)ok;
        if (iters) iters[i] = it;
    }
}

// ---- VisualFrontEnd::kltTracking, two stages without a host round trip -------------------------------
// stage 1: keypoints with a prior, 2 pyramid levels (nbpyrlvl = 1, src/visual_front_end.cpp:190)
__global__ __launch_bounds__(64) void klt_stage1_kernel(ov2_pyr_view pv, ov2_pyr_view cv, klt_params P, int n,
                                                        const float2 *__restrict__ kps,
                                                        const float2 *__restrict__ prior,
                                                        const unsigned char *__restrict__ has_prior,
                                                        const int *__restrict__ img_idx, float2 *__restrict__ out_xy,
                                                        unsigned char *__restrict__ out_status,
                                                        unsigned *__restrict__ counts /* [batch][64] slots: low 16 bits n3d, high 16 good */,
                                                        unsigned *__restrict__ iters)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    const float2 kp = kps[i];
    if (!has_prior[i]) {
        if (lane == 0) { out_xy[i] = kp; out_status[i] = 0; if (iters) iters[i] = 0; }
        return;
    }
    const int b = img_idx ? img_idx[i] : 0;
    float2 pr = prior[i];
    unsigned it = 0;
    const int nl = min(1, pv.nlevels - 1);
    const int ok = fb_track_one(pv, cv, b, kp.x, kp.y, pr.x, pr.y, P, nl, lane, it);
    if (lane == 0) {
        out_xy[i] = pr;  // tracked position, or the failed forward result that seeds stage 2 (:217-219)
        out_status[i] = (unsigned char)ok;
        if (iters) iters[i] = it;
        // one atomic per keypoint, spread over 64 slots per image: same-address atomics serialise at ~10-16 ns
        // each (measured: they, not the tracking, bounded this kernel when every wave hit one counter)
        atomicAdd(&counts[64 * b + (i & 63)], 1u + ((unsigned)ok << 16));
    }
}

// stage 2: keypoints without prior + stage-1 failures, full pyramid (src/visual_front_end.cpp:237-270)
__global__ __launch_bounds__(64) void klt_stage2_kernel(ov2_pyr_view pv, ov2_pyr_view cv, klt_params P, int n,
                                                        const float2 *__restrict__ kps,
                                                        const unsigned char *__restrict__ has_prior,
                                                        const int *__restrict__ img_idx, float2 *__restrict__ out_xy,
                                                        unsigned char *__restrict__ out_status,
                                                        const unsigned *__restrict__ counts, int *__restrict__ p3p_req,
                                                        unsigned *__restrict__ iters)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    const int b = img_idx ? img_idx[i] : 0;
    const unsigned cw = counts[64 * b + lane];
    const int n3 = (int)wave_sum_i64((int)(cw & 0xffffu)), good = (int)wave_sum_i64((int)(cw >> 16));
    const bool drop = n3 > 0 && (double)good < 0.33 * (double)n3;  // :228
    if (lane == 0 && p3p_req && drop) p3p_req[b] = 1;
    const bool hp = has_prior[i] != 0;
    if (hp && out_status[i]) {  // tracked in stage 1
        if (lane == 0 && iters) iters[n + i] = 0;
        return;
    }
    const float2 kp = kps[i];
    float2 pr = (hp && !drop) ? out_xy[i] : kp;
    unsigned it = 0;
    const int ok = fb_track_one(pv, cv, b, kp.x, kp.y, pr.x, pr.y, P, P.nlevels, lane, it);
    if (lane == 0) {
        out_xy[i] = pr;
        out_status[i] = (unsigned char)ok;
        if (iters) iters[n + i] = it;  // second half of the work-word array = stage 2
    }
}

ov2_status make_params(ov2_ctx *c, const ov2_pyr *prev, const ov2_pyr *cur, int win, int nlevels, int max_iter,
                       float eps, float err_th, float fb_th, klt_params *P)
{
    if (!prev || !cur) return ov2_set_err(c, OV2_ERR_INVALID, "null pyramid");
    const ov2_pyr_view &a = prev->buf->view, &b = cur->buf->view;
    // assert(vprevpyr.size() == vcurpyr.size())  src/feature_tracker.cpp:41
    if (a.nlevels != b.nlevels || a.pad != b.pad || a.batch != b.batch || a.lv[0].w != b.lv[0].w ||
        a.lv[0].h != b.lv[0].h)
        return ov2_set_err(c, OV2_ERR_INVALID, "prev/cur pyramids differ in geometry");
    if (win < 3 || win > 11 || (win & 1) == 0 || win > a.pad)
        return ov2_set_err(c, OV2_ERR_INVALID, "win=%d unsupported (odd, 3..11, <= pyramid pad %d)", win, a.pad);
    if (nlevels < 0) return ov2_set_err(c, OV2_ERR_INVALID, "nlevels < 0");
    if (a.nlevels < nlevels + 1) nlevels = a.nlevels - 1;  // :50-52
    P->win = win;
    P->nlevels = nlevels;
    P->max_iter = max_iter < 0 ? 0 : (max_iter > 100 ? 100 : max_iter);  // TermCriteria clamps
    double e = (double)eps;
    if (e < 0.) e = 0.;
    if (e > 10.) e = 10.;
    P->eps2 = e * e;
    P->err_th = err_th;
    P->fb_th = (double)fb_th;
    P->min_eig_thr = 1e-4f;  // calcOpticalFlowPyrLK default minEigThreshold, not passed by the reference
    return OV2_OK;
}

}  // namespace

extern "C" ov2_status ov2_klt_track_fb_dev(ov2_ctx *c, const ov2_pyr *prev, const ov2_pyr *cur, int win, int nlevels,
                                           int max_iter, float eps, float err_th, float fb_th, int n,
                                           const float *d_kps, float *d_priors, uint8_t *d_status,
                                           const int32_t *d_img_idx, uint32_t *d_iters)
{
    if (!c) return OV2_ERR_INVALID;
    if (n == 0) return OV2_OK;  // src/feature_tracker.cpp:43-46
    if (n < 0 || !d_kps || !d_priors || !d_status) return ov2_set_err(c, OV2_ERR_INVALID, "null/negative argument");
    klt_params P;
    ov2_status s = make_params(c, prev, cur, win, nlevels, max_iter, eps, err_th, fb_th, &P);
    if (s != OV2_OK) return s;
    OV2_HIP(c, hipSetDevice(c->device));
    OV2_LAUNCH(c, OV2_K_KLT_FB, klt_fb_kernel, dim3(n), dim3(64), 0, c->stream, prev->buf->view, cur->buf->view, P, n,
                       reinterpret_cast<const float2 *>(d_kps), reinterpret_cast<float2 *>(d_priors), d_status,
                       d_img_idx, d_iters);
    OV2_HIP(c, hipGetLastError());
    return OV2_OK;
}

extern "C" ov2_status ov2_klt_track_fb(ov2_ctx *c, const ov2_pyr *prev, const ov2_pyr *cur, int win, int nlevels,
                                       int max_iter, float eps, float err_th, float fb_th, int n, const float *kps,
                                       float *priors, uint8_t *status)
{
    if (!c) return OV2_ERR_INVALID;
    if (n == 0) return OV2_OK;
    if (n < 0 || !kps || !priors || !status) return ov2_set_err(c, OV2_ERR_INVALID, "null/negative argument");
    const size_t nb = (size_t)n * 8;
    void *scr = nullptr;
    ov2_status s = ov2_scratch(c, 2 * nb + (size_t)n + 64, &scr);
    if (s != OV2_OK) return s;
    float *d_kps = (float *)scr, *d_pri = (float *)((char *)scr + nb);
    uint8_t *d_st = (uint8_t *)scr + 2 * nb;
    OV2_HIP(c, hipMemcpyAsync(d_kps, kps, nb, hipMemcpyHostToDevice, c->stream));
    OV2_HIP(c, hipMemcpyAsync(d_pri, priors, nb, hipMemcpyHostToDevice, c->stream));
    s = ov2_klt_track_fb_dev(c, prev, cur, win, nlevels, max_iter, eps, err_th, fb_th, n, d_kps, d_pri, d_st, nullptr,
                             nullptr);
    if (s != OV2_OK) return s;
    OV2_HIP(c, hipMemcpyAsync(priors, d_pri, nb, hipMemcpyDeviceToHost, c->stream));
    OV2_HIP(c, hipMemcpyAsync(status, d_st, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    OV2_HIP(c, hipStreamSynchronize(c->stream));
    return OV2_OK;
}

extern "C" ov2_status ov2_klt_tracking_frame_dev(ov2_ctx *c, const ov2_pyr *prev, const ov2_pyr *cur, int win,
                                                 int nlevels_full, int max_iter, float eps, float err_th, float fb_th,
                                                 int n, const float *d_kps, const float *d_prior,
                                                 const uint8_t *d_has_prior, const int32_t *d_img_idx, float *d_out_xy,
                                                 uint8_t *d_out_status, int32_t *d_p3p_req, uint32_t *d_iters)
{
    if (!c) return OV2_ERR_INVALID;
    if (n == 0) return OV2_OK;
    if (n < 0 || !d_kps || !d_prior || !d_has_prior || !d_out_xy || !d_out_status)
        return ov2_set_err(c, OV2_ERR_INVALID, "null/negative argument");
    klt_params P;
    ov2_status s = make_params(c, prev, cur, win, nlevels_full, max_iter, eps, err_th, fb_th, &P);
    if (s != OV2_OK) return s;
    OV2_HIP(c, hipSetDevice(c->device));
    const int B = prev->buf->batch;
    void *scr = nullptr;
    s = ov2_scratch(c, (size_t)B * 64 * sizeof(unsigned) + 256, &scr);
    if (s != OV2_OK) return s;
    unsigned *counts = (unsigned *)scr;
    OV2_HIP(c, hipMemsetAsync(counts, 0, (size_t)B * 64 * sizeof(unsigned), c->stream));
    if (d_p3p_req) OV2_HIP(c, hipMemsetAsync(d_p3p_req, 0, (size_t)B * sizeof(int), c->stream));
    OV2_LAUNCH(c, OV2_K_KLT_STAGE1, klt_stage1_kernel, dim3(n), dim3(64), 0, c->stream, prev->buf->view, cur->buf->view, P, n,
                       reinterpret_cast<const float2 *>(d_kps), reinterpret_cast<const float2 *>(d_prior), d_has_prior,
                       d_img_idx, reinterpret_cast<float2 *>(d_out_xy), d_out_status, counts, d_iters);
    OV2_LAUNCH(c, OV2_K_KLT_STAGE2, klt_stage2_kernel, dim3(n), dim3(64), 0, c->stream, prev->buf->view, cur->buf->view, P, n,
                       reinterpret_cast<const float2 *>(d_kps), d_has_prior, d_img_idx,
                       reinterpret_cast<float2 *>(d_out_xy), d_out_status, counts, d_p3p_req, d_iters);
    OV2_HIP(c, hipGetLastError());
    return OV2_OK;
}
