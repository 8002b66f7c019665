// stereo.hip -- keyframe-rate stereo matching around the KLT kernels (MapManager::stereoMatching, reference
// src/map_manager.cpp:367-611): the SAD line search that seeds rectified rigs (FeatureTracker::getLineMinSAD,
// src/feature_tracker.cpp:140-213, called at src/map_manager.cpp:429), the two fbKltTracking calls without the 33 %
// rule (:493-580, klt.hip) and the epipolar gate with the rectified row snap / the Sampson distance (:583-604,
// MultiViewGeometry::computeSampsonDistance src/multi_view_geometry.cpp:798-821).
// Integer SAD on 16-bit fixed-point bilinear samples (cv::getRectSubPix 8U -> 8U) => bit-exact against the oracle;
// the gate's float expressions are evaluated in the oracle's order (-ffp-contract=off).
#include "ov2_internal.h"
#include "ov2_cam.h"

namespace {

enum { K_SAD = OV2_K_MAP + 2, K_GATE = OV2_K_MAP + 3 };

// one candidate position's (or the template's) cv::getRectSubPix weights and source geometry
struct subpix_geo {
    int a11, a12, a21, a22, b1, b2;   // scale_fixpt(cvRound(v * 65536))
    int col0;                         // source column of window column 0 (after adjustRect)
    int rx, rw;                       // window columns [rx, rw) are interior, the rest reuse the edge column
};

__device__ __forceinline__ int cv_round_dev(float v) { return (int)__builtin_rintf(v); }   // v_rndne: round half to even

__device__ __forceinline__ subpix_geo make_geo(float cxc, float b, int w, int ww)
{
    subpix_geo g;
    const int ipx = (int)floorf(cxc);
    const float a = cxc - (float)ipx;
    g.a11 = cv_round_dev((1.f - a) * (1.f - b) * 65536.f);
    g.a12 = cv_round_dev(a * (1.f - b) * 65536.f);
    g.a21 = cv_round_dev((1.f - a) * b * 65536.f);
    g.a22 = cv_round_dev(a * b * 65536.f);
    g.b1 = cv_round_dev((1.f - b) * 65536.f);
    g.b2 = cv_round_dev(b * 65536.f);
    int off = 0;
    if (ipx >= 0) { off = ipx; g.rx = 0; }
    else { g.rx = -ipx; if (g.rx > ww) g.rx = ww; }
    if (ipx < w - ww) g.rw = ww;
    else {
        g.rw = w - ipx - 1;
        if (g.rw < 0) { off += g.rw; g.rw = 0; }
    }
    g.col0 = off - g.rx;
    return g;
}

// pixel (row pair A/B, window column j) of the sampled rectangle; A, B point at the two source rows
__device__ __forceinline__ int subpix_px(const subpix_geo &g, const unsigned char *A, const unsigned char *B, int j)
{
    int v;
    if (j < g.rx) v = A[g.col0 + g.rx] * g.b1 + B[g.col0 + g.rx] * g.b2;
    else if (j >= g.rw) v = A[g.col0 + g.rw] * g.b1 + B[g.col0 + g.rw] * g.b2;
    else {
        const int c = g.col0 + j;
        v = A[c] * g.a11 + A[c + 1] * g.a12 + B[c] * g.a21 + B[c + 1] * g.a22;
    }
    return (v + (1 << 15)) >> 16;   // cast_8u
}

// One wave per point.  LDS: the (ws + 1) source rows of the right image the search can touch (whole rows: every
// candidate reads the same rows at a different column offset), the row pair table and the template patch.
// Lane s, s + 64, ... evaluates candidate c = x -/+ s: its own fixed-point weights (the float chain c -= 1 is followed
// literally, so a candidate's fraction is whatever the reference's loop variable holds), ws x ws samples from LDS, exact
// integer SAD; the wave then takes the minimum of (sad / n as float, s) -- the reference keeps the FIRST minimum.
__global__ __launch_bounds__(64) void sad_kernel(ov2_pyr_view lv, ov2_pyr_view rv, int level, int nwinsize, int go_left, int n,
                                                 const float2 *__restrict__ pts, const int *__restrict__ img_idx,
                                                 float *__restrict__ xprior, float *__restrict__ l1err, int ws_max)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    const ov2_level_desc &LL = lv.lv[level], &LR = rv.lv[level];
    const int w = LL.w, h = LL.h, b = img_idx ? img_idx[i] : 0;
    const unsigned char *iml = lv.base + LL.img_off + LL.img_bstride * b + (size_t)lv.pad * LL.istride + OV2_LM;
    const unsigned char *imr = rv.base + LR.img_off + LR.img_bstride * b + (size_t)rv.pad * LR.istride + OV2_LM;
    const float x = pts[i].x, y = pts[i].y;
    float out_x = -1.f, out_e = 0.f;
    // the reference only calls with keypoints of the image; outside it the window arithmetic below is unbounded
    const bool inside = x >= 0.f && x < (float)w && y >= 0.f && y < (float)h && (nwinsize & 1);
    int halfwin = nwinsize / 2;
    if (inside) {   // src/feature_tracker.cpp:155-162, `int += float`
        if (x - (float)halfwin < 0) halfwin = (int)((float)halfwin + (x - (float)halfwin));
        if (x + (float)halfwin >= (float)w) halfwin = (int)((float)halfwin + (x + (float)halfwin - (float)w - 1.f));
        if (y - (float)halfwin < 0) halfwin = (int)((float)halfwin + (y - (float)halfwin));
        if (y + (float)halfwin >= (float)h) halfwin = (int)((float)halfwin + (y + (float)halfwin - (float)h - 1.f));
    }
    const int ws = 2 * halfwin + 1;
    if (!inside || halfwin <= 0 || ws > ws_max) {
        if (lane == 0) { xprior[i] = -1.f; if (l1err) l1err[i] = 0.f; }
        return;
    }
    // rows: same for the template and every candidate (same y)
    const float cyc = y - (float)(ws - 1) * 0.5f;
    const int ipy = (int)floorf(cyc);
    const float bfr = cyc - (float)ipy;
    int ry, rh, row0 = 0;
    if (ipy >= 0) { row0 = ipy; ry = 0; } else ry = -ipy;
    if (ipy < h - ws) rh = ws;
    else {
        rh = h - ipy - 1;
        if (rh < 0) { row0 += rh; rh = 0; }
    }
    // LDS layout: [rowA (ws ints) | rowB (ws ints) | patch (ws*ws bytes, padded to 4) | strip ((ws+1) rows x w bytes)]
    int *rowA = reinterpret_cast<int *>(lds), *rowB = rowA + ws_max;
    unsigned char *patch = reinterpret_cast<unsigned char *>(rowB + ws_max);
    unsigned char *strip = patch + ((ws_max * ws_max + 3) & ~3);
    if (lane == 0) {
        int cur = row0;
        for (int r = 0; r < ws; ++r) {
            const int nxt = (r < ry || r >= rh) ? cur : cur + 1;
            rowA[r] = cur; rowB[r] = nxt;
            if (r < rh) cur = nxt;
        }
    }
    __syncthreads();
    const int r_lo = rowA[0], r_hi = rowB[ws - 1];
    for (int k = lane; k < (r_hi - r_lo + 1) * w; k += 64) {
        const int rr = k / w, cc = k - rr * w;
        strip[k] = imr[(size_t)(r_lo + rr) * LR.istride + cc];
    }
    {   // template patch from the left image (global reads: ws*ws samples once per point)
        const subpix_geo g = make_geo(x - (float)(ws - 1) * 0.5f, bfr, w, ws);
        for (int t = lane; t < ws * ws; t += 64) {
            const int r = t / ws, j = t - r * ws;
            patch[t] = (unsigned char)subpix_px(g, iml + (size_t)rowA[r] * LL.istride, iml + (size_t)rowB[r] * LL.istride, j);
        }
    }
    __syncthreads();
    const float nb = (float)(ws * ws);
    unsigned long long best = ~0ull;   // (float bits of sad / n) << 32 | candidate index; errors are >= 0 so the bits order like the values
    for (int s0 = 0;; s0 += 64) {
        // candidate s = s0 + lane: follow the loop variable of the reference (c = x; c -= 1 / c += 1)
        const int s = s0 + lane;
        float c = x;
        for (int k = 0; k < s; ++k) c = go_left ? c - 1.f : c + 1.f;
        float c_first = x;   // candidate s0 (lane 0) decides whether the wave is done
        for (int k = 0; k < s0; ++k) c_first = go_left ? c_first - 1.f : c_first + 1.f;
        const bool wave_live = go_left ? (c_first >= (float)halfwin) : (c_first < (float)(w - halfwin));
        if (!wave_live) break;
        const bool live = go_left ? (c >= (float)halfwin) : (c < (float)(w - halfwin));
        if (live) {
            const subpix_geo g = make_geo(c - (float)(ws - 1) * 0.5f, bfr, w, ws);
            int sad = 0;
            for (int r = 0; r < ws; ++r) {
                const unsigned char *A = strip + (rowA[r] - r_lo) * w, *B = strip + (rowB[r] - r_lo) * w;
                for (int j = 0; j < ws; ++j) {
                    const int d = (int)patch[r * ws + j] - subpix_px(g, A, B, j);
                    sad += d < 0 ? -d : d;
                }
            }
            float err = (float)(double)sad;
            err /= nb;
            if (err < 255.f) {   // minsad starts at 255: a candidate must beat it
                const unsigned long long key = ((unsigned long long)__float_as_uint(err) << 32) | (unsigned)s;
                if (key < best) best = key;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o);
        if (other < best) best = other;
    }
    if (best != ~0ull) {
        const int s = (int)(best & 0xffffffffull);
        float c = x;
        for (int k = 0; k < s; ++k) c = go_left ? c - 1.f : c + 1.f;
        out_x = c;
        out_e = __uint_as_float((unsigned)(best >> 32));
    } else {
        out_e = 255.f;
    }
    if (lane == 0) { xprior[i] = out_x; if (l1err) l1err[i] = out_e; }
}

struct gate_params {
    double F[9];
    int rectified;
    ov2_cam_model rcam;   // the right camera's lens model (model 0: undistortImagePoint is the identity)
};

// src/map_manager.cpp:583-604: the gate compares UNDISTORTED pixels (runpx = pcalib_rightcam_->undistortImagePoint(r), :586);
// what is stored is the raw right pixel (its row snapped onto lunpx.y for rectified rigs, :592)
__global__ __launch_bounds__(256) void epi_gate_kernel(int n, gate_params G, const float2 *__restrict__ kps,
                                                       const float2 *__restrict__ lunpx, float2 *__restrict__ rxy,
                                                       unsigned char *__restrict__ status)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !status[i]) return;
    const float2 l = lunpx ? lunpx[i] : kps[i];
    float2 r = rxy[i];
    float2 ru;
    ov2_cam_undistort(G.rcam, r.x, r.y, ru.x, ru.y);
    float epi_err;
    if (G.rectified) {
        epi_err = fabsf(l.y - ru.y);
        r.y = l.y;            // :592: the right keypoint is put on the left row before the gate decides
        rxy[i] = r;
    } else {
        const double lv[3] = {(double)l.x, (double)l.y, 1.0}, rv[3] = {(double)ru.x, (double)ru.y, 1.0};
        double Fl[3], Ftr[3];
        for (int k = 0; k < 3; ++k) {
            Fl[k] = G.F[3 * k] * lv[0] + G.F[3 * k + 1] * lv[1] + G.F[3 * k + 2] * lv[2];
            Ftr[k] = G.F[k] * rv[0] + G.F[3 + k] * rv[1] + G.F[6 + k] * rv[2];
        }
        float num = (float)(Ftr[0] * lv[0] + Ftr[1] * lv[1] + Ftr[2] * lv[2]);
        num *= num;
        const float x1 = (float)Ftr[0], x2 = (float)Fl[0], y1 = (float)Ftr[1], y2 = (float)Fl[1];
        const float den = x1 * x1 + y1 * y1 + x2 * x2 + y2 * y2;
        epi_err = sqrtf(num / den);
    }
    if (!((double)epi_err <= 2.)) status[i] = 0;
}

size_t sad_lds_bytes(int ws_max, int w) { return (size_t)2 * ws_max * sizeof(int) + ((ws_max * ws_max + 3) & ~3) + (size_t)(ws_max + 1) * w + 16; }

}  // namespace

extern "C" ov2_status ov2_line_min_sad_dev(ov2_ctx *c, const ov2_pyr *left, const ov2_pyr *right, int level, int nwinsize,
                                           int go_left, int n, const float *d_pts_xy, const int32_t *d_img_idx,
                                           float *d_xprior, float *d_l1err)
{
    if (!c) return OV2_ERR_INVALID;
    if (n == 0) return OV2_OK;
    if (n < 0 || !left || !right || !d_pts_xy || !d_xprior) return ov2_set_err(c, OV2_ERR_INVALID, "null/negative argument");
    const ov2_pyr_view &a = left->buf->view, &b = right->buf->view;
    if (a.nlevels != b.nlevels || a.batch != b.batch || a.lv[0].w != b.lv[0].w || a.lv[0].h != b.lv[0].h)
        return ov2_set_err(c, OV2_ERR_INVALID, "left/right pyramids differ in geometry");
    if (level < 0 || level >= a.nlevels) return ov2_set_err(c, OV2_ERR_INVALID, "level %d outside the pyramid", level);
    if (nwinsize < 1 || nwinsize > 15) return ov2_set_err(c, OV2_ERR_INVALID, "nwinsize=%d unsupported (1..15)", nwinsize);
    // the reference's border arithmetic can GROW the half window near the right / bottom border (:157-162): at most
    // hw -> 2 hw - 1 twice
    const int hw0 = nwinsize / 2, ws_max = 2 * (4 * hw0 + 1) + 1;
    const size_t lds = sad_lds_bytes(ws_max, a.lv[level].w);
    if (lds > 60 * 1024) return ov2_set_err(c, OV2_ERR_UNSUPPORTED, "level %d is too wide for the SAD strip (%zu B of LDS)", level, lds);
    OV2_HIP(c, hipSetDevice(c->device));
    ov2_status s;
    if ((s = ov2_pyr_wait_ready(c, left)) != OV2_OK || (s = ov2_pyr_wait_ready(c, right)) != OV2_OK) return s;
    OV2_LAUNCH(c, K_SAD, sad_kernel, dim3(n), dim3(64), lds, c->stream, a, b, level, nwinsize, go_left ? 1 : 0, n,
               reinterpret_cast<const float2 *>(d_pts_xy), d_img_idx, d_xprior, d_l1err, ws_max);
    OV2_HIP(c, hipGetLastError());
    return OV2_OK;
}

extern "C" ov2_status ov2_line_min_sad(ov2_ctx *c, const ov2_pyr *left, const ov2_pyr *right, int level, int nwinsize,
                                       int go_left, int n, const float *pts_xy, float *xprior, float *l1err)
{
    if (!c) return OV2_ERR_INVALID;
    if (n == 0) return OV2_OK;
    if (n < 0 || !pts_xy || !xprior) return ov2_set_err(c, OV2_ERR_INVALID, "null/negative argument");
    void *hs = nullptr, *ds = nullptr;
    const size_t nb = (size_t)n * 8, need = nb + 2 * (size_t)n * 4;
    ov2_status s = ov2_staging(c, need, &hs, &ds);
    if (s != OV2_OK) return s;
    memcpy(hs, pts_xy, nb);
    OV2_HIP(c, hipMemcpyAsync(ds, hs, nb, hipMemcpyHostToDevice, c->stream));
    float *d_x = (float *)((char *)ds + nb), *d_e = d_x + n;
    s = ov2_line_min_sad_dev(c, left, right, level, nwinsize, go_left, n, (const float *)ds, nullptr, d_x, d_e);
    if (s != OV2_OK) return s;
    OV2_HIP(c, hipMemcpyAsync((char *)hs + nb, d_x, 2 * (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    OV2_HIP(c, hipStreamSynchronize(c->stream));
    memcpy(xprior, (char *)hs + nb, (size_t)n * 4);
    if (l1err) memcpy(l1err, (char *)hs + nb + (size_t)n * 4, (size_t)n * 4);
    return OV2_OK;
}

extern "C" ov2_status ov2_stereo_matching_dev(ov2_ctx *c, const ov2_pyr *left, const ov2_pyr *right, int win, int nlevels_full,
                                              int max_iter, float eps, float err_th, float fb_th, int n, const float *d_kps_xy,
                                              const float *d_prior_xy, const uint8_t *d_has_prior, const int32_t *d_img_idx,
                                              const float *d_lunpx_xy, int rectified, const double *F_rl,
                                              const ov2_cam_model *right_cam, float *d_out_rxy, uint8_t *d_out_status,
                                              uint32_t *d_iters)
{
    if (!c) return OV2_ERR_INVALID;
    if (n == 0) return OV2_OK;
    if (!rectified && !F_rl) return ov2_set_err(c, OV2_ERR_INVALID, "F_rl is required for the Sampson gate");
    ov2_status s = ov2_klt_two_stage_dev(c, left, right, win, nlevels_full, max_iter, eps, err_th, fb_th, n, d_kps_xy, d_prior_xy,
                                         d_has_prior, d_img_idx, d_out_rxy, d_out_status, nullptr, d_iters, 0);
    if (s != OV2_OK) return s;
    gate_params G;
    for (int k = 0; k < 9; ++k) G.F[k] = F_rl ? F_rl[k] : 0.0;
    G.rectified = rectified ? 1 : 0;
    G.rcam = ov2_cam_normalised(right_cam);
    if (G.rcam.model < 0 || G.rcam.model > 2) return ov2_set_err(c, OV2_ERR_INVALID, "unknown lens model %d", G.rcam.model);
    OV2_LAUNCH(c, K_GATE, epi_gate_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, G,
               reinterpret_cast<const float2 *>(d_kps_xy), reinterpret_cast<const float2 *>(d_lunpx_xy),
               reinterpret_cast<float2 *>(d_out_rxy), d_out_status);
    OV2_HIP(c, hipGetLastError());
    return OV2_OK;
}

extern "C" ov2_status ov2_stereo_matching(ov2_ctx *c, const ov2_pyr *left, const ov2_pyr *right, int win, int nlevels_full,
                                          int max_iter, float eps, float err_th, float fb_th, int n, const float *kps_xy,
                                          const float *prior_xy, const uint8_t *has_prior, const float *lunpx_xy, int rectified,
                                          const double *F_rl, const ov2_cam_model *right_cam, float *out_rxy, uint8_t *out_status)
{
    if (!c) return OV2_ERR_INVALID;
    if (n == 0) return OV2_OK;
    if (n < 0 || !kps_xy || !prior_xy || !has_prior || !out_rxy || !out_status)
        return ov2_set_err(c, OV2_ERR_INVALID, "null/negative argument");
    // staging block: kps | prior | lunpx | out (n x 8 each) | has_prior | status (n each)
    const size_t nb = (size_t)n * 8, need = 4 * nb + 2 * (size_t)n + 64;
    void *hs = nullptr, *ds = nullptr;
    ov2_status s = ov2_staging(c, need, &hs, &ds);
    if (s != OV2_OK) return s;
    char *h = (char *)hs, *d = (char *)ds;
    memcpy(h, kps_xy, nb);
    memcpy(h + nb, prior_xy, nb);
    if (lunpx_xy) memcpy(h + 2 * nb, lunpx_xy, nb);
    memcpy(h + 4 * nb, has_prior, (size_t)n);
    OV2_HIP(c, hipMemcpyAsync(d, h, 3 * nb, hipMemcpyHostToDevice, c->stream));
    OV2_HIP(c, hipMemcpyAsync(d + 4 * nb, h + 4 * nb, (size_t)n, hipMemcpyHostToDevice, c->stream));
    s = ov2_stereo_matching_dev(c, left, right, win, nlevels_full, max_iter, eps, err_th, fb_th, n, (const float *)d,
                                (const float *)(d + nb), (const uint8_t *)(d + 4 * nb), nullptr,
                                lunpx_xy ? (const float *)(d + 2 * nb) : nullptr, rectified, F_rl, right_cam, (float *)(d + 3 * nb),
                                (uint8_t *)(d + 4 * nb + n), nullptr);
    if (s != OV2_OK) return s;
    OV2_HIP(c, hipMemcpyAsync(h + 3 * nb, d + 3 * nb, nb, hipMemcpyDeviceToHost, c->stream));
    OV2_HIP(c, hipMemcpyAsync(h + 4 * nb + n, d + 4 * nb + n, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    OV2_HIP(c, hipStreamSynchronize(c->stream));
    memcpy(out_rxy, h + 3 * nb, nb);
    memcpy(out_status, h + 4 * nb + n, (size_t)n);
    return OV2_OK;
}
