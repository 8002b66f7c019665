/*
 * ov2_oracle_stereo.c -- CPU restatement of the keyframe-rate stereo matching pieces around fbKltTracking.
 * TEST INFRASTRUCTURE ONLY (see ov2_oracle.h).  Follows (reference, /root/reference):
 *   FeatureTracker::getLineMinSAD            src/feature_tracker.cpp:140-213   (called at src/map_manager.cpp:429)
 *   MultiViewGeometry::computeSampsonDistance src/multi_view_geometry.cpp:798-821 (called at src/map_manager.cpp:595)
 *   MapManager::stereoMatching               src/map_manager.cpp:367-611: the two fbKltTracking calls (:497-580) and the
 *                                            epipolar gate with the rectified row snap (:583-604), on flat arrays
 * cv::getRectSubPix (8U -> 8U) is OpenCV's imgproc/samplers.cpp getRectSubPix_Cn_<uchar, uchar, int, scale_fixpt,
 * cast_8u> + adjustRect, restated from the published source (OpenCV is not vendored, version unpinned => parity
 * unpinned; pinned by tests/test_oracle_stereo.py against an independent numpy restatement).
 */
#include "ov2_oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

static inline int cv_round(float v) { return (int)lrintf(v); }
static inline int cv_floor(float v) { return (int)floorf(v); }

/* cv::getRectSubPix(src 8UC1, Size(ww, wh), center, dst 8UC1): bilinear sample of a ww x wh rectangle centred on
 * `center`, 16-bit fixed-point weights, BORDER_REPLICATE through adjustRect (the columns / rows that fall outside
 * reuse the edge column / row with the vertical weights b1, b2 only -- exactly as the source does). */
void ov2o_get_rect_sub_pix_u8(const uint8_t *src, int w, int h, int stride, int ww, int wh, float cx, float cy, uint8_t *dst)
{
    cx -= (float)(ww - 1) * 0.5f;
    cy -= (float)(wh - 1) * 0.5f;
    const int ipx = cv_floor(cx), ipy = cv_floor(cy);
    const float a = cx - (float)ipx, b = cy - (float)ipy;
    /* scale_fixpt: cvRound(value * (1 << 16)) */
    const int a11 = cv_round((1.f - a) * (1.f - b) * 65536.f), a12 = cv_round(a * (1.f - b) * 65536.f);
    const int a21 = cv_round((1.f - a) * b * 65536.f), a22 = cv_round(a * b * 65536.f);
    const int b1 = cv_round((1.f - b) * 65536.f), b2 = cv_round(b * 65536.f);
#define CAST8(v) ((uint8_t)(((v) + (1 << 15)) >> 16))
    if (0 <= ipx && ipx < w - ww && 0 <= ipy && ipy < h - wh) {
        const uint8_t *s = src + (size_t)ipy * stride + ipx;
        for (int i = 0; i < wh; ++i, s += stride)
            for (int j = 0; j < ww; ++j)
                dst[i * ww + j] = CAST8(s[j] * a11 + s[j + 1] * a12 + s[j + stride] * a21 + s[j + stride + 1] * a22);
        return;
    }
    /* adjustRect */
    int rx, ry, rw, rh;
    ptrdiff_t off = 0;
    if (ipx >= 0) { off += ipx; rx = 0; }
    else { rx = -ipx; if (rx > ww) rx = ww; }
    if (ipx < w - ww) rw = ww;
    else {
        rw = w - ipx - 1;
        if (rw < 0) { off += rw; rw = 0; }
    }
    if (ipy >= 0) { off += (ptrdiff_t)ipy * stride; ry = 0; }
    else ry = -ipy;
    if (ipy < h - wh) rh = wh;
    else {
        rh = h - ipy - 1;
        if (rh < 0) { off += (ptrdiff_t)rh * stride; rh = 0; }
    }
    const uint8_t *s = src + off - rx;
    for (int i = 0; i < wh; ++i) {
        const uint8_t *s2 = s + stride;
        if (i < ry || i >= rh) s2 -= stride;
        int v = s[rx] * b1 + s2[rx] * b2;
        for (int j = 0; j < rx; ++j) dst[i * ww + j] = CAST8(v);
        v = s[rw] * b1 + s2[rw] * b2;
        for (int j = rw; j < ww; ++j) dst[i * ww + j] = CAST8(v);
        for (int j = rx; j < rw; ++j) dst[i * ww + j] = CAST8(s[j] * a11 + s[j + 1] * a12 + s2[j] * a21 + s2[j + 1] * a22);
        if (i < rh) s = s2;
    }
#undef CAST8
}

/* FeatureTracker::getLineMinSAD on two images of the same size (src/feature_tracker.cpp:140-213) */
void ov2o_line_min_sad_img(const uint8_t *iml, const uint8_t *imr, int w, int h, int stride, float x, float y, int nwinsize,
                           int go_left, float *xprior, float *l1err)
{
    *xprior = -1.f;
    if (nwinsize % 2 == 0) return;                         /* :146-149 (l1err untouched) */
    int halfwin = nwinsize / 2;
    /* :155-162 -- `int += float`: the sum is formed in float and truncated */
    if (x - (float)halfwin < 0) halfwin = (int)((float)halfwin + (x - (float)halfwin));
    if (x + (float)halfwin >= (float)w) halfwin = (int)((float)halfwin + (x + (float)halfwin - (float)w - 1.f));
    if (y - (float)halfwin < 0) halfwin = (int)((float)halfwin + (y - (float)halfwin));
    if (y + (float)halfwin >= (float)h) halfwin = (int)((float)halfwin + (y + (float)halfwin - (float)h - 1.f));
    if (halfwin <= 0) return;                              /* :164-166 */
    const int ws = 2 * halfwin + 1, nbwinpx = ws * ws;
    float minsad = 255.f;
    uint8_t *patch = (uint8_t *)malloc((size_t)nbwinpx), *target = (uint8_t *)malloc((size_t)nbwinpx);
    ov2o_get_rect_sub_pix_u8(iml, w, h, stride, ws, ws, x, y, patch);
    float err = *l1err;
    if (go_left) {
        for (float c = x; c >= (float)halfwin; c -= 1.f) {
            ov2o_get_rect_sub_pix_u8(imr, w, h, stride, ws, ws, c, y, target);
            int sad = 0;
            for (int k = 0; k < nbwinpx; ++k) sad += abs((int)patch[k] - (int)target[k]);   /* cv::norm(NORM_L1), exact */
            err = (float)(double)sad;                       /* double -> float l1err */
            err /= (float)nbwinpx;
            if (err < minsad) { minsad = err; *xprior = c; }
        }
    } else {
        for (float c = x; c < (float)(w - halfwin); c += 1.f) {
            ov2o_get_rect_sub_pix_u8(imr, w, h, stride, ws, ws, c, y, target);
            int sad = 0;
            for (int k = 0; k < nbwinpx; ++k) sad += abs((int)patch[k] - (int)target[k]);
            err = (float)(double)sad;
            err /= (float)nbwinpx;
            if (err < minsad) { minsad = err; *xprior = c; }
        }
    }
    (void)err;
    *l1err = minsad;                                        /* :212 */
    free(patch); free(target);
}

/* n points against level `level` of two pyramids (MapManager::stereoMatching passes vleftpyr.at(2 * nklt_pyr_lvl),
 * i.e. the coarsest level image, and kp.px_ * 2^-nklt_pyr_lvl, src/map_manager.cpp:425-431) */
void ov2o_line_min_sad(const ov2o_pyr *left, const ov2o_pyr *right, int level, int nwinsize, int go_left, int n,
                       const float *pts_xy, float *xprior, float *l1err)
{
    const ov2o_level *L = &left->lv[level], *R = &right->lv[level];
    const uint8_t *il = L->img + (size_t)L->pad * L->stride + L->pad, *ir = R->img + (size_t)R->pad * R->stride + R->pad;
    for (int i = 0; i < n; ++i) {
        float e = 0.f;
        ov2o_line_min_sad_img(il, ir, L->w, L->h, L->stride, pts_xy[2 * i], pts_xy[2 * i + 1], nwinsize, go_left, &xprior[i], &e);
        if (l1err) l1err[i] = e;
    }
}

/* MultiViewGeometry::computeSampsonDistance(Frl, Point2f left, Point2f right) src/multi_view_geometry.cpp:798-821:
 * doubles inside the products, every named intermediate a float.  F row-major. */
float ov2o_sampson_distance(const double F[9], float lx, float ly, float rx, float ry)
{
    const double l[3] = {(double)lx, (double)ly, 1.0}, r[3] = {(double)rx, (double)ry, 1.0};
    double Fl[3], Ftr[3];
    for (int i = 0; i < 3; ++i) {
        Fl[i] = F[3 * i] * l[0] + F[3 * i + 1] * l[1] + F[3 * i + 2] * l[2];
        Ftr[i] = F[i] * r[0] + F[3 + i] * r[1] + F[6 + i] * r[2];
    }
    float num = (float)(Ftr[0] * l[0] + Ftr[1] * l[1] + Ftr[2] * l[2]);   /* (r^T F) l: the product is left-associative */
    num *= num;
    const float x1 = (float)Ftr[0], x2 = (float)Fl[0], y1 = (float)Ftr[1], y2 = (float)Fl[1];
    const float den = x1 * x1 + y1 * y1 + x2 * x2 + y2 * y2;
    return sqrtf(num / den);
}

/* ---- lens models (see ov2_oracle.h) ------------------------------------------------------------------------------ */
static double coef(const ov2_cam_model *c, int i) { return (c && i < c->n_coeffs && i < 5) ? c->D[i] : 0.0; }

void ov2o_cam_undistort(const ov2_cam_model *c, float u_, float v_, float *ou, float *ov)
{
    if (!c || c->model == 0 || c->n_coeffs <= 0) { *ou = u_; *ov = v_; return; }
    const double u = u_, v = v_, fx = c->K[0], fy = c->K[1], cx = c->K[2], cy = c->K[3];
    if (c->model == 1) {   /* cv::undistortPoints(src, dst, K, D, noArray(), K): five fixed-point sweeps, no epsilon test */
        const double k1 = coef(c, 0), k2 = coef(c, 1), p1 = coef(c, 2), p2 = coef(c, 3), k3 = coef(c, 4);
        double x = (u - cx) * (1. / fx), y = (v - cy) * (1. / fy);
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; ++j) {
            const double r2 = x * x + y * y;
            const double icdist = 1. / (1. + ((k3 * r2 + k2) * r2 + k1) * r2);
            if (icdist < 0) { x = (u - cx) * (1. / fx); y = (v - cy) * (1. / fy); break; }
            const double dX = 2. * p1 * x * y + p2 * (r2 + 2. * x * x), dY = p1 * (r2 + 2. * y * y) + 2. * p2 * x * y;
            x = (x0 - dX) * icdist;
            y = (y0 - dY) * icdist;
        }
        *ou = (float)(fx * x + cx); *ov = (float)(fy * y + cy);
        return;
    }
    /* cv::fisheye::undistortPoints: Newton on theta, at most ten steps */
    const double pwx = (u - cx) / fx, pwy = (v - cy) / fy;
    double theta_d = sqrt(pwx * pwx + pwy * pwy);
    const double hp = 1.5707963267948966;
    theta_d = theta_d < -hp ? -hp : (theta_d > hp ? hp : theta_d);
    double scale = 1.0;
    if (theta_d > 1e-8) {
        double theta = theta_d;
        for (int j = 0; j < 10; ++j) {
            const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t6 * t2;
            const double k0t2 = coef(c, 0) * t2, k1t4 = coef(c, 1) * t4, k2t6 = coef(c, 2) * t6, k3t8 = coef(c, 3) * t8;
            const double fix = (theta * (1 + k0t2 + k1t4 + k2t6 + k3t8) - theta_d) / (1 + 3 * k0t2 + 5 * k1t4 + 7 * k2t6 + 9 * k3t8);
            theta -= fix;
            if (fabs(fix) < 1e-8) break;
        }
        scale = tan(theta) / theta_d;
    }
    *ou = (float)(fx * (pwx * scale) + cx); *ov = (float)(fy * (pwy * scale) + cy);
}

void ov2o_cam_project_dist(const ov2_cam_model *c, const double K[4], const double pc[3], float *px, float *py)
{
    const double invz = 1. / pc[2], x = pc[0] * invz, y = pc[1] * invz, fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    if (!c || c->model == 0 || c->n_coeffs <= 0) { *px = (float)(fx * x + cx); *py = (float)(fy * y + cy); return; }
    const double xf = (double)(float)x, yf = (double)(float)y;   /* the reference hands OpenCV a Point3f / Point2f */
    if (c->model == 1) {
        const double k1 = coef(c, 0), k2 = coef(c, 1), p1 = coef(c, 2), p2 = coef(c, 3), k3 = coef(c, 4);
        const double r2 = xf * xf + yf * yf, r4 = r2 * r2, r6 = r4 * r2;
        const double a1 = 2 * xf * yf, a2 = r2 + 2 * xf * xf, a3 = r2 + 2 * yf * yf;
        const double cdist = 1 + k1 * r2 + k2 * r4 + k3 * r6;
        const double xd = xf * cdist + p1 * a1 + p2 * a2, yd = yf * cdist + p1 * a3 + p2 * a1;
        *px = (float)(xd * fx + cx); *py = (float)(yd * fy + cy);
        return;
    }
    const double r = sqrt(xf * xf + yf * yf), theta = atan(r);
    const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
    const double theta_d = theta * (1 + coef(c, 0) * t2 + coef(c, 1) * t4 + coef(c, 2) * t6 + coef(c, 3) * t8);
    const double inv_r = r > 1e-8 ? 1.0 / r : 1.0, cdist = r > 1e-8 ? theta_d * inv_r : 1.0;
    *px = (float)(fx * (xf * cdist) + cx); *py = (float)(fy * (yf * cdist) + cy);
}

/* The tracking + gating part of MapManager::stereoMatching on flat arrays (src/map_manager.cpp:493-604), for a camera
 * pair without distortion (Dcv_.empty(): undistortImagePoint is the identity, src/camera_calibration.cpp:317-319):
 *   has_prior[i] != 0 : keypoint i goes through the 2-level call with prior_xy[i] (v3dkps / v3dpriors, :497-541);
 *                       failures are re-queued on the full pyramid with the UPDATED prior (:533-537)
 *   otherwise         : full pyramid from prior_xy[i] (vkps / vpriors, :544-580: prior = kp.px_ or the SAD prior)
 *   gate (:583-604)   : rectified: |lunpx.y - r.y| <= 2, and the right point is snapped to the left row BEFORE it is
 *                       stored; else Sampson distance with F_rl.  out_status[i] = 1 iff updateKeypointStereo is reached,
 *                       out_rxy[i] = the stored right pixel (the snapped one when rectified) or the last forward result. */
void ov2o_stereo_matching(const ov2o_pyr *left, const ov2o_pyr *right, int win, int nlevels_full, float err_th, float fb_th,
                          int max_iter, float eps, int n, const float *kps_xy, const float *prior_xy,
                          const uint8_t *has_prior, const float *lunpx_xy /* NULL = kps_xy */, int rectified,
                          const double F_rl[9], const ov2_cam_model *right_cam, float *out_rxy, uint8_t *out_status)
{
    int *ida = (int *)malloc((size_t)(n + 1) * sizeof(int)), *idb = (int *)malloc((size_t)(n + 1) * sizeof(int));
    float *ka = (float *)malloc((size_t)(n + 1) * 8), *pa = (float *)malloc((size_t)(n + 1) * 8);
    float *kb = (float *)malloc((size_t)(n + 1) * 8), *pb = (float *)malloc((size_t)(n + 1) * 8);
    uint8_t *st = (uint8_t *)malloc((size_t)n + 1), *trk = (uint8_t *)calloc((size_t)n + 1, 1);
    int na = 0, nb = 0;
    for (int i = 0; i < n; ++i) {
        out_status[i] = 0; out_rxy[2 * i] = prior_xy[2 * i]; out_rxy[2 * i + 1] = prior_xy[2 * i + 1];
        if (has_prior[i]) { memcpy(ka + 2 * na, kps_xy + 2 * i, 8); memcpy(pa + 2 * na, prior_xy + 2 * i, 8); ida[na++] = i; }
        else { memcpy(kb + 2 * nb, kps_xy + 2 * i, 8); memcpy(pb + 2 * nb, prior_xy + 2 * i, 8); idb[nb++] = i; }
    }
    if (na > 0) {                                            /* :497-541 */
        ov2o_fb_klt_tracking(left, right, win, 1, err_th, fb_th, max_iter, eps, na, ka, pa, st, NULL);
        for (int k = 0; k < na; ++k) {
            const int i = ida[k];
            out_rxy[2 * i] = pa[2 * k]; out_rxy[2 * i + 1] = pa[2 * k + 1];
            if (st[k]) trk[i] = 1;
            else { memcpy(kb + 2 * nb, ka + 2 * k, 8); memcpy(pb + 2 * nb, pa + 2 * k, 8); idb[nb++] = i; }
        }
    }
    if (nb > 0) {                                            /* :544-580 */
        ov2o_fb_klt_tracking(left, right, win, nlevels_full, err_th, fb_th, max_iter, eps, nb, kb, pb, st, NULL);
        for (int k = 0; k < nb; ++k) {
            const int i = idb[k];
            out_rxy[2 * i] = pb[2 * k]; out_rxy[2 * i + 1] = pb[2 * k + 1];
            trk[i] = st[k];
        }
    }
    for (int i = 0; i < n; ++i) {                            /* :583-604 */
        if (!trk[i]) continue;
        const float lx = lunpx_xy ? lunpx_xy[2 * i] : kps_xy[2 * i], ly = lunpx_xy ? lunpx_xy[2 * i + 1] : kps_xy[2 * i + 1];
        float rx, ry;                                        /* runpx = pcalib_rightcam_->undistortImagePoint(r) (:586) */
        ov2o_cam_undistort(right_cam, out_rxy[2 * i], out_rxy[2 * i + 1], &rx, &ry);
        float epi_err;
        if (rectified) {
            epi_err = fabsf(ly - ry);
            out_rxy[2 * i + 1] = ly;                         /* :592: snapped whether or not the gate passes */
        } else {
            epi_err = ov2o_sampson_distance(F_rl, lx, ly, rx, ry);
        }
        if ((double)epi_err <= 2.) out_status[i] = 1;        /* `epi_err <= 2.`: float against a double literal */
    }
    free(ida); free(idb); free(ka); free(pa); free(kb); free(pb); free(st); free(trk);
}
