/*
 * ov2_oracle_match.c -- CPU restatement of the keyframe descriptor path (first half).
 * TEST INFRASTRUCTURE ONLY (see ov2_oracle.h).  Follows (reference, /root/reference):
 *   FeatureExtractor::describeBRIEF  src/feature_extractor.cpp:224-285 -> cv::xfeatures2d::BriefDescriptorExtractor
 *       (opencv_contrib xfeatures2d/src/brief.cpp: 32 bytes, PATCH_SIZE 48, KERNEL_SIZE 9, box sums from the integral
 *       image at the ROUNDED keypoint, keypoints closer than 28 px to the border dropped).  opencv_contrib is not
 *       vendored and the 256 test pairs live in its generated_32.i: the pattern is a caller-supplied table here =>
 *       parity unpinned for the table, the arithmetic around it is restated.
 *   Mapper::matchToMap               src/mapper.cpp:576-774 (+ MapPoint::computeMinDescDist src/map_point.cpp:236-252,
 *       Frame::getSurroundingKeypoints(Point2f) src/frame.cpp:624-650) on flat arrays.
 */
#include "ov2_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* smoothedSum of brief.cpp: 9 x 9 box around (img_y + y, img_x + x), from the integral image there; here the exact sum */
static int box9(const uint8_t *img, int stride, int cy, int cx)
{
    int s = 0;
    for (int dy = -4; dy <= 4; ++dy) {
        const uint8_t *r = img + (size_t)(cy + dy) * stride + cx;
        for (int dx = -4; dx <= 4; ++dx) s += r[dx];
    }
    return s;
}

void ov2o_describe_brief(const uint8_t *img, int w, int h, int stride, int n, const float *pts_xy, const int8_t *pattern,
                         uint8_t *desc, uint8_t *valid)
{
    /* KeyPointsFilter::runByImageBorder(keypoints, size, PATCH_SIZE / 2 + KERNEL_SIZE / 2 = 28): Rect_<float>::contains */
    const int border = 28;
    for (int i = 0; i < n; ++i) {
        const float x = pts_xy[2 * i], y = pts_xy[2 * i + 1];
        uint8_t *d = desc + (size_t)i * 32;
        memset(d, 0, 32);
        const int ok = (w > 2 * border && h > 2 * border) && x >= (float)border && x < (float)(w - border) && y >= (float)border &&
                       y < (float)(h - border);
        valid[i] = (uint8_t)ok;
        if (!ok) continue;
        const int px = (int)((double)x + 0.5), py = (int)((double)y + 0.5);   /* (int)(kpt.pt.x + 0.5) */
        for (int k = 0; k < 256; ++k) {
            const int8_t *t = pattern + 4 * k;                                 /* SMOOTHED(y1, x1) < SMOOTHED(y2, x2) */
            const int a = box9(img, stride, py + t[0], px + t[1]), b = box9(img, stride, py + t[2], px + t[3]);
            if (a < b) d[k >> 3] |= (uint8_t)(1u << (7 - (k & 7)));
        }
    }
}

static void pose_rt(const double *T, double R[9], double t[3])
{
    double x = T[3], y = T[4], z = T[5], w = T[6];
    const double n = sqrt(x * x + y * y + z * z + w * w);
    x /= n; y /= n; z /= n; w /= n;
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
    t[0] = T[0]; t[1] = T[1]; t[2] = T[2];
}

/* Frame::projWorldToCam with Tcw = Twc^-1: R^T (p - t) */
static void world_to_cam(const double *Twc, const double *p, double c[3])
{
    double R[9], t[3];
    pose_rt(Twc, R, t);
    const double d[3] = {p[0] - t[0], p[1] - t[1], p[2] - t[2]};
    for (int r = 0; r < 3; ++r) c[r] = R[r] * d[0] + R[3 + r] * d[1] + R[6 + r] * d[2];
}

static int hamming32(const uint8_t *a, const uint8_t *b)
{
    int s = 0;
    for (int k = 0; k < 32; ++k) s += __builtin_popcount((unsigned)(a[k] ^ b[k]));
    return s;
}

void ov2o_match_to_map(const ov2_match_input *in, float fmaxprojerr, float fdistratio, int32_t *match_cand, float *match_dist)
{
    const int nkp = in->n_kp;
    for (int k = 0; k < nkp; ++k) { match_cand[k] = -1; match_dist[k] = 0.f; }
    if (in->n_cand <= 0) return;                                            /* :580-583 */
    const float vfov = (float)(0.5 * in->img_h / in->K[1]), hfov = (float)(0.5 * in->img_w / in->K[0]);   /* :586-587 */
    const float maxradfov = hfov > vfov ? atanf(hfov) : atanf(vfov);
    const float view_th = cosf(maxradfov);
    float dmaxpxdist = fmaxprojerr;
    if (in->nb3dkps < 30) dmaxpxdist *= 2.f;                                /* :600-603 */
    const int nbw = (int)ceilf((float)in->img_w / (float)in->cell), nbh = (int)ceilf((float)in->img_h / (float)in->cell);
    const int ncells = nbw * nbh;
    float *kbest = (float *)malloc(sizeof(float) * (size_t)(nkp + 1));      /* per keypoint: best distance so far (:754-771) */
    for (int k = 0; k < nkp; ++k) kbest[k] = 1024.f;
    for (int c = 0; c < in->n_cand; ++c) {
        const int cd0 = in->cand_desc_ptr[c], cd1 = in->cand_desc_ptr[c + 1];
        if (cd1 == cd0) continue;                                           /* plm->desc_.empty() :622 */
        const double *wpt = in->cand_wpt + 3 * (size_t)c;
        double campt[3];
        world_to_cam(in->Twc, wpt, campt);
        if (campt[2] < 0.1) continue;                                       /* :631 */
        const float view_angle = (float)(campt[2] / sqrt(campt[0] * campt[0] + campt[1] * campt[1] + campt[2] * campt[2]));
        if (fabsf(view_angle) < view_th) continue;                          /* :637 */
        float px, py;                                        /* Frame::projWorldToImageDist (:631) */
        ov2o_cam_project_dist(in->cam, in->K, campt, &px, &py);
        if (!(px >= 0 && py >= 0 && px < (float)in->img_w && py < (float)in->img_h)) continue;   /* isInImage :643 */
        const float mindist = (float)((double)(32.f * fdistratio) * 8.);    /* desc_.cols * fdistratio * 8. */
        int bestid = -1, secid = -1;
        float bestdist = mindist, secdist = mindist;
        const int rkp = (int)floorf(py / (float)in->cell), ckp = (int)floorf(px / (float)in->cell);   /* src/frame.cpp:629-630 */
        for (int r = rkp - 1; r < rkp + 1; ++r)
            for (int cc = ckp - 1; cc < ckp + 1; ++cc) {
                const int idx = r * nbw + cc;
                if (r < 0 || cc < 0 || idx >= ncells) continue;
                for (int g = in->grid_ptr[idx]; g < in->grid_ptr[idx + 1]; ++g) {
                    const int k = in->grid_kp[g];
                    const float dx = px - in->kp_px[2 * k], dy = py - in->kp_px[2 * k + 1];
                    const float pxdist = (float)sqrt((double)dx * dx + (double)dy * dy);   /* cv::norm(Point2f) */
                    if (pxdist > dmaxpxdist) continue;                      /* :669 */
                    const int kd0 = in->kp_desc_ptr[k], kd1 = in->kp_desc_ptr[k + 1];
                    if (kd1 == kd0) continue;                               /* pkplm->desc_.empty() :683 */
                    /* never both observed in one keyframe (:686-695) */
                    int a = in->cand_kf_ptr[c], a1 = in->cand_kf_ptr[c + 1], b = in->kp_kf_ptr[k], b1 = in->kp_kf_ptr[k + 1], shared = 0;
                    while (a < a1 && b < b1) {
                        if (in->cand_kfids[a] == in->kp_kfids[b]) { shared = 1; break; }
                        if (in->cand_kfids[a] < in->kp_kfids[b]) ++a; else ++b;
                    }
                    if (shared) continue;
                    /* mean reprojection distance of the candidate's point in the keyframes that observe the keypoint (:700-719) */
                    float coprojpx = 0.f;
                    size_t nbcokp = 0;
                    for (int e = in->kp_kf_ptr[k]; e < in->kp_kf_ptr[k + 1]; ++e) {
                        const int kfid = in->kp_kfids[e];
                        if (kfid < 0 || kfid >= in->n_kf) continue;
                        double cp[3];
                        world_to_cam(in->kf_Twc + 7 * (size_t)kfid, wpt, cp);
                        float qx, qy;
                        ov2o_cam_project_dist(in->cam, in->K, cp, &qx, &qy);
                        const float ex = in->kp_kf_px[2 * e] - qx, ey = in->kp_kf_px[2 * e + 1] - qy;
                        coprojpx = (float)((double)coprojpx + sqrt((double)ex * ex + (double)ey * ey));
                        nbcokp++;
                    }
                    if (coprojpx / (float)nbcokp > dmaxpxdist) continue;    /* :717 (0 / 0 = NaN: not greater, kept) */
                    float dist = 1000.f;                                    /* MapPoint::computeMinDescDist */
                    for (int i = cd0; i < cd1; ++i)
                        for (int j = kd0; j < kd1; ++j) {
                            const float hd = (float)hamming32(in->cand_descs + 32 * (size_t)i, in->kp_descs + 32 * (size_t)j);
                            if (hd < dist) dist = hd;
                        }
                    if (dist <= bestdist) { secdist = bestdist; secid = bestid; bestdist = dist; bestid = k; }   /* :723-733 */
                    else if (dist <= secdist) { secdist = dist; secid = k; }
                }
            }
        if (bestid != -1 && secid != -1 && 0.9 * (double)secdist < (double)bestdist) bestid = -1;   /* :736-740 */
        if (bestid < 0) continue;
        if (bestdist <= kbest[bestid]) { kbest[bestid] = bestdist; match_cand[bestid] = c; match_dist[bestid] = bestdist; }   /* :754-771 */
    }
    free(kbest);
}
