/*
 * ov2_oracle_det.c -- CPU restatement of the keyframe-rate detectors (TEST INFRASTRUCTURE ONLY, see ov2_oracle.h).
 * Compile with -ffp-contract=off (every fp32 expression rounds exactly as written; the HIP side matches bit for bit).
 *
 * Reference call sites (/root/reference): FeatureExtractor::detectSingleScale src/feature_extractor.cpp:288-440,
 * FeatureExtractor::detectGridFAST :443-570, cv::cornerSubPix call at :429-437 / :560-566; caller
 * MapManager::extractKeypoints src/map_manager.cpp:286-340.
 * The arithmetic is OpenCV's (GaussianBlur 3x3 fixed point, cornerMinEigenVal(3,3), FAST-9/16 + cornerScore + NMS,
 * cv::circle filled, getRectSubPix bilinear, cornerSubPix) -- not vendored, version unpinned => parity unpinned; the
 * restatement follows the published algorithms and fixes one evaluation order per fp32 expression (stated inline).
 *
 * Two properties of the reference that shape this file:
 *  (1) both detectors mutate ONE shared float mask from inside cv::parallel_for_ (src/feature_extractor.cpp:341,372,
 *      388,499,527): the result depends on the thread schedule.  Canonical order here (and on the GPU): the 2x2
 *      colouring of the cell grid, colour = (r&1)*2 + (c&1) ascending, row-major inside a colour.  Same-colour cells
 *      are >= one cell apart and the mask discs have radius cell/4, so they never interact: this is one valid
 *      schedule of the reference's loop that can also run in parallel.
 *  (2) detectGridFAST hands the CV_32F mask to FastFeatureDetector::detect, whose pixel-mask filter reads it with
 *      at<uchar>(y, x) (features2d KeyPointsFilter::runByPixelsMask): inside the cell ROI byte x of a row is byte
 *      (x & 3) of float (x >> 2); 1.0f = 00 00 80 3F, so a FAST corner survives iff (x & 3) >= 2 and
 *      maskf[y0+y][x0 + (x >> 2)] != 0.  That is what the reference computes, and what is restated here.
 */
#include "ov2_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * (n - 1) - i;
    }
    return i;
}
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int cv_round(float v) { return (int)lrintf(v); }
static inline int cv_floor(float v) { return (int)floorf(v); }

/* cv::circle(img, center, radius, value, FILLED) for LINE_8 / shift 0: drawing.cpp Circle() midpoint spans, clipped */
void ov2o_draw_disc_u8(uint8_t *mask, int w, int h, int cx, int cy, int radius, uint8_t value)
{
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        const int ys[4] = {cy - dy, cy + dy, cy - dx, cy + dx};
        const int xa[4] = {cx - dx, cx - dx, cx - dy, cx - dy};
        const int xb[4] = {cx + dx, cx + dx, cx + dy, cx + dy};
        for (int k = 0; k < 4; ++k) {
            if (ys[k] < 0 || ys[k] >= h) continue;
            const int x0 = xa[k] < 0 ? 0 : xa[k], x1 = xb[k] >= w ? w - 1 : xb[k];
            for (int x = x0; x <= x1; ++x) mask[(size_t)ys[k] * w + x] = value;
        }
        dy++;
        err += plus;
        plus += 2;
        const int m = (err <= 0) - 1;
        err -= minus & m;
        dx += m;
        minus -= m & 2;
    }
}

/* GaussianBlur(im(hroi), 3x3, sigma 0) (8u fixed point: [1 2 1]^2 / 16, round half up, borders read the parent image,
 * REFLECT_101 at the image border) then cornerMinEigenVal(blockSize 3, ksize 3) on the blurred cell (REFLECT_101 at
 * the CELL border).  fp32 evaluation order fixed as written. */
void ov2o_min_eig_cell(const uint8_t *img, int w, int h, int stride, int x0, int y0, int cell, float *hmap)
{
    const int n = cell;
    uint8_t *bl = (uint8_t *)malloc((size_t)n * n);
    float *dx = (float *)malloc(sizeof(float) * n * n), *dy = (float *)malloc(sizeof(float) * n * n);
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
            int s = 0;
            static const int k[3] = {1, 2, 1};
            for (int j = -1; j <= 1; ++j)
                for (int i = -1; i <= 1; ++i)
                    s += k[j + 1] * k[i + 1] * img[(size_t)reflect101(y0 + y + j, h) * stride + reflect101(x0 + x + i, w)];
            bl[y * n + x] = (uint8_t)((s + 8) >> 4);
        }
    /* Sobel 3x3 with scale = 1 / (2^(ksize-1) * blockSize * 255) folded into the smoothing taps (cv::Sobel) */
    const float sc = (float)(1.0 / (4.0 * 3.0 * 255.0)), sc2 = sc * 2.f;
#define B(yy, xx) ((float)bl[reflect101((yy), n) * n + reflect101((xx), n)])
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
            /* Dx: row pass p[x+1]-p[x-1] (exact), column pass 2s*r0 + s*(r-1 + r+1) */
            const float rm = B(y - 1, x + 1) - B(y - 1, x - 1), r0 = B(y, x + 1) - B(y, x - 1), rp = B(y + 1, x + 1) - B(y + 1, x - 1);
            dx[y * n + x] = sc2 * r0 + sc * (rm + rp);
            /* Dy: row pass (p[x-1]+p[x+1])*s + p[x]*2s, column pass r+1 - r-1 */
            const float tm = (B(y - 1, x - 1) + B(y - 1, x + 1)) * sc + B(y - 1, x) * sc2;
            const float tp = (B(y + 1, x - 1) + B(y + 1, x + 1)) * sc + B(y + 1, x) * sc2;
            dy[y * n + x] = tp - tm;
        }
#undef B
    /* cov = (dx^2, dx dy, dy^2) box-summed 3x3 (sums in double like boxFilter's CV_64F accumulator, cast to float) */
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
            double a = 0, b = 0, c = 0;
            for (int j = -1; j <= 1; ++j)
                for (int i = -1; i <= 1; ++i) {
                    const int q = reflect101(y + j, n) * n + reflect101(x + i, n);
                    a += (double)(dx[q] * dx[q]);
                    b += (double)(dx[q] * dy[q]);
                    c += (double)(dy[q] * dy[q]);
                }
            const float fa = (float)a * 0.5f, fb = (float)b, fc = (float)c * 0.5f;
            hmap[y * n + x] = (fa + fc) - sqrtf((fa - fc) * (fa - fc) + fb * fb);
        }
    free(bl); free(dx); free(dy);
}

/* masked arg-max in row-major scan order (cv::minMaxLoc returns the first maximum) */
static void masked_argmax(const float *hmap, const uint8_t *mask, int w, int x0, int y0, int n, float *best, int *bx, int *by)
{
    float m = -FLT_MAX;
    int mx = 0, my = 0;
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
            const float v = mask[(size_t)(y0 + y) * w + x0 + x] ? hmap[y * n + x] : 0.f;   /* hmap.mul(mask) */
            if (v > m) { m = v; mx = x; my = y; }
        }
    *best = m; *bx = mx; *by = my;
}

/* ------------------------------------------------------------------------------------------ */
/* cornerSubPix                                                                                 */

static float rect_subpix(const uint8_t *img, int w, int h, int stride, int ipx, int ipy, float a11, float a12, float a21,
                         float a22, int i, int j)
{   /* getRectSubPix_Cn_<uchar,float,float> with the out-of-image taps clamped (BORDER_REPLICATE) */
    const int x0 = clampi(ipx + j, 0, w - 1), x1 = clampi(ipx + j + 1, 0, w - 1);
    const int y0 = clampi(ipy + i, 0, h - 1), y1 = clampi(ipy + i + 1, 0, h - 1);
    return (float)img[(size_t)y0 * stride + x0] * a11 + (float)img[(size_t)y0 * stride + x1] * a12 +
           (float)img[(size_t)y1 * stride + x0] * a21 + (float)img[(size_t)y1 * stride + x1] * a22;
}

void ov2o_corner_subpix(const uint8_t *img, int w, int h, int stride, int n, float *xy, int hw, int max_iter, double eps)
{
    const int win = 2 * hw + 1, bw = win + 2;
    float mask[15 * 15], buf[17 * 17];
    if (hw > 7) return;
    if (max_iter < 1) max_iter = 1;
    if (max_iter > 100) max_iter = 100;   /* MAX_ITERS clamp of cornerSubPix */
    eps = eps < 0 ? 0 : eps;
    eps *= eps;
    for (int i = 0; i < win; ++i) {
        const float y = (float)(i - hw) / (float)hw;
        const float vy = expf(-y * y);
        for (int j = 0; j < win; ++j) {
            const float x = (float)(j - hw) / (float)hw;
            mask[i * win + j] = (float)(vy * expf(-x * x));
        }
    }
    for (int p = 0; p < n; ++p) {
        const float cTx = xy[2 * p], cTy = xy[2 * p + 1];
        float cIx = cTx, cIy = cTy;
        int iter = 0;
        double err = 0;
        do {
            float cx = cIx - (float)(bw - 1) * 0.5f, cy = cIy - (float)(bw - 1) * 0.5f;
            const int ipx = cv_floor(cx), ipy = cv_floor(cy);
            const float a = cx - (float)ipx, b = cy - (float)ipy;
            const float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b;
            for (int i = 0; i < bw; ++i)
                for (int j = 0; j < bw; ++j) buf[i * bw + j] = rect_subpix(img, w, h, stride, ipx, ipy, a11, a12, a21, a22, i, j);
            /* the five sums over the window: cv::cornerSubPix adds the terms row-major in double; the canonical order
             * here is the one a 16-lane DPP row produces -- leaf l = terms l, l+16, l+32, ... added in that order
             * (term t = i*win + j), then the 16-leaf pairwise tree ((l0+l1)+(l2+l3))+... -- so that the GPU
             * reproduces it bit for bit; the two orders differ at the 1e-16 level only */
            double acc[5][16];
            for (int t = 0; t < 16; ++t) for (int q = 0; q < 5; ++q) acc[q][t] = 0.0;
            for (int i = 0; i < win; ++i) {
                const double py = i - hw;
                for (int j = 0; j < win; ++j) {   /* row-major t ascending == ascending k within every leaf */
                    const float *sp = buf + (i + 1) * bw + (j + 1);
                    const double m = mask[i * win + j];
                    const double tgx = sp[1] - sp[-1];
                    const double tgy = sp[bw] - sp[-bw];
                    const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
                    const double px = j - hw;
                    const int l = (i * win + j) & 15;
                    acc[0][l] += gxx; acc[1][l] += gxy; acc[2][l] += gyy;
                    acc[3][l] += gxx * px + gxy * py;
                    acc[4][l] += gxy * px + gyy * py;
                }
            }
            for (int o = 1; o < 16; o <<= 1)
                for (int q = 0; q < 5; ++q) {
                    double tmp[16];
                    for (int t = 0; t < 16; ++t) tmp[t] = acc[q][t] + acc[q][t ^ o];
                    for (int t = 0; t < 16; ++t) acc[q][t] = tmp[t];
                }
            const double A = acc[0][0], Bm = acc[1][0], C = acc[2][0], bb1 = acc[3][0], bb2 = acc[4][0];
            const double det = A * C - Bm * Bm;
            if (fabs(det) <= DBL_EPSILON * DBL_EPSILON) break;
            const double scale = 1.0 / det;
            const float nx = (float)(cIx + C * scale * bb1 - Bm * scale * bb2);
            const float ny = (float)(cIy - Bm * scale * bb1 + A * scale * bb2);
            err = (double)((nx - cIx) * (nx - cIx) + (ny - cIy) * (ny - cIy));
            cIx = nx; cIy = ny;
            if (cIx < 0 || cIx >= (float)w || cIy < 0 || cIy >= (float)h) break;
        } while (++iter < max_iter && err > eps);
        if (fabsf(cIx - cTx) > (float)hw || fabsf(cIy - cTy) > (float)hw) { cIx = cTx; cIy = cTy; }
        xy[2 * p] = cIx; xy[2 * p + 1] = cIy;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* grid bookkeeping shared by both detectors                                                    */

typedef struct {
    int nwcells, nhcells, nbcells;
    uint8_t *occ;    /* (nhcells+1) x (nwcells+1) */
    uint8_t *mask;   /* w x h, 1 = free */
} grid;

static void grid_init(grid *g, int w, int h, int cell, int n_cur, const float *cur_xy)
{
    g->nhcells = h / cell; g->nwcells = w / cell; g->nbcells = g->nhcells * g->nwcells;
    g->occ = (uint8_t *)calloc((size_t)(g->nhcells + 1) * (g->nwcells + 1), 1);
    g->mask = (uint8_t *)malloc((size_t)w * h);
    memset(g->mask, 1, (size_t)w * h);
    const int r = cell / 4;
    for (int k = 0; k < n_cur; ++k) {
        const float x = cur_xy[2 * k], y = cur_xy[2 * k + 1];
        const int cr = (int)(y / (float)cell), cc = (int)(x / (float)cell);   /* voccupcells[px.y/cell][px.x/cell] */
        if (cr >= 0 && cr <= g->nhcells && cc >= 0 && cc <= g->nwcells) g->occ[cr * (g->nwcells + 1) + cc] = 1;
        ov2o_draw_disc_u8(g->mask, w, h, cv_round(x), cv_round(y), r, 0);
    }
}

static int in_roi(int x, int y, const int roi[4])
{
    return !(x < roi[0] || y < roi[1] || x >= roi[0] + roi[2] || y >= roi[1] + roi[3]);
}

void ov2o_detect_single_scale(const uint8_t *img, int w, int h, int stride, int cell, int n_cur, const float *cur_xy,
                              const int roi[4], double *dmaxquality, int do_subpix, int *n_out, float *out_xy)
{
    grid g;
    grid_init(&g, w, h, cell, n_cur, cur_xy);
    const int nb = g.nbcells, r = cell / 4;
    float *first = (float *)malloc(sizeof(float) * 2 * (nb ? nb : 1)), *second = (float *)malloc(sizeof(float) * 2 * (nb ? nb : 1));
    uint8_t *hf = (uint8_t *)calloc((size_t)(nb ? nb : 1), 1), *hs = (uint8_t *)calloc((size_t)(nb ? nb : 1), 1);
    float *hmap = (float *)malloc(sizeof(float) * cell * cell);
    int nboccup = 0;
    for (int colour = 0; colour < 4; ++colour)
        for (int i = 0; i < nb; ++i) {
            const int rr = i / g.nwcells, cc = i % g.nwcells;
            if (((rr & 1) * 2 + (cc & 1)) != colour) continue;
            if (g.occ[rr * (g.nwcells + 1) + cc]) { nboccup++; continue; }
            const int x = cc * cell, y = rr * cell;
            if (!(x + cell < w - 1 && y + cell < h - 1)) continue;
            ov2o_min_eig_cell(img, w, h, stride, x, y, cell, hmap);
            float best; int bx, by;
            masked_argmax(hmap, g.mask, w, x, y, cell, &best, &bx, &by);
            bx += x; by += y;
            if (!in_roi(bx, by, roi)) continue;
            if ((double)best >= *dmaxquality) {
                first[2 * i] = (float)bx; first[2 * i + 1] = (float)by; hf[i] = 1;
                ov2o_draw_disc_u8(g.mask, w, h, bx, by, r, 0);
            }
            masked_argmax(hmap, g.mask, w, x, y, cell, &best, &bx, &by);
            bx += x; by += y;
            if (!in_roi(bx, by, roi)) continue;
            if ((double)best >= *dmaxquality) {
                second[2 * i] = (float)bx; second[2 * i + 1] = (float)by; hs[i] = 1;
                ov2o_draw_disc_u8(g.mask, w, h, bx, by, r, 0);
            }
        }
    int n = 0;
    for (int i = 0; i < nb; ++i)
        if (hf[i]) { out_xy[2 * n] = first[2 * i]; out_xy[2 * n + 1] = first[2 * i + 1]; ++n; }
    if (n + nboccup < nb) {   /* :397-412 fill up with second candidates */
        const int nbsec = nb - (n + nboccup);
        int k = 0;
        for (int i = 0; i < nb && k < nbsec; ++i)
            if (hs[i]) { out_xy[2 * n] = second[2 * i]; out_xy[2 * n + 1] = second[2 * i + 1]; ++n; ++k; }
    }
    if ((double)n < 0.33 * (double)(nb - nboccup)) *dmaxquality /= 2.;      /* :418-423 */
    else if ((double)n > 0.9 * (double)(nb - nboccup)) *dmaxquality *= 1.5;
    if (n > 0 && do_subpix) ov2o_corner_subpix(img, w, h, stride, n, out_xy, 3, 30, 0.01);
    *n_out = n;
    free(first); free(second); free(hf); free(hs); free(hmap); free(g.occ); free(g.mask);
}

/* ------------------------------------------------------------------------------------------ */
/* FAST-9/16 (features2d fast.cpp FAST_t<16>, fast_score.cpp cornerScore<16>)                    */

static const int fast_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int fast_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

int ov2o_fast_score(const uint8_t *img, int stride, int x, int y, int threshold)
{
    const uint8_t *p = img + (size_t)y * stride + x;
    const int v = p[0];
    int d[25];
    for (int k = 0; k < 25; ++k) d[k] = v - p[fast_dy[k & 15] * stride + fast_dx[k & 15]];
    /* corner test: 9 contiguous pixels darker than v - t or brighter than v + t */
    int is_corner = 0;
    for (int s = 0; s < 16 && !is_corner; ++s) {
        int br = 1, dk = 1;
        for (int k = 0; k < 9; ++k) {
            const int dd = d[(s + k) & 15];
            if (!(dd < -threshold)) br = 0;   /* pixel brighter than v + t  <=>  v - p < -t */
            if (!(dd > threshold)) dk = 0;
        }
        is_corner = br | dk;
    }
    if (!is_corner) return 0;
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        a = a < d[k + 3] ? a : d[k + 3];
        if (a <= a0) continue;
        for (int q = 4; q <= 8; ++q) a = a < d[k + q] ? a : d[k + q];
        int m = a < d[k] ? a : d[k];
        a0 = a0 > m ? a0 : m;
        m = a < d[k + 9] ? a : d[k + 9];
        a0 = a0 > m ? a0 : m;
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        for (int q = 3; q <= 5; ++q) b = b > d[k + q] ? b : d[k + q];
        if (b >= b0) continue;
        for (int q = 6; q <= 8; ++q) b = b > d[k + q] ? b : d[k + q];
        int m = b > d[k] ? b : d[k];
        b0 = b0 < m ? b0 : m;
        m = b > d[k + 9] ? b : d[k + 9];
        b0 = b0 < m ? b0 : m;
    }
    return -b0 - 1;
}

void ov2o_detect_grid_fast(const uint8_t *img, int w, int h, int stride, int cell, int n_cur, const float *cur_xy,
                           const int roi[4], int *nfast_th, int do_subpix, int *n_out, float *out_xy)
{
    (void)roi;   /* detectGridFAST never reads its roi argument */
    grid g;
    grid_init(&g, w, h, cell, n_cur, cur_xy);
    const int nb = g.nbcells, r = cell / 4;
    float *first = (float *)malloc(sizeof(float) * 2 * (nb ? nb : 1));
    uint8_t *hf = (uint8_t *)calloc((size_t)(nb ? nb : 1), 1);
    int *score = (int *)malloc(sizeof(int) * cell * cell);
    int nboccup = 0, nbempty = 0;
    int th = *nfast_th;
    th = th < 0 ? 0 : (th > 255 ? 255 : th);
    for (int colour = 0; colour < 4; ++colour)
        for (int i = 0; i < nb; ++i) {
            const int rr = i / g.nwcells, cc = i % g.nwcells;
            if (((rr & 1) * 2 + (cc & 1)) != colour) continue;
            if (g.occ[rr * (g.nwcells + 1) + cc]) { nboccup++; continue; }
            nbempty++;
            const int x0 = cc * cell, y0 = rr * cell;
            if (!(x0 + cell < w - 1 && y0 + cell < h - 1)) continue;
            /* FAST on the cell ROI: a 3-px border of the ROI is never a corner; pixels outside the ROI are not read */
            memset(score, 0, sizeof(int) * cell * cell);
            for (int y = 3; y < cell - 3; ++y)
                for (int x = 3; x < cell - 3; ++x) score[y * cell + x] = ov2o_fast_score(img, stride, x0 + x, y0 + y, th);
            int best = -1, bx = 0, by = 0;
            for (int y = 3; y < cell - 3; ++y)
                for (int x = 3; x < cell - 3; ++x) {
                    const int s = score[y * cell + x];
                    if (s <= 0) continue;
                    int nms = 1;   /* strictly greater than the 8 neighbours */
                    for (int j = -1; j <= 1 && nms; ++j)
                        for (int ii = -1; ii <= 1; ++ii)
                            if ((j || ii) && !(s > score[(y + j) * cell + x + ii])) { nms = 0; break; }
                    if (!nms) continue;
                    /* the CV_32F mask read as bytes, see the header of this file */
                    if ((x & 3) < 2 || !g.mask[(size_t)(y0 + y) * w + x0 + (x >> 2)]) continue;
                    if (s > best) { best = s; bx = x; by = y; }   /* first maximum in row-major order */
                }
            if (best >= 20) {
                first[2 * i] = (float)(bx + x0); first[2 * i + 1] = (float)(by + y0); hf[i] = 1;
                ov2o_draw_disc_u8(g.mask, w, h, bx + x0, by + y0, r, 0);
            }
        }
    int n = 0;
    for (int i = 0; i < nb; ++i)
        if (hf[i]) { out_xy[2 * n] = first[2 * i]; out_xy[2 * n + 1] = first[2 * i + 1]; ++n; }
    if ((double)n < 0.5 * (double)nbempty && nbempty > 10) *nfast_th = (int)((double)*nfast_th * 0.66);   /* int nfast_th_ *= 0.66 */
    else if (n == nbempty) *nfast_th = (int)((double)*nfast_th * 1.5);
    if (n > 0 && do_subpix) ov2o_corner_subpix(img, w, h, stride, n, out_xy, 3, 30, 0.01);
    *n_out = n;
    free(first); free(hf); free(score); free(g.occ); free(g.mask);
}
