/*
 * ov2_oracle_fe.c -- CPU restatement of the per-frame front-end arithmetic (TEST INFRASTRUCTURE ONLY,
 * see ov2_oracle.h).  Compile with -ffp-contract=off: every float expression below must round exactly
 * as written so the HIP kernels (built with the same flag) agree bit for bit.
 *
 * What is restated (reference call sites in /root/reference):
 *   ov2o_pyramid_build   <- cv::buildOpticalFlowPyramid   src/visual_front_end.cpp:1172, src/mapper.cpp:81
 *   ov2o_clahe           <- cv::CLAHE::apply              src/visual_front_end.cpp:1159, src/mapper.cpp:76
 *   ov2o_calc_optical_flow_pyr_lk <- cv::calcOpticalFlowPyrLK  src/feature_tracker.cpp:66,113
 *   ov2o_fb_klt_tracking <- FeatureTracker::fbKltTracking src/feature_tracker.cpp:35-137
 *   ov2o_klt_tracking_frame <- VisualFrontEnd::kltTracking src/visual_front_end.cpp:132-275
 * OpenCV itself is not in the container (parity unpinned, see header): the arithmetic follows OpenCV's
 * published lkpyramid.cpp / pyramids.cpp / clahe.cpp semantics as written down in SURVEY.md Appendix A.
 * Pinned instead by independent numpy restatements written from that specification (tests/test_oracle_fe.py):
 * pyrDown / Scharr / borders, the whole of CLAHE (byte-equal), one LKTrackerInvoker level and the coarse-to-fine
 * chaining (bit-equal positions, status, err).
 *
 * One deliberate choice: the LK sums (A11,A12,A22,b1,b2) are accumulated EXACTLY in int64 and
 * converted to float once.  OpenCV does that on ARM (acctype=int64) and uses order-dependent float
 * lanes on x86; the exact sum is the order-independent member of that family and is what lets a
 * 64-lane wave reduction agree bit for bit with this scalar loop.
 */
#define _GNU_SOURCE
#include "ov2_oracle.h"

#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <float.h>
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

static inline int cv_round(float v) { return (int)lrintf(v); }  /* round-half-even */
static inline int cv_floor(float v) { return (int)floorf(v); }
static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* ------------------------------------------------------------------------------------------ */
/* worker pool (OpenCV's parallel_for_): ov2o_set_num_threads(n > 1) keeps n - 1 pthreads, created once and pinned
 * one per allowed CPU (threads created per call start on the caller's core and are only migrated by the next
 * load-balancing tick, i.e. after a millisecond-scale call has already finished); the caller works too.  Used by the
 * loops OpenCV parallelises on this path: CLAHE tiles / rows, pyrDown / Scharr rows, the points of one LK call.  Every
 * index is computed independently of the others, so results do not depend on n.  For the N-thread CPU baseline of
 * bench.py; default 1 thread. */
typedef void (*ov2o_range_fn)(void *ctx, int i0, int i1);
typedef struct { ov2o_range_fn fn; void *ctx; int i0, i1; } pool_span;

#define POOL_MAX_THREADS 1024
static struct {
    pthread_mutex_t mu; pthread_cond_t cv_go, cv_done;
    pthread_t th[POOL_MAX_THREADS];
    int n_workers;            /* live worker threads (excluding the caller) */
    int want;                 /* threads requested by ov2o_set_num_threads */
    int n_spans, next_span, pending, stop;
    pool_span *spans;
} g_pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, {0}, 0, 1, 0, 0, 0, 0, NULL};

static void *pool_worker(void *arg)
{
    (void)arg;
    pthread_mutex_lock(&g_pool.mu);
    for (;;) {
        while (!g_pool.stop && g_pool.next_span >= g_pool.n_spans) pthread_cond_wait(&g_pool.cv_go, &g_pool.mu);
        if (g_pool.stop) break;
        const pool_span sp = g_pool.spans[g_pool.next_span++];
        pthread_mutex_unlock(&g_pool.mu);
        sp.fn(sp.ctx, sp.i0, sp.i1);
        pthread_mutex_lock(&g_pool.mu);
        if (--g_pool.pending == 0) pthread_cond_signal(&g_pool.cv_done);
    }
    pthread_mutex_unlock(&g_pool.mu);
    return NULL;
}

void ov2o_set_num_threads(int n)
{
    if (n < 1) n = 1;
    if (n > POOL_MAX_THREADS) n = POOL_MAX_THREADS;
    pthread_mutex_lock(&g_pool.mu);
    g_pool.want = n;
    cpu_set_t allowed;
    const int have_aff = sched_getaffinity(0, sizeof(allowed), &allowed) == 0;
    int cpu = -1;
    for (int k = 0; k < g_pool.n_workers && have_aff; ++k)      /* CPUs already handed out */
        do { cpu = (cpu + 1) % CPU_SETSIZE; } while (!CPU_ISSET(cpu, &allowed));
    while (g_pool.n_workers < n - 1) {
        if (pthread_create(&g_pool.th[g_pool.n_workers], NULL, pool_worker, NULL) != 0) break;
        if (have_aff && CPU_COUNT(&allowed) > 0) {   /* worker k on the (k+1)-th allowed CPU, wrapping */
            do { cpu = (cpu + 1) % CPU_SETSIZE; } while (!CPU_ISSET(cpu, &allowed));
            cpu_set_t one; CPU_ZERO(&one); CPU_SET(cpu, &one);
            (void)pthread_setaffinity_np(g_pool.th[g_pool.n_workers], sizeof(one), &one);
        }
        ++g_pool.n_workers;
    }
    pthread_mutex_unlock(&g_pool.mu);
}
int ov2o_get_num_threads(void) { return g_pool.want; }

/* fn(ctx, i0, i1) over [0, n) in spans of at least `grain` indices */
static void ov2o_parallel_for(int n, int grain, ov2o_range_fn fn, void *ctx)
{
    if (n <= 0) return;
    int nt = g_pool.want < g_pool.n_workers + 1 ? g_pool.want : g_pool.n_workers + 1;
    if (grain < 1) grain = 1;
    if (nt > (n + grain - 1) / grain) nt = (n + grain - 1) / grain;
    if (nt <= 1) { fn(ctx, 0, n); return; }
    int ns = nt * 4;          /* a few spans per thread: indices differ in cost */
    if (ns > (n + grain - 1) / grain) ns = (n + grain - 1) / grain;
    pool_span *sp = (pool_span *)malloc((size_t)ns * sizeof(pool_span));
    if (!sp) { fn(ctx, 0, n); return; }
    for (int t = 0; t < ns; ++t) {
        sp[t].fn = fn; sp[t].ctx = ctx;
        sp[t].i0 = (int)((long long)n * t / ns); sp[t].i1 = (int)((long long)n * (t + 1) / ns);
    }
    pthread_mutex_lock(&g_pool.mu);
    if (g_pool.spans) {       /* a parallel region is already running (nested or concurrent caller): stay serial */
        pthread_mutex_unlock(&g_pool.mu);
        free(sp);
        fn(ctx, 0, n);
        return;
    }
    g_pool.spans = sp; g_pool.n_spans = ns; g_pool.next_span = 0; g_pool.pending = ns;
    pthread_cond_broadcast(&g_pool.cv_go);
    while (g_pool.next_span < g_pool.n_spans) {      /* the caller works too */
        const pool_span me = sp[g_pool.next_span++];
        pthread_mutex_unlock(&g_pool.mu);
        me.fn(me.ctx, me.i0, me.i1);
        pthread_mutex_lock(&g_pool.mu);
        --g_pool.pending;
    }
    while (g_pool.pending > 0) pthread_cond_wait(&g_pool.cv_done, &g_pool.mu);
    g_pool.spans = NULL; g_pool.n_spans = 0; g_pool.next_span = 0;
    pthread_mutex_unlock(&g_pool.mu);
    free(sp);
}

/* ------------------------------------------------------------------------------------------ */
/* pyramid                                                                                      */

static void fill_border_reflect(ov2o_level *L)
{
    const int p = L->pad, w = L->w, h = L->h, s = L->stride;
    for (int y = -p; y < h + p; ++y) {
        const int sy = reflect101(y, h);
        uint8_t *row = L->img + (size_t)(y + p) * s + p;
        const uint8_t *srow = L->img + (size_t)(sy + p) * s + p;
        for (int x = -p; x < w + p; ++x) {
            if (y >= 0 && y < h && x >= 0 && x < w) continue;
            row[x] = srow[reflect101(x, w)];
        }
    }
}

/* pyrDown: [1 4 6 4 1]x[1 4 6 4 1], (sum+128)>>8, REFLECT_101, dst=((w+1)/2,(h+1)/2). */
typedef struct { const ov2o_level *S; ov2o_level *D; } pyr_down_job;
static void pyr_down_rows(void *ctx, int y0, int y1)
{
    static const int k[5] = {1, 4, 6, 4, 1};
    const ov2o_level *S = ((pyr_down_job *)ctx)->S;
    ov2o_level *D = ((pyr_down_job *)ctx)->D;
    for (int y = y0; y < y1; ++y) {
        for (int x = 0; x < D->w; ++x) {
            int acc = 0;
            for (int j = 0; j < 5; ++j) {
                const int sy = reflect101(2 * y - 2 + j, S->h);
                const uint8_t *srow = S->img + (size_t)(sy + S->pad) * S->stride + S->pad;
                int racc = 0;
                for (int i = 0; i < 5; ++i) racc += k[i] * srow[reflect101(2 * x - 2 + i, S->w)];
                acc += k[j] * racc;
            }
            D->img[(size_t)(y + D->pad) * D->stride + D->pad + x] = (uint8_t)((acc + 128) >> 8);
        }
    }
}
static void pyr_down(const ov2o_level *S, ov2o_level *D)
{
    pyr_down_job j = {S, D};
    ov2o_parallel_for(D->h, 8, pyr_down_rows, &j);
}

/* calcSharrDeriv: Ix = t0[x+1]-t0[x-1], t0 = 3(r-1 + r+1) + 10 r0 ; Iy = 3(t1[x-1]+t1[x+1]) + 10 t1[x],
 * t1 = r+1 - r-1 ; rows/cols REFLECT_101 ; padding of the gradient plane stays zero. */
static void scharr_rows(void *ctx, int y0, int y1)
{
    ov2o_level *L = (ov2o_level *)ctx;
    const int p = L->pad, w = L->w, h = L->h, s = L->stride;
    for (int y = y0; y < y1; ++y) {
        const uint8_t *r0 = L->img + (size_t)(reflect101(y - 1, h) + p) * s + p;
        const uint8_t *r1 = L->img + (size_t)(y + p) * s + p;
        const uint8_t *r2 = L->img + (size_t)(reflect101(y + 1, h) + p) * s + p;
        int16_t *g = L->grad + ((size_t)(y + p) * s + p) * 2;
        for (int x = 0; x < w; ++x) {
            const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
            const int t0m = (r0[xm] + r2[xm]) * 3 + r1[xm] * 10;
            const int t0p = (r0[xp] + r2[xp]) * 3 + r1[xp] * 10;
            const int t1m = r2[xm] - r0[xm];
            const int t1c = r2[x] - r0[x];
            const int t1p = r2[xp] - r0[xp];
            g[2 * x + 0] = (int16_t)(t0p - t0m);
            g[2 * x + 1] = (int16_t)((t1p + t1m) * 3 + t1c * 10);
        }
    }
}
static void scharr(ov2o_level *L) { ov2o_parallel_for(L->h, 8, scharr_rows, L); }

static int level_alloc(ov2o_level *L, int w, int h, int pad)
{
    L->w = w; L->h = h; L->pad = pad; L->stride = w + 2 * pad;
    const size_t n = (size_t)(h + 2 * pad) * L->stride;
    L->img = (uint8_t *)calloc(n, 1);
    L->grad = (int16_t *)calloc(n * 2, sizeof(int16_t));
    return L->img && L->grad;
}

ov2o_pyr *ov2o_pyramid_build(const uint8_t *img, int w, int h, int stride, int win, int max_level)
{
    ov2o_pyr *P = (ov2o_pyr *)calloc(1, sizeof(ov2o_pyr));
    if (!P) return NULL;
    if (max_level > 7) max_level = 7;
    int cw = w, ch = h;
    for (int l = 0; l <= max_level; ++l) {
        ov2o_level *L = &P->lv[l];
        if (!level_alloc(L, cw, ch, win)) { ov2o_pyramid_free(P); return NULL; }
        if (l == 0) {
            for (int y = 0; y < h; ++y)
                memcpy(L->img + (size_t)(y + win) * L->stride + win, img + (size_t)y * stride, (size_t)w);
        } else {
            pyr_down(&P->lv[l - 1], L);
        }
        fill_border_reflect(L);
        scharr(L);
        P->nlevels = l + 1;
        /* early stop exactly as buildOpticalFlowPyramid: next level would be <= win */
        cw = (cw + 1) / 2; ch = (ch + 1) / 2;
        if (cw <= win || ch <= win) break;
    }
    return P;
}

void ov2o_pyramid_free(ov2o_pyr *p)
{
    if (!p) return;
    for (int l = 0; l < 8; ++l) { free(p->lv[l].img); free(p->lv[l].grad); }
    free(p);
}

int ov2o_pyr_nlevels(const ov2o_pyr *p) { return p->nlevels; }
void ov2o_pyr_level_info(const ov2o_pyr *p, int l, int *w, int *h, int *pad, int *stride)
{
    *w = p->lv[l].w; *h = p->lv[l].h; *pad = p->lv[l].pad; *stride = p->lv[l].stride;
}
const uint8_t *ov2o_pyr_image(const ov2o_pyr *p, int l) { return p->lv[l].img; }
const int16_t *ov2o_pyr_grad(const ov2o_pyr *p, int l) { return p->lv[l].grad; }

/* ------------------------------------------------------------------------------------------ */
/* CLAHE (8-bit, one channel)                                                                   */

typedef struct {
    const uint8_t *src; uint8_t *dst, *lut;
    int w, h, stride, dst_stride, tiles_x, tiles_y, tw, th, clip_limit;
    float lut_scale;
} clahe_job;

/* CLAHE_CalcLut_Body: clipped histogram + LUT of tiles [t0, t1) (tile index = ty * tiles_x + tx) */
static void clahe_tiles(void *ctx, int t0, int t1)
{
    const clahe_job *J = (const clahe_job *)ctx;
    const int hist_size = 256, w = J->w, h = J->h, tw = J->tw, th = J->th;
    for (int t = t0; t < t1; ++t) {
        const int ty = t / J->tiles_x, tx = t - ty * J->tiles_x;
        int hist[256];
        memset(hist, 0, sizeof(hist));
        for (int y = ty * th; y < (ty + 1) * th; ++y) {
            /* copyMakeBorder(src, ext, 0, eh-h, 0, ew-w, REFLECT_101): ext(y,x)=src(refl(y),refl(x)) */
            const uint8_t *row = J->src + (size_t)reflect101(y, h) * J->stride;
            for (int x = tx * tw; x < (tx + 1) * tw; ++x) hist[row[reflect101(x, w)]]++;
        }
        if (J->clip_limit > 0) {
            int clipped = 0;
            for (int i = 0; i < hist_size; ++i)
                if (hist[i] > J->clip_limit) { clipped += hist[i] - J->clip_limit; hist[i] = J->clip_limit; }
            const int batch = clipped / hist_size;
            int residual = clipped - batch * hist_size;
            for (int i = 0; i < hist_size; ++i) hist[i] += batch;
            if (residual != 0) {
                int step = hist_size / residual;
                if (step < 1) step = 1;
                for (int i = 0; i < hist_size && residual > 0; i += step, residual--) hist[i]++;
            }
        }
        uint8_t *tl = J->lut + (size_t)t * hist_size;
        int sum = 0;
        for (int i = 0; i < hist_size; ++i) {
            sum += hist[i];
            tl[i] = sat_u8(cv_round((float)sum * J->lut_scale));
        }
    }
}

/* CLAHE_Interpolation_Body: rows [y0, y1) */
static void clahe_rows(void *ctx, int y0, int y1)
{
    const clahe_job *J = (const clahe_job *)ctx;
    const int hist_size = 256, w = J->w, tiles_x = J->tiles_x, tiles_y = J->tiles_y;
    const float inv_tw = 1.0f / (float)J->tw;
    const float inv_th = 1.0f / (float)J->th;
    for (int y = y0; y < y1; ++y) {
        const float tyf = (float)y * inv_th - 0.5f;
        int ty1 = cv_floor(tyf);
        int ty2 = ty1 + 1;
        const float ya = tyf - (float)ty1;
        const float ya1 = 1.0f - ya;
        if (ty1 < 0) ty1 = 0;
        if (ty2 > tiles_y - 1) ty2 = tiles_y - 1;
        const uint8_t *p1 = J->lut + (size_t)ty1 * tiles_x * hist_size;
        const uint8_t *p2 = J->lut + (size_t)ty2 * tiles_x * hist_size;
        for (int x = 0; x < w; ++x) {
            const float txf = (float)x * inv_tw - 0.5f;
            int tx1 = cv_floor(txf);
            int tx2 = tx1 + 1;
            const float xa = txf - (float)tx1;
            const float xa1 = 1.0f - xa;
            if (tx1 < 0) tx1 = 0;
            if (tx2 > tiles_x - 1) tx2 = tiles_x - 1;
            const int v = J->src[(size_t)y * J->stride + x];
            const int i1 = tx1 * hist_size + v, i2 = tx2 * hist_size + v;
            const float res = ((float)p1[i1] * xa1 + (float)p1[i2] * xa) * ya1 +
                              ((float)p2[i1] * xa1 + (float)p2[i2] * xa) * ya;
            J->dst[(size_t)y * J->dst_stride + x] = sat_u8(cv_round(res));
        }
    }
}

void ov2o_clahe(const uint8_t *src, int w, int h, int stride, float clipf, int tiles_x, int tiles_y,
                uint8_t *dst, int dst_stride)
{
    const int hist_size = 256;
    /* tile geometry: image extended (REFLECT_101) on the right/bottom when not divisible */
    int ew = w, eh = h;
    if (w % tiles_x != 0 || h % tiles_y != 0) {
        ew = w + (tiles_x - (w % tiles_x));
        eh = h + (tiles_y - (h % tiles_y));
    }
    const int tw = ew / tiles_x, th = eh / tiles_y;
    const int tile_total = tw * th;
    const double clip = (double)clipf;
    int clip_limit = 0;
    if (clip > 0.0) {
        clip_limit = (int)(clip * tile_total / hist_size);
        if (clip_limit < 1) clip_limit = 1;
    }
    uint8_t *lut = (uint8_t *)malloc((size_t)tiles_x * tiles_y * hist_size);
    clahe_job J = {src, dst, lut, w, h, stride, dst_stride, tiles_x, tiles_y, tw, th, clip_limit,
                   (float)(hist_size - 1) / (float)tile_total};
    ov2o_parallel_for(tiles_x * tiles_y, 1, clahe_tiles, &J);
    ov2o_parallel_for(h, 8, clahe_rows, &J);
    free(lut);
}

/* ------------------------------------------------------------------------------------------ */
/* pyramidal Lucas-Kanade                                                                       */

#define W_BITS 14
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

static inline void lk_weights(float a, float b, int *w00, int *w01, int *w10, int *w11)
{
    *w00 = cv_round((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
    *w01 = cv_round(a * (1.f - b) * (float)(1 << W_BITS));
    *w10 = cv_round((1.f - a) * b * (float)(1 << W_BITS));
    *w11 = (1 << W_BITS) - *w00 - *w01 - *w10;
}

/* one LKTrackerInvoker pass (one level) for one point; returns executed iterations */
static int lk_level(const ov2o_level *I, const ov2o_level *J, int level, int max_level,
                    const float *prev_pt_in, float *next_pt /* in/out, in level-0.. see caller */,
                    uint8_t *status, float *err, int win, int max_iter, double eps2, float min_eig_thr)
{
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    const float half = (float)(win - 1) * 0.5f;
    const float lscale = (float)(1. / (double)(1 << level));
    float px = prev_pt_in[0] * lscale, py = prev_pt_in[1] * lscale;
    float nx, ny;
    if (level == max_level) { nx = next_pt[0] * lscale; ny = next_pt[1] * lscale; }
    else { nx = next_pt[0] * 2.f; ny = next_pt[1] * 2.f; }
    next_pt[0] = nx; next_pt[1] = ny;

    px -= half; py -= half;
    const int ipx = cv_floor(px), ipy = cv_floor(py);
    if (ipx < -win || ipx >= I->w || ipy < -win || ipy >= I->h) {
        if (level == 0) { *status = 0; *err = 0.f; }
        return 0;
    }
    int w00, w01, w10, w11;
    lk_weights(px - (float)ipx, py - (float)ipy, &w00, &w01, &w10, &w11);

    int16_t Iw[32 * 32], Ixw[32 * 32], Iyw[32 * 32];
    int64_t sA11 = 0, sA12 = 0, sA22 = 0;
    const int sI = I->stride, p = I->pad;
    for (int y = 0; y < win; ++y) {
        const uint8_t *src = I->img + (size_t)(y + ipy + p) * sI + (ipx + p);
        const int16_t *ds = I->grad + ((size_t)(y + ipy + p) * sI + (ipx + p)) * 2;
        for (int x = 0; x < win; ++x) {
            const int iv = DESCALE(src[x] * w00 + src[x + 1] * w01 + src[x + sI] * w10 + src[x + sI + 1] * w11,
                                   W_BITS - 5);
            const int ix = DESCALE(ds[2 * x] * w00 + ds[2 * x + 2] * w01 + ds[2 * (x + sI)] * w10 +
                                   ds[2 * (x + sI) + 2] * w11, W_BITS);
            const int iy = DESCALE(ds[2 * x + 1] * w00 + ds[2 * x + 3] * w01 + ds[2 * (x + sI) + 1] * w10 +
                                   ds[2 * (x + sI) + 3] * w11, W_BITS);
            Iw[y * win + x] = (int16_t)iv; Ixw[y * win + x] = (int16_t)ix; Iyw[y * win + x] = (int16_t)iy;
            sA11 += (int64_t)ix * ix; sA12 += (int64_t)ix * iy; sA22 += (int64_t)iy * iy;
        }
    }
    const float A11 = (float)(double)sA11 * FLT_SCALE;
    const float A12 = (float)(double)sA12 * FLT_SCALE;
    const float A22 = (float)(double)sA22 * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    const float min_eig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                          (float)(2 * win * win);
    *err = min_eig;  /* OPTFLOW_LK_GET_MIN_EIGENVALS */
    if (min_eig < min_eig_thr || D < FLT_EPSILON) {
        if (level == 0) *status = 0;
        return 0;
    }
    D = 1.f / D;

    nx -= half; ny -= half;
    float pdx = 0.f, pdy = 0.f;
    const int sJ = J->stride, pj = J->pad;
    int j;
    for (j = 0; j < max_iter; ++j) {
        const int inx = cv_floor(nx), iny = cv_floor(ny);
        if (inx < -win || inx >= J->w || iny < -win || iny >= J->h) {
            if (level == 0) *status = 0;
            break;
        }
        lk_weights(nx - (float)inx, ny - (float)iny, &w00, &w01, &w10, &w11);
        int64_t sb1 = 0, sb2 = 0;
        for (int y = 0; y < win; ++y) {
            const uint8_t *jp = J->img + (size_t)(y + iny + pj) * sJ + (inx + pj);
            for (int x = 0; x < win; ++x) {
                const int diff = DESCALE(jp[x] * w00 + jp[x + 1] * w01 + jp[x + sJ] * w10 + jp[x + sJ + 1] * w11,
                                         W_BITS - 5) - Iw[y * win + x];
                sb1 += (int64_t)diff * Ixw[y * win + x];
                sb2 += (int64_t)diff * Iyw[y * win + x];
            }
        }
        const float b1 = (float)(double)sb1 * FLT_SCALE;
        const float b2 = (float)(double)sb2 * FLT_SCALE;
        const float dx = (A12 * b2 - A22 * b1) * D;
        const float dy = (A12 * b1 - A11 * b2) * D;
        nx += dx; ny += dy;
        next_pt[0] = nx + half; next_pt[1] = ny + half;
        /* delta.ddot(delta) <= criteria.epsilon : both sides double in OpenCV */
        if ((double)dx * dx + (double)dy * dy <= eps2) { ++j; break; }
        /* std::abs(float) < 0.01 (a double literal): for a float v,  v < 0.01  <=>  v <= 0.01f */
        if (j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) {
            next_pt[0] -= dx * 0.5f; next_pt[1] -= dy * 0.5f;
            ++j; break;
        }
        pdx = dx; pdy = dy;
    }
    return j;
}

/* cv::parallel_for_ of calcOpticalFlowPyrLK: OpenCV splits the POINTS of one call over its worker pool
 * (LKTrackerInvoker is the parallel body); same here through ov2o_parallel_for. */
typedef struct {
    const ov2o_pyr *prev, *next;
    int win, max_level, max_iter;
    const float *prev_xy; float *next_xy; uint8_t *status; float *err; int *iters;
    double eps2; float min_eig_thr;
} lk_job;

static void lk_points(void *ctx, int i0, int i1)
{
    const lk_job *s = (const lk_job *)ctx;
    for (int i = i0; i < i1; ++i) {
        s->status[i] = 1; s->err[i] = 0.f;
        for (int l = s->max_level; l >= 0; --l) {
            const int it = lk_level(&s->prev->lv[l], &s->next->lv[l], l, s->max_level, s->prev_xy + 2 * i, s->next_xy + 2 * i,
                                    s->status + i, s->err + i, s->win, s->max_iter, s->eps2, s->min_eig_thr);
            if (s->iters) s->iters[(size_t)i * (s->max_level + 1) + l] = it;
        }
    }
}

void ov2o_calc_optical_flow_pyr_lk(const ov2o_pyr *prev, const ov2o_pyr *next, int n,
                                   const float *prev_xy, float *next_xy, uint8_t *status, float *err,
                                   int win, int max_level, int max_iter, float eps, float min_eig_thr,
                                   int *iters)
{
    if (max_level > prev->nlevels - 1) max_level = prev->nlevels - 1;
    if (max_level > next->nlevels - 1) max_level = next->nlevels - 1;
    /* TermCriteria clamps of calcOpticalFlowPyrLK */
    if (max_iter < 0) max_iter = 0;
    if (max_iter > 100) max_iter = 100;
    double e = eps; if (e < 0.) e = 0.; if (e > 10.) e = 10.;
    lk_job job = {prev, next, win, max_level, max_iter, prev_xy, next_xy, status, err, iters, e * e /* epsilon *= epsilon (double) */,
                  min_eig_thr};
    ov2o_parallel_for(n, 16, lk_points, &job);
}

/* ------------------------------------------------------------------------------------------ */
/* FeatureTracker::fbKltTracking, src/feature_tracker.cpp:35-137                                */

static inline int in_border(float x, float y, int cols, int rows)
{   /* src/feature_tracker.cpp:216-221, BORDER_SIZE = 1 */
    return 1.f <= x && x < (float)cols - 1.f && 1.f <= y && y < (float)rows - 1.f;
}

void ov2o_fb_klt_tracking(const ov2o_pyr *prev, const ov2o_pyr *cur, int win, int nlevels,
                          float err_th, float fb_th, int max_iter, float eps, int n,
                          const float *kps_xy, float *priors_xy, uint8_t *status, int64_t *iter_count)
{
    if (n <= 0) return;                                       /* :43-46 */
    if (prev->nlevels < nlevels + 1) nlevels = prev->nlevels - 1;  /* :50-52 */
    uint8_t *st = (uint8_t *)malloc((size_t)n);
    float *er = (float *)malloc((size_t)n * sizeof(float));
    int *its = (int *)calloc((size_t)n * (nlevels + 1), sizeof(int));
    ov2o_calc_optical_flow_pyr_lk(prev, cur, n, kps_xy, priors_xy, st, er, win, nlevels, max_iter, eps,
                                  1e-4f, its);
    int64_t total = 0;
    for (size_t k = 0; k < (size_t)n * (nlevels + 1); ++k) total += its[k];

    float *newk = (float *)malloc((size_t)n * 2 * sizeof(float));
    float *back = (float *)malloc((size_t)n * 2 * sizeof(float));
    int *idx = (int *)malloc((size_t)n * sizeof(int));
    int m = 0;
    for (int i = 0; i < n; ++i) {
        status[i] = 0;
        if (!st[i]) continue;                                  /* :81 */
        if (er[i] > err_th) continue;                          /* :86 */
        if (!in_border(priors_xy[2 * i], priors_xy[2 * i + 1], cur->lv[0].w, cur->lv[0].h)) continue; /* :91 */
        newk[2 * m] = priors_xy[2 * i]; newk[2 * m + 1] = priors_xy[2 * i + 1];
        back[2 * m] = kps_xy[2 * i]; back[2 * m + 1] = kps_xy[2 * i + 1];
        status[i] = 1; idx[m++] = i;
    }
    if (m > 0) {
        int *its2 = (int *)calloc((size_t)m, sizeof(int));
        ov2o_calc_optical_flow_pyr_lk(cur, prev, m, newk, back, st, er, win, 0, max_iter, eps, 1e-4f, its2);  /* :113 */
        for (int k = 0; k < m; ++k) {
            total += its2[k];
            const int i = idx[k];
            if (!st[k]) { status[i] = 0; continue; }          /* :123 */
            const float dx = kps_xy[2 * i] - back[2 * k], dy = kps_xy[2 * i + 1] - back[2 * k + 1];
            const double nrm = sqrt((double)dx * dx + (double)dy * dy);   /* cv::norm(Point2f) */
            if (nrm > (double)fb_th) status[i] = 0;            /* :128 */
        }
        free(its2);
    }
    if (iter_count) *iter_count = total;
    free(st); free(er); free(its); free(newk); free(back); free(idx);
}

/* ------------------------------------------------------------------------------------------ */
/* VisualFrontEnd::kltTracking batching, src/visual_front_end.cpp:132-275                       */

void ov2o_klt_tracking_frame(const ov2o_pyr *prev, const ov2o_pyr *cur, int win, int nlevels_full,
                             float err_th, float fb_th, int max_iter, float eps, int n,
                             const float *kps_xy, const float *prior_xy, const uint8_t *has_prior,
                             float *out_xy, uint8_t *out_status, int *p3p_req)
{
    int *id3 = (int *)malloc((size_t)n * sizeof(int)), *id2 = (int *)malloc((size_t)n * sizeof(int));
    float *k3 = (float *)malloc((size_t)n * 2 * sizeof(float)), *p3 = (float *)malloc((size_t)n * 2 * sizeof(float));
    float *k2 = (float *)malloc((size_t)n * 2 * sizeof(float)), *p2 = (float *)malloc((size_t)n * 2 * sizeof(float));
    uint8_t *st = (uint8_t *)malloc((size_t)n);
    int n3 = 0, n2 = 0;
    if (p3p_req) *p3p_req = 0;
    for (int i = 0; i < n; ++i) {                                /* :155-184 */
        out_status[i] = 0; out_xy[2 * i] = kps_xy[2 * i]; out_xy[2 * i + 1] = kps_xy[2 * i + 1];
        if (has_prior[i]) {
            k3[2 * n3] = kps_xy[2 * i]; k3[2 * n3 + 1] = kps_xy[2 * i + 1];
            p3[2 * n3] = prior_xy[2 * i]; p3[2 * n3 + 1] = prior_xy[2 * i + 1];
            id3[n3++] = i;
        } else {
            k2[2 * n2] = kps_xy[2 * i]; k2[2 * n2 + 1] = kps_xy[2 * i + 1];
            p2[2 * n2] = kps_xy[2 * i]; p2[2 * n2 + 1] = kps_xy[2 * i + 1];
            id2[n2++] = i;
        }
    }
    if (n3 > 0) {                                                /* :187-234 */
        ov2o_fb_klt_tracking(prev, cur, win, 1, err_th, fb_th, max_iter, eps, n3, k3, p3, st, NULL);
        int good = 0;
        const int n2_before = n2;
        for (int k = 0; k < n3; ++k) {
            if (st[k]) {
                out_status[id3[k]] = 1; out_xy[2 * id3[k]] = p3[2 * k]; out_xy[2 * id3[k] + 1] = p3[2 * k + 1];
                ++good;
            } else {
                k2[2 * n2] = k3[2 * k]; k2[2 * n2 + 1] = k3[2 * k + 1];
                p2[2 * n2] = p3[2 * k]; p2[2 * n2 + 1] = p3[2 * k + 1];   /* forward result kept as prior :219 */
                id2[n2++] = id3[k];
            }
        }
        (void)n2_before;
        if ((double)good < 0.33 * (double)n3) {                  /* :228-233 */
            if (p3p_req) *p3p_req = 1;
            memcpy(p2, k2, (size_t)n2 * 2 * sizeof(float));
        }
    }
    if (n2 > 0) {                                                /* :237-270 */
        ov2o_fb_klt_tracking(prev, cur, win, nlevels_full, err_th, fb_th, max_iter, eps, n2, k2, p2, st, NULL);
        for (int k = 0; k < n2; ++k) {
            out_xy[2 * id2[k]] = p2[2 * k]; out_xy[2 * id2[k] + 1] = p2[2 * k + 1];
            out_status[id2[k]] = st[k];
        }
    }
    free(id3); free(id2); free(k3); free(p3); free(k2); free(p2); free(st);
}
