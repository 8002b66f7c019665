// detect.hip -- keyframe-rate grid detectors + cornerSubPix on gfx950.
//
// Replaces (reference, /root/reference): FeatureExtractor::detectSingleScale src/feature_extractor.cpp:288-440,
// FeatureExtractor::detectGridFAST :443-570 and the cv::cornerSubPix(3x3, 30, 0.01) call that ends both; caller
// MapManager::extractKeypoints src/map_manager.cpp:286-340.  Arithmetic = OpenCV semantics as restated in
// oracle/ov2_oracle_det.c (same fp32 evaluation order, contraction off => bit-identical results).
//
// The reference mutates ONE shared mask from a racy cv::parallel_for_ over the cells; the canonical schedule used by
// the oracle and here is the 2x2 colouring of the cell grid (colour = (r&1)*2+(c&1) ascending): same-colour cells are
// at least one cell apart while a mask disc has radius cell/4, so one launch per colour runs its cells in parallel
// without interaction and the four launches reproduce the sequential result exactly.
//
// Kernels: det_mask_kernel (occupancy + discs of the existing keypoints), det_mineig_kernel / det_fast_kernel (one
// workgroup per cell: response map in LDS, masked arg-max, disc, second arg-max), subpix_kernel (one wave per point).
#include "ov2_internal.h"

#include <cmath>

namespace {

#define DET_MAX_CELL 64
#define DET_MAX_R 16

struct disc_shape {
    int r;
    signed char hw[2 * DET_MAX_R + 1];   // half-width of row offset dy+r of cv::circle's filled midpoint circle
};

disc_shape make_disc(int radius)
{   // drawing.cpp Circle(): spans (cy+-dy: +-dx) and (cy+-dx: +-dy)
    disc_shape s;
    s.r = radius;
    for (int i = 0; i < 2 * DET_MAX_R + 1; ++i) s.hw[i] = -1;
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        auto upd = [&](int row, int half) {
            if (s.hw[radius + row] < half) s.hw[radius + row] = (signed char)half;
            if (s.hw[radius - row] < half) s.hw[radius - row] = (signed char)half;
        };
        upd(dy, dx);
        upd(dx, dy);
        dy++;
        err += plus;
        plus += 2;
        const int m = (err <= 0) - 1;
        err -= minus & m;
        dx += m;
        minus -= m & 2;
    }
    return s;
}

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// all threads of the workgroup zero the disc around (cx, cy) in the w x h mask
__device__ __forceinline__ void draw_disc(unsigned char *mask, int w, int h, int cx, int cy, const disc_shape &ds)
{
    const int side = 2 * ds.r + 1;
    for (int i = threadIdx.x; i < side * side; i += blockDim.x) {
        const int oy = i / side - ds.r, ox = i % side - ds.r;
        const int hw = ds.hw[ds.r + oy];
        const int x = cx + ox, y = cy + oy;
        if (hw >= 0 && ox >= -hw && ox <= hw && x >= 0 && x < w && y >= 0 && y < h) mask[(size_t)y * w + x] = 0;
    }
}

// existing keypoints: occupancy of their grid cell (voccupcells, :316-320 / :466-470) + their discs in the mask.
// One wave per keypoint; `valid` (optional) marks the keypoints that count (e.g. the tracking status).
__global__ __launch_bounds__(64) void det_mask_kernel(const float2 *__restrict__ cur, const int *__restrict__ cur_img,
                                                      const unsigned char *__restrict__ valid, int n_cur, int cell,
                                                      int nwcells, int nhcells, unsigned char *__restrict__ occ_all,
                                                      unsigned char *__restrict__ mask_all, int w, int h, disc_shape ds)
{
    for (int k = blockIdx.x; k < n_cur; k += gridDim.x) {
        if (valid && !valid[k]) continue;
        const float2 p = cur[k];
        const int bimg = cur_img[k];
        if (threadIdx.x == 0) {
            const int cr = (int)(p.y / (float)cell), cc = (int)(p.x / (float)cell);
            if (cr >= 0 && cr <= nhcells && cc >= 0 && cc <= nwcells)
                occ_all[(size_t)bimg * (nhcells + 1) * (nwcells + 1) + cr * (nwcells + 1) + cc] = 1;
        }
        draw_disc(mask_all + (size_t)bimg * w * h, w, h, (int)__builtin_rintf(p.x), (int)__builtin_rintf(p.y), ds);
    }
}

// block-wide exclusive prefix of one flag per thread (256 threads); returns the rank, *total = count of the block
__device__ __forceinline__ int block_rank_256(bool flag, int *sh4, int *total)
{
    const unsigned long long m = __ballot(flag);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) sh4[wv] = __popcll(m);
    __syncthreads();
    int off = 0;
    for (int k = 0; k < wv; ++k) off += sh4[k];
    *total = sh4[0] + sh4[1] + sh4[2] + sh4[3];
    __syncthreads();
    return off + rank;
}

// one workgroup per image: the free, border-valid cells (:338-350) of each colour are appended to the four global
// work lists (one atomic per image and colour reserves the range; the order inside a colour is irrelevant)
__global__ __launch_bounds__(256) void det_worklist_kernel(int cell, int nwcells, int nhcells, int w, int h,
                                                           const unsigned char *__restrict__ occ_all,
                                                           int2 *__restrict__ work /* 4 lists of list_cap */, int list_cap,
                                                           unsigned *__restrict__ wcount /* [4] */)
{
    __shared__ int sh4[4];
    __shared__ int base[4];
    const int b = blockIdx.x, nb = nwcells * nhcells, tid = threadIdx.x;
    const unsigned char *occ = occ_all + (size_t)b * (nhcells + 1) * (nwcells + 1);
    for (int colour = 0; colour < 4; ++colour) {
        int done = 0;   // cells of this colour already placed (block-uniform)
        // pass 1: count, pass 2: place -- two passes keep the reservation to one atomic per (image, colour)
        int cnt = 0;
        for (int i0 = 0; i0 < nb; i0 += 256) {
            const int i = i0 + tid;
            bool f = false;
            if (i < nb) {
                const int rr = i / nwcells, cc = i - rr * nwcells;
                f = ((rr & 1) * 2 + (cc & 1)) == colour && !occ[rr * (nwcells + 1) + cc] &&
                    (cc * cell + cell < w - 1 && rr * cell + cell < h - 1);
            }
            int tot;
            (void)block_rank_256(f, sh4, &tot);
            cnt += tot;
        }
        if (tid == 0) base[colour] = cnt ? (int)atomicAdd(&wcount[colour], (unsigned)cnt) : 0;
        __syncthreads();
        for (int i0 = 0; i0 < nb; i0 += 256) {
            const int i = i0 + tid;
            bool f = false;
            if (i < nb) {
                const int rr = i / nwcells, cc = i - rr * nwcells;
                f = ((rr & 1) * 2 + (cc & 1)) == colour && !occ[rr * (nwcells + 1) + cc] &&
                    (cc * cell + cell < w - 1 && rr * cell + cell < h - 1);
            }
            int tot;
            const int rank = block_rank_256(f, sh4, &tot);
            if (f) work[(size_t)colour * list_cap + base[colour] + done + rank] = make_int2(b, i);
            done += tot;
        }
    }
}

// ordered key: larger value wins, on ties the smaller index (cv::minMaxLoc returns the first maximum)
__device__ __forceinline__ unsigned long long argmax_key(float v, int idx)
{
    unsigned b = __float_as_uint(v);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((unsigned long long)b << 32) | (unsigned)(0x7fffffff - idx);
}

__device__ __forceinline__ unsigned long long block_max_u64(unsigned long long v, unsigned long long *sh)
{
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long t = __shfl_xor(v, o);
        v = t > v ? t : v;
    }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long r = sh[0];
    for (int k = 1; k < (int)(blockDim.x >> 6); ++k) r = sh[k] > r ? sh[k] : r;
    __syncthreads();
    return r;
}

struct det_out {          // per cell
    float fx, fy, sx, sy;
    int has_first, has_second, occupied, pad;
};

// p / n for 0 <= p < 2^13 and 2 <= n <= 66 without the ~35-instruction integer division: (p + 0.5) / n is at least
// 1/(2n) away from every integer, far more than the fp32 error of the product
__device__ __forceinline__ int div_small(int p, float inv_n) { return (int)(((float)p + 0.5f) * inv_n); }

// zero the disc around (cx, cy) in the global mask and in the cell-local copy mk (cell origin x0,y0, side n)
__device__ __forceinline__ void draw_disc_both(unsigned char *mask, int w, int h, unsigned char *mk, int x0, int y0, int n,
                                               int cx, int cy, const disc_shape &ds)
{
    const int side = 2 * ds.r + 1;
    const float inv_side = 1.f / (float)side;
    for (int i = threadIdx.x; i < side * side; i += blockDim.x) {
        const int q = div_small(i, inv_side);
        const int oy = q - ds.r, ox = i - q * side - ds.r;
        const int hw = ds.hw[ds.r + oy];
        const int x = cx + ox, y = cy + oy;
        if (hw >= 0 && ox >= -hw && ox <= hw && x >= 0 && x < w && y >= 0 && y < h) {
            mask[(size_t)y * w + x] = 0;
            const int lx = x - x0, ly = y - y0;
            if (lx >= 0 && lx < n && ly >= 0 && ly < n) mk[ly * n + lx] = 0;
        }
    }
}

// detectSingleScale: one workgroup per free cell of the current colour.  Everything after the first touch of the
// image runs out of LDS: the (cell+2)^2 source patch (the padded plane already holds the image-border REFLECT_101),
// the blurred cell and the Sobel responses stored WITH their per-cell REFLECT_101 ring (so the 3x3 stencils carry no
// border logic), the response map and a local copy of the mask that also receives this cell's own disc.
__global__ __launch_bounds__(256) void det_mineig_kernel(const unsigned char *__restrict__ img0, size_t img_bstride,
                                                         int istride, int w, int h, int cell, int nwcells, int nhcells,
                                                         const int2 *__restrict__ work /* (image, cell) of this colour */,
                                                         const unsigned *__restrict__ work_count,
                                                         unsigned char *__restrict__ mask_all, disc_shape ds, int rx,
                                                         int ry, int rw, int rh, const double *__restrict__ quality_all,
                                                         det_out *__restrict__ out_all)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ unsigned long long shk[4];
    if (blockIdx.x >= *work_count) return;   // the grid is an upper bound; the list length lives on the device
    const int2 item = work[blockIdx.x];   // free, border-valid cells only (det_worklist_kernel did :338-350)
    const int bimg = item.x, i = item.y;
    const unsigned char *img = img0 + img_bstride * bimg;
    unsigned char *mask = mask_all + (size_t)bimg * w * h;
    det_out *out = out_all + (size_t)bimg * nwcells * nhcells;
    const double quality = quality_all[bimg];
    const int rr = i / nwcells, cc = i - rr * nwcells;
    const int tid = threadIdx.x, nth = blockDim.x, n = cell, n2 = cell * cell, m = cell + 2, m2 = m * m;
    const int x0 = cc * cell, y0 = rr * cell;
    const float inv_n = 1.f / (float)n, inv_m = 1.f / (float)m;
    float *dxs = reinterpret_cast<float *>(lds_raw);   // m2, ring = per-cell reflection
    float *dys = dxs + m2;                              // m2
    float *hmap = dys + m2;                             // n2
    unsigned char *src = reinterpret_cast<unsigned char *>(hmap + n2);   // m2: parent pixels, rows/cols -1 .. n
    unsigned char *bl = src + m2;                                        // m2: blurred cell + reflected ring
    unsigned char *mk = bl + m2;                                         // n2: mask of the cell
    for (int p = tid; p < m2; p += nth) {
        const int y = div_small(p, inv_m), x = p - y * m;
        src[p] = img[(ptrdiff_t)(y0 + y - 1) * istride + (x0 + x - 1)];
    }
    for (int p = tid; p < n2; p += nth) {
        const int y = div_small(p, inv_n), x = p - y * n;
        mk[p] = mask[(size_t)(y0 + y) * w + x0 + x];
    }
    __syncthreads();
    // GaussianBlur 3x3 (fixed point, round half up) on parent pixels beyond the cell; ring entry (y,x) = value of the
    // cell pixel (reflect101(y), reflect101(x)), which is what the per-cell REFLECT_101 Sobel reads there
    for (int p = tid; p < m2; p += nth) {
        const int y = div_small(p, inv_m), x = p - y * m;
        const int q = (reflect101(y - 1, n) + 1) * m + reflect101(x - 1, n) + 1;
        const int s = (src[q - m - 1] + 2 * src[q - m] + src[q - m + 1]) + 2 * (src[q - 1] + 2 * src[q] + src[q + 1]) +
                      (src[q + m - 1] + 2 * src[q + m] + src[q + m + 1]);
        bl[p] = (unsigned char)((s + 8) >> 4);
    }
    __syncthreads();
    const float sc = (float)(1.0 / (4.0 * 3.0 * 255.0)), sc2 = sc * 2.f;
    for (int p = tid; p < m2; p += nth) {
        const int y = div_small(p, inv_m), x = p - y * m;
        const int q = (reflect101(y - 1, n) + 1) * m + reflect101(x - 1, n) + 1;
        const float a00 = (float)bl[q - m - 1], a01 = (float)bl[q - m], a02 = (float)bl[q - m + 1];
        const float a10 = (float)bl[q - 1], a12 = (float)bl[q + 1];
        const float a20 = (float)bl[q + m - 1], a21 = (float)bl[q + m], a22 = (float)bl[q + m + 1];
        const float rm = a02 - a00, r0 = a12 - a10, rp = a22 - a20;
        dxs[p] = sc2 * r0 + sc * (rm + rp);
        const float tm = (a00 + a02) * sc + a01 * sc2;
        const float tp = (a20 + a22) * sc + a21 * sc2;
        dys[p] = tp - tm;
    }
    __syncthreads();
    for (int p = tid; p < n2; p += nth) {
        const int y = div_small(p, inv_n), x = p - y * n;
        const int q0 = (y + 1) * m + x + 1;
        double a = 0, b = 0, c = 0;
#pragma unroll
        for (int j = -1; j <= 1; ++j)
#pragma unroll
            for (int ii = -1; ii <= 1; ++ii) {
                const float gx = dxs[q0 + j * m + ii], gy = dys[q0 + j * m + ii];
                a += (double)(gx * gx);
                b += (double)(gx * gy);
                c += (double)(gy * gy);
            }
        const float fa = (float)a * 0.5f, fb = (float)b, fc = (float)c * 0.5f;
        hmap[p] = (fa + fc) - __fsqrt_rn((fa - fc) * (fa - fc) + fb * fb);
    }
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
        unsigned long long key = argmax_key(-3.4028234663852886e38f, 0x7ffffffe);
        for (int p = tid; p < n2; p += nth) {
            const float v = mk[p] ? hmap[p] : 0.f;
            const unsigned long long k2 = argmax_key(v, p);
            key = k2 > key ? k2 : key;
        }
        key = block_max_u64(key, shk);
        const int idx = 0x7fffffff - (int)(key & 0xffffffffu);
        unsigned kb = (unsigned)(key >> 32);
        kb = (kb & 0x80000000u) ? (kb & 0x7fffffffu) : ~kb;
        const float best = __uint_as_float(kb);
        const int iy = div_small(idx, inv_n);
        const int bx = x0 + idx - iy * n, by = y0 + iy;
        if (bx < rx || by < ry || bx >= rx + rw || by >= ry + rh) return;   // `continue` of the reference: cell done
        if ((double)best >= quality) {
            if (tid == 0) {
                if (pass == 0) { out[i].fx = (float)bx; out[i].fy = (float)by; out[i].has_first = 1; }
                else { out[i].sx = (float)bx; out[i].sy = (float)by; out[i].has_second = 1; }
            }
            draw_disc_both(mask, w, h, mk, x0, y0, n, bx, by, ds);
        }
        __syncthreads();
    }
}

// FAST-9/16 score of the centre pixel p (cornerScore<16>); 0 when not a corner at `threshold`
__device__ inline int fast_score(const unsigned char *p, int stride, int threshold)
{
    const int ox[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
    const int oy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
    const int v = p[0];
    int d[25];
#pragma unroll
    for (int k = 0; k < 16; ++k) d[k] = v - p[oy[k] * stride + ox[k]];
#pragma unroll
    for (int k = 16; k < 25; ++k) d[k] = d[k - 16];
    unsigned br = 0, dk = 0;   // bit k: ring pixel k brighter / darker than the centre by more than the threshold
#pragma unroll
    for (int k = 0; k < 16; ++k) { br |= (d[k] < -threshold ? 1u : 0u) << k; dk |= (d[k] > threshold ? 1u : 0u) << k; }
    br |= br << 16; dk |= dk << 16;
    bool corner = false;
#pragma unroll
    for (int s = 0; s < 16; ++s) corner |= (((br >> s) & 0x1ffu) == 0x1ffu) | (((dk >> s) & 0x1ffu) == 0x1ffu);
    if (!corner) return 0;
    int a0 = threshold;
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        int a = min(min(d[k + 1], d[k + 2]), d[k + 3]);
        if (a <= a0) continue;
        a = min(a, min(min(d[k + 4], d[k + 5]), min(min(d[k + 6], d[k + 7]), d[k + 8])));
        a0 = max(a0, min(a, d[k]));
        a0 = max(a0, min(a, d[k + 9]));
    }
    int b0 = -a0;
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        int b = max(max(d[k + 1], d[k + 2]), max(d[k + 3], max(d[k + 4], d[k + 5])));
        if (b >= b0) continue;
        b = max(b, max(d[k + 6], max(d[k + 7], d[k + 8])));
        b0 = min(b0, max(b, d[k]));
        b0 = min(b0, max(b, d[k + 9]));
    }
    return -b0 - 1;
}

// detectGridFAST: one workgroup per cell of the current colour
__global__ __launch_bounds__(256) void det_fast_kernel(const unsigned char *__restrict__ img0, size_t img_bstride,
                                                       int istride, int w, int h, int cell, int nwcells, int nhcells,
                                                       const int2 *__restrict__ work,
                                                       const unsigned *__restrict__ work_count,
                                                       unsigned char *__restrict__ mask_all, disc_shape ds,
                                                       const double *__restrict__ thresh_all,
                                                       det_out *__restrict__ out_all)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ unsigned long long shk[4];
    if (blockIdx.x >= *work_count) return;
    const int2 item = work[blockIdx.x];
    const int bimg = item.x, i = item.y;
    const unsigned char *img = img0 + img_bstride * bimg;
    unsigned char *mask = mask_all + (size_t)bimg * w * h;
    det_out *out = out_all + (size_t)bimg * nwcells * nhcells;
    const int threshold = min(max((int)thresh_all[bimg], 0), 255);
    const int rr = i / nwcells, cc = i % nwcells;
    const int tid = threadIdx.x, nth = blockDim.x, n = cell, n2 = cell * cell;
    const int x0 = cc * cell, y0 = rr * cell;
    int *score = reinterpret_cast<int *>(lds_raw);
    for (int p = tid; p < n2; p += nth) {
        const int y = p / n, x = p - y * n;
        int s = 0;
        if (x >= 3 && x < n - 3 && y >= 3 && y < n - 3) s = fast_score(img + (size_t)(y0 + y) * istride + x0 + x, istride, threshold);
        score[p] = s;
    }
    __syncthreads();
    unsigned long long key = 0ull;   // scores are > 0
    for (int p = tid; p < n2; p += nth) {
        const int y = p / n, x = p - y * n;
        const int s = score[p];
        if (s <= 0) continue;
        bool nms = true;
#pragma unroll
        for (int j = -1; j <= 1; ++j)
#pragma unroll
            for (int ii = -1; ii <= 1; ++ii)
                if ((j || ii) && !(s > score[(y + j) * n + x + ii])) nms = false;
        if (!nms) continue;
        // the reference passes its CV_32F mask to FastFeatureDetector::detect, which reads it as bytes (oracle header)
        if ((x & 3) < 2 || !mask[(size_t)(y0 + y) * w + x0 + (x >> 2)]) continue;
        const unsigned long long k2 = ((unsigned long long)(unsigned)s << 32) | (unsigned)(0x7fffffff - p);
        key = k2 > key ? k2 : key;
    }
    key = block_max_u64(key, shk);
    const int best = (int)(key >> 32);
    if (best >= 20) {
        const int idx = 0x7fffffff - (int)(key & 0xffffffffu);
        const int bx = x0 + idx % n, by = y0 + idx / n;
        if (tid == 0) { out[i].fx = (float)bx; out[i].fy = (float)by; out[i].has_first = 1; }
        draw_disc(mask, w, h, bx, by, ds);
    }
}

// one workgroup per image: candidates in cell order (:393-412 / :532-538), the second candidates of detectSingleScale
// up to the number of still empty cells, the adaptive threshold (:418-423 / :546-552), and the image's points appended
// to the global list cornerSubPix walks (order across images irrelevant).
__global__ __launch_bounds__(256) void det_assemble_kernel(int mode, int nwcells, int nhcells,
                                                           const unsigned char *__restrict__ occ_all,
                                                           const det_out *__restrict__ out_all, double *__restrict__ thresh,
                                                           int *__restrict__ n_out, float2 *__restrict__ out_xy, int out_cap,
                                                           int *__restrict__ pt_ref, int *__restrict__ pt_img,
                                                           unsigned *__restrict__ npts)
{
    __shared__ int sh4[4];
    __shared__ int gbase;
    const int b = blockIdx.x, nb = nwcells * nhcells, tid = threadIdx.x;
    const unsigned char *occ = occ_all + (size_t)b * (nhcells + 1) * (nwcells + 1);
    const det_out *ho = out_all + (size_t)b * nb;
    float2 *o = out_xy + (size_t)b * out_cap;
    int n = 0, nboccup = 0;
    for (int i0 = 0; i0 < nb; i0 += 256) {
        const int i = i0 + tid;
        const bool f = i < nb && ho[i].has_first;
        bool oc = false;
        if (i < nb) { const int rr = i / nwcells, cc = i - rr * nwcells; oc = occ[rr * (nwcells + 1) + cc] != 0; }
        int tot, toc;
        const int rank = block_rank_256(f, sh4, &tot);
        (void)block_rank_256(oc, sh4, &toc);
        if (f) o[n + rank] = make_float2(ho[i].fx, ho[i].fy);
        n += tot; nboccup += toc;
    }
    const int nbempty = nb - nboccup;
    double th = thresh[b];
    if (mode == OV2_DETECT_MINEIG) {
        if (n + nboccup < nb) {
            const int nbsec = nb - (n + nboccup);
            int k = 0;
            for (int i0 = 0; i0 < nb && k < nbsec; i0 += 256) {
                const int i = i0 + tid;
                const bool f = i < nb && ho[i].has_second;
                int tot;
                const int rank = block_rank_256(f, sh4, &tot);
                if (f && k + rank < nbsec) o[n + k + rank] = make_float2(ho[i].sx, ho[i].sy);
                k += tot;
            }
            n += min(k, nbsec);
        }
        if ((double)n < 0.33 * (double)(nb - nboccup)) th /= 2.;
        else if ((double)n > 0.9 * (double)(nb - nboccup)) th *= 1.5;
    } else {
        const int cur_th = (int)th;
        if ((double)n < 0.5 * (double)nbempty && nbempty > 10) th = (double)(int)((double)cur_th * 0.66);
        else if (n == nbempty) th = (double)(int)((double)cur_th * 1.5);
    }
    if (tid == 0) {
        thresh[b] = th;
        n_out[b] = n;
        gbase = n ? (int)atomicAdd(npts, (unsigned)n) : 0;
    }
    __syncthreads();
    for (int k = tid; k < n; k += 256) { pt_ref[gbase + k] = b * out_cap + k; pt_img[gbase + k] = b; }
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | lo);
}

// 16-leaf pairwise tree ((l0+l1)+(l2+l3))+... over a DPP row, total in every lane of the row.  After the two quad steps
// all lanes of a quad agree, so the mirror steps pair equal partial sums exactly like xor 4 / xor 8 would.
__device__ __forceinline__ double row_tree_f64(double v)
{
    v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);   // row_half_mirror
    v += dpp_f64<0x140>(v);   // row_mirror
    return v;
}

// cv::cornerSubPix (win = (HW,HW), zeroZone none): FOUR points per wave, one per DPP row of 16 lanes.
// Per iteration: the (2HW+4)^2 source pixels are fetched once (clamped = getRectSubPix's BORDER_REPLICATE) into LDS,
// the (2HW+3)^2 bilinear samples are formed from LDS, window term t is owned by lane t % 16 (terms t, t+16, ... added
// in that order), the five sums are closed by the 16-leaf tree above (the oracle uses the same order => bit-equal)
// and every lane of the row solves the 2x2 system, so the scalar work of an iteration is shared by four points.
template <int HW>
__global__ __launch_bounds__(64) void subpix_kernel(const unsigned char *__restrict__ img0, size_t img_bstride,
                                                    const int *__restrict__ pt_img, const int *__restrict__ pt_ref,
                                                    const unsigned *__restrict__ npts, int istride, int w, int h,
                                                    float2 *__restrict__ pts, int max_iter, double eps2,
                                                    const float *__restrict__ wmask)
{
    const int n = (int)*npts;                       // the grid is an upper bound; the point count lives on the device
    if ((int)blockIdx.x * 4 >= n) return;
    constexpr int WIN = 2 * HW + 1, BW = WIN + 2, SW = BW + 1, NT = WIN * WIN, NK = (NT + 15) / 16;
    __shared__ unsigned char src[4][(SW * SW + 3) & ~3];
    __shared__ float buf[4][BW * BW];
    const int lane = threadIdx.x, sub = lane & 15, row = lane >> 4;
    const int p = blockIdx.x * 4 + row;
    const bool act = p < n;
    const int pp = act ? p : n - 1;
    const unsigned char *img = img0 + img_bstride * pt_img[pp];
    const int ref = pt_ref[pp];                     // where the point lives in the per-image output array
    const float cTx = pts[ref].x, cTy = pts[ref].y;
    float cIx = cTx, cIy = cTy;
    int iter = 0;
    bool go = act;
    double wm[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) wm[k] = (sub + 16 * k < NT) ? (double)wmask[sub + 16 * k] : 0.0;
    while (__any(go)) {
        const float cx = cIx - (float)(BW - 1) * 0.5f, cy = cIy - (float)(BW - 1) * 0.5f;
        const int ipx = (int)floorf(cx), ipy = (int)floorf(cy);
        const float a = cx - (float)ipx, b = cy - (float)ipy;
        const float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b;
        if (go) {
#pragma unroll
            for (int q = sub; q < SW * SW; q += 16) {
                const int i = q / SW, j = q - i * SW;
                const int xa = min(max(ipx + j, 0), w - 1), ya = min(max(ipy + i, 0), h - 1);
                src[row][q] = img[(size_t)ya * istride + xa];
            }
        }
        __syncthreads();
        if (go) {
#pragma unroll
            for (int q = sub; q < BW * BW; q += 16) {
                const int i = q / BW, j = q - i * BW;
                const unsigned char *sp = &src[row][i * SW + j];
                buf[row][q] = (float)sp[0] * a11 + (float)sp[1] * a12 + (float)sp[SW] * a21 + (float)sp[SW + 1] * a22;
            }
        }
        __syncthreads();
        double A = 0, Bm = 0, C = 0, bb1 = 0, bb2 = 0;
        if (go) {
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int t = sub + 16 * k;
                if (t < NT) {
                    const int i = t / WIN, j = t - i * WIN;
                    const float *sp = &buf[row][(i + 1) * BW + (j + 1)];
                    const double m = wm[k];
                    const double tgx = sp[1] - sp[-1];
                    const double tgy = sp[BW] - sp[-BW];
                    const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
                    const double px = j - HW, py = i - HW;
                    A += gxx; Bm += gxy; C += gyy;
                    bb1 += gxx * px + gxy * py;
                    bb2 += gxy * px + gyy * py;
                }
            }
        }
        A = row_tree_f64(A); Bm = row_tree_f64(Bm); C = row_tree_f64(C);
        bb1 = row_tree_f64(bb1); bb2 = row_tree_f64(bb2);
        if (go) {
            const double det = A * C - Bm * Bm;
            if (fabs(det) <= 2.220446049250313e-16 * 2.220446049250313e-16) go = false;
            else {
                const double scale = 1.0 / det;
                const float nx = (float)(cIx + C * scale * bb1 - Bm * scale * bb2);
                const float ny = (float)(cIy - Bm * scale * bb1 + A * scale * bb2);
                const double err = (double)((nx - cIx) * (nx - cIx) + (ny - cIy) * (ny - cIy));
                cIx = nx; cIy = ny;
                if (cIx < 0 || cIx >= (float)w || cIy < 0 || cIy >= (float)h) go = false;
                else go = (++iter < max_iter) && (err > eps2);
            }
        }
    }
    if (fabsf(cIx - cTx) > (float)HW || fabsf(cIy - cTy) > (float)HW) { cIx = cTx; cIy = cTy; }
    if (act && sub == 0) pts[ref] = make_float2(cIx, cIy);
}

}  // namespace

// Everything on the device, nothing synchronous: mask + occupancy of the existing keypoints, per-colour work lists,
// the four colour launches, assembly in cell order + threshold adaptation, cornerSubPix.  Work-list lengths and the
// point count never visit the host: the dependent grids are upper bounds and their workgroups compare against the
// device-side counters.
extern "C" ov2_status ov2_detect_grid_batch_dev(ov2_ctx *c, const ov2_pyr *pyr, int cell, int mode, double *d_thresh,
                                                int n_cur, const float *d_cur_xy, const int32_t *d_cur_img,
                                                const uint8_t *d_cur_valid, const int *roi, int do_subpix,
                                                int32_t *d_n_out, float *d_out_xy, int out_cap)
{
    if (!c) return OV2_ERR_INVALID;
    if (!pyr || !d_thresh || !d_n_out || !d_out_xy || n_cur < 0) return ov2_set_err(c, OV2_ERR_INVALID, "null argument");
    if (n_cur > 0 && (!d_cur_xy || !d_cur_img)) return ov2_set_err(c, OV2_ERR_INVALID, "null keypoint arrays");
    if (cell < 8 || cell > DET_MAX_CELL || cell / 4 > DET_MAX_R)
        return ov2_set_err(c, OV2_ERR_INVALID, "cell size %d unsupported (8..%d)", cell, DET_MAX_CELL);
    if (mode != OV2_DETECT_FAST && mode != OV2_DETECT_MINEIG) return ov2_set_err(c, OV2_ERR_INVALID, "mode %d", mode);
    OV2_HIP(c, hipSetDevice(c->device));
    {
        const ov2_status ws = ov2_pyr_wait_ready(c, pyr);
        if (ws != OV2_OK) return ws;
    }
    const ov2_pyr_view &v = pyr->buf->view;
    const ov2_level_desc &L = v.lv[0];
    const int B = pyr->buf->batch, w = L.w, h = L.h;
    const unsigned char *img = v.base + L.img_off + (size_t)v.pad * L.istride + OV2_LM;
    const int nh = h / cell, nw = w / cell, nb = nh * nw;
    hipStream_t st = c->stream;
    OV2_HIP(c, hipMemsetAsync(d_n_out, 0, sizeof(int) * B, st));
    if (nb == 0) return OV2_OK;
    if (out_cap < 2 * nb) return ov2_set_err(c, OV2_ERR_INVALID, "out_cap %d < 2 * cells (%d)", out_cap, 2 * nb);
    const int rx = roi ? roi[0] : 0, ry = roi ? roi[1] : 0, rw = roi ? roi[2] : w, rh = roi ? roi[3] : h;
    // scratch: [counters | occupancy | det_out] zeroed per call, then masks | work lists | point refs | weights
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t occ_sz = (size_t)(nh + 1) * (nw + 1);
    const int list_cap = ((nh + 1) / 2) * ((nw + 1) / 2) * B;           // cells of one colour, all images
    const size_t off_cnt = 0, off_occ = 256, off_out = off_occ + up(occ_sz * B);
    const size_t zero_bytes = off_out + up(sizeof(det_out) * nb * B);
    const size_t off_mask = zero_bytes, off_work = off_mask + up((size_t)w * h * B);
    const size_t off_ref = off_work + up(sizeof(int2) * 4 * (size_t)list_cap), off_pimg = off_ref + up(sizeof(int) * (size_t)nb * B);
    const size_t off_wm = off_pimg + up(sizeof(int) * (size_t)nb * B), total = off_wm + 1024;
    void *scr = nullptr;
    ov2_status s = ov2_scratch(c, total, &scr);
    if (s != OV2_OK) return s;
    char *base = (char *)scr;
    unsigned *cnt = (unsigned *)(base + off_cnt);           // [0..3] work-list lengths, [4] number of points
    unsigned char *occ = (unsigned char *)(base + off_occ), *mask = (unsigned char *)(base + off_mask);
    det_out *dout = (det_out *)(base + off_out);
    int2 *dwork = (int2 *)(base + off_work);
    int *pt_ref = (int *)(base + off_ref), *pt_img = (int *)(base + off_pimg);
    float *dwm = (float *)(base + off_wm);
    OV2_HIP(c, hipMemsetAsync(base, 0, zero_bytes, st));
    OV2_HIP(c, hipMemsetAsync(mask, 1, (size_t)w * h * B, st));
    const disc_shape ds = make_disc(cell / 4);
    if (n_cur > 0)
        OV2_LAUNCH(c, OV2_K_DETECT + 1, det_mask_kernel, dim3(std::min(n_cur, 65536)), dim3(64), 0, st,
                   reinterpret_cast<const float2 *>(d_cur_xy), d_cur_img, d_cur_valid, n_cur, cell, nw, nh, occ, mask, w, h, ds);
    OV2_LAUNCH(c, OV2_K_DETECT + 5, det_worklist_kernel, dim3(B), dim3(256), 0, st, cell, nw, nh, w, h, occ, dwork, list_cap, cnt);
    const int nthreads = (cell * cell <= 256) ? 64 : 256;   // small cells: one wave walks the cell
    const size_t mineig_lds = (size_t)(cell + 2) * (cell + 2) * 10 + (size_t)cell * cell * 5;
    for (int colour = 0; colour < 4; ++colour) {
        // cells of this colour per image (upper bound of the list): rows of parity colour>>1 x columns of parity colour&1
        const int rows_c = (nh + 1 - (colour >> 1)) / 2, cols_c = (nw + 1 - (colour & 1)) / 2, nitems = rows_c * cols_c * B;
        if (nitems <= 0) continue;
        const int2 *wl = dwork + (size_t)colour * list_cap;
        if (mode == OV2_DETECT_MINEIG)
            OV2_LAUNCH(c, OV2_K_DETECT, det_mineig_kernel, dim3(nitems), dim3(nthreads), mineig_lds, st, img, L.img_bstride, L.istride,
                       w, h, cell, nw, nh, wl, cnt + colour, mask, ds, rx, ry, rw, rh, d_thresh, dout);
        else
            OV2_LAUNCH(c, OV2_K_DETECT, det_fast_kernel, dim3(nitems), dim3(nthreads), (size_t)cell * cell * 4, st, img, L.img_bstride,
                       L.istride, w, h, cell, nw, nh, wl, cnt + colour, mask, ds, d_thresh, dout);
    }
    OV2_LAUNCH(c, OV2_K_DETECT + 5, det_assemble_kernel, dim3(B), dim3(256), 0, st, mode, nw, nh, occ, dout, d_thresh, d_n_out,
               reinterpret_cast<float2 *>(d_out_xy), out_cap, pt_ref, pt_img, cnt + 4);
    if (do_subpix) {
        const int hw = 3, win = 7;
        float wm[49];
        for (int i = 0; i < win; ++i) {
            const float y = (float)(i - hw) / (float)hw;
            const float vy = expf(-y * y);
            for (int j = 0; j < win; ++j) {
                const float x = (float)(j - hw) / (float)hw;
                wm[i * win + j] = (float)(vy * expf(-x * x));
            }
        }
        OV2_HIP(c, hipMemcpyAsync(dwm, wm, sizeof(wm), hipMemcpyHostToDevice, st));   // 196 B: copied at enqueue time
        OV2_LAUNCH(c, OV2_K_DETECT + 2, subpix_kernel<3>, dim3(((size_t)nb * B + 3) / 4), dim3(64), 0, st, img, L.img_bstride, pt_img,
                   pt_ref, cnt + 4, L.istride, w, h, reinterpret_cast<float2 *>(d_out_xy), 30, 0.01 * 0.01, dwm);
    }
    OV2_HIP(c, hipGetLastError());
    return OV2_OK;
}

// host-pointer form: stages the arguments through pinned memory, runs the device pipeline, one synchronisation
extern "C" ov2_status ov2_detect_grid_batch(ov2_ctx *c, const ov2_pyr *pyr, int cell, int mode, double *thresh,
                                            const int *n_cur, const float *cur_xy, const int *roi, int do_subpix,
                                            int *n_out, float *out_xy, int out_cap)
{
    if (!c) return OV2_ERR_INVALID;
    if (!pyr || !thresh || !n_out || !out_xy || !n_cur) return ov2_set_err(c, OV2_ERR_INVALID, "null argument");
    const int B = pyr->buf->batch;
    int ncur_tot = 0;
    for (int b = 0; b < B; ++b) {
        if (n_cur[b] < 0) return ov2_set_err(c, OV2_ERR_INVALID, "negative keypoint count");
        ncur_tot += n_cur[b];
    }
    if (ncur_tot && !cur_xy) return ov2_set_err(c, OV2_ERR_INVALID, "null cur_xy");
    if (out_cap <= 0) return ov2_set_err(c, OV2_ERR_INVALID, "out_cap %d", out_cap);
    OV2_HIP(c, hipSetDevice(c->device));
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    // pinned staging (host) and its device twin: [thresh B | n_out B | cur_img | cur_xy | out_xy B x cap]
    const size_t o_th = 0, o_no = up(sizeof(double) * B), o_ci = o_no + up(sizeof(int) * B);
    const size_t o_cx = o_ci + up(sizeof(int) * (size_t)ncur_tot), o_ox = o_cx + up(8 * (size_t)ncur_tot);
    const size_t total = o_ox + up(8 * (size_t)B * out_cap);
    char *hp = nullptr, *dp = nullptr;
    ov2_status s = ov2_staging(c, total, (void **)&hp, (void **)&dp);
    if (s != OV2_OK) return s;
    memcpy(hp + o_th, thresh, sizeof(double) * B);
    {
        int *ci = (int *)(hp + o_ci);
        int k = 0;
        for (int b = 0; b < B; ++b)
            for (int q = 0; q < n_cur[b]; ++q) ci[k++] = b;
        if (ncur_tot) memcpy(hp + o_cx, cur_xy, 8 * (size_t)ncur_tot);
    }
    hipStream_t st = c->stream;
    OV2_HIP(c, hipMemcpyAsync(dp, hp, o_ox, hipMemcpyHostToDevice, st));          // everything before the output block
    s = ov2_detect_grid_batch_dev(c, pyr, cell, mode, (double *)(dp + o_th), ncur_tot, (const float *)(dp + o_cx),
                                  (const int32_t *)(dp + o_ci), nullptr, roi, do_subpix, (int32_t *)(dp + o_no),
                                  (float *)(dp + o_ox), out_cap);
    if (s != OV2_OK) return s;
    OV2_HIP(c, hipMemcpyAsync(hp, dp, o_ci, hipMemcpyDeviceToHost, st));          // thresholds + counts
    OV2_HIP(c, hipMemcpyAsync(hp + o_ox, dp + o_ox, 8 * (size_t)B * out_cap, hipMemcpyDeviceToHost, st));
    OV2_HIP(c, hipStreamSynchronize(st));
    memcpy(thresh, hp + o_th, sizeof(double) * B);
    memcpy(n_out, hp + o_no, sizeof(int) * B);
    for (int b = 0; b < B; ++b)
        if (n_out[b] > 0) memcpy(out_xy + (size_t)b * out_cap * 2, hp + o_ox + 8 * (size_t)b * out_cap, 8 * (size_t)n_out[b]);
    return OV2_OK;
}

extern "C" ov2_status ov2_detect_grid(ov2_ctx *c, const ov2_pyr *pyr, int b, int cell, int mode, double *thresh,
                                      int n_cur, const float *cur_xy, const int *roi, int do_subpix, int *n_out,
                                      float *out_xy)
{
    if (!c) return OV2_ERR_INVALID;
    if (!pyr || !thresh || !n_out || !out_xy || n_cur < 0) return ov2_set_err(c, OV2_ERR_INVALID, "null argument");
    const int B = pyr->buf->batch;
    if (b < 0 || b >= B) return ov2_set_err(c, OV2_ERR_INVALID, "image index %d out of the batch", b);
    if (B == 1) {
        const int cap = 2 * (pyr->buf->view.lv[0].w / (cell > 0 ? cell : 1)) * (pyr->buf->view.lv[0].h / (cell > 0 ? cell : 1));
        return ov2_detect_grid_batch(c, pyr, cell, mode, thresh, &n_cur, cur_xy, roi, do_subpix, n_out, out_xy, cap > 0 ? cap : 1);
    }
    // one image of a batch: run the batch call with the other images fully masked out is wasteful; detect on a
    // single-image view of the pyramid instead
    ov2_pyr_buf one = *pyr->buf;
    one.batch = 1;
    one.view.batch = 1;
    for (int l = 0; l < one.view.nlevels; ++l) {
        one.view.lv[l].img_off += one.view.lv[l].img_bstride * b;
        one.view.lv[l].grad_off += one.view.lv[l].grad_bstride * b;
    }
    ov2_pyr tmp;
    tmp.refs.store(1);
    tmp.ctx = pyr->ctx;
    tmp.buf = &one;
    const int cap = 2 * (one.view.lv[0].w / (cell > 0 ? cell : 1)) * (one.view.lv[0].h / (cell > 0 ? cell : 1));
    return ov2_detect_grid_batch(c, &tmp, cell, mode, thresh, &n_cur, cur_xy, roi, do_subpix, n_out, out_xy, cap > 0 ? cap : 1);
}
