// pyramid.hip -- CLAHE + optical-flow pyramid (pyrDown 5x5 + Scharr) for gfx950.
//
// Replaces (reference, /root/reference): VisualFrontEnd::preprocessImage src/visual_front_end.cpp:1143-1177
// = cv::CLAHE::apply + cv::buildOpticalFlowPyramid(img, pyr, Size(9,9), 3); right image src/mapper.cpp:76-81.
// Arithmetic = OpenCV semantics restated in oracle/ov2_oracle_fe.c (integer exact; CLAHE interpolation in
// fp32 with contraction off, v_rndne for cvRound) -- results are bit-identical to the oracle.
//
// HBM layout (one allocation per pyramid batch, see ov2_level_desc): per level a u8 plane and an int16x2
// (Ix,Iy) plane, each padded by `pad` (=win) pixels; the interior of every row starts at column OV2_LM=16
// so that 4-pixel groups are dword (u8) / 16-byte (gradient) aligned.  u8 padding = REFLECT_101, gradient
// padding = 0 (zeroed once when the pooled buffer is created; kernels never write it).
//
// Kernels (all HBM-streaming, byte/int16 work -- no MFMA on purpose):
//   clahe_lut_wave_kernel one wave per (tile, image): LDS histogram -> clip/redistribute -> scan -> LUT
//   level0_kernel         CLAHE bilinear LUT interpolation (or plain copy) -> padded level-0 plane
//   level_kernel          one pass over level l staged in LDS (64x16 tile + 2-px halo):
//                         writes Scharr(l) as 16-byte stores and pyrDown(l) -> level l+1 (+ its reflect border)
#include "ov2_internal.h"

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int reflect101(int i, int n)
{
    // one reflection is enough for |overshoot| < n (callers guarantee it)
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

__device__ __forceinline__ unsigned char sat_u8_rn(float v)
{
    int r = (int)__builtin_rintf(v);  // v_rndne_f32: round half to even == cvRound
    return (unsigned char)min(max(r, 0), 255);
}

// write one u8 pixel of a padded plane together with every REFLECT_101 copy of it in the border.
__device__ __forceinline__ void store_reflections(unsigned char *plane, int istride, int pad, int W, int H, int x,
                                                  int y, unsigned char v, bool write_self)
{
    int xs[3], ys[3], nx = 0, ny = 0;
    xs[nx++] = x;
    if (x >= 1 && x <= pad) xs[nx++] = -x;
    if (x >= W - 1 - pad && x <= W - 2) xs[nx++] = 2 * (W - 1) - x;
    ys[ny++] = y;
    if (y >= 1 && y <= pad) ys[ny++] = -y;
    if (y >= H - 1 - pad && y <= H - 2) ys[ny++] = 2 * (H - 1) - y;
    for (int j = 0; j < ny; ++j)
        for (int i = 0; i < nx; ++i) {
            if (i == 0 && j == 0 && !write_self) continue;
            plane[(size_t)(ys[j] + pad) * istride + OV2_LM + xs[i]] = v;
        }
}

// ---------------------------------------------------------------------------------------------------
// CLAHE LUT, one WAVE per tile (four tiles per workgroup, no workgroup barrier anywhere): the tile's ~2.6 k pixels are
// 12 dwords per lane, the histogram lives in four interleaved LDS copies private to the wave, lane l owns bins
// 4l .. 4l+3 for the clip / redistribute / scan steps (wave shuffles), and writes its four LUT bytes as one dword.
// A workgroup per tile (the first version) is a chain of barrier-separated phases with eight workgroups resident per CU
// (4.2 rounds for 64 images x 135 tiles); here every tile of the batch is resident at once.
#ifndef CLW_COPIES
#define CLW_COPIES 4
#endif
__global__ __launch_bounds__(256) void clahe_lut_wave_kernel(const unsigned char *__restrict__ src, int w, int h,
                                                             int sstride, size_t sbstride, int tiles_x, int tiles_y,
                                                             int tw, int th, int clip_limit, float lut_scale,
                                                             unsigned char *__restrict__ lut)
{
    __shared__ int hist_all[4][256 * CLW_COPIES];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wv, b = blockIdx.y;
    if (tile >= tiles_x * tiles_y) return;           // wave-uniform
    int *hist = hist_all[wv];
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const unsigned char *img = src + sbstride * b;
#pragma unroll
    for (int k = 0; k < 256 * CLW_COPIES / 64 / 4; ++k)
        *reinterpret_cast<int4 *>(&hist[(k * 64 + lane) * 4]) = make_int4(0, 0, 0, 0);
    __builtin_amdgcn_wave_barrier();
    const int x0 = tx * tw, y0 = ty * th, copy = lane & (CLW_COPIES - 1);
    // Tiles of the last tile row / column reach into the REFLECT_101 extension (CLAHE_Impl::apply pads the image to a
    // multiple of the tile grid): an extension pixel is the mirror image of an in-image pixel, so when all mirror sources
    // lie inside this tile's own in-image part (every usual geometry) the tile is histogrammed from that part alone, the
    // mirrored columns / rows counting twice (weights 1, 2, 4) -- the same dword loads as any other tile.
    const int xe = min(x0 + tw, w), ye = min(y0 + th, h);
    const int xm_lo = 2 * w - 1 - x0 - tw, ym_lo = 2 * h - 1 - y0 - th;   // first mirrored column / row (when there is an extension)
    const bool ext_x = x0 + tw > w, ext_y = y0 + th > h;
    if ((!ext_x || (x0 < w - 1 && xm_lo >= x0)) && (!ext_y || (y0 < h - 1 && ym_lo >= y0))) {
        // rows as aligned dwords, pixels of the first / last dword outside [x0, xe) masked
        const int xa = x0 & ~3, ndw = (xe - xa + 3) >> 2;
        const int total = ndw * (ye - y0);
        const unsigned char *base = img + (size_t)y0 * sstride + xa;
        constexpr int CH = 6;
        // the lane's index advances by 64 per load: (row, dword) tracked incrementally instead of divided out
        const int a64 = 64 / ndw, r64 = 64 - a64 * ndw;
        int yy = lane / ndw, dd = lane - yy * ndw;
        for (int i0 = 0; i0 < total; i0 += 64 * CH) {
            unsigned v[CH];
            int xs[CH], ys[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const int i = i0 + k * 64 + lane;
                xs[k] = -1000;
                ys[k] = 0;
                v[k] = 0;
                if (i < total) {
                    v[k] = *reinterpret_cast<const unsigned *>(base + (unsigned)(yy * sstride + 4 * dd));
                    xs[k] = xa + 4 * dd - x0;
                    ys[k] = y0 + yy;
                }
                dd += r64; yy += a64;
                if (dd >= ndw) { dd -= ndw; ++yy; }
            }
            if (!ext_x && !ext_y) {   // wave-uniform
#pragma unroll
                for (int k = 0; k < CH; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if ((unsigned)(xs[k] + j) < (unsigned)tw) atomicAdd(&hist[((v[k] >> (8 * j)) & 255) * CLW_COPIES + copy], 1);
            } else {
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const int wy = (ext_y && ys[k] >= ym_lo && ys[k] <= h - 2) ? 2 : 1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int x = x0 + xs[k] + j;
                        if (x >= x0 && x < xe)
                            atomicAdd(&hist[((v[k] >> (8 * j)) & 255) * CLW_COPIES + copy], (ext_x && x >= xm_lo && x <= w - 2) ? 2 * wy : wy);
                    }
                }
            }
        }
    } else {
        // exotic geometries (an extension wider than the in-image part of its tile): pixel by pixel through the reflection
        const int npx = tw * th;
        for (int i0 = 0; i0 < npx; i0 += 64 * 8) {
            int v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + k * 64 + lane;
                v[k] = -1;
                if (i < npx) {
                    const int yy = i / tw, xx = i - yy * tw;
                    v[k] = img[(size_t)reflect101(y0 + yy, h) * sstride + reflect101(x0 + xx, w)];
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (v[k] >= 0) atomicAdd(&hist[v[k] * CLW_COPIES + copy], 1);
        }
    }
    __builtin_amdgcn_wave_barrier();
    // lane l: bins 4l .. 4l+3
    int hv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int sum = 0;
#pragma unroll
        for (int q = 0; q < CLW_COPIES / 4; ++q) {
            const int4 c = *reinterpret_cast<const int4 *>(&hist[(4 * lane + k) * CLW_COPIES + 4 * q]);
            sum += (c.x + c.y) + (c.z + c.w);
        }
        hv[k] = sum;
    }
    if (clip_limit > 0) {
        int over = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (hv[k] > clip_limit) { over += hv[k] - clip_limit; hv[k] = clip_limit; }
        for (int o = 32; o > 0; o >>= 1) over += __shfl_xor(over, o);
        const int clipped = over;
        const int batch = clipped / 256, residual = clipped - batch * 256;
        // step = 256 / residual and bin / step by float reciprocals: for 0 <= a <= 256 and 1 <= b <= 256,
        // floor(a / b) = (int)((a + 0.5f) * (1.0f / b)) exactly (the true quotient is at least 0.5 / 256 away from an
        // integer boundary, the float error is below 1e-4) -- eight integer divisions of ~25 instructions each per lane gone
        int step = residual ? (int)(256.5f * (1.0f / (float)residual)) : 1;
        if (step < 1) step = 1;
        const float inv_step = 1.0f / (float)step;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int bin = 4 * lane + k;
            hv[k] += batch;
            const int q = (int)(((float)bin + 0.5f) * inv_step);
            if (residual != 0 && bin - q * step == 0 && q < residual) hv[k] += 1;
        }
    }
    // inclusive scan over the 256 bins
    hv[1] += hv[0]; hv[2] += hv[1]; hv[3] += hv[2];
    int tot = hv[3];
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(tot, o);
        if (lane >= o) tot += t;
    }
    const int below = tot - hv[3];
    unsigned outw = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) outw |= (unsigned)sat_u8_rn((float)(hv[k] + below) * lut_scale) << (8 * k);
    reinterpret_cast<unsigned *>(lut + ((size_t)b * tiles_x * tiles_y + tile) * 256)[lane] = outw;
}

// ---------------------------------------------------------------------------------------------------
// level 0: CLAHE interpolation (use_clahe) or copy, 4 px per thread, into the padded plane + reflect border.
// grid (ceil(w/256), h, batch), 64 threads.
__global__ __launch_bounds__(64) void level0_kernel(const unsigned char *__restrict__ src, int w, int h, int sstride,
                                                    size_t sbstride, int use_clahe,
                                                    const unsigned char *__restrict__ lut, int tiles_x, int tiles_y,
                                                    float inv_tw, float inv_th, ov2_pyr_view pv)
{
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y, b = blockIdx.z;
    if (x0 >= w) return;
    const ov2_level_desc L = pv.lv[0];
    unsigned char *plane = pv.base + L.img_off + L.img_bstride * b;
    const unsigned char *srow = src + sbstride * b + (size_t)y * sstride;
    unsigned char out[4];
    const int nvalid = min(4, w - x0);
    unsigned int raw = 0;
    if (nvalid == 4) raw = *reinterpret_cast<const unsigned int *>(srow + x0);  // sstride, x0 multiples of 4
    else for (int i = 0; i < nvalid; ++i) raw |= (unsigned int)srow[x0 + i] << (8 * i);

    if (use_clahe) {
        const float tyf = (float)y * inv_th - 0.5f;
        int ty1 = (int)floorf(tyf);
        int ty2 = ty1 + 1;
        const float ya = tyf - (float)ty1;
        const float ya1 = 1.0f - ya;
        ty1 = max(ty1, 0);
        ty2 = min(ty2, tiles_y - 1);
        const unsigned char *p1 = lut + ((size_t)b * tiles_y + ty1) * tiles_x * 256;
        const unsigned char *p2 = lut + ((size_t)b * tiles_y + ty2) * tiles_x * 256;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int x = x0 + i;
            const float txf = (float)x * inv_tw - 0.5f;
            int tx1 = (int)floorf(txf);
            int tx2 = tx1 + 1;
            const float xa = txf - (float)tx1;
            const float xa1 = 1.0f - xa;
            tx1 = max(tx1, 0);
            tx2 = min(tx2, tiles_x - 1);
            const int v = (raw >> (8 * i)) & 255;
            const int i1 = tx1 * 256 + v, i2 = tx2 * 256 + v;
            const float res = ((float)p1[i1] * xa1 + (float)p1[i2] * xa) * ya1 +
                              ((float)p2[i1] * xa1 + (float)p2[i2] * xa) * ya;
            out[i] = sat_u8_rn(res);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = (unsigned char)((raw >> (8 * i)) & 255);
    }
    unsigned char *drow = plane + (size_t)(y + pv.pad) * L.istride + OV2_LM;
    if (nvalid == 4) {
        *reinterpret_cast<unsigned int *>(drow + x0) =
            (unsigned int)out[0] | ((unsigned int)out[1] << 8) | ((unsigned int)out[2] << 16) | ((unsigned int)out[3] << 24);
    } else {
        for (int i = 0; i < nvalid; ++i) drow[x0 + i] = out[i];
    }
    const bool yedge = (y <= pv.pad) || (y >= h - 1 - pv.pad);
    const bool xedge = (x0 <= pv.pad) || (x0 + 3 >= w - 1 - pv.pad);
    if (yedge || xedge)
        for (int i = 0; i < nvalid; ++i) store_reflections(plane, L.istride, pv.pad, w, h, x0 + i, y, out[i], false);
}

// ---------------------------------------------------------------------------------------------------
// level 0, CLAHE path, tiled: one workgroup = 64 x 64 px, the <= (64/tw+3) x (64/th+3) tile LUTs that its pixels
// interpolate between are staged in LDS once (3 KB for EuRoC geometry) instead of 16 L1/L2 byte gathers per thread.
// Same arithmetic as level0_kernel (which remains the fallback for exotic tile geometries and the copy path).
__global__ __launch_bounds__(256) void level0_clahe_tiled_kernel(const unsigned char *__restrict__ src, int w, int h,
                                                                 int sstride, size_t sbstride,
                                                                 const unsigned char *__restrict__ lut, int tiles_x,
                                                                 int tiles_y, float inv_tw, float inv_th, int ncx_max,
                                                                 ov2_pyr_view pv)
{
    // tile = 64 x 64 px; a thread owns 4 px in each of 4 rows (y0+ty, +16, +32, +48) and issues its 4 source loads
    // back to back: the kernel is bound by the number of memory round trips per resident wave, not by bandwidth.
    extern __shared__ __attribute__((aligned(16))) unsigned char llds[];
    const int tid = threadIdx.x, b = blockIdx.z;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 64;
    const int xl = min(x0 + 63, w - 1), yl = min(y0 + 63, h - 1);
    const int cx_lo = max((int)floorf((float)x0 * inv_tw - 0.5f), 0);
    const int cx_hi = min((int)floorf((float)xl * inv_tw - 0.5f) + 1, tiles_x - 1);
    const int cy_lo = max((int)floorf((float)y0 * inv_th - 0.5f), 0);
    const int cy_hi = min((int)floorf((float)yl * inv_th - 0.5f) + 1, tiles_y - 1);
    const int ncx = cx_hi - cx_lo + 1, ncy = cy_hi - cy_lo + 1;
    const int x = x0 + 4 * (tid & 15), ty = tid >> 4;
    const int nvalid = min(4, w - x);
    const unsigned char *sb = src + sbstride * b;
    unsigned int raw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int y = y0 + ty + 16 * k;
        raw[k] = 0;
        if (x < w && y < h) {
            const unsigned char *srow = sb + (size_t)y * sstride;
            if (nvalid == 4) raw[k] = *reinterpret_cast<const unsigned int *>(srow + x);
            else for (int i = 0; i < nvalid; ++i) raw[k] |= (unsigned int)srow[x + i] << (8 * i);
        }
    }
    const unsigned char *lb = lut + (size_t)b * tiles_x * tiles_y * 256;
    for (int i = tid; i < ncx * ncy * 64; i += 256) {   // dwords
        const int t = i >> 6, d = i & 63;
        const int cy = t / ncx, cx = t - cy * ncx;
        reinterpret_cast<unsigned int *>(llds)[(cy * ncx_max + cx) * 64 + d] =
            reinterpret_cast<const unsigned int *>(lb + ((size_t)(cy_lo + cy) * tiles_x + cx_lo + cx) * 256)[d];
    }
    __syncthreads();
    if (x >= w) return;
    const ov2_level_desc L = pv.lv[0];
    unsigned char *plane = pv.base + L.img_off + L.img_bstride * b;
    // the x interpolation terms are shared by the 4 rows
    float xa[4], xa1[4];
    int o1[4], o2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float txf = (float)(x + i) * inv_tw - 0.5f;
        int tx1 = (int)floorf(txf);
        int tx2 = tx1 + 1;
        xa[i] = txf - (float)tx1;
        xa1[i] = 1.0f - xa[i];
        tx1 = max(tx1, 0);
        tx2 = min(tx2, tiles_x - 1);
        o1[i] = min(max(tx1 - cx_lo, 0), ncx - 1) * 256;
        o2[i] = min(max(tx2 - cx_lo, 0), ncx - 1) * 256;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int y = y0 + ty + 16 * k;
        if (y >= h) break;
        const float tyf = (float)y * inv_th - 0.5f;
        int ty1 = (int)floorf(tyf);
        int ty2 = ty1 + 1;
        const float ya = tyf - (float)ty1;
        const float ya1 = 1.0f - ya;
        ty1 = max(ty1, 0);
        ty2 = min(ty2, tiles_y - 1);
        const unsigned char *p1 = llds + (size_t)(ty1 - cy_lo) * ncx_max * 256;
        const unsigned char *p2 = llds + (size_t)(ty2 - cy_lo) * ncx_max * 256;
        unsigned char out[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int v = (raw[k] >> (8 * i)) & 255;
            const int i1 = o1[i] + v, i2 = o2[i] + v;
            const float res = ((float)p1[i1] * xa1[i] + (float)p1[i2] * xa[i]) * ya1 + ((float)p2[i1] * xa1[i] + (float)p2[i2] * xa[i]) * ya;
            out[i] = sat_u8_rn(res);
        }
        unsigned char *drow = plane + (size_t)(y + pv.pad) * L.istride + OV2_LM;
        if (nvalid == 4) {
            *reinterpret_cast<unsigned int *>(drow + x) =
                (unsigned int)out[0] | ((unsigned int)out[1] << 8) | ((unsigned int)out[2] << 16) | ((unsigned int)out[3] << 24);
        } else {
            for (int i = 0; i < nvalid; ++i) drow[x + i] = out[i];
        }
        const bool yedge = (y <= pv.pad) || (y >= h - 1 - pv.pad);
        const bool xedge = (x <= pv.pad) || (x + 3 >= w - 1 - pv.pad);
        if (yedge || xedge)
            for (int i = 0; i < nvalid; ++i) store_reflections(plane, L.istride, pv.pad, w, h, x + i, y, out[i], false);
    }
}

// ---------------------------------------------------------------------------------------------------
// level l -> Scharr(l) [+ pyrDown -> level l+1].  Tile 64x16 px of level l, 256 threads.
// grid (ceil(w/64), ceil(h/16), batch).
#define TILE_W 64
// 64 rows per workgroup: a thread owns 4 Scharr groups and 4 pyrDown outputs, so each resident wave carries several
// memory round trips' worth of work (with 16-row tiles the kernel was bound by round trips per wave, not bandwidth)
#define TILE_H 16
#define LDS_ROWS (TILE_H + 4)  // rows y0-2 .. y0+TILE_H+1
#define LDS_DW 18              // dwords per row: cols x0-4 .. x0+67

__global__ __launch_bounds__(256) void level_kernel(ov2_pyr_view pv, int l, int has_next, int do_scharr)
{
    __shared__ unsigned int tile[LDS_ROWS][LDS_DW];
    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * TILE_W, y0 = blockIdx.y * TILE_H;
    const ov2_level_desc L = pv.lv[l];
    const int pad = pv.pad;
    const unsigned char *img = pv.base + L.img_off + L.img_bstride * b;

    // stage: dword loads from the padded plane (border already holds REFLECT_101, so no index logic)
    for (int i = tid; i < LDS_ROWS * LDS_DW; i += 256) {
        const int r = i / LDS_DW, c = i - r * LDS_DW;
        int row = y0 - 2 + r + pad;                       // padded row index
        row = min(row, L.h + 2 * pad - 1);                // tiles hanging over the bottom edge
        int dw = (OV2_LM + x0 - 4) / 4 + c;               // dword column
        dw = min(dw, L.istride / 4 - 1);                  // tiles hanging over the right edge
        tile[r][c] = *reinterpret_cast<const unsigned int *>(img + (size_t)row * L.istride + 4 * dw);
    }
    __syncthreads();

    // ---- Scharr: 4 px per thread ------------------------------------------------------------------
    if (do_scharr)
    for (int rr = 0; rr < TILE_H / 16; ++rr) {
        const int tx = tid & 15, ty = (tid >> 4) + 16 * rr;
        const int x = x0 + 4 * tx, y = y0 + ty;
        if (y < L.h && x < L.w) {
            // bytes x-4 .. x+7 of rows y-1,y,y+1  ->  need cols x-1 .. x+4
            int t0[6], t1[6];
            unsigned int ra[3], rb[3], rc[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                ra[k] = tile[ty + 1][tx + k];
                rb[k] = tile[ty + 2][tx + k];
                rc[k] = tile[ty + 3][tx + k];
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const int byte = 3 + c;  // col x-1 is byte 3 of the 12 loaded
                const int a = (ra[byte >> 2] >> (8 * (byte & 3))) & 255;
                const int m = (rb[byte >> 2] >> (8 * (byte & 3))) & 255;
                const int d = (rc[byte >> 2] >> (8 * (byte & 3))) & 255;
                t0[c] = (a + d) * 3 + m * 10;
                t1[c] = d - a;
            }
            short g[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                g[2 * i] = (short)(t0[i + 2] - t0[i]);
                g[2 * i + 1] = (short)((t1[i] + t1[i + 2]) * 3 + t1[i + 1] * 10);
            }
            short *grow = reinterpret_cast<short *>(pv.base + L.grad_off + L.grad_bstride * b) +
                          ((size_t)(y + pad) * L.gstride + OV2_LM + x) * 2;
            if (x + 3 < L.w) {
                uint4 v;
                v.x = (unsigned short)g[0] | ((unsigned int)(unsigned short)g[1] << 16);
                v.y = (unsigned short)g[2] | ((unsigned int)(unsigned short)g[3] << 16);
                v.z = (unsigned short)g[4] | ((unsigned int)(unsigned short)g[5] << 16);
                v.w = (unsigned short)g[6] | ((unsigned int)(unsigned short)g[7] << 16);
                *reinterpret_cast<uint4 *>(grow) = v;
            } else {
                for (int i = 0; i < L.w - x; ++i) { grow[2 * i] = g[2 * i]; grow[2 * i + 1] = g[2 * i + 1]; }
            }
        }
    }

    // ---- pyrDown: 32 x (TILE_H/2) outputs per tile, TILE_H/16 per thread -----------------------------------
    if (has_next)
    for (int rr = 0; rr < TILE_H / 16; ++rr) {
        const ov2_level_desc N = pv.lv[l + 1];
        const int lx = tid & 31, ly = (tid >> 5) + 8 * rr;
        const int xo = (x0 >> 1) + lx, yo = (y0 >> 1) + ly;
        const int c0 = 2 + 2 * lx;  // first byte (col 2xo-2) inside the LDS row
        int acc = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const unsigned int d0 = tile[2 * ly + j][c0 >> 2], d1 = tile[2 * ly + j][(c0 >> 2) + 1];
            const unsigned long long v = (((unsigned long long)d1 << 32) | d0) >> (8 * (c0 & 3));
            const int p0 = (int)(v & 255), p1 = (int)((v >> 8) & 255), p2 = (int)((v >> 16) & 255),
                      p3 = (int)((v >> 24) & 255), p4 = (int)((v >> 32) & 255);
            const int r = p0 + p4 + 4 * (p1 + p3) + 6 * p2;
            const int kj = (j == 0 || j == 4) ? 1 : ((j == 2) ? 6 : 4);
            acc += kj * r;
        }
        const unsigned int val = (unsigned int)((acc + 128) >> 8);
        // pack 4 neighbouring outputs into one dword store where the group is complete
        const unsigned int v1 = __shfl_down(val, 1), v2 = __shfl_down(val, 2), v3 = __shfl_down(val, 3);
        unsigned char *nplane = pv.base + N.img_off + N.img_bstride * b;
        const bool valid = xo < N.w && yo < N.h;
        const int gx = xo & ~3;  // group origin
        const bool full = (gx + 3 < N.w) && yo < N.h;
        if (full) {
            if ((lx & 3) == 0)
                *reinterpret_cast<unsigned int *>(nplane + (size_t)(yo + pad) * N.istride + OV2_LM + xo) =
                    val | (v1 << 8) | (v2 << 16) | (v3 << 24);
        } else if (valid) {
            nplane[(size_t)(yo + pad) * N.istride + OV2_LM + xo] = (unsigned char)val;
        }
        if (valid) {
            const bool edge = (xo <= pad) || (xo >= N.w - 1 - pad) || (yo <= pad) || (yo >= N.h - 1 - pad);
            if (edge) store_reflections(nplane, N.istride, pad, N.w, N.h, xo, yo, (unsigned char)val, false);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// pyrDown alone (what a build runs: the Scharr planes are written on demand).  Tile = 128 x 32 px of level l in LDS
// (+ 2-px halo, rows as dwords from the column 4 left of the tile), a thread owns FOUR adjacent outputs: per source row
// one 16-byte LDS read holds the 11 pixels they touch, the 5 x 5 binomial is applied as 1-4-6-4-1 along the row and
// down the five rows (exact integers), and the four results leave as one dword store.  Same arithmetic as
// level_kernel's pyrDown part (oracle: pyr_down_rows); grid (ceil(w/128), ceil(h/32), batch), 256 threads.
#define PD_TW 128
#define PD_TH 32
#define PD_ROWS (PD_TH + 4)       // rows y0-2 .. y0+PD_TH+1
#define PD_DW (PD_TW / 4 + 4)     // dwords per row: cols x0-4 .. x0+PD_TW+11

__global__ __launch_bounds__(256) void pyrdown_kernel(ov2_pyr_view pv, int l)
{
    __shared__ __attribute__((aligned(16))) unsigned int tile[PD_ROWS][PD_DW];
    const int tid = threadIdx.x, b = blockIdx.z;
    const int x0 = blockIdx.x * PD_TW, y0 = blockIdx.y * PD_TH;
    const ov2_level_desc L = pv.lv[l], N = pv.lv[l + 1];
    const int pad = pv.pad;
    const unsigned char *img = pv.base + L.img_off + L.img_bstride * b;
    for (int i = tid; i < PD_ROWS * PD_DW; i += 256) {
        const int r = i / PD_DW, c = i - r * PD_DW;
        const int row = min(y0 - 2 + r + pad, L.h + 2 * pad - 1);            // tiles hanging over the bottom edge
        const int dw = min((OV2_LM + x0 - 4) / 4 + c, L.istride / 4 - 1);    // ... and over the right edge
        tile[r][c] = *reinterpret_cast<const unsigned int *>(img + (size_t)row * L.istride + 4 * dw);
    }
    __syncthreads();
    // thread -> outputs (xo .. xo+3, yo): 16 groups per output row, 16 output rows
    const int gx = tid & 15, ly = tid >> 4;
    const int xo = (x0 >> 1) + 4 * gx, yo = (y0 >> 1) + ly;
    if (yo >= N.h || xo >= N.w) return;
    int acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        // source columns 2 xo - 2 .. 2 xo + 8 = bytes 2 .. 12 of the 16 bytes from LDS column 8 gx (tile column 0 = x0 - 4);
        // output o takes bytes 2 + 2o .. 6 + 2o: v_dot4_u32_u8 with (1, 4, 6, 4) on the first four, the fifth as addend
        const uint4 q = *reinterpret_cast<const uint4 *>(&tile[2 * ly + j][2 * gx]);
        const unsigned d0 = __builtin_amdgcn_alignbyte(q.y, q.x, 2), d2 = __builtin_amdgcn_alignbyte(q.z, q.y, 2);
        const unsigned wts = 0x04060401u;
        int r[4];
        r[0] = (int)__builtin_amdgcn_udot4(d0, wts, (q.y >> 16) & 255u, false);
        r[1] = (int)__builtin_amdgcn_udot4(q.y, wts, q.z & 255u, false);
        r[2] = (int)__builtin_amdgcn_udot4(d2, wts, (q.z >> 16) & 255u, false);
        r[3] = (int)__builtin_amdgcn_udot4(q.z, wts, q.w & 255u, false);
        const int kj = (j == 0 || j == 4) ? 1 : ((j == 2) ? 6 : 4);
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[o] += kj * r[o];
    }
    unsigned char out[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) out[o] = (unsigned char)((acc[o] + 128) >> 8);
    unsigned char *nplane = pv.base + N.img_off + N.img_bstride * b;
    unsigned char *dp = nplane + (size_t)(yo + pad) * N.istride + OV2_LM + xo;
    const int nv = min(4, N.w - xo);
    if (nv == 4) *reinterpret_cast<unsigned int *>(dp) = (unsigned)out[0] | ((unsigned)out[1] << 8) | ((unsigned)out[2] << 16) | ((unsigned)out[3] << 24);
    else for (int o = 0; o < nv; ++o) dp[o] = out[o];
    const bool edge = (xo <= pad) || (xo + 3 >= N.w - 1 - pad) || (yo <= pad) || (yo >= N.h - 1 - pad);
    if (edge)
        for (int o = 0; o < nv; ++o) store_reflections(nplane, N.istride, pad, N.w, N.h, xo + o, yo, out[o], false);
}

// ---------------------------------------------------------------------------------------------------
// level 0 (CLAHE path) AND the first pyrDown in one pass: the workgroup forms the CLAHE output of a 128 x 32 px tile
// plus the 2-px ring the 5 x 5 binomial reaches into (36 rows x 36 dwords, columns x0-4 .. x0+139), keeps it in LDS,
// writes the tile's own pixels to the level-0 plane (128-byte rows: whole lines) and then runs pyrdown_kernel's
// arithmetic from LDS -- level 0 is not read back, one launch tail less.  Ring positions outside the image take the value
// of their REFLECT_101 source pixel, which is what the padded plane holds there.
// The interpolation makes ONE LDS read per pixel: for every interpolation cell the tile touches (the rectangle between
// four neighbouring tile centres; <= 3 x 2 for EuRoC geometry) the four surrounding LUTs are interleaved into a
// 256-entry table of byte quads, so a pixel value fetches all four taps with one dword read (v_cvt_f32_ubyte0..3 unpack
// them); the per-pixel arithmetic is the expression of level0_kernel evaluated two taps at a time by the packed fp32
// instructions (each product and sum rounded separately), so the bytes are the same.  The kernel is bound by vector
// instruction issue, so tiles whose staged region lies inside the image and away from the plane's reflected border
// (workgroup-uniform test) take a path without any reflect / clamp / edge-store logic, and every address is a uniform
// base plus a 32-bit lane offset.
template <bool INTERIOR>
__device__ __forceinline__ void l0pd_stage(const unsigned char *__restrict__ sb, int w, int h, int sstride,
                                           unsigned char *__restrict__ plane, int istride, int pad, float inv_tw, float inv_th,
                                           int txA, int tyA, int ncellx, int ncelly, int x0, int y0,
                                           unsigned int (*tile)[PD_DW], const unsigned int *quad)
{
    const int tid = threadIdx.x;
    const int c = tid % PD_DW, g = tid / PD_DW;   // dword column of the staged region (252 threads busy), rows g, g + 7, ...
    if (g >= 7) return;
    const int x = x0 - 4 + 4 * c;
    const bool col_live = INTERIOR || x < w + 4;          // columns further right feed no output
    const bool col_in = INTERIOR || (x >= 0 && x + 3 < w);   // the four pixels are four consecutive source bytes
    int xr[4];
    unsigned cellx[4];   // byte offset of the pixel's cell column in the quad tables
    float xa[4], xa1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        xr[i] = INTERIOR ? x + i : min(max(reflect101(x + i, w), 0), w - 1);
        const float txf = (float)xr[i] * inv_tw - 0.5f;
        const float fl = floorf(txf);
        xa[i] = txf - fl;
        xa1[i] = 1.0f - xa[i];
        cellx[i] = (unsigned)min(max((int)fl - txA, 0), ncellx - 1) * 1024u;
    }
    unsigned int raw[6];
    {   // the source loads of all rows are in flight together
        const int yfirst = y0 - 2 + g;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int r = g + 7 * k, y = yfirst + 7 * k;
            raw[k] = 0;
            if (INTERIOR) {
                if (r < PD_ROWS) raw[k] = *reinterpret_cast<const unsigned int *>(sb + (unsigned)(y * sstride + x));
            } else if (col_live && r < PD_ROWS && y < h + 2) {
                const unsigned ro = (unsigned)(min(max(reflect101(y, h), 0), h - 1) * sstride);
                if (col_in) raw[k] = *reinterpret_cast<const unsigned int *>(sb + (ro + (unsigned)x));
                else
                    for (int i = 0; i < 4; ++i) raw[k] |= (unsigned int)sb[ro + (unsigned)xr[i]] << (8 * i);
            }
        }
    }
    __syncthreads();   // quad tables complete
    const bool own_col = c >= 1 && c <= PD_TW / 4 && (INTERIOR || x < w);
    const unsigned char *qb = reinterpret_cast<const unsigned char *>(quad);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int r = g + 7 * k, y = y0 - 2 + r;
        if (r >= PD_ROWS) break;
        unsigned int packed = 0;
        if (INTERIOR || (col_live && y < h + 2)) {
            const int yr = INTERIOR ? y : min(max(reflect101(y, h), 0), h - 1);
            const float tyf = (float)yr * inv_th - 0.5f;
            const float fl = floorf(tyf);
            const float ya = tyf - fl;
            const v2f yw = {1.0f - ya, ya};
            const unsigned celly = (unsigned)(min(max((int)fl - tyA, 0), ncelly - 1) * ncellx) * 1024u;
            unsigned int rb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // ((tl xa1 + tr xa) ya1 + (bl xa1 + br xa) ya), every product and sum rounded on its own, two at a time
                // (v_pk_mul_f32 / v_pk_add_f32); the result lies in [0, 255], so adding 2^23 leaves cvRound(res) in the
                // low mantissa byte (round-to-nearest-even, like v_rndne)
                const unsigned q = *reinterpret_cast<const unsigned int *>(qb + (celly + cellx[i] + 4u * ((raw[k] >> (8 * i)) & 255u)));
                const v2f left = {(float)(q & 255u), (float)((q >> 16) & 255u)}, right = {(float)((q >> 8) & 255u), (float)(q >> 24)};
                const v2f xl = {xa1[i], xa1[i]}, xr2 = {xa[i], xa[i]};
                const v2f tb = (left * xl + right * xr2) * yw;
                rb[i] = __float_as_uint((tb.x + tb.y) + 8388608.0f);
            }
            packed = __builtin_amdgcn_perm(__builtin_amdgcn_perm(rb[3], rb[2], 0x0c0c0400u),
                                           __builtin_amdgcn_perm(rb[1], rb[0], 0x0c0c0400u), 0x05040100u);
            // the tile's own pixels go to the level-0 plane
            if (own_col && r >= 2 && r < PD_TH + 2 && (INTERIOR || y < h)) {
                const unsigned po = (unsigned)((y + pad) * istride + OV2_LM + x);
                if (INTERIOR) {
                    *reinterpret_cast<unsigned int *>(plane + po) = packed;
                } else {
                    const int nvalid = min(4, w - x);
                    if (nvalid == 4) *reinterpret_cast<unsigned int *>(plane + po) = packed;
                    else for (int i = 0; i < nvalid; ++i) plane[po + i] = (unsigned char)(packed >> (8 * i));
                    const bool yedge = (y <= pad) || (y >= h - 1 - pad);
                    const bool xedge = (x <= pad) || (x + 3 >= w - 1 - pad);
                    if (yedge || xedge)
                        for (int i = 0; i < nvalid; ++i)
                            store_reflections(plane, istride, pad, w, h, x + i, y, (unsigned char)(packed >> (8 * i)), false);
                }
            }
        }
        tile[r][c] = packed;
    }
}

__global__ __launch_bounds__(256) void level0_clahe_pyrdown_kernel(const unsigned char *__restrict__ src, int w, int h,
                                                                   int sstride, size_t sbstride,
                                                                   const unsigned char *__restrict__ lut, int tiles_x,
                                                                   int tiles_y, float inv_tw, float inv_th, ov2_pyr_view pv)
{
    extern __shared__ __attribute__((aligned(16))) unsigned int fl[];
    unsigned int (*tile)[PD_DW] = reinterpret_cast<unsigned int (*)[PD_DW]>(fl);   // PD_ROWS x PD_DW
    unsigned int *quad = fl + PD_ROWS * PD_DW;                                     // cells x 256 byte quads
    const int tid = threadIdx.x, b = blockIdx.z;
    const int x0 = blockIdx.x * PD_TW, y0 = blockIdx.y * PD_TH;
    const int pad = pv.pad;
    const int xmin = max(x0 - 4, 0), xmax = min(x0 + PD_TW + 11, w - 1);
    const int ymin = max(y0 - 2, 0), ymax = min(y0 + PD_TH + 1, h - 1);
    const int txA = (int)floorf((float)xmin * inv_tw - 0.5f), txB = (int)floorf((float)xmax * inv_tw - 0.5f);
    const int tyA = (int)floorf((float)ymin * inv_th - 0.5f), tyB = (int)floorf((float)ymax * inv_th - 0.5f);
    const int ncellx = txB - txA + 1, ncelly = tyB - tyA + 1;
    // ---- byte-quad tables of the cells: item = (cell, group of four values); value v -> (tl | tr << 8 | bl << 16 | br << 24)
    const unsigned char *lb = lut + (size_t)b * tiles_x * tiles_y * 256;
    for (int i = tid; i < ncellx * ncelly * 64; i += 256) {
        const int cell = i >> 6, v4 = i & 63;
        const int jy = cell / ncellx, jx = cell - jy * ncellx;
        const int c1 = min(max(txA + jx, 0), tiles_x - 1), c2 = min(max(txA + jx + 1, 0), tiles_x - 1);
        const int r1 = min(max(tyA + jy, 0), tiles_y - 1), r2 = min(max(tyA + jy + 1, 0), tiles_y - 1);
        const unsigned A = reinterpret_cast<const unsigned int *>(lb + (unsigned)((r1 * tiles_x + c1) * 256))[v4];
        const unsigned Bq = reinterpret_cast<const unsigned int *>(lb + (unsigned)((r1 * tiles_x + c2) * 256))[v4];
        const unsigned Cq = reinterpret_cast<const unsigned int *>(lb + (unsigned)((r2 * tiles_x + c1) * 256))[v4];
        const unsigned D = reinterpret_cast<const unsigned int *>(lb + (unsigned)((r2 * tiles_x + c2) * 256))[v4];
        // 4 x 4 byte transpose: AB = (A0 B0 A1 B1 | A2 B2 A3 B3), CD alike, then the halves are paired
        const unsigned ab_lo = __builtin_amdgcn_perm(Bq, A, 0x05010400u), ab_hi = __builtin_amdgcn_perm(Bq, A, 0x07030602u);
        const unsigned cd_lo = __builtin_amdgcn_perm(D, Cq, 0x05010400u), cd_hi = __builtin_amdgcn_perm(D, Cq, 0x07030602u);
        uint4 o;
        o.x = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x05040100u); o.y = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x07060302u);
        o.z = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x05040100u); o.w = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x07060302u);
        *reinterpret_cast<uint4 *>(quad + cell * 256 + 4 * v4) = o;
    }
    const ov2_level_desc L = pv.lv[0];
    const unsigned char *sb = src + sbstride * b;
    unsigned char *plane = pv.base + L.img_off + L.img_bstride * b;
    // staged region inside the image, own pixels clear of the rows / columns whose reflections the padded plane holds
    const bool interior = x0 - 4 >= 0 && x0 + PD_TW + 11 < w && y0 - 2 >= 0 && y0 + PD_TH + 1 < h &&
                          x0 > pad && x0 + PD_TW - 1 < w - 1 - pad && y0 > pad && y0 + PD_TH - 1 < h - 1 - pad;
    if (interior) l0pd_stage<true>(sb, w, h, sstride, plane, L.istride, pad, inv_tw, inv_th, txA, tyA, ncellx, ncelly, x0, y0, tile, quad);
    else l0pd_stage<false>(sb, w, h, sstride, plane, L.istride, pad, inv_tw, inv_th, txA, tyA, ncellx, ncelly, x0, y0, tile, quad);
    __syncthreads();
    // ---- pyrDown of the staged tile (pyrdown_kernel's second half, l = 0)
    const ov2_level_desc N = pv.lv[1];
    const int gx = tid & 15, ly = tid >> 4;
    const int xo = (x0 >> 1) + 4 * gx, yo = (y0 >> 1) + ly;
    if (yo >= N.h || xo >= N.w) return;
    int acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const uint4 q = *reinterpret_cast<const uint4 *>(&tile[2 * ly + j][2 * gx]);
        const unsigned d0 = __builtin_amdgcn_alignbyte(q.y, q.x, 2), d2 = __builtin_amdgcn_alignbyte(q.z, q.y, 2);
        const unsigned wts = 0x04060401u;
        int rr[4];
        rr[0] = (int)__builtin_amdgcn_udot4(d0, wts, (q.y >> 16) & 255u, false);
        rr[1] = (int)__builtin_amdgcn_udot4(q.y, wts, q.z & 255u, false);
        rr[2] = (int)__builtin_amdgcn_udot4(d2, wts, (q.z >> 16) & 255u, false);
        rr[3] = (int)__builtin_amdgcn_udot4(q.z, wts, q.w & 255u, false);
        const int kj = (j == 0 || j == 4) ? 1 : ((j == 2) ? 6 : 4);
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[o] += kj * rr[o];
    }
    const unsigned outp = (unsigned)((acc[0] + 128) >> 8) | ((unsigned)((acc[1] + 128) >> 8) << 8) |
                          ((unsigned)((acc[2] + 128) >> 8) << 16) | ((unsigned)((acc[3] + 128) >> 8) << 24);
    unsigned char *nplane = pv.base + N.img_off + N.img_bstride * b;
    const unsigned no = (unsigned)((yo + pad) * N.istride + OV2_LM + xo);
    const int nv = min(4, N.w - xo);
    if (nv == 4) *reinterpret_cast<unsigned int *>(nplane + no) = outp;
    else for (int o = 0; o < nv; ++o) nplane[no + o] = (unsigned char)(outp >> (8 * o));
    const bool edge = (xo <= pad) || (xo + 3 >= N.w - 1 - pad) || (yo <= pad) || (yo >= N.h - 1 - pad);
    if (edge)
        for (int o = 0; o < nv; ++o) store_reflections(nplane, N.istride, pad, N.w, N.h, xo + o, yo, (unsigned char)(outp >> (8 * o)), false);
}

// ---------------------------------------------------------------------------------------------------
// host side

ov2_status acquire_buf(ov2_ctx *c, int w, int h, int pad, int max_level, int batch, ov2_pyr_buf **out)
{
    {
        // prefer a pooled buffer whose last consumers have already finished: taking the one released a moment ago
        // would chain the new build behind the KLT that still reads it and defeat the two-stream overlap.  Up to
        // OV2_PYR_RING buffers per geometry are kept so that a finished one is normally available.
        std::lock_guard<std::mutex> g(c->mu);
        int pending = -1, same = 0;
        for (size_t i = 0; i < c->pool.size(); ++i) {
            ov2_pyr_buf *b = c->pool[i];
            if (!(b->w == w && b->h == h && b->pad == pad && b->max_level == max_level && b->batch == batch)) continue;
            ++same;
            if ((!b->has_free_ev || hipEventQuery(b->free_ev) == hipSuccess) &&
                (!b->has_free_ev2 || hipEventQuery(b->free_ev2) == hipSuccess)) {
                c->pool.erase(c->pool.begin() + i);
                *out = b;
                return OV2_OK;
            }
            if (pending < 0) pending = (int)i;
        }
        if (pending >= 0 && same >= 2) {   // two idle-but-pending buffers already exist: reuse rather than grow
            ov2_pyr_buf *b = c->pool[pending];
            c->pool.erase(c->pool.begin() + pending);
            *out = b;
            return OV2_OK;
        }
    }
    ov2_pyr_buf *b = new ov2_pyr_buf();
    b->w = w; b->h = h; b->pad = pad; b->max_level = max_level; b->batch = batch;
    b->lut = nullptr;
    ov2_pyr_view &v = b->view;
    memset(&v, 0, sizeof(v));
    v.pad = pad;
    v.batch = batch;
    size_t off = 0;
    int cw = w, ch = h, nl = 0;
    for (int l = 0; l <= max_level && l < OV2_MAX_LEVELS; ++l) {
        ov2_level_desc &L = v.lv[l];
        L.w = cw; L.h = ch;
        L.istride = ov2_round_up(OV2_LM + cw + pad, 64);
        L.gstride = ov2_round_up(OV2_LM + cw + pad, 16);
        L.rows = ch + 2 * pad;
        L.img_bstride = ((size_t)L.istride * L.rows + 255) / 256 * 256;
        L.grad_bstride = ((size_t)L.gstride * 4 * L.rows + 255) / 256 * 256;
        L.img_off = off; off += L.img_bstride * batch;
        L.grad_off = off; off += L.grad_bstride * batch;
        nl = l + 1;
        cw = (cw + 1) / 2; ch = (ch + 1) / 2;
        if (cw <= pad || ch <= pad) break;  // buildOpticalFlowPyramid early stop (next level <= winSize)
    }
    v.nlevels = nl;
    b->bytes = off + 4096;
    hipError_t e = hipMalloc((void **)&b->base, b->bytes);
    if (e != hipSuccess) {
        delete b;
        return ov2_set_err(c, OV2_ERR_NOMEM, "pyramid hipMalloc(%zu): %s", off, hipGetErrorString(e));
    }
    e = hipMemsetAsync(b->base, 0, b->bytes, c->stream_pyr);  // gradient padding = 0 for the lifetime of the buffer
    if (e != hipSuccess) {
        (void)hipFree(b->base);
        delete b;
        return ov2_set_err(c, OV2_ERR_HIP, "hipMemsetAsync: %s", hipGetErrorString(e));
    }
    if (hipEventCreateWithFlags(&b->ready_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&b->grad_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&b->free_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&b->free_ev2, hipEventDisableTiming) != hipSuccess) {
        (void)hipFree(b->base);
        delete b;
        return ov2_set_err(c, OV2_ERR_HIP, "hipEventCreate failed");
    }
    b->has_free_ev = false;
    b->has_free_ev2 = false;
    b->grad_built = false;
    v.base = b->base;
    *out = b;
    return OV2_OK;
}

}  // namespace

extern "C" ov2_status ov2_pyramid_build_images(ov2_ctx *c, const ov2_images *im, int win, int max_level,
                                               int use_clahe, float clip, int tiles_x, int tiles_y, ov2_pyr **out)
{
    if (!c || !im || !out) return OV2_ERR_INVALID;
    *out = nullptr;
    if (win < 3 || win > OV2_LM || max_level < 0 || max_level >= OV2_MAX_LEVELS)
        return ov2_set_err(c, OV2_ERR_INVALID, "win=%d (3..%d) max_level=%d (0..%d)", win, OV2_LM, max_level,
                           OV2_MAX_LEVELS - 1);
    if (im->w <= win || im->h <= win) return ov2_set_err(c, OV2_ERR_INVALID, "image smaller than window");
    if (use_clahe && (tiles_x < 1 || tiles_y < 1 || tiles_x > 64 || tiles_y > 64 || tiles_x > im->w || tiles_y > im->h))
        return ov2_set_err(c, OV2_ERR_INVALID, "CLAHE tiles %dx%d", tiles_x, tiles_y);
    OV2_HIP(c, hipSetDevice(c->device));
    ov2_pyr_buf *buf = nullptr;
    ov2_status s = acquire_buf(c, im->w, im->h, win, max_level, im->batch, &buf);
    if (s != OV2_OK) return s;
    const ov2_pyr_view &v = buf->view;
    const int B = im->batch;
    hipStream_t sp = c->stream_pyr;
    // a pooled buffer may still be read by kernels of the main stream (its last consumers): wait for their release mark
    if (buf->has_free_ev) OV2_HIP(c, hipStreamWaitEvent(sp, buf->free_ev, 0));
    if (buf->has_free_ev2) {   // readers another context enqueued on its own stream (ov2_pyr_release_from)
        OV2_HIP(c, hipStreamWaitEvent(sp, buf->free_ev2, 0));
        buf->has_free_ev2 = false;
    }

    float inv_tw = 0.f, inv_th = 0.f;
    if (use_clahe) {
        if (!buf->lut) {
            hipError_t e = hipMalloc((void **)&buf->lut, (size_t)B * 64 * 64 * 256);
            if (e != hipSuccess) {
                std::lock_guard<std::mutex> g(c->mu);
                c->pool.push_back(buf);
                return ov2_set_err(c, OV2_ERR_NOMEM, "CLAHE LUT hipMalloc");
            }
        }
        int ew = im->w, eh = im->h;
        if (im->w % tiles_x != 0 || im->h % tiles_y != 0) {  // CLAHE_Impl::apply: extend right/bottom
            ew = im->w + (tiles_x - im->w % tiles_x);
            eh = im->h + (tiles_y - im->h % tiles_y);
        }
        const int tw = ew / tiles_x, th = eh / tiles_y;
        const int total = tw * th;
        const float lut_scale = 255.0f / (float)total;
        int clip_limit = 0;
        if ((double)clip > 0.0) {
            clip_limit = (int)((double)clip * total / 256);
            if (clip_limit < 1) clip_limit = 1;
        }
        inv_tw = 1.0f / (float)tw;
        inv_th = 1.0f / (float)th;
        OV2_LAUNCH_ON(c, OV2_K_CLAHE_LUT, sp, clahe_lut_wave_kernel, dim3((tiles_x * tiles_y + 3) / 4, B), dim3(256), 0, sp, im->base, im->w,
                           im->h, im->stride, im->bstride, tiles_x, tiles_y, tw, th, clip_limit, lut_scale, buf->lut);
    }
    int ncx_max = 0, ncy_max = 0;
    size_t l0_lds = 0;
    if (use_clahe) {
        // upper bound of the LUT window of a 64 x 16 px tile; tw/th = 1 / inv
        const float tw = 1.0f / inv_tw, th = 1.0f / inv_th;
        ncx_max = (int)(64.0f / tw) + 3;
        ncy_max = (int)(64.0f / th) + 3;
        l0_lds = (size_t)ncx_max * ncy_max * 256;
    }
    // fused level 0 + first pyrDown: LDS = the staged tile + one 1-KB quad table per interpolation cell it can touch
    int first_down = 0;
    size_t fused_lds = 0;
    if (use_clahe && v.nlevels >= 2) {
        const float tw = 1.0f / inv_tw, th = 1.0f / inv_th;
        const int cells = ((int)((float)(PD_TW + 16) / tw) + 2) * ((int)((float)(PD_TH + 4) / th) + 2);
        fused_lds = ((size_t)PD_ROWS * PD_DW + (size_t)cells * 256) * 4;
    }
    static const bool no_fused = getenv("OV2_PYR_NO_FUSED") != nullptr;   // experiments: the two-kernel path
    if (use_clahe && v.nlevels >= 2 && fused_lds <= 64 * 1024 && im->w >= 8 && im->h >= 8 && !no_fused) {
        OV2_LAUNCH_ON(c, OV2_K_LEVEL0, sp, level0_clahe_pyrdown_kernel, dim3((im->w + PD_TW - 1) / PD_TW, (im->h + PD_TH - 1) / PD_TH, B),
                      dim3(256), fused_lds, sp, im->base, im->w, im->h, im->stride, im->bstride, buf->lut, tiles_x, tiles_y, inv_tw,
                      inv_th, v);
        first_down = 1;
    } else if (use_clahe && l0_lds <= 48 * 1024)
        OV2_LAUNCH_ON(c, OV2_K_LEVEL0, sp, level0_clahe_tiled_kernel, dim3((im->w + 63) / 64, (im->h + 63) / 64, B), dim3(256), l0_lds,
                   sp, im->base, im->w, im->h, im->stride, im->bstride, buf->lut, tiles_x, tiles_y, inv_tw, inv_th,
                   ncx_max, v);
    else {
        OV2_LAUNCH_ON(c, OV2_K_LEVEL0, sp, level0_kernel, dim3((im->w + 255) / 256, im->h, B), dim3(64), 0, sp, im->base, im->w,
                      im->h, im->stride, im->bstride, use_clahe, buf->lut, tiles_x, tiles_y, inv_tw, inv_th, v);
    }
    // pyrDown chain only: the gradient planes are written when a consumer asks for them (ov2_pyr_need_grad)
    { std::lock_guard<std::mutex> g(c->mu); buf->grad_built = false; }
    for (int l = first_down; l + 1 < v.nlevels; ++l) {
        const ov2_level_desc &L = v.lv[l];
        OV2_LAUNCH_ON(c, OV2_K_LEVEL, sp, pyrdown_kernel, dim3((L.w + PD_TW - 1) / PD_TW, (L.h + PD_TH - 1) / PD_TH, B), dim3(256), 0, sp,
                      v, l);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        std::lock_guard<std::mutex> g(c->mu);
        c->pool.push_back(buf);
        return ov2_set_err(c, OV2_ERR_HIP, "pyramid launch: %s", hipGetErrorString(e));
    }
    OV2_HIP(c, hipEventRecord(buf->ready_ev, sp));
    ov2_pyr *p = new ov2_pyr();
    p->refs.store(1);
    p->ctx = c;
    p->buf = buf;
    *out = p;
    return OV2_OK;
}

extern "C" ov2_status ov2_pyramid_build(ov2_ctx *c, const uint8_t *img, int w, int h, int stride, int win,
                                        int max_level, int use_clahe, float clip, int tiles_x, int tiles_y,
                                        ov2_pyr **out)
{
    if (!c || !img || !out || w <= 0 || h <= 0 || stride < w) return OV2_ERR_INVALID;
    if (c->tmp_img && (c->tmp_img->w != w || c->tmp_img->h != h)) {
        OV2_HIP(c, hipStreamSynchronize(c->stream));
        ov2_images_destroy(c->tmp_img);
        c->tmp_img = nullptr;
    }
    if (!c->tmp_img) {
        ov2_status s = ov2_images_create(c, 1, w, h, &c->tmp_img);
        if (s != OV2_OK) return s;
    }
    ov2_status s = ov2_images_upload(c, c->tmp_img, 0, img, stride);
    if (s != OV2_OK) return s;
    s = ov2_pyramid_build_images(c, c->tmp_img, win, max_level, use_clahe, clip, tiles_x, tiles_y, out);
    if (s != OV2_OK) return s;
    OV2_HIP(c, hipStreamSynchronize(c->stream_pyr));  // the staging image may be overwritten by the next call
    return OV2_OK;
}

extern "C" void ov2_pyr_retain(ov2_pyr *p)
{
    if (p) p->refs.fetch_add(1);
}

extern "C" void ov2_pyr_release(ov2_pyr *p)
{
    if (!p) return;
    if (p->refs.fetch_sub(1) == 1) {
        // stream-ordered reuse: later builds on the same ctx stream run after every kernel that reads this buffer
        // when those readers were enqueued on the same ctx.  Cross-ctx readers must synchronise before release.
        // every consumer of this pyramid was enqueued on the main stream before this release: mark that point
        if (hipEventRecord(p->buf->free_ev, p->ctx->stream) == hipSuccess) p->buf->has_free_ev = true;
        std::lock_guard<std::mutex> g(p->ctx->mu);
        p->ctx->pool.push_back(p->buf);
        delete p;
    }
}

// A consumer on ANOTHER context (the reference's mapper thread works on the keyframe's pyramid while the front-end
// moves on, src/mapper.cpp:76-97) enqueued its readers on `user`'s main stream: mark that point on the buffer, so
// the next build into it waits for them, then drop the reference.  One foreign consumer stream per pyramid.
extern "C" void ov2_pyr_release_from(ov2_ctx *user, ov2_pyr *p)
{
    if (!p) return;
    if (user && user != p->ctx) {
        (void)hipSetDevice(user->device);
        std::lock_guard<std::mutex> g(p->ctx->mu);
        if (hipEventRecord(p->buf->free_ev2, user->stream) == hipSuccess) p->buf->has_free_ev2 = true;
    }
    ov2_pyr_release(p);
}

extern "C" int ov2_pyr_batch(const ov2_pyr *p) { return p ? p->buf->batch : 0; }
extern "C" int ov2_pyr_nlevels(const ov2_pyr *p) { return p ? p->buf->view.nlevels : 0; }

extern "C" ov2_status ov2_pyr_level_size(const ov2_pyr *p, int level, int *w, int *h, int *pad)
{
    if (!p || level < 0 || level >= p->buf->view.nlevels) return OV2_ERR_INVALID;
    if (w) *w = p->buf->view.lv[level].w;
    if (h) *h = p->buf->view.lv[level].h;
    if (pad) *pad = p->buf->view.pad;
    return OV2_OK;
}

extern "C" ov2_status ov2_pyr_download_level(ov2_ctx *c, const ov2_pyr *p, int b, int level, uint8_t *img,
                                             int16_t *grad)
{
    if (!c || !p || b < 0 || b >= p->buf->batch || level < 0 || level >= p->buf->view.nlevels) return OV2_ERR_INVALID;
    const ov2_pyr_view &v = p->buf->view;
    const ov2_level_desc &L = v.lv[level];
    const int pw = L.w + 2 * v.pad, ph = L.h + 2 * v.pad;
    const int x_off = OV2_LM - v.pad;
    if (grad) {
        const ov2_status gs = ov2_pyr_need_grad(c, p);
        if (gs != OV2_OK) return gs;
    }
    OV2_HIP(c, hipStreamSynchronize(c->stream_pyr));
    OV2_HIP(c, hipStreamSynchronize(c->stream));
    if (img)
        OV2_HIP(c, hipMemcpy2D(img, pw, v.base + L.img_off + L.img_bstride * b + x_off, L.istride, pw, ph,
                               hipMemcpyDeviceToHost));
    if (grad)
        OV2_HIP(c, hipMemcpy2D(grad, (size_t)pw * 4, v.base + L.grad_off + L.grad_bstride * b + (size_t)x_off * 4,
                               (size_t)L.gstride * 4, (size_t)pw * 4, ph, hipMemcpyDeviceToHost));
    return OV2_OK;
}

// Scharr planes of every level, written once per pyramid by the first consumer that needs them (the 8 / 16-lane KLT
// kernels, ov2_pyr_download_level) on ITS main stream, behind the build; later consumers wait for that point.
ov2_status ov2_pyr_need_grad(ov2_ctx *c, const ov2_pyr *p)
{
    if (!c || !p) return OV2_ERR_INVALID;
    ov2_pyr_buf *buf = p->buf;
    // The HIP calls run under the owning context's mutex (two consumer contexts must not both write the planes), so their
    // failures are collected and reported AFTER the guard is gone: ov2_set_err locks the same mutex when c == p->ctx.
    hipError_t err = hipSuccess;
    const char *what = "";
    {
        std::lock_guard<std::mutex> g(p->ctx->mu);
        if (buf->grad_built) {
            err = hipStreamWaitEvent(c->stream, buf->grad_ev, 0); what = "hipStreamWaitEvent(grad_ev)";
        } else if ((err = hipStreamWaitEvent(c->stream, buf->ready_ev, 0)) != hipSuccess) {
            what = "hipStreamWaitEvent(ready_ev)";
        } else {
            const ov2_pyr_view &v = buf->view;
            for (int l = 0; l < v.nlevels; ++l) {
                const ov2_level_desc &L = v.lv[l];
                OV2_LAUNCH(c, OV2_K_LEVEL, level_kernel, dim3((L.w + TILE_W - 1) / TILE_W, (L.h + TILE_H - 1) / TILE_H, buf->batch), dim3(256), 0,
                           c->stream, v, l, 0, 1);
            }
            if ((err = hipGetLastError()) != hipSuccess) what = "level_kernel launch";
            else if ((err = hipEventRecord(buf->grad_ev, c->stream)) != hipSuccess) what = "hipEventRecord(grad_ev)";
            else buf->grad_built = true;
        }
    }
    if (err != hipSuccess) return ov2_set_err(c, OV2_ERR_HIP, "%s failed: %s", what, hipGetErrorString(err));
    return OV2_OK;
}

ov2_status ov2_pyr_wait_ready(ov2_ctx *c, const ov2_pyr *p)
{
    if (!p) return OV2_ERR_INVALID;
    OV2_HIP(c, hipStreamWaitEvent(c->stream, p->buf->ready_ev, 0));
    return OV2_OK;
}
