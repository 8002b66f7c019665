// klt.hip -- forward-backward pyramidal Lucas-Kanade, EIGHT keypoints per 64-lane wavefront (gfx950).
//
// Replaces (reference, /root/reference): FeatureTracker::fbKltTracking src/feature_tracker.cpp:35-137
// (= 2x cv::calcOpticalFlowPyrLK + the status / err / inBorder / forward-backward gates) and the two-stage
// batching of VisualFrontEnd::kltTracking src/visual_front_end.cpp:132-275.
// Arithmetic: OpenCV LKTrackerInvoker semantics as restated in oracle/ov2_oracle_fe.c -- 14-bit bilinear
// weights (v_rndne == cvRound), CV_DESCALE fixed point, EXACT integer sums for A11/A12/A22/b1/b2 converted to
// fp32 once (order independent, so the lane reduction is bit-identical to the oracle's scalar loop), fp32
// 2x2 solve with contraction off, fp64 for the two comparisons OpenCV does in double.
//
// Mapping: one keypoint per group of 8 or 16 lanes (8 or 4 keypoints per wave), lane = window column walking the win
// rows of its column.
//   - every "scalar" of the LK update is per-group vector math, so the keypoints of a wave share each instruction;
//   - the windows (template I + its Scharr gradients, search window J) are fetched row-wise with 16-byte loads into LDS
//     and read from there as pixel pairs: the kernel was bound by the texture addresser when every lane gathered two
//     bytes per row (see klt_lds below);
//   - sums over the window = DPP reduction (quad_perm, row_half_mirror, row_mirror), no readlane;
//   - template (I, Ix, Iy per window pixel) stays in registers for the whole level, two pixels per register;
//   - bilinear taps as v_dot2_i32_i16 on v_perm_b32-spread pixel pairs.
// The whole pyramid loop, the gates and the backward pass run inside one launch; no host round trip.
#include <cstdlib>

#include "ov2_internal.h"

namespace {

struct klt_params {
    int win, nlevels, max_iter;
    double eps2;     // criteria.epsilon^2 (double, as in calcOpticalFlowPyrLK)
    float err_th;    // nklt_err
    double fb_th;    // fmax_fbklt_dist promoted to double for the cv::norm comparison
    float min_eig_thr;
    int y_pickup;            // k > 0: every k-th wave of the first launch takes one batch of stragglers itself; 0: none does
    int y_after, y_groups;   // three-lane kernels: a level pass hands its last <= y_groups running keypoints over to the resume
                             // launch once it has made y_after iterations (0 = never); see klt_rec
    int rule33;      // 1: VisualFrontEnd::kltTracking's "< 33 % good => drop every re-queued prior" (src/visual_front_end.cpp:228-233);
                     // 0: MapManager::stereoMatching re-queues the failures with the updated prior, no such rule (src/map_manager.cpp:533-537)
};

#define W_BITS 14

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}

// sum over the GL (8 or 16) lanes of a keypoint's lane group, result in every lane of the group
template <int GL>
__device__ __forceinline__ int row_sum_i32(int v)
{
    v += dpp_i32<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_i32<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_i32<0x141>(v);   // row_half_mirror: lane l <-> 7 - l of its half row
    if (GL == 16) v += dpp_i32<0x140>(v);   // row_mirror
    return v;
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64k(double v)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | lo);
}

// exact: the addends are integers below 2^32, every partial sum is an integer below 2^53
template <int GL>
__device__ __forceinline__ double row_sum_f64(double v)
{
    v += dpp_f64k<0xB1>(v);
    v += dpp_f64k<0x4E>(v);
    v += dpp_f64k<0x141>(v);
    if (GL == 16) v += dpp_f64k<0x140>(v);
    return v;
}

__device__ __forceinline__ void lk_weights(float a, float b, int &w00, int &w01, int &w10, int &w11)
{
    w00 = (int)__builtin_rintf((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
    w01 = (int)__builtin_rintf(a * (1.f - b) * (float)(1 << W_BITS));
    w10 = (int)__builtin_rintf((1.f - a) * b * (float)(1 << W_BITS));
    w11 = (1 << W_BITS) - w00 - w01 - w10;
}

__device__ __forceinline__ unsigned ld_u16(const unsigned char *p)
{
    unsigned short r;
    __builtin_memcpy(&r, p, 2);   // unaligned 16-bit load: (x, x+1) of one row
    return r;
}

// 16 bytes from a dword-aligned address: one global_load_dwordx4
__device__ __forceinline__ uint4 ld_b128(const unsigned char *p)
{
    return *reinterpret_cast<const uint4 *>(__builtin_assume_aligned(p, 4));
}

// LDS staging of one keypoint's windows.  The kernel is bound by the texture addresser, not by VALU or HBM (TA busy
// 80-90 % of the launch with one 2-byte gather per lane, row and iteration): the rows of a window are now fetched
// with 16-byte loads (one lane per row: (WIN+1) x 16 B cover the WIN+1 columns a bilinear window touches) into LDS,
// and the lanes read their column's pixel pairs from there.  Per group: two u8 windows (template I, search J) of
// (WIN+1) rows x 16 B and one gradient window of (WIN+1) rows x GSEG x 16 B; +16 B so that the groups of a wave start
// in different banks.
template <int WIN>
struct klt_lds {
    static constexpr int ROWS = WIN + 1;
    static constexpr int GSEG = ((WIN + 1) * 4 + 15) / 16;
    static constexpr int WB = ROWS * 16 + 16;
    static constexpr int GB = ROWS * GSEG * 16 + 16;
};

// rows r = sub, sub + GL, ... of a (WIN+1) x 16 B window.  src is DWORD-ALIGNED (the caller rounds the window's first
// column down to a multiple of 4 and keeps the remainder as the read offset: (WIN+1) + 3 <= 16 bytes): a byte-aligned
// 16-byte load is legal but is split by the memory pipeline (measured: 345 us instead of 204 us for the launch).
#ifndef KLT_STAGE_DWORD
#define KLT_STAGE_DWORD 0
#endif
template <int WIN, int GL>
__device__ __forceinline__ void stage_u8(const unsigned char *src, int stride, unsigned char *lw, int sub)
{
#if KLT_STAGE_DWORD
    // four lanes per row, one dword each: the lanes of a quad share a cache line
    constexpr int NE4 = (WIN + 1) * 4, NK4 = (NE4 + GL - 1) / GL;
#pragma unroll
    for (int k = 0; k < NK4; ++k) {
        const int e = sub + GL * k;
        if (e < NE4)
            *reinterpret_cast<unsigned *>(lw + e * 4) =
                *reinterpret_cast<const unsigned *>(src + (size_t)(e >> 2) * stride + (e & 3) * 4);
    }
    return;
#endif
    constexpr int NK = (WIN + 1 + GL - 1) / GL;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int r = sub + GL * k;
        if (r < WIN + 1) *reinterpret_cast<uint4 *>(lw + r * 16) = ld_b128(src + (size_t)r * stride);
    }
}

// 16-byte segments s = sub, sub + GL, ... of the (WIN+1) x GSEG gradient window ((Ix,Iy) int16 pairs, 4 B per pixel)
template <int WIN, int GL>
__device__ __forceinline__ void stage_grad(const int *src, int gstride, unsigned char *lg, int sub)
{
    constexpr int GSEG = klt_lds<WIN>::GSEG, NS = (WIN + 1) * GSEG, NK = (NS + GL - 1) / GL;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int sg = sub + GL * k;
        if (sg < NS) {
            const int row = sg / GSEG, part = sg - row * GSEG;
            *reinterpret_cast<uint4 *>(lg + sg * 16) =
                ld_b128(reinterpret_cast<const unsigned char *>(src + (size_t)row * gstride) + part * 16);
        }
    }
}

// Pixel pairs (p[x], p[x+1]) from a staged window with ALIGNED LDS reads: the dword pair around the column + v_alignbyte
// (an odd-address ds_read_u16 is legal but slow).  The column's dword and byte shift are the same for every row of the
// window (rows are 16 B apart), so they are formed once and the rows become immediate offsets.
struct lds_col {
    const unsigned *q;   // dword holding the column's first byte, row 0
    unsigned sh;         // byte shift inside it
};
__device__ __forceinline__ lds_col lds_col_of(const unsigned char *row0 /* 16-byte aligned */, int byte_off)
{
    lds_col c;
    c.q = reinterpret_cast<const unsigned *>(row0 + (byte_off & ~3));
    c.sh = (unsigned)byte_off & 3u;
    return c;
}
__device__ __forceinline__ unsigned lds_pair(const lds_col &c, int row)
{
    const unsigned *q = c.q + 4 * row;
    return __builtin_amdgcn_alignbyte(q[1], q[0], c.sh);
}

typedef short ov2_s16x2 __attribute__((ext_vector_type(2)));

// v_dot2_i32_i16: a.lo * b.lo + a.hi * b.hi + c (signed 16-bit halves, 32-bit accumulate, no clamp)
__device__ __forceinline__ int dot2(unsigned a, unsigned b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(ov2_s16x2, a), __builtin_bit_cast(ov2_s16x2, b), c, false);
}
// first tap of a chain, accumulator = a rounding CONSTANT: the VOP3P form reads it from an SGPR; the two-address
// v_dot2c the compiler prefers would need a v_mov of the constant per chain (13 per LK iteration)
__device__ __forceinline__ int dot2k(unsigned a, unsigned b, int k)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
    return r;
}
// v_perm_b32 byte shuffles: (byte0, byte1) of a two-pixel load -> two zero-extended 16-bit halves
__device__ __forceinline__ unsigned spread_u8x2(unsigned t) { return __builtin_amdgcn_perm(0u, t, 0x0c010c00u); }
// (a.lo16, b.lo16) and (a.hi16, b.hi16)
__device__ __forceinline__ unsigned pack_lo16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
__device__ __forceinline__ unsigned pack_hi16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
__device__ __forceinline__ unsigned pk_sub16(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, (ov2_s16x2)(__builtin_bit_cast(ov2_s16x2, a) - __builtin_bit_cast(ov2_s16x2, b)));
}

typedef unsigned short ov2_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_add16(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, (ov2_u16x2)(__builtin_bit_cast(ov2_u16x2, a) + __builtin_bit_cast(ov2_u16x2, b)));
}
__device__ __forceinline__ unsigned pk_mul16(unsigned a, unsigned k)   // both halves times k (low 16 bits)
{
    return __builtin_bit_cast(unsigned, (ov2_u16x2)(__builtin_bit_cast(ov2_u16x2, a) * (ov2_u16x2)((unsigned short)k)));
}
__device__ __forceinline__ unsigned pk_mad16(unsigned a, unsigned k, unsigned c)   // a * k + c per half
{
    return __builtin_bit_cast(unsigned, (ov2_u16x2)(__builtin_bit_cast(ov2_u16x2, a) * (ov2_u16x2)((unsigned short)k) + __builtin_bit_cast(ov2_u16x2, c)));
}

// (acc >> SH) written into the HIGH 16 bits of `keep`, whose low half stays (SDWA destination select): shift and pack of the
// second value of a 16-bit pair in one instruction
template <int SH>
__device__ __forceinline__ unsigned shr_pack_hi(unsigned keep, unsigned acc)
{
    asm("v_lshrrev_b32_sdwa %0, %2, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(keep) : "v"(acc), "n"(SH));
    return keep;
}
template <int SH>
__device__ __forceinline__ unsigned ashr_pack_hi(unsigned keep, int acc)
{
    asm("v_ashrrev_i32_sdwa %0, %2, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(keep) : "v"(acc), "n"(SH));
    return keep;
}

struct level_ptrs {
    const unsigned char *img;
    const int *grad;   // (Ix,Iy) int16 pair per pixel
    int istride, gstride, w, h, rows;
    const unsigned char *base;   // the pyramid allocation (wave-uniform) and the planes' byte offsets inside it: the three-lane
    unsigned img_o, grad_o;      // path addresses memory as uniform base + 32-bit lane offset (one VGPR per address)
};

__device__ __forceinline__ level_ptrs level_of(const ov2_pyr_view &v, int l, int b)
{
    level_ptrs p;
    const ov2_level_desc &L = v.lv[l];
    p.img = v.base + L.img_off + L.img_bstride * b;
    p.grad = reinterpret_cast<const int *>(v.base + L.grad_off + L.grad_bstride * b);
    p.istride = L.istride; p.gstride = L.gstride; p.w = L.w; p.h = L.h; p.rows = L.rows;
    p.base = v.base; p.img_o = (unsigned)(L.img_off + L.img_bstride * b); p.grad_o = (unsigned)(L.grad_off + L.grad_bstride * b);
    return p;
}

// One LKTrackerInvoker pass (one pyramid level) for the keypoint of this lane group (GL = 8 or 16 lanes).  `run`
// (group-uniform) says whether the group takes part; everything below is group-uniform except the lane's share of the
// window.  Lane `sub` < OC owns window column `sub` and walks its WIN rows (row y+1's load is row y's lower
// neighbours).  With GL = 8 a 9- or 11-wide window has EC = WIN - 8 columns left: their EC*WIN pixels are dealt out one
// per lane and round (pixel e = sub + GL*round), each with its own four taps.  Returns iterations executed.
template <int WIN, int GL>
__device__ __forceinline__ int lk_level(const level_ptrs &I, const level_ptrs &J, int pad, int level, int max_level,
                                        bool run, float kx, float ky, float &nx_io, float &ny_io, int &status,
                                        float &err, const klt_params &P, int sub, unsigned &passes,
                                        unsigned char *lwI, unsigned char *lwJ, unsigned char *lg)
{
    constexpr int GROW = klt_lds<WIN>::GSEG * 4;   // ints per row of the staged gradient window
    constexpr int OC = WIN < GL ? WIN : GL;        // columns with an owner lane
    constexpr int NE = (WIN - OC) * WIN;           // pixels of the remaining columns
    constexpr int NR = (NE + GL - 1) / GL;         // rounds to deal them out
    constexpr int NRP = (NR + 1) / 2;
    constexpr int NP = (WIN + 1) / 2;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    const float half = (float)(WIN - 1) * 0.5f;
    const float lscale = __builtin_ldexpf(1.f, -level);   // 1.f / (1 << level), exactly; one v_ldexp_f32 instead of an IEEE division
    float px = kx * lscale, py = ky * lscale;
    float nx, ny;
    if (level == max_level) { nx = nx_io * lscale; ny = ny_io * lscale; }
    else { nx = nx_io * 2.f; ny = ny_io * 2.f; }
    if (run) { nx_io = nx; ny_io = ny; }

    px -= half; py -= half;
    const int ipx = (int)floorf(px), ipy = (int)floorf(py);
    if (run && (ipx < -WIN || ipx >= I.w || ipy < -WIN || ipy >= I.h)) {
        if (level == 0) { status = 0; err = 0.f; }
        run = false;
    }
    if (!__any(run)) return 0;
    if (run) passes += 1u;
    int w00, w01, w10, w11;
    lk_weights(px - (float)ipx, py - (float)ipy, w00, w01, w10, w11);

    // template (I, Ix, Iy per window pixel), two pixels per register as int16 pairs; idle lanes / groups hold zero
    // gradients, so whatever they compute below drops out of the sums
    const bool col = sub < OC;
    const int sc = col ? sub : 0;
    unsigned Iv2[NP], Ix2[NP], Iy2[NP];
    unsigned IvE[NRP > 0 ? NRP : 1], IxE[NRP > 0 ? NRP : 1], IyE[NRP > 0 ? NRP : 1];
    int eoff[NR > 0 ? NR : 1];   // byte offset of the lane's round-r pixel inside a u8 plane (rows * stride + column)
    int sA11 = 0, sA12 = 0, sA22 = 0;
    {
        // bilinear taps as two v_dot2_i32_i16: (p[x], p[x+1]) . (w00, w01) + (p[x], p[x+1])' . (w10, w11); the weights
        // fit int16 (w11 can be -1 after rounding), pixels and Scharr gradients too
        const unsigned W01 = pack_lo16((unsigned)w00, (unsigned)w01), W23 = pack_lo16((unsigned)w10, (unsigned)w11);
        const int bx0 = run ? (OV2_LM + ipx) : OV2_LM, by = run ? (ipy + pad) : pad;
        // stage the template windows and, with them, the search window of the first iteration (its position is known)
        {
            const float sx = nx - half, sy = ny - half;
            const int inx0 = (int)floorf(sx), iny0 = (int)floorf(sy);
            const bool in0 = run && !(inx0 < -WIN || inx0 >= J.w || iny0 < -WIN || iny0 >= J.h);
            __syncthreads();   // the previous pass may still read these buffers
            const int jx0 = OV2_LM + (in0 ? inx0 : 0);
            stage_u8<WIN, GL>(I.img + (size_t)by * I.istride + (bx0 & ~3), I.istride, lwI, sub);
            stage_grad<WIN, GL>(I.grad + (size_t)by * I.gstride + bx0, I.gstride, lg, sub);
            stage_u8<WIN, GL>(J.img + (size_t)((in0 ? iny0 : 0) + pad) * J.istride + (jx0 & ~3), J.istride, lwJ, sub);
            __syncthreads();
        }
        const lds_col ci = lds_col_of(lwI, (bx0 & 3) + sc);
        const int *gp = reinterpret_cast<const int *>(lg) + sc;
        unsigned T = spread_u8x2(lds_pair(ci, 0));
        unsigned GX, GY;
        {
            const unsigned g0 = (unsigned)gp[0], g1 = (unsigned)gp[1];
            GX = pack_lo16(g0, g1); GY = pack_hi16(g0, g1);
        }
        const bool on = run && col;
        unsigned piv = 0, pix = 0, piy = 0;
#pragma unroll
        for (int y = 0; y < WIN; ++y) {
            gp += GROW;
            const unsigned B = spread_u8x2(lds_pair(ci, y + 1));
            const unsigned h0 = (unsigned)gp[0], h1 = (unsigned)gp[1];
            const unsigned HX = pack_lo16(h0, h1), HY = pack_hi16(h0, h1);
            const unsigned iv = (unsigned)dot2(B, W23, dot2k(T, W01, 1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
            int ix = dot2(HX, W23, dot2k(GX, W01, 1 << (W_BITS - 1))) >> W_BITS;
            int iy = dot2(HY, W23, dot2k(GY, W01, 1 << (W_BITS - 1))) >> W_BITS;
            if (y & 1) {
                Iv2[y >> 1] = pack_lo16(piv, iv); Ix2[y >> 1] = pack_lo16(pix, (unsigned)ix); Iy2[y >> 1] = pack_lo16(piy, (unsigned)iy);
            } else if (y == WIN - 1) {   // odd window: the phantom last row has zero gradients
                Iv2[y >> 1] = iv; Ix2[y >> 1] = (unsigned)ix & 0xffffu; Iy2[y >> 1] = (unsigned)iy & 0xffffu;
            } else {
                piv = iv; pix = (unsigned)ix; piy = (unsigned)iy;
            }
            T = B; GX = HX; GY = HY;
        }
        // idle lanes / groups hold zero gradients; Hessian sums from the packed pairs: one dot2 per row PAIR and term
        // (|Ix|,|Iy| <= 4080: two products stay below 2^26)
        {
            const unsigned onm = on ? 0xffffffffu : 0u;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                Ix2[q] &= onm; Iy2[q] &= onm;
                sA11 = dot2(Ix2[q], Ix2[q], sA11); sA12 = dot2(Ix2[q], Iy2[q], sA12); sA22 = dot2(Iy2[q], Iy2[q], sA22);
            }
        }
        // pixels of the columns without an owner lane
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int e = sub + GL * r;
            const bool ev = e < NE;
            const int ec = ev ? e : 0;
            const int ecol = OC + ec / WIN, erow = ec - (ec / WIN) * WIN;
            eoff[r] = erow * 16 + ecol;
            const lds_col ce = lds_col_of(lwI + erow * 16, (bx0 & 3) + ecol);
            const int *gpe = reinterpret_cast<const int *>(lg) + erow * GROW + ecol;
            const unsigned T0 = spread_u8x2(lds_pair(ce, 0)), T1 = spread_u8x2(lds_pair(ce, 1));
            const unsigned g0 = (unsigned)gpe[0], g1 = (unsigned)gpe[1];
            const unsigned h0 = (unsigned)gpe[GROW], h1 = (unsigned)gpe[GROW + 1];
            const unsigned iv = (unsigned)dot2(T1, W23, dot2k(T0, W01, 1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
            int ix = dot2(pack_lo16(h0, h1), W23, dot2k(pack_lo16(g0, g1), W01, 1 << (W_BITS - 1))) >> W_BITS;
            int iy = dot2(pack_hi16(h0, h1), W23, dot2k(pack_hi16(g0, g1), W01, 1 << (W_BITS - 1))) >> W_BITS;
            if (!(run && ev)) { ix = 0; iy = 0; }
            sA11 += __mul24(ix, ix); sA12 += __mul24(ix, iy); sA22 += __mul24(iy, iy);
            if (r & 1) {
                IvE[r >> 1] = pack_lo16(piv, iv); IxE[r >> 1] = pack_lo16(pix, (unsigned)ix); IyE[r >> 1] = pack_lo16(piy, (unsigned)iy);
            } else if (r == NR - 1) {
                IvE[r >> 1] = iv; IxE[r >> 1] = (unsigned)ix & 0xffffu; IyE[r >> 1] = (unsigned)iy & 0xffffu;
            } else {
                piv = iv; pix = (unsigned)ix; piy = (unsigned)iy;
            }
        }
    }
    // |Ix|,|Iy| <= 4080 for u8 images: WIN*WIN <= 121 products stay below 2^31 -> exact in int32
    const float A11 = (float)(double)row_sum_i32<GL>(sA11) * FLT_SCALE;
    const float A12 = (float)(double)row_sum_i32<GL>(sA12) * FLT_SCALE;
    const float A22 = (float)(double)row_sum_i32<GL>(sA22) * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    const float min_eig = __fdiv_rn(A22 + A11 - __fsqrt_rn((A11 - A22) * (A11 - A22) + 4.f * A12 * A12),
                                    (float)(2 * WIN * WIN));
    if (run) {
        err = min_eig;  // OPTFLOW_LK_GET_MIN_EIGENVALS
        if (min_eig < P.min_eig_thr || D < 1.1920929e-07f /* FLT_EPSILON */) {
            if (level == 0) status = 0;
            run = false;
        }
    }
    D = __fdiv_rn(1.f, D);

    nx -= half; ny -= half;
    float pdx = 0.f, pdy = 0.f;
    int iters = 0;
    for (int j = 0; j < P.max_iter; ++j) {
        if (!__any(run)) break;
        const int inx = (int)floorf(nx), iny = (int)floorf(ny);
        if (run && (inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h)) {
            if (level == 0) status = 0;
            run = false;
        }
        lk_weights(nx - (float)inx, ny - (float)iny, w00, w01, w10, w11);
        if (j > 0) {   // iteration 0's window came with the template
            __syncthreads();
            stage_u8<WIN, GL>(J.img + (size_t)((run ? iny : 0) + pad) * J.istride + ((OV2_LM + (run ? inx : 0)) & ~3), J.istride, lwJ, sub);
            __syncthreads();
        }
        const int jo = (run ? inx : 0) & 3;   // OV2_LM is a multiple of 4
        const lds_col cj = lds_col_of(lwJ, jo + sc);
        const unsigned W01 = pack_lo16((unsigned)w00, (unsigned)w01), W23 = pack_lo16((unsigned)w10, (unsigned)w11);
        int pb1 = 0, pb2 = 0;
        unsigned T = spread_u8x2(lds_pair(cj, 0));
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            unsigned B = spread_u8x2(lds_pair(cj, 2 * q + 1));
            // taps + rounding >= 1 (w11 >= -1), so the logical shift is the arithmetic one
            const unsigned j0 = (unsigned)dot2(B, W23, dot2k(T, W01, 1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
            T = B;
            unsigned j1 = 0;
            if (2 * q + 1 < WIN) {
                B = spread_u8x2(lds_pair(cj, 2 * q + 2));
                j1 = (unsigned)dot2(B, W23, dot2k(T, W01, 1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
                T = B;
            }
            // |jv - I| <= 8160 fits int16: v_pk_sub_i16 on the row pair, then one dot2 per gradient component
            const unsigned d2 = pk_sub16(j0 | (j1 << 16), Iv2[q]);
            pb1 = dot2(d2, Ix2[q], pb1);   // |diff| <= 8160, |Ix| <= 4080
            pb2 = dot2(d2, Iy2[q], pb2);
        }
#pragma unroll
        for (int q = 0; q < NRP; ++q) {   // the lane's pixels of the ownerless columns, two rounds per register
            const lds_col pe = lds_col_of(lwJ + (eoff[2 * q] & ~15), jo + (eoff[2 * q] & 15));
            const unsigned j0 = (unsigned)dot2(spread_u8x2(lds_pair(pe, 1)), W23,
                                               dot2k(spread_u8x2(lds_pair(pe, 0)), W01, 1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
            unsigned j1 = 0;
            if (2 * q + 1 < NR) {
                const lds_col pf = lds_col_of(lwJ + (eoff[2 * q + 1] & ~15), jo + (eoff[2 * q + 1] & 15));
                j1 = (unsigned)dot2(spread_u8x2(lds_pair(pf, 1)), W23,
                                    dot2k(spread_u8x2(lds_pair(pf, 0)), W01, 1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
            }
            const unsigned d2 = pk_sub16(j0 | (j1 << 16), IvE[q]);
            pb1 = dot2(d2, IxE[q], pb1);
            pb2 = dot2(d2, IyE[q], pb2);
        }
        // a lane holds <= 16 products of <= 3.4e7: fits int32; the group total may not, so sum exactly in f64.
        // Usual case (wave-uniform test): every lane's partial is below 2^27, the group total fits int32 and
        // v_cvt_f32_i32 rounds it once, exactly like (float)(double)total -- DPP-fused integer adds instead of DPP
        // moves + f64 adds.
        float b1, b2;
        if (__all((unsigned)(pb1 + (1 << 27)) < (1u << 28) && (unsigned)(pb2 + (1 << 27)) < (1u << 28))) {
            b1 = (float)row_sum_i32<GL>(pb1) * FLT_SCALE;
            b2 = (float)row_sum_i32<GL>(pb2) * FLT_SCALE;
        } else {
            b1 = (float)row_sum_f64<GL>((double)pb1) * FLT_SCALE;
            b2 = (float)row_sum_f64<GL>((double)pb2) * FLT_SCALE;
        }
        const float dx = (A12 * b2 - A22 * b1) * D;
        const float dy = (A12 * b1 - A11 * b2) * D;
        if (run) {
            ++iters;
            nx += dx; ny += dy;
            nx_io = nx + half; ny_io = ny + half;
            if ((double)dx * dx + (double)dy * dy <= P.eps2) run = false;
            // std::abs(float) < 0.01 (double literal)  <=>  <= 0.01f
            else if (j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) {
                nx_io -= dx * 0.5f; ny_io -= dy * 0.5f;
                run = false;
            }
            pdx = dx; pdy = dy;
        }
    }
    return iters;
}

// ---- three lanes per keypoint (9 x 9 windows) ---------------------------------------------------------------------
// The 8-lane mapping above spends most of a level pass on per-keypoint arithmetic that every lane of the group repeats
// (positions, weights, 2 x 2 solve, convergence tests: ~70 of the ~190 vector instructions of an LK iteration, and
// more around the template).  Here a keypoint takes THREE lanes, each owning three adjacent columns of the 9 x 9 window,
// so a wave carries 20 keypoints (five groups per 16-lane DPP row, lane 15 of every row idles) and that overhead is
// shared by 2.5 x as many keypoints:
//   - one aligned dword pair + v_alignbyte yields the four pixels (x .. x+3) a lane's three columns touch in a row;
//   - the template rows (I: 8 B, gradient: 16 B per lane and row) go from global memory straight to registers -- no
//     LDS for them;
//   - the SEARCH image is staged once per level pass as a 14-row x 48-byte region around the start position
//     (margin 2 rows / >= 11 columns): the later iterations of the pass read it from LDS without another trip to
//     memory, and a window that leaves the region re-stages it (rare);
//   - sums over the three lanes: two DPP row shifts + two selects.
// Arithmetic per window pixel is the one of lk_level (same taps, exact integer sums), so results are bit-identical.
#ifndef KLT3_RROWS
#define KLT3_RROWS 14
#endif
#ifndef KLT3_RBYTES
#define KLT3_RBYTES 48
#endif
struct klt3 {
    // Region rows are 48 bytes (one 16-byte piece per lane: a load instruction touches ONE image row per keypoint; the
    // vector memory front end spends ~2 cycles per distinct cache line of an instruction, scripts/micro/ta_bench.hip) or
    // 32 bytes (pieces dealt out 3 per instruction over the 2 x RROWS pieces: smaller LDS block and fewer registers in flight).
    static constexpr int RROWS = KLT3_RROWS, RBYTES = KLT3_RBYTES;
    static constexpr int RSZ = RROWS * RBYTES + 16;   // +16: the regions of a wave start in different banks
    static constexpr int NLD = RBYTES == 48 ? RROWS : (2 * RROWS + 2) / 3;   // 16-byte loads per lane
    static constexpr int ROW_SPAN = RROWS - 10, ROW_MARGIN = ROW_SPAN / 2;   // the window's first row may sit ROW_SPAN rows into the region
    static constexpr int COL_ALIGN = RBYTES == 48 ? 16 : 4;
    static constexpr int COL_LEAD = RBYTES == 48 ? 12 : 9, COL_SPAN = RBYTES == 48 ? 38 : 20;   // ... and its first column 0 .. COL_SPAN bytes
};

// lane -> (keypoint slot of the wave, lane of the group); GL = 8 / 16: power-of-two groups, GL = 3: see above
template <int GL>
struct klt_map {
    static constexpr int KPW = 64 / GL;
    static __device__ __forceinline__ int slot(int tid) { return tid / GL; }
    static __device__ __forceinline__ int sub(int tid) { return tid & (GL - 1); }
    static __device__ __forceinline__ bool lane_ok(int) { return true; }
};
template <>
struct klt_map<3> {
    static constexpr int KPW = 20;
    static __device__ __forceinline__ int slot(int tid) { return (tid >> 4) * 5 + min(((tid & 15) * 11) >> 5, 4); }   // (t * 11) >> 5 == t / 3 for t < 16
    static __device__ __forceinline__ int sub(int tid) { const int t = tid & 15; return t - 3 * ((t * 11) >> 5); }
    static __device__ __forceinline__ bool lane_ok(int tid) { return (tid & 15) != 15; }
};

template <int WIN, int GL>
struct klt_smem {   // LDS bytes of one wave
    static constexpr int BYTES = (64 / GL) * (2 * klt_lds<WIN>::WB + klt_lds<WIN>::GB);
};
template <int WIN>
struct klt_smem<WIN, 3> {
    static constexpr int BYTES = klt_map<3>::KPW * klt3::RSZ;
};

// sum over the three lanes of a group, result in all three (c = lane of the group)
__device__ __forceinline__ int sum3_i32(int v, int c)
{
    int t = v + dpp_i32<0x111>(v);   // row_shr:1  lane l reads l - 1
    t += dpp_i32<0x112>(v);          // row_shr:2  complete in lane c == 2
    const int u1 = dpp_i32<0x101>(t), u2 = dpp_i32<0x102>(t);   // row_shl:1 / 2: lane l reads l + 1 / l + 2
    return c == 2 ? t : (c == 1 ? u1 : u2);
}
__device__ __forceinline__ double sum3_f64(double v, int c)
{
    double t = v + dpp_f64k<0x111>(v);
    t += dpp_f64k<0x112>(v);
    const double u1 = dpp_f64k<0x101>(t), u2 = dpp_f64k<0x102>(t);
    return c == 2 ? t : (c == 1 ? u1 : u2);
}

// (byte k, byte k + 1) of a dword as two zero-extended 16-bit halves
template <int K>
__device__ __forceinline__ unsigned spread_pair(unsigned t)
{
    return __builtin_amdgcn_perm(0u, t, 0x0c000c00u + (unsigned)K * 0x00010001u + 0x00010000u);
}

// origin of the staged search region (padded row, byte column) for a window at (inx, iny)
__device__ __forceinline__ void klt3_region_origin(const level_ptrs &J, int pad, int inx, int iny, int &rr0, int &cc0)
{
    rr0 = min(max(iny + pad - klt3::ROW_MARGIN, 0), J.rows - klt3::RROWS);
    cc0 = min(max((OV2_LM + inx - klt3::COL_LEAD) & ~(klt3::COL_ALIGN - 1), 0), J.istride - klt3::RBYTES);
}

// the group's three lanes fetch the region: 48-byte rows: lane c takes bytes 16c .. 16c + 15 of every row; 32-byte rows:
// the 2 x RROWS pieces are dealt out three per instruction (piece s = 3t + c: row s / 2, half s % 2)
struct klt3_segs { uint4 v[klt3::NLD]; };
__device__ __forceinline__ void klt3_load_region(const level_ptrs &J, int rr0, int cc0, int c, klt3_segs &S)
{
    const unsigned o = J.img_o + (unsigned)(rr0 * J.istride + cc0);
#pragma unroll
    for (int t = 0; t < klt3::NLD; ++t) {
        unsigned a;
        if (klt3::RBYTES == 48) a = o + (unsigned)(t * J.istride + 16 * c);
        else {
            const int sg = min(3 * t + c, 2 * klt3::RROWS - 1);
            a = o + (unsigned)((sg >> 1) * J.istride + (sg & 1) * 16);
        }
#ifdef KLT_EXP_NOLOADS
        S.v[t] = make_uint4(a + t, a ^ 0x5a5a5a5au, a * 3u + t, 0x40302010u);
#else
        S.v[t] = ld_b128(J.base + (size_t)a);
#endif
    }
}
__device__ __forceinline__ void klt3_store_region(const klt3_segs &S, int c, bool store, unsigned char *lj)
{
    if (store) {
#pragma unroll
        for (int t = 0; t < klt3::NLD; ++t) {
            if (klt3::RBYTES == 48) *reinterpret_cast<uint4 *>(lj + t * klt3::RBYTES + 16 * c) = S.v[t];
            else if (3 * t + c < 2 * klt3::RROWS) *reinterpret_cast<uint4 *>(lj + (3 * t + c) * 16) = S.v[t];
        }
    }
}

// One LKTrackerInvoker pass for the keypoint of this three-lane group; lane c owns window columns 3c .. 3c + 2.
// `lane_ok` is false for the idle sixteenth lane of a DPP row (it computes along on zeros and never stores).
// ---- divergence tail: yield and resume ----------------------------------------------------------------------------
// A wave of 20 keypoints runs a level pass until its SLOWEST keypoint has converged: on a hard stream (large flow, poor
// priors) a few keypoints iterate 15-30 times while 16-19 groups of the wave idle through every one of those iterations
// at full issue cost.  So a pass that has made y_after iterations and is down to <= y_groups running keypoints stops: the
// stragglers write their state (level, iteration count, position, previous step) to a list and leave the wave, which goes
// on to the next level with the others.  A second, small launch (klt_resume_kernel) packs the stragglers of ALL waves 20
// to a wave, rebuilds each one's template at its level (same integers) and continues the SAME iteration sequence from the
// stored state, then the remaining levels and the backward pass.  Per-keypoint arithmetic is untouched, so results stay
// bit-identical to the scalar oracle; what changes is which wave executes an iteration.
// MEASURED (64 x 2048 keypoints, front-end alone, easy / hard stream in k frames/s; off = 208 / 121): stragglers to the resume
// launch only, yield (6, 5): 184 / 109, (10, 3): 187 / 107 -- the first launch gets 38 us shorter, but the stragglers are the
// longest chains of the call and a launch of ~330 of their waves cannot end before its slowest one (~70 us with most of the
// device idle); letting the first launch's own waves take batches of stragglers as they finish (OV2_KLT_PICKUP = k: every
// k-th wave, one compare-and-swap attempt each): 136 / 82 at best -- a wave that picks up a batch becomes a long chain
// itself.  (On the way: a retry loop around that compare-and-swap held the launch for 5-20 ms -- thousands of waves finish
// within microseconds of each other -- and a release / acquire fence pair per record at agent scope cost the same: it
// writes back / invalidates the issuing XCD's whole L2.)  The waiting lanes are cheaper than any of this, so the switch
// is OFF by default (ov2_klt_set_yield / OV2_KLT_YIELD turn it on); the tests run every setting against the oracle.
struct klt_rec {
    int i;               // keypoint
    int info;            // bits 0-3 level, bit 4: backward pass, bit 5: list A (2-level call), bits 8-15: iterations made in the pass
    float nx, ny;        // position of the pass (level scale), = nx_io / ny_io: what the pass would return if it ended here
    float rx, ry;        // the loop's own running position (nx_io - half is NOT it: nx_io = fl(running + half) has been rounded)
    float pdx, pdy;      // previous step (the oscillation test reads it)
    float fx, fy;        // backward pass: the forward result
    unsigned it, passes; // work so far
    float err;           // backward pass: the forward minimum eigenvalue (gate already taken)
};
// In memory a record is KLT_REC_WORDS 64-bit words, each (epoch << 32) | one 32-bit field: every word validates itself, so
// a record can be handed from a running wave to another one of the SAME launch with relaxed device-scope atomic stores /
// loads and no fence -- on this device a release / acquire pair at agent scope writes back / invalidates the whole L2 of
// the issuing XCD (measured: the launch went from 0.14 to 8.5 ms with a fence per record).  The buffer belongs to the
// context and is only ever written here, with the call's serial number as epoch: a word of an older call never matches.
#define KLT_REC_WORDS 13
#define KLT_REC_STRIDE 16   // 64-bit words per record slot (128 bytes)
static_assert(sizeof(klt_rec) == 4 * KLT_REC_WORDS, "one tagged word per 32-bit field");
struct lk_res { bool here; int j0; float pdx, pdy, rx, ry; };   // this lane's group resumes IN this pass
struct lk_yld { int after, groups; bool yielded; int j; float pdx, pdy, rx, ry; };

template <int WIN>
__device__ __forceinline__ int lk_level3(const level_ptrs &I, const level_ptrs &J, int pad, int level, int max_level,
                                         bool run, float kx, float ky, float &nx_io, float &ny_io, int &status,
                                         float &err, const klt_params &P, int c, bool lane_ok, unsigned &passes,
                                         unsigned char *lj, const lk_res &R, lk_yld &Y)
{
    static_assert(WIN == 9, "three lanes x three columns");
    constexpr int NPX = 3 * WIN, NQ = (NPX + 1) / 2;   // window pixels per lane, flat e = 3 * row + column; pairs of them
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    const float half = (float)(WIN - 1) * 0.5f;
    const float lscale = __builtin_ldexpf(1.f, -level);
    float px = kx * lscale, py = ky * lscale;
    float nx, ny;
    if (R.here) { nx = nx_io; ny = ny_io; }   // a resumed pass continues where it stopped
    else if (level == max_level) { nx = nx_io * lscale; ny = ny_io * lscale; }
    else { nx = nx_io * 2.f; ny = ny_io * 2.f; }
    if (run) { nx_io = nx; ny_io = ny; }

    px -= half; py -= half;
    const int ipx = (int)floorf(px), ipy = (int)floorf(py);
    if (run && (ipx < -WIN || ipx >= I.w || ipy < -WIN || ipy >= I.h)) {
        if (level == 0) { status = 0; err = 0.f; }
        run = false;
    }
    if (!__any(run)) return 0;
    if (run) passes += 1u;
    int w00, w01, w10, w11;
    lk_weights(px - (float)ipx, py - (float)ipy, w00, w01, w10, w11);

    unsigned Iv2[NQ], Ix2[NQ], Iy2[NQ];
    int sA11 = 0, sA12 = 0, sA22 = 0;
    int rr0, cc0;
    {
        // search region around the first iteration's window, fetched together with the template rows
        const float sx = nx - half, sy = ny - half;
        const int inx0 = (int)floorf(sx), iny0 = (int)floorf(sy);
        const bool in0 = run && !(inx0 < -WIN || inx0 >= J.w || iny0 < -WIN || iny0 >= J.h);
        klt3_region_origin(J, pad, in0 ? inx0 : 0, in0 ? iny0 : 0, rr0, cc0);
        const int bx0 = (run ? (OV2_LM + ipx) : OV2_LM) + 3 * c, by = run ? (ipy + pad) : pad;
        // Template rows.  The Scharr derivatives are formed HERE from the image rows (one 12-byte load per lane and row, 12
        // rows: the window's 10 plus a ring of one) instead of being read from a gradient plane: the int16 (Ix, Iy)
        // planes are 4 of the 5 bytes per pixel a pyramid holds and their windows made this kernel HBM-bound (and the
        // pyramid build write-bound).  Same integers as calcSharrDeriv / level_kernel's Scharr part: [3 10 3]' x [-1 0 1]
        // and its transpose in any order, zero outside the image (the derivative border of buildOpticalFlowPyramid).
        const int ax = bx0 - 1;   // lane's first byte: window column 3c - 1
        const unsigned io = I.img_o + (unsigned)(by * I.istride + (ax & ~3));
        const unsigned shI = (unsigned)ax & 3u;
        struct row3 { unsigned x, y, z; };
        row3 irow[WIN + 3];
        // loaded row p = image row by - 1 + p; only the first / last can leave the plane, and then only under positions the
        // border mask zeroes: they re-read their neighbour
#ifdef KLT_EXP_NOLOADS
#define KLT3_LOAD_ROW(p) (irow[p].x = io + (p), irow[p].y = io * 7u + (p), irow[p].z = io ^ ((p) * 0x01010101u))
#else
#define KLT3_LOAD_ROW(p)                                                                                                  \
    do {                                                                                                                  \
        const int dr = (p) == 0 ? (by > 0 ? -1 : 0) : ((p) == WIN + 2 ? (by + WIN + 1 < I.rows ? WIN + 1 : WIN) : (p) - 1); \
        __builtin_memcpy(&irow[p], __builtin_assume_aligned(I.base + (size_t)(io + (unsigned)(dr * I.istride)), 4), 12);   \
    } while (0)
#endif
        // two batches (the first travels with the region, the second sits behind the barrier that publishes the region):
        // keeps the registers in flight down
        constexpr int R1 = 6;
        __syncthreads();   // the previous pass may still read the region
        klt3_segs S;
        klt3_load_region(J, rr0, cc0, c, S);
#pragma unroll
        for (int p = 0; p < R1; ++p) KLT3_LOAD_ROW(p);
        klt3_store_region(S, c, lane_ok, lj);
        __syncthreads();   // region staged
#pragma unroll
        for (int p = R1; p < WIN + 3; ++p) KLT3_LOAD_ROW(p);
#undef KLT3_LOAD_ROW

        const unsigned W01 = pack_lo16((unsigned)w00, (unsigned)w01), W23 = pack_lo16((unsigned)w10, (unsigned)w11);
        // derivative positions outside the image are zero: masks per column pair and per row (always applied: a
        // wave-uniform branch around them costs 60 more registers than it saves time); idle lanes / groups get all-zero
        // masks, so their gradients -- and with them everything they add to the sums below -- vanish
        const int x0 = ipx + 3 * c;
        const unsigned wlim = (run && lane_ok) ? (unsigned)I.w : 0u, hlim = (run && lane_ok) ? (unsigned)I.h : 0u;
        unsigned cm[2];   // columns (0, 1) and (2, 3) of the lane
#pragma unroll
        for (int j = 0; j < 2; ++j)
            cm[j] = ((unsigned)(x0 + 2 * j) < wlim ? 0xffffu : 0u) | ((unsigned)(x0 + 2 * j + 1) < wlim ? 0xffff0000u : 0u);
        // per loaded row: pixel pairs P_k = (q_k, q_k+1), k = 0..4, of the lane's six pixels q_0..q_5 (window columns
        // 3c - 1 .. 3c + 4); horizontal terms for the column pairs (0, 1) and (2, 3): Hd = P_k+2 - P_k, Sm = 3 (P_k + P_k+2)
        // + 10 P_k+1 with k = 0 / 2 (packed 16-bit); the middle pair (1, 2) of a derivative row is cut out of the two
        unsigned Pm[2][3] = {};              // P_1..P_3 (the window's own pixel pairs) of the two previous loaded rows
        unsigned Hd[3][2] = {}, Sm[3][2] = {};   // [age][pair]: ages 0 / 1 / 2 = loaded rows p - 2 / p - 1 / p
        unsigned GXp[3] = {0, 0, 0}, GYp[3] = {0, 0, 0};   // derivative pairs of the previous window row
        unsigned piv = 0, pix = 0, piy = 0;
#pragma unroll
        for (int p = 0; p < WIN + 3; ++p) {
            const unsigned e0 = __builtin_amdgcn_alignbyte(irow[p].y, irow[p].x, shI);
            const unsigned e1 = __builtin_amdgcn_alignbyte(irow[p].z, irow[p].y, shI);
            unsigned P[5];
            P[0] = spread_pair<0>(e0); P[1] = spread_pair<1>(e0); P[2] = spread_pair<2>(e0);
            P[3] = __builtin_amdgcn_perm(e1, e0, 0x0c040c03u);   // (e0 byte 3, e1 byte 0)
            P[4] = spread_pair<0>(e1);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                Hd[0][j] = Hd[1][j]; Hd[1][j] = Hd[2][j]; Sm[0][j] = Sm[1][j]; Sm[1][j] = Sm[2][j];
                Hd[2][j] = pk_sub16(P[2 * j + 2], P[2 * j]);
                Sm[2][j] = pk_mad16(P[2 * j + 1], 10u, pk_mul16(pk_add16(P[2 * j], P[2 * j + 2]), 3u));
            }
            if (p >= 2) {
                // derivative pairs of window row r = p - 2 (loaded rows p - 2, p - 1, p)
                const unsigned rm = (unsigned)(ipy + p - 2) < hlim ? 0xffffffffu : 0u;
                unsigned GX[3], GY[3];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned m = cm[j] & rm;
                    GX[2 * j] = pk_mad16(Hd[1][j], 10u, pk_mul16(pk_add16(Hd[0][j], Hd[2][j]), 3u)) & m;
                    GY[2 * j] = pk_sub16(Sm[2][j], Sm[0][j]) & m;
                }
                // (column 1, column 2) = (high half of pair 0, low half of pair 2)
                GX[1] = __builtin_amdgcn_perm(GX[2], GX[0], 0x05040302u); GY[1] = __builtin_amdgcn_perm(GY[2], GY[0], 0x05040302u);
                if (p >= 3) {
                    // window pixel row y = p - 3: intensities from loaded rows y + 1, y + 2, derivatives of rows y, y + 1
                    const int y = p - 3;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const int e = 3 * y + i;
                        const unsigned av = (unsigned)dot2(Pm[1][i], W23, dot2k(Pm[0][i], W01, 1 << (W_BITS - 5 - 1)));
                        const int ax_ = dot2(GX[i], W23, dot2k(GXp[i], W01, 1 << (W_BITS - 1)));
                        const int ay_ = dot2(GY[i], W23, dot2k(GYp[i], W01, 1 << (W_BITS - 1)));
                        if (e & 1) {
                            Iv2[e >> 1] = shr_pack_hi<W_BITS - 5>(piv, av);
                            Ix2[e >> 1] = ashr_pack_hi<W_BITS>(pix, ax_);
                            Iy2[e >> 1] = ashr_pack_hi<W_BITS>(piy, ay_);
                        } else if (e == NPX - 1) {   // the odd last pixel: its phantom partner has zero gradients
                            Iv2[e >> 1] = av >> (W_BITS - 5);
                            Ix2[e >> 1] = (unsigned)(ax_ >> W_BITS) & 0xffffu; Iy2[e >> 1] = (unsigned)(ay_ >> W_BITS) & 0xffffu;
                        } else {
                            piv = av >> (W_BITS - 5); pix = (unsigned)(ax_ >> W_BITS); piy = (unsigned)(ay_ >> W_BITS);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) { GXp[j] = GX[j]; GYp[j] = GY[j]; }
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) { Pm[0][j] = Pm[1][j]; Pm[1][j] = P[j + 1]; }
        }
        // |Ix|, |Iy| <= 4080: 27 products per lane and 81 per group stay below 2^31 -> exact in int32
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            sA11 = dot2(Ix2[q], Ix2[q], sA11); sA12 = dot2(Ix2[q], Iy2[q], sA12); sA22 = dot2(Iy2[q], Iy2[q], sA22);
        }
    }
    const float A11 = (float)(double)sum3_i32(sA11, c) * FLT_SCALE;
    const float A12 = (float)(double)sum3_i32(sA12, c) * FLT_SCALE;
    const float A22 = (float)(double)sum3_i32(sA22, c) * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    const float min_eig = __fdiv_rn(A22 + A11 - __fsqrt_rn((A11 - A22) * (A11 - A22) + 4.f * A12 * A12),
                                    (float)(2 * WIN * WIN));
#ifndef KLT_EXP_FIXED_ITERS
    if (run) {
        err = min_eig;  // OPTFLOW_LK_GET_MIN_EIGENVALS
        if (min_eig < P.min_eig_thr || D < 1.1920929e-07f /* FLT_EPSILON */) {
            if (level == 0) status = 0;
            run = false;
        }
    }
#else
    err = min_eig * 1e-9f;
#endif
    D = __fdiv_rn(1.f, D);

    nx -= half; ny -= half;
    if (R.here) { nx = R.rx; ny = R.ry; }   // bit for bit the value the loop held when it stopped
    float pdx = R.here ? R.pdx : 0.f, pdy = R.here ? R.pdy : 0.f;
    int jl = R.here ? R.j0 : 0;   // iterations the GROUP has made in this pass (a resumed group starts above zero)
    int iters = 0;
    for (int trip = 0; trip < P.max_iter; ++trip) {
        if (!__any(run)) break;
        if (Y.after > 0 && trip >= Y.after) {   // wave-uniform: the pass is down to its stragglers -> they leave (see klt_rec)
            if (__popcll(__ballot(run)) <= 3 * Y.groups) {
                if (run) { Y.yielded = true; Y.j = jl; Y.pdx = pdx; Y.pdy = pdy; Y.rx = nx; Y.ry = ny; run = false; }
                break;
            }
        }
        const int inx = (int)floorf(nx), iny = (int)floorf(ny);
        if (run && (inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h)) {
            if (level == 0) status = 0;
            run = false;
        }
        int w00j, w01j, w10j, w11j;
        lk_weights(nx - (float)inx, ny - (float)iny, w00j, w01j, w10j, w11j);
        int rb = run ? iny + pad - rr0 : 0, cb = run ? OV2_LM + inx - cc0 : 0;
        const bool out = (unsigned)rb > (unsigned)klt3::ROW_SPAN || (unsigned)cb > (unsigned)klt3::COL_SPAN;
        if (__any(out)) {   // the window left the staged region: fetch a new one around it
            __syncthreads();
            if (out) {
                klt3_region_origin(J, pad, inx, iny, rr0, cc0);
                klt3_segs S;
                klt3_load_region(J, rr0, cc0, c, S);
                klt3_store_region(S, c, lane_ok, lj);
                rb = iny + pad - rr0; cb = OV2_LM + inx - cc0;
            }
            __syncthreads();
        }
        const unsigned W01 = pack_lo16((unsigned)w00j, (unsigned)w01j), W23 = pack_lo16((unsigned)w10j, (unsigned)w11j);
        const int cq = cb + 3 * c;
        const unsigned *q = reinterpret_cast<const unsigned *>(lj + rb * klt3::RBYTES + (cq & ~3));
        const unsigned sh = (unsigned)cq & 3u;
        int pb1 = 0, pb2 = 0;
        unsigned T[3];
        {
            const unsigned p4 = __builtin_amdgcn_alignbyte(q[1], q[0], sh);
            T[0] = spread_pair<0>(p4); T[1] = spread_pair<1>(p4); T[2] = spread_pair<2>(p4);
        }
        unsigned pj = 0;
#pragma unroll
        for (int y = 0; y < WIN; ++y) {
            const unsigned *qr = q + (klt3::RBYTES / 4) * (y + 1);
            const unsigned p4 = __builtin_amdgcn_alignbyte(qr[1], qr[0], sh);
            unsigned B[3];
            B[0] = spread_pair<0>(p4); B[1] = spread_pair<1>(p4); B[2] = spread_pair<2>(p4);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int e = 3 * y + i;
                // taps + rounding >= 1 (w11 >= -1), so the logical shift is the arithmetic one
                const unsigned aj = (unsigned)dot2(B[i], W23, dot2k(T[i], W01, 1 << (W_BITS - 5 - 1)));
                if ((e & 1) || e == NPX - 1) {
                    // |jv - I| <= 8160 fits int16: v_pk_sub_i16 on the pixel pair, then one dot2 per gradient component
                    const unsigned d2 = pk_sub16((e & 1) ? shr_pack_hi<W_BITS - 5>(pj, aj) : (aj >> (W_BITS - 5)), Iv2[e >> 1]);
                    pb1 = dot2(d2, Ix2[e >> 1], pb1);
                    pb2 = dot2(d2, Iy2[e >> 1], pb2);
                } else {
                    pj = aj >> (W_BITS - 5);
                }
                T[i] = B[i];
            }
        }
        // a lane holds 27 products of <= 3.4e7: fits int32; the group total may not, so sum exactly in f64 unless every
        // partial of the wave is below 2^27 (then the total fits int32 and v_cvt_f32_i32 rounds it once, like
        // (float)(double)total)
        float b1, b2;
        if (__all((unsigned)(pb1 + (1 << 27)) < (1u << 28) && (unsigned)(pb2 + (1 << 27)) < (1u << 28))) {
            b1 = (float)sum3_i32(pb1, c) * FLT_SCALE;
            b2 = (float)sum3_i32(pb2, c) * FLT_SCALE;
        } else {
            b1 = (float)sum3_f64((double)pb1, c) * FLT_SCALE;
            b2 = (float)sum3_f64((double)pb2, c) * FLT_SCALE;
        }
        const float dx = (A12 * b2 - A22 * b1) * D;
        const float dy = (A12 * b1 - A11 * b2) * D;
        if (run) {
            ++iters;
            nx += dx; ny += dy;
            nx_io = nx + half; ny_io = ny + half;
#ifdef KLT_EXP_FIXED_ITERS   // timing experiments: every pass runs exactly this many iterations around the start position
            nx -= dx; ny -= dy;
            if (trip + 1 >= KLT_EXP_FIXED_ITERS) run = false;
#else
            if ((double)dx * dx + (double)dy * dy <= P.eps2) run = false;
            else if (jl > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) {
                nx_io -= dx * 0.5f; ny_io -= dy * 0.5f;
                run = false;
            }
#endif
            pdx = dx; pdy = dy;
            if (++jl >= P.max_iter) run = false;   // criteria.maxCount
        }
    }
    return iters;
}

// FeatureTracker::fbKltTracking for the keypoint of this DPP row (act = row has a keypoint).
// returns status (0/1); fx,fy = forward result; work = iterations | level passes << 16.
// `smem` = the wave's LDS block (klt_smem<WIN, GL>::BYTES), `slot` = the keypoint's slot in the wave
template <int WIN, int GL>
__device__ __forceinline__ int lk_pass(const level_ptrs &I, const level_ptrs &J, int pad, int level, int max_level, bool run,
                                       float kx, float ky, float &nx_io, float &ny_io, int &status, float &err,
                                       const klt_params &P, int sub, bool lane_ok, unsigned &passes, unsigned char *smem, int slot)
{
    if constexpr (GL == 3) {
        const lk_res R0 = {false, 0, 0.f, 0.f, 0.f, 0.f};
        lk_yld Y0 = {0, 0, false, 0, 0.f, 0.f, 0.f, 0.f};
        return lk_level3<WIN>(I, J, pad, level, max_level, run, kx, ky, nx_io, ny_io, status, err, P, sub, lane_ok, passes,
                              smem + slot * klt3::RSZ, R0, Y0);
    } else {
        constexpr int KPW = 64 / GL, WB = klt_lds<WIN>::WB, GB = klt_lds<WIN>::GB;
        return lk_level<WIN, GL>(I, J, pad, level, max_level, run, kx, ky, nx_io, ny_io, status, err, P, sub, passes,
                                 smem + slot * WB, smem + KPW * WB + slot * WB, smem + 2 * KPW * WB + slot * GB);
    }
}

template <int WIN, int GL>
__device__ __forceinline__ int fb_track(const ov2_pyr_view &pv, const ov2_pyr_view &cv, int b, bool act, float kx,
                                          float ky, float &fx, float &fy, const klt_params &P, int nlevels, int sub,
                                          bool lane_ok, unsigned &work, unsigned char *smem, int slot)
{
    int status = 1;
    float err = 0.f;
    unsigned it = 0, passes = 0;
    for (int l = nlevels; l >= 0; --l) {
        const level_ptrs I = level_of(pv, l, b), J = level_of(cv, l, b);
        it += lk_pass<WIN, GL>(I, J, pv.pad, l, nlevels, act, kx, ky, fx, fy, status, err, P, sub, lane_ok, passes, smem, slot);
    }
    // gates of src/feature_tracker.cpp:79-101
    const int W0 = cv.lv[0].w, H0 = cv.lv[0].h;
    int ok = act && status && !(err > P.err_th) &&
             (1.f <= fx && fx < (float)W0 - 1.f && 1.f <= fy && fy < (float)H0 - 1.f);
#ifdef KLT_EXP_FIXED_ITERS
    ok = act;
#endif
    // backward pass cur -> prev on level 0 from the original keypoint (src/feature_tracker.cpp:113)
    int st2 = 1;
    float e2 = 0.f, bx = kx, by = ky;
    {
        const level_ptrs I = level_of(cv, 0, b), J = level_of(pv, 0, b);
        it += lk_pass<WIN, GL>(I, J, cv.pad, 0, 0, ok != 0, fx, fy, bx, by, st2, e2, P, sub, lane_ok, passes, smem, slot);
    }
    if (ok) {
        if (!st2) ok = 0;
        else {
            const float dx = kx - bx, dy = ky - by;
            const double nrm = __dsqrt_rn((double)dx * dx + (double)dy * dy);  // cv::norm(Point2f) is double
            if (nrm > P.fb_th) ok = 0;
        }
    }
    work = it | (passes << 16);
    return ok;
}

// fb_track for the three-lane mapping with the yield / resume protocol of klt_rec.  rin: the record this group resumes from
// (null: a fresh keypoint).  Returns the status as fb_track does; `yielded` = the group left in the middle of a pass and
// `rout` holds its state (nothing else of the keypoint may be written then).
template <int WIN>
__device__ __forceinline__ int fb_track3(const ov2_pyr_view &pv, const ov2_pyr_view &cv, int b, bool act, float kx, float ky,
                                         float &fx, float &fy, const klt_params &P, int nlevels, int top_level, int sub,
                                         bool lane_ok, unsigned &work, unsigned char *smem, int slot, int y_after, int y_groups,
                                         const klt_rec *rin, klt_rec &rout, bool &yielded)
{
    int status = 1;
    float err = 0.f;
    unsigned it = rin ? rin->it : 0u, passes = rin ? rin->passes : 0u;
    const bool rback = rin && (rin->info & 16);
    const int rlevel = rin ? (rin->info & 15) : nlevels;
    const int rj = rin ? ((rin->info >> 8) & 0xff) : 0;
    yielded = false;
    unsigned char *lds = smem + slot * klt3::RSZ;
    if (rin && !rback) { fx = rin->nx; fy = rin->ny; }
    for (int l = nlevels; l >= 0; --l) {
        const level_ptrs I = level_of(pv, l, b), J = level_of(cv, l, b);
        const bool run = act && !yielded && !rback && l <= rlevel;
        const lk_res R = {rin && !rback && l == rlevel, rj, rin ? rin->pdx : 0.f, rin ? rin->pdy : 0.f, rin ? rin->rx : 0.f, rin ? rin->ry : 0.f};
        lk_yld Y = {y_after, y_groups, false, 0, 0.f, 0.f, 0.f, 0.f};
        it += lk_level3<WIN>(I, J, pv.pad, l, top_level, run, kx, ky, fx, fy, status, err, P, sub, lane_ok, passes, lds, R, Y);
        if (Y.yielded) {
            yielded = true;
            rout.info = l | (Y.j << 8); rout.nx = fx; rout.ny = fy; rout.rx = Y.rx; rout.ry = Y.ry; rout.pdx = Y.pdx; rout.pdy = Y.pdy; rout.fx = 0.f; rout.fy = 0.f;
            rout.err = 0.f;
        }
    }
    const int W0 = cv.lv[0].w, H0 = cv.lv[0].h;
    int ok;
    if (rback) { ok = act; fx = rin->fx; fy = rin->fy; err = rin->err; }   // the gates were taken before the backward pass began
    else ok = act && !yielded && status && !(err > P.err_th) && (1.f <= fx && fx < (float)W0 - 1.f && 1.f <= fy && fy < (float)H0 - 1.f);
    int st2 = 1;
    float e2 = 0.f, bx = rback ? rin->nx : kx, by = rback ? rin->ny : ky;
    {
        const level_ptrs I = level_of(cv, 0, b), J = level_of(pv, 0, b);
        const lk_res R = {rback, rj, rin ? rin->pdx : 0.f, rin ? rin->pdy : 0.f, rin ? rin->rx : 0.f, rin ? rin->ry : 0.f};
        lk_yld Y = {y_after, y_groups, false, 0, 0.f, 0.f, 0.f, 0.f};
        it += lk_level3<WIN>(I, J, cv.pad, 0, 0, ok != 0, fx, fy, bx, by, st2, e2, P, sub, lane_ok, passes, lds, R, Y);
        if (Y.yielded) {
            yielded = true;
            rout.info = 16 | (Y.j << 8); rout.nx = bx; rout.ny = by; rout.rx = Y.rx; rout.ry = Y.ry; rout.pdx = Y.pdx; rout.pdy = Y.pdy; rout.fx = fx; rout.fy = fy;
            rout.err = err;
        }
    }
    if (ok && !yielded) {
        if (!st2) ok = 0;
        else {
            const float dx = kx - bx, dy = ky - by;
            const double nrm = __dsqrt_rn((double)dx * dx + (double)dy * dy);  // cv::norm(Point2f) is double
            if (nrm > P.fb_th) ok = 0;
        }
    }
    rout.it = it; rout.passes = passes;
    work = it | (passes << 16);
    return ok;
}

// the keypoints that yielded in this wave go to the resume list (one atomic per wave that has any)
// A record is PUBLISHED by its epoch word (the call's serial number), stored after a fence behind the other fields:
// waves of the same launch pick records up while their producers are still running (klt_take_yielded).
__device__ __forceinline__ void klt_push_yielded(bool push, const klt_rec &r, unsigned long long *__restrict__ ylist, unsigned *__restrict__ ycnt, int epoch)
{
    const unsigned long long ym = __ballot(push);
    if (!ym) return;
    const int lane = (int)threadIdx.x, leader = __ffsll((long long)ym) - 1;
    int base = 0;
    if (lane == leader) base = (int)atomicAdd(ycnt, (unsigned)__popcll(ym));
    base = __shfl(base, leader);
    if (push) {
        unsigned long long *dst = ylist + (size_t)(base + __popcll(ym & ((1ull << lane) - 1ull))) * KLT_REC_STRIDE;
        unsigned f[KLT_REC_WORDS];
        __builtin_memcpy(f, &r, sizeof(f));
        const unsigned long long tag = (unsigned long long)(unsigned)epoch << 32;
#pragma unroll
        for (int k = 0; k < KLT_REC_WORDS; ++k) (void)atomicExch(&dst[k], tag | f[k]);   // read-modify-write atomics meet at the device's coherence point
    }
}

// reads the record at slot `idx`, waiting for every word of it to carry this call's epoch
__device__ __forceinline__ klt_rec klt_read_rec(const unsigned long long *__restrict__ ylist, int idx, int epoch)
{
    const unsigned long long *src = ylist + (size_t)idx * KLT_REC_STRIDE;
    unsigned f[KLT_REC_WORDS];
#pragma unroll
    for (int k = 0; k < KLT_REC_WORDS; ++k) {
        unsigned long long v;
        // a read-modify-write (OR with zero), not a load: a plain device-scope load is served by the issuing XCD's own L2, which
        // keeps returning the line it fetched before the producer (on another XCD) wrote it -- measured: milliseconds of spinning
        while ((unsigned)((v = atomicOr(const_cast<unsigned long long *>(&src[k]), 0ull)) >> 32) != (unsigned)epoch)
            __builtin_amdgcn_s_sleep(1);   // its producer is a running wave of this launch: the word is on its way
        f[k] = (unsigned)v;
    }
    klt_rec r;
    __builtin_memcpy(&r, f, sizeof(f));
    return r;
}

// A wave that has finished its own keypoints takes up to KPW published-or-reserved records off the list (one CAS per
// wave); 0 = nothing there right now (records reserved later belong to the resume launch).  cnt[3] = records reserved,
// cnt[4] = records taken.
__device__ __forceinline__ int klt_take_yielded(unsigned *__restrict__ cnt, int kpw, int &first)
{
    int t = 0, nrec = 0;
    if (threadIdx.x == 0) {
        // ONE attempt: thousands of waves finish within microseconds of each other, and a retry loop on this one address
        // turned into a compare-and-swap storm that held the launch for milliseconds (measured)
        const unsigned taken = atomicOr(&cnt[4], 0u), avail = atomicOr(&cnt[3], 0u);
        if (taken < avail) {
            const unsigned want = min((unsigned)kpw, avail - taken);
            if (atomicCAS(&cnt[4], taken, taken + want) == taken) { t = (int)taken; nrec = (int)want; }
        }
    }
    first = __shfl(t, 0);
    return __shfl(nrec, 0);
}

// Lanes per keypoint, chosen per call: 8 lanes (eight keypoints per wave) halve the wave-instructions per keypoint and
// win once the launch holds several rounds of waves (measured: 106k vs 103k frames/s at 64 x 2048 keypoints per call);
// 16 lanes (four per wave) give twice as many, shorter waves and win while the launch is latency-bound (68k vs 60k
// frames/s at 16 x 2048).
#define KLT_GL8_MIN_KPS 65536
// occupancy bounds handed to the register allocator (min, max waves per SIMD).  Measured on 64 x 2048 keypoints,
// first launch: register-squeezed to 6-7 waves 228 us, unconstrained (5 waves) 209 us, capped at 3 / 4 / 5 waves
// 204 / 203.5 / 205 us -- the kernel is issue-bound, extra waves only add pressure, so the allocator gets room.
#ifndef KLT_WAVE_CAP
#define KLT_WAVE_CAP 4
#endif
#ifndef KLT3_WAVE_MIN
#define KLT3_WAVE_MIN 2
#endif
#define KLT_WAVES(W, G) ((G) == 3 ? KLT3_WAVE_MIN : 1), KLT_WAVE_CAP

// 64 threads = 64 / GL keypoints.  grid = ceil(n / (64 / GL))
template <int WIN, int KLT_GL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(KLT_WAVES(WIN, KLT_GL)))) void klt_fb_kernel(ov2_pyr_view pv, ov2_pyr_view cv, klt_params P, int n,
                                                    const float2 *__restrict__ kps, float2 *__restrict__ priors,
                                                    unsigned char *__restrict__ status,
                                                    const int *__restrict__ img_idx, unsigned *__restrict__ iters)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[klt_smem<WIN, KLT_GL>::BYTES];
    using M = klt_map<KLT_GL>;
    constexpr int KLT_KPW = M::KPW;
    const int slot = M::slot((int)threadIdx.x), sub = M::sub((int)threadIdx.x), i = blockIdx.x * KLT_KPW + slot;
    const bool lane_ok = M::lane_ok((int)threadIdx.x);
    const bool act = i < n && lane_ok;
    const int ii = i < n ? i : 0;
    const int b = img_idx ? img_idx[ii] : 0;
    const float2 kp = kps[ii];
    float2 pr = priors[ii];
    unsigned work = 0;
    const int ok = fb_track<WIN, KLT_GL>(pv, cv, b, act, kp.x, kp.y, pr.x, pr.y, P, P.nlevels, sub, lane_ok, work, smem, slot);
    if (act && sub == 0) {
        priors[i] = pr;
        status[i] = (unsigned char)ok;
        if (iters) iters[i] = work;
    }
}

// ---- VisualFrontEnd::kltTracking, two stages without a host round trip -------------------------------
// Only a fraction of the keypoints is live in each stage (those with a prior in stage 1; the rest + the stage-1
// failures in stage 2).  A small compaction pass writes the indices of the live keypoints (order irrelevant: every
// keypoint is tracked independently) and their count; the tracking kernels are single-wave workgroups, workgroup g
// takes live keypoints [g*KPW, (g+1)*KPW) and workgroups beyond the count retire at once.  So every launched wave is
// full, there is no intra-workgroup imbalance, and the grid needs no host-side knowledge of the live count.

// pass 1 lists the keypoints with a prior (list A, 2 pyramid levels) and those without (list B, full pyramid: they do
// not depend on the outcome of the first group, so both groups run in ONE launch); pass 2 lists the failures of
// list A, which are re-tracked on the full pyramid once the per-image tally of list A is complete (the 33 % rule).
__global__ __launch_bounds__(256) void klt_compact_kernel(int n, int pass, const unsigned char *__restrict__ has_prior,
                                                          const unsigned char *__restrict__ out_status,
                                                          unsigned *__restrict__ iters, int *__restrict__ list_a,
                                                          int *__restrict__ list_b, unsigned *__restrict__ cnt,
                                                          int *__restrict__ p3p_req, int batch)
{
    __shared__ int wsum[2][4];
    __shared__ int wbase[2];
    const int f = blockIdx.x * 256 + threadIdx.x;
    bool la = false, lb = false;
    if (f < n) {
        const bool hp = has_prior[f] != 0;
        if (pass == 1) {
            la = hp; lb = !hp;
            if (iters) iters[hp ? n + f : f] = 0;   // the slot of the stage the keypoint does not take (so far)
        } else {
            la = hp && out_status[f] == 0;
        }
    }
    if (pass == 1 && p3p_req)
        for (int k = f; k < batch; k += gridDim.x * 256) p3p_req[k] = 0;
    const unsigned long long ma = __ballot(la), mb = __ballot(lb);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int ra = __popcll(ma & below), rb = __popcll(mb & below);
    if (lane == 0) { wsum[0][wv] = __popcll(ma); wsum[1][wv] = __popcll(mb); }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int q = threadIdx.x;
        const int tot = wsum[q][0] + wsum[q][1] + wsum[q][2] + wsum[q][3];
        wbase[q] = tot ? (int)atomicAdd(&cnt[pass == 1 ? q : 2], (unsigned)tot) : 0;
    }
    __syncthreads();
    int oa = wbase[0], ob = wbase[1];
    for (int k = 0; k < wv; ++k) { oa += wsum[0][k]; ob += wsum[1][k]; }
    if (la) list_a[oa + ra] = f;
    if (lb) list_b[ob + rb] = f;
}

// what the first launch does with a finished keypoint (shared with the resume launch): result, status, work word, the
// per-image tally of list A and the failures of list A appended to the re-tracking list
__device__ __forceinline__ void klt_stage1_finish(bool mine, bool is_a, int i, int b, int n, int ok, float2 pr, unsigned work,
                                                  float2 *__restrict__ out_xy, unsigned char *__restrict__ out_status,
                                                  unsigned *__restrict__ counts, unsigned *__restrict__ iters,
                                                  unsigned *__restrict__ cnt, int *__restrict__ list_c)
{
    if (mine) {
        out_xy[i] = pr;   // tracked position, or the failed forward result that seeds the re-tracking (:217-219)
        out_status[i] = (unsigned char)ok;
        if (iters) iters[is_a ? i : n + i] = work;   // second half of the work-word array = full-pyramid passes
        // one atomic per keypoint, spread over 64 slots per image: same-address atomics serialise at ~10-16 ns
        // each (measured: they, not the tracking, bounded this kernel when every wave hit one counter)
        if (is_a) atomicAdd(&counts[64 * b + (i & 63)], 1u + ((unsigned)ok << 16));
    }
    // the failures of list A are what the second launch re-tracks: appended to its list here (one atomic per wave that
    // has any; the order of the list does not matter, every keypoint is tracked on its own) -- a compaction launch less
    // in the frame's chain
    const bool retry = mine && is_a && !ok;
    const unsigned long long rm = __ballot(retry);
    if (rm) {
        const int lane = (int)threadIdx.x, leader = __ffsll((long long)rm) - 1;
        int base = 0;
        if (lane == leader) base = (int)atomicAdd(&cnt[2], (unsigned)__popcll(rm));
        base = __shfl(base, leader);
        if (retry) list_c[base + __popcll(rm & ((1ull << lane) - 1ull))] = i;
    }
}

// one wave continues the records [first, first + nrec) of the list: each its pass, the levels below it and the backward pass
template <int WIN>
__device__ __forceinline__ void klt_resume_wave(const ov2_pyr_view &pv, const ov2_pyr_view &cv, const klt_params &P, int n,
                                                const float2 *__restrict__ kps, const int *__restrict__ img_idx,
                                                float2 *__restrict__ out_xy, unsigned char *__restrict__ out_status,
                                                unsigned *__restrict__ counts, unsigned *__restrict__ iters,
                                                unsigned *__restrict__ cnt, int *__restrict__ list_c,
                                                const unsigned long long *__restrict__ ylist, int first, int nrec, int epoch,
                                                unsigned char *smem)
{
    using M = klt_map<3>;
    const int slot = M::slot((int)threadIdx.x), sub = M::sub((int)threadIdx.x);
    const bool lane_ok = M::lane_ok((int)threadIdx.x);
    const bool act = slot < nrec && lane_ok;
    const klt_rec rec = klt_read_rec(ylist, first + (slot < nrec ? slot : nrec - 1), epoch);
    const int i = rec.i;
    const bool is_a = (rec.info & 32) != 0;
    const float2 kp = kps[i];
    const int b = img_idx ? img_idx[i] : 0;
    float2 pr = make_float2(rec.nx, rec.ny);
    unsigned work = 0;
    klt_rec dummy = {};
    bool yielded = false;
    // every level of the pyramid is walked (the groups of a wave resume at different ones); top_level beyond it: a level
    // below a group's resume level always starts from twice the position of the level above
    const int ok = fb_track3<WIN>(pv, cv, b, act, kp.x, kp.y, pr.x, pr.y, P, pv.nlevels - 1, pv.nlevels, sub, lane_ok, work, smem, slot, 0, 0,
                                  &rec, dummy, yielded);
    klt_stage1_finish(act && sub == 0, is_a, i, b, n, ok, pr, work, out_xy, out_status, counts, iters, cnt, list_c);
}

// the stragglers the first launch left behind (klt_rec; most are taken by its own waves), 20 to a wave
template <int WIN>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(KLT_WAVES(WIN, 3)))) void klt_resume_kernel(ov2_pyr_view pv, ov2_pyr_view cv, klt_params P, int n,
                                                        const float2 *__restrict__ kps, const int *__restrict__ img_idx,
                                                        float2 *__restrict__ out_xy, unsigned char *__restrict__ out_status,
                                                        unsigned *__restrict__ counts, unsigned *__restrict__ iters,
                                                        unsigned *__restrict__ cnt, int *__restrict__ list_c,
                                                        const unsigned long long *__restrict__ ylist, int epoch)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[klt_smem<WIN, 3>::BYTES];
    constexpr int KLT_KPW = klt_map<3>::KPW;
    const int first = (int)cnt[4] + (int)blockIdx.x * KLT_KPW, total = (int)cnt[3];   // what the first launch's waves did not take
    if (first >= total) return;
    klt_resume_wave<WIN>(pv, cv, P, n, kps, img_idx, out_xy, out_status, counts, iters, cnt, list_c, ylist, first,
                         min(KLT_KPW, total - first), epoch, smem);
}

// first launch: list A = keypoints with a prior on 2 pyramid levels (nbpyrlvl = 1, src/visual_front_end.cpp:190),
// list B = keypoints without prior on the full pyramid from their own position (:237-270, vpriors = vkps)
template <int WIN, int KLT_GL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(KLT_WAVES(WIN, KLT_GL)))) void klt_stage1_kernel(ov2_pyr_view pv, ov2_pyr_view cv, klt_params P, int n,
                                                        const float2 *__restrict__ kps,
                                                        const float2 *__restrict__ prior,
                                                        const int *__restrict__ img_idx, float2 *__restrict__ out_xy,
                                                        unsigned char *__restrict__ out_status,
                                                        unsigned *__restrict__ counts /* [batch][64] slots: low 16 bits n3d, high 16 good */,
                                                        unsigned *__restrict__ iters, const int *__restrict__ list_a,
                                                        const int *__restrict__ list_b, unsigned *__restrict__ cnt,
                                                        int *__restrict__ list_c, unsigned long long *__restrict__ ylist, int epoch)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[klt_smem<WIN, KLT_GL>::BYTES];
    using M = klt_map<KLT_GL>;
    constexpr int KLT_KPW = M::KPW;
    const int total_a = (int)cnt[0], total_b = (int)cnt[1];
    // list B first: its waves run the full pyramid (about twice the work of a list-A wave), so the short list-A waves
    // fill the tail of the launch instead of the long ones forming it
    const int groups_b = (total_b + KLT_KPW - 1) / KLT_KPW;
    const bool is_a = (int)blockIdx.x >= groups_b;
    const int g = is_a ? (int)blockIdx.x - groups_b : (int)blockIdx.x;
    const int total = is_a ? total_a : total_b;
    if (g * KLT_KPW >= total) return;
    const int slot = M::slot((int)threadIdx.x), sub = M::sub((int)threadIdx.x), idx = g * KLT_KPW + slot;
    const bool lane_ok = M::lane_ok((int)threadIdx.x);
    const bool act = idx < total && lane_ok;
    const int i = (is_a ? list_a : list_b)[idx < total ? idx : total - 1];
    const float2 kp = kps[i];
    const int b = img_idx ? img_idx[i] : 0;
    // kltTracking: keypoints without a prior start from their own position (vpriors = vkps, :181-183); stereoMatching's
    // full-pyramid list carries priors of its own (kp.px_ or the SAD prior, src/map_manager.cpp:419-436,486-488)
    float2 pr = (is_a || !P.rule33) ? prior[i] : kp;
    unsigned work = 0;
    const int nl = is_a ? min(1, pv.nlevels - 1) : P.nlevels;
    int ok;
    bool yielded = false;
    if constexpr (KLT_GL == 3) {
        klt_rec rec = {};
        ok = fb_track3<WIN>(pv, cv, b, act, kp.x, kp.y, pr.x, pr.y, P, nl, nl, sub, lane_ok, work, smem, slot, ylist ? P.y_after : 0,
                            P.y_groups, nullptr, rec, yielded);
        if (ylist) {   // wave-uniform
            rec.i = i; rec.info |= is_a ? 32 : 0;
            klt_push_yielded(act && sub == 0 && yielded, rec, ylist, &cnt[3], epoch);
        }
    } else {
        ok = fb_track<WIN, KLT_GL>(pv, cv, b, act, kp.x, kp.y, pr.x, pr.y, P, nl, sub, lane_ok, work, smem, slot);
    }
    klt_stage1_finish(act && sub == 0 && !yielded, is_a, i, b, n, ok, pr, work, out_xy, out_status, counts, iters, cnt, list_c);
    if constexpr (KLT_GL == 3) {
        // done with its own keypoints, the wave continues stragglers other waves have left -- ONE batch of at most 20: a wave
        // that kept taking batches turned the launch's last few waves into the only workers of a long serial tail (measured:
        // milliseconds); what is left when the launch ends goes to the resume launch, 20 to a wave across the whole device
        if (ylist && P.y_pickup > 0 && ((int)blockIdx.x % P.y_pickup) == 0) {   // every y_pickup-th wave tries (the counters are one cache line)
            int first;
            const int nrec = klt_take_yielded(cnt, KLT_KPW, first);
            if (nrec > 0) {
                __syncthreads();   // the wave's LDS region is reused
                klt_resume_wave<WIN>(pv, cv, P, n, kps, img_idx, out_xy, out_status, counts, iters, cnt, list_c, ylist, first, nrec, epoch, smem);
            }
        }
    }
}

// second launch: the failures of list A, full pyramid, from the failed forward result -- or from the keypoint itself
// when less than 33 % of the image's list-A keypoints were tracked (:228-233, which also raises bp3preq_)
template <int WIN, int KLT_GL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(KLT_WAVES(WIN, KLT_GL)))) void klt_stage2_kernel(ov2_pyr_view pv, ov2_pyr_view cv, klt_params P, int n,
                                                        const float2 *__restrict__ kps,
                                                        const int *__restrict__ img_idx, float2 *__restrict__ out_xy,
                                                        unsigned char *__restrict__ out_status,
                                                        const unsigned *__restrict__ counts, int *__restrict__ p3p_req,
                                                        unsigned *__restrict__ iters, const int *__restrict__ list_c,
                                                        const unsigned *__restrict__ cnt, int batch,
                                                        unsigned *__restrict__ next_counts, int next_words)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[klt_smem<WIN, KLT_GL>::BYTES];
    using M = klt_map<KLT_GL>;
    constexpr int KLT_KPW = M::KPW;
    const int slot = M::slot((int)threadIdx.x), sub = M::sub((int)threadIdx.x);
    const bool lane_ok = M::lane_ok((int)threadIdx.x);
    // this is the last kernel of a call: it leaves the OTHER counter buffer zeroed for the next call (nothing of this call
    // reads it), so no memset sits in front of every frame's tracking chain
    for (int q = blockIdx.x * 64 + threadIdx.x; q < next_words; q += gridDim.x * 64) next_counts[q] = 0u;
    // per-image tally of the first launch: its 64 slots shared out over the lanes of the group
    auto tally = [&](int b, int &n3, int &good) {
        n3 = 0; good = 0;
        if constexpr (KLT_GL == 3) {
            for (int q = sub; q < 64; q += 3) {
                const unsigned cw = counts[64 * b + q];
                n3 += (int)(cw & 0xffffu); good += (int)(cw >> 16);
            }
            n3 = sum3_i32(n3, sub); good = sum3_i32(good, sub);
        } else {
            for (int q = 0; q < 64 / KLT_GL; ++q) {
                const unsigned cw = counts[64 * b + sub * (64 / KLT_GL) + q];
                n3 += (int)(cw & 0xffffu); good += (int)(cw >> 16);
            }
            n3 = row_sum_i32<KLT_GL>(n3); good = row_sum_i32<KLT_GL>(good);
        }
    };
    // the 33 % flag is per image and must be raised even when no failure of that image is re-tracked
    if (p3p_req && P.rule33) {
        for (int b0 = blockIdx.x * KLT_KPW; b0 < batch; b0 += gridDim.x * KLT_KPW) {   // wave-uniform trip count (DPP sums inside)
            const int b = b0 + slot;
            int n3, good;
            tally(min(b, batch - 1), n3, good);
            if (b < batch && lane_ok && sub == 0 && n3 > 0 && (double)good < 0.33 * (double)n3) p3p_req[b] = 1;
        }
    }
    const int total = (int)cnt[2];
    if ((int)blockIdx.x * KLT_KPW >= total) return;
    const int idx = blockIdx.x * KLT_KPW + slot;
    const bool act = idx < total && lane_ok;
    const int i = list_c[idx < total ? idx : total - 1];
    const int b = img_idx ? img_idx[i] : 0;
    int n3, good;
    tally(b, n3, good);
    const bool drop = P.rule33 && n3 > 0 && (double)good < 0.33 * (double)n3;  // :228
    const float2 kp = kps[i];
    float2 pr = drop ? kp : out_xy[i];
    unsigned work = 0;
    const int ok = fb_track<WIN, KLT_GL>(pv, cv, b, act, kp.x, kp.y, pr.x, pr.y, P, P.nlevels, sub, lane_ok, work, smem, slot);
    if (act && sub == 0) {
        out_xy[i] = pr;
        out_status[i] = (unsigned char)ok;
        if (iters) iters[n + i] = work;
    }
}

// Lanes per keypoint for a call of n keypoints.  9 x 9 windows (the reference's nklt_win_size) take the three-lane
// mapping (20 keypoints per wave, Scharr derivatives formed in the kernel) at EVERY call size: measured per frame-batch of
// 308 keypoints (EuRoC size, scripts/klt_small_time.py), pyramid build + two-stage tracking back to back: 1 sequence 64.1 us
// against 71.4 us with the 16-lane kernels (which first need the four gradient-plane launches), 8 sequences 77.9 / 90.5,
// 64 sequences 109.8 / 182.5; synchronised latency of one frame 103 / 121 us.  ov2_klt_set_lanes (or OV2_KLT_LANES = 3 / 8 /
// 16 in the environment) forces one mapping: tests run every mapping against the oracle.
#define KLT_GL3_MIN_KPS 0
int klt_lanes_for(const ov2_ctx *c, int n, int win, const ov2_pyr *a, const ov2_pyr *b)
{
    static const int env = [] { const char *e = getenv("OV2_KLT_LANES"); return e ? atoi(e) : 0; }();
    const int forced = c->klt_lanes ? c->klt_lanes : env;
    // the three-lane kernels address a pyramid as base + 32-bit offset: allocations of 4 GB and more take the other kernels
    const bool fits32 = a->buf->bytes < (1ull << 32) && b->buf->bytes < (1ull << 32);
    if (forced == 8 || forced == 16 || (forced == 3 && win == 9 && fits32)) return forced;
    if (win == 9 && n >= KLT_GL3_MIN_KPS && fits32) return 3;
    return n >= KLT_GL8_MIN_KPS ? 8 : 16;
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
    P->rule33 = 1;
    P->y_after = c->klt_yield_after; P->y_groups = c->klt_yield_groups; P->y_pickup = c->klt_yield_pickup;
    return OV2_OK;
}

}  // namespace

extern "C" ov2_status ov2_klt_set_lanes(ov2_ctx *c, int lanes)
{
    if (!c) return OV2_ERR_INVALID;
    if (lanes != 0 && lanes != 3 && lanes != 8 && lanes != 16) return ov2_set_err(c, OV2_ERR_INVALID, "lanes per keypoint: 0 (auto), 3, 8 or 16");
    c->klt_lanes = lanes;
    return OV2_OK;
}

extern "C" ov2_status ov2_klt_set_yield(ov2_ctx *c, int after, int groups, int pickup)
{
    if (!c) return OV2_ERR_INVALID;
    if (after < 0 || after > 100 || groups < 0 || groups > 20 || pickup < 0)
        return ov2_set_err(c, OV2_ERR_INVALID, "yield: 0 <= after <= 100, 0 <= groups <= 20, pickup >= 0");
    c->klt_yield_after = groups > 0 ? after : 0;
    c->klt_yield_groups = groups;
    c->klt_yield_pickup = pickup;
    return OV2_OK;
}

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
    if ((s = ov2_pyr_wait_ready(c, prev)) != OV2_OK || (s = ov2_pyr_wait_ready(c, cur)) != OV2_OK) return s;
    const int lanes = klt_lanes_for(c, n, win, prev, cur);
    if (lanes != 3 && ((s = ov2_pyr_need_grad(c, prev)) != OV2_OK || (s = ov2_pyr_need_grad(c, cur)) != OV2_OK)) return s;
#define KLT_FB_GL(W, G)                                                                                        \
    OV2_LAUNCH(c, OV2_K_KLT_FB, (klt_fb_kernel<W, G>), dim3((n + klt_map<G>::KPW - 1) / klt_map<G>::KPW), dim3(64), 0, c->stream,    \
               prev->buf->view, cur->buf->view, P, n, reinterpret_cast<const float2 *>(d_kps),                 \
               reinterpret_cast<float2 *>(d_priors), d_status, d_img_idx, d_iters)
#define KLT_FB(W)                                                                                              \
    do {                                                                                                       \
        if (lanes == 8) KLT_FB_GL(W, 8);                                                                          \
        else KLT_FB_GL(W, 16);                                                                                 \
    } while (0)
    switch (win) {
    case 3: KLT_FB(3); break;
    case 5: KLT_FB(5); break;
    case 7: KLT_FB(7); break;
    case 9:
        if (lanes == 3) KLT_FB_GL(9, 3);
        else KLT_FB(9);
        break;
    default: KLT_FB(11); break;
    }
#undef KLT_FB
#undef KLT_FB_GL
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
    return ov2_klt_two_stage_dev(c, prev, cur, win, nlevels_full, max_iter, eps, err_th, fb_th, n, d_kps, d_prior, d_has_prior,
                                 d_img_idx, d_out_xy, d_out_status, d_p3p_req, d_iters, 1);
}

// the two-stage batching shared by VisualFrontEnd::kltTracking (rule33 = 1) and MapManager::stereoMatching (rule33 = 0)
ov2_status ov2_klt_two_stage_dev(ov2_ctx *c, const ov2_pyr *prev, const ov2_pyr *cur, int win, int nlevels_full, int max_iter,
                                 float eps, float err_th, float fb_th, int n, const float *d_kps, const float *d_prior,
                                 const uint8_t *d_has_prior, const int32_t *d_img_idx, float *d_out_xy,
                                 uint8_t *d_out_status, int32_t *d_p3p_req, uint32_t *d_iters, int rule33)
{
    if (!c) return OV2_ERR_INVALID;
    if (n == 0) return OV2_OK;
    if (n < 0 || !d_kps || !d_prior || !d_has_prior || !d_out_xy || !d_out_status)
        return ov2_set_err(c, OV2_ERR_INVALID, "null/negative argument");
    klt_params P;
    ov2_status s = make_params(c, prev, cur, win, nlevels_full, max_iter, eps, err_th, fb_th, &P);
    if (s != OV2_OK) return s;
    P.rule33 = rule33 ? 1 : 0;
    OV2_HIP(c, hipSetDevice(c->device));
    if ((s = ov2_pyr_wait_ready(c, prev)) != OV2_OK || (s = ov2_pyr_wait_ready(c, cur)) != OV2_OK) return s;
    const int lanes = klt_lanes_for(c, n, win, prev, cur);
    if (lanes != 3 && ((s = ov2_pyr_need_grad(c, prev)) != OV2_OK || (s = ov2_pyr_need_grad(c, cur)) != OV2_OK)) return s;
    const int B = prev->buf->batch;
    // scratch: [tallies B x 64 | list lengths (3)] zeroed per call, then the three keypoint lists
    // counters: [tallies B x 64 | list lengths (3)] in the ctx's double buffer (zeroed by the previous call's last kernel);
    // scratch: the three keypoint lists
    const size_t cnt_words = (size_t)B * 64 + 16;
    if (cnt_words > c->klt_counts_words) {
        OV2_HIP(c, hipStreamSynchronize(c->stream));
        for (int i = 0; i < 2; ++i) {
            if (c->klt_counts[i]) OV2_HIP(c, hipFree(c->klt_counts[i]));
            c->klt_counts[i] = nullptr;
        }
        c->klt_counts_words = 0;
        const size_t words = cnt_words + cnt_words / 2;
        for (int i = 0; i < 2; ++i) OV2_HIP(c, hipMalloc((void **)&c->klt_counts[i], words * sizeof(unsigned)));
        c->klt_counts_words = words;
        c->klt_counts_dirty = true;
    }
    if (c->klt_counts_dirty)
        for (int i = 0; i < 2; ++i) OV2_HIP(c, hipMemsetAsync(c->klt_counts[i], 0, c->klt_counts_words * sizeof(unsigned), c->stream));
    c->klt_counts_dirty = true;   // until this call's last kernel is enqueued
    void *scr = nullptr;
    const bool use_yield = lanes == 3 && P.y_after > 0 && P.y_groups > 0;
    const int epoch = ++c->klt_epoch;   // publishes this call's straggler records (the list's memory is reused from call to call)
    s = ov2_scratch(c, 3 * (size_t)n * sizeof(int) + 256, &scr);
    if (s != OV2_OK) return s;
    unsigned *counts = c->klt_counts[c->klt_counts_cur], *live_cnt = counts + (size_t)B * 64;
    unsigned *next_counts = c->klt_counts[c->klt_counts_cur ^ 1];
    const int next_words = (int)c->klt_counts_words;
    int *list_a = (int *)scr, *list_b = list_a + n, *list_c = list_b + n;
    unsigned long long *ylist = nullptr;
    if (use_yield) {   // the straggler records live in a buffer of the context that nothing else writes (see KLT_REC_WORDS)
        const size_t need = (size_t)n * KLT_REC_STRIDE * sizeof(unsigned long long);
        if (need > c->klt_ybuf_bytes) {
            OV2_HIP(c, hipStreamSynchronize(c->stream));
            if (c->klt_ybuf) OV2_HIP(c, hipFree(c->klt_ybuf));
            c->klt_ybuf = nullptr; c->klt_ybuf_bytes = 0;
            const size_t want = need + need / 2;
            OV2_HIP(c, hipMalloc(&c->klt_ybuf, want));
            OV2_HIP(c, hipMemsetAsync(c->klt_ybuf, 0, want, c->stream));
            c->klt_ybuf_bytes = want;
        }
        ylist = (unsigned long long *)c->klt_ybuf;
    }
    const dim3 cgrid((n + 255) / 256);
#define KLT_STAGES_GL(W, G)                                                                                     \
    do {                                                                                                        \
        const dim3 tgrid((n + klt_map<G>::KPW - 1) / klt_map<G>::KPW + 1);   /* groups of list A + groups of list B <= n/KPW + 2 */ \
        OV2_LAUNCH(c, OV2_K_DETECT + 4, klt_compact_kernel, cgrid, dim3(256), 0, c->stream, n, 1, d_has_prior,  \
                   d_out_status, d_iters, list_a, list_b, live_cnt, d_p3p_req, B);                              \
        OV2_LAUNCH(c, OV2_K_KLT_STAGE1, (klt_stage1_kernel<W, G>), dim3(tgrid.x + 1), dim3(64), 0, c->stream,   \
                   prev->buf->view, cur->buf->view, P, n, reinterpret_cast<const float2 *>(d_kps),              \
                   reinterpret_cast<const float2 *>(d_prior), d_img_idx, reinterpret_cast<float2 *>(d_out_xy),  \
                   d_out_status, counts, d_iters, list_a, list_b, live_cnt, list_c, ylist, epoch);             \
        if constexpr (G == 3) {                                                                                 \
            if (use_yield)   /* the stragglers of the first launch, before the re-tracking launch reads its list and tallies */ \
                OV2_LAUNCH(c, OV2_K_KLT_STAGE1, (klt_resume_kernel<W>), tgrid, dim3(64), 0, c->stream,          \
                           prev->buf->view, cur->buf->view, P, n, reinterpret_cast<const float2 *>(d_kps), d_img_idx, \
                           reinterpret_cast<float2 *>(d_out_xy), d_out_status, counts, d_iters, live_cnt, list_c, ylist, epoch); \
        }                                                                                                       \
        OV2_LAUNCH(c, OV2_K_KLT_STAGE2, (klt_stage2_kernel<W, G>), tgrid, dim3(64), 0, c->stream,               \
                   prev->buf->view, cur->buf->view, P, n, reinterpret_cast<const float2 *>(d_kps), d_img_idx,   \
                   reinterpret_cast<float2 *>(d_out_xy), d_out_status, counts, d_p3p_req, d_iters, list_c,      \
                   live_cnt, B, next_counts, next_words);                                                       \
    } while (0)
#define KLT_STAGES(W)                                                                                           \
    do {                                                                                                        \
        if (lanes == 8) KLT_STAGES_GL(W, 8);                                                                       \
        else KLT_STAGES_GL(W, 16);                                                                              \
    } while (0)
    switch (win) {
    case 3: KLT_STAGES(3); break;
    case 5: KLT_STAGES(5); break;
    case 7: KLT_STAGES(7); break;
    case 9:
        if (lanes == 3) KLT_STAGES_GL(9, 3);
        else KLT_STAGES(9);
        break;
    default: KLT_STAGES(11); break;
    }
#undef KLT_STAGES
#undef KLT_STAGES_GL
    OV2_HIP(c, hipGetLastError());
    c->klt_counts_cur ^= 1;
    c->klt_counts_dirty = false;
    return OV2_OK;
}
