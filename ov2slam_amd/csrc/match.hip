// match.hip -- keyframe descriptors and map matching (first half):
//   brief_kernel   FeatureExtractor::describeBRIEF src/feature_extractor.cpp:224-285 (cv::xfeatures2d::BriefDescriptorExtractor:
//                  32 bytes, patch 48, 9 x 9 box sums at the rounded keypoint) with a caller-supplied test table
//   match_*        Mapper::matchToMap src/mapper.cpp:576-774 on flat arrays (ov2_match_input)
// Integer box sums and Hamming distances => bit-exact against the oracle; the float gates are evaluated in its order.
#include "ov2_internal.h"
#include "ov2_cam.h"

namespace {

enum { K_BRIEF = OV2_K_MAP + 4, K_MATCH = OV2_K_MAP + 5 };

// one wave per keypoint: the 57 x 57 pixels the 256 tests can touch are staged in LDS, lane l evaluates tests l, l + 64, ...
__global__ __launch_bounds__(64) void brief_kernel(ov2_pyr_view pv, int n, const float2 *__restrict__ pts,
                                                   const int *__restrict__ img_idx, const signed char *__restrict__ pattern,
                                                   unsigned char *__restrict__ desc, unsigned char *__restrict__ valid, int b0)
{
    __shared__ unsigned char patch[57 * 60];
    __shared__ unsigned bits[8];
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    const ov2_level_desc &L = pv.lv[0];
    const int w = L.w, h = L.h, b = img_idx ? img_idx[i] : b0;
    const float x = pts[i].x, y = pts[i].y;
    const int border = 28;   // PATCH_SIZE / 2 + KERNEL_SIZE / 2
    const bool ok = (w > 2 * border && h > 2 * border) && x >= (float)border && x < (float)(w - border) && y >= (float)border &&
                    y < (float)(h - border);
    if (!ok) {
        if (lane < 32) desc[(size_t)i * 32 + lane] = 0;
        if (lane == 0) valid[i] = 0;
        return;
    }
    const int px = (int)((double)x + 0.5), py = (int)((double)y + 0.5);
    const unsigned char *img = pv.base + L.img_off + L.img_bstride * b + (size_t)pv.pad * L.istride + OV2_LM;
    for (int k = lane; k < 57 * 57; k += 64) {
        const int r = k / 57, c = k - r * 57;
        patch[r * 60 + c] = img[(size_t)(py - 28 + r) * L.istride + (px - 28 + c)];
    }
    if (lane < 8) bits[lane] = 0;
    __syncthreads();
    for (int k = lane; k < 256; k += 64) {
        const signed char *t = pattern + 4 * k;
        int a = 0, bsum = 0;
        const int ya = 28 + t[0], xa = 28 + t[1], yb = 28 + t[2], xb = 28 + t[3];
        for (int dy = -4; dy <= 4; ++dy)
            for (int dx = -4; dx <= 4; ++dx) {
                a += patch[(ya + dy) * 60 + xa + dx];
                bsum += patch[(yb + dy) * 60 + xb + dx];
            }
        if (a < bsum) atomicOr(&bits[k >> 5], 1u << (k & 31));
    }
    __syncthreads();
    if (lane < 32) {   // byte j holds tests 8 j .. 8 j + 7, the first one in its most significant bit
        const unsigned wv = bits[lane >> 2] >> ((lane & 3) * 8);
        unsigned char o = 0;
        for (int q = 0; q < 8; ++q) o |= (unsigned char)(((wv >> q) & 1u) << (7 - q));
        desc[(size_t)i * 32 + lane] = o;
    }
    if (lane == 0) valid[i] = 1;
}

struct match_dev {   // device twin of ov2_match_input
    double Twc[7], K[4];
    ov2_cam_model cam;   // Frame::projWorldToImageDist (model 0: the pinhole map with K)
    int img_w, img_h, cell, nb3dkps, n_kp, n_cand, n_kf, nbw, ncells;
    const float2 *kp_px; const int *kp_desc_ptr; const unsigned char *kp_descs; const int *kp_kf_ptr, *kp_kfids; const float2 *kp_kf_px;
    const int *grid_ptr, *grid_kp;
    const double *cand_wpt; const int *cand_desc_ptr; const unsigned char *cand_descs; const int *cand_kf_ptr, *cand_kfids;
    const double *kf_Twc;
};

__device__ inline void world_to_cam(const double *Twc, const double *p, double c[3])
{
    double x = Twc[3], y = Twc[4], z = Twc[5], w = Twc[6];
    const double n = sqrt(x * x + y * y + z * z + w * w);
    x /= n; y /= n; z /= n; w /= n;
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    const double R[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
    const double d[3] = {p[0] - Twc[0], p[1] - Twc[1], p[2] - Twc[2]};
    for (int r = 0; r < 3; ++r) c[r] = R[r] * d[0] + R[3 + r] * d[1] + R[6 + r] * d[2];
}

__device__ inline int hamming32(const unsigned char *a, const unsigned char *b)
{
    const unsigned long long *pa = reinterpret_cast<const unsigned long long *>(a), *pb = reinterpret_cast<const unsigned long long *>(b);
    return __popcll(pa[0] ^ pb[0]) + __popcll(pa[1] ^ pb[1]) + __popcll(pa[2] ^ pb[2]) + __popcll(pa[3] ^ pb[3]);
}

// one wave per candidate map point: gates, then the neighbouring keypoints 64 at a time (one per lane: pixel gate,
// co-observation test, mean reprojection distance, minimum Hamming distance over the two descriptor sets), and the
// best / second-best bookkeeping replayed in the reference's order (it breaks ties by position).  The winner goes to
// the keypoint's slot by a 64-bit atomicMin on (distance, ~candidate index): smallest distance, later candidate on ties.
__global__ __launch_bounds__(64) void match_cand_kernel(match_dev M, float dmaxpxdist, float mindist, float view_th,
                                                        unsigned long long *__restrict__ kp_best)
{
    const int c = blockIdx.x, lane = threadIdx.x;
    if (c >= M.n_cand) return;
    const int cd0 = M.cand_desc_ptr[c], cd1 = M.cand_desc_ptr[c + 1];
    if (cd1 == cd0) return;
    const double *wpt = M.cand_wpt + 3 * (size_t)c;
    double campt[3];
    world_to_cam(M.Twc, wpt, campt);
    if (campt[2] < 0.1) return;
    const float view_angle = (float)(campt[2] / sqrt(campt[0] * campt[0] + campt[1] * campt[1] + campt[2] * campt[2]));
    if (fabsf(view_angle) < view_th) return;
    float px, py;
    ov2_cam_project_dist(M.cam, campt, px, py);
    if (!(px >= 0 && py >= 0 && px < (float)M.img_w && py < (float)M.img_h)) return;
    int bestid = -1, secid = -1;
    float bestdist = mindist, secdist = mindist;
    const int rkp = (int)floorf(py / (float)M.cell), ckp = (int)floorf(px / (float)M.cell);
    for (int cellk = 0; cellk < 4; ++cellk) {
        const int r = rkp - 1 + (cellk >> 1), cc = ckp - 1 + (cellk & 1);
        const int idx = r * M.nbw + cc;
        if (r < 0 || cc < 0 || idx >= M.ncells) continue;
        const int g0 = M.grid_ptr[idx], g1 = M.grid_ptr[idx + 1];
        for (int base = g0; base < g1; base += 64) {
            const int g = base + lane;
            float dist = -1.f;   // < 0: this keypoint is not a candidate
            int k = -1;
            if (g < g1) {
                k = M.grid_kp[g];
                const float dx = px - M.kp_px[k].x, dy = py - M.kp_px[k].y;
                const float pxdist = (float)sqrt((double)dx * dx + (double)dy * dy);
                const int kd0 = M.kp_desc_ptr[k], kd1 = M.kp_desc_ptr[k + 1];
                bool cand_ok = !(pxdist > dmaxpxdist) && kd1 > kd0;
                if (cand_ok) {
                    int a = M.cand_kf_ptr[c], a1 = M.cand_kf_ptr[c + 1], b = M.kp_kf_ptr[k], b1 = M.kp_kf_ptr[k + 1];
                    while (a < a1 && b < b1) {
                        const int ka = M.cand_kfids[a], kb = M.kp_kfids[b];
                        if (ka == kb) { cand_ok = false; break; }
                        if (ka < kb) ++a; else ++b;
                    }
                }
                if (cand_ok) {
                    float coprojpx = 0.f;
                    int nbcokp = 0;
                    for (int e = M.kp_kf_ptr[k]; e < M.kp_kf_ptr[k + 1]; ++e) {
                        const int kfid = M.kp_kfids[e];
                        if (kfid < 0 || kfid >= M.n_kf) continue;
                        double cp[3];
                        world_to_cam(M.kf_Twc + 7 * (size_t)kfid, wpt, cp);
                        float qx, qy;
                        ov2_cam_project_dist(M.cam, cp, qx, qy);
                        const float ex = M.kp_kf_px[e].x - qx, ey = M.kp_kf_px[e].y - qy;
                        coprojpx = (float)((double)coprojpx + sqrt((double)ex * ex + (double)ey * ey));
                        ++nbcokp;
                    }
                    if (coprojpx / (float)nbcokp > dmaxpxdist) cand_ok = false;
                }
                if (cand_ok) {
                    float dmin = 1000.f;
                    for (int i = cd0; i < cd1; ++i)
                        for (int j = kd0; j < kd1; ++j) {
                            const float hd = (float)hamming32(M.cand_descs + 32 * (size_t)i, M.kp_descs + 32 * (size_t)j);
                            if (hd < dmin) dmin = hd;
                        }
                    dist = dmin;
                }
            }
            // replay in keypoint order (wave-uniform): src/mapper.cpp:723-733
            const int cnt = min(64, g1 - base);
            for (int q = 0; q < cnt; ++q) {
                const float dq = __shfl(dist, q);
                const int kq = __shfl(k, q);
                if (dq < 0.f) continue;
                if (dq <= bestdist) { secdist = bestdist; secid = bestid; bestdist = dq; bestid = kq; }
                else if (dq <= secdist) { secdist = dq; secid = kq; }
            }
        }
    }
    if (bestid != -1 && secid != -1 && 0.9 * (double)secdist < (double)bestdist) bestid = -1;
    if (bestid < 0) return;
    if (lane == 0)   // Hamming distances are small integers: exact in the key
        atomicMin(&kp_best[bestid], ((unsigned long long)(unsigned)(int)bestdist << 32) | (unsigned long long)(0xffffffffu - (unsigned)c));
}

__global__ void match_out_kernel(int n_kp, const unsigned long long *__restrict__ kp_best, int *__restrict__ match_cand,
                                 float *__restrict__ match_dist)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_kp) return;
    const unsigned long long v = kp_best[k];
    if (v == ~0ull) { match_cand[k] = -1; match_dist[k] = 0.f; }
    else { match_cand[k] = (int)(0xffffffffu - (unsigned)(v & 0xffffffffull)); match_dist[k] = (float)(unsigned)(v >> 32); }
}

}  // namespace

extern "C" ov2_status ov2_describe_brief_dev(ov2_ctx *c, const ov2_pyr *pyr, int n, const float *d_pts_xy, const int32_t *d_img_idx,
                                             const int8_t *d_pattern, uint8_t *d_desc, uint8_t *d_valid)
{
    if (!c) return OV2_ERR_INVALID;
    if (n == 0) return OV2_OK;
    if (n < 0 || !pyr || !d_pts_xy || !d_pattern || !d_desc || !d_valid) return ov2_set_err(c, OV2_ERR_INVALID, "null/negative argument");
    OV2_HIP(c, hipSetDevice(c->device));
    ov2_status s = ov2_pyr_wait_ready(c, pyr);
    if (s != OV2_OK) return s;
    OV2_LAUNCH(c, K_BRIEF, brief_kernel, dim3(n), dim3(64), 0, c->stream, pyr->buf->view, n, reinterpret_cast<const float2 *>(d_pts_xy),
               d_img_idx, reinterpret_cast<const signed char *>(d_pattern), d_desc, d_valid, 0);
    OV2_HIP(c, hipGetLastError());
    return OV2_OK;
}

extern "C" ov2_status ov2_describe_brief(ov2_ctx *c, const ov2_pyr *pyr, int b, int n, const float *pts_xy, const int8_t *pattern,
                                         uint8_t *desc, uint8_t *valid)
{
    if (!c) return OV2_ERR_INVALID;
    if (n == 0) return OV2_OK;
    if (n < 0 || !pyr || !pts_xy || !pattern || !desc || !valid) return ov2_set_err(c, OV2_ERR_INVALID, "null/negative argument");
    if (b < 0 || b >= pyr->buf->batch) return ov2_set_err(c, OV2_ERR_INVALID, "image %d outside the pyramid batch", b);
    for (int k = 0; k < 1024; ++k)
        if (pattern[k] < -24 || pattern[k] > 24) return ov2_set_err(c, OV2_ERR_INVALID, "BRIEF test offset %d outside [-24, 24]", (int)pattern[k]);
    void *hs = nullptr, *ds = nullptr;
    const size_t nb = (size_t)n * 8, need = nb + 1024 + (size_t)n * 33 + 64;
    ov2_status s = ov2_staging(c, need, &hs, &ds);
    if (s != OV2_OK) return s;
    char *h = (char *)hs, *d = (char *)ds;
    memcpy(h, pts_xy, nb);
    memcpy(h + nb, pattern, 1024);
    OV2_HIP(c, hipSetDevice(c->device));
    OV2_HIP(c, hipMemcpyAsync(d, h, nb + 1024, hipMemcpyHostToDevice, c->stream));
    if ((s = ov2_pyr_wait_ready(c, pyr)) != OV2_OK) return s;
    uint8_t *d_desc = (uint8_t *)(d + nb + 1024), *d_valid = d_desc + (size_t)n * 32;
    OV2_LAUNCH(c, K_BRIEF, brief_kernel, dim3(n), dim3(64), 0, c->stream, pyr->buf->view, n, reinterpret_cast<const float2 *>(d), nullptr,
               reinterpret_cast<const signed char *>(d + nb), d_desc, d_valid, b);
    OV2_HIP(c, hipMemcpyAsync(h + nb + 1024, d_desc, (size_t)n * 33, hipMemcpyDeviceToHost, c->stream));
    OV2_HIP(c, hipStreamSynchronize(c->stream));
    memcpy(desc, h + nb + 1024, (size_t)n * 32);
    memcpy(valid, h + nb + 1024 + (size_t)n * 32, (size_t)n);
    return OV2_OK;
}

extern "C" ov2_status ov2_match_to_map(ov2_ctx *c, const ov2_match_input *in, float fmaxprojerr, float fdistratio,
                                       int32_t *match_cand, float *match_dist)
{
    if (!c) return OV2_ERR_INVALID;
    if (!in || !match_cand || !match_dist || in->n_kp < 0 || in->n_cand < 0 || in->cell <= 0 || in->img_w <= 0 || in->img_h <= 0)
        return ov2_set_err(c, OV2_ERR_INVALID, "bad ov2_match_input");
    const int nkp = in->n_kp, nc = in->n_cand;
    for (int k = 0; k < nkp; ++k) { match_cand[k] = -1; match_dist[k] = 0.f; }
    if (nkp == 0 || nc == 0) return OV2_OK;                                  // src/mapper.cpp:580-583
    const int nbw = (int)ceilf((float)in->img_w / (float)in->cell), nbh = (int)ceilf((float)in->img_h / (float)in->cell);
    const int ncells = nbw * nbh;
    const size_t n_kpd = (size_t)in->kp_desc_ptr[nkp], n_kpk = (size_t)in->kp_kf_ptr[nkp], n_g = (size_t)in->grid_ptr[ncells];
    const size_t n_cd = (size_t)in->cand_desc_ptr[nc], n_ck = (size_t)in->cand_kf_ptr[nc];
    // one staging block, every array 16-byte aligned
    size_t off = 0;
    auto place = [&](size_t bytes) { const size_t o = off; off += (bytes + 15) / 16 * 16; return o; };
    const size_t o_kpx = place(nkp * 8), o_kdp = place((nkp + 1) * 4), o_kd = place(n_kpd * 32), o_kkp = place((nkp + 1) * 4),
                 o_kk = place(n_kpk * 4), o_kkx = place(n_kpk * 8), o_gp = place((ncells + 1) * 4), o_gk = place(n_g * 4),
                 o_cw = place((size_t)nc * 24), o_cdp = place((nc + 1) * 4), o_cd = place(n_cd * 32), o_ckp = place((nc + 1) * 4),
                 o_ck = place(n_ck * 4), o_kf = place((size_t)std::max(in->n_kf, 0) * 56), o_in = off;
    const size_t o_best = place((size_t)nkp * 8), o_mc = place((size_t)nkp * 4), o_md = place((size_t)nkp * 4);
    void *hs = nullptr, *ds = nullptr;
    ov2_status s = ov2_staging(c, off + 64, &hs, &ds);
    if (s != OV2_OK) return s;
    char *h = (char *)hs, *d = (char *)ds;
    memcpy(h + o_kpx, in->kp_px, nkp * 8); memcpy(h + o_kdp, in->kp_desc_ptr, (nkp + 1) * 4); memcpy(h + o_kd, in->kp_descs, n_kpd * 32);
    memcpy(h + o_kkp, in->kp_kf_ptr, (nkp + 1) * 4); memcpy(h + o_kk, in->kp_kfids, n_kpk * 4); memcpy(h + o_kkx, in->kp_kf_px, n_kpk * 8);
    memcpy(h + o_gp, in->grid_ptr, (ncells + 1) * 4); memcpy(h + o_gk, in->grid_kp, n_g * 4);
    memcpy(h + o_cw, in->cand_wpt, (size_t)nc * 24); memcpy(h + o_cdp, in->cand_desc_ptr, (nc + 1) * 4); memcpy(h + o_cd, in->cand_descs, n_cd * 32);
    memcpy(h + o_ckp, in->cand_kf_ptr, (nc + 1) * 4); memcpy(h + o_ck, in->cand_kfids, n_ck * 4);
    if (in->n_kf > 0) memcpy(h + o_kf, in->kf_Twc, (size_t)in->n_kf * 56);
    OV2_HIP(c, hipSetDevice(c->device));
    OV2_HIP(c, hipMemcpyAsync(d, h, o_in, hipMemcpyHostToDevice, c->stream));
    OV2_HIP(c, hipMemsetAsync(d + o_best, 0xff, (size_t)nkp * 8, c->stream));
    match_dev M;
    for (int i = 0; i < 7; ++i) M.Twc[i] = in->Twc[i];
    for (int i = 0; i < 4; ++i) M.K[i] = in->K[i];
    M.cam = ov2_cam_normalised(in->cam);
    for (int i = 0; i < 4; ++i) M.cam.K[i] = in->K[i];   // one set of intrinsics: the frame's
    if (M.cam.model < 0 || M.cam.model > 2) return ov2_set_err(c, OV2_ERR_INVALID, "unknown lens model %d", M.cam.model);
    M.img_w = in->img_w; M.img_h = in->img_h; M.cell = in->cell; M.nb3dkps = in->nb3dkps; M.n_kp = nkp; M.n_cand = nc; M.n_kf = in->n_kf;
    M.nbw = nbw; M.ncells = ncells;
    M.kp_px = (const float2 *)(d + o_kpx); M.kp_desc_ptr = (const int *)(d + o_kdp); M.kp_descs = (const unsigned char *)(d + o_kd);
    M.kp_kf_ptr = (const int *)(d + o_kkp); M.kp_kfids = (const int *)(d + o_kk); M.kp_kf_px = (const float2 *)(d + o_kkx);
    M.grid_ptr = (const int *)(d + o_gp); M.grid_kp = (const int *)(d + o_gk);
    M.cand_wpt = (const double *)(d + o_cw); M.cand_desc_ptr = (const int *)(d + o_cdp); M.cand_descs = (const unsigned char *)(d + o_cd);
    M.cand_kf_ptr = (const int *)(d + o_ckp); M.cand_kfids = (const int *)(d + o_ck); M.kf_Twc = (const double *)(d + o_kf);
    // thresholds exactly as the reference forms them (:586-603, :651)
    const float vfov = (float)(0.5 * in->img_h / in->K[1]), hfov = (float)(0.5 * in->img_w / in->K[0]);
    const float maxradfov = hfov > vfov ? atanf(hfov) : atanf(vfov);
    const float view_th = cosf(maxradfov);
    float dmaxpxdist = fmaxprojerr;
    if (in->nb3dkps < 30) dmaxpxdist *= 2.f;
    const float mindist = (float)((double)(32.f * fdistratio) * 8.);
    OV2_LAUNCH(c, K_MATCH, match_cand_kernel, dim3(nc), dim3(64), 0, c->stream, M, dmaxpxdist, mindist, view_th,
               (unsigned long long *)(d + o_best));
    OV2_LAUNCH(c, K_MATCH, match_out_kernel, dim3((nkp + 255) / 256), dim3(256), 0, c->stream, nkp, (const unsigned long long *)(d + o_best),
               (int *)(d + o_mc), (float *)(d + o_md));
    OV2_HIP(c, hipMemcpyAsync(h + o_mc, d + o_mc, (o_md - o_mc) + (size_t)nkp * 4, hipMemcpyDeviceToHost, c->stream));
    OV2_HIP(c, hipStreamSynchronize(c->stream));
    memcpy(match_cand, h + o_mc, (size_t)nkp * 4);
    memcpy(match_dist, h + o_md, (size_t)nkp * 4);
    return OV2_OK;
}
