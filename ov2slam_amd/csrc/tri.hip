// tri.hip -- two-view triangulation of keypoint pairs + the mapper's acceptance gates, one pair per lane.
//
// Replaces the per-keypoint bodies of Mapper::triangulateStereo (src/mapper.cpp:346-461) and
// Mapper::triangulateTemporal (:191-344) of the reference: MultiViewGeometry::triangulate (src/multi_view_geometry.cpp:
// 53-61 -> opengvTriangulate2 :85-99, OpenGV's mid-point method) or the rectified disparity form (:411-422), depth and
// reprojection gates, world projection, rotation-compensated parallax.  Arithmetic and operation order are those of
// oracle/ov2_oracle_tri.c (f64, contraction off).  HBM-bound streaming work: 72-92 B read, 25-57 B written per pair.
#include <hip/hip_runtime.h>

#include <cstring>

#include "ov2_internal.h"

namespace {

struct tri_args {
    int n, method, G;
    const double *T_ab, *Twc_a;
    const int *grp;
    const double *bv_a, *bv_b;
    const float *unpx_a, *unpx_b;
    double Ka[4], Kb[4];
    float max_err;
    double *pt_a, *wpt, *parallax;
    unsigned char *status;
};

__device__ __forceinline__ void quat_R(const double *T, double R[9])
{
    const double x = T[3], y = T[4], z = T[5], w = T[6];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}

// CameraCalibration::projectCamToImage (src/camera_calibration.cpp:243-252)
__device__ __forceinline__ void project(const double K[4], const double p[3], float &u, float &v)
{
    const double invz = 1. / p[2];
    const double x = p[0] * invz, y = p[1] * invz;
    u = (float)(K[0] * x + K[2]);
    v = (float)(K[1] * y + K[3]);
}

// cv::norm(Point2f - Point2f): float differences, square root in double
__device__ __forceinline__ double norm2f(float ax, float ay, float bx, float by)
{
    const float dx = ax - bx, dy = ay - by;
    return __dsqrt_rn((double)dx * dx + (double)dy * dy);
}

__global__ __launch_bounds__(256) void tri_kernel(tri_args A)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    int g = A.grp ? A.grp[i] : 0;
    g = min(max(g, 0), A.G - 1);   // validated on the host for host arrays; clamped for device-resident ones
    const double *T = A.T_ab + 7 * g;
    double R[9];
    quat_R(T, R);
    const double f1[3] = {A.bv_a[3 * i], A.bv_a[3 * i + 1], A.bv_a[3 * i + 2]};
    const double f2[3] = {A.bv_b[3 * i], A.bv_b[3 * i + 1], A.bv_b[3 * i + 2]};
    const double f2u[3] = {R[0] * f2[0] + R[1] * f2[1] + R[2] * f2[2], R[3] * f2[0] + R[4] * f2[1] + R[5] * f2[2],
                           R[6] * f2[0] + R[7] * f2[1] + R[8] * f2[2]};
    const float ua = A.unpx_a[2 * i], va = A.unpx_a[2 * i + 1], ub = A.unpx_b[2 * i], vb = A.unpx_b[2 * i + 1];
    if (A.parallax) {
        float ru, rv;
        project(A.Kb, f2u, ru, rv);
        A.parallax[i] = norm2f(ua, va, ru, rv);
    }
    double X[3];
    if (A.method == OV2_TRI_RECTIFIED) {
        const float disp = ua - ub;
        if (disp < 0.f) {
            A.status[i] = OV2_TRI_NEG_DISP;
            A.pt_a[3 * i] = A.pt_a[3 * i + 1] = A.pt_a[3 * i + 2] = 0.0;
            if (A.wpt) A.wpt[3 * i] = A.wpt[3 * i + 1] = A.wpt[3 * i + 2] = 0.0;
            return;
        }
        const double base = __dsqrt_rn(T[0] * T[0] + T[1] * T[1] + T[2] * T[2]);
        const float z = (float)(A.Ka[0] * base / fabs((double)disp));
        X[0] = (double)z * ((double)ua / A.Ka[0] - A.Ka[2] / A.Ka[0]);
        X[1] = (double)z * ((double)va / A.Ka[1] - A.Ka[3] / A.Ka[1]);
        X[2] = (double)z;
    } else {
        const double a00 = f1[0] * f1[0] + f1[1] * f1[1] + f1[2] * f1[2];
        const double a10 = f1[0] * f2u[0] + f1[1] * f2u[1] + f1[2] * f2u[2];
        const double a01 = -a10;
        const double a11 = -(f2u[0] * f2u[0] + f2u[1] * f2u[1] + f2u[2] * f2u[2]);
        const double b0 = T[0] * f1[0] + T[1] * f1[1] + T[2] * f1[2];
        const double b1 = T[0] * f2u[0] + T[1] * f2u[1] + T[2] * f2u[2];
        const double invdet = 1. / (a00 * a11 - a01 * a10);
        const double l0 = (a11 * invdet) * b0 + (-a01 * invdet) * b1;
        const double l1 = (-a10 * invdet) * b0 + (a00 * invdet) * b1;
#pragma unroll
        for (int k = 0; k < 3; ++k) X[k] = (l0 * f1[k] + (T[k] + l1 * f2u[k])) / 2.;
    }
    const double d[3] = {X[0] - T[0], X[1] - T[1], X[2] - T[2]};
    const double Xb[3] = {R[0] * d[0] + R[3] * d[1] + R[6] * d[2], R[1] * d[0] + R[4] * d[1] + R[7] * d[2],
                          R[2] * d[0] + R[5] * d[1] + R[8] * d[2]};
    int st = OV2_TRI_OK;
    if (X[2] < 0.1 || Xb[2] < 0.1) st = OV2_TRI_BEHIND;
    else {
        float pu, pv, qu, qv;
        project(A.Ka, X, pu, pv);
        project(A.Kb, Xb, qu, qv);
        const float ldist = (float)norm2f(pu, pv, ua, va), rdist = (float)norm2f(qu, qv, ub, vb);
        if (ldist > A.max_err || rdist > A.max_err) st = OV2_TRI_REPROJ;
    }
    A.status[i] = (unsigned char)st;
#pragma unroll
    for (int k = 0; k < 3; ++k) A.pt_a[3 * i + k] = X[k];
    if (A.wpt) {
        const double *W = A.Twc_a + 7 * g;
        double Rw[9];
        quat_R(W, Rw);
#pragma unroll
        for (int k = 0; k < 3; ++k) A.wpt[3 * i + k] = Rw[3 * k] * X[0] + Rw[3 * k + 1] * X[1] + Rw[3 * k + 2] * X[2] + W[k];
    }
}

}  // namespace

extern "C" ov2_status ov2_triangulate_pairs_dev(ov2_ctx *c, int n, int method, int G, const double *d_T_ab, const double *d_Twc_a,
                                                const int32_t *d_grp, const double *d_bv_a, const double *d_bv_b,
                                                const float *d_unpx_a, const float *d_unpx_b, const double *K_a,
                                                const double *K_b, float max_reproj_err, double *d_pt_a, double *d_wpt,
                                                double *d_parallax, uint8_t *d_status)
{
    if (!c) return OV2_ERR_INVALID;
    if (n < 0 || G < 1 || (method != OV2_TRI_MIDPOINT && method != OV2_TRI_RECTIFIED))
        return ov2_set_err(c, OV2_ERR_INVALID, "ov2_triangulate_pairs: n=%d G=%d method=%d", n, G, method);
    if (n == 0) return OV2_OK;
    if (!d_T_ab || !d_bv_a || !d_bv_b || !d_unpx_a || !d_unpx_b || !K_a || !K_b || !d_pt_a || !d_status || (d_wpt && !d_Twc_a))
        return ov2_set_err(c, OV2_ERR_INVALID, "ov2_triangulate_pairs: null argument");
    OV2_HIP(c, hipSetDevice(c->device));
    tri_args A;
    A.n = n; A.method = method; A.G = G; A.T_ab = d_T_ab; A.Twc_a = d_Twc_a; A.grp = d_grp; A.bv_a = d_bv_a; A.bv_b = d_bv_b;
    A.unpx_a = d_unpx_a; A.unpx_b = d_unpx_b;
    for (int k = 0; k < 4; ++k) { A.Ka[k] = K_a[k]; A.Kb[k] = K_b[k]; }
    A.max_err = max_reproj_err; A.pt_a = d_pt_a; A.wpt = d_wpt; A.parallax = d_parallax; A.status = d_status;
    OV2_LAUNCH(c, OV2_K_MAP + 1, tri_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, A);
    return OV2_OK;
}

extern "C" ov2_status ov2_triangulate_pairs(ov2_ctx *c, int n, int method, int G, const double *T_ab, const double *Twc_a,
                                            const int32_t *grp, const double *bv_a, const double *bv_b, const float *unpx_a,
                                            const float *unpx_b, const double *K_a, const double *K_b, float max_reproj_err,
                                            double *pt_a, double *wpt, double *parallax, uint8_t *status)
{
    if (!c) return OV2_ERR_INVALID;
    if (n < 0 || G < 1) return ov2_set_err(c, OV2_ERR_INVALID, "ov2_triangulate_pairs: n=%d G=%d", n, G);
    if (n == 0) return OV2_OK;
    if (!T_ab || !bv_a || !bv_b || !unpx_a || !unpx_b || !K_a || !K_b || !pt_a || !status || (wpt && !Twc_a))
        return ov2_set_err(c, OV2_ERR_INVALID, "ov2_triangulate_pairs: null argument");
    if (grp)
        for (int i = 0; i < n; ++i)
            if (grp[i] < 0 || grp[i] >= G) return ov2_set_err(c, OV2_ERR_INVALID, "pose-pair index %d of pair %d outside [0, %d)", grp[i], i, G);
    OV2_HIP(c, hipSetDevice(c->device));
    // one pinned block: inputs in, outputs back
    const size_t N = (size_t)n;
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 15) & ~(size_t)15; return o; };
    const size_t oT = carve(56 * (size_t)G), oW = carve(Twc_a ? 56 * (size_t)G : 0), oG = carve(grp ? 4 * N : 0), oA = carve(24 * N),
                 oB = carve(24 * N), oUa = carve(8 * N), oUb = carve(8 * N), in_end = off;
    const size_t oP = carve(24 * N), oX = carve(wpt ? 24 * N : 0), oL = carve(parallax ? 8 * N : 0), oS = carve(N);
    void *hv, *dv;
    ov2_status s = ov2_staging(c, off, &hv, &dv);
    if (s != OV2_OK) return s;
    unsigned char *h = (unsigned char *)hv, *d = (unsigned char *)dv;
    memcpy(h + oT, T_ab, 56 * (size_t)G);
    if (Twc_a) memcpy(h + oW, Twc_a, 56 * (size_t)G);
    if (grp) memcpy(h + oG, grp, 4 * N);
    memcpy(h + oA, bv_a, 24 * N); memcpy(h + oB, bv_b, 24 * N); memcpy(h + oUa, unpx_a, 8 * N); memcpy(h + oUb, unpx_b, 8 * N);
    OV2_HIP(c, hipMemcpyAsync(d, h, in_end, hipMemcpyHostToDevice, c->stream));
    s = ov2_triangulate_pairs_dev(c, n, method, G, (const double *)(d + oT), Twc_a ? (const double *)(d + oW) : nullptr,
                                  grp ? (const int32_t *)(d + oG) : nullptr, (const double *)(d + oA), (const double *)(d + oB),
                                  (const float *)(d + oUa), (const float *)(d + oUb), K_a, K_b, max_reproj_err, (double *)(d + oP),
                                  wpt ? (double *)(d + oX) : nullptr, parallax ? (double *)(d + oL) : nullptr, d + oS);
    if (s != OV2_OK) return s;
    OV2_HIP(c, hipMemcpyAsync(h + oP, d + oP, off - oP, hipMemcpyDeviceToHost, c->stream));
    OV2_HIP(c, hipStreamSynchronize(c->stream));
    memcpy(pt_a, h + oP, 24 * N);
    if (wpt) memcpy(wpt, h + oX, 24 * N);
    if (parallax) memcpy(parallax, h + oL, 8 * N);
    memcpy(status, h + oS, N);
    return OV2_OK;
}
