/*
 * ov2_oracle_tri.c -- CPU restatement of the two-view triangulation + acceptance gates of the mapper
 * (TEST INFRASTRUCTURE ONLY, see ov2_oracle.h).
 *
 * Reference (/root/reference): Mapper::triangulateStereo src/mapper.cpp:346-461, Mapper::triangulateTemporal :191-344,
 * Mapper::computeTriangulation :463-467 -> MultiViewGeometry::triangulate src/multi_view_geometry.cpp:53-61 ->
 * opengvTriangulate2 :85-99 -> opengv::triangulation::triangulate2.  OpenGV is an un-vendored dependency
 * (CMakeLists.txt, find_package(opengv)); its triangulate2 is the published mid-point method: with f2' = R12 f2,
 *   A = [ f1.f1  -f1.f2' ; f1.f2'  -f2'.f2' ],  b = [ t12.f1 ; t12.f2' ],  lambda = A^-1 b,
 *   X = (lambda0 f1 + t12 + lambda1 f2') / 2.
 * It is restated here from that definition; the reference holds no fixture for it => parity unpinned, pinned instead
 * by tests/test_oracle_tri.py (exact recovery of noise-free points, least-squares closest approach, gates).
 * Projection: CameraCalibration::projectCamToImage src/camera_calibration.cpp:243-252 (double math, float result);
 * distances: cv::norm(Point2f) = sqrt in double of float differences, stored in a float (:438-439, :326-327).
 */
#include "ov2_oracle.h"

#include <math.h>

static void quat_Rt(const double *T, double R[9])
{
    const double x = T[3], y = T[4], z = T[5], w = T[6];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}

static void project(const double K[4], const double p[3], float px[2])
{
    const double invz = 1. / p[2];
    const double x = p[0] * invz, y = p[1] * invz;
    px[0] = (float)(K[0] * x + K[2]);
    px[1] = (float)(K[1] * y + K[3]);
}

static double norm2f(float ax, float ay, float bx, float by)
{
    const float dx = ax - bx, dy = ay - by;
    return sqrt((double)dx * dx + (double)dy * dy);
}

int ov2o_triangulate_pairs(int n, int method, int G, const double *T_ab, const double *Twc_a, const int *grp,
                           const double *bv_a, const double *bv_b, const float *unpx_a, const float *unpx_b,
                           const double *K_a, const double *K_b, float max_reproj_err, double *pt_a, double *wpt,
                           double *parallax, unsigned char *status)
{
    for (int i = 0; i < n; ++i) {
        const int g = grp ? grp[i] : 0;
        if (g < 0 || g >= G) return -1;
        const double *T = T_ab + 7 * g;
        double R[9];
        quat_Rt(T, R);
        const double *f1 = bv_a + 3 * i, *f2 = bv_b + 3 * i;
        const double f2u[3] = {R[0] * f2[0] + R[1] * f2[1] + R[2] * f2[2], R[3] * f2[0] + R[4] * f2[1] + R[5] * f2[2],
                               R[6] * f2[0] + R[7] * f2[1] + R[8] * f2[2]};
        double X[3];
        int st = 0;
        if (parallax) {   /* rotation-compensated parallax, :301-302 */
            float rp[2];
            project(K_b, f2u, rp);
            parallax[i] = norm2f(unpx_a[2 * i], unpx_a[2 * i + 1], rp[0], rp[1]);
        }
        if (method == 1) {   /* rectified pair, :411-422 */
            const float disp = unpx_a[2 * i] - unpx_b[2 * i];
            if (disp < 0.f) { status[i] = 3; pt_a[3 * i] = pt_a[3 * i + 1] = pt_a[3 * i + 2] = 0; if (wpt) wpt[3 * i] = wpt[3 * i + 1] = wpt[3 * i + 2] = 0; continue; }
            const double base = sqrt(T[0] * T[0] + T[1] * T[1] + T[2] * T[2]);
            const float z = (float)(K_a[0] * base / fabs((double)disp));
            const double vx = (double)unpx_a[2 * i], vy = (double)unpx_a[2 * i + 1];
            /* iK = K^-1 = [1/fx 0 -cx/fx; 0 1/fy -cy/fy; 0 0 1] */
            X[0] = (double)z * (vx / K_a[0] - K_a[2] / K_a[0]);
            X[1] = (double)z * (vy / K_a[1] - K_a[3] / K_a[1]);
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
            for (int k = 0; k < 3; ++k) X[k] = (l0 * f1[k] + (T[k] + l1 * f2u[k])) / 2.;
        }
        /* view b: T_ab^-1 * X = R'(X - t) */
        const double d[3] = {X[0] - T[0], X[1] - T[1], X[2] - T[2]};
        const double Xb[3] = {R[0] * d[0] + R[3] * d[1] + R[6] * d[2], R[1] * d[0] + R[4] * d[1] + R[7] * d[2],
                              R[2] * d[0] + R[5] * d[1] + R[8] * d[2]};
        if (X[2] < 0.1 || Xb[2] < 0.1) st = 1;   /* :428, :316 */
        else {
            float pa[2], pb[2];
            project(K_a, X, pa);
            project(K_b, Xb, pb);
            const float ldist = (float)norm2f(pa[0], pa[1], unpx_a[2 * i], unpx_a[2 * i + 1]);
            const float rdist = (float)norm2f(pb[0], pb[1], unpx_b[2 * i], unpx_b[2 * i + 1]);
            if (ldist > max_reproj_err || rdist > max_reproj_err) st = 2;   /* :441-446, :330-336 */
        }
        status[i] = (unsigned char)st;
        for (int k = 0; k < 3; ++k) pt_a[3 * i + k] = X[k];
        if (wpt) {   /* Frame::projCamToWorld src/frame.cpp:802-809 */
            double Rw[9];
            const double *W = Twc_a + 7 * g;
            quat_Rt(W, Rw);
            for (int k = 0; k < 3; ++k) wpt[3 * i + k] = Rw[3 * k] * X[0] + Rw[3 * k + 1] * X[1] + Rw[3 * k + 2] * X[2] + W[k];
        }
    }
    return 0;
}
