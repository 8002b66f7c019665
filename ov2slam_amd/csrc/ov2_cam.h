// ov2_cam.h -- CameraCalibration::undistortImagePoint / projectCamToImageDist (src/camera_calibration.cpp:254-332) for the
// kernels: the arithmetic of ov2::CameraCalibration in the host mirror (host/ov2_host.cpp), which restates OpenCV's
// undistortPoints (five fixed-point sweeps), projectPoints, fisheye::undistortPoints (Newton on theta) and
// fisheye::distortPoints from their published definitions (OpenCV is not vendored by the reference: parity unpinned).
#pragma once
#include "../../include/ov2slam_hip.h"

__host__ __device__ inline void ov2_cam_undistort(const ov2_cam_model &c, float u_, float v_, float &ou, float &ov)
{
    if (c.model == 0 || c.n_coeffs <= 0) { ou = u_; ov = v_; return; }
    const double u = u_, v = v_, fx = c.K[0], fy = c.K[1], cx = c.K[2], cy = c.K[3];
    if (c.model == 1) {
        const double k1 = c.D[0], k2 = c.D[1], p1 = c.D[2], p2 = c.D[3], k3 = c.D[4];
        double x = (u - cx) * (1. / fx), y = (v - cy) * (1. / fy);
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; ++j) {   // TermCriteria(COUNT, 5): no epsilon test
            const double r2 = x * x + y * y;
            const double icdist = 1. / (1. + ((k3 * r2 + k2) * r2 + k1) * r2);
            if (icdist < 0) { x = (u - cx) * (1. / fx); y = (v - cy) * (1. / fy); break; }
            const double dX = 2. * p1 * x * y + p2 * (r2 + 2. * x * x), dY = p1 * (r2 + 2. * y * y) + 2. * p2 * x * y;
            x = (x0 - dX) * icdist;
            y = (y0 - dY) * icdist;
        }
        ou = (float)(fx * x + cx); ov = (float)(fy * y + cy);
        return;
    }
    const double pwx = (u - cx) / fx, pwy = (v - cy) / fy;
    double theta_d = sqrt(pwx * pwx + pwy * pwy);
    const double hp = 1.5707963267948966;
    theta_d = theta_d < -hp ? -hp : (theta_d > hp ? hp : theta_d);
    double scale = 1.0;
    if (theta_d > 1e-8) {
        double theta = theta_d;
        for (int j = 0; j < 10; ++j) {
            const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t6 * t2;
            const double k0t2 = c.D[0] * t2, k1t4 = c.D[1] * t4, k2t6 = c.D[2] * t6, k3t8 = c.D[3] * t8;
            const double fix = (theta * (1 + k0t2 + k1t4 + k2t6 + k3t8) - theta_d) / (1 + 3 * k0t2 + 5 * k1t4 + 7 * k2t6 + 9 * k3t8);
            theta -= fix;
            if (fabs(fix) < 1e-8) break;
        }
        scale = tan(theta) / theta_d;
    }
    ou = (float)(fx * (pwx * scale) + cx); ov = (float)(fy * (pwy * scale) + cy);
}

__host__ __device__ inline void ov2_cam_project_dist(const ov2_cam_model &c, const double pc[3], float &px, float &py)
{
    const double invz = 1. / pc[2], x = pc[0] * invz, y = pc[1] * invz, fx = c.K[0], fy = c.K[1], cx = c.K[2], cy = c.K[3];
    if (c.model == 0 || c.n_coeffs <= 0) { px = (float)(fx * x + cx); py = (float)(fy * y + cy); return; }
    const double xf = (double)(float)x, yf = (double)(float)y;   // the reference hands OpenCV a Point3f / Point2f
    if (c.model == 1) {
        const double k1 = c.D[0], k2 = c.D[1], p1 = c.D[2], p2 = c.D[3], k3 = c.D[4];
        const double r2 = xf * xf + yf * yf, r4 = r2 * r2, r6 = r4 * r2;
        const double a1 = 2 * xf * yf, a2 = r2 + 2 * xf * xf, a3 = r2 + 2 * yf * yf;
        const double cdist = 1 + k1 * r2 + k2 * r4 + k3 * r6;
        const double xd = xf * cdist + p1 * a1 + p2 * a2, yd = yf * cdist + p1 * a3 + p2 * a1;
        px = (float)(xd * fx + cx); py = (float)(yd * fy + cy);
        return;
    }
    const double r = sqrt(xf * xf + yf * yf), theta = atan(r);
    const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
    const double theta_d = theta * (1 + c.D[0] * t2 + c.D[1] * t4 + c.D[2] * t6 + c.D[3] * t8);
    const double inv_r = r > 1e-8 ? 1.0 / r : 1.0, cdist = r > 1e-8 ? theta_d * inv_r : 1.0;
    px = (float)(fx * (xf * cdist) + cx); py = (float)(fy * (yf * cdist) + cy);
}

// a model with the unused coefficients zeroed (callers may pass fewer than five); null -> model 0
inline ov2_cam_model ov2_cam_normalised(const ov2_cam_model *in)
{
    ov2_cam_model m;
    for (int i = 0; i < 4; ++i) m.K[i] = in ? in->K[i] : 0.0;
    m.model = in ? in->model : 0;
    m.n_coeffs = in ? (in->n_coeffs > 5 ? 5 : in->n_coeffs) : 0;
    for (int i = 0; i < 5; ++i) m.D[i] = (in && i < m.n_coeffs) ? in->D[i] : 0.0;
    if (m.n_coeffs <= 0) m.model = 0;
    return m;
}
