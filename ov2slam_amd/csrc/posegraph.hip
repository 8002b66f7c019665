// posegraph.hip -- pose-graph optimisation of keyframe / frame CHAINS on gfx950, the whole LM loop in one launch.
//
// Replaces (reference, /root/reference): the Ceres problems of Optimizer::localPoseGraph src/optimizer.cpp:2346-2592 (loop
// closer: keyframes loop .. new joined by odometry edges + the loop edge, loop keyframe constant, 10 iterations) and
// Optimizer::fullPoseGraph :2783-2870 (every frame between constant keyframes, 100 iterations): cost functor
// LeftSE3RelativePoseError (src/ceres_parametrization.cpp:30-102: r = log(Twc1^-1 Twc0 Tc0c1), jacobians (I + J_c/2) Adj),
// SE3LeftParameterization, LEVENBERG_MARQUARDT with the Ceres radius rules, Jacobi scaling, no loss function.  Restated
// on the CPU in oracle/ov2_oracle_pg.c (dense normal equations there).
//
// Both graphs are chains: an edge joins two free poses that are neighbours in the order of the free poses, or has a
// constant end.  J'J is then block tridiagonal (6 x 6 blocks) inside every run of coupled free poses, and runs are
// independent.  One workgroup of 256 threads: edges (residual, jacobians), pose blocks (normal-equation blocks, column
// norms, LM diagonal, Plus) and runs (block-tridiagonal Cholesky + the two substitutions, one thread per run: a few
// hundred dependent 6 x 6 operations) are strided over the threads; sums over edges close with a fixed-order tree; the
// trust-region bookkeeping is done by thread 0 in LDS.  A loop closure of 200 keyframes is 10 x ~0.5 ms of mostly
// serial latency -- this is glue on the loop closer's path, not a throughput kernel; it exists so that the solver's
// results do not depend on a host Ceres.
#include <algorithm>
#include <vector>

#include "ov2_internal.h"

namespace {

struct pg_dev {
    int n_pose, n_edge, nf, n_seg;
    const int *edge_i, *edge_j, *fidx, *pose_of_f, *inc_ptr, *inc, *seg0, *seg1;
    const double *T_ij;
    double *x, *cand, *best;                       // 7 n_pose each
    double *r, *Ji, *Jj;                           // 6 / 36 / 36 per edge
    double *D, *O, *g, *sqn, *scale, *diag, *lmd, *step, *Ld, *Lo, *y;   // per free block: 36, 36, 6, 6, 6, 6, 6, 6, 36, 36, 6
    double *part;                                  // 256 partial sums
    ov2_pg_result *res;
};

struct pg_opt {
    int max_iters, jacobi, max_invalid;
    double ftol, initial_radius, max_radius, min_radius, min_d, max_d, min_rel, ptol, gtol;
};

struct se3m { double R[9], t[3]; };

__device__ inline void pg_from7(const double *p, se3m &T)
{
    double x = p[3], y = p[4], z = p[5], w = p[6];
    const double n = sqrt(x * x + y * y + z * z + w * w);
    x /= n; y /= n; z /= n; w /= n;
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    T.R[0] = 1 - (tyy + tzz); T.R[1] = txy - twz;       T.R[2] = txz + twy;
    T.R[3] = txy + twz;       T.R[4] = 1 - (txx + tzz); T.R[5] = tyz - twx;
    T.R[6] = txz - twy;       T.R[7] = tyz + twx;       T.R[8] = 1 - (txx + tyy);
    T.t[0] = p[0]; T.t[1] = p[1]; T.t[2] = p[2];
}

__device__ inline void pg_mul(const se3m &A, const se3m &B, se3m &C)
{
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) C.R[3 * i + j] = A.R[3 * i] * B.R[j] + A.R[3 * i + 1] * B.R[3 + j] + A.R[3 * i + 2] * B.R[6 + j];
        C.t[i] = A.t[i] + (A.R[3 * i] * B.t[0] + A.R[3 * i + 1] * B.t[1] + A.R[3 * i + 2] * B.t[2]);
    }
}

__device__ inline void pg_inv(const se3m &A, se3m &C)
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C.R[3 * i + j] = A.R[3 * j + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) C.t[i] = -(C.R[3 * i] * A.t[0] + C.R[3 * i + 1] * A.t[1] + C.R[3 * i + 2] * A.t[2]);
}

// rotation matrix -> unit quaternion (Eigen's branch order), then Sophus SO3::logAndTheta + SE3::log
__device__ inline void pg_log(const se3m &T, double out[6])
{
    const double eps = 1e-10;
    const double *R = T.R;
    double qx, qy, qz, qw;
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        double t = sqrt(tr + 1.0);
        qw = 0.5 * t; t = 0.5 / t;
        qx = (R[7] - R[5]) * t; qy = (R[2] - R[6]) * t; qz = (R[3] - R[1]) * t;
    } else if (R[0] >= R[4] && R[0] >= R[8]) {          // i = 0 (ties as in the oracle: i moves only on a strict ">")
        double t = sqrt(R[0] - R[4] - R[8] + 1.0);
        qx = 0.5 * t; t = 0.5 / t;
        qw = (R[7] - R[5]) * t; qy = (R[3] + R[1]) * t; qz = (R[6] + R[2]) * t;
    } else if (R[4] > R[0] && R[4] >= R[8]) {           // i = 1
        double t = sqrt(R[4] - R[8] - R[0] + 1.0);
        qy = 0.5 * t; t = 0.5 / t;
        qw = (R[2] - R[6]) * t; qz = (R[7] + R[5]) * t; qx = (R[1] + R[3]) * t;
    } else {                                            // i = 2
        double t = sqrt(R[8] - R[0] - R[4] + 1.0);
        qz = 0.5 * t; t = 0.5 / t;
        qw = (R[3] - R[1]) * t; qx = (R[2] + R[6]) * t; qy = (R[5] + R[7]) * t;
    }
    const double sn = qx * qx + qy * qy + qz * qz;
    double f, theta;
    if (sn < eps * eps) {
        const double w2 = qw * qw;
        f = 2.0 / qw - (2.0 / 3.0) * sn / (qw * w2);
        theta = 2.0 * sn / qw;
    } else {
        const double n = sqrt(sn);
        if (fabs(qw) < eps) f = (qw > 0 ? 3.14159265358979323846 : -3.14159265358979323846) / n;
        else f = 2.0 * atan(n / qw) / n;
        theta = f * n;
    }
    const double om[3] = {f * qx, f * qy, f * qz};
    const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double c;
    if (fabs(theta) < eps) c = 1.0 / 12.0;
    else { const double h = 0.5 * theta; c = (1.0 - theta * cos(h) / (2.0 * sin(h))) / (theta * theta); }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double s = 0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double o2 = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
            s += (((i == j) ? 1.0 : 0.0) - 0.5 * O[3 * i + j] + c * o2) * T.t[j];
        }
        out[i] = s;
    }
    out[3] = om[0]; out[4] = om[1]; out[5] = om[2];
}

// J = sign * (I + sg/2 [So Sr; 0 So]) * Adj(T), row-major 6 x 6, columns scaled by sc (may be null)
__device__ inline void pg_jac(const double r[6], const se3m &T, double sg, double sign, const double *sc, double *J)
{
    const double *rho = r, *om = r + 3;
    const double Sr[9] = {0, -rho[2], rho[1], rho[2], 0, -rho[0], -rho[1], rho[0], 0};
    const double So[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    const double *t = T.t, *R = T.R;
    const double H[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
    double HR[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) HR[3 * i + j] = H[3 * i] * R[j] + H[3 * i + 1] * R[3 + j] + H[3 * i + 2] * R[6 + j];
    // M = I + sg/2 [So Sr; 0 So];  A = [R HR; 0 R];  M A = [ (I + a So) R , (I + a So) HR + a Sr R ; 0 , (I + a So) R ]
    const double a = 0.5 * sg;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            // evaluated as the 6 x 6 product of the oracle, term by term in the same order (k = 0..5)
            double tl = 0, trr = 0, br = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double m = ((i == k) ? 1.0 : 0.0) + a * So[3 * i + k];
                tl += m * R[3 * k + j];
                trr += m * HR[3 * k + j];
                br += m * R[3 * k + j];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) trr += (a * Sr[3 * i + k]) * R[3 * k + j];
            const double s0 = sc ? sc[j] : 1.0, s1 = sc ? sc[j + 3] : 1.0;
            J[6 * i + j] = (sign * tl) * s0;
            J[6 * i + j + 3] = (sign * trr) * s1;
            J[6 * (i + 3) + j] = 0.0;
            J[6 * (i + 3) + j + 3] = (sign * br) * s1;
        }
}

__device__ inline void pg_eval_edge(const pg_dev &d, const double *poses, int e, bool jac, bool scaled)
{
    const int i = d.edge_i[e], j = d.edge_j[e];
    se3m T0, T1, T01, T1i, A, E;
    pg_from7(poses + 7 * i, T0);
    pg_from7(poses + 7 * j, T1);
    pg_from7(d.T_ij + 7 * (size_t)e, T01);
    pg_inv(T1, T1i);
    pg_mul(T1i, T0, A);
    pg_mul(A, T01, E);
    double r[6];
    pg_log(E, r);
#pragma unroll
    for (int k = 0; k < 6; ++k) d.r[6 * (size_t)e + k] = r[k];
    if (!jac) return;
    const int fi = d.fidx[i], fj = d.fidx[j];
    pg_jac(r, T1i, -1.0, 1.0, (scaled && fi >= 0) ? d.scale + 6 * fi : nullptr, d.Ji + 36 * (size_t)e);
    se3m P, Pi;
    pg_mul(T0, T01, P);
    pg_inv(P, Pi);
    pg_jac(r, Pi, 1.0, -1.0, (scaled && fj >= 0) ? d.scale + 6 * fj : nullptr, d.Jj + 36 * (size_t)e);
}

// fixed-order sum of one value per thread (256 threads), result to every thread
__device__ inline double pg_block_sum(double v, double *sh)
{
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) sh[t] += sh[t + s];
        __syncthreads();
    }
    const double tot = sh[0];
    __syncthreads();
    return tot;
}

// residuals (+ jacobians) of every edge, cost
__device__ inline double pg_evaluate(const pg_dev &d, const double *poses, bool jac, bool scaled, double *sh)
{
    double c = 0.0;
    for (int e = threadIdx.x; e < d.n_edge; e += 256) {
        pg_eval_edge(d, poses, e, jac, scaled);
        double s = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) { const double v = d.r[6 * (size_t)e + k]; s += v * v; }
        c += 0.5 * s;
    }
    __threadfence_block();
    return pg_block_sum(c, sh);
}

// blocks of J'J and J'r of every free pose: D_f (diagonal block), O_f (block (f+1, f)), g_f, column norms
__device__ inline void pg_normal(const pg_dev &d)
{
    for (int f = threadIdx.x; f < d.nf; f += 256) {
        double D[36], O[36], g[6];
#pragma unroll
        for (int k = 0; k < 36; ++k) { D[k] = 0.0; O[k] = 0.0; }
#pragma unroll
        for (int k = 0; k < 6; ++k) g[k] = 0.0;
        for (int q = d.inc_ptr[f]; q < d.inc_ptr[f + 1]; ++q) {
            const int e = d.inc[q] >> 1, side = d.inc[q] & 1;
            const double *J = (side ? d.Jj : d.Ji) + 36 * (size_t)e, *Jo = (side ? d.Ji : d.Jj) + 36 * (size_t)e;
            const double *r = d.r + 6 * (size_t)e;
            const int other = side ? d.edge_i[e] : d.edge_j[e];
            const bool up = d.fidx[other] == f + 1;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
#pragma unroll
                for (int b = 0; b < 6; ++b) {
                    double s = 0, so = 0;
#pragma unroll
                    for (int k = 0; k < 6; ++k) { s += J[6 * k + a] * J[6 * k + b]; so += Jo[6 * k + a] * J[6 * k + b]; }
                    D[6 * a + b] += s;
                    if (up) O[6 * a + b] += so;     // rows: block f + 1, columns: block f
                }
                double gg = 0;
#pragma unroll
                for (int k = 0; k < 6; ++k) gg += J[6 * k + a] * r[k];
                g[a] += gg;
            }
        }
#pragma unroll
        for (int k = 0; k < 36; ++k) { d.D[36 * (size_t)f + k] = D[k]; d.O[36 * (size_t)f + k] = O[k]; }
#pragma unroll
        for (int k = 0; k < 6; ++k) { d.g[6 * f + k] = g[k]; d.sqn[6 * f + k] = D[7 * k]; }
    }
    __threadfence_block();
    __syncthreads();
}

// (J'J + D'D) y = J'r by runs of coupled free poses; step = -y.  Returns (to every thread) whether every run factorised.
__device__ inline bool pg_solve(const pg_dev &d, int *flag)
{
    if (threadIdx.x == 0) *flag = 1;
    __syncthreads();
    for (int sg = threadIdx.x; sg < d.n_seg; sg += 256) {
        const int s0 = d.seg0[sg], s1 = d.seg1[sg];
        bool ok = true;
        double Lp[36], yp[6];     // Lo_{f-1}, y_{f-1}
#pragma unroll
        for (int k = 0; k < 36; ++k) Lp[k] = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) yp[k] = 0.0;
        for (int f = s0; f <= s1 && ok; ++f) {
            double A[36], b[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) {
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    double s = d.D[36 * (size_t)f + 6 * a + c];
#pragma unroll
                    for (int k = 0; k < 6; ++k) s -= Lp[6 * a + k] * Lp[6 * c + k];
                    A[6 * a + c] = s;
                }
                const double l = d.lmd[6 * f + a];
                A[7 * a] += l * l;
                double s = d.g[6 * f + a];
#pragma unroll
                for (int k = 0; k < 6; ++k) s -= Lp[6 * a + k] * yp[k];
                b[a] = s;
            }
            // 6 x 6 Cholesky (lower, in A) and y_f = L^-1 b
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                double dj = A[7 * j];
#pragma unroll
                for (int k = 0; k < j; ++k) dj -= A[6 * j + k] * A[6 * j + k];
                if (!(dj > 0.0)) ok = false;
                dj = sqrt(dj);
                A[7 * j] = dj;
#pragma unroll
                for (int i = j + 1; i < 6; ++i) {
                    double s = A[6 * i + j];
#pragma unroll
                    for (int k = 0; k < j; ++k) s -= A[6 * i + k] * A[6 * j + k];
                    A[6 * i + j] = s / dj;
                }
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                double s = b[i];
#pragma unroll
                for (int k = 0; k < i; ++k) s -= A[6 * i + k] * yp[k];
                yp[i] = s / A[7 * i];
            }
#pragma unroll
            for (int k = 0; k < 36; ++k) d.Ld[36 * (size_t)f + k] = A[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) d.y[6 * f + k] = yp[k];
            if (f < s1) {   // Lo_f = O_f L^-T: row a of Lo solves L x = (row a of O)
#pragma unroll
                for (int a = 0; a < 6; ++a) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        double s = d.O[36 * (size_t)f + 6 * a + i];
#pragma unroll
                        for (int k = 0; k < i; ++k) s -= A[6 * i + k] * Lp[6 * a + k];
                        Lp[6 * a + i] = s / A[7 * i];
                    }
                }
#pragma unroll
                for (int k = 0; k < 36; ++k) d.Lo[36 * (size_t)f + k] = Lp[k];
            }
        }
        if (!ok) { *flag = 0; continue; }
        // backward: x_f = L_f^-T (y_f - Lo_f' x_{f+1})
        double xn[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) xn[k] = 0.0;
        for (int f = s1; f >= s0; --f) {
            double b[6], L[36];
#pragma unroll
            for (int k = 0; k < 36; ++k) L[k] = d.Ld[36 * (size_t)f + k];
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                double s = d.y[6 * f + a];
                if (f < s1) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) s -= d.Lo[36 * (size_t)f + 6 * k + a] * xn[k];
                }
                b[a] = s;
            }
#pragma unroll
            for (int i = 5; i >= 0; --i) {
                double s = b[i];
#pragma unroll
                for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * xn[k];
                xn[i] = s / L[7 * i];
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) d.step[6 * f + k] = -xn[k];
        }
    }
    __threadfence_block();
    __syncthreads();
    return *flag != 0;
}

// SE3LeftParameterization::Plus (exp(delta) * x), as pnp.hip / the oracle
__device__ inline void pg_se3_plus(const double *x, const double *dl, double *out)
{
    const double *u = dl, *w = dl + 3;
    const double eps = 1e-10;
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double theta, imag, real;
    if (th2 < eps * eps) {
        theta = 0.0;
        const double th4 = th2 * th2;
        imag = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
        real = 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4;
    } else {
        theta = sqrt(th2);
        const double half = 0.5 * theta;
        imag = sin(half) / theta;
        real = cos(half);
    }
    const double ax = imag * w[0], ay = imag * w[1], az = imag * w[2], aw = real;
    double Ra[9], V[9];
    {
        const double tx = 2 * ax, ty = 2 * ay, tz = 2 * az;
        const double twx = tx * aw, twy = ty * aw, twz = tz * aw, txx = tx * ax, txy = ty * ax, txz = tz * ax;
        const double tyy = ty * ay, tyz = tz * ay, tzz = tz * az;
        Ra[0] = 1 - (tyy + tzz); Ra[1] = txy - twz;       Ra[2] = txz + twy;
        Ra[3] = txy + twz;       Ra[4] = 1 - (txx + tzz); Ra[5] = tyz - twx;
        Ra[6] = txz - twy;       Ra[7] = tyz + twx;       Ra[8] = 1 - (txx + tyy);
    }
    if (theta < eps) {
#pragma unroll
        for (int i = 0; i < 9; ++i) V[i] = Ra[i];
    } else {
        const double O[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
        const double t2 = theta * theta;
        const double c1 = (1.0 - cos(theta)) / t2, c2 = (theta - sin(theta)) / (t2 * theta);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double s = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) s += O[3 * i + k] * O[3 * k + j];
                V[3 * i + j] = ((i == j) ? 1.0 : 0.0) + c1 * O[3 * i + j] + c2 * s;
            }
    }
    double b0 = x[3], b1 = x[4], b2 = x[5], b3 = x[6];
    const double nb = sqrt(b0 * b0 + b1 * b1 + b2 * b2 + b3 * b3);
    b0 /= nb; b1 /= nb; b2 /= nb; b3 /= nb;
    double q3 = aw * b3 - ax * b0 - ay * b1 - az * b2;
    double q0 = aw * b0 + ax * b3 + ay * b2 - az * b1;
    double q1 = aw * b1 + ay * b3 + az * b0 - ax * b2;
    double q2 = aw * b2 + az * b3 + ax * b1 - ay * b0;
    const double nq = sqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
    q0 /= nq; q1 /= nq; q2 /= nq; q3 /= nq;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        out[i] = (V[3 * i] * u[0] + V[3 * i + 1] * u[1] + V[3 * i + 2] * u[2]) + (Ra[3 * i] * x[0] + Ra[3 * i + 1] * x[1] + Ra[3 * i + 2] * x[2]);
    out[3] = q0; out[4] = q1; out[5] = q2; out[6] = q3;
}

// out = Plus(x, delta) on the free poses (the other poses are copied); delta = -v (PG_NEG), v * scale (PG_STEP: the step
// of the scaled problem) or -v / scale (PG_NEG_UNSCALE: the gradient of the unscaled problem from the scaled one)
enum { PG_NEG = 0, PG_STEP = 1, PG_NEG_UNSCALE = 2 };
__device__ inline void pg_plus(const pg_dev &d, const double *x, const double *v, int mode, double *out)
{
    for (int i = threadIdx.x; i < d.n_pose; i += 256) {
        const int f = d.fidx[i];
        if (f < 0) {
#pragma unroll
            for (int k = 0; k < 7; ++k) out[7 * i + k] = x[7 * i + k];
        } else {
            double dl[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const double vv = v[6 * f + k], sc = d.scale[6 * f + k];
                dl[k] = mode == PG_STEP ? vv * sc : (mode == PG_NEG_UNSCALE ? -vv / sc : -vv);
            }
            pg_se3_plus(x + 7 * i, dl, out + 7 * i);
        }
    }
    __threadfence_block();
    __syncthreads();
}

__device__ inline double pg_gmax(const pg_dev &d, const double *x, const double *c, double *sh)
{
    double m = 0.0;
    for (int f = threadIdx.x; f < d.nf; f += 256) {
        const int i = d.pose_of_f[f];
#pragma unroll
        for (int k = 0; k < 7; ++k) m = fmax(m, fabs(x[7 * i + k] - c[7 * i + k]));
    }
    const int t = threadIdx.x;
    sh[t] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) sh[t] = fmax(sh[t], sh[t + s]);
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

__device__ inline double pg_norm2(const pg_dev &d, const double *a, const double *b, double *sh)
{
    double s = 0.0;
    for (int f = threadIdx.x; f < d.nf; f += 256) {
        const int i = d.pose_of_f[f];
#pragma unroll
        for (int k = 0; k < 7; ++k) { const double v = a[7 * i + k] - (b ? b[7 * i + k] : 0.0); s += v * v; }
    }
    return pg_block_sum(s, sh);
}

__device__ inline void pg_copy(const pg_dev &d, double *dst, const double *src)
{
    for (int k = threadIdx.x; k < 7 * d.n_pose; k += 256) dst[k] = src[k];
    __threadfence_block();
    __syncthreads();
}

__device__ inline void pg_logit(ov2_pg_result *R, double cost, double change, double radius, double rel, double model, int valid, int ok)
{
    if (threadIdx.x != 0 || R->n_log >= OV2_BA_MAX_LOG) return;
    ov2_ba_iter *it = &R->log[R->n_log++];
    it->cost = cost; it->cost_change = change; it->radius = radius; it->relative_decrease = rel;
    it->model_cost_change = model; it->step_is_valid = valid; it->step_is_successful = ok;
}

// the whole minimisation (trust_region_minimizer.cc as restated in oracle/ov2_oracle_pg.c); every thread carries the
// scalar state redundantly (all sums are broadcast), so the control flow is uniform
__global__ __launch_bounds__(256) void pg_minimize_kernel(pg_dev d, pg_opt o)
{
    __shared__ double sh[256];
    __shared__ int flag;
    ov2_pg_result *R = d.res;
    if (threadIdx.x == 0) { R->n_log = 0; R->termination = OV2_BA_TERM_MAX_ITER; }
    const int m = 6 * d.nf;
    for (int k = threadIdx.x; k < m; k += 256) d.scale[k] = 1.0;
    pg_copy(d, d.best, d.x);
    double x_cost = pg_evaluate(d, d.x, true, false, sh);
    pg_normal(d);
    // gradient_max_norm = || x - Plus(x, -g) ||_inf with the gradient of the unscaled problem
    pg_plus(d, d.x, d.g, PG_NEG, d.cand);
    double gmax = pg_gmax(d, d.x, d.cand, sh);
    if (o.jacobi) {
        for (int k = threadIdx.x; k < m; k += 256) d.scale[k] = 1.0 / (1.0 + sqrt(d.sqn[k]));
        __threadfence_block();
        __syncthreads();
        x_cost = pg_evaluate(d, d.x, true, true, sh);
        pg_normal(d);
    }
    if (threadIdx.x == 0) R->initial_cost = x_cost;
    double minimum_cost = x_cost, x_norm = -1.0, radius = o.initial_radius, decrease_factor = 2.0;
    int reuse_diagonal = 0, invalid_steps = 0, iteration = 0, last_ok = 1, term = OV2_BA_TERM_MAX_ITER;
    pg_logit(R, x_cost, 0.0, radius, 0.0, 0.0, 1, 1);
    for (;;) {
        if (iteration >= o.max_iters) { term = OV2_BA_TERM_MAX_ITER; break; }
        if (last_ok && gmax <= o.gtol) { term = OV2_BA_TERM_GTOL; break; }
        if (radius <= o.min_radius) { term = OV2_BA_TERM_MIN_RADIUS; break; }
        ++iteration;
        for (int k = threadIdx.x; k < m; k += 256) {
            if (!reuse_diagonal) d.diag[k] = fmin(fmax(d.sqn[k], o.min_d), o.max_d);
            d.lmd[k] = sqrt(d.diag[k] / radius);
        }
        __threadfence_block();
        __syncthreads();
        reuse_diagonal = 1;
        bool finite = pg_solve(d, &flag);
        {
            double bad = 0.0;
            for (int k = threadIdx.x; k < m; k += 256) if (!isfinite(d.step[k])) bad = 1.0;
            if (pg_block_sum(bad, sh) > 0.0) finite = false;
        }
        double model_change = 0.0;
        bool valid = false;
        if (finite) {
            double mc = 0.0;
            for (int e = threadIdx.x; e < d.n_edge; e += 256) {
                const int fi = d.fidx[d.edge_i[e]], fj = d.fidx[d.edge_j[e]];
                const double *Ji = d.Ji + 36 * (size_t)e, *Jj = d.Jj + 36 * (size_t)e, *r = d.r + 6 * (size_t)e;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    double mk = 0;
                    if (fi >= 0) {
#pragma unroll
                        for (int c = 0; c < 6; ++c) mk += Ji[6 * k + c] * d.step[6 * fi + c];
                    }
                    if (fj >= 0) {
#pragma unroll
                        for (int c = 0; c < 6; ++c) mk += Jj[6 * k + c] * d.step[6 * fj + c];
                    }
                    mc -= mk * (r[k] + mk / 2.0);
                }
            }
            model_change = pg_block_sum(mc, sh);
            valid = model_change > 0.0;
        }
        if (!valid) {
            if (++invalid_steps >= o.max_invalid) { term = OV2_BA_TERM_FAILURE; break; }
            radius = radius / decrease_factor; decrease_factor *= 2.0;   // StepRejected
            last_ok = 0;
            pg_logit(R, x_cost, 0.0, radius, 0.0, model_change, 0, 0);
            continue;
        }
        invalid_steps = 0;
        pg_plus(d, d.x, d.step, PG_STEP, d.cand);
        // the candidate's residuals overwrite r: the accepted branch re-evaluates, the rejected one restores them
        const double cand_cost = pg_evaluate(d, d.cand, false, false, sh);
        const double step_norm = sqrt(pg_norm2(d, d.x, d.cand, sh));
        bool stop = false;
        if (step_norm <= o.ptol * (x_norm + o.ptol)) { term = OV2_BA_TERM_PTOL; stop = true; }
        const double cost_change = x_cost - cand_cost;
        if (!stop && fabs(cost_change) <= o.ftol * x_cost) {
            term = OV2_BA_TERM_FTOL;
            pg_logit(R, x_cost, cost_change, radius, 0.0, model_change, 1, 0);
            stop = true;
        }
        if (stop) break;
        const double rel = (cand_cost >= 1.7976931348623157e308) ? -1.7976931348623157e308 : (x_cost - cand_cost) / model_change;
        if (rel > o.min_rel) {
            pg_copy(d, d.x, d.cand);
            x_norm = sqrt(pg_norm2(d, d.x, nullptr, sh));
            x_cost = pg_evaluate(d, d.x, true, o.jacobi != 0, sh);
            pg_normal(d);
            pg_plus(d, d.x, d.g, o.jacobi ? PG_NEG_UNSCALE : PG_NEG, d.cand);
            gmax = pg_gmax(d, d.x, d.cand, sh);
            radius = fmin(o.max_radius, radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3)));   // StepAccepted
            decrease_factor = 2.0;
            reuse_diagonal = 0;
            last_ok = 1;
            if (x_cost < minimum_cost) { minimum_cost = x_cost; pg_copy(d, d.best, d.x); }
            pg_logit(R, x_cost, cost_change, radius, rel, model_change, 1, 1);
        } else {
            radius = radius / decrease_factor; decrease_factor *= 2.0;
            last_ok = 0;
            (void)pg_evaluate(d, d.x, false, false, sh);   // r back to the residuals at x (the model uses them next round)
            pg_logit(R, cand_cost, cost_change, radius, rel, model_change, 1, 0);
        }
    }
    pg_copy(d, d.x, d.best);
    if (threadIdx.x == 0) { R->final_cost = minimum_cost; R->termination = term; }
}

}  // namespace

extern "C" ov2_status ov2_pose_graph_solve(ov2_ctx *c, const ov2_pg_problem *P, const ov2_ba_options *o, ov2_pg_result *R)
{
    if (!c) return OV2_ERR_INVALID;
    if (!P || !o || !R) return ov2_set_err(c, OV2_ERR_INVALID, "ov2_pose_graph_solve: null argument");
    memset(R, 0, sizeof(*R));
    const int n = P->n_pose, E = P->n_edge;
    if (n < 0 || E < 0 || (n && (!P->pose || !P->pose_const)) || (E && (!P->edge_i || !P->edge_j || !P->T_ij)))
        return ov2_set_err(c, OV2_ERR_INVALID, "ov2_pose_graph_solve: null / negative field");
    std::vector<int> fidx((size_t)n), pose_of_f;
    for (int i = 0; i < n; ++i) {
        fidx[i] = P->pose_const[i] ? -1 : (int)pose_of_f.size();
        if (!P->pose_const[i]) pose_of_f.push_back(i);
    }
    const int nf = (int)pose_of_f.size();
    if (nf == 0 || E == 0) { R->termination = OV2_BA_TERM_SKIPPED; return OV2_OK; }
    // chain structure, incidence lists, runs of coupled free poses
    std::vector<std::vector<int>> inc((size_t)nf);
    std::vector<char> couple((size_t)nf, 0);
    for (int e = 0; e < E; ++e) {
        const int i = P->edge_i[e], j = P->edge_j[e];
        if (i < 0 || i >= n || j < 0 || j >= n || i == j) return ov2_set_err(c, OV2_ERR_INVALID, "edge %d joins poses %d and %d", e, i, j);
        const int fi = fidx[i], fj = fidx[j];
        if (fi >= 0) inc[fi].push_back(2 * e);
        if (fj >= 0) inc[fj].push_back(2 * e + 1);
        if (fi >= 0 && fj >= 0) {
            if (std::abs(fi - fj) != 1)
                return ov2_set_err(c, OV2_ERR_UNSUPPORTED, "edge %d joins free poses %d and %d that are not neighbours: only chains "
                                   "(localPoseGraph / fullPoseGraph) are solved", e, i, j);
            couple[std::min(fi, fj)] = 1;
        }
    }
    std::vector<int> inc_ptr((size_t)nf + 1, 0), inc_flat, seg0, seg1;
    for (int f = 0; f < nf; ++f) { inc_ptr[f + 1] = inc_ptr[f] + (int)inc[f].size(); inc_flat.insert(inc_flat.end(), inc[f].begin(), inc[f].end()); }
    for (int f = 0; f < nf;) {
        int g = f;
        while (g + 1 < nf && couple[g]) ++g;
        seg0.push_back(f); seg1.push_back(g);
        f = g + 1;
    }
    const int n_seg = (int)seg0.size();
    OV2_HIP(c, hipSetDevice(c->device));
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    // staging block: [inputs | poses || result] mirrored on the device, followed by the device-only workspace
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o_ = off; off += up(bytes); return o_; };
    const size_t o_ei = carve(sizeof(int) * E), o_ej = carve(sizeof(int) * E), o_fi = carve(sizeof(int) * n), o_pf = carve(sizeof(int) * nf);
    const size_t o_ip = carve(sizeof(int) * (nf + 1)), o_in = carve(sizeof(int) * (inc_flat.size() + 1));
    const size_t o_s0 = carve(sizeof(int) * n_seg), o_s1 = carve(sizeof(int) * n_seg), o_T = carve(sizeof(double) * 7 * E);
    const size_t o_x = carve(sizeof(double) * 7 * n), in_end = off;
    const size_t o_res = carve(sizeof(ov2_pg_result)), io_end = off;
    const size_t o_cand = carve(sizeof(double) * 7 * n), o_best = carve(sizeof(double) * 7 * n);
    const size_t o_r = carve(sizeof(double) * 6 * E), o_Ji = carve(sizeof(double) * 36 * E), o_Jj = carve(sizeof(double) * 36 * E);
    const size_t o_D = carve(sizeof(double) * 36 * nf), o_O = carve(sizeof(double) * 36 * nf), o_Ld = carve(sizeof(double) * 36 * nf);
    const size_t o_Lo = carve(sizeof(double) * 36 * nf), o_v = carve(sizeof(double) * 6 * nf * 8), o_part = carve(sizeof(double) * 256);
    char *hp = nullptr, *dp = nullptr;
    ov2_status s = ov2_staging(c, off, (void **)&hp, (void **)&dp);
    if (s != OV2_OK) return s;
    memcpy(hp + o_ei, P->edge_i, sizeof(int) * E); memcpy(hp + o_ej, P->edge_j, sizeof(int) * E);
    memcpy(hp + o_fi, fidx.data(), sizeof(int) * n); memcpy(hp + o_pf, pose_of_f.data(), sizeof(int) * nf);
    memcpy(hp + o_ip, inc_ptr.data(), sizeof(int) * (nf + 1));
    if (!inc_flat.empty()) memcpy(hp + o_in, inc_flat.data(), sizeof(int) * inc_flat.size());
    memcpy(hp + o_s0, seg0.data(), sizeof(int) * n_seg); memcpy(hp + o_s1, seg1.data(), sizeof(int) * n_seg);
    memcpy(hp + o_T, P->T_ij, sizeof(double) * 7 * E);
    memcpy(hp + o_x, P->pose, sizeof(double) * 7 * n);
    hipStream_t st = c->stream;
    OV2_HIP(c, hipMemcpyAsync(dp, hp, in_end, hipMemcpyHostToDevice, st));
    pg_dev d;
    d.n_pose = n; d.n_edge = E; d.nf = nf; d.n_seg = n_seg;
    d.edge_i = (const int *)(dp + o_ei); d.edge_j = (const int *)(dp + o_ej); d.fidx = (const int *)(dp + o_fi);
    d.pose_of_f = (const int *)(dp + o_pf); d.inc_ptr = (const int *)(dp + o_ip); d.inc = (const int *)(dp + o_in);
    d.seg0 = (const int *)(dp + o_s0); d.seg1 = (const int *)(dp + o_s1); d.T_ij = (const double *)(dp + o_T);
    d.x = (double *)(dp + o_x); d.cand = (double *)(dp + o_cand); d.best = (double *)(dp + o_best);
    d.r = (double *)(dp + o_r); d.Ji = (double *)(dp + o_Ji); d.Jj = (double *)(dp + o_Jj);
    d.D = (double *)(dp + o_D); d.O = (double *)(dp + o_O); d.Ld = (double *)(dp + o_Ld); d.Lo = (double *)(dp + o_Lo);
    double *v = (double *)(dp + o_v);
    const size_t m = 6 * (size_t)nf;
    d.g = v; d.sqn = v + m; d.scale = v + 2 * m; d.diag = v + 3 * m; d.lmd = v + 4 * m; d.step = v + 5 * m; d.y = v + 6 * m;
    d.part = (double *)(dp + o_part);
    d.res = (ov2_pg_result *)(dp + o_res);
    pg_opt po;
    po.max_iters = o->max_iters; po.jacobi = o->jacobi_scaling; po.max_invalid = o->max_consecutive_invalid_steps;
    po.ftol = o->function_tolerance; po.initial_radius = o->initial_radius; po.max_radius = o->max_radius; po.min_radius = o->min_radius;
    po.min_d = o->min_lm_diagonal; po.max_d = o->max_lm_diagonal; po.min_rel = o->min_relative_decrease;
    po.ptol = o->parameter_tolerance; po.gtol = o->gradient_tolerance;
    OV2_LAUNCH(c, OV2_K_MAP + 6, pg_minimize_kernel, dim3(1), dim3(256), 0, st, d, po);
    OV2_HIP(c, hipGetLastError());
    OV2_HIP(c, hipMemcpyAsync(hp + o_x, dp + o_x, io_end - o_x, hipMemcpyDeviceToHost, st));
    OV2_HIP(c, hipStreamSynchronize(st));
    memcpy(P->pose, hp + o_x, sizeof(double) * 7 * n);
    memcpy(R, hp + o_res, sizeof(*R));
    return OV2_OK;
}
