/*
 * ov2_oracle_ba.c -- CPU restatement of Optimizer::localBA's numerical core (TEST INFRASTRUCTURE ONLY, see
 * ov2_oracle.h): the reference's analytic cost functors + the subset of Ceres 2.0.0 that ceres::Solve runs for it.
 *
 * Reference files followed (/root/reference):
 *   cost functors ............ src/ceres_parametrization.cpp:107-196 (L_XYZ) :198-298 (R_XYZ) :361-470 (L_INV)
 *                               :476-575 (RANCH_INV) :579-709 (R_INV)
 *   SE3 left update .......... include/ceres_parametrization/ceres_parametrization/se3left_parametrization.hpp:39-73
 *   Sophus exp / product ..... Thirdparty/Sophus/sophus/se3.hpp:763-784, so3.hpp:585-621, so3.hpp:329-343
 *   problem set-up / options . src/optimizer.cpp:43-479 ; flagging + L2 re-solve :484-735
 *   Ceres (Thirdparty/ceres-solver/internal/ceres):
 *     trust_region_minimizer.cc:67-827, levenberg_marquardt_strategy.cc:66-164,
 *     trust_region_step_evaluator.cc:52-112, residual_block.cc:69-204, corrector.cc:42-157, loss_function.cc:48-62,
 *     schur_complement_solver.cc:118-175, schur_eliminator_impl.h:179-694, invert_psd_matrix.h:51-72
 * Parity: the linear algebra is pinned by the known-answer vectors in Ceres' own tests
 * (linear_least_squares_problems.cc:66-180, corrector_test.cc, levenberg_marquardt_strategy_test.cc:81-111) -- see
 * tests/golden/ceres_known_answers.json; the reference's own localBA outputs are "parity unpinned" (no fixtures, not
 * buildable here).  Deviation kept on purpose: outlier flags are taken at the final accepted state,
 * not at "whatever point Ceres evaluated last"; wall-clock caps are not applied.
 */
#include "ov2_oracle_ba.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* SE3                                                                                          */

static void quat_normalize(double q[4])
{
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

/* Eigen::Quaternion::toRotationMatrix, q = (x,y,z,w), R row-major */
static void quat_to_R(const double q[4], double R[9])
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

/* Sophus::SE3d(q, t): the SO3 constructor normalises the quaternion */
static void pose_Rt(const double p[7], double R[9], double t[3])
{
    double q[4] = {p[3], p[4], p[5], p[6]};
    quat_normalize(q);
    quat_to_R(q, R);
    t[0] = p[0]; t[1] = p[1]; t[2] = p[2];
}

static void mat3_vec(const double R[9], const double v[3], double o[3])
{
    o[0] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
    o[1] = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
    o[2] = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
}

static void mat3T_vec(const double R[9], const double v[3], double o[3])
{
    o[0] = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
    o[1] = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
    o[2] = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
}

/* Sophus::SE3::exp, tangent = [upsilon, omega]; out = [t, q(x,y,z,w)] */
void ov2o_se3_exp(const double a[6], double out[7])
{
    const double *u = a, *w = a + 3;
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
    double q[4] = {imag * w[0], imag * w[1], imag * w[2], real};
    double V[9];
    if (theta < eps) {
        quat_to_R(q, V);
    } else {
        const double O[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
        double O2[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0;
                for (int k = 0; k < 3; ++k) s += O[3 * i + k] * O[3 * k + j];
                O2[3 * i + j] = s;
            }
        const double t2 = theta * theta;
        const double c1 = (1.0 - cos(theta)) / t2, c2 = (theta - sin(theta)) / (t2 * theta);
        for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c1 * O[i] + c2 * O2[i];
    }
    mat3_vec(V, u, out);
    out[3] = q[0]; out[4] = q[1]; out[5] = q[2]; out[6] = q[3];
}

/* SE3LeftParameterization::Plus: x+ = exp(delta) * x */
void ov2o_se3_plus(const double x[7], const double d[6], double out[7])
{
    double e[7];
    ov2o_se3_exp(d, e);
    double b[4] = {x[3], x[4], x[5], x[6]};
    quat_normalize(b);
    const double ax = e[3], ay = e[4], az = e[5], aw = e[6];
    double q[4];
    q[3] = aw * b[3] - ax * b[0] - ay * b[1] - az * b[2];
    q[0] = aw * b[0] + ax * b[3] + ay * b[2] - az * b[1];
    q[1] = aw * b[1] + ay * b[3] + az * b[0] - ax * b[2];
    q[2] = aw * b[2] + az * b[3] + ax * b[1] - ay * b[0];
    quat_normalize(q);
    double Ra[9], rt[3];
    quat_to_R(e + 3, Ra);
    mat3_vec(Ra, x, rt);
    out[0] = e[0] + rt[0]; out[1] = e[1] + rt[1]; out[2] = e[2] + rt[2];
    out[3] = q[0]; out[4] = q[1]; out[5] = q[2]; out[6] = q[3];
}

/* ------------------------------------------------------------------------------------------ */
/* cost functors                                                                                */

/* J(2x3) * hat(w) */
static void j_hat(const double J[6], const double w[3], double o[6])
{
    for (int r = 0; r < 2; ++r) {
        const double a = J[3 * r], b = J[3 * r + 1], c = J[3 * r + 2];
        o[3 * r + 0] = b * w[2] - c * w[1];
        o[3 * r + 1] = c * w[0] - a * w[2];
        o[3 * r + 2] = a * w[1] - b * w[0];
    }
}

void ov2o_ba_eval_residual(const ov2_ba_problem *P, const double *poses, const double *lms, int i, int want_jac,
                           ov2o_res_eval *o)
{
    const int type = P->res_type[i];
    const int l = P->res_lm[i];
    const double sigma = P->res_sigma ? P->res_sigma[i] : 1.0;
    const double inv_sigma = 1.0 / sigma;
    const int is_right = (type == OV2_BA_R_XYZ || type == OV2_BA_R_INV || type == OV2_BA_RANCH_INV);
    const double *K = is_right ? P->calib_r : P->calib_l;
    double Rrl[9] = {0}, trl[3] = {0, 0, 0};
    if (is_right) pose_Rt(P->T_rl, Rrl, trl);

    double wpt[3] = {0, 0, 0}, anchpt[3] = {0, 0, 0}, Rwa[9], twa[3], zanch = 0.0;
    const int inv = (type == OV2_BA_L_INV || type == OV2_BA_R_INV || type == OV2_BA_RANCH_INV);
    if (inv) {
        zanch = 1.0 / lms[l];
        const double *Kl = P->calib_l;
        /* invK * [u v 1] with K = [fx 0 cx; 0 fy cy; 0 0 1] */
        const double ua = P->lm_anchor_uv[2 * l], va = P->lm_anchor_uv[2 * l + 1];
        anchpt[0] = zanch * ((ua - Kl[2]) / Kl[0]);
        anchpt[1] = zanch * ((va - Kl[3]) / Kl[1]);
        anchpt[2] = zanch;
        if (type != OV2_BA_RANCH_INV) {
            pose_Rt(poses + 7 * P->lm_anchor_pose[l], Rwa, twa);
            mat3_vec(Rwa, anchpt, wpt);
            wpt[0] += twa[0]; wpt[1] += twa[1]; wpt[2] += twa[2];
        }
    } else {
        wpt[0] = lms[3 * l]; wpt[1] = lms[3 * l + 1]; wpt[2] = lms[3 * l + 2];
    }

    double Rwc[9], twc[3], lcam[3], cam[3];
    double M[9]; /* rotation that maps world (or anchor-camera, for RANCH) directions into the measuring camera */
    if (type == OV2_BA_RANCH_INV) {
        mat3_vec(Rrl, anchpt, cam);
        cam[0] += trl[0]; cam[1] += trl[1]; cam[2] += trl[2];
        memcpy(M, Rrl, sizeof(M));
    } else {
        pose_Rt(poses + 7 * P->res_pose[i], Rwc, twc);
        const double d[3] = {wpt[0] - twc[0], wpt[1] - twc[1], wpt[2] - twc[2]};
        mat3T_vec(Rwc, d, lcam); /* Tcw * wpt */
        if (is_right) {
            mat3_vec(Rrl, lcam, cam);
            cam[0] += trl[0]; cam[1] += trl[1]; cam[2] += trl[2];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) {
                    double s = 0;
                    for (int k = 0; k < 3; ++k) s += Rrl[3 * r + k] * Rwc[3 * c + k]; /* Rrl * Rcw */
                    M[3 * r + c] = s;
                }
        } else {
            memcpy(cam, lcam, sizeof(cam));
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) M[3 * r + c] = Rwc[3 * c + r]; /* Rcw */
        }
    }
    const double invz = 1.0 / cam[2];
    const double pu = K[0] * cam[0] * invz + K[2], pv = K[1] * cam[1] * invz + K[3];
    o->r[0] = inv_sigma * (pu - P->res_uv[2 * i]);
    o->r[1] = inv_sigma * (pv - P->res_uv[2 * i + 1]);
    o->chi2 = o->r[0] * o->r[0] + o->r[1] * o->r[1];
    o->depth_positive = cam[2] > 0.0;
    if (!want_jac) return;

    const double invz2 = invz * invz;
    const double Jc[6] = {invz * K[0], 0.0, -cam[0] * invz2 * K[0], 0.0, invz * K[1], -cam[1] * invz2 * K[1]};
    double JR[6];
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 3; ++c)
            JR[3 * r + c] = Jc[3 * r] * M[c] + Jc[3 * r + 1] * M[3 + c] + Jc[3 * r + 2] * M[6 + c];
    memset(o->Jk, 0, sizeof(o->Jk));
    memset(o->Ja, 0, sizeof(o->Ja));
    memset(o->Jl, 0, sizeof(o->Jl));
    if (type != OV2_BA_RANCH_INV) {
        double JRh[6];
        j_hat(JR, wpt, JRh);
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 3; ++c) {
                o->Jk[6 * r + c] = -inv_sigma * JR[3 * r + c];      /* [-J_R | J_R hat(wpt)] */
                o->Jk[6 * r + 3 + c] = inv_sigma * JRh[3 * r + c];
                if (inv) {
                    o->Ja[6 * r + c] = inv_sigma * JR[3 * r + c];   /* [ J_R | -J_R hat(wpt)] */
                    o->Ja[6 * r + 3 + c] = -inv_sigma * JRh[3 * r + c];
                }
            }
    }
    if (inv) {
        double jl[3];
        if (type == OV2_BA_RANCH_INV) {
            jl[0] = -zanch * anchpt[0]; jl[1] = -zanch * anchpt[1]; jl[2] = -zanch * anchpt[2];
        } else {
            double t[3];
            mat3_vec(Rwa, anchpt, t);
            jl[0] = -zanch * t[0]; jl[1] = -zanch * t[1]; jl[2] = -zanch * t[2];
        }
        for (int r = 0; r < 2; ++r)
            o->Jl[r] = inv_sigma * (JR[3 * r] * jl[0] + JR[3 * r + 1] * jl[1] + JR[3 * r + 2] * jl[2]);
    } else {
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 3; ++c) o->Jl[3 * r + c] = inv_sigma * JR[3 * r + c];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* loss + corrector                                                                             */

void ov2o_huber(double a, double s, double rho[3])
{
    const double b = a * a;
    if (s > b) {
        const double r = sqrt(s);
        rho[0] = 2.0 * a * r - b;
        rho[1] = a / r;
        if (rho[1] < DBL_MIN) rho[1] = DBL_MIN;
        rho[2] = -rho[1] / (2.0 * s);
    } else {
        rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
}

/* Corrector (corrector.cc:42-157) on a residual of `nr` rows with `nblk` jacobian blocks */
void ov2o_corrector(double sq_norm, const double rho[3], int nr, double *res, int nblk, double **jac, const int *ncols)
{
    const double sqrt_rho1 = sqrt(rho[1]);
    double residual_scaling, alpha_sq_norm;
    if (sq_norm == 0.0 || rho[2] <= 0.0) {
        residual_scaling = sqrt_rho1;
        alpha_sq_norm = 0.0;
    } else {
        const double D = 1.0 + 2.0 * sq_norm * rho[2] / rho[1];
        const double alpha = 1.0 - sqrt(D);
        residual_scaling = sqrt_rho1 / (1 - alpha);
        alpha_sq_norm = alpha / sq_norm;
    }
    /* residual_block.cc:190-200: jacobians are corrected first (they need the uncorrected residual) */
    for (int b = 0; b < nblk; ++b) {
        double *J = jac[b];
        if (!J) continue;
        const int nc = ncols[b];
        if (alpha_sq_norm == 0.0) {
            for (int k = 0; k < nr * nc; ++k) J[k] *= sqrt_rho1;
            continue;
        }
        for (int c = 0; c < nc; ++c) {
            double rtj = 0.0;
            for (int r = 0; r < nr; ++r) rtj += J[r * nc + c] * res[r];
            for (int r = 0; r < nr; ++r) J[r * nc + c] = sqrt_rho1 * (J[r * nc + c] - alpha_sq_norm * res[r] * rtj);
        }
    }
    for (int r = 0; r < nr; ++r) res[r] *= residual_scaling;
}

/* ------------------------------------------------------------------------------------------ */
/* block-sparse Schur solve (generic sizes; SchurEliminator + dense Cholesky of the reduced system) */

static int chol_inplace(double *A, int n) /* lower Cholesky in place; 0 = ok */
{
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0.0)) return -1;
        d = sqrt(d);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * n + j];
            const double *ri = A + (size_t)i * n, *rj = A + (size_t)j * n;
            for (int k = 0; k < j; ++k) s -= ri[k] * rj[k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    return 0;
}

static void chol_solve(const double *L, int n, double *b)
{
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[(size_t)i * n + k] * b[k];
        b[i] = s / L[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= L[(size_t)k * n + i] * b[k];
        b[i] = s / L[(size_t)i * n + i];
    }
}

/* inverse of a small SPD matrix (invert_psd_matrix.h: LLT solve against identity) */
static int inv_spd(const double *A, int n, double *out)
{
    double L[64];
    if (n > 8) return -1;
    memcpy(L, A, sizeof(double) * n * n);
    if (chol_inplace(L, n)) return -1;
    for (int c = 0; c < n; ++c) {
        double e[8] = {0};
        e[c] = 1.0;
        chol_solve(L, n, e);
        for (int r = 0; r < n; ++r) out[r * n + c] = e[r];
    }
    return 0;
}

int ov2o_schur_solve(const ov2o_bs_problem *p, double *S_out, double *rhs_out, double *x)
{
    const int R = p->R, E = p->E, F = p->F, MF = p->maxf;
    const int m = p->n_f * F, ne = p->n_e * E;
    double *S = (double *)calloc((size_t)(m ? m : 1) * (m ? m : 1), sizeof(double));
    double *rhs = (double *)calloc((size_t)(m ? m : 1), sizeof(double));
    double *inv_ete_all = (double *)calloc((size_t)(p->n_e ? p->n_e : 1) * E * E, sizeof(double));
    int rc = 0;
    /* D_f^2 on the diagonal */
    if (p->D)
        for (int k = 0; k < m; ++k) S[(size_t)k * m + k] += p->D[ne + k] * p->D[ne + k];
    /* rows by e block */
    int *cnt = (int *)calloc((size_t)p->n_e + 1, sizeof(int));
    for (int r = 0; r < p->n_rows; ++r)
        if (p->row_e[r] >= 0) cnt[p->row_e[r] + 1]++;
    for (int e = 0; e < p->n_e; ++e) cnt[e + 1] += cnt[e];
    int *rows = (int *)malloc(sizeof(int) * (size_t)(cnt[p->n_e] ? cnt[p->n_e] : 1));
    int *fill = (int *)calloc((size_t)p->n_e + 1, sizeof(int));
    for (int r = 0; r < p->n_rows; ++r)
        if (p->row_e[r] >= 0) rows[cnt[p->row_e[r]] + fill[p->row_e[r]]++] = r;

    int cap = 64;
    int *flist = (int *)malloc(sizeof(int) * cap);
    double *buf = (double *)malloc(sizeof(double) * cap * E * F);
    double ete[64], g[8], ieg[8];

    for (int e = 0; e < p->n_e && rc == 0; ++e) {
        const int n_r = cnt[e + 1] - cnt[e];
        if (n_r * MF > cap) {
            cap = n_r * MF;
            flist = (int *)realloc(flist, sizeof(int) * cap);
            buf = (double *)realloc(buf, sizeof(double) * cap * E * F);
        }
        int nfl = 0;
        memset(ete, 0, sizeof(double) * E * E);
        memset(g, 0, sizeof(double) * E);
        if (p->D)
            for (int k = 0; k < E; ++k) ete[k * E + k] = p->D[e * E + k] * p->D[e * E + k];
        for (int q = 0; q < n_r; ++q) {
            const int r = rows[cnt[e] + q];
            const double *Je = p->Je + (size_t)r * R * E;
            const double *b = p->b + (size_t)r * R;
            for (int i = 0; i < E; ++i) {
                for (int j = 0; j < E; ++j) {
                    double s = 0;
                    for (int k = 0; k < R; ++k) s += Je[k * E + i] * Je[k * E + j];
                    ete[i * E + j] += s;
                }
                double s = 0;
                for (int k = 0; k < R; ++k) s += Je[k * E + i] * b[k];
                g[i] += s;
            }
            for (int c = 0; c < MF; ++c) {
                const int fb = p->row_f[(size_t)r * MF + c];
                if (fb < 0) continue;
                const double *Jf = p->Jf + ((size_t)r * MF + c) * R * F;
                int pos = -1;
                for (int k = 0; k < nfl; ++k) if (flist[k] == fb) { pos = k; break; }
                if (pos < 0) { pos = nfl++; flist[pos] = fb; memset(buf + (size_t)pos * E * F, 0, sizeof(double) * E * F); }
                double *B = buf + (size_t)pos * E * F;  /* E' F */
                for (int i = 0; i < E; ++i)
                    for (int j = 0; j < F; ++j) {
                        double s = 0;
                        for (int k = 0; k < R; ++k) s += Je[k * E + i] * Jf[k * F + j];
                        B[i * F + j] += s;
                    }
                /* EBlockRowOuterProduct: F_c' F_c2 for every pair of f cells of the row */
                for (int c2 = 0; c2 < MF; ++c2) {
                    const int fb2 = p->row_f[(size_t)r * MF + c2];
                    if (fb2 < 0) continue;
                    const double *Jf2 = p->Jf + ((size_t)r * MF + c2) * R * F;
                    for (int i = 0; i < F; ++i)
                        for (int j = 0; j < F; ++j) {
                            double s = 0;
                            for (int k = 0; k < R; ++k) s += Jf[k * F + i] * Jf2[k * F + j];
                            S[(size_t)(fb * F + i) * m + fb2 * F + j] += s;
                        }
                }
            }
        }
        double *iete = inv_ete_all + (size_t)e * E * E;
        if (inv_spd(ete, E, iete)) { rc = -2; break; }
        for (int i = 0; i < E; ++i) {
            double s = 0;
            for (int j = 0; j < E; ++j) s += iete[i * E + j] * g[j];
            ieg[i] = s;
        }
        /* rhs += F'(b - E inv(ete) g) */
        for (int q = 0; q < n_r; ++q) {
            const int r = rows[cnt[e] + q];
            const double *Je = p->Je + (size_t)r * R * E;
            double sj[8];
            for (int k = 0; k < R; ++k) {
                double s = p->b[(size_t)r * R + k];
                for (int i = 0; i < E; ++i) s -= Je[k * E + i] * ieg[i];
                sj[k] = s;
            }
            for (int c = 0; c < MF; ++c) {
                const int fb = p->row_f[(size_t)r * MF + c];
                if (fb < 0) continue;
                const double *Jf = p->Jf + ((size_t)r * MF + c) * R * F;
                for (int j = 0; j < F; ++j) {
                    double s = 0;
                    for (int k = 0; k < R; ++k) s += Jf[k * F + j] * sj[k];
                    rhs[fb * F + j] += s;
                }
            }
        }
        /* S -= (E'F)' inv(ete) (E'F) over all f-block pairs of the chunk (ChunkOuterProduct) */
        for (int a = 0; a < nfl; ++a) {
            double T[8 * 8]; /* inv(ete) * B_a' ... compute B_a' * iete : F x E */
            const double *Ba = buf + (size_t)a * E * F;
            for (int i = 0; i < F; ++i)
                for (int j = 0; j < E; ++j) {
                    double s = 0;
                    for (int k = 0; k < E; ++k) s += Ba[k * F + i] * iete[k * E + j];
                    T[i * E + j] = s;
                }
            for (int b2 = 0; b2 < nfl; ++b2) {
                const double *Bb = buf + (size_t)b2 * E * F;
                for (int i = 0; i < F; ++i)
                    for (int j = 0; j < F; ++j) {
                        double s = 0;
                        for (int k = 0; k < E; ++k) s += T[i * E + k] * Bb[k * F + j];
                        S[(size_t)(flist[a] * F + i) * m + flist[b2] * F + j] -= s;
                    }
            }
        }
    }
    /* rows without an e block (NoEBlockRowsUpdate) */
    for (int r = 0; r < p->n_rows && rc == 0; ++r) {
        if (p->row_e[r] >= 0) continue;
        for (int c = 0; c < MF; ++c) {
            const int fb = p->row_f[(size_t)r * MF + c];
            if (fb < 0) continue;
            const double *Jf = p->Jf + ((size_t)r * MF + c) * R * F;
            for (int j = 0; j < F; ++j) {
                double s = 0;
                for (int k = 0; k < R; ++k) s += Jf[k * F + j] * p->b[(size_t)r * R + k];
                rhs[fb * F + j] += s;
            }
            for (int c2 = 0; c2 < MF; ++c2) {
                const int fb2 = p->row_f[(size_t)r * MF + c2];
                if (fb2 < 0) continue;
                const double *Jf2 = p->Jf + ((size_t)r * MF + c2) * R * F;
                for (int i = 0; i < F; ++i)
                    for (int j = 0; j < F; ++j) {
                        double s = 0;
                        for (int k = 0; k < R; ++k) s += Jf[k * F + i] * Jf2[k * F + j];
                        S[(size_t)(fb * F + i) * m + fb2 * F + j] += s;
                    }
            }
        }
    }
    if (rc == 0) {
        if (S_out) memcpy(S_out, S, sizeof(double) * (size_t)m * m);
        if (rhs_out) memcpy(rhs_out, rhs, sizeof(double) * (size_t)m);
        double *z = x + ne;
        memcpy(z, rhs, sizeof(double) * (size_t)m);
        if (m > 0) {
            if (chol_inplace(S, m)) rc = -1;
            else chol_solve(S, m, z);
        }
    }
    /* back-substitution: y_e = inv(ete) sum E'(b - sum F z) */
    for (int e = 0; e < p->n_e && rc == 0; ++e) {
        double acc[8] = {0};
        for (int q = cnt[e]; q < cnt[e + 1]; ++q) {
            const int r = rows[q];
            double sj[8];
            for (int k = 0; k < R; ++k) sj[k] = p->b[(size_t)r * R + k];
            for (int c = 0; c < MF; ++c) {
                const int fb = p->row_f[(size_t)r * MF + c];
                if (fb < 0) continue;
                const double *Jf = p->Jf + ((size_t)r * MF + c) * R * F;
                for (int k = 0; k < R; ++k) {
                    double s = 0;
                    for (int j = 0; j < F; ++j) s += Jf[k * F + j] * x[ne + fb * F + j];
                    sj[k] -= s;
                }
            }
            const double *Je = p->Je + (size_t)r * R * E;
            for (int i = 0; i < E; ++i)
                for (int k = 0; k < R; ++k) acc[i] += Je[k * E + i] * sj[k];
        }
        const double *iete = inv_ete_all + (size_t)e * E * E;
        for (int i = 0; i < E; ++i) {
            double s = 0;
            for (int j = 0; j < E; ++j) s += iete[i * E + j] * acc[j];
            x[e * E + i] = s;
        }
    }
    free(S); free(rhs); free(inv_ete_all); free(cnt); free(rows); free(fill); free(flist); free(buf);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* reduced program + evaluator                                                                  */

typedef struct {
    const ov2_ba_problem *P;
    int e;                 /* landmark block size (1 or 3) */
    int n_e, n_f, n_act;   /* free landmark blocks, free pose blocks, active residuals */
    int *eidx, *fidx;      /* landmark -> e block / pose -> f block (or -1) */
    int *lm_of_e, *pose_of_f;
    int *act;              /* active residual ids */
    int use_loss;
    double huber_a;
    /* evaluation outputs (row = active residual) */
    double *res;           /* n_act x 2 */
    double *Je;            /* n_act x 2 x e */
    double *Jf;            /* n_act x 2 cells x 2 x 6 : cell 0 = observing pose, cell 1 = anchor pose */
    int *row_e, *row_f;    /* n_act, n_act x 2 */
    int ncols;
} prog;

static void prog_free(prog *g)
{
    free(g->eidx); free(g->fidx); free(g->lm_of_e); free(g->pose_of_f); free(g->act);
    free(g->res); free(g->Je); free(g->Jf); free(g->row_e); free(g->row_f);
}

/* program.cc RemoveFixedBlocks: constant and unused parameter blocks leave the program */
static void prog_build(prog *g, const ov2_ba_problem *P, const uint8_t *active, int use_loss, double huber_a)
{
    memset(g, 0, sizeof(*g));
    g->P = P;
    g->e = P->inv_depth ? 1 : 3;
    g->use_loss = use_loss;
    g->huber_a = huber_a;
    g->eidx = (int *)malloc(sizeof(int) * (size_t)(P->n_lm + 1));
    g->fidx = (int *)malloc(sizeof(int) * (size_t)(P->n_pose + 1));
    g->act = (int *)malloc(sizeof(int) * (size_t)(P->n_res + 1));
    for (int l = 0; l < P->n_lm; ++l) g->eidx[l] = -1;
    for (int p = 0; p < P->n_pose; ++p) g->fidx[p] = -1;
    for (int i = 0; i < P->n_res; ++i) {
        if (active && !active[i]) continue;
        g->act[g->n_act++] = i;
        g->eidx[P->res_lm[i]] = 0;
        const int t = P->res_type[i];
        if (t != OV2_BA_RANCH_INV && !P->pose_const[P->res_pose[i]]) g->fidx[P->res_pose[i]] = 0;
        if ((t == OV2_BA_L_INV || t == OV2_BA_R_INV) && !P->pose_const[P->lm_anchor_pose[P->res_lm[i]]])
            g->fidx[P->lm_anchor_pose[P->res_lm[i]]] = 0;
    }
    g->lm_of_e = (int *)malloc(sizeof(int) * (size_t)(P->n_lm + 1));
    g->pose_of_f = (int *)malloc(sizeof(int) * (size_t)(P->n_pose + 1));
    for (int l = 0; l < P->n_lm; ++l)
        if (g->eidx[l] == 0) { g->eidx[l] = g->n_e; g->lm_of_e[g->n_e++] = l; }
    for (int p = 0; p < P->n_pose; ++p)
        if (g->fidx[p] == 0) { g->fidx[p] = g->n_f; g->pose_of_f[g->n_f++] = p; }
    g->ncols = g->n_e * g->e + g->n_f * 6;
    const size_t na = (size_t)(g->n_act ? g->n_act : 1);
    g->res = (double *)calloc(na * 2, sizeof(double));
    g->Je = (double *)calloc(na * 2 * g->e, sizeof(double));
    g->Jf = (double *)calloc(na * 2 * 12, sizeof(double));
    g->row_e = (int *)malloc(sizeof(int) * na);
    g->row_f = (int *)malloc(sizeof(int) * na * 2);
    for (int a = 0; a < g->n_act; ++a) {
        const int i = g->act[a], t = P->res_type[i];
        g->row_e[a] = g->eidx[P->res_lm[i]];
        g->row_f[2 * a] = (t == OV2_BA_RANCH_INV) ? -1 : g->fidx[P->res_pose[i]];
        g->row_f[2 * a + 1] = (t == OV2_BA_L_INV || t == OV2_BA_R_INV) ? g->fidx[P->lm_anchor_pose[P->res_lm[i]]] : -1;
    }
}

/* ProgramEvaluator::Evaluate (+ ResidualBlock::Evaluate): cost = 1/2 sum rho(|r|^2); residuals and jacobians
 * robustified by the corrector; local parameterisation = first 6 columns of the 7-wide pose jacobian */
static double prog_evaluate(prog *g, const double *poses, const double *lms, int want_jac)
{
    double cost = 0.0;
    const int e = g->e;
    for (int a = 0; a < g->n_act; ++a) {
        ov2o_res_eval ev;
        ov2o_ba_eval_residual(g->P, poses, lms, g->act[a], want_jac, &ev);
        const double s = ev.chi2;
        double rho[3] = {s, 1.0, 0.0};
        if (g->use_loss) ov2o_huber(g->huber_a, s, rho);
        cost += 0.5 * rho[0];
        if (!want_jac) continue;
        double r[2] = {ev.r[0], ev.r[1]};
        if (g->use_loss) {
            double *jac[3] = {ev.Jk, ev.Ja, ev.Jl};
            const int nc[3] = {6, 6, e};
            ov2o_corrector(s, rho, 2, r, 3, jac, nc);
        }
        g->res[2 * a] = r[0]; g->res[2 * a + 1] = r[1];
        memcpy(g->Je + (size_t)a * 2 * e, ev.Jl, sizeof(double) * 2 * e);
        memcpy(g->Jf + (size_t)a * 24, ev.Jk, sizeof(double) * 12);
        memcpy(g->Jf + (size_t)a * 24 + 12, ev.Ja, sizeof(double) * 12);
    }
    return cost;
}

/* squared column norms of the (scaled) jacobian, and J'r */
static void prog_colnorm_grad(const prog *g, double *sqn, double *grad)
{
    const int e = g->e, ne = g->n_e * e;
    memset(sqn, 0, sizeof(double) * (size_t)g->ncols);
    if (grad) memset(grad, 0, sizeof(double) * (size_t)g->ncols);
    for (int a = 0; a < g->n_act; ++a) {
        const double *Je = g->Je + (size_t)a * 2 * e, *r = g->res + 2 * a;
        const int eb = g->row_e[a];
        for (int c = 0; c < e; ++c) {
            sqn[eb * e + c] += Je[c] * Je[c] + Je[e + c] * Je[e + c];
            if (grad) grad[eb * e + c] += Je[c] * r[0] + Je[e + c] * r[1];
        }
        for (int k = 0; k < 2; ++k) {
            const int fb = g->row_f[2 * a + k];
            if (fb < 0) continue;
            const double *Jf = g->Jf + (size_t)a * 24 + 12 * k;
            for (int c = 0; c < 6; ++c) {
                sqn[ne + fb * 6 + c] += Jf[c] * Jf[c] + Jf[6 + c] * Jf[6 + c];
                if (grad) grad[ne + fb * 6 + c] += Jf[c] * r[0] + Jf[6 + c] * r[1];
            }
        }
    }
}

static void prog_scale_columns(prog *g, const double *sc)
{
    const int e = g->e, ne = g->n_e * e;
    for (int a = 0; a < g->n_act; ++a) {
        double *Je = g->Je + (size_t)a * 2 * e;
        const int eb = g->row_e[a];
        for (int c = 0; c < e; ++c) { Je[c] *= sc[eb * e + c]; Je[e + c] *= sc[eb * e + c]; }
        for (int k = 0; k < 2; ++k) {
            const int fb = g->row_f[2 * a + k];
            if (fb < 0) continue;
            double *Jf = g->Jf + (size_t)a * 24 + 12 * k;
            for (int c = 0; c < 6; ++c) { Jf[c] *= sc[ne + fb * 6 + c]; Jf[6 + c] *= sc[ne + fb * 6 + c]; }
        }
    }
}

/* Evaluator::Plus on the free blocks: poses by SE3 left update, landmarks additive */
static void prog_plus(const prog *g, const double *poses, const double *lms, const double *delta, double *oposes,
                      double *olms)
{
    const ov2_ba_problem *P = g->P;
    const int e = g->e, ne = g->n_e * e;
    memcpy(oposes, poses, sizeof(double) * 7 * (size_t)P->n_pose);
    memcpy(olms, lms, sizeof(double) * (size_t)e * P->n_lm);
    for (int k = 0; k < g->n_e; ++k)
        for (int c = 0; c < e; ++c) olms[g->lm_of_e[k] * e + c] = lms[g->lm_of_e[k] * e + c] + delta[k * e + c];
    for (int k = 0; k < g->n_f; ++k) ov2o_se3_plus(poses + 7 * g->pose_of_f[k], delta + ne + 6 * k, oposes + 7 * g->pose_of_f[k]);
}

static double prog_xnorm2_diff(const prog *g, const double *pa, const double *la, const double *pb, const double *lb)
{
    /* || x_a - x_b ||^2 over the free blocks in GLOBAL size (7 per pose); pb == NULL -> ||x_a||^2 */
    const int e = g->e;
    double s = 0;
    for (int k = 0; k < g->n_e; ++k)
        for (int c = 0; c < e; ++c) {
            const int ix = g->lm_of_e[k] * e + c;
            const double d = la[ix] - (lb ? lb[ix] : 0.0);
            s += d * d;
        }
    for (int k = 0; k < g->n_f; ++k)
        for (int c = 0; c < 7; ++c) {
            const int ix = g->pose_of_f[k] * 7 + c;
            const double d = pa[ix] - (pb ? pb[ix] : 0.0);
            s += d * d;
        }
    return s;
}

/* ------------------------------------------------------------------------------------------ */
/* TrustRegionMinimizer + LevenbergMarquardtStrategy                                            */

static void log_iter(ov2_ba_result *R, double cost, double change, double radius, double rel, double model, int valid,
                     int ok)
{
    if (!R || R->n_log >= OV2_BA_MAX_LOG) return;
    ov2_ba_iter *it = &R->log[R->n_log++];
    it->cost = cost; it->cost_change = change; it->radius = radius; it->relative_decrease = rel;
    it->model_cost_change = model; it->step_is_valid = valid; it->step_is_successful = ok;
}

/* LevenbergMarquardtStrategy: the three state updates minimize() uses, exported so that the values Ceres' own test holds
 * (levenberg_marquardt_strategy_test.cc:81-150) pin THIS code (tests/test_oracle_ba.py). */
void ov2o_lm_step_accepted(double *radius, double *decrease_factor, double step_quality, double max_radius)
{   /* StepAccepted, levenberg_marquardt_strategy.cc:147-153 */
    *radius = *radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * step_quality - 1.0, 3));
    *radius = fmin(max_radius, *radius);
    *decrease_factor = 2.0;
}

void ov2o_lm_step_rejected(double *radius, double *decrease_factor)
{   /* StepRejected / StepIsInvalid, :155-164 */
    *radius = *radius / *decrease_factor;
    *decrease_factor *= 2.0;
}

void ov2o_lm_diagonal(int n, const double *colnorm2, double min_diag, double max_diag, double radius, double *diag, double *D)
{   /* ComputeStep :76-89: diagonal_ = clamp(J'J diagonal), lm_diagonal_ = sqrt(diagonal_ / radius_) */
    for (int k = 0; k < n; ++k) {
        if (colnorm2) diag[k] = fmin(fmax(colnorm2[k], min_diag), max_diag);
        D[k] = sqrt(diag[k] / radius);
    }
}

/* eval_pose / eval_lm receive the state of the LAST residual evaluation (ResidualBlock::Evaluate -> the cost functors'
 * cached chi2err_ / isdepthpositive_, src/ceres_parametrization.cpp:136-146): x after IterationZero or an accepted step,
 * the candidate after a rejected step or a FTOL / PTOL exit (trust_region_minimizer.cc:108-131; Ceres does not
 * re-evaluate after Solve).  The reference flags outliers from those cached values (src/optimizer.cpp:500-592). */
static int minimize(const ov2_ba_problem *P, double *poses, double *lms, const uint8_t *active, int use_loss,
                    const ov2_ba_options *o, int max_iters, ov2_ba_result *R, double *initial_cost, double *final_cost,
                    double *eval_pose, double *eval_lm)
{
    prog g;
    prog_build(&g, P, active, use_loss, o->huber_delta);
    const int e = g.e, nc = g.ncols, ne = g.n_e * e;
    int term = OV2_BA_TERM_MAX_ITER;
    const size_t npose = (size_t)P->n_pose * 7, nlm = (size_t)P->n_lm * e;
    memcpy(eval_pose, poses, sizeof(double) * npose);
    memcpy(eval_lm, lms, sizeof(double) * nlm);
    if (g.n_act == 0 || nc == 0) {
        *initial_cost = *final_cost = 0.0;
        prog_free(&g);
        return OV2_BA_TERM_SKIPPED;
    }
    double *xp = (double *)malloc(sizeof(double) * npose), *xl = (double *)malloc(sizeof(double) * (nlm ? nlm : 1));
    double *cp = (double *)malloc(sizeof(double) * npose), *cl = (double *)malloc(sizeof(double) * (nlm ? nlm : 1));
    double *scale = (double *)malloc(sizeof(double) * nc), *grad = (double *)malloc(sizeof(double) * nc);
    double *diag = (double *)malloc(sizeof(double) * nc), *lmd = (double *)malloc(sizeof(double) * nc);
    double *step = (double *)malloc(sizeof(double) * nc), *delta = (double *)malloc(sizeof(double) * nc);
    double *tmp = (double *)malloc(sizeof(double) * nc);
    memcpy(xp, poses, sizeof(double) * npose);
    memcpy(xl, lms, sizeof(double) * nlm);
    for (int k = 0; k < nc; ++k) scale[k] = 1.0;

    /* IterationZero -> EvaluateGradientAndJacobian */
    double x_cost = prog_evaluate(&g, xp, xl, 1);
    prog_colnorm_grad(&g, tmp, grad);
    if (o->jacobi_scaling) {
        for (int k = 0; k < nc; ++k) scale[k] = 1.0 / (1.0 + sqrt(tmp[k]));
        prog_scale_columns(&g, scale);
    }
    double gmax = 0.0;
    {   /* gradient_max_norm = || x - Plus(x, -g) ||_inf */
        for (int k = 0; k < nc; ++k) tmp[k] = -grad[k];
        prog_plus(&g, xp, xl, tmp, cp, cl);
        for (int k = 0; k < g.n_e; ++k)
            for (int c = 0; c < e; ++c) gmax = fmax(gmax, fabs(xl[g.lm_of_e[k] * e + c] - cl[g.lm_of_e[k] * e + c]));
        for (int k = 0; k < g.n_f; ++k)
            for (int c = 0; c < 7; ++c) gmax = fmax(gmax, fabs(xp[g.pose_of_f[k] * 7 + c] - cp[g.pose_of_f[k] * 7 + c]));
    }
    *initial_cost = x_cost;
    double minimum_cost = x_cost;                 /* parameters_ <- x_ */
    memcpy(poses, xp, sizeof(double) * npose);
    memcpy(lms, xl, sizeof(double) * nlm);
    double x_norm = -1.0;                         /* Init(): "invalid value" until the first successful step */
    double radius = o->initial_radius, decrease_factor = 2.0;
    int reuse_diagonal = 0, invalid_steps = 0, iteration = 0;
    int last_ok = 1;
    log_iter(R, x_cost, 0.0, radius, 0.0, 0.0, 1, 1);

    for (;;) {
        /* FinalizeIterationAndCheckIfMinimizerCanContinue (time cap not applied) */
        if (iteration >= max_iters) { term = OV2_BA_TERM_MAX_ITER; break; }
        if (last_ok && gmax <= o->gradient_tolerance) { term = OV2_BA_TERM_GTOL; break; }
        if (radius <= o->min_radius) { term = OV2_BA_TERM_MIN_RADIUS; break; }
        ++iteration;

        /* LevenbergMarquardtStrategy::ComputeStep */
        if (!reuse_diagonal) {
            prog_colnorm_grad(&g, tmp, NULL);
            ov2o_lm_diagonal(nc, tmp, o->min_lm_diagonal, o->max_lm_diagonal, radius, diag, lmd);
        } else {
            ov2o_lm_diagonal(nc, NULL, o->min_lm_diagonal, o->max_lm_diagonal, radius, diag, lmd);
        }
        ov2o_bs_problem bs;
        bs.R = 2; bs.E = e; bs.F = 6; bs.maxf = 2; bs.n_rows = g.n_act; bs.n_e = g.n_e; bs.n_f = g.n_f;
        bs.row_e = g.row_e; bs.row_f = g.row_f; bs.Je = g.Je; bs.Jf = g.Jf; bs.b = g.res; bs.D = lmd;
        const int lin = ov2o_schur_solve(&bs, NULL, NULL, step);
        reuse_diagonal = 1;
        int finite = (lin == 0);
        if (finite) for (int k = 0; k < nc; ++k) if (!isfinite(step[k])) { finite = 0; break; }
        double model_change = 0.0;
        int valid = 0;
        if (finite) {
            for (int k = 0; k < nc; ++k) step[k] = -step[k];
            /* model_cost_change = -m.(r + m/2), m = J step */
            for (int a = 0; a < g.n_act; ++a) {
                double m[2] = {0, 0};
                const double *Je = g.Je + (size_t)a * 2 * e;
                for (int c = 0; c < e; ++c) { m[0] += Je[c] * step[g.row_e[a] * e + c]; m[1] += Je[e + c] * step[g.row_e[a] * e + c]; }
                for (int k = 0; k < 2; ++k) {
                    const int fb = g.row_f[2 * a + k];
                    if (fb < 0) continue;
                    const double *Jf = g.Jf + (size_t)a * 24 + 12 * k;
                    for (int c = 0; c < 6; ++c) { m[0] += Jf[c] * step[ne + fb * 6 + c]; m[1] += Jf[6 + c] * step[ne + fb * 6 + c]; }
                }
                model_change -= m[0] * (g.res[2 * a] + m[0] / 2.0) + m[1] * (g.res[2 * a + 1] + m[1] / 2.0);
            }
            valid = model_change > 0.0;
        }
        if (!valid) {   /* HandleInvalidStep */
            if (++invalid_steps >= o->max_consecutive_invalid_steps) { term = OV2_BA_TERM_FAILURE; break; }
            ov2o_lm_step_rejected(&radius, &decrease_factor); reuse_diagonal = 1;   /* StepIsInvalid == StepRejected */
            last_ok = 0;
            log_iter(R, x_cost, 0.0, radius, 0.0, model_change, 0, 0);
            continue;
        }
        invalid_steps = 0;
        for (int k = 0; k < nc; ++k) delta[k] = step[k] * scale[k];
        prog_plus(&g, xp, xl, delta, cp, cl);
        const double cand_cost = prog_evaluate(&g, cp, cl, 0);
        memcpy(eval_pose, cp, sizeof(double) * npose);   /* the functors now hold the candidate's chi2 / depth sign */
        memcpy(eval_lm, cl, sizeof(double) * nlm);
        /* ParameterToleranceReached */
        const double step_norm = sqrt(prog_xnorm2_diff(&g, xp, xl, cp, cl));
        if (step_norm <= o->parameter_tolerance * (x_norm + o->parameter_tolerance)) { term = OV2_BA_TERM_PTOL; break; }
        /* FunctionToleranceReached: returns WITHOUT taking the candidate */
        const double cost_change = x_cost - cand_cost;
        if (fabs(cost_change) <= o->function_tolerance * x_cost) {
            term = OV2_BA_TERM_FTOL;
            log_iter(R, x_cost, cost_change, radius, 0.0, model_change, 1, 0);
            break;
        }
        const double rel = (cand_cost >= DBL_MAX) ? -DBL_MAX : (x_cost - cand_cost) / model_change;
        if (rel > o->min_relative_decrease) {   /* HandleSuccessfulStep */
            memcpy(xp, cp, sizeof(double) * npose);
            memcpy(xl, cl, sizeof(double) * nlm);
            x_norm = sqrt(prog_xnorm2_diff(&g, xp, xl, NULL, NULL));
            x_cost = prog_evaluate(&g, xp, xl, 1);   /* same point as the candidate just evaluated */
            prog_colnorm_grad(&g, tmp, grad);
            if (o->jacobi_scaling) prog_scale_columns(&g, scale);
            gmax = 0.0;
            for (int k = 0; k < nc; ++k) tmp[k] = -grad[k];
            prog_plus(&g, xp, xl, tmp, cp, cl);
            for (int k = 0; k < g.n_e; ++k)
                for (int c = 0; c < e; ++c) gmax = fmax(gmax, fabs(xl[g.lm_of_e[k] * e + c] - cl[g.lm_of_e[k] * e + c]));
            for (int k = 0; k < g.n_f; ++k)
                for (int c = 0; c < 7; ++c) gmax = fmax(gmax, fabs(xp[g.pose_of_f[k] * 7 + c] - cp[g.pose_of_f[k] * 7 + c]));
            ov2o_lm_step_accepted(&radius, &decrease_factor, rel, o->max_radius);   /* StepAccepted */
            reuse_diagonal = 0;
            last_ok = 1;
            if (x_cost < minimum_cost) {
                minimum_cost = x_cost;
                memcpy(poses, xp, sizeof(double) * npose);
                memcpy(lms, xl, sizeof(double) * nlm);
            }
            log_iter(R, x_cost, cost_change, radius, rel, model_change, 1, 1);
        } else {                                /* StepRejected */
            ov2o_lm_step_rejected(&radius, &decrease_factor); reuse_diagonal = 1;
            last_ok = 0;
            log_iter(R, cand_cost, cost_change, radius, rel, model_change, 1, 0);
        }
    }
    *final_cost = minimum_cost;
    free(xp); free(xl); free(cp); free(cl); free(scale); free(grad); free(diag); free(lmd); free(step); free(delta); free(tmp);
    prog_free(&g);
    return term;
}

/* ------------------------------------------------------------------------------------------ */
/* Optimizer::localBA numerical core: robust solve -> flag -> L2 re-solve -> flag                */

void ov2o_ba_default_options(ov2_ba_options *o, float robust_mono_th)
{
    memset(o, 0, sizeof(*o));
    o->huber_delta = (double)sqrtf(robust_mono_th);  /* HuberLoss(std::sqrt(mono_th)) with float mono_th */
    o->chi2_th = (double)robust_mono_th;
    o->max_iters = 5;
    o->l2_refine = 1;
    o->l2_max_iters = 10;
    o->function_tolerance = 1e-3;
    o->initial_radius = 1e4; o->max_radius = 1e16; o->min_radius = 1e-32;
    o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32;
    o->min_relative_decrease = 1e-3; o->parameter_tolerance = 1e-8; o->gradient_tolerance = 1e-10;
    o->jacobi_scaling = 1;
    o->max_consecutive_invalid_steps = 5;
}

static int flag_outliers(const ov2_ba_problem *P, const double *poses, const double *lms, uint8_t *active,
                         const ov2_ba_options *o, ov2_ba_result *R, int pass, int *n_left, int *n_right)
{
    int nbad = 0;
    *n_left = *n_right = 0;
    for (int i = 0; i < P->n_res; ++i) {
        if (!active[i]) continue;
        ov2o_res_eval ev;
        ov2o_ba_eval_residual(P, poses, lms, i, 0, &ev);
        if (R->chi2) R->chi2[i] = ev.chi2;
        if (R->depth_positive) R->depth_positive[i] = (uint8_t)ev.depth_positive;
        if (ev.chi2 > o->chi2_th || !ev.depth_positive) {
            active[i] = 0;
            if (R->outlier) R->outlier[i] = (uint8_t)pass;
            ++nbad;
        } else {
            const int t = P->res_type[i];
            if (t == OV2_BA_L_XYZ || t == OV2_BA_L_INV) ++*n_left;
            else if (t == OV2_BA_R_XYZ || t == OV2_BA_R_INV) ++*n_right;
        }
    }
    return nbad;
}

int ov2o_ba_solve(const ov2_ba_problem *P, const ov2_ba_options *o, ov2_ba_result *R)
{
    uint8_t *active = (uint8_t *)malloc((size_t)(P->n_res ? P->n_res : 1));
    memset(active, 1, (size_t)P->n_res);
    R->n_log = 0; R->l2_done = 0; R->n_outliers_pass1 = R->n_outliers_pass2 = 0;
    R->l2_initial_cost = R->l2_final_cost = 0.0; R->l2_termination = OV2_BA_TERM_SKIPPED;
    if (R->outlier) memset(R->outlier, 0, (size_t)P->n_res);
    const int use_loss = o->huber_delta > 0.0;
    const int e = P->inv_depth ? 1 : 3;
    double *ep = (double *)malloc(sizeof(double) * ((size_t)P->n_pose * 7 + 1));
    double *el = (double *)malloc(sizeof(double) * ((size_t)P->n_lm * e + 1));
    R->termination = minimize(P, P->pose, P->lm, active, use_loss, o, o->max_iters, R, &R->initial_cost, &R->final_cost, ep, el);
    R->n_log_robust = R->n_log;
    int n_left, n_right;
    R->n_outliers_pass1 = flag_outliers(P, ep, el, active, o, R, 1, &n_left, &n_right);
    if (o->l2_refine && use_loss && R->n_outliers_pass1 > 0) {
        /* loss dropped only if both the left and the right list are non-empty (src/optimizer.cpp:606-608) */
        const int keep_loss = !(n_left > 0 && n_right > 0);
        R->l2_termination = minimize(P, P->pose, P->lm, active, keep_loss, o, o->l2_max_iters, R, &R->l2_initial_cost,
                                     &R->l2_final_cost, ep, el);
        R->l2_done = 1;
        R->n_outliers_pass2 = flag_outliers(P, ep, el, active, o, R, 2, &n_left, &n_right);
    }
    free(active); free(ep); free(el);
    return 0;
}
