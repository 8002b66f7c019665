/*
 * ov2_oracle_pg.c -- CPU restatement of the pose-graph problems.  TEST INFRASTRUCTURE ONLY.
 * Follows (reference, /root/reference):
 *   LeftSE3RelativePoseError::Evaluate     src/ceres_parametrization.cpp:30-102 (+ se3left_parametrization.hpp:76-99)
 *   Sophus SE3::log / SO3::logAndTheta     Thirdparty/Sophus/sophus/se3.hpp:223-256, so3.hpp:247-290; Adj se3.hpp:103-111
 *   the trust-region loop                  Thirdparty ceres 2.0.0 trust_region_minimizer.cc (as ov2_oracle_ba.c minimize())
 * The linear solve is a DENSE Cholesky of J'J + D'D (the reference asks for SPARSE_NORMAL_CHOLESKY: same solution), on
 * purpose different from the block-tridiagonal solver of the HIP kernel it checks.
 */
#include "ov2_oracle_ba.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double R[9], t[3]; } se3m;

static void q_to_R(const double *p7, double R[9])
{
    double x = p7[3], y = p7[4], z = p7[5], w = p7[6];
    const double n = sqrt(x * x + y * y + z * z + w * w);
    x /= n; y /= n; z /= n; w /= n;
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
static se3m from7(const double *p7) { se3m T; q_to_R(p7, T.R); T.t[0] = p7[0]; T.t[1] = p7[1]; T.t[2] = p7[2]; return T; }
static se3m mul(const se3m *A, const se3m *B)
{
    se3m C;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) C.R[3 * i + j] = A->R[3 * i] * B->R[j] + A->R[3 * i + 1] * B->R[3 + j] + A->R[3 * i + 2] * B->R[6 + j];
        C.t[i] = A->t[i] + (A->R[3 * i] * B->t[0] + A->R[3 * i + 1] * B->t[1] + A->R[3 * i + 2] * B->t[2]);
    }
    return C;
}
static se3m inv(const se3m *A)
{
    se3m C;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C.R[3 * i + j] = A->R[3 * j + i];
    for (int i = 0; i < 3; ++i) C.t[i] = -(C.R[3 * i] * A->t[0] + C.R[3 * i + 1] * A->t[1] + C.R[3 * i + 2] * A->t[2]);
    return C;
}
/* rotation matrix -> unit quaternion (x, y, z, w), w >= 0 branch order of Eigen::Quaternion(Matrix3) */
static void R_to_q(const double R[9], double q[4])
{
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        double t = sqrt(tr + 1.0);
        q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double t = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (R[3 * k + j] - R[3 * j + k]) * t;
        q[j] = (R[3 * j + i] + R[3 * i + j]) * t;
        q[k] = (R[3 * k + i] + R[3 * i + k]) * t;
    }
}
/* Sophus SE3::log: tangent = [upsilon, omega] */
static void se3_log(const se3m *T, double out[6])
{
    const double eps = 1e-10;
    double q[4];
    R_to_q(T->R, q);
    const double sn = q[0] * q[0] + q[1] * q[1] + q[2] * q[2], w = q[3];
    double f, theta;
    if (sn < eps * eps) {
        const double w2 = w * w;
        f = 2.0 / w - (2.0 / 3.0) * sn / (w * w2);
        theta = 2.0 * sn / w;
    } else {
        const double n = sqrt(sn);
        if (fabs(w) < eps) f = (w > 0 ? 3.14159265358979323846 : -3.14159265358979323846) / n;
        else f = 2.0 * atan(n / w) / n;
        theta = f * n;
    }
    const double om[3] = {f * q[0], f * q[1], f * q[2]};
    const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double O2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
    double c;
    if (fabs(theta) < eps) c = 1.0 / 12.0;
    else { const double h = 0.5 * theta; c = (1.0 - theta * cos(h) / (2.0 * sin(h))) / (theta * theta); }
    for (int i = 0; i < 3; ++i) {
        double s = 0;
        for (int j = 0; j < 3; ++j) s += (((i == j) ? 1.0 : 0.0) - 0.5 * O[3 * i + j] + c * O2[3 * i + j]) * T->t[j];
        out[i] = s;
    }
    out[3] = om[0]; out[4] = om[1]; out[5] = om[2];
}
static void adj(const se3m *T, double A[36])
{
    const double *t = T->t, *R = T->R;
    const double H[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
    memset(A, 0, sizeof(double) * 36);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            A[6 * i + j] = R[3 * i + j];
            A[6 * (i + 3) + j + 3] = R[3 * i + j];
            A[6 * i + j + 3] = H[3 * i] * R[j] + H[3 * i + 1] * R[3 + j] + H[3 * i + 2] * R[6 + j];
        }
}

/* one edge: residual (6) and the two 6x6 local jacobians (row-major) */
void ov2o_pg_eval_edge(const double *pose_i, const double *pose_j, const double *T_ij, double r[6], double Ji[36], double Jj[36])
{
    const se3m Twc0 = from7(pose_i), Twc1 = from7(pose_j), T01 = from7(T_ij);
    const se3m Tc1w = inv(&Twc1);
    const se3m Tc1c0 = mul(&Tc1w, &Twc0);
    const se3m err = mul(&Tc1c0, &T01);
    se3_log(&err, r);
    if (!Ji && !Jj) return;
    const double *rho = r, *om = r + 3;
    const double Sr[9] = {0, -rho[2], rho[1], rho[2], 0, -rho[0], -rho[1], rho[0], 0};
    const double So[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    for (int side = 0; side < 2; ++side) {
        double *J = side ? Jj : Ji;
        if (!J) continue;
        const double sg = side ? 1.0 : -1.0;   /* J_c0 = -[So Sr; 0 So], J_c1 = +[So Sr; 0 So] */
        double M[36];
        memset(M, 0, sizeof(M));
        for (int i = 0; i < 6; ++i) M[7 * i] = 1.0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                M[6 * i + j] += 0.5 * sg * So[3 * i + j];
                M[6 * i + j + 3] += 0.5 * sg * Sr[3 * i + j];
                M[6 * (i + 3) + j + 3] += 0.5 * sg * So[3 * i + j];
            }
        double A[36];
        if (!side) adj(&Tc1w, A);
        else { const se3m P = mul(&Twc0, &T01); const se3m Pi = inv(&P); adj(&Pi, A); }
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) {
                double s = 0;
                for (int k = 0; k < 6; ++k) s += M[6 * i + k] * A[6 * k + j];
                J[6 * i + j] = side ? -s : s;
            }
    }
}

static int chol_dense(double *A, int n)
{
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0.0)) return -1;
        d = sqrt(d);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    return 0;
}

typedef struct {
    const ov2_pg_problem *P;
    int nf, m;
    int *fidx;         /* pose -> free block or -1 */
    int *pose_of_f;
    double *r, *Ji, *Jj;   /* per edge, jacobians already column-scaled */
} pg;

static double pg_evaluate(pg *g, const double *poses, int jac, const double *scale)
{
    const ov2_pg_problem *P = g->P;
    double cost = 0.0;
    for (int e = 0; e < P->n_edge; ++e) {
        const int i = P->edge_i[e], j = P->edge_j[e];
        double rc[6];
        double *r = jac ? g->r + 6 * (size_t)e : rc;   /* a cost-only evaluation (the candidate) leaves the stored residuals alone */
        ov2o_pg_eval_edge(poses + 7 * i, poses + 7 * j, P->T_ij + 7 * (size_t)e, r, jac ? g->Ji + 36 * (size_t)e : NULL,
                          jac ? g->Jj + 36 * (size_t)e : NULL);
        double s = 0;
        for (int k = 0; k < 6; ++k) s += r[k] * r[k];
        cost += 0.5 * s;
        if (jac && scale) {
            const int fi = g->fidx[i], fj = g->fidx[j];
            for (int a = 0; a < 6; ++a)
                for (int c = 0; c < 6; ++c) {
                    if (fi >= 0) g->Ji[36 * (size_t)e + 6 * a + c] *= scale[6 * fi + c];
                    if (fj >= 0) g->Jj[36 * (size_t)e + 6 * a + c] *= scale[6 * fj + c];
                }
        }
    }
    return cost;
}

/* H = J'J (dense m x m), g = J'r, column norms */
static void pg_normal(const pg *g, double *H, double *grad, double *sqn)
{
    const ov2_pg_problem *P = g->P;
    const int m = g->m;
    if (H) memset(H, 0, sizeof(double) * (size_t)m * m);
    if (grad) memset(grad, 0, sizeof(double) * m);
    memset(sqn, 0, sizeof(double) * m);
    for (int e = 0; e < P->n_edge; ++e) {
        const int f[2] = {g->fidx[P->edge_i[e]], g->fidx[P->edge_j[e]]};
        const double *J[2] = {g->Ji + 36 * (size_t)e, g->Jj + 36 * (size_t)e}, *r = g->r + 6 * (size_t)e;
        for (int a = 0; a < 2; ++a) {
            if (f[a] < 0) continue;
            for (int c = 0; c < 6; ++c) {
                double s = 0, gg = 0;
                for (int k = 0; k < 6; ++k) { s += J[a][6 * k + c] * J[a][6 * k + c]; gg += J[a][6 * k + c] * r[k]; }
                sqn[6 * f[a] + c] += s;
                if (grad) grad[6 * f[a] + c] += gg;
            }
            if (!H) continue;
            for (int b = 0; b < 2; ++b) {
                if (f[b] < 0) continue;
                for (int c = 0; c < 6; ++c)
                    for (int d = 0; d < 6; ++d) {
                        double s = 0;
                        for (int k = 0; k < 6; ++k) s += J[a][6 * k + c] * J[b][6 * k + d];
                        H[(size_t)(6 * f[a] + c) * m + 6 * f[b] + d] += s;
                    }
            }
        }
    }
}

static void pg_plus(const pg *g, const double *poses, const double *delta, double *out)
{
    memcpy(out, poses, sizeof(double) * 7 * (size_t)g->P->n_pose);
    for (int f = 0; f < g->nf; ++f) ov2o_se3_plus(poses + 7 * g->pose_of_f[f], delta + 6 * f, out + 7 * g->pose_of_f[f]);
}

static double pg_norm2(const pg *g, const double *a, const double *b)
{
    double s = 0;
    for (int f = 0; f < g->nf; ++f)
        for (int c = 0; c < 7; ++c) {
            const double v = a[7 * g->pose_of_f[f] + c] - (b ? b[7 * g->pose_of_f[f] + c] : 0.0);
            s += v * v;
        }
    return s;
}

static void pg_log(ov2_pg_result *R, double cost, double change, double radius, double rel, double model, int valid, int ok)
{
    if (!R || R->n_log >= OV2_BA_MAX_LOG) return;
    ov2_ba_iter *it = &R->log[R->n_log++];
    it->cost = cost; it->cost_change = change; it->radius = radius; it->relative_decrease = rel;
    it->model_cost_change = model; it->step_is_valid = valid; it->step_is_successful = ok;
}

int ov2o_pose_graph_solve(const ov2_pg_problem *P, const ov2_ba_options *o, ov2_pg_result *R)
{
    memset(R, 0, sizeof(*R));
    pg g;
    g.P = P;
    g.fidx = (int *)malloc(sizeof(int) * (size_t)(P->n_pose + 1));
    g.pose_of_f = (int *)malloc(sizeof(int) * (size_t)(P->n_pose + 1));
    g.nf = 0;
    for (int i = 0; i < P->n_pose; ++i) {
        g.fidx[i] = P->pose_const[i] ? -1 : g.nf;
        if (!P->pose_const[i]) g.pose_of_f[g.nf++] = i;
    }
    const int m = g.m = 6 * g.nf, np7 = 7 * P->n_pose;
    if (m == 0 || P->n_edge == 0) { R->termination = OV2_BA_TERM_SKIPPED; free(g.fidx); free(g.pose_of_f); return 0; }
    g.r = (double *)malloc(sizeof(double) * 6 * (size_t)P->n_edge);
    g.Ji = (double *)malloc(sizeof(double) * 36 * (size_t)P->n_edge);
    g.Jj = (double *)malloc(sizeof(double) * 36 * (size_t)P->n_edge);
    double *x = (double *)malloc(sizeof(double) * np7), *c = (double *)malloc(sizeof(double) * np7), *best = (double *)malloc(sizeof(double) * np7);
    double *H = (double *)malloc(sizeof(double) * (size_t)m * m), *A = (double *)malloc(sizeof(double) * (size_t)m * m);
    double *grad = (double *)malloc(sizeof(double) * m), *sqn = (double *)malloc(sizeof(double) * m), *scale = (double *)malloc(sizeof(double) * m);
    double *diag = (double *)malloc(sizeof(double) * m), *lmd = (double *)malloc(sizeof(double) * m), *step = (double *)malloc(sizeof(double) * m);
    double *tmp = (double *)malloc(sizeof(double) * m);
    memcpy(x, P->pose, sizeof(double) * np7);
    memcpy(best, x, sizeof(double) * np7);
    for (int k = 0; k < m; ++k) scale[k] = 1.0;
    double x_cost = pg_evaluate(&g, x, 1, NULL);
    pg_normal(&g, NULL, grad, sqn);
    if (o->jacobi_scaling) {
        for (int k = 0; k < m; ++k) scale[k] = 1.0 / (1.0 + sqrt(sqn[k]));
        x_cost = pg_evaluate(&g, x, 1, scale);
        pg_normal(&g, NULL, grad, sqn);
    }
    double gmax = 0.0;
    {
        for (int k = 0; k < m; ++k) tmp[k] = -grad[k] / scale[k];   /* the gradient of the unscaled problem */
        pg_plus(&g, x, tmp, c);
        for (int f = 0; f < g.nf; ++f)
            for (int k = 0; k < 7; ++k) gmax = fmax(gmax, fabs(x[7 * g.pose_of_f[f] + k] - c[7 * g.pose_of_f[f] + k]));
    }
    R->initial_cost = x_cost;
    double minimum_cost = x_cost, x_norm = -1.0, radius = o->initial_radius, decrease_factor = 2.0;
    int reuse_diagonal = 0, invalid_steps = 0, iteration = 0, last_ok = 1, term = OV2_BA_TERM_MAX_ITER;
    pg_log(R, x_cost, 0.0, radius, 0.0, 0.0, 1, 1);
    for (;;) {
        if (iteration >= o->max_iters) { term = OV2_BA_TERM_MAX_ITER; break; }
        if (last_ok && gmax <= o->gradient_tolerance) { term = OV2_BA_TERM_GTOL; break; }
        if (radius <= o->min_radius) { term = OV2_BA_TERM_MIN_RADIUS; break; }
        ++iteration;
        pg_normal(&g, H, grad, sqn);
        ov2o_lm_diagonal(m, reuse_diagonal ? NULL : sqn, o->min_lm_diagonal, o->max_lm_diagonal, radius, diag, lmd);
        reuse_diagonal = 1;
        memcpy(A, H, sizeof(double) * (size_t)m * m);
        for (int k = 0; k < m; ++k) A[(size_t)k * m + k] += lmd[k] * lmd[k];
        int finite = chol_dense(A, m) == 0;
        double model_change = 0.0;
        int valid = 0;
        if (finite) {
            /* (J'J + D'D) y = J'r ; step = -y */
            for (int i = 0; i < m; ++i) { double s = grad[i]; for (int k = 0; k < i; ++k) s -= A[(size_t)i * m + k] * step[k]; step[i] = s / A[(size_t)i * m + i]; }
            for (int i = m - 1; i >= 0; --i) { double s = step[i]; for (int k = i + 1; k < m; ++k) s -= A[(size_t)k * m + i] * step[k]; step[i] = s / A[(size_t)i * m + i]; }
            for (int k = 0; k < m; ++k) { step[k] = -step[k]; if (!isfinite(step[k])) finite = 0; }
        }
        if (finite) {
            for (int e = 0; e < P->n_edge; ++e) {
                const int f[2] = {g.fidx[P->edge_i[e]], g.fidx[P->edge_j[e]]};
                const double *J[2] = {g.Ji + 36 * (size_t)e, g.Jj + 36 * (size_t)e}, *r = g.r + 6 * (size_t)e;
                for (int k = 0; k < 6; ++k) {
                    double mk = 0;
                    for (int a = 0; a < 2; ++a)
                        if (f[a] >= 0) for (int cc = 0; cc < 6; ++cc) mk += J[a][6 * k + cc] * step[6 * f[a] + cc];
                    model_change -= mk * (r[k] + mk / 2.0);
                }
            }
            valid = model_change > 0.0;
        }
        if (!valid) {
            if (++invalid_steps >= o->max_consecutive_invalid_steps) { term = OV2_BA_TERM_FAILURE; break; }
            ov2o_lm_step_rejected(&radius, &decrease_factor);
            last_ok = 0;
            pg_log(R, x_cost, 0.0, radius, 0.0, model_change, 0, 0);
            continue;
        }
        invalid_steps = 0;
        for (int k = 0; k < m; ++k) tmp[k] = step[k] * scale[k];
        pg_plus(&g, x, tmp, c);
        const double cand_cost = pg_evaluate(&g, c, 0, NULL);
        const double step_norm = sqrt(pg_norm2(&g, x, c));
        if (step_norm <= o->parameter_tolerance * (x_norm + o->parameter_tolerance)) { term = OV2_BA_TERM_PTOL; break; }
        const double cost_change = x_cost - cand_cost;
        if (fabs(cost_change) <= o->function_tolerance * x_cost) {
            term = OV2_BA_TERM_FTOL;
            pg_log(R, x_cost, cost_change, radius, 0.0, model_change, 1, 0);
            break;
        }
        const double rel = (cand_cost >= DBL_MAX) ? -DBL_MAX : (x_cost - cand_cost) / model_change;
        if (rel > o->min_relative_decrease) {
            memcpy(x, c, sizeof(double) * np7);
            x_norm = sqrt(pg_norm2(&g, x, NULL));
            x_cost = pg_evaluate(&g, x, 1, o->jacobi_scaling ? scale : NULL);
            pg_normal(&g, NULL, grad, sqn);
            gmax = 0.0;
            for (int k = 0; k < m; ++k) tmp[k] = -grad[k] / scale[k];
            pg_plus(&g, x, tmp, c);
            for (int f = 0; f < g.nf; ++f)
                for (int k = 0; k < 7; ++k) gmax = fmax(gmax, fabs(x[7 * g.pose_of_f[f] + k] - c[7 * g.pose_of_f[f] + k]));
            ov2o_lm_step_accepted(&radius, &decrease_factor, rel, o->max_radius);
            reuse_diagonal = 0;
            last_ok = 1;
            if (x_cost < minimum_cost) { minimum_cost = x_cost; memcpy(best, x, sizeof(double) * np7); }
            pg_log(R, x_cost, cost_change, radius, rel, model_change, 1, 1);
        } else {
            ov2o_lm_step_rejected(&radius, &decrease_factor);
            last_ok = 0;
            pg_log(R, cand_cost, cost_change, radius, rel, model_change, 1, 0);
        }
    }
    /* re-evaluate the stored jacobian state is not needed; parameters_ <- the minimum-cost point */
    memcpy(P->pose, best, sizeof(double) * np7);
    R->final_cost = minimum_cost;
    R->termination = term;
    free(g.fidx); free(g.pose_of_f); free(g.r); free(g.Ji); free(g.Jj); free(x); free(c); free(best); free(H); free(A);
    free(grad); free(sqn); free(scale); free(diag); free(lmd); free(step); free(tmp);
    return 0;
}
