/*
 * ov2_oracle_pnp.c -- CPU restatement of MultiViewGeometry::ceresPnP (TEST INFRASTRUCTURE ONLY, see ov2_oracle.h).
 *
 * Reference (/root/reference): src/multi_view_geometry.cpp:492-586 (problem, options, chi2 flags, L2 re-solve),
 * cost functor ReprojectionErrorSE3 src/ceres_parametrization.cpp:301-358, SE3LeftParameterization, and the same
 * Ceres 2.0.0 TrustRegionMinimizer / LevenbergMarquardtStrategy semantics as ov2_oracle_ba.c.  The reference asks for
 * DENSE_QR on [J; D]; the regularised normal equations (J'J + D'D) y = J'r solved here by Cholesky are the same
 * least-squares problem (6 unknowns, condition number small), differing at the 1e-15 level.
 * Deviations kept on purpose: no 5 ms wall-clock cap; outlier flags are read at the last EVALUATED pose, as the reference's cached functor fields are.
 * Parity: unpinned by any reference fixture; pinned by the shared BA pieces (see ov2_oracle_ba.c) and the tests.
 */
#include "ov2_oracle_ba.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static void quat_R(const double *p, double R[9])
{
    double x = p[3], y = p[4], z = p[5], w = p[6];
    const double n = sqrt(x * x + y * y + z * z + w * w);
    x /= n; y /= n; z /= n; w /= n;
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

/* ReprojectionErrorSE3: r (2), local jacobian 2x6 = sqrt_info * [-J_R | J_R hat(wpt)] */
static void pnp_eval(const double *Twc, const double K[4], const double *wpt, const double *unpx, double inv_sigma,
                     int want_jac, double r[2], double J[12], double *chi2, int *depth_pos)
{
    double R[9];
    quat_R(Twc, R);
    const double d[3] = {wpt[0] - Twc[0], wpt[1] - Twc[1], wpt[2] - Twc[2]};
    const double cam[3] = {R[0] * d[0] + R[3] * d[1] + R[6] * d[2], R[1] * d[0] + R[4] * d[1] + R[7] * d[2],
                           R[2] * d[0] + R[5] * d[1] + R[8] * d[2]};
    const double invz = 1.0 / cam[2];
    r[0] = inv_sigma * ((K[0] * cam[0] * invz + K[2]) - unpx[0]);
    r[1] = inv_sigma * ((K[1] * cam[1] * invz + K[3]) - unpx[1]);
    *chi2 = r[0] * r[0] + r[1] * r[1];
    *depth_pos = cam[2] > 0.0;
    if (!want_jac) return;
    const double invz2 = invz * invz;
    const double Jc[6] = {invz * K[0], 0.0, -cam[0] * invz2 * K[0], 0.0, invz * K[1], -cam[1] * invz2 * K[1]};
    for (int q = 0; q < 2; ++q) {
        double a[3];
        for (int c = 0; c < 3; ++c) a[c] = Jc[3 * q] * R[3 * c] + Jc[3 * q + 1] * R[3 * c + 1] + Jc[3 * q + 2] * R[3 * c + 2];   /* J_cam * Rcw */
        J[6 * q + 0] = -inv_sigma * a[0]; J[6 * q + 1] = -inv_sigma * a[1]; J[6 * q + 2] = -inv_sigma * a[2];
        J[6 * q + 3] = inv_sigma * (a[1] * wpt[2] - a[2] * wpt[1]);
        J[6 * q + 4] = inv_sigma * (a[2] * wpt[0] - a[0] * wpt[2]);
        J[6 * q + 5] = inv_sigma * (a[0] * wpt[1] - a[1] * wpt[0]);
    }
}

static int chol6_solve(const double *A, const double *b, double *x)
{
    double L[36];
    memcpy(L, A, sizeof(L));
    for (int j = 0; j < 6; ++j) {
        double d = L[j * 6 + j];
        for (int k = 0; k < j; ++k) d -= L[j * 6 + k] * L[j * 6 + k];
        if (!(d > 0.0)) return -1;
        d = sqrt(d);
        L[j * 6 + j] = d;
        for (int i = j + 1; i < 6; ++i) {
            double s = L[i * 6 + j];
            for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * L[j * 6 + k];
            L[i * 6 + j] = s / d;
        }
    }
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * x[k];
        x[i] = s / L[i * 6 + i];
    }
    for (int i = 5; i >= 0; --i) {
        double s = x[i];
        for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * x[k];
        x[i] = s / L[i * 6 + i];
    }
    return 0;
}

typedef struct { int n; const double *unpx, *wpts; const int *scales; const double *K; const uint8_t *active; int use_loss; double a; } pnp_pb;

/* cost (+ unscaled H = J'J, g = J'r of the robustified problem) at pose x */
static double pnp_accumulate(const pnp_pb *P, const double *x, double *H, double *g)
{
    double cost = 0.0;
    if (H) { memset(H, 0, sizeof(double) * 36); memset(g, 0, sizeof(double) * 6); }
    for (int i = 0; i < P->n; ++i) {
        if (!P->active[i]) continue;
        double r[2], J[12], chi2;
        int dp;
        const double inv_sigma = 1.0 / (P->scales ? pow(2., P->scales[i]) : 1.0);
        pnp_eval(x, P->K, P->wpts + 3 * i, P->unpx + 2 * i, inv_sigma, H != NULL, r, J, &chi2, &dp);
        double rho[3] = {chi2, 1.0, 0.0};
        if (P->use_loss) ov2o_huber(P->a, chi2, rho);
        cost += 0.5 * rho[0];
        if (!H) continue;
        if (P->use_loss) {
            double *jac[1] = {J};
            const int nc[1] = {6};
            ov2o_corrector(chi2, rho, 2, r, 1, jac, nc);
        }
        for (int a = 0; a < 6; ++a) {
            g[a] += J[a] * r[0] + J[6 + a] * r[1];
            for (int b = 0; b < 6; ++b) H[a * 6 + b] += J[a] * J[b] + J[6 + a] * J[6 + b];
        }
    }
    return cost;
}

/* Teval receives the pose of the LAST residual evaluation: the reference reads the cost functors' cached chi2err_ /
 * isdepthpositive_ after Solve (src/multi_view_geometry.cpp:559-571), i.e. x after an accepted last step, the candidate
 * after a rejected one or a FTOL / PTOL exit (Ceres does not re-evaluate after Solve) */
static int pnp_minimize(const pnp_pb *P, double *Twc, const ov2_ba_options *o, int max_iters, int *n_iter, double *Teval)
{
    double x[7], cand[7], H[36], g[6], scale[6], diag[6];
    memcpy(x, Twc, sizeof(x));
    memcpy(Teval, x, sizeof(x));
    double x_cost = pnp_accumulate(P, x, H, g);
    for (int c = 0; c < 6; ++c) scale[c] = o->jacobi_scaling ? 1.0 / (1.0 + sqrt(H[c * 6 + c])) : 1.0;
    double minimum_cost = x_cost, x_norm = -1.0, radius = o->initial_radius, dec = 2.0;
    int reuse = 0, invalid = 0, iteration = 0, term = OV2_BA_TERM_MAX_ITER;
    *n_iter = 0;
    for (;;) {
        if (iteration >= max_iters) { term = OV2_BA_TERM_MAX_ITER; break; }
        if (radius <= o->min_radius) { term = OV2_BA_TERM_MIN_RADIUS; break; }
        ++iteration;
        *n_iter = iteration;
        double Hs[36], gs[6], A[36], y[6], step[6];
        for (int a = 0; a < 6; ++a) {
            gs[a] = g[a] * scale[a];
            for (int b = 0; b < 6; ++b) Hs[a * 6 + b] = H[a * 6 + b] * scale[a] * scale[b];
        }
        if (!reuse) for (int c = 0; c < 6; ++c) diag[c] = fmin(fmax(Hs[c * 6 + c], o->min_lm_diagonal), o->max_lm_diagonal);
        reuse = 1;
        memcpy(A, Hs, sizeof(A));
        for (int c = 0; c < 6; ++c) A[c * 6 + c] += diag[c] / radius;
        int ok = chol6_solve(A, gs, y) == 0;
        double model_change = 0.0;
        if (ok) {
            for (int c = 0; c < 6; ++c) { step[c] = -y[c]; if (!isfinite(step[c])) ok = 0; }
            double sg = 0, shs = 0;
            for (int a = 0; a < 6; ++a) {
                sg += step[a] * gs[a];
                for (int b = 0; b < 6; ++b) shs += step[a] * Hs[a * 6 + b] * step[b];
            }
            model_change = -(sg + 0.5 * shs);
        }
        if (!ok || !(model_change > 0.0)) {
            if (++invalid >= o->max_consecutive_invalid_steps) { term = OV2_BA_TERM_FAILURE; break; }
            radius /= dec; dec *= 2.0;
            continue;
        }
        invalid = 0;
        double delta[6];
        for (int c = 0; c < 6; ++c) delta[c] = step[c] * scale[c];
        ov2o_se3_plus(x, delta, cand);
        const double cand_cost = pnp_accumulate(P, cand, NULL, NULL);
        memcpy(Teval, cand, sizeof(cand));
        double sn = 0;
        for (int c = 0; c < 7; ++c) sn += (x[c] - cand[c]) * (x[c] - cand[c]);
        if (sqrt(sn) <= o->parameter_tolerance * (x_norm + o->parameter_tolerance)) { term = OV2_BA_TERM_PTOL; break; }
        const double cost_change = x_cost - cand_cost;
        if (fabs(cost_change) <= o->function_tolerance * x_cost) { term = OV2_BA_TERM_FTOL; break; }
        const double rel = cost_change / model_change;
        if (rel > o->min_relative_decrease) {
            memcpy(x, cand, sizeof(x));
            x_norm = 0;
            for (int c = 0; c < 7; ++c) x_norm += x[c] * x[c];
            x_norm = sqrt(x_norm);
            x_cost = pnp_accumulate(P, x, H, g);
            radius = fmin(o->max_radius, radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3)));
            dec = 2.0; reuse = 0;
            if (x_cost < minimum_cost) { minimum_cost = x_cost; memcpy(Twc, x, sizeof(x)); }
        } else {
            radius /= dec; dec *= 2.0;
        }
    }
    return term;
}

/* returns 1 = success (summary.IsSolutionUsable()), 0 = failure or every point flagged (:567-569) */
int ov2o_pnp_solve(int n, const double *unpx, const double *wpts, const int *scales, const double K[4], double *Twc,
                   int max_iters, float chi2th, int use_robust, int l2_after_robust, uint8_t *outlier, int *iters)
{
    ov2_ba_options o;
    ov2o_ba_default_options(&o, chi2th);   /* a = sqrtf(chi2th), chi2_th, Ceres LM defaults */
    uint8_t *active = (uint8_t *)malloc((size_t)(n ? n : 1));
    memset(active, 1, (size_t)n);
    memset(outlier, 0, (size_t)n);
    pnp_pb P = {n, unpx, wpts, scales, K, active, use_robust, o.huber_delta};
    double T[7];
    memcpy(T, Twc, sizeof(T));
    int it1 = 0, it2 = 0;
    double Te[7];
    int term = pnp_minimize(&P, T, &o, max_iters, &it1, Te);
    int nbad = 0;
    for (int i = 0; i < n; ++i) {
        double r[2], J[12], chi2;
        int dp;
        const double inv_sigma = 1.0 / (scales ? pow(2., scales[i]) : 1.0);
        pnp_eval(Te, K, wpts + 3 * i, unpx + 2 * i, inv_sigma, 0, r, J, &chi2, &dp);
        if (chi2 > (double)chi2th || !dp) {
            outlier[i] = 1; ++nbad;
            if (l2_after_robust) active[i] = 0;
        }
    }
    if (iters) { iters[0] = it1; iters[1] = 0; }
    if (nbad == n) { free(active); return 0; }
    if (l2_after_robust && nbad > 0) {
        P.use_loss = 0;
        term = pnp_minimize(&P, T, &o, max_iters, &it2, Te);
        if (iters) iters[1] = it2;
    }
    memcpy(Twc, T, sizeof(T));
    free(active);
    return term != OV2_BA_TERM_FAILURE;
}
