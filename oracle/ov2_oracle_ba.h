/* ov2_oracle_ba.h -- BA half of the CPU oracle (TEST INFRASTRUCTURE ONLY, see ov2_oracle.h).
 * Uses the POD problem/option/result structs of the public C ABI (types only; no product code is linked). */
#ifndef OV2_ORACLE_BA_H
#define OV2_ORACLE_BA_H

#include "../include/ov2slam_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ov2o_res_eval {
    double r[2];
    double Jk[12];   /* 2x6 local jacobian of the observing pose  [-J_R | J_R hat(X_w)] */
    double Ja[12];   /* 2x6 local jacobian of the anchor pose (inverse-depth types) */
    double Jl[6];    /* 2x3 (XYZ) or 2x1 (inverse depth) */
    double chi2;
    int depth_positive;
} ov2o_res_eval;

void ov2o_se3_exp(const double tangent[6], double out7[7]);
void ov2o_se3_plus(const double x7[7], const double delta[6], double out7[7]);
void ov2o_ba_eval_residual(const ov2_ba_problem *P, const double *poses, const double *lms, int i, int want_jac,
                           ov2o_res_eval *out);
void ov2o_huber(double a, double s, double rho[3]);
void ov2o_corrector(double sq_norm, const double rho[3], int nr, double *res, int nblk, double **jac, const int *ncols);

/* generic block-sparse least squares  min |A x - b|^2 + |D x|^2  with the Schur structure Ceres exploits:
 * row blocks of R rows, every row has at most one e cell (block size E, eliminated) and up to maxf f cells (size F). */
typedef struct ov2o_bs_problem {
    int R, E, F, maxf;
    int n_rows, n_e, n_f;
    const int *row_e;   /* n_rows: e block id or -1 */
    const int *row_f;   /* n_rows x maxf: f block id or -1 */
    const double *Je;   /* n_rows x R x E */
    const double *Jf;   /* n_rows x maxf x R x F */
    const double *b;    /* n_rows x R */
    const double *D;    /* n_e*E + n_f*F or NULL */
} ov2o_bs_problem;
/* returns 0, -1 (reduced system not positive definite), -2 (E'E singular). S_out (n_f*F)^2, rhs_out optional. */
int ov2o_schur_solve(const ov2o_bs_problem *p, double *S_out, double *rhs_out, double *x);

/* MultiViewGeometry::ceresPnP (src/multi_view_geometry.cpp:492-586): motion-only LM on one pose. */
int ov2o_pnp_solve(int n, const double *unpx, const double *wpts, const int *scales, const double K[4], double *Twc,
                   int max_iters, float chi2th, int use_robust, int l2_after_robust, uint8_t *outlier, int *iters);

void ov2o_ba_default_options(ov2_ba_options *o, float robust_mono_th);
int ov2o_ba_solve(const ov2_ba_problem *P, const ov2_ba_options *o, ov2_ba_result *R);

/* pose graphs (ov2_oracle_pg.c): LeftSE3RelativePoseError on one edge (6 residuals, two 6x6 local jacobians, row-major;
 * either jacobian may be NULL) and the LM solve of a whole graph (dense normal equations) */
void ov2o_pg_eval_edge(const double *pose_i, const double *pose_j, const double *T_ij, double r[6], double Ji[36], double Jj[36]);
int ov2o_pose_graph_solve(const ov2_pg_problem *P, const ov2_ba_options *o, ov2_pg_result *R);

/* LevenbergMarquardtStrategy state updates used by the oracle's minimize() (levenberg_marquardt_strategy.cc:76-164) */
void ov2o_lm_step_accepted(double *radius, double *decrease_factor, double step_quality, double max_radius);
void ov2o_lm_step_rejected(double *radius, double *decrease_factor);
void ov2o_lm_diagonal(int n, const double *colnorm2, double min_diag, double max_diag, double radius, double *diag, double *D);

#ifdef __cplusplus
}
#endif
#endif
