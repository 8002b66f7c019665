// ba.hip -- local bundle adjustment on gfx950: Levenberg-Marquardt + landmark Schur complement, all f64.
//
// Replaces (reference, /root/reference): the ceres::Solve + chi2 flagging + L2 re-solve inside
// Optimizer::localBA src/optimizer.cpp:439-735, with the cost functors of src/ceres_parametrization.cpp:107-709
// and the Ceres 2.0.0 trust-region/LM/Schur semantics restated in oracle/ov2_oracle_ba.c.
// BATCHED: one call solves B independent windows (ov2_ba_solve_batch; ov2_ba_solve is B = 1).  The windows are laid end
// to end -- residual blocks, landmarks and poses get global indices, rows stay sorted by (landmark, pose) so every
// window owns a contiguous range of rows / landmark blocks / pose blocks -- and every O(residuals) / O(landmarks) kernel
// is simply a larger grid.  What is per window: the reduced camera system S_w (its own dense block of the S pool, its
// own Cholesky workgroup), the scalar reductions (cost, model change, norms) and the WHOLE trust-region state machine
// of Ceres -- radius update, accept / reject, tolerances -- which lives in a device-side record per window (ba_win)
// and is advanced by single-thread epilogues of the per-window reduction kernels.  The host enqueues the fixed launch
// chain of max_iters LM rounds without a single synchronisation; a window that has terminated is skipped by every
// kernel.  A window's arithmetic does not depend on which other windows share the batch: its reductions run over its
// own ranges in a fixed order, so the result of window w is bitwise the same for B = 1 and B = 64.
//
// HBM layout (one workspace per batch, reused across calls through the ctx):
//   rows (= residual blocks) sorted by (landmark, observing pose), SoA: type / pose / landmark / F-block ids /
//   measurement; per row storage of the robustified, UNSCALED jacobian: res[2], Je[2e], U[2x6] (ONE pose block: the
//   anchor-pose block of an anchored row is -U; the consumers apply the Jacobi scale of the column as they read)
//   landmark CSR row_ptr[n_e+1]; S (reduced camera system) dense column-major m x m (m = 6 * free poses), lower
//   triangle authoritative; every reduction (cost, model change, norms, S, rhs, column norms) has one writer per element
//   and a fixed summation order => bitwise reproducible, independent of the batch a window shares.
//
// Kernels:  ba_eval (residual + analytic jacobian + Huber corrector)  -> rows
//           ba_colnorm16 / ba_pose_normal (column norms, gradient, F'F per pose: whatever only changes with the jacobian)
//           bs_landmark + bs_gather (per landmark: (E'E + D)^-1 and the Schur cells; per pose / pose pair: the blocks of S)
//           ba_chol (one workgroup per window, left-looking blocked Cholesky in LDS, rhs carried as an extra row; back solve)
//           ba_backsub16 (per landmark: y_e, then J*step and the model cost change)
//           ba_plus (SE3 left update / additive), ba_flag (chi2 + depth flags), ba_winreduce (ordered sums + the
//           per-window trust-region state machine); bb_* / bs_* structure kernels build the program on the device;
//           bb_gather / bb_scatter serve device-resident callers (ov2_ba_solve_batch_dev)
// MFMA only in the multi-workgroup Cholesky's SYRK (v_mfma_f64_16x16x4_f64): the other contractions are 6x6 / 6x1 / 6x3
// blocks and every kernel of an LM round waits on memory (SQ counters: 52-82 % of the wave cycles, scripts/profile_ba_sq.sh).
#include "ov2_internal.h"

#include <hipcub/hipcub.hpp>

#include <chrono>
#include <thread>
#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <numeric>

namespace {

enum { K_EVAL = OV2_K_BA_FIRST, K_COLNORM, K_SCALE, K_LMDIAG, K_SINIT, K_SCHUR, K_CHOL, K_BACKSUB, K_PLUS, K_FLAG,
       K_REDUCE, K_MISC };

#define WPT_S 4
struct ba_wconst { double Kl[4], Kr[4], Rrl[9], trl[3]; };   // per window: calibrations + right<-left extrinsic

// Device-side record of one window: its ranges in the batch-wide arrays and the state of Ceres' TrustRegionMinimizer
// (trust_region_minimizer.cc) + LevenbergMarquardtStrategy (levenberg_marquardt_strategy.cc) for it.
struct ba_win {
    int row0, row1;      // sorted rows
    int e0, e1;          // landmark blocks of the reduced program
    int f0, f1;          // pose blocks of the reduced program
    int vb0, vb1;        // virtual 256-row blocks (cost partials)
    int m;               // 6 * (f1 - f0)
    int skip;            // not part of this program (second solve of a window that needs no L2 refinement)
    long long S_off;     // doubles into the S pool
    // trust-region state
    double radius, decrease_factor, x_cost, x_norm, gmax, minimum_cost, initial_cost;
    double model_change, cand_cost;
    int iteration, invalid_steps, reuse_diagonal, refresh_diag, last_ok;
    int active;          // takes part in the current round
    int valid;           // the round's step is valid: Plus + candidate evaluation run
    int accepted;        // the round's step was accepted: copy, jacobian at the new x
    int done, termination, use_loss, max_iters;
    int eval_at_cand;    // the last residual evaluation of this window happened at the candidate (chi2 flags are read there)
    int chol_fail;
    int nbad, n_left, n_right;   // ba_flag tallies of the current pass
    int n_log;
    ov2_ba_iter log[OV2_BA_MAX_LOG];
};

struct ba_dev {
    int e;               // landmark block size 1 | 3 (one parametrisation per batch)
    int B;               // windows
    int n_rows, n_e, n_f, n_pose, n_lm, m, nc;   // batch totals (m = 6 * n_f)
    ba_win *W;
    const ba_wconst *wc;
    const int *row_win, *win_of_e, *win_of_f;     // window of a sorted row / landmark block / pose block
    const int *vb_start;                          // B + 1: first virtual block of every window
    int *n_active;                                // per LM round: windows that enter it (the host stops enqueuing at zero)
    double *Spool;
    // rows (sorted)
    const unsigned char *type;
    const int *pose, *lm, *anch;   // global pose / landmark / anchor-pose indices (anch = -1 for XYZ)
    const int *eb, *fk, *fa;       // reduced-program block ids (fk/fa = -1: constant or absent)
    const double *uv, *inv_sigma;  // measurement, 1/sigma per row
    const double *lm_auv;          // anchor pixel per LANDMARK (batch-wide index): the rows of a landmark sit in adjacent lanes, one line serves them
    const int *row_ptr;            // n_e + 1
    const int *lm_of_e, *pose_of_f;
    // jacobian storage
    // per row: residual, landmark block and the 2 x 3 block G of the robustified, UNSCALED jacobian; the 2 x 6 pose block of
    // a row is U = [-G | G x p] (p = the landmark's world point, kept once per landmark block in wpt): 48 bytes per row
    // instead of 96.  The anchor-pose block of an anchored inverse-depth row is -U, zero for every other type.
    double *res, *Je, *G, *wpt;    // wpt: WPT_S doubles per landmark block (x, y, z, 0): two 16-byte loads / stores
    const int4 *rowrec;            // per sorted row: (landmark, anchor pose, pose, landmark block | type << 28): what the evaluation reads, in one load
    const int *ent_lm;             // per pose -> rows entry: landmark block of its row (what ba_pose_normal needs beside the row)
    // vectors over columns (E part first: n_e*e, then F part: n_f*6)
    double *scale, *sqn, *grad, *diag, *lmd, *step;
    double *rhs, *iete, *ieg;
    double *FFp;         // per pose block: lower triangle (21) of F'F over all its rows, refreshed with the jacobian
    double *part;        // partial sums (max(n_rows blocks, n_e, n_f + n_e))
    // L2 re-solve on the SAME program: rows flagged by the robust pass are dead (zero residual and jacobian), blocks whose
    // rows all died stay out of the parameter norms (Ceres' reduced program would not contain them)
    int masked;
    unsigned char *dead;       // per sorted row
    unsigned char *e_dead;     // per landmark block: no live row left
    unsigned char *f_live;     // per pose block: some live row refers to it
};

struct ba_lmopt {   // the solver options the device-side state machine needs
    double min_d, max_d, max_radius, min_radius, min_rel, ptol, gtol, ftol;
    int max_invalid, jacobi;
};

// ------------------------------------------------------------------------------------------------------
// SE3 helpers (same formulas as the oracle / Sophus / Eigen)

__host__ __device__ inline void quat_to_R(const double q[4], double R[9])
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

__host__ __device__ inline void pose_Rt(const double *p, double R[9], double t[3])
{
    double q[4] = {p[3], p[4], p[5], p[6]};
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
    quat_to_R(q, R);
    t[0] = p[0]; t[1] = p[1]; t[2] = p[2];
}

__device__ inline void se3_plus(const double *x, const double *d, double *out)
{
    // Sophus::SE3::exp(d) * SE3(q,t)   (se3left_parametrization.hpp:41-60)
    const double *u = d, *w = d + 3;
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
    const double a[4] = {imag * w[0], imag * w[1], imag * w[2], real};
    double Ra[9], V[9];
    quat_to_R(a, Ra);
    if (theta < eps) {
        for (int i = 0; i < 9; ++i) V[i] = Ra[i];
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
    double b[4] = {x[3], x[4], x[5], x[6]};
    const double nb = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3]);
    b[0] /= nb; b[1] /= nb; b[2] /= nb; b[3] /= nb;
    double q[4];
    q[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    q[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    q[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    q[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    const double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int r = 0; r < 3; ++r)
        out[r] = (V[3 * r] * u[0] + V[3 * r + 1] * u[1] + V[3 * r + 2] * u[2]) +
                 (Ra[3 * r] * x[0] + Ra[3 * r + 1] * x[1] + Ra[3 * r + 2] * x[2]);
    out[3] = q[0] / nq; out[4] = q[1] / nq; out[5] = q[2] / nq; out[6] = q[3] / nq;
}

// ------------------------------------------------------------------------------------------------------
// one residual block: r, local jacobians, chi2, depth sign  (src/ceres_parametrization.cpp, 5 functors)

struct row_eval {
    double r[2], G[6], wp[3], Jl[6], chi2;   // pose block = [-G | G x wp]; the anchor-pose block of an anchored row is its negative
    bool depth_pos;
    int type, eb;   // from the row record
};

// R | t of a pose: from the workgroup's LDS table of its window's poses (EV_RT_STRIDE doubles per pose, filled by
// pose_Rt) when there is one, else converted from the quaternion pose in global memory -- the same values either way
#define EV_RT_STRIDE 13
#define EV_RT_MAX 128
__device__ __forceinline__ void fetch_Rt(const double *__restrict__ poses, const double *sRt, int p0, int gp, double R[9], double t[3])
{
    if (sRt) {
        const double *q = sRt + (gp - p0) * EV_RT_STRIDE;
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = q[i];
        t[0] = q[9]; t[1] = q[10]; t[2] = q[11];
    } else pose_Rt(poses + 7 * (size_t)gp, R, t);
}

template <bool JAC>
__device__ inline void eval_row(const ba_dev &d, const ba_wconst &wc, const double *__restrict__ poses,
                                const double *__restrict__ lms, int row, row_eval &o, const double *sRt = nullptr, int p0 = 0)
{
    const int4 rec = d.rowrec[row];
    const int type = (int)((unsigned)rec.w >> 28), l = rec.x;
    o.type = type; o.eb = rec.w & 0x0fffffff;
    const double inv_sigma = d.inv_sigma[row];
    const bool is_right = (type == OV2_BA_R_XYZ || type == OV2_BA_R_INV || type == OV2_BA_RANCH_INV);
    const bool inv = (type >= OV2_BA_L_INV);
    const double *K = is_right ? wc.Kr : wc.Kl;
    double wpt[3] = {0, 0, 0}, anchpt[3] = {0, 0, 0}, Rwa[9], zanch = 0.0;
    if (inv) {
        zanch = 1.0 / lms[l];
        anchpt[0] = zanch * ((d.lm_auv[2 * (size_t)l] - wc.Kl[2]) / wc.Kl[0]);
        anchpt[1] = zanch * ((d.lm_auv[2 * (size_t)l + 1] - wc.Kl[3]) / wc.Kl[1]);
        anchpt[2] = zanch;
        if (type != OV2_BA_RANCH_INV) {
            double twa[3];
            fetch_Rt(poses, sRt, p0, rec.y, Rwa, twa);
            for (int r = 0; r < 3; ++r)
                wpt[r] = (Rwa[3 * r] * anchpt[0] + Rwa[3 * r + 1] * anchpt[1] + Rwa[3 * r + 2] * anchpt[2]) + twa[r];
        }
    } else {
        wpt[0] = lms[3 * l]; wpt[1] = lms[3 * l + 1]; wpt[2] = lms[3 * l + 2];
    }
    double cam[3], M[9];
    if (type == OV2_BA_RANCH_INV) {
        for (int r = 0; r < 3; ++r)
            cam[r] = (wc.Rrl[3 * r] * anchpt[0] + wc.Rrl[3 * r + 1] * anchpt[1] + wc.Rrl[3 * r + 2] * anchpt[2]) + wc.trl[r];
        for (int i = 0; i < 9; ++i) M[i] = wc.Rrl[i];
    } else {
        double Rwc[9], twc[3], lcam[3];
        fetch_Rt(poses, sRt, p0, rec.z, Rwc, twc);
        const double dd[3] = {wpt[0] - twc[0], wpt[1] - twc[1], wpt[2] - twc[2]};
        for (int r = 0; r < 3; ++r) lcam[r] = Rwc[r] * dd[0] + Rwc[3 + r] * dd[1] + Rwc[6 + r] * dd[2];
        if (is_right) {
            for (int r = 0; r < 3; ++r)
                cam[r] = (wc.Rrl[3 * r] * lcam[0] + wc.Rrl[3 * r + 1] * lcam[1] + wc.Rrl[3 * r + 2] * lcam[2]) + wc.trl[r];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) {
                    double s = 0;
                    for (int k = 0; k < 3; ++k) s += wc.Rrl[3 * r + k] * Rwc[3 * c + k];
                    M[3 * r + c] = s;
                }
        } else {
            cam[0] = lcam[0]; cam[1] = lcam[1]; cam[2] = lcam[2];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) M[3 * r + c] = Rwc[3 * c + r];
        }
    }
    const double invz = 1.0 / cam[2];
    o.r[0] = inv_sigma * ((K[0] * cam[0] * invz + K[2]) - d.uv[2 * row]);
    o.r[1] = inv_sigma * ((K[1] * cam[1] * invz + K[3]) - d.uv[2 * row + 1]);
    o.chi2 = o.r[0] * o.r[0] + o.r[1] * o.r[1];
    o.depth_pos = cam[2] > 0.0;
    if (!JAC) return;
    const double invz2 = invz * invz;
    const double Jc[6] = {invz * K[0], 0.0, -cam[0] * invz2 * K[0], 0.0, invz * K[1], -cam[1] * invz2 * K[1]};
    double JR[6];
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 3; ++c) JR[3 * r + c] = Jc[3 * r] * M[c] + Jc[3 * r + 1] * M[3 + c] + Jc[3 * r + 2] * M[6 + c];
    for (int i = 0; i < 6; ++i) { o.G[i] = 0.0; o.Jl[i] = 0.0; }
    for (int i = 0; i < 3; ++i) o.wp[i] = wpt[i];
    if (type != OV2_BA_RANCH_INV)   // d r / d (left se3 increment of Twc) = [-G | G x p], G = J_r / sigma (p = world point)
        for (int i = 0; i < 6; ++i) o.G[i] = inv_sigma * JR[i];
    if (inv) {
        double jl[3];
        if (type == OV2_BA_RANCH_INV) {
            jl[0] = -zanch * anchpt[0]; jl[1] = -zanch * anchpt[1]; jl[2] = -zanch * anchpt[2];
        } else {
            for (int r = 0; r < 3; ++r)
                jl[r] = -zanch * (Rwa[3 * r] * anchpt[0] + Rwa[3 * r + 1] * anchpt[1] + Rwa[3 * r + 2] * anchpt[2]);
        }
        for (int r = 0; r < 2; ++r) o.Jl[r] = inv_sigma * (JR[3 * r] * jl[0] + JR[3 * r + 1] * jl[1] + JR[3 * r + 2] * jl[2]);
    } else {
        for (int i = 0; i < 6; ++i) o.Jl[i] = inv_sigma * JR[i];
    }
}

__device__ inline void huber(double a, double s, double rho[3])
{
    const double b = a * a;
    if (s > b) {
        const double r = sqrt(s);
        rho[0] = 2.0 * a * r - b;
        rho[1] = fmax(2.2250738585072014e-308, a / r);
        rho[2] = -rho[1] / (2.0 * s);
    } else {
        rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
}

// block-wide ordered sum of one double per thread (blockDim = 256) -> lane 0 of the block returns the total
__device__ inline double block_sum_256(double v, double *sh)
{
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) sh[t] += sh[t + s];
        __syncthreads();
    }
    return sh[0];
}

// ------------------------------------------------------------------------------------------------------
// K_EVAL: residual (+ jacobian), loss, corrector, Jacobi scaling; cost partial per workgroup

// The solver's kernels are few, short waves that share their CUs with the front-end's long
// KLT waves: raising the wave priority lets them win the SIMD issue arbitration instead of taking turns
// (s_setprio is per wave and costs one scalar instruction; measured: 467 -> 475 LM it/s concurrent, front-end unchanged).
#ifndef BA_WAVE_PRIO
#define BA_WAVE_PRIO() __builtin_amdgcn_s_setprio(3)
#endif

// window that owns virtual block b: the last w with vb_start[w] <= b (windows without rows own no block)
__device__ __forceinline__ int win_of_vblock(const int *__restrict__ vb_start, int B, int b)
{
    int lo = 0, hi = B;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (vb_start[mid] <= b) lo = mid; else hi = mid;
    }
    return lo;
}

// which windows a launch works for
enum { EV_ZERO = 0,   // iteration zero: every window of the program
       EV_CAND = 1,   // candidate evaluation: windows whose round produced a valid step
       EV_ACC = 2 };  // jacobian at the new x: windows whose step was accepted

__device__ __forceinline__ bool win_runs(const ba_win &W, int mode)
{
    return mode == EV_ZERO ? !W.skip : (mode == EV_CAND ? (W.active && W.valid) : W.accepted != 0);
}

// workgroup b covers 256 rows of ONE window (virtual blocks: a window with r rows owns ceil(r / 256) of them), so the
// cost partial part[b] belongs to one window and a window's partials are summed in the order a lone solve would use
template <bool JAC, int E>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void ba_eval_kernel(ba_dev d, const double *__restrict__ poses,
                                                      const double *__restrict__ lms, int mode, double huber_a,
                                                      double *__restrict__ part, const int *__restrict__ pose_off)
{
    BA_WAVE_PRIO();
    __shared__ double sh[256];
    __shared__ double sRt_buf[EV_RT_MAX * EV_RT_STRIDE];
    __shared__ int sw;
    if (threadIdx.x == 0) sw = win_of_vblock(d.vb_start, d.B, blockIdx.x);
    __syncthreads();
    const int w = sw;
    const ba_win &W = d.W[w];
    if (!win_runs(W, mode)) return;   // workgroup-uniform
    // the window's poses as R | t, converted once per workgroup instead of twice per row (a row's pose and its landmark's
    // anchor pose: two dependent 56-byte gathers and two quaternion normalisations per lane)
    const int p0 = pose_off[w], np = pose_off[w + 1] - p0;
    const double *sRt = (np <= EV_RT_MAX) ? sRt_buf : nullptr;
    if (sRt) {
        if ((int)threadIdx.x < np) {
            double R[9], t[3];
            pose_Rt(poses + 7 * (size_t)(p0 + (int)threadIdx.x), R, t);
            double *q = sRt_buf + threadIdx.x * EV_RT_STRIDE;
#pragma unroll
            for (int i = 0; i < 9; ++i) q[i] = R[i];
            q[9] = t[0]; q[10] = t[1]; q[11] = t[2];
        }
        __syncthreads();
    }
    const int row = W.row0 + ((int)blockIdx.x - W.vb0) * 256 + (int)threadIdx.x;
    const int use_loss = W.use_loss;
    double c = 0.0;
    const bool dead = d.masked && row < W.row1 && d.dead[row];
    if (dead) {
        if (JAC) {
            constexpr int e = E;
            d.res[2 * row] = 0.0; d.res[2 * row + 1] = 0.0;
            for (int cc = 0; cc < 2 * e; ++cc) d.Je[(size_t)row * 2 * e + cc] = 0.0;
            double2 *G2 = reinterpret_cast<double2 *>(d.G + (size_t)row * 6);
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) G2[cc] = make_double2(0.0, 0.0);
        }
    } else if (row < W.row1) {
        row_eval ev;
        eval_row<JAC>(d, d.wc[w], poses, lms, row, ev, sRt, p0);
        double rho[3] = {ev.chi2, 1.0, 0.0};
        if (use_loss) huber(huber_a, ev.chi2, rho);
        c = 0.5 * rho[0];
        if (JAC) {
            // Corrector (corrector.cc:42-157); for Huber rho'' <= 0 always -> plain sqrt(rho') scaling, the general
            // branch is kept for completeness
            const double sqrt_rho1 = sqrt(rho[1]);
            double rs = sqrt_rho1, asn = 0.0;
            if (use_loss && !(ev.chi2 == 0.0 || rho[2] <= 0.0)) {
                const double D = 1.0 + 2.0 * ev.chi2 * rho[2] / rho[1];
                const double alpha = 1.0 - sqrt(D);
                rs = sqrt_rho1 / (1 - alpha);
                asn = alpha / ev.chi2;
            }
            constexpr int e = E;   // compile-time: the Jl accesses below must not become scratch-backed dynamic indexing
            if (use_loss) {
                if (asn == 0.0) {
                    for (int i = 0; i < 6; ++i) ev.G[i] *= sqrt_rho1;
                    for (int i = 0; i < 2 * e; ++i) ev.Jl[i] *= sqrt_rho1;
                } else {
                    for (int cc = 0; cc < 3; ++cc) {   // the corrector mixes the two rows of a block: the same mix of the rows of G
                        const double rtj = ev.G[cc] * ev.r[0] + ev.G[3 + cc] * ev.r[1];
                        ev.G[cc] = sqrt_rho1 * (ev.G[cc] - asn * ev.r[0] * rtj);
                        ev.G[3 + cc] = sqrt_rho1 * (ev.G[3 + cc] - asn * ev.r[1] * rtj);
                    }
                    for (int cc = 0; cc < e; ++cc) {
                        const double rtj = ev.Jl[cc] * ev.r[0] + ev.Jl[e + cc] * ev.r[1];
                        ev.Jl[cc] = sqrt_rho1 * (ev.Jl[cc] - asn * ev.r[0] * rtj);
                        ev.Jl[e + cc] = sqrt_rho1 * (ev.Jl[e + cc] - asn * ev.r[1] * rtj);
                    }
                }
                ev.r[0] *= rs; ev.r[1] *= rs;
            }
            d.res[2 * row] = ev.r[0];
            d.res[2 * row + 1] = ev.r[1];
            // Stored UNSCALED (the consumers apply the Jacobi scaling of the column as they read) and as the 2 x 3 block G: the
            // pose block is [-G | G x p] with the landmark's world point p, written once per landmark block.
            for (int cc = 0; cc < 2 * e; ++cc) d.Je[(size_t)row * 2 * e + cc] = ev.Jl[cc];
            double2 *G2 = reinterpret_cast<double2 *>(d.G + (size_t)row * 6);   // 48-byte rows: three 16-byte stores
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) G2[cc] = make_double2(ev.G[2 * cc], ev.G[2 * cc + 1]);
            if (ev.type != OV2_BA_RANCH_INV) {   // every such row of a landmark holds the same point (same anchor, same depth)
                const int eb = ev.eb;
                double2 *W2 = reinterpret_cast<double2 *>(d.wpt + (size_t)eb * WPT_S);
                W2[0] = make_double2(ev.wp[0], ev.wp[1]); W2[1] = make_double2(ev.wp[2], 0.0);
            }
        }
    }
    const double tot = block_sum_256(c, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// ordered sum of in[0..n) by the 256 threads of a workgroup (thread t takes t, t + 256, ...; then the tree)
__device__ __forceinline__ double ordered_sum(const double *__restrict__ in, int n, double *sh)
{
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) v += in[i];
    const double tot = block_sum_256(v, sh);
    __syncthreads();
    return tot;
}

// ... of the concatenation a[0..na) | b[0..nb)
__device__ __forceinline__ double ordered_sum2(const double *__restrict__ a, int na, const double *__restrict__ b, int nb,
                                               double *sh)
{
    double v = 0.0;
    for (int i = threadIdx.x; i < na + nb; i += 256) v += (i < na) ? a[i] : b[i - na];
    const double tot = block_sum_256(v, sh);
    __syncthreads();
    return tot;
}

__device__ __forceinline__ double block_max_256(double v, double *sh)
{
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

__device__ inline void win_log(ba_win &W, double cost, double change, double radius, double rel, double model, int valid, int ok)
{
    if (W.n_log >= OV2_BA_MAX_LOG) return;
    ov2_ba_iter *it = &W.log[W.n_log++];
    it->cost = cost; it->cost_change = change; it->radius = radius; it->relative_decrease = rel;
    it->model_cost_change = model; it->step_is_valid = valid; it->step_is_successful = ok;
}

// HandleInvalidStep (trust_region_minimizer.cc:562-580): StepIsInvalid == StepRejected for the LM strategy
__device__ inline void win_invalid_step(ba_win &W, const ba_lmopt &o, double model_change)
{
    W.valid = 0;
    if (++W.invalid_steps >= o.max_invalid) { W.termination = OV2_BA_TERM_FAILURE; W.done = 1; W.active = 0; return; }
    W.radius /= W.decrease_factor; W.decrease_factor *= 2.0;
    W.last_ok = 0;
    win_log(W, W.x_cost, 0.0, W.radius, 0.0, model_change, 0, 0);
}

// FinalizeIterationAndCheckIfMinimizerCanContinue + the head of the next iteration (trust_region_minimizer.cc:104-131)
__device__ inline void win_begin_round(ba_win &W, const ba_lmopt &o)
{
    W.active = 0; W.valid = 0; W.accepted = 0;
    if (W.skip || W.done) return;
    if (W.iteration >= W.max_iters) { W.termination = OV2_BA_TERM_MAX_ITER; W.done = 1; return; }
    if (W.last_ok && W.gmax <= o.gtol) { W.termination = OV2_BA_TERM_GTOL; W.done = 1; return; }
    if (W.radius <= o.min_radius) { W.termination = OV2_BA_TERM_MIN_RADIUS; W.done = 1; return; }
    ++W.iteration;
    W.active = 1;
    W.refresh_diag = W.reuse_diagonal ? 0 : 1;   // LevenbergMarquardtStrategy::ComputeStep :76-89
    W.reuse_diagonal = 1;
    W.chol_fail = 0;
}

// Per-window reductions + the single-thread epilogues that advance the window's trust-region state.  One workgroup
// per window; every sum runs over the window's own range in the order a lone solve uses.
enum { WR_JAC = 0,     // after a jacobian evaluation (iteration zero, or the new x of an accepted step): cost, gradient
                       // max-norm -> bookkeeping of the accepted step, then the head of the next round for EVERY window
       WR_MODEL = 1,   // after the back-substitution: model cost change -> step validity
       WR_CAND = 2 };  // after the candidate evaluation: |step|, |x+|, candidate cost -> tolerances, accept / reject

template <int MODE>
__global__ __launch_bounds__(256) void ba_winreduce_kernel(ba_dev d, ba_lmopt o, const double *__restrict__ xp,
                                                           const double *__restrict__ part_cost,
                                                           const double *__restrict__ part_step,
                                                           const double *__restrict__ part_norm,
                                                           const double *__restrict__ part_model, int first,
                                                           double initial_radius, int next_round)
{
    BA_WAVE_PRIO();
    __shared__ double sh[256];
    const int w = blockIdx.x;
    ba_win &W = d.W[w];
    const int e = d.e;
    if (MODE == WR_JAC) {
        const bool fin = first ? !W.skip : (W.accepted != 0);   // workgroup-uniform
        double cost = 0.0, gmax = 0.0;
        if (fin) {
            cost = ordered_sum(part_cost + W.vb0, W.vb1 - W.vb0, sh);
            // gradient_max_norm = || x - Plus(x, -g) ||_inf with g the gradient of the UNSCALED problem (grad holds the
            // scaled one): landmark blocks are additive, pose blocks go through the SE3 Plus
            double v = 0.0;
            for (int i = W.e0 * e + (int)threadIdx.x; i < W.e1 * e; i += 256)
                v = fmax(v, fabs(o.jacobi ? d.grad[i] / d.scale[i] : d.grad[i]));
            const int ne = d.n_e * e;
            for (int f = W.f0 + (int)threadIdx.x; f < W.f1; f += 256) {
                double dl[6], out7[7];
                for (int k = 0; k < 6; ++k) {
                    const int i = ne + f * 6 + k;
                    dl[k] = -(o.jacobi ? d.grad[i] / d.scale[i] : d.grad[i]);
                }
                const double *x = xp + 7 * d.pose_of_f[f];
                se3_plus(x, dl, out7);
                for (int k = 0; k < 7; ++k) v = fmax(v, fabs(x[k] - out7[k]));
            }
            gmax = block_max_256(v, sh);
        }
        if (threadIdx.x == 0) {
            if (fin) {
                if (first) {   // IterationZero
                    W.x_cost = cost; W.gmax = gmax;
                    W.initial_cost = cost; W.minimum_cost = cost;
                    W.x_norm = -1.0;            // Init(): "invalid value" until the first successful step
                    W.radius = initial_radius; W.decrease_factor = 2.0;
                    W.reuse_diagonal = 0; W.invalid_steps = 0; W.iteration = 0; W.last_ok = 1;
                    W.done = 0; W.termination = OV2_BA_TERM_MAX_ITER; W.eval_at_cand = 0;
                    win_log(W, cost, 0.0, W.radius, 0.0, 0.0, 1, 1);
                } else {       // HandleSuccessfulStep: the jacobian at the new x is in place
                    const double rel = W.cand_cost;   // WR_CAND parked the step's relative decrease / cost change here
                    const double cost_change = W.x_cost - cost;
                    W.x_cost = cost; W.gmax = gmax;
                    const double t = 2.0 * rel - 1.0;
                    W.radius = W.radius / fmax(1.0 / 3.0, 1.0 - t * t * t);   // StepAccepted
                    W.radius = fmin(o.max_radius, W.radius);
                    W.decrease_factor = 2.0;
                    W.reuse_diagonal = 0;
                    W.last_ok = 1;
                    W.eval_at_cand = 0;
                    if (cost < W.minimum_cost) W.minimum_cost = cost;
                    win_log(W, cost, cost_change, W.radius, rel, W.model_change, 1, 1);
                }
            }
            win_begin_round(W, o);
            if (W.active) atomicAdd(&d.n_active[next_round], 1);
        }
    } else if (MODE == WR_MODEL) {
        if (!W.active) return;
        const double msum = ordered_sum(part_model + W.e0, W.e1 - W.e0, sh);
        if (threadIdx.x == 0) {
            const double model_change = -msum;
            W.model_change = model_change;
            const bool valid = !W.chol_fail && isfinite(model_change) && model_change > 0.0;
            if (valid) W.valid = 1;
            else win_invalid_step(W, o, model_change);
        }
    } else {
        if (!(W.active && W.valid)) return;
        const int ne_w = W.e1 - W.e0, nf_w = W.f1 - W.f0;
        const double step2 = ordered_sum2(part_step + W.e0, ne_w, part_step + d.n_e + W.f0, nf_w, sh);
        const double xnorm2 = ordered_sum2(part_norm + W.e0, ne_w, part_norm + d.n_e + W.f0, nf_w, sh);
        const double cand_cost = ordered_sum(part_cost + W.vb0, W.vb1 - W.vb0, sh);
        if (threadIdx.x == 0) {
            if (!isfinite(step2)) { win_invalid_step(W, o, W.model_change); return; }
            W.invalid_steps = 0;
            W.eval_at_cand = 1;   // the cost functors were last evaluated at the candidate
            const double step_norm = sqrt(step2);
            const double cost_change = W.x_cost - cand_cost;
            if (step_norm <= o.ptol * (W.x_norm + o.ptol)) {            // ParameterToleranceReached
                W.termination = OV2_BA_TERM_PTOL; W.done = 1; W.active = 0;
            } else if (fabs(cost_change) <= o.ftol * W.x_cost) {        // FunctionToleranceReached: returns WITHOUT taking the candidate
                W.termination = OV2_BA_TERM_FTOL; W.done = 1; W.active = 0;
                win_log(W, W.x_cost, cost_change, W.radius, 0.0, W.model_change, 1, 0);
            } else {
                const double rel = isfinite(cand_cost) ? cost_change / W.model_change : -1e300;
                if (rel > o.min_rel) {                                   // IsStepSuccessful
                    W.accepted = 1;
                    W.x_norm = sqrt(xnorm2);
                    W.cand_cost = rel;                                   // read back by WR_JAC
                } else {                                                 // HandleUnsuccessfulStep
                    W.radius = W.radius / W.decrease_factor; W.decrease_factor *= 2.0;
                    W.last_ok = 0;
                    win_log(W, cand_cost, cost_change, W.radius, rel, W.model_change, 1, 0);
                }
            }
        }
    }
}

// x <- candidate for the windows whose step was accepted (whole window ranges: constant / unused blocks are equal in
// both buffers).  win_of_pose / win_of_lm come from the window offsets by bisection.
__device__ __forceinline__ int win_of_index(const int *__restrict__ off, int B, int i)
{
    int lo = 0, hi = B;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

// the same for a wave whose lanes hold ascending indices (i = first + lane): the bisection runs once, on the first
// lane's index with wave-uniform (scalar) loads, and a lane past the end of that window steps forward
__device__ __forceinline__ int win_of_index_asc(const int *__restrict__ off, int B, int i)
{
    const int i0 = __builtin_amdgcn_readfirstlane(i);
    int lo = 0, hi = B;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= i0) lo = mid; else hi = mid;
    }
    while (lo + 1 < B && off[lo + 1] <= i) ++lo;
    return lo;
}

__global__ __launch_bounds__(256) void ba_accept_kernel(ba_dev d, const int *__restrict__ pose_off,
                                                        const int *__restrict__ lm_off, double *__restrict__ xp,
                                                        const double *__restrict__ cp, double *__restrict__ xl,
                                                        const double *__restrict__ cl)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < d.n_pose) {
        if (d.W[win_of_index(pose_off, d.B, i)].accepted)
            for (int k = 0; k < 7; ++k) xp[7 * i + k] = cp[7 * i + k];
    }
    if (i < d.n_lm) {
        if (d.W[win_of_index(lm_off, d.B, i)].accepted)
            for (int k = 0; k < d.e; ++k) xl[(size_t)i * d.e + k] = cl[(size_t)i * d.e + k];
    }
}

// ------------------------------------------------------------------------------------------------------
// K_COLNORM helpers / scaling

// lower-triangle element order of a symmetric 6x6 block
__constant__ signed char c_tri_i[21] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5};
__constant__ signed char c_tri_j[21] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5};
__constant__ signed char c_tri_diag[6] = {0, 6, 11, 15, 18, 20};

// Jacobi scale of every column from the norms of the FIRST jacobian, and in the same pass what the unscaled norms /
// gradient / F'F become under it (column i scales by s_i: squared norm by s_i^2, gradient entry by s_i, F'F entry (i, j) by
// s_i s_j) -- the column-norm kernels are not run a second time over the rows at iteration zero.
__global__ void ba_make_scale_kernel(ba_dev d)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.nc) return;
    const double s = 1.0 / (1.0 + sqrt(d.sqn[i]));
    d.scale[i] = s;
    d.sqn[i] = d.sqn[i] * s * s;
    d.grad[i] = d.grad[i] * s;
    const int ne = d.n_e * d.e;
    if (i >= ne && (i - ne) % 6 == 0) {   // first column of a pose block also rescales the block's F'F; the six scales are
        const int f = (i - ne) / 6;       // re-derived from its diagonal (= the unscaled squared norms, touched by this thread only)
        double *FF = d.FFp + (size_t)f * 21;
        double sf[6];
        for (int c = 0; c < 6; ++c) sf[c] = 1.0 / (1.0 + sqrt(FF[c_tri_diag[c]]));
        for (int t = 0; t < 21; ++t) FF[t] = FF[t] * sf[c_tri_i[t]] * sf[c_tri_j[t]];
    }
}

// LevenbergMarquardtStrategy::ComputeStep :76-89, per window (blockIdx.y)
// LM diagonal of the window's columns (refreshed from the column norms after an accepted step) and, in the same launch,
// S_w = diag(D_f^2), rhs_w = 0.  The S part recomputes the few D_f it needs instead of reading what other threads of
// this launch write.
__global__ void ba_lmdiag_sinit_kernel(ba_dev d, double min_d, double max_d)
{
    BA_WAVE_PRIO();
    const ba_win &W = d.W[blockIdx.y];
    if (!W.active) return;
    const int e = d.e, refresh_diag = W.refresh_diag;
    const double radius = W.radius;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nce = (W.e1 - W.e0) * e, m = W.m;
    if (i < (size_t)(nce + m)) {
        const size_t col = i < (size_t)nce ? (size_t)W.e0 * e + i : (size_t)d.n_e * e + (size_t)W.f0 * 6 + (i - nce);
        if (refresh_diag) d.diag[col] = fmin(fmax(d.sqn[col], min_d), max_d);
        d.lmd[col] = sqrt(d.diag[col] / radius);
    }
    const size_t mm = (size_t)m * m;
    if (i < mm) {
        const int r = (int)(i % m), c = (int)(i / m);
        double v = 0.0;
        if (r == c) {
            const size_t col = (size_t)d.n_e * e + (size_t)W.f0 * 6 + r;
            const double dg = refresh_diag ? fmin(fmax(d.sqn[col], min_d), max_d) : d.diag[col];
            const double dv = sqrt(dg / radius);   // the same D the column's own thread stores in lmd
            v = dv * dv;
        }
        d.Spool[W.S_off + i] = v;
    }
    if (i < (size_t)m) d.rhs[(size_t)W.f0 * 6 + i] = 0.0;
}

// ------------------------------------------------------------------------------------------------------
// shared device helpers of the landmark kernels (SchurEliminator::Eliminate, schur_eliminator_impl.h:179-308)

// Packed 64-bit sort keys with runtime field widths (every sort is a hipcub SortKeys over the bit range that matters; the
// radix sort is stable, so whatever sits in the bits below the range rides along in its original order).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | lo);
}

__device__ __forceinline__ double readlane_f64(double v, int lane)   // lane must be wave-uniform
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | lo);
}

// n doubles (n even) from a 16-byte aligned address as 16-byte loads
template <int N>
__device__ __forceinline__ void load_d2(const double *__restrict__ p, double *out)
{
    const double2 *q = reinterpret_cast<const double2 *>(p);
#pragma unroll
    for (int i = 0; i < N / 2; ++i) { const double2 v = q[i]; out[2 * i] = v.x; out[2 * i + 1] = v.y; }
}

// world point of landmark block e (two 16-byte loads)
__device__ __forceinline__ void load_wpt(const ba_dev &d, size_t e, double *p)
{
    const double2 *q = reinterpret_cast<const double2 *>(d.wpt + e * WPT_S);
    const double2 a = q[0], b = q[1];
    p[0] = a.x; p[1] = a.y; p[2] = b.x;
}

// the 2 x 6 pose block of row r: U = [-G | G x p], p = world point of the row's landmark block
__device__ __forceinline__ void load_U(const ba_dev &d, size_t r, const double *__restrict__ p, double *U)
{
    double G[6];
    load_d2<6>(d.G + r * 6, G);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double a = G[3 * k], b = G[3 * k + 1], c = G[3 * k + 2];
        U[6 * k] = -a; U[6 * k + 1] = -b; U[6 * k + 2] = -c;
        U[6 * k + 3] = b * p[2] - c * p[1]; U[6 * k + 4] = c * p[0] - a * p[2]; U[6 * k + 5] = a * p[1] - b * p[0];
    }
}

__device__ __forceinline__ double row_sum(double v)
{
    v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);   // row_half_mirror
    v += dpp_f64<0x140>(v);   // row_mirror
    return v;
}

template <int E>
__device__ __forceinline__ void invert_ete(const double *ete, double *ie)
{
    if (E == 1) {
        ie[0] = 1.0 / ete[0];
    } else {
        const double a = ete[0], b = ete[1], c = ete[2], dd = ete[4], ee = ete[5], f = ete[8];
        const double A = dd * f - ee * ee, B = c * ee - b * f, C = b * ee - c * dd;
        const double id = 1.0 / (a * A + b * B + c * C);
        ie[0] = A * id; ie[1] = B * id; ie[2] = C * id;
        ie[3] = B * id; ie[4] = (a * f - c * c) * id; ie[5] = (b * c - a * ee) * id;
        ie[6] = C * id; ie[7] = ie[5]; ie[8] = (a * dd - b * b) * id;
    }
}

// sum over the GW (8 | 16) lanes of a landmark group, result in all of them
template <int GW>
__device__ __forceinline__ double grp_sum(double v)
{
    v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);   // row_half_mirror
    if (GW == 16) v += dpp_f64<0x140>(v);   // row_mirror
    return v;
}

#ifndef BS_LM_GW
#define BS_LM_GW 8   // lanes per landmark in the landmark-major kernels (column norms, elimination, back-substitution): 12 rows / 7
                     // cells per landmark are typical (six stereo observers + the anchor); with 16 lanes a wave carried four landmarks
                     // and ~6 KB of traffic through its ~10 us of dependent loads (bs_landmark 311 -> 248 us with 8)
#endif
#define BS_LM_PER_WG (256 / BS_LM_GW)

// column norms + gradient of the landmark (E) columns: GW lanes per landmark, no atomics
template <int GW>
__global__ __launch_bounds__(256) void ba_colnorm16_kernel(ba_dev d, int mode)
{
    BA_WAVE_PRIO();
    const int l = blockIdx.x * (256 / GW) + (int)(threadIdx.x / GW), sub = threadIdx.x % GW;
    const bool live = l < d.n_e && win_runs(d.W[d.win_of_e[l]], mode);
    const int e = d.e;
    const int r0 = live ? d.row_ptr[l] : 0, r1 = live ? d.row_ptr[l + 1] : 0;
    double se[3] = {0, 0, 0}, ge[3] = {0, 0, 0}, sc[3] = {1, 1, 1};
    if (live) for (int c = 0; c < e; ++c) sc[c] = d.scale[l * e + c];
    for (int base = r0; base < r1; base += GW) {
        const int r = base + sub;
        if (r < r1) {
            const double *Je = d.Je + (size_t)r * 2 * e;
            const double b0 = d.res[2 * r], b1 = d.res[2 * r + 1];
            for (int c = 0; c < e; ++c) {
                const double j0 = Je[c] * sc[c], j1 = Je[e + c] * sc[c];
                se[c] += j0 * j0 + j1 * j1; ge[c] += j0 * b0 + j1 * b1;
            }
        }
    }
    for (int c = 0; c < e; ++c) { se[c] = grp_sum<GW>(se[c]); ge[c] = grp_sum<GW>(ge[c]); }
    if (live && sub == 0)
        for (int c = 0; c < e; ++c) { d.sqn[l * e + c] = se[c]; d.grad[l * e + c] = ge[c]; }
}


// Normal-equation pieces of the pose (F) columns that only change with the jacobian: one workgroup per free pose
// gathers its (row, cell) entries through the pose -> entries CSR and forms F'F (lower triangle, 21 values: its diagonal
// is the column norms) and F'b (the gradient).  No atomics (the per-pose sums were the most contended addresses of the
// whole solve), fixed summation order => bitwise reproducible.  The Schur complement reuses F'F and F'b in every LM
// round until the next accepted step instead of re-deriving them from the rows.
// (Measured alternative, round 3: a pose-major copy of (G | residual), one 64-byte record per entry, written by the jacobian
// evaluation and read contiguously here and by the F'Fa terms of the gather.  This kernel 265 -> 181 us, but the evaluation
// 211 -> 320 us per launch (two scattered 64-byte stores per row, 1 GB per full evaluation of 8 M rows) and the gather 552 ->
// 708 us: 27.9 ms per batch of 64 windows against 24.4.  The rows of a batch are 640 MB, far beyond L2 and MALL; every extra
// copy is paid at HBM rate.)
__global__ __launch_bounds__(256) void ba_pose_normal_kernel(ba_dev d, const int *__restrict__ pose_ptr,
                                                             const int *__restrict__ pose_ent, int mode)
{
    BA_WAVE_PRIO();
    __shared__ double sh[4][27];
    const int f = blockIdx.x, tid = threadIdx.x;
    if (!win_runs(d.W[d.win_of_f[f]], mode)) return;
    double acc[27], sf[6];
#pragma unroll
    for (int c = 0; c < 27; ++c) acc[c] = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) sf[c] = d.scale[d.n_e * d.e + f * 6 + c];
    for (int k = pose_ptr[f] + tid; k < pose_ptr[f + 1]; k += 256) {
        const int ent = pose_ent[k], r = ent >> 1;
        double J[12], bb[2], wpk[3];
        load_wpt(d, (size_t)d.ent_lm[k], wpk);
        load_U(d, (size_t)r, wpk, J);
        load_d2<2>(d.res + 2 * (size_t)r, bb);
        // the pose's block of this row, Jacobi-scaled; as the row's anchor pose (entry bit 0) the block is -U
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            J[c] *= sf[c]; J[6 + c] *= sf[c];
            if (ent & 1) { J[c] = -J[c]; J[6 + c] = -J[6 + c]; }
        }
        const double b0 = bb[0], b1 = bb[1];
#pragma unroll
        for (int t = 0; t < 21; ++t) acc[t] += J[c_tri_i[t]] * J[c_tri_j[t]] + J[6 + c_tri_i[t]] * J[6 + c_tri_j[t]];
#pragma unroll
        for (int c = 0; c < 6; ++c) acc[21 + c] += J[c] * b0 + J[6 + c] * b1;
    }
#pragma unroll
    for (int c = 0; c < 27; ++c) {
        const double v = row_sum(acc[c]);
        const double w = (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
        if ((tid & 63) == 0) sh[tid >> 6][c] = w;
    }
    __syncthreads();
    if (tid < 27) {
        const double t = (sh[0][tid] + sh[1][tid]) + (sh[2][tid] + sh[3][tid]);
        const int ne = d.n_e * d.e;
        if (tid < 21) {
            d.FFp[(size_t)f * 21 + tid] = t;
            for (int c = 0; c < 6; ++c)
                if (c_tri_diag[c] == tid) d.sqn[ne + f * 6 + c] = t;
        } else d.grad[ne + f * 6 + tid - 21] = t;
    }
}

// back-substitution + model cost change, 16 lanes per landmark
template <int E, int GW>
__global__ __launch_bounds__(256) void ba_backsub16_kernel(ba_dev d, double *__restrict__ part)
{
    BA_WAVE_PRIO();
    {   // the pose part of the step is -z (z = the reduced solve's result in rhs); written here to save a launch
        const int gi = blockIdx.x * 256 + threadIdx.x;
        if (gi < d.m) d.step[d.n_e * d.e + gi] = -d.rhs[gi];
    }
    const int l = blockIdx.x * (256 / GW) + (int)(threadIdx.x / GW), sub = threadIdx.x % GW;
    const bool live = l < d.n_e && d.W[d.win_of_e[l]].active;
    const int r0 = live ? d.row_ptr[l] : 0, r1 = live ? d.row_ptr[l + 1] : 0;
    const double *__restrict__ sf = d.scale + (size_t)d.n_e * E;   // Jacobi scales of the pose columns
    double acc[E], se[E], wp[3];
    for (int i = 0; i < E; ++i) { acc[i] = 0.0; se[i] = live ? d.scale[(size_t)l * E + i] : 1.0; }
    wp[0] = wp[1] = wp[2] = 0.0;
    if (live) load_wpt(d, (size_t)l, wp);
    // One row of the landmark: its jacobian entries scaled (je), residual (b) and p = F_row * step_poses, the part of J * step
    // that comes from the row's pose and anchor-pose blocks.  The kernel needs p twice (E' (r + F z) for the landmark step,
    // then J * step for the model cost): the first two trips of a group keep (je, b, p) in registers, so the second sweep
    // re-reads nothing for landmarks of up to 2 GW rows -- before, it gathered the 24 (z, scale) values per row and the
    // 48-byte block again (the kernel was bound by the number of gather instructions, not by bytes).
    struct rowv { double je[2 * E], b0, b1, p0, p1; };
    auto load_row = [&](int r, rowv &v) {
        const double *Je = d.Je + (size_t)r * 2 * E;
        double Uu[12];
        load_U(d, (size_t)r, wp, Uu);
        for (int i = 0; i < E; ++i) { v.je[i] = Je[i] * se[i]; v.je[E + i] = Je[E + i] * se[i]; }
        v.b0 = d.res[2 * r]; v.b1 = d.res[2 * r + 1];
        double p0 = 0.0, p1 = 0.0;
        const int fk = d.fk[r], fa = d.fa[r];
        if (fk >= 0) for (int c = 0; c < 6; ++c) { const double z = d.rhs[fk * 6 + c], sk = sf[fk * 6 + c]; p0 -= (Uu[c] * sk) * z; p1 -= (Uu[6 + c] * sk) * z; }
        if (fa >= 0) for (int c = 0; c < 6; ++c) { const double z = d.rhs[fa * 6 + c], sa = sf[fa * 6 + c]; p0 += (Uu[c] * sa) * z; p1 += (Uu[6 + c] * sa) * z; }
        v.p0 = p0; v.p1 = p1;
    };
    auto sweep1 = [&](const rowv &v) {
        const double sj0 = v.b0 + v.p0, sj1 = v.b1 + v.p1;
        for (int i = 0; i < E; ++i) acc[i] += v.je[i] * sj0 + v.je[E + i] * sj1;
    };
    rowv c0, c1;
    const bool in0 = r0 + sub < r1, in1 = r0 + GW + sub < r1;
    if (in0) { load_row(r0 + sub, c0); sweep1(c0); }
    if (in1) { load_row(r0 + GW + sub, c1); sweep1(c1); }
    for (int base = r0 + 2 * GW; base < r1; base += GW) {
        const int r = base + sub;
        if (r < r1) { rowv v; load_row(r, v); sweep1(v); }
    }
    double y[E];
    for (int i = 0; i < E; ++i) acc[i] = grp_sum<GW>(acc[i]);
    for (int i = 0; i < E; ++i) {
        double s = 0;
        for (int j = 0; j < E; ++j) s += (live ? d.iete[(size_t)l * E * E + i * E + j] : 0.0) * acc[j];
        y[i] = s;
    }
    double mc = 0.0;
    auto sweep2 = [&](const rowv &v) {
        double m0 = 0.0, m1 = 0.0;
        for (int i = 0; i < E; ++i) { m0 -= v.je[i] * y[i]; m1 -= v.je[E + i] * y[i]; }
        m0 += v.p0; m1 += v.p1;
        mc += m0 * (v.b0 + m0 / 2.0) + m1 * (v.b1 + m1 / 2.0);
    };
    if (in0) sweep2(c0);
    if (in1) sweep2(c1);
    for (int base = r0 + 2 * GW; base < r1; base += GW) {
        const int r = base + sub;
        if (r < r1) { rowv v; load_row(r, v); sweep2(v); }
    }
    mc = grp_sum<GW>(mc);
    if (live && sub == 0) {
        for (int i = 0; i < E; ++i) d.step[l * E + i] = -y[i];
        part[l] = mc;
    }
}

// ------------------------------------------------------------------------------------------------------
// K_CHOL: dense Cholesky of the reduced camera system by one workgroup (schur_complement_solver.cc:217-229 does an
// Eigen LLT; :319-355 a sparse one -- same factor).  Column-major, lower triangle.  Left-looking, panels of NB
// columns held in LDS; the right-hand side rides along as row m, so the forward substitution is free; the backward
// substitution is done panel by panel afterwards.  z (solution) overwrites rhs.

// 512 threads: 256 VGPRs per lane, so the register-blocked phases need no scratch (with 1024 threads / 128 VGPRs the
// kernel spilled ~100 registers, and every dispatch that needs scratch stalled for milliseconds in the runtime's
// queue-scratch management -- measured: 12-30 ms per minimize() instead of 3)
#define CHOL_THREADS 512
template <int NB>
__global__ __launch_bounds__(CHOL_THREADS) void ba_chol_kernel(ba_dev d)
{
    BA_WAVE_PRIO();
    extern __shared__ __attribute__((aligned(16))) double lds[];   // the only LDS object of this kernel
    ba_win &W = d.W[blockIdx.x];                                   // one workgroup per window
    if (!W.active || W.m == 0) return;
    double *__restrict__ A = d.Spool + W.S_off;
    double *__restrict__ rhs = d.rhs + (size_t)W.f0 * 6;
    const int m = W.m;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int M1 = m + 1;               // augmented row count (row m = right-hand side)
    constexpr int PS = NB + 1;          // panel row stride (bank spread)
    constexpr int KC = 64, HALF = NB / 2;
    double *P = lds;                    // panel: rows x NB
    double *Lj = lds + (((size_t)M1 * PS + 1) & ~(size_t)1); // 16-B aligned; KC x NB chunk of the previous columns of rows j0..j0+nb  ([kk][c])
    volatile int *failp = reinterpret_cast<volatile int *>(Lj + NB * KC);
    if (tid == 0) *failp = 0;
    __syncthreads();
    for (int j0 = 0; j0 < m; j0 += NB) {
        const int nb = min(NB, m - j0);
        const int rows = M1 - j0;       // panel rows j0 .. m
        for (int idx = tid; idx < rows * nb; idx += nth) {
            const int c = idx / rows, i = idx - c * rows;
            const int gi = j0 + i;
            P[i * PS + c] = (gi < m) ? A[(size_t)(j0 + c) * m + gi] : rhs[j0 + c];
        }
        __syncthreads();
        // P -= L[j0.., 0..j0) * L[j0..j0+nb, 0..j0)^T ; thread = (row, half of the panel columns)
        const int groups = (rows * 2 <= nth) ? 2 : 1;
        for (int k0 = 0; k0 < j0; k0 += KC) {
            const int kc = min(KC, j0 - k0);
            for (int idx = tid; idx < NB * KC; idx += nth) {
                const int kk = idx / NB, c = idx - kk * NB;
                Lj[idx] = (kk < kc && c < nb) ? A[(size_t)(k0 + kk) * m + j0 + c] : 0.0;
            }
            __syncthreads();
            for (int w = tid; w < rows * groups; w += nth) {
                const int i = w % rows, g = w / rows;
                const int gi = j0 + i;
                for (int h = g; h < 2; h += groups) {
                    const int c0 = h * HALF;
                    double acc[HALF];
#pragma unroll
                    for (int c = 0; c < HALF; ++c) acc[c] = 0.0;
                    // row m (rhs) of previous columns lives in rhs[] after their panel was written back
                    const double *src = (gi < m) ? (A + (size_t)k0 * m + gi) : (rhs + k0);
                    const size_t sstep = (gi < m) ? (size_t)m : 1;
                    // register double buffer of UB L[i][k] values: the next batch is in flight while this one is used
                    // (sized for the 256-VGPR budget of a 512-thread workgroup: a dispatch that needs
                    // scratch costs milliseconds of queue-scratch management on this runtime, see CHOL_THREADS)
                    constexpr int UB = 16;
                    double cur[UB], nxt[UB];
#pragma unroll
                    for (int u = 0; u < UB; ++u) cur[u] = (u < kc) ? src[(size_t)u * sstep] : 0.0;
                    for (int kk0 = 0; kk0 < kc; kk0 += UB) {
#pragma unroll
                        for (int u = 0; u < UB; ++u) nxt[u] = (kk0 + UB + u < kc) ? src[(size_t)(kk0 + UB + u) * sstep] : 0.0;
#pragma unroll
                        for (int u = 0; u < UB; ++u) {
#pragma unroll
                            for (int c = 0; c < HALF; ++c) acc[c] += cur[u] * Lj[(kk0 + u) * NB + c0 + c];
                        }
#pragma unroll
                        for (int u = 0; u < UB; ++u) cur[u] = nxt[u];
                    }
#pragma unroll
                    for (int c = 0; c < HALF; ++c)
                        if (c0 + c < nb) P[i * PS + c0 + c] -= acc[c];
                }
            }
            __syncthreads();
        }
        // (a) the nb x nb diagonal block: ONE wave, lane i keeps row i in registers, pivots travel by v_readlane
        if (tid < 64) {
            double row[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) row[c] = (tid < nb && c < nb) ? P[tid * PS + c] : ((tid == c) ? 1.0 : 0.0);
            int bad = 0;
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                const double dcc = readlane_f64(row[c], c);
                if (c < nb && !(dcc > 0.0)) bad = 1;
                const double dsq = sqrt(dcc > 0.0 ? dcc : 1.0);
                row[c] = (tid == c) ? dsq : row[c] / dsq;
#pragma unroll
                for (int c2 = c + 1; c2 < NB; ++c2) {
                    const double l = readlane_f64(row[c], c2);
                    row[c2] -= row[c] * l;      // rows above the diagonal collect garbage that is never read
                }
            }
            if (tid < nb) {
#pragma unroll
                for (int c = 0; c < NB; ++c)
                    if (c <= tid && c < nb) P[tid * PS + c] = row[c];
            }
            if (bad && tid == 0) *failp = 1;
        }
        __syncthreads();
        if (*failp) break;
        // (b) rows below the block (and the rhs row): x L_D^T = P[i,:], forward substitution, one thread per row
        for (int i = nb + tid; i < rows; i += nth) {
            double x[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) x[c] = (c < nb) ? P[i * PS + c] : 0.0;
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                if (c < nb) {
                    double v = x[c];
#pragma unroll
                    for (int k = 0; k < c; ++k) v -= x[k] * P[c * PS + k];
                    x[c] = v / P[c * PS + c];
                }
            }
#pragma unroll
            for (int c = 0; c < NB; ++c)
                if (c < nb) P[i * PS + c] = x[c];
        }
        __syncthreads();
        for (int idx = tid; idx < rows * nb; idx += nth) {
            const int c = idx / rows, i = idx - c * rows;
            const int gi = j0 + i;
            if (gi < m) { if (i >= c) A[(size_t)(j0 + c) * m + gi] = P[i * PS + c]; }
            else rhs[j0 + c] = P[i * PS + c];   // y = L^-1 b rides along as row m
        }
        __syncthreads();
    }
    if (*failp) {
        if (tid == 0) W.chol_fail = 1;
        return;
    }
    __threadfence_block();
    __syncthreads();
    // backward substitution L^T z = y, panel by panel from the bottom: (1) every wave takes panel columns and forms
    // t_c = sum_{i below the panel} L[i][c] z[i] with coalesced column reads, (2) wave 0 solves the nb x nb triangle
    // with the block's columns in registers (pivots by v_readlane).  z lives in LDS.
    {
        double *zb = lds, *tpart = lds + m;
        const int lane = tid & 63, wave = tid >> 6, nwaves = nth >> 6;
        for (int i = tid; i < m; i += nth) zb[i] = rhs[i];
        __syncthreads();
        for (int j0 = ((m - 1) / NB) * NB; j0 >= 0; j0 -= NB) {
            const int nb = min(NB, m - j0);
            for (int c = wave; c < nb; c += nwaves) {
                double sacc = 0.0;
                for (int i = j0 + nb + lane; i < m; i += 64) sacc += A[(size_t)(j0 + c) * m + i] * zb[i];
                for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);
                if (lane == 0) tpart[c] = sacc;
            }
            __syncthreads();
            if (tid < 64) {
                double y = (lane < nb) ? zb[j0 + lane] - tpart[lane] : 0.0;
                double colD[NB];   // lane i: colD[j] = L[j0+j][j0+i], j >= i
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    colD[j] = (lane < nb && j < nb && j >= lane) ? A[(size_t)(j0 + lane) * m + j0 + j] : ((j == lane) ? 1.0 : 0.0);
#pragma unroll
                for (int j = NB - 1; j >= 0; --j) {
                    const double zj = readlane_f64(y, j) / readlane_f64(colD[j], j);
                    if (lane == j) y = zj;
                    else if (lane < j) y -= colD[j] * zj;
                }
                if (lane < nb) zb[j0 + lane] = y;
            }
            __syncthreads();
        }
        for (int i = tid; i < m; i += nth) rhs[i] = zb[i];
    }
}

// ------------------------------------------------------------------------------------------------------
// K_CHOL, one workgroup per window with the panel update on the matrix cores.  Same left-looking blocked factorisation
// as ba_chol_kernel (32-column panel in LDS, right-hand side as row m, diagonal block by one wave, rows below by
// forward substitution, blocked backward substitution) -- what changes is the O(m^3) part: the update of a panel with
// the columns before it, P -= L[j0.., 0..j0) L[j0..j0+32, 0..j0)^T, is a GEMM, and ba_chol_kernel spent it one LDS read
// per fused multiply-add (every thread a row, 16 columns each).  Here every wave takes 16-row tiles of the panel (both
// 16-column halves at once, sharing the row operand) and walks k in steps of four with v_mfma_f64_16x16x4_f64, both
// operands straight from the factor in L2 (for a fixed k, 16 consecutive rows are 128 contiguous bytes): no LDS traffic
// and no barrier inside the update.
typedef double ov2_v4f64 __attribute__((ext_vector_type(4)));
// 1 / d by v_rcp_f64 + two Newton steps (within an ulp of the correctly rounded quotient): an IEEE f64 division is a ~35
// instruction sequence here, and the substitution loops of the factorisation made one per row and column
__device__ __forceinline__ double chol_rcp(double d)
{
    double x = __builtin_amdgcn_rcp(d);
    x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
    x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
    return x;
}

#ifdef OV2_CHOL_PROF
__device__ unsigned long long g_chol_prof[8];   // phase clocks (100 MHz ticks) of workgroup 0: update, diagonal, substitution, write-back, backward
#define CHOL_TICK(slot) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); g_chol_prof[slot] += t_ - t_prev; t_prev = t_; } } while (0)
#else
#define CHOL_TICK(slot) do { } while (0)
#endif

// sqrt(d) and 1 / sqrt(d) for a pivot (d > 0, far from the ends of the exponent range: a sum of squared jacobian entries plus
// the LM diagonal): v_rsq_f64 + two Newton steps for the reciprocal root, the root as d * that with one correction.  The
// pivot sits on the critical path of the one-wave diagonal block: sqrt() followed by a reciprocal was ~60 dependent
// instructions per column, this is 14.
__device__ __forceinline__ void chol_sqrt_rcp(double d, double &root, double &inv)
{
    double y = __builtin_amdgcn_rsq(d);
    const double hd = 0.5 * d;
    y = __builtin_fma(y, __builtin_fma(-hd * y, y, 0.5), y);
    y = __builtin_fma(y, __builtin_fma(-hd * y, y, 0.5), y);
    double r = d * y;
    r = __builtin_fma(__builtin_fma(-r, r, d), 0.5 * y, r);
    root = r;
    inv = __builtin_fma(__builtin_fma(-r, y, 1.0), y, y);   // 1 / root (y is 1 / sqrt(d); one step onto the rounded root)
}

#ifndef CHOL_TRIP
#define CHOL_TRIP 8   // k-steps per trip of the panel update (4 columns each): 32 = the panel width divides every j0
#endif
__global__ __launch_bounds__(CHOL_THREADS) void ba_chol_mfma_kernel(ba_dev d)
{
    BA_WAVE_PRIO();
#ifdef OV2_CHOL_PROF
    unsigned long long t_prev = wall_clock64();
#endif
    constexpr int NB = 32, PS = NB + 1;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    ba_win &W = d.W[blockIdx.x];
    if (!W.active || W.m == 0) return;
    double *__restrict__ A = d.Spool + W.S_off;
    double *__restrict__ rhs = d.rhs + (size_t)W.f0 * 6;
    const int m = W.m;
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, wave = tid >> 6, nwaves = nth >> 6;
    const int M1 = m + 1;
    double *P = lds;                                   // panel: rows x NB, stride PS
    double *invd = lds + (((size_t)M1 * PS + 1) & ~(size_t)1);   // reciprocals of the panel's diagonal (NB)
    volatile int *failp = reinterpret_cast<volatile int *>(invd + NB);
    if (tid == 0) *failp = 0;
    __syncthreads();
    const int li = lane & 15, kq = lane >> 4;
    for (int j0 = 0; j0 < m; j0 += NB) {
        const int nb = min(NB, m - j0);
        const int rows = M1 - j0;                      // panel rows j0 .. m (row m = right-hand side)
        const int ntile = (rows + 15) >> 4;
        // the two 16-column blocks of the panel as the MFMA's first operand: lane (li, kq) holds L[j0 + 16 h + li][k + kq]
        const int ca = j0 + li, cb = j0 + 16 + li;
        const bool ca_ok = li < nb, cb_ok = 16 + li < nb;
        for (int t = wave; t < ntile; t += nwaves) {
            const int r0 = j0 + 16 * t, rr = r0 + li;  // this lane's row of the tile (rr == m: the right-hand side row)
            const double *rsrc = rr < m ? A + rr : rhs;
            const size_t rstep = rr < m ? (size_t)m : 1;
            const bool r_ok = rr <= m;
            ov2_v4f64 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
            // j0 is a multiple of 32: eight k-steps per trip, their 24 operand loads issued together (one L2 round trip per
            // trip; issued step by step every matrix instruction waited for its own loads -- 53 us per panel at m = 354).
            // Measured and not kept: double-buffered trips (48 operands + 48 addresses live: 256 VGPRs, 636 B of scratch);
            // two tiles per wave and trip sharing the column operands (32 loads instead of 2 x 24: update phase 694 -> 765 us)
            // ; the panel's 32 x j0 column block staged through LDS once per 32 columns for all waves, a wave keeping the accumulators
            // of its three tiles (one round trip feeds 48 matrix instructions): 693 -> 942 us -- the two barriers per step make every
            // step as slow as the slowest wave's loads, where independent waves overlap each other's latencies
            for (int k = 0; k < j0; k += 4 * CHOL_TRIP) {
                double b[CHOL_TRIP], a0[CHOL_TRIP], a1[CHOL_TRIP];
#pragma unroll
                for (int u = 0; u < CHOL_TRIP; ++u) {
                    const size_t kk = (size_t)(k + 4 * u + kq);
                    const double *col = A + kk * m;
                    b[u] = r_ok ? rsrc[kk * rstep] : 0.0;
                    a0[u] = ca_ok ? col[ca] : 0.0;
                    a1[u] = cb_ok ? col[cb] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < CHOL_TRIP; ++u) {
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b[u], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b[u], acc1, 0, 0, 0);
                }
            }
            // acc_h[q] = sum_k L[j0 + 16 h + kq + 4 q][k] L[rr][k]: the update of P[rr][16 h + kq + 4 q]
            if (r_ok) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = kq + 4 * q, c1 = 16 + kq + 4 * q;
                    if (c0 < nb) P[(rr - j0) * PS + c0] = (rr < m ? A[(size_t)(j0 + c0) * m + rr] : rhs[j0 + c0]) - acc0[q];
                    if (c1 < nb) P[(rr - j0) * PS + c1] = (rr < m ? A[(size_t)(j0 + c1) * m + rr] : rhs[j0 + c1]) - acc1[q];
                }
            }
        }
        __syncthreads();
        CHOL_TICK(0);
        // (a) the nb x nb diagonal block: ONE wave, lane i keeps row i in registers, pivots travel by v_readlane
        if (tid < 64) {
            double row[NB];
            int tl = tid;   // opaque: the 32 identity-padding constants (tl == c ? 1 : 0) are otherwise hoisted out of the panel
            asm volatile("" : "+v"(tl));   // loop and held in 64 VGPRs for the whole kernel (spills)
#pragma unroll
            for (int c = 0; c < NB; ++c) row[c] = (tl < nb && c < nb) ? P[tl * PS + c] : ((tl == c) ? 1.0 : 0.0);
            int bad = 0;
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                const double dcc = readlane_f64(row[c], c);
                if (!(dcc > 0.0)) bad = 1;   // (columns beyond a short last panel are identity columns: 1 > 0)
                double dsq, inv;
                chol_sqrt_rcp(dcc > 0.0 ? dcc : 1.0, dsq, inv);
                if (tl == c) invd[c] = inv;
                row[c] = (tl == c) ? dsq : row[c] * inv;
#pragma unroll
                for (int c2 = c + 1; c2 < NB; ++c2) {
                    const double l = readlane_f64(row[c], c2);
                    row[c2] = __builtin_fma(-row[c], l, row[c2]);      // rows above the diagonal collect garbage that is never read
                }
            }
            if (tl < nb) {
#pragma unroll
                for (int c = 0; c < NB; ++c)
                    if (c <= tl) P[tl * PS + c] = row[c];
            }
            if (bad && tid == 0) *failp = 1;
        }
        __syncthreads();
        CHOL_TICK(1);
        if (*failp) break;
        // (b) rows below the block (and the rhs row): x L_D^T = P[i,:], forward substitution, one thread per row
        for (int i = nb + tid; i < rows; i += nth) {
            double x[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) x[c] = P[i * PS + c];
            // The test "column inside a short last panel" is made on a value the compiler cannot see through, once per column:
            // written as `c < nb` it is 32 wave-uniform conditions kept live across the whole unrolled body (more scalar
            // registers than there are: spilled to scratch through VGPR lanes), and without any test the 496 reads of L_D are
            // invariant in the row loop and hoisted out of it (4 KB of spills per lane)
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                int nbv = nb;
                asm volatile("" : "+v"(nbv));
                if (c < nbv) {
                    double v = x[c];
#pragma unroll
                    for (int k = 0; k < c; ++k) v = __builtin_fma(-x[k], P[c * PS + k], v);
                    x[c] = v * invd[c];
                }
            }
#pragma unroll
            for (int c = 0; c < NB; ++c)
                if (c < nb) P[i * PS + c] = x[c];
        }
        __syncthreads();
        CHOL_TICK(2);
        for (int c = wave; c < nb; c += nwaves) {   // a column per wave: consecutive lanes, consecutive rows
            double *__restrict__ colA = A + (size_t)(j0 + c) * m + j0;
            for (int i = c + lane; i < rows; i += 64) {
                if (j0 + i < m) colA[i] = P[i * PS + c];
                else rhs[j0 + c] = P[i * PS + c];   // y = L^-1 b rides along as row m
            }
        }
        __threadfence_block();
        __syncthreads();
        CHOL_TICK(3);
    }
    if (*failp) {
        if (tid == 0) W.chol_fail = 1;
        return;
    }
    __threadfence_block();
    __syncthreads();
    // backward substitution L^T z = y, as in ba_chol_kernel
    {
        double *zb = lds, *tpart = lds + m;
        for (int i = tid; i < m; i += nth) zb[i] = rhs[i];
        __syncthreads();
        for (int j0 = ((m - 1) / NB) * NB; j0 >= 0; j0 -= NB) {
            const int nb = min(NB, m - j0);
            for (int c = wave; c < nb; c += nwaves) {
                double sacc = 0.0;
                for (int i = j0 + nb + lane; i < m; i += 64) sacc += A[(size_t)(j0 + c) * m + i] * zb[i];
                for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);
                if (lane == 0) tpart[c] = sacc;
            }
            __syncthreads();
            if (tid < 64) {
                double y = (lane < nb) ? zb[j0 + lane] - tpart[lane] : 0.0;
                double colD[NB];   // lane i: colD[j] = L[j0+j][j0+i], j >= i
#pragma unroll
                for (int j = 0; j < NB; ++j)   // (the diagonal goes through dinv, entries above it are never used)
                    colD[j] = (lane < nb && j < nb && j > lane) ? A[(size_t)(j0 + lane) * m + j0 + j] : 0.0;
                const double dinv = chol_rcp(lane < nb ? A[(size_t)(j0 + lane) * m + j0 + lane] : 1.0);   // lane j: 1 / L[j0+j][j0+j]
#pragma unroll
                for (int j = NB - 1; j >= 0; --j) {
                    const double zj = readlane_f64(y, j) * readlane_f64(dinv, j);
                    if (lane == j) y = zj;
                    else if (lane < j) y -= colD[j] * zj;
                }
                if (lane < nb) zb[j0 + lane] = y;
            }
            __syncthreads();
        }
        for (int i = tid; i < m; i += nth) rhs[i] = zb[i];
    }
    CHOL_TICK(4);
}

// ------------------------------------------------------------------------------------------------------
// K_CHOL, multi-workgroup form for m >= CHOL_MULTI_MIN: right-looking blocked Cholesky, two launches per 32-column
// panel.  (A) every workgroup (one wave) factors the 32 x 32 diagonal block redundantly in registers -- 3 us, cheaper
// than a third launch -- and solves x L_D^T = a for 64 rows below it (the right-hand side rides along as row m);
// L_D goes to a side buffer (the rows below are still reading the unfactored block from A).  (B) the trailing matrix
// gets A22 -= L21 L21^T on the matrix cores: one wave per 16 x 16 tile of the lower triangle, eight
// v_mfma_f64_16x16x4_f64 per tile (operands straight from L2: lanes of one k read 16 consecutive doubles), plus the
// rhs row.  Then one workgroup does the blocked backward substitution.  m = 480: 1.52 -> ~0.6 ms per factorisation
// + solve against the one-workgroup kernel above; no upper limit on m any more.
#define CHOL_NB 32
#define CHOL_MULTI_MIN 400   // measured (3-iteration minimize): m = 480: 9.9 -> 5.9 ms against the one-workgroup kernel.  At
                             // m = 250 the 16 launches win 6 % on an idle GPU (3.28 -> 3.08 ms) but lose 9 % beside a busy
                             // front-end (332 vs 363 LM iterations/s: every dispatch queues), so the single launch stays there.
                             // Round 3, batches of 64 distinct windows with the one-workgroup kernel's panel update on the
                             // matrix cores (ba_chol_mfma_kernel): m <= 354: 24.4 ms per batch against 25.5 (this path) and 26.2
                             // (ba_chol_kernel); m ~ 440 (75 keyframes): 17.1 against 16.0 -- the crossover sits near 400

__global__ __launch_bounds__(64) void ba_chol_panel_kernel(ba_dev d, double *__restrict__ Dpool, size_t dstride, int k0)
{
    BA_WAVE_PRIO();
    constexpr int NB = CHOL_NB;
    __shared__ double LD[NB][NB + 1];
    ba_win &W = d.W[blockIdx.y];   // window = blockIdx.y; the x grid is sized for the largest window
    const int m = W.m;
    if (!W.active || k0 >= m) return;
    if (*(volatile int *)&W.chol_fail) return;
    double *__restrict__ A = d.Spool + W.S_off;
    double *__restrict__ rhs = d.rhs + (size_t)W.f0 * 6;
    double *__restrict__ Dbuf = Dpool + (size_t)blockIdx.y * dstride;
    const int lane = threadIdx.x;
    const int nb = min(NB, m - k0);
    double row[NB];   // lane i: row i of the diagonal block (lower part), identity outside
#pragma unroll
    for (int c = 0; c < NB; ++c)
        row[c] = (lane < nb && c <= lane) ? A[(size_t)(k0 + c) * m + k0 + lane] : ((lane == c) ? 1.0 : 0.0);
    int bad = 0;
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        const double dcc = readlane_f64(row[c], c);
        if (c < nb && !(dcc > 0.0)) bad = 1;
        const double dsq = sqrt(dcc > 0.0 ? dcc : 1.0);
        row[c] = (lane == c) ? dsq : row[c] / dsq;
#pragma unroll
        for (int c2 = c + 1; c2 < NB; ++c2) {
            const double l = readlane_f64(row[c], c2);
            row[c2] -= row[c] * l;      // entries above the diagonal collect garbage that is never read
        }
    }
    if (bad) {
        if (blockIdx.x == 0 && lane == 0) W.chol_fail = 1;
        return;
    }
    if (lane < NB) {
#pragma unroll
        for (int c = 0; c < NB; ++c) LD[lane][c] = row[c];
    }
    if (blockIdx.x == 0 && lane < NB) {
        double *D = Dbuf + (size_t)(k0 / NB) * NB * NB;   // column-major block: D[c * NB + r] = L[k0 + r][k0 + c]
#pragma unroll
        for (int c = 0; c < NB; ++c) D[c * NB + lane] = (c <= lane) ? row[c] : 0.0;
    }
    __syncthreads();
    // rows below the block (r < m) and the right-hand side (r == m): forward substitution against L_D
    const int r = k0 + nb + blockIdx.x * 64 + lane;
    if (r > m) return;
    double x[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) x[c] = (c < nb) ? ((r < m) ? A[(size_t)(k0 + c) * m + r] : rhs[k0 + c]) : 0.0;
    // left-looking per entry: a chain of ~500 dependent FMAs (2 us).  The right-looking variant (each final x[c] updating
    // all later entries at once) has more parallelism, but the scheduler hoisted its 496 LDS reads and the kernel
    // spilled 624 registers into 2.5 KB of scratch per lane -- and scratch costs milliseconds per dispatch here.
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        if (c < nb) {
            double v = x[c];
#pragma unroll
            for (int k = 0; k < c; ++k) v -= x[k] * LD[c][k];
            x[c] = v / LD[c][c];
        }
    }
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        if (c < nb) {
            if (r < m) A[(size_t)(k0 + c) * m + r] = x[c];
            else rhs[k0 + c] = x[c];
        }
    }
}

__global__ __launch_bounds__(64) void ba_chol_syrk_kernel(ba_dev d, int k0)
{
    BA_WAVE_PRIO();
    constexpr int NB = CHOL_NB;
    const ba_win &W = d.W[blockIdx.y];
    const int m = W.m;
    if (!W.active || k0 + NB >= m) return;
    if (*(volatile const int *)&W.chol_fail) return;
    double *__restrict__ A = d.Spool + W.S_off;
    double *__restrict__ rhs = d.rhs + (size_t)W.f0 * 6;
    const int t0 = k0 + NB, t = m - t0;
    const int nt = (t + 15) / 16, ntile = nt * (nt + 1) / 2;
    const int lane = threadIdx.x, b = blockIdx.x;
    if (b < ntile) {
        int ti = (int)((sqrtf(8.f * (float)b + 1.f) - 1.f) * 0.5f);
        while ((ti + 1) * (ti + 2) / 2 <= b) ++ti;
        while (ti * (ti + 1) / 2 > b) --ti;
        const int tj = b - ti * (ti + 1) / 2;
        const int r0 = t0 + 16 * ti, c0 = t0 + 16 * tj;
        const int i = lane & 15, kq = lane >> 4;
        // D'[i][j] = sum_k L[c0+i][k] L[r0+j][k] = C[r0+j][c0+i]: the result's lane index walks ROWS of the column-major
        // trailing matrix, so the update below is four runs of 16 consecutive doubles per register
        ov2_v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < NB; kk += 4) {
            const double *col = A + (size_t)(k0 + kk + kq) * m;
            const double a = (c0 + i < m) ? col[c0 + i] : 0.0;
            const double bb = (r0 + i < m) ? col[r0 + i] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc, 0, 0, 0);
        }
        const int rowi = r0 + i;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cj = c0 + kq + 4 * q;
            if (rowi < m && cj < m && rowi >= cj) A[(size_t)cj * m + rowi] -= acc[q];
        }
    } else {
        const int j = t0 + (b - ntile) * 64 + lane;   // the right-hand side row of the trailing part
        if (j < m) {
            double sacc = 0.0;
#pragma unroll 8
            for (int k = 0; k < NB; ++k) sacc += rhs[k0 + k] * A[(size_t)(k0 + k) * m + j];
            rhs[j] -= sacc;
        }
    }
}

// backward substitution L^T z = y after the panel kernels: panel by panel from the bottom, (1) every wave takes panel
// columns and forms t_c = sum_{i below the panel} L[i][c] z[i] with coalesced column reads, (2) wave 0 solves the
// nb x nb triangle with the block's columns (from the side buffer) in registers, pivots by v_readlane.  z in LDS.
__global__ __launch_bounds__(256) void ba_chol_backward_kernel(ba_dev d, const double *__restrict__ Dpool, size_t dstride)
{
    BA_WAVE_PRIO();
    constexpr int NB = CHOL_NB;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const ba_win &W = d.W[blockIdx.x];
    const int m = W.m;
    if (!W.active || m == 0) return;
    if (*(volatile const int *)&W.chol_fail) return;
    const double *__restrict__ A = d.Spool + W.S_off;
    double *__restrict__ rhs = d.rhs + (size_t)W.f0 * 6;
    const double *__restrict__ Dbuf = Dpool + (size_t)blockIdx.x * dstride;
    double *zb = lds, *tpart = lds + m;
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, wave = tid >> 6, nwaves = nth >> 6;
    for (int i = tid; i < m; i += nth) zb[i] = rhs[i];
    __syncthreads();
    for (int j0 = ((m - 1) / NB) * NB; j0 >= 0; j0 -= NB) {
        const int nb = min(NB, m - j0);
        for (int c = wave; c < nb; c += nwaves) {
            double sacc = 0.0;
            for (int i = j0 + nb + lane; i < m; i += 64) sacc += A[(size_t)(j0 + c) * m + i] * zb[i];
            for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);
            if (lane == 0) tpart[c] = sacc;
        }
        __syncthreads();
        if (tid < 64) {
            const double *D = Dbuf + (size_t)(j0 / NB) * NB * NB;
            double y = (lane < nb) ? zb[j0 + lane] - tpart[lane] : 0.0;
            double colD[NB];   // lane i: colD[j] = L[j0+j][j0+i], j >= i
#pragma unroll
            for (int j = 0; j < NB; ++j)
                colD[j] = (lane < nb && j < nb && j >= lane) ? D[lane * NB + j] : ((j == lane) ? 1.0 : 0.0);
#pragma unroll
            for (int j = NB - 1; j >= 0; --j) {
                const double zj = readlane_f64(y, j) / readlane_f64(colD[j], j);
                if (lane == j) y = zj;
                else if (lane < j) y -= colD[j] * zj;
            }
            if (lane < nb) zb[j0 + lane] = y;
        }
        __syncthreads();
    }
    for (int i = tid; i < m; i += nth) rhs[i] = zb[i];
}



// K_PLUS: candidate = Plus(x, step .* scale); partials: |x - cand|^2 and |cand|^2 over the free blocks (global size)
__global__ __launch_bounds__(64) void ba_plus_kernel(ba_dev d, const double *__restrict__ xp, const double *__restrict__ xl,
                                                     double *__restrict__ cp, double *__restrict__ cl,
                                                     const double *__restrict__ delta_vec, int use_scale,
                                                     double *__restrict__ part_step, double *__restrict__ part_norm)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int e = d.e, ne = d.n_e * e;
    if (i < d.n_e + d.n_f) {
        const ba_win &W = d.W[i < d.n_e ? d.win_of_e[i] : d.win_of_f[i - d.n_e]];
        if (!(W.active && W.valid)) return;
    }
    if (d.masked && i < d.n_e + d.n_f && (i < d.n_e ? d.e_dead[i] != 0 : d.f_live[i - d.n_e] == 0)) {
        part_step[i] = 0.0; part_norm[i] = 0.0;   // not a block of the reduced program: its step is zero, the candidate keeps x
        return;
    }
    if (i < d.n_e) {
        const int l = d.lm_of_e[i];
        double s2 = 0, n2 = 0;
        for (int c = 0; c < e; ++c) {
            const double dl = delta_vec[i * e + c] * (use_scale ? d.scale[i * e + c] : 1.0);
            const double v = xl[l * e + c] + dl;
            cl[l * e + c] = v;
            s2 += (xl[l * e + c] - v) * (xl[l * e + c] - v);
            n2 += v * v;
        }
        part_step[i] = s2; part_norm[i] = n2;
    } else if (i < d.n_e + d.n_f) {
        const int f = i - d.n_e, p = d.pose_of_f[f];
        double dl[6], out[7];
        for (int c = 0; c < 6; ++c) dl[c] = delta_vec[ne + f * 6 + c] * (use_scale ? d.scale[ne + f * 6 + c] : 1.0);
        se3_plus(xp + 7 * p, dl, out);
        double s2 = 0, n2 = 0;
        for (int c = 0; c < 7; ++c) {
            s2 += (xp[7 * p + c] - out[c]) * (xp[7 * p + c] - out[c]);
            n2 += out[c] * out[c];
            cp[7 * p + c] = out[c];
        }
        part_step[i] = s2; part_norm[i] = n2;
    }
}

// K_FLAG: chi2 / depth flags of every row of the program, written at the ORIGINAL residual index.  The reference reads
// the cost functors' cached chi2err_ / isdepthpositive_ (src/optimizer.cpp:500-592; src/ceres_parametrization.cpp:136-146),
// i.e. the values of the LAST Evaluate() call, and Ceres does not re-evaluate after Solve: after an accepted last step that
// is the final x, after a FTOL / PTOL exit or a rejected last step it is the candidate (trust_region_minimizer.cc:108-131).
// Rows that fail the test leave the device-side active set and are marked dead in the program (the L2 re-solve runs the
// same program with them masked); the per-window tallies drive the decision for the L2 refinement (src/optimizer.cpp:603-608).
__global__ __launch_bounds__(256) void ba_flag_kernel(ba_dev d, const double *__restrict__ xp, const double *__restrict__ xl,
                                                      const double *__restrict__ cp, const double *__restrict__ cl,
                                                      const int *__restrict__ rows, double chi2_th, int pass,
                                                      double *__restrict__ chi2, unsigned char *__restrict__ depth,
                                                      unsigned char *__restrict__ active, unsigned char *__restrict__ outlier)
{
    BA_WAVE_PRIO();
    const int row = blockIdx.x * 256 + threadIdx.x;
    const bool in = row < d.n_rows;
    const int w = d.row_win[in ? row : d.n_rows - 1];
    ba_win &W = d.W[w];
    bool bad = false, left = false, right = false;
    if (in && !(pass == 2 && (W.skip || d.dead[row]))) {   // second pass: the rows of the L2 program only
        const bool at_cand = W.eval_at_cand != 0;
        row_eval ev;
        eval_row<false>(d, d.wc[w], at_cand ? cp : xp, at_cand ? cl : xl, row, ev);
        // the per-residual outputs go to the ORIGINAL residual index (scattered 8- and 1-byte stores): written only when a
        // window of the batch asked for them (null otherwise); the solver itself needs the dead byte of the sorted row
        const int i = rows[row];
        if (chi2) chi2[i] = ev.chi2;
        if (depth) depth[i] = ev.depth_pos ? 1 : 0;
        if (ev.chi2 > chi2_th || !ev.depth_pos) {
            if (outlier) outlier[i] = (unsigned char)pass;
            d.dead[row] = 1;
            bad = true;
        } else {
            const int t = d.type[row];
            left = (t == OV2_BA_L_XYZ || t == OV2_BA_L_INV);
            right = (t == OV2_BA_R_XYZ || t == OV2_BA_R_INV);
        }
    }
    // tallies: the three counters of a window share a cache line and same-line atomics serialise, so they are summed over
    // the workgroup first (one atomic per counter and workgroup when it sits inside one window -- the usual case; a
    // workgroup that straddles a window boundary falls back to one atomic per wave / lane)
    __shared__ int tal[3][4];
    __shared__ int wfirst[4], wmixed[4];
    const int wv = threadIdx.x >> 6;
    const int w0 = __builtin_amdgcn_readfirstlane(w);
    const bool mixed = __ballot(w != w0) != 0ull;
    const int nb = __popcll(__ballot(bad)), nl = __popcll(__ballot(left)), nr = __popcll(__ballot(right));
    if ((threadIdx.x & 63) == 0) { tal[0][wv] = nb; tal[1][wv] = nl; tal[2][wv] = nr; wfirst[wv] = w0; wmixed[wv] = mixed ? 1 : 0; }
    __syncthreads();
    const bool block_uniform = !(wmixed[0] | wmixed[1] | wmixed[2] | wmixed[3]) && wfirst[0] == wfirst[1] && wfirst[1] == wfirst[2] &&
                               wfirst[2] == wfirst[3];
    if (block_uniform) {
        if (threadIdx.x < 3) {
            const int tot = tal[threadIdx.x][0] + tal[threadIdx.x][1] + tal[threadIdx.x][2] + tal[threadIdx.x][3];
            ba_win &W0 = d.W[wfirst[0]];
            if (tot) atomicAdd(threadIdx.x == 0 ? &W0.nbad : (threadIdx.x == 1 ? &W0.n_left : &W0.n_right), tot);
        }
    } else if (!mixed) {
        if ((threadIdx.x & 63) == 0) {
            ba_win &W0 = d.W[w0];
            if (nb) atomicAdd(&W0.nbad, nb);
            if (nl) atomicAdd(&W0.n_left, nl);
            if (nr) atomicAdd(&W0.n_right, nr);
        }
    } else {
        if (bad) atomicAdd(&W.nbad, 1);
        if (left) atomicAdd(&W.n_left, 1);
        if (right) atomicAdd(&W.n_right, 1);
    }
}

// which blocks still have a live row after the robust pass (thread per landmark block)
__global__ __launch_bounds__(256) void ba_live_kernel(ba_dev d)
{
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= d.n_e) return;
    if (d.W[d.win_of_e[l]].skip) { d.e_dead[l] = 0; return; }
    bool any = false;
    for (int r = d.row_ptr[l]; r < d.row_ptr[l + 1]; ++r) {
        if (d.dead[r]) continue;
        any = true;
        const int fk = d.fk[r], fa = d.fa[r];
        if (fk >= 0) d.f_live[fk] = 1;
        if (fa >= 0) d.f_live[fa] = 1;
    }
    d.e_dead[l] = any ? 0 : 1;
}

// ------------------------------------------------------------------------------------------------------
// Device-side program build (Optimizer::localBA's problem set-up as Ceres sees it: program.cc RemoveFixedBlocks +
// LexicographicallyOrderResidualBlocks).  The flat problem is uploaded once; which blocks are in use, their reduced
// numbering, the row order (landmark block, observing pose block, original index), the per-row records, the landmark
// CSR and the pose -> (row, cell) CSR are all produced by kernels -- the host only learns six integers.  The L2
// re-solve rebuilds from the device-side `active` flags without any host work.  (The host version of this walk took
// 1.8 + 1.35 ms of the 8.9 ms solve at 128 k rows.)

struct ba_raw {   // the B flat problems laid end to end; indices inside a window stay window-local
    const unsigned char *type; const int *pose, *lm; const double *uv, *sigma;   // per residual block (n_res)
    const int *lm_anch; const double *lm_auv;                                     // per landmark (inv-depth only)
    const unsigned char *pose_const;                                              // per pose
    unsigned char *active;                                                        // per residual block, device-owned
    const int *res_off, *lm_off, *pose_off;                                       // B + 1 each: where a window starts
    int B, n_res, n_lm, n_pose, inv_depth;                                        // batch totals
};

// One window of a DEVICE-RESIDENT batch (ov2_ba_solve_batch_dev): where its arrays live and where they go in the
// batch-wide arrays; the per-window outputs are device pointers as well.
struct ba_gsrc {
    const unsigned char *type, *pose_const;
    const int *pose, *lm, *anch;
    const double *uv, *sigma, *auv, *xp, *xl;
    double *out_pose, *out_lm, *out_chi2;
    unsigned char *out_depth, *out_outlier;
    int n_res, n_lm, n_pose, r0, l0, p0;
};

struct ba_gdst {
    unsigned char *type, *pose_const;
    int *pose, *lm, *anch;
    double *uv, *sigma, *auv, *xp, *xl;
    int e, inv_depth;
};

template <typename T>
__device__ __forceinline__ void gcopy(T *__restrict__ dst, const T *__restrict__ src, size_t n, size_t tid, size_t nthr)
{
    for (size_t i = tid; i < n; i += nthr) dst[i] = src[i];
}

// lays the windows' device arrays end to end (what upload_batch's host staging + H2D copy does for host problems);
// grid (G, B): the G workgroups of a window stride over each of its arrays
__global__ __launch_bounds__(256) void bb_gather_kernel(const ba_gsrc *__restrict__ T, ba_gdst D)
{
    const ba_gsrc g = T[blockIdx.y];
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nthr = (size_t)gridDim.x * 256;
    const size_t nr = (size_t)g.n_res, nl = (size_t)g.n_lm, np = (size_t)g.n_pose, r0 = (size_t)g.r0, l0 = (size_t)g.l0, p0 = (size_t)g.p0;
    gcopy(D.type + r0, g.type, nr, tid, nthr);
    gcopy(D.pose + r0, g.pose, nr, tid, nthr);
    gcopy(D.lm + r0, g.lm, nr, tid, nthr);
    gcopy(D.uv + 2 * r0, g.uv, 2 * nr, tid, nthr);
    if (D.sigma) {
        if (g.sigma) gcopy(D.sigma + r0, g.sigma, nr, tid, nthr);
        else for (size_t i = tid; i < nr; i += nthr) D.sigma[r0 + i] = 1.0;
    }
    if (D.inv_depth) {
        gcopy(D.anch + l0, g.anch, nl, tid, nthr);
        gcopy(D.auv + 2 * l0, g.auv, 2 * nl, tid, nthr);
    }
    gcopy(D.xl + (size_t)D.e * l0, g.xl, (size_t)D.e * nl, tid, nthr);
    gcopy(D.pose_const + p0, g.pose_const, np, tid, nthr);
    gcopy(D.xp + 7 * p0, g.xp, 7 * np, tid, nthr);
}

// ... and hands the solved states and the per-residual outputs back to the windows' own device arrays
__global__ __launch_bounds__(256) void bb_scatter_kernel(const ba_gsrc *__restrict__ T, const double *__restrict__ xp,
                                                         const double *__restrict__ xl, int e, const double *__restrict__ chi2,
                                                         const unsigned char *__restrict__ depth,
                                                         const unsigned char *__restrict__ outlier)
{
    const ba_gsrc g = T[blockIdx.y];
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nthr = (size_t)gridDim.x * 256;
    gcopy(g.out_pose, xp + 7 * (size_t)g.p0, 7 * (size_t)g.n_pose, tid, nthr);
    gcopy(g.out_lm, xl + (size_t)e * g.l0, (size_t)e * g.n_lm, tid, nthr);
    if (g.out_chi2) gcopy(g.out_chi2, chi2 + g.r0, (size_t)g.n_res, tid, nthr);
    if (g.out_depth) gcopy(g.out_depth, depth + g.r0, (size_t)g.n_res, tid, nthr);
    if (g.out_outlier) gcopy(g.out_outlier, outlier + g.r0, (size_t)g.n_res, tid, nthr);
}

typedef unsigned long long u64;
enum { BH_ROWS = 0, BH_NE, BH_NF, BH_ERR, BH_ENT, BH_VB, BH_MMAX, BH_STOT, BH_PAIRCAP, BH_ROWSLOT = 12 /* 64 partial row counts */, BH_N = 12 + 64 };
// sort keys of inactive residual blocks carry one bit above the live key bits: they sort behind every live key

__global__ __launch_bounds__(256) void bb_mark_kernel(ba_raw R, const ba_win *__restrict__ W, int *__restrict__ used_lm,
                                                      int *__restrict__ used_pose, u64 *__restrict__ hdr)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= R.n_res || !R.active[i]) return;
    const int w = win_of_index_asc(R.res_off, R.B, i);
    if (W[w].skip) return;
    const int nl = R.lm_off[w + 1] - R.lm_off[w], np = R.pose_off[w + 1] - R.pose_off[w];
    const int t = R.type[i], l = R.lm[i], p = R.pose[i];
    int err = 0;
    if (l < 0 || l >= nl) err = 1;
    else if (t > OV2_BA_RANCH_INV) err = 2;
    else if ((t >= OV2_BA_L_INV) != (R.inv_depth != 0)) err = 3;
    else if (t != OV2_BA_RANCH_INV && (p < 0 || p >= np)) err = 4;
    else if (R.inv_depth && (R.lm_anch[R.lm_off[w] + l] < 0 || R.lm_anch[R.lm_off[w] + l] >= np)) err = 5;
    if (err) { atomicMax(&hdr[BH_ERR], ((u64)err << 56) | ((u64)(unsigned)w << 32) | (u64)(unsigned)(i - R.res_off[w])); return; }
    const int gl = R.lm_off[w] + l, gp = R.pose_off[w] + p;
    used_lm[gl] = 1;
    if (t != OV2_BA_RANCH_INV && !R.pose_const[gp]) used_pose[gp] = 1;
    if (t == OV2_BA_L_INV || t == OV2_BA_R_INV) {
        const int ga = R.pose_off[w] + R.lm_anch[gl];
        if (!R.pose_const[ga]) used_pose[ga] = 1;
    }
}

// reduced numbering from the exclusive scans of the use flags: idx[i] = rank or -1, list[rank] = i
__global__ __launch_bounds__(256) void bb_number_kernel(const int *__restrict__ used, const int *__restrict__ scan, int n,
                                                        int *__restrict__ idx, int *__restrict__ list)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const bool f = used[i] != 0;
    idx[i] = f ? scan[i] : -1;
    if (f) list[scan[i]] = i;
}

// sort key of an active residual block: (landmark block | observing pose block + 1 (0 = anchor-camera residual) | index);
// only the two upper fields are sorted, blocks with the same (landmark, pose) keep their original order (the two cameras
// of one keyframe) and the index rides along
__global__ __launch_bounds__(256) void bb_keys_kernel(ba_raw R, const ba_win *__restrict__ W, const int *__restrict__ eidx,
                                                      const int *__restrict__ fidx, u64 *__restrict__ keys,
                                                      u64 *__restrict__ hdr, int nbits, int fb, int dead_bit)
{
    __shared__ int wsum[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool a = i < R.n_res && R.active[i] != 0;
    u64 k = 1ull << dead_bit;
    if (a) {
        const int w = win_of_index_asc(R.res_off, R.B, i);
        if (W[w].skip) a = false;
        else {
            const int t = R.type[i];
            const int fkey = (t == OV2_BA_RANCH_INV) ? -1 : fidx[R.pose_off[w] + R.pose[i]];
            k = ((u64)(unsigned)eidx[R.lm_off[w] + R.lm[i]] << (fb + nbits)) | ((u64)(unsigned)(fkey + 1) << nbits) | (u64)(unsigned)i;
        }
    }
    if (i < R.n_res) keys[i] = k;
    const u64 m = __ballot(a);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        // one of 64 slots per workgroup: 32 k same-address atomics serialise at ~10 ns each (they, not the key arithmetic,
        // made this kernel 0.39 ms); bb_rows_kernel adds the slots up
        if (tot) atomicAdd(&hdr[BH_ROWSLOT + (blockIdx.x & 63)], (u64)tot);
    }
}

__global__ __launch_bounds__(64) void bb_rows_kernel(u64 *__restrict__ hdr)
{
    u64 v = hdr[BH_ROWSLOT + threadIdx.x];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (threadIdx.x == 0) hdr[BH_ROWS] = v;
}

struct ba_prog_out {   // the writable twins of the const program arrays in ba_dev
    unsigned char *type; int *pose, *lm, *anch, *eb, *fk, *fa; double *uv, *isg; int *rows, *row_win;
    int4 *rowrec;
};

__global__ __launch_bounds__(256) void bb_fill_kernel(ba_raw R, const u64 *__restrict__ keys,
                                                      const int *__restrict__ eidx, const int *__restrict__ fidx,
                                                      const u64 *__restrict__ hdr, ba_prog_out O, int nbits)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= (int)hdr[BH_ROWS]) return;
    const int i = (int)(keys[r] & ((1ull << nbits) - 1ull)), t = R.type[i];
    const int w = win_of_index(R.res_off, R.B, i);
    const int l = R.lm_off[w] + R.lm[i];                                     // batch-wide indices from here on
    const int pk = (t == OV2_BA_RANCH_INV) ? -1 : R.pose_off[w] + R.pose[i];
    const int pa = R.inv_depth ? R.pose_off[w] + R.lm_anch[l] : -1;
    O.rows[r] = i;
    O.row_win[r] = w;
    O.type[r] = (unsigned char)t;
    O.pose[r] = pk < 0 ? R.pose_off[w] : pk;
    O.lm[r] = l;
    O.anch[r] = pa;
    O.eb[r] = eidx[l];
    O.fk[r] = pk < 0 ? -1 : fidx[pk];
    O.fa[r] = (t == OV2_BA_L_INV || t == OV2_BA_R_INV) ? fidx[pa] : -1;
    O.uv[2 * r] = R.uv[2 * i]; O.uv[2 * r + 1] = R.uv[2 * i + 1];
    O.isg[r] = 1.0 / (R.sigma ? R.sigma[i] : 1.0);
    O.rowrec[r] = make_int4(l, pa, pk < 0 ? R.pose_off[w] : pk, eidx[l] | (t << 28));   // (landmark blocks < 2^28: checked with the sort keys)
}

__global__ __launch_bounds__(256) void bb_rowptr_kernel(const int *__restrict__ eb, const u64 *__restrict__ hdr,
                                                        int *__restrict__ row_ptr)
{
    const int r = blockIdx.x * 256 + threadIdx.x, n = (int)hdr[BH_ROWS];
    if (r >= n) return;
    if (r == 0 || eb[r] != eb[r - 1]) row_ptr[eb[r]] = r;   // every landmark block of the reduced program has a row
    if (r == n - 1) row_ptr[(int)hdr[BH_NE]] = n;
}

// pose -> (row, cell) entries: key = (pose block << 32) | (2 * row + cell), dead slots carry the bit above the pose
// field; sorted on the pose field only, so the entries of a pose stay in row order
__global__ __launch_bounds__(256) void bb_posekeys_kernel(const int *__restrict__ fk, const int *__restrict__ fa,
                                                          const u64 *__restrict__ hdr, int n_cap, u64 *__restrict__ pk, int fb)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_cap) return;
    const u64 dead = 1ull << (32 + fb);
    u64 k0 = dead | (u64)(unsigned)(2 * r), k1 = dead | (u64)(unsigned)(2 * r + 1);
    if (r < (int)hdr[BH_ROWS]) {
        if (fk[r] >= 0) k0 = ((u64)(unsigned)fk[r] << 32) | (u64)(unsigned)(2 * r);
        if (fa[r] >= 0) k1 = ((u64)(unsigned)fa[r] << 32) | (u64)(unsigned)(2 * r + 1);
    }
    pk[2 * r] = k0; pk[2 * r + 1] = k1;
}

__global__ __launch_bounds__(256) void bb_poseptr_kernel(const u64 *__restrict__ pks, int n2, u64 *__restrict__ hdr,
                                                         int *__restrict__ pose_ptr, int *__restrict__ pose_ent, int fb,
                                                         const int *__restrict__ eb, int *__restrict__ ent_lm)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n2) return;
    const u64 k = pks[j] >> 32, dead = 1ull << fb;
    if (k == dead) {
        if (j == 0) { hdr[BH_ENT] = 0; pose_ptr[(int)hdr[BH_NF]] = 0; }
        return;
    }
    const int ent = (int)(pks[j] & 0xffffffffull);
    pose_ent[j] = ent;
    ent_lm[j] = eb[ent >> 1];
    if (j == 0 || (pks[j - 1] >> 32) != k) pose_ptr[(int)k] = j;   // every pose block of the reduced program has an entry
    if (j + 1 == n2 || (pks[j + 1] >> 32) == dead) { hdr[BH_ENT] = (u64)(j + 1); pose_ptr[(int)hdr[BH_NF]] = j + 1; }
}

__device__ __forceinline__ int lower_bound_i32(const int *__restrict__ a, int n, int v)
{   // first index with a[i] >= v
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// window of every landmark / pose block of the reduced program
__global__ __launch_bounds__(256) void bb_winfill_kernel(ba_raw R, const u64 *__restrict__ hdr, const int *__restrict__ lm_of_e,
                                                         const int *__restrict__ pose_of_f, int *__restrict__ win_of_e,
                                                         int *__restrict__ win_of_f)
{
    const int k = blockIdx.x * 256 + threadIdx.x, ne = (int)hdr[BH_NE], nf = (int)hdr[BH_NF];
    if (k < ne) win_of_e[k] = win_of_index(R.lm_off, R.B, lm_of_e[k]);
    if (k < nf) win_of_f[k] = win_of_index(R.pose_off, R.B, pose_of_f[k]);
}

// ranges of every window in the reduced program + the prefix sums that depend on them (virtual blocks, S pool).  One
// workgroup; B is at most a few thousand.
__global__ __launch_bounds__(256) void bb_winranges_kernel(ba_raw R, const int *__restrict__ escan, const int *__restrict__ fscan,
                                                           const int *__restrict__ row_win, u64 *__restrict__ hdr,
                                                           ba_win *__restrict__ W, int *__restrict__ vb_start)
{
    const int n_rows = (int)hdr[BH_ROWS];
    for (int w = threadIdx.x; w < R.B; w += 256) {
        ba_win &X = W[w];
        X.e0 = escan[R.lm_off[w]]; X.e1 = escan[R.lm_off[w + 1]];
        X.f0 = fscan[R.pose_off[w]]; X.f1 = fscan[R.pose_off[w + 1]];
        X.row0 = lower_bound_i32(row_win, n_rows, w); X.row1 = lower_bound_i32(row_win, n_rows, w + 1);
        X.m = 6 * (X.f1 - X.f0);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int vb = 0, mmax = 0;
        long long soff = 0, pcap = 0;
        for (int w = 0; w < R.B; ++w) {
            ba_win &X = W[w];
            vb_start[w] = vb;
            X.vb0 = vb; vb += (X.row1 - X.row0 + 255) / 256; X.vb1 = vb;
            X.S_off = soff; soff += (long long)X.m * X.m;
            const long long nf = X.f1 - X.f0;
            pcap += nf * (nf + 1) / 2;
            if (X.m > mmax) mmax = X.m;
        }
        vb_start[R.B] = vb;
        hdr[BH_VB] = (u64)vb; hdr[BH_MMAX] = (u64)mmax; hdr[BH_STOT] = (u64)soff; hdr[BH_PAIRCAP] = (u64)pcap;
    }
}

// ------------------------------------------------------------------------------------------------------
// Schur complement without atomics.  Structure (once per program): the pose CELLS of every landmark (one per run of
// rows of the same free observing pose, plus the anchor pose of an inverse-depth landmark) in CSR form, every unordered
// cell pair of a landmark keyed by its pose pair and sorted (so each off-diagonal 6x6 block of S owns a contiguous
// list of (cell, cell) entries), and the cells of every pose.  Per LM iteration: (1) per landmark: (E'E + D)^-1, E'b and
// per cell W = F'E, F'F, F'Fa, F'(b - E (E'E)^-1 E'b) -> global; (2) one wave per diagonal block gathers its cells,
// one wave per off-diagonal block gathers its entries.  Every block of S has exactly one writer, so S and the rhs are
// bitwise reproducible, and the 10 M f64 atomics per iteration of the previous form (its bound: atomic request rate)
// are gone.

template <int E> struct cell_rec { static constexpr int STRIDE = (E == 1) ? 8 : 22; };   // doubles per cell record (16-byte multiples)

struct ba_cells {
    const int *cell_ptr;      // n_e + 1
    const int *cell_f;        // pose block of the cell
    const int *cell_row;      // first row of its run, -1 = anchor cell (always the last cell of its landmark)
    const int *cell_lm;       // landmark block of the cell
    const int *cell_rank;     // position of the cell in the pose-major order (cells of a pose are contiguous there)
    // per cell, one record at its pose-major position: V = W L (6E doubles) with W = F'E and (E'E + D)^-1 = L L' (a scalar
    // root for inverse depths, a 3 x 3 Cholesky factor for XYZ), then h = L' E'b (E doubles).  The Schur complement is
    // symmetric in V: S[hi, lo] -= V_hi V_lo', rhs -= V h -- one 64-byte (E = 1) / 176-byte (E = 3) record per cell
    // instead of W, W (E'E + D)^-1 and W (E'E + D)^-1 E'b (144 / 336 bytes), and both sides of a pair read the same array.
    double *V;
    const int4 *qrow;         // pose-major position of an observing cell -> (first row, rows) of its run in the sorted rows, landmark block
};

__global__ __launch_bounds__(256) void bs_count_kernel(ba_dev d, int *__restrict__ ncell, int *__restrict__ npair)
{
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l > d.n_e) return;
    int n = 0;
    if (l < d.n_e) {
        int cur = -2, anchor = 0;
        for (int r = d.row_ptr[l]; r < d.row_ptr[l + 1]; ++r) {
            const int fk = d.fk[r];
            if (fk >= 0 && fk != cur) { cur = fk; ++n; }
            if (d.fa[r] >= 0) anchor = 1;
        }
        n += anchor;
    }
    ncell[l] = n;                 // entry n_e = 0, so the exclusive scans end with the totals
    npair[l] = n * (n - 1) / 2;
}

__global__ __launch_bounds__(256) void bs_cells_kernel(ba_dev d, const int *__restrict__ cell_ptr, int *__restrict__ cell_f,
                                                       int *__restrict__ cell_row, int *__restrict__ cell_lm, u64 *__restrict__ ckey)
{
    // eight lanes per landmark, a lane per row: a row opens a cell when its pose block is free and differs from the row
    // before (rows of a landmark are sorted by pose block, constant poses first); the cell's position is the number of such
    // rows before it.  The anchor cell (the free anchor pose of an anchored landmark) comes last.
    const int l = blockIdx.x * 32 + (int)(threadIdx.x >> 3), sub = threadIdx.x & 7, lane = threadIdx.x & 63;
    const bool on = l < d.n_e;
    const int c0 = on ? cell_ptr[l] : 0;
    const int r0 = on ? d.row_ptr[l] : 0, r1 = on ? d.row_ptr[l + 1] : 0;
    const int gshift = lane & ~7;
    int count = 0, fam = -1;
    // every group of the wave makes the same number of trips (the ballots need all lanes): the longest landmark of the wave
    int ntrip = (r1 - r0 + 7) >> 3;
    for (int o = 32; o > 0; o >>= 1) ntrip = max(ntrip, __shfl_xor(ntrip, o));
    for (int t = 0; t < ntrip; ++t) {
        const int r = r0 + 8 * t + sub;
        const bool in = r < r1;
        const int fk = in ? d.fk[r] : -1, fa = in ? d.fa[r] : -1;
        const bool opens = in && fk >= 0 && (r == r0 || d.fk[r - 1] != fk);
        const unsigned gb = (unsigned)((__ballot(opens) >> gshift) & 0xffull), ga = (unsigned)((__ballot(fa >= 0) >> gshift) & 0xffull);
        if (opens) {
            const int c = c0 + count + __popc(gb & ((1u << sub) - 1u));
            cell_f[c] = fk; cell_row[c] = r; cell_lm[c] = l;
            ckey[c] = ((u64)(unsigned)fk << 32) | (u64)(unsigned)c;
        }
        count += __popc(gb);
        if (ga) { const int v = __shfl(fa, gshift + (__ffs(ga) - 1)); fam = v; }
    }
    if (on && sub == 0 && fam >= 0) {
        const int c = c0 + count;
        cell_f[c] = fam; cell_row[c] = -1; cell_lm[c] = l;
        ckey[c] = ((u64)(unsigned)fam << 32) | (u64)(unsigned)c;
    }
}

// the pair entries of every landmark (cell p, cell q > p), eight lanes per landmark: the entries of a landmark are
// contiguous (pair_off), lane j writes entries j, j + 8, ...  (One thread per landmark wrote 64 scattered words per store
// instruction: 1.3 GB of write traffic for 0.2 GB of entries, 507 us per batch.)
__global__ __launch_bounds__(256) void bs_pairs_kernel(ba_dev d, const int *__restrict__ cell_ptr, const int *__restrict__ pair_off,
                                                       const int *__restrict__ cell_f, u64 *__restrict__ pkey, int *__restrict__ pent,
                                                       int fb, int pb)
{
    const int l = blockIdx.x * 32 + (int)(threadIdx.x >> 3), sub = threadIdx.x & 7;
    if (l >= d.n_e) return;
    const int c0 = cell_ptr[l], nc = cell_ptr[l + 1] - c0;
    const int np = nc * (nc - 1) / 2, o0 = pair_off[l];
    int p = 0, first = 0;   // entries first .. first + (nc - 1 - p) belong to cell p
    for (int idx = sub; idx < np; idx += 8) {
        while (idx >= first + (nc - 1 - p)) { first += nc - 1 - p; ++p; }
        const int q = p + 1 + (idx - first);
        const int fp = cell_f[c0 + p], fq = cell_f[c0 + q];
        const int hi = fp >= fq ? p : q, lo = fp >= fq ? q : p;   // cell of the larger pose block first
        const int fh = fp >= fq ? fp : fq, fl = fp >= fq ? fq : fp;
        const int o = o0 + idx;
        // key = (pose hi | pose lo | entry index), sorted on the pose pair: a pair's entries stay in landmark order;
        // the entry's two cells are looked up through its index
        pkey[o] = ((((u64)(unsigned)fh << fb) | (u64)(unsigned)fl) << pb) | (u64)(unsigned)o;
        reinterpret_cast<int2 *>(pent)[o] = make_int2(c0 + hi, c0 + lo);
    }
}

// boundaries of the pose-sorted cell list -> pcell_ptr (n_f + 1) and the cell ids
__global__ __launch_bounds__(256) void bs_posecells_kernel(const u64 *__restrict__ ck, int n, int n_f,
                                                           int *__restrict__ pcell_ptr, int *__restrict__ pcell_ent)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int f = (int)(ck[j] >> 32);
    pcell_ent[j] = (int)(ck[j] & 0xffffffffull);
    if (j == 0 || (int)(ck[j - 1] >> 32) != f) pcell_ptr[f] = j;
    if (j == n - 1) pcell_ptr[n_f] = n;
}

// segment heads of the sorted pair list (a new pose pair starts) ...
__global__ __launch_bounds__(256) void bs_heads_kernel(const u64 *__restrict__ pk, int n, int *__restrict__ head, int pb)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j > n) return;
    head[j] = (j < n && (j == 0 || (pk[j] >> pb) != (pk[j - 1] >> pb))) ? 1 : 0;   // entry n = 0: its rank is the pair count
}

// ... and, from their exclusive ranks, the start of every pair's segment (+ the end of the last one)
__global__ __launch_bounds__(256) void bs_segs_kernel(const u64 *__restrict__ pk, int n, const int *__restrict__ head,
                                                      const int *__restrict__ rank, int *__restrict__ seg_start,
                                                      u64 *__restrict__ pair_key, int pb)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    if (head[j]) { seg_start[rank[j]] = j; pair_key[rank[j]] = pk[j] >> pb; }
    if (j == n - 1) seg_start[rank[n]] = n;
}

// position of a cell in the pose-major order (inverse of the pose-sorted cell list)
__global__ __launch_bounds__(256) void bs_rank_kernel(const int *__restrict__ pcell_ent, int n, int *__restrict__ cell_rank)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < n) cell_rank[pcell_ent[j]] = j;
}

// observing cell -> its run of rows, stored at the cell's pose-major position (what the gather walks to form F'Fa)
__global__ __launch_bounds__(256) void bs_qrow_kernel(ba_dev d, const int *__restrict__ cell_f, const int *__restrict__ cell_row,
                                                      const int *__restrict__ cell_lm, const int *__restrict__ cell_rank, int n,
                                                      int4 *__restrict__ qrow)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    const int r0 = cell_row[c];
    int cnt = 0;
    if (r0 >= 0) {
        const int r1 = d.row_ptr[cell_lm[c] + 1], fk = cell_f[c];
        while (r0 + cnt < r1 && d.fk[r0 + cnt] == fk) ++cnt;
    }
    qrow[cell_rank[c]] = make_int4(r0, cnt, cell_lm[c], 0);
}

// pair entries: cell ids -> pose-major positions, bit 31 = "this cell is the landmark's anchor cell"
__global__ __launch_bounds__(256) void bs_pent_kernel(int *__restrict__ pent, size_t n2, const int *__restrict__ cell_rank,
                                                      const int *__restrict__ cell_row)
{
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n2) return;
    const int c = pent[k];
    pent[k] = cell_rank[c] | (cell_row[c] < 0 ? (int)0x80000000u : 0);
}

// the entries in the order of the sorted pair list (what the gather reads, coalesced): sorted position -> (hi, lo) cells
__global__ __launch_bounds__(256) void bs_pent_sorted_kernel(const u64 *__restrict__ pk, int n, int pb, const int *__restrict__ pent,
                                                             int2 *__restrict__ out)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    const size_t ent = (size_t)(pk[q] & ((1ull << pb) - 1ull));
    out[q] = make_int2(pent[2 * ent], pent[2 * ent + 1]);
}

// (1) per landmark, 16 lanes: (E'E + D)^-1, (E'E + D)^-1 E'b, and per pose cell W = F'E, W (E'E + D)^-1, W (E'E + D)^-1 E'b
// (+ F'Fa for the cells that observe an anchored landmark), written at the cell's pose-major position so that the
// gathers of (2) read contiguous memory.  F'F and F'b are not formed here: they only change with the jacobian
// (ba_pose_normal_kernel).
template <int E, int GW>
__global__ __launch_bounds__(256) void bs_landmark_kernel(ba_dev d, ba_cells C)
{
    BA_WAVE_PRIO();
    const int grp = threadIdx.x / GW, sub = threadIdx.x % GW;
    const int l = blockIdx.x * (256 / GW) + grp;
    const bool live = l < d.n_e && d.W[d.win_of_e[l]].active;
    const int r0 = live ? d.row_ptr[l] : 0, r1 = live ? d.row_ptr[l + 1] : 0;
    // the cell list of the landmark is fetched WITH its rows, and the first cell of every lane is walked before the
    // reductions below: the kernel is a chain of dependent loads per 16-lane group (rows -> sums -> cell list -> cell rows ->
    // cell position -> store); taking the cell side out of that chain shortens it by two round trips (332 -> 311 us per launch).
    // Every value is computed by the same expressions as before.
    const int c0 = live ? C.cell_ptr[l] : 0, nc = live ? C.cell_ptr[l + 1] - c0 : 0;
    const double *__restrict__ sf = d.scale + (size_t)d.n_e * E;   // Jacobi scales of the pose columns
    double ete[E * E], g[E], se[E], wp[3];
    for (int i = 0; i < E * E; ++i) ete[i] = 0.0;
    for (int i = 0; i < E; ++i) { g[i] = 0.0; se[i] = live ? d.scale[(size_t)l * E + i] : 1.0; }
    wp[0] = wp[1] = wp[2] = 0.0;
    if (live) load_wpt(d, (size_t)l, wp);
    const bool has_anchor = nc > 0 && C.cell_row[c0 + nc - 1] < 0;
    const int nobs = has_anchor ? nc - 1 : nc;
    // W = F'E of one cell: the lane walks the (one or two) rows of the cell's run once
    auto cell_W = [&](int c, double *Wk) {
        const int fk = C.cell_f[c0 + c];
        double sk[6];
#pragma unroll
        for (int i = 0; i < 6 * E; ++i) Wk[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) sk[i] = sf[fk * 6 + i];
        for (int r2 = C.cell_row[c0 + c]; r2 < r1 && d.fk[r2] == fk; ++r2) {
            double J2[12], Je2[2 * E];
            load_U(d, (size_t)r2, wp, J2);
            load_d2<2 * E>(d.Je + (size_t)r2 * 2 * E, Je2);
#pragma unroll
            for (int i = 0; i < 6; ++i) { J2[i] *= sk[i]; J2[6 + i] *= sk[i]; }
#pragma unroll
            for (int k = 0; k < E; ++k) { Je2[k] *= se[k]; Je2[E + k] *= se[k]; }
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int k = 0; k < E; ++k) Wk[i * E + k] += J2[i] * Je2[k] + J2[6 + i] * Je2[E + k];
        }
    };
    double W0[6 * E];
    const bool own0 = sub < nobs;
    int rank0 = 0;
    if (own0) { rank0 = C.cell_rank[c0 + sub]; cell_W(sub, W0); }
    // one sweep over the rows for E'E, E'b AND the anchor cell's W (every row of an anchored landmark carries the anchor
    // block -U): the rows were read a second time for it before
    const int fa_cell = has_anchor ? C.cell_f[c0 + nc - 1] : 0;
    double Wa[6 * E], sa[6];
    for (int i = 0; i < 6 * E; ++i) Wa[i] = 0.0;
    for (int i = 0; i < 6; ++i) sa[i] = has_anchor ? sf[fa_cell * 6 + i] : 0.0;
    for (int base = r0; base < r1; base += GW) {
        const int r = base + sub;
        if (r < r1) {
            double Je[2 * E];
            load_d2<2 * E>(d.Je + (size_t)r * 2 * E, Je);
            for (int i = 0; i < E; ++i) { Je[i] *= se[i]; Je[E + i] *= se[i]; }
            const double b0 = d.res[2 * r], b1 = d.res[2 * r + 1];
            for (int i = 0; i < E; ++i) {
                for (int j = 0; j < E; ++j) ete[i * E + j] += Je[i] * Je[j] + Je[E + i] * Je[E + j];
                g[i] += Je[i] * b0 + Je[E + i] * b1;
            }
            if (has_anchor && d.fa[r] >= 0) {
                double Ja[12];
                load_U(d, (size_t)r, wp, Ja);
                for (int i = 0; i < 6; ++i) { Ja[i] = -(Ja[i] * sa[i]); Ja[6 + i] = -(Ja[6 + i] * sa[i]); }
                for (int i = 0; i < 6; ++i)
                    for (int k = 0; k < E; ++k) Wa[i * E + k] += Ja[i] * Je[k] + Ja[6 + i] * Je[E + k];
            }
        }
    }
    for (int i = 0; i < E * E; ++i) ete[i] = grp_sum<GW>(ete[i]);
    for (int i = 0; i < E; ++i) g[i] = grp_sum<GW>(g[i]);
    if (live) for (int i = 0; i < E; ++i) { const double dv = d.lmd[l * E + i]; ete[i * E + i] += dv * dv; }
    else for (int i = 0; i < E; ++i) ete[i * E + i] = 1.0;
    double ie[E * E], ieg[E];
    invert_ete<E>(ete, ie);
    for (int i = 0; i < E; ++i) {
        double sacc = 0;
        for (int j = 0; j < E; ++j) sacc += ie[i * E + j] * g[j];
        ieg[i] = sacc;
    }
    if (live && sub == 0) {
        for (int i = 0; i < E * E; ++i) d.iete[(size_t)l * E * E + i] = ie[i];
        for (int i = 0; i < E; ++i) d.ieg[(size_t)l * E + i] = ieg[i];
    }
    if (!live) return;
    // L with (E'E + D)^-1 = L L' and h = L' E'b (the same for every cell of the landmark)
    double Lf[E * E], hv[E];
    if (E == 1) {
        Lf[0] = sqrt(ie[0]);
        hv[0] = Lf[0] * g[0];
    } else {
        const double l00 = sqrt(ie[0]), l10 = ie[3] / l00, l20 = ie[6] / l00;
        const double l11 = sqrt(ie[4] - l10 * l10), l21 = (ie[7] - l20 * l10) / l11;
        const double l22 = sqrt(ie[8] - l20 * l20 - l21 * l21);
        Lf[0] = l00; Lf[1] = 0.0; Lf[2] = 0.0; Lf[3] = l10; Lf[4] = l11; Lf[5] = 0.0; Lf[6] = l20; Lf[7] = l21; Lf[8] = l22;
        for (int cc = 0; cc < E; ++cc) {
            double sacc = 0.0;
            for (int k = cc; k < E; ++k) sacc += Lf[k * E + cc] * g[k];
            hv[cc] = sacc;
        }
    }
    // writes the record (V = W L | h) of one cell at its pose-major position
    auto store_cell = [&](int rank, const double *Wk) {
        constexpr int CS = cell_rec<E>::STRIDE;
        double rec[CS];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int cc = 0; cc < E; ++cc) {
                double w = 0.0;
#pragma unroll
                for (int k = cc; k < E; ++k) w += Wk[i * E + k] * Lf[k * E + cc];
                rec[i * E + cc] = w;
            }
#pragma unroll
        for (int cc = 0; cc < E; ++cc) rec[6 * E + cc] = hv[cc];
#pragma unroll
        for (int i = 7 * E; i < CS; ++i) rec[i] = 0.0;
        double2 *dst = reinterpret_cast<double2 *>(C.V + (size_t)rank * CS);
#pragma unroll
        for (int i = 0; i < CS / 2; ++i) dst[i] = make_double2(rec[2 * i], rec[2 * i + 1]);
    };
    // observing cells: ONE LANE PER CELL.  (F'Fa, the coupling of an observing pose with the landmark's anchor pose, is not
    // materialised: the gather forms it from the same rows.)
    if (own0) store_cell(rank0, W0);
    for (int c = sub + GW; c < nobs; c += GW) {   // landmarks with more observing cells than the group has lanes
        double Wk[6 * E];
        cell_W(c, Wk);
        store_cell(C.cell_rank[c0 + c], Wk);
    }
    if (has_anchor) {   // W of the anchor cell: summed over all rows by the group (anchor block of a row = -U)
        for (int i = 0; i < 6 * E; ++i) Wa[i] = grp_sum<GW>(Wa[i]);
        if (sub == 0) store_cell(C.cell_rank[c0 + nc - 1], Wa);
    }
}

__device__ __forceinline__ double wave_total(double v)
{
    v = row_sum(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// (2a) one workgroup per pose block: diagonal block of S (lower triangle) = D_f^2 + F'F - sum_cells W (E'E + D)^-1 W' and
// its rhs segment = F'b - sum_cells W (E'E + D)^-1 E'b.  The pose's cells are one contiguous run of the pose-major arrays;
// threads stride over them, every thread keeps the 27 partial sums, fixed-order wave + LDS reduction at the end.
template <int E>
__device__ __forceinline__ void bs_diag_block(const ba_dev &d, const ba_cells &C, const int *__restrict__ pcell_ptr, int f,
                                              double (*red)[27])
{
    const int tid = threadIdx.x;
    const ba_win &Wn = d.W[d.win_of_f[f]];
    if (!Wn.active) return;   // workgroup-uniform
    double acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = 0.0;
    for (int q = pcell_ptr[f] + tid; q < pcell_ptr[f + 1]; q += 256) {
        constexpr int CS = cell_rec<E>::STRIDE;
        double v[CS];
        load_d2<CS>(C.V + (size_t)q * CS, v);
#pragma unroll
        for (int t = 0; t < 21; ++t) {
            const int i = c_tri_i[t], j = c_tri_j[t];
            double p = 0.0;
#pragma unroll
            for (int cc = 0; cc < E; ++cc) p += v[i * E + cc] * v[j * E + cc];
            acc[t] -= p;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            double p = 0.0;
#pragma unroll
            for (int cc = 0; cc < E; ++cc) p += v[i * E + cc] * v[6 * E + cc];
            acc[21 + i] -= p;
        }
    }
#pragma unroll
    for (int t = 0; t < 27; ++t) {
        const double w = wave_total(acc[t]);
        if ((tid & 63) == 0) red[tid >> 6][t] = w;
    }
    __syncthreads();
    if (tid < 27) {
        const double v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        const int lf = f - Wn.f0;   // block index inside the window's own S
        if (tid >= 21) d.rhs[f * 6 + tid - 21] = d.grad[d.n_e * d.e + f * 6 + tid - 21] + v;   // its only writer
        else d.Spool[Wn.S_off + (size_t)(lf * 6 + c_tri_j[tid]) * Wn.m + lf * 6 + c_tri_i[tid]] += d.FFp[(size_t)f * 21 + tid] + v;   // on top of the LM diagonal of ba_sinit
    }
}

// (2b) one wave per pose pair (hi > lo): S[hi, lo] = sum over the landmarks seen by both of  -W_hi (E'E + D)^-1 W_lo^T,
// plus F'Fa when one of the two cells is the landmark's anchor cell.  Lanes stride over the pair's entries; an entry
// names the two cells by their pose-major positions, which grow with the landmark inside both poses' runs, so
// consecutive entries read nearly consecutive records.
// (Measured alternative, round 3: lane = 16 g + s with g = one 3 x 3 quarter of the block and s = one of 16 entry slots -- nine
// accumulators and 116 VGPRs instead of 36 and 170, nine DPP row sums instead of 36 cross-wave totals.  705 us per launch
// against 549 for this form on 64 distinct 50-keyframe windows, 732 / 888 us with two / four entries per lane in flight:
// every entry is then fetched by four lanes, and the kernel is bound by the number of scattered (lane, load) addresses the
// texture path resolves, not by VALU work, registers or latency.)
template <int E>
__device__ __forceinline__ void bs_pair_block(const ba_dev &d, const ba_cells &C, int n_pairs,
                                              const u64 *__restrict__ pair_key, const int *__restrict__ seg_start,
                                              const int2 *__restrict__ pent, int fb, int pidx, int dbg = 0)
{
    const int lane = threadIdx.x & 63;
    if (pidx >= n_pairs) return;
    const int hi = (int)(pair_key[pidx] >> fb), lo = (int)(pair_key[pidx] & ((1ull << fb) - 1ull));
    const ba_win &Wn = d.W[d.win_of_f[hi]];   // both pose blocks of a pair belong to one window
    if (!Wn.active) return;
    double acc[36];   // element (i of hi, j of lo) at i + 6 j
#pragma unroll
    for (int t = 0; t < 36; ++t) acc[t] = 0.0;
    const int s0 = seg_start[pidx], s1 = seg_start[pidx + 1];
    const double *__restrict__ sf = d.scale + (size_t)d.n_e * E;
    double s_hi[6], s_lo[6];   // Jacobi scales of the two poses' columns (wave-uniform)
#pragma unroll
    for (int i = 0; i < 6; ++i) { s_hi[i] = sf[hi * 6 + i]; s_lo[i] = sf[lo * 6 + i]; }
    for (int q0 = s0; q0 < s1; q0 += 64) {
        const int q = q0 + lane;
        const bool in = q < s1;
        const int2 pe = in ? pent[q] : make_int2(0, 0);
        const int eh = pe.x, el = pe.y;
        if (in) {
            const size_t qh = (size_t)(eh & 0x7fffffff), ql = (size_t)(el & 0x7fffffff);
            double T[6 * E], Wl[6 * E];
            {   // the V part of the two records (16-byte loads)
                constexpr int CS = cell_rec<E>::STRIDE;
                const double2 *th = reinterpret_cast<const double2 *>(C.V + qh * CS), *wl = reinterpret_cast<const double2 *>(C.V + ql * CS);
#pragma unroll
                for (int i = 0; i < 3 * E; ++i) { const double2 a = th[i], b = wl[i]; T[2 * i] = a.x; T[2 * i + 1] = a.y; Wl[2 * i] = b.x; Wl[2 * i + 1] = b.y; }
            }
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    double w = 0.0;
#pragma unroll
                    for (int cc = 0; cc < E; ++cc) w += T[i * E + cc] * Wl[j * E + cc];
                    acc[i + 6 * j] -= w;
                }
            // an entry in which one of the two cells is the landmark's anchor cell adds F'Fa = sum over the observing cell's
            // rows of (its pose block)' (the anchor's block); the anchor block of a row is -U, so whichever of hi / lo
            // observes, element (i of hi, j of lo) is -((U s_hi)_i (U s_lo)_j) summed over the two residual components.
            // Formed here from the one or two 48-byte rows of the run, lane-parallel like the rest of the entry (nothing
            // materialised per cell: the 288-byte F'Fa records were the largest store of the whole Schur complement).
            if (!(dbg & 4) && (eh < 0 || el < 0)) {
                const int4 rr = C.qrow[el < 0 ? qh : ql];
                double wp[3];   // the rows of a cell belong to one landmark
                load_wpt(d, (size_t)rr.z, wp);
                for (int k = 0; k < rr.y; ++k) {
                    double Uu[12], H[12], L[12];
                    load_U(d, (size_t)(rr.x + k), wp, Uu);
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        H[i] = Uu[i] * s_hi[i]; H[6 + i] = Uu[6 + i] * s_hi[i];
                        L[i] = Uu[i] * s_lo[i]; L[6 + i] = Uu[6 + i] * s_lo[i];
                    }
#pragma unroll
                    for (int j = 0; j < 6; ++j)
#pragma unroll
                        for (int i = 0; i < 6; ++i) acc[i + 6 * j] -= H[i] * L[j] + H[6 + i] * L[6 + j];
                }
            }
        }
    }
    // the 36 totals are wave-uniform: lane t keeps total t, then 36 lanes write the block with one store.  An
    // off-diagonal block has this wave as its only writer and was zeroed by ba_sinit: plain stores (a serial
    // read-modify-write of 36 elements by one lane made these short waves latency-bound)
    double mine = 0.0;
    if (dbg & 8) {
#pragma unroll
        for (int t = 0; t < 36; ++t) mine += acc[t];
    } else {
#pragma unroll
    for (int t = 0; t < 36; ++t) {
        const double v = wave_total(acc[t]);
        if (lane == t) mine = v;
    }
    }
    if (lane < 36) {
        const int i = lane % 6, j = lane / 6;
        double *dst = d.Spool + Wn.S_off + (size_t)((lo - Wn.f0) * 6 + j) * Wn.m + (hi - Wn.f0) * 6 + i;
        if (hi != lo) *dst = mine;
        else if (i >= j) *dst += mine;
    }
}

// (2) ONE launch for all of S: workgroups [0, n_f) gather the diagonal blocks (+ rhs), the following ones take four
// pose pairs each (one per wave).  One launch instead of two matters beside a busy front-end, where every dispatch of
// the BA stream queues behind resident waves.
template <int E>
__global__ __launch_bounds__(256) void bs_gather_kernel(ba_dev d, ba_cells C, const int *__restrict__ pcell_ptr,
                                                        const int *__restrict__ n_pairs, const u64 *__restrict__ pair_key,
                                                        const int *__restrict__ seg_start, const int2 *__restrict__ pent, int fb,
                                                        int dbg_only)
{
    BA_WAVE_PRIO();
    __shared__ double red[4][27];
    if ((dbg_only & 3) && (((dbg_only & 3) == 1) != ((int)blockIdx.x < d.n_f))) return;   // timing experiments: 1 = diagonal part only, 2 = pairs only
    if ((int)blockIdx.x < d.n_f) bs_diag_block<E>(d, C, pcell_ptr, blockIdx.x, red);
    else {
        // Consecutive pairs share their `hi` pose, i.e. they re-read the same run of cell records; workgroups are dealt
        // round-robin to the eight XCDs, so the pair blocks are renumbered to give every XCD a contiguous range of pairs
        // (its L2 then holds the runs its pairs share).  The host rounds the grid's pair blocks up to a multiple of 8.
        const int np = *n_pairs, pb = (int)blockIdx.x - d.n_f;
        const int npb = (((np + 3) >> 2) + 7) & ~7;   // pair blocks that have work (the grid is sized for an upper bound), a multiple of 8
        if (pb >= npb) return;
        const int blk = (pb & 7) * (npb >> 3) + (pb >> 3);
        bs_pair_block<E>(d, C, np, pair_key, seg_start, pent, fb, blk * 4 + (int)(threadIdx.x >> 6), dbg_only);
    }
}

// ------------------------------------------------------------------------------------------------------
// host side

struct dev_buf {
    void *p = nullptr;
    size_t cap = 0;
};

struct ba_workspace {
    std::vector<dev_buf> bufs;
    size_t next = 0;
};

// device memory of a solve is carved from one arena owned by the ctx and kept across solves (no hipMalloc/hipFree on
// the keyframe path); ov2_ba_solve sizes it from an upper bound before the first carve.
template <typename T>
ov2_status dalloc(ov2_ctx *c, size_t &off, T **out, size_t n)
{
    const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) / 256 * 256;
    if (off + bytes > c->ba_arena_cap)
        return ov2_set_err(c, OV2_ERR_NOMEM, "BA arena exhausted (%zu + %zu > %zu)", off, bytes, c->ba_arena_cap);
    *out = (T *)((char *)c->ba_arena + off);
    off += bytes;
    return OV2_OK;
}

template <typename T>
ov2_status dalloc2(ov2_ctx *c, size_t &off, T **out, size_t n)
{
    const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) / 256 * 256;
    if (off + bytes > c->ba_arena2_cap)
        return ov2_set_err(c, OV2_ERR_NOMEM, "BA structure arena exhausted (%zu + %zu > %zu)", off, bytes, c->ba_arena2_cap);
    *out = (T *)((char *)c->ba_arena2 + off);
    off += bytes;
    return OV2_OK;
}

template <typename T>
ov2_status dupload(ov2_ctx *c, size_t &off, const T **out, const std::vector<T> &v)
{
    T *p = nullptr;
    ov2_status s = dalloc(c, off, &p, v.size());
    if (s != OV2_OK) return s;
    if (!v.empty()) OV2_HIP(c, hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
    *out = p;
    return OV2_OK;
}

// upload path: host arrays are built in a PINNED mirror of the arena at the same offsets and sent with one
// hipMemcpyAsync.  (Measured: the same data in pageable std::vectors, ~15 copies of 0.5-2 MB, took 10-30 ms to arrive
// on this runtime; the pinned single copy takes ~0.3 ms.)
template <typename T>
ov2_status hcarve(ov2_ctx *c, size_t &off, T **host, const T **dev, size_t n)
{
    const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) / 256 * 256;
    if (off + bytes > c->ba_arena_cap || off + bytes > c->ba_host_cap)
        return ov2_set_err(c, OV2_ERR_NOMEM, "BA arena exhausted (%zu + %zu > %zu)", off, bytes, c->ba_arena_cap);
    *host = (T *)((char *)c->ba_host + off);
    *dev = (const T *)((char *)c->ba_arena + off);
    off += bytes;
    return OV2_OK;
}

struct ba_solver {
    ov2_ctx *c;
    int B;
    const ov2_ba_problem *P;     // B problems
    const ov2_ba_options *o;
    size_t arena_off = 0;
    ba_dev d;
    ba_raw raw;                  // the flat problems on the device (uploaded once per batch) + the active flags
    size_t persist_end = 0;      // arena offset behind everything that lives for the whole batch: programs are (re)built from here
    std::vector<int> res_off, lm_off, pose_off;   // host prefix sums (B + 1)
    ba_win *W = nullptr;         // device window records
    ba_wconst *wc = nullptr;
    std::vector<ba_win> hW;      // host mirror of the window records (pageable; staged through the pinned block)
    const int *rows = nullptr;   // device: sorted row -> batch-wide residual index
    // atomic-free Schur complement: cells, pose -> cells, pose pairs -> (cell, cell) entries (see the bs_* kernels)
    ba_cells cells;
    const int *pcell_ptr = nullptr, *pcell_ent = nullptr;
    const int *n_pairs = nullptr, *seg_start = nullptr;
    const u64 *pair_key = nullptr, *pair_val = nullptr;   // pair_val: the sorted (pose pair | entry) keys
    const int2 *pair_ent = nullptr;                       // sorted entry -> its two cells (pose-major positions, bit 31 = anchor cell)
    int fb = 1, pb = 1;                                   // bit widths of a pose block id / of a pair entry index in the packed keys
    long long pair_cap = 0;
    double *xp = nullptr, *xl = nullptr, *cp = nullptr, *cl = nullptr;  // device states (batch-wide)
    double *chold = nullptr;      // diagonal blocks of the Cholesky factors (multi-workgroup path), one slab per window
    size_t chold_stride = 0;
    double *chi2_dev = nullptr;
    unsigned char *depth_dev = nullptr, *outlier_dev = nullptr;
    int vblocks = 0, mmax = 0, cover_max = 0;
    const int *pose_ptr = nullptr, *pose_ent = nullptr;   // pose -> (row*2 + cell) CSR
    int n_res = 0, n_lm = 0, n_pose = 0, e = 1;           // batch totals
    bool any_sigma = false;
    bool on_device = false;      // ov2_ba_solve_batch_dev: the problems' arrays and the per-residual outputs are device memory
    const ov2_ba_result *Rdev = nullptr;   // ... the results (for their output pointers)
    const ba_gsrc *gtab = nullptr;         // ... device table of the windows' arrays
    bool want_chi2 = false, want_depth = false, want_out = false;   // some window of the batch asked for the per-residual output
    bool meas_pending = false;   // the measurement upload is still travelling on the copy stream (first program build waits for it)
};

#define BA_LAUNCH(S, id, ...) OV2_LAUNCH((S).c, id, __VA_ARGS__)

// Lays the B flat problems end to end in the PINNED mirror of the arena head and sends them with one copy (measured in
// round 1: the same data in pageable vectors, ~15 copies of 0.5-2 MB, took 10-30 ms to arrive; the pinned single copy
// ~0.3 ms for one window).  Indices stay window-local; the kernels add the window offsets.
ov2_status upload_batch(ba_solver &S)
{
    ov2_ctx *c = S.c;
    const int B = S.B;
    const size_t n = (size_t)S.n_res, L = (size_t)S.n_lm, NP = (size_t)S.n_pose, e = (size_t)S.e;
    ov2_status s;
    S.arena_off = 0;
    unsigned char *h_type, *h_pc; int *h_pose, *h_lm, *h_anch, *h_ro, *h_lo, *h_po; double *h_uv, *h_sig, *h_auv, *h_xp, *h_xl;
    ba_win *h_W; ba_wconst *h_wc;
    ba_raw &R = S.raw;
    memset(&R, 0, sizeof(R));
    R.B = B; R.n_res = S.n_res; R.n_lm = S.n_lm; R.n_pose = S.n_pose; R.inv_depth = S.e == 1 ? 1 : 0;
    const double *d_xp, *d_xl; const ba_win *d_W; const ba_wconst *d_wc;
#define HC(devp, hostp, count) if ((s = hcarve(c, S.arena_off, &hostp, &devp, (size_t)(count))) != OV2_OK) return s
    // the measurements (two thirds of the bytes) come last: they travel on a second stream while the program build
    // already numbers and sorts the residual blocks, which only needs the index arrays of the head
    HC(R.type, h_type, n); HC(R.pose, h_pose, n); HC(R.lm, h_lm, n);
    HC(R.lm_anch, h_anch, L); HC(R.lm_auv, h_auv, 2 * L); HC(R.pose_const, h_pc, NP);
    HC(R.res_off, h_ro, B + 1); HC(R.lm_off, h_lo, B + 1); HC(R.pose_off, h_po, B + 1);
    HC(d_xp, h_xp, 7 * NP); HC(d_xl, h_xl, e * L); HC(d_W, h_W, B); HC(d_wc, h_wc, B);
    const size_t head_end = S.arena_off;
    HC(R.uv, h_uv, 2 * n);
    if (S.any_sigma) { HC(R.sigma, h_sig, n); } else { R.sigma = nullptr; h_sig = nullptr; }
    ba_gsrc *h_tab = nullptr; const ba_gsrc *d_tab = nullptr;
    if (S.on_device) HC(d_tab, h_tab, B);
#undef HC
    if (S.on_device) {
        // the arrays stay where they are: only the window table, the offsets and the window records cross the link, one
        // kernel lays the windows end to end
        for (int w = 0; w < B; ++w) {
            const ov2_ba_problem &P = S.P[w];
            ba_gsrc &g = h_tab[w];
            g.type = P.res_type; g.pose_const = P.pose_const; g.pose = P.res_pose; g.lm = P.res_lm; g.anch = P.lm_anchor_pose;
            g.uv = P.res_uv; g.sigma = P.res_sigma; g.auv = P.lm_anchor_uv; g.xp = P.pose; g.xl = P.lm;
            g.out_pose = P.pose; g.out_lm = P.lm;
            g.out_chi2 = S.Rdev[w].chi2; g.out_depth = S.Rdev[w].depth_positive; g.out_outlier = S.Rdev[w].outlier;
            g.n_res = P.n_res; g.n_lm = P.n_lm; g.n_pose = P.n_pose;
            g.r0 = S.res_off[w]; g.l0 = S.lm_off[w]; g.p0 = S.pose_off[w];
            ba_wconst &K = h_wc[w];
            for (int i = 0; i < 4; ++i) { K.Kl[i] = P.calib_l[i]; K.Kr[i] = P.calib_r[i]; }
            pose_Rt(P.T_rl, K.Rrl, K.trl);
        }
        memcpy(h_ro, S.res_off.data(), sizeof(int) * (B + 1));
        memcpy(h_lo, S.lm_off.data(), sizeof(int) * (B + 1));
        memcpy(h_po, S.pose_off.data(), sizeof(int) * (B + 1));
        memcpy(h_W, S.hW.data(), sizeof(ba_win) * B);
        auto h2d = [&](const void *dev, const void *host, size_t bytes) {
            return hipMemcpyAsync(const_cast<void *>(dev), host, bytes, hipMemcpyHostToDevice, c->stream);
        };
        OV2_HIP(c, h2d(R.res_off, h_ro, (size_t)((const char *)d_xp - (const char *)R.res_off)));           // the three offset arrays
        OV2_HIP(c, h2d(d_W, h_W, head_end - (size_t)((const char *)d_W - (const char *)c->ba_arena)));     // window records + constants
        OV2_HIP(c, h2d(d_tab, h_tab, sizeof(ba_gsrc) * B));
        ba_gdst D;
        D.type = const_cast<unsigned char *>(R.type); D.pose_const = const_cast<unsigned char *>(R.pose_const);
        D.pose = const_cast<int *>(R.pose); D.lm = const_cast<int *>(R.lm); D.anch = const_cast<int *>(R.lm_anch);
        D.uv = const_cast<double *>(R.uv); D.sigma = const_cast<double *>(R.sigma); D.auv = const_cast<double *>(R.lm_auv);
        D.xp = const_cast<double *>(d_xp); D.xl = const_cast<double *>(d_xl); D.e = S.e; D.inv_depth = R.inv_depth;
        BA_LAUNCH(S, K_MISC, bb_gather_kernel, dim3(64, B), dim3(256), 0, c->stream, d_tab, D);
        S.gtab = d_tab;
    } else {
    auto stage_window = [&](int w) {
        const ov2_ba_problem &P = S.P[w];
        const size_t r0 = (size_t)S.res_off[w], l0 = (size_t)S.lm_off[w], p0 = (size_t)S.pose_off[w];
        const size_t nr = (size_t)P.n_res, nl = (size_t)P.n_lm, np = (size_t)P.n_pose;
        if (nr) {
            memcpy(h_type + r0, P.res_type, nr);
            memcpy(h_pose + r0, P.res_pose, nr * sizeof(int));
            memcpy(h_lm + r0, P.res_lm, nr * sizeof(int));
            memcpy(h_uv + 2 * r0, P.res_uv, 2 * nr * sizeof(double));
            if (h_sig) {
                if (P.res_sigma) memcpy(h_sig + r0, P.res_sigma, nr * sizeof(double));
                else for (size_t k = 0; k < nr; ++k) h_sig[r0 + k] = 1.0;
            }
        }
        if (nl) {
            if (P.inv_depth) {
                memcpy(h_anch + l0, P.lm_anchor_pose, nl * sizeof(int));
                memcpy(h_auv + 2 * l0, P.lm_anchor_uv, 2 * nl * sizeof(double));
            }
            memcpy(h_xl + e * l0, P.lm, e * nl * sizeof(double));
        }
        if (np) {
            memcpy(h_pc + p0, P.pose_const, np);
            memcpy(h_xp + 7 * p0, P.pose, 7 * np * sizeof(double));
        }
        ba_wconst &K = h_wc[w];
        for (int i = 0; i < 4; ++i) { K.Kl[i] = P.calib_l[i]; K.Kr[i] = P.calib_r[i]; }
        pose_Rt(P.T_rl, K.Rrl, K.trl);
    };
    {   // a batch of 64 windows is ~200 MB of host copies: a few threads, windows dealt round-robin (disjoint destinations)
        const int nthr = (n > (4u << 20) && B >= 4) ? std::min(8, B) : 1;
        if (nthr <= 1) {
            for (int w = 0; w < B; ++w) stage_window(w);
        } else {
            std::vector<std::thread> th;
            for (int t = 1; t < nthr; ++t) th.emplace_back([&, t] { for (int w = t; w < B; w += nthr) stage_window(w); });
            for (int w = 0; w < B; w += nthr) stage_window(w);
            for (auto &x : th) x.join();
        }
    }
    memcpy(h_ro, S.res_off.data(), sizeof(int) * (B + 1));
    memcpy(h_lo, S.lm_off.data(), sizeof(int) * (B + 1));
    memcpy(h_po, S.pose_off.data(), sizeof(int) * (B + 1));
    memcpy(h_W, S.hW.data(), sizeof(ba_win) * B);
    if (!c->ba_copy_stream) {
        OV2_HIP(c, hipStreamCreateWithFlags(&c->ba_copy_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) OV2_HIP(c, hipEventCreateWithFlags(&c->ba_copy_ev[i], hipEventDisableTiming));
    }
    OV2_HIP(c, hipMemcpyAsync(c->ba_arena, c->ba_host, head_end, hipMemcpyHostToDevice, c->stream));
    // the two copies share the link: the second starts when the first is through, the kernels of the build run beside it
    OV2_HIP(c, hipEventRecord(c->ba_copy_ev[0], c->stream));
    OV2_HIP(c, hipStreamWaitEvent(c->ba_copy_stream, c->ba_copy_ev[0], 0));
    OV2_HIP(c, hipMemcpyAsync((char *)c->ba_arena + head_end, (char *)c->ba_host + head_end, S.arena_off - head_end, hipMemcpyHostToDevice,
                              c->ba_copy_stream));
    OV2_HIP(c, hipEventRecord(c->ba_copy_ev[1], c->ba_copy_stream));
    S.meas_pending = true;
    }
    S.xp = const_cast<double *>(d_xp); S.xl = const_cast<double *>(d_xl);
    S.W = const_cast<ba_win *>(d_W); S.wc = const_cast<ba_wconst *>(d_wc);
    // device-only arrays that live for the whole batch
    unsigned char *act = nullptr;
    if ((s = dalloc(c, S.arena_off, &act, n)) != OV2_OK) return s;
    R.active = act;
    OV2_HIP(c, hipMemsetAsync(act, 1, std::max<size_t>(n, 1), c->stream));
    if ((s = dalloc(c, S.arena_off, &S.cp, 7 * NP)) != OV2_OK) return s;
    if ((s = dalloc(c, S.arena_off, &S.cl, e * L)) != OV2_OK) return s;
    if ((s = dalloc(c, S.arena_off, &S.chi2_dev, n)) != OV2_OK) return s;       // indexed by batch-wide residual
    if ((s = dalloc(c, S.arena_off, &S.depth_dev, n)) != OV2_OK) return s;
    if ((s = dalloc(c, S.arena_off, &S.outlier_dev, n)) != OV2_OK) return s;
    OV2_HIP(c, hipMemsetAsync(S.outlier_dev, 0, std::max<size_t>(n, 1), c->stream));
    if (NP) OV2_HIP(c, hipMemcpyAsync(S.cp, S.xp, 7 * NP * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    if (L) OV2_HIP(c, hipMemcpyAsync(S.cl, S.xl, e * L * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    S.persist_end = S.arena_off;
    return OV2_OK;
}

int bits_for(size_t v)
{
    int b = 1;
    while (b < 63 && (1ull << b) <= v) ++b;
    return b;
}

// Builds the reduced program of the currently active residual blocks of the windows that are not skipped (see the bb_*
// kernels) and carves the solver's work arrays behind it.  Two small D2H + synchronisations tell the host the sizes.
ov2_status build_program(ba_solver &S)
{
    ov2_ctx *c = S.c;
    ba_dev &d = S.d;
    memset(&d, 0, sizeof(d));
    const int e = S.e, B = S.B;
    d.e = e; d.B = B;
    d.n_pose = S.n_pose; d.n_lm = S.n_lm;
    d.W = S.W; d.wc = S.wc;
    hipStream_t st = c->stream;
    const ba_raw &R = S.raw;
    const int n = S.n_res, L = S.n_lm, NP = S.n_pose;
    S.arena_off = S.persist_end;
    ov2_status s;
    int *used_lm, *used_pose, *escan, *fscan, *eidx, *fidx, *lm_of_e, *pose_of_f, *rows, *row_ptr, *pose_ptr, *pose_ent, *ent_lm;
    int *win_of_e, *win_of_f, *vb_start;
    u64 *hdr, *keys, *keys2, *pk, *pk2;
    ba_prog_out O;
#define AL(ptr, count) if ((s = dalloc(c, S.arena_off, &ptr, (size_t)(count))) != OV2_OK) return s
    AL(hdr, BH_N); AL(used_lm, L + 1); AL(used_pose, NP + 1); AL(escan, L + 1); AL(fscan, NP + 1); AL(eidx, L); AL(fidx, NP);
    AL(lm_of_e, L); AL(pose_of_f, NP); AL(win_of_e, L); AL(win_of_f, NP); AL(vb_start, B + 1);
    AL(keys, n); AL(keys2, n); AL(pk, 2 * (size_t)n); AL(pk2, 2 * (size_t)n);
    AL(O.type, n); AL(O.pose, n); AL(O.lm, n); AL(O.anch, n); AL(O.eb, n); AL(O.fk, n); AL(O.fa, n);
    AL(O.uv, 2 * (size_t)n); AL(O.isg, n); AL(O.rows, n); AL(O.row_win, n); AL(O.rowrec, n);
    AL(row_ptr, L + 1); AL(pose_ptr, NP + 1); AL(pose_ent, 2 * (size_t)n); AL(ent_lm, 2 * (size_t)n);
    rows = O.rows;
    // row keys: [landmark block : bits(L)] [pose block + 1 : fb] [residual index : nbits], one dead bit above
    const int nbits = bits_for((size_t)n), fb = bits_for((size_t)NP + 1);
    const int dead_bit = bits_for((size_t)L) + fb + nbits;
    S.fb = fb;
    if ((long long)L >= (1ll << 28))
        return ov2_set_err(c, OV2_ERR_UNSUPPORTED, "batch too large for the packed row records (%d landmarks): split it", L);
    if (dead_bit > 62)
        return ov2_set_err(c, OV2_ERR_UNSUPPORTED, "batch too large for the packed sort keys (%d residual blocks, %d landmarks, %d "
                           "poses): split it", n, L, NP);
    size_t tmp1 = 0, tmp2 = 0, tmp3 = 0;
    OV2_HIP(c, hipcub::DeviceRadixSort::SortKeys(nullptr, tmp1, keys, keys2, std::max(n, 1), 0, 64, st));
    OV2_HIP(c, hipcub::DeviceRadixSort::SortKeys(nullptr, tmp2, pk, pk2, std::max(2 * n, 1), 0, 64, st));
    OV2_HIP(c, hipcub::DeviceScan::ExclusiveSum(nullptr, tmp3, used_lm, escan, std::max(L, NP) + 1, st));
    size_t tmp_bytes = std::max(std::max(tmp1, tmp2), tmp3);
    unsigned char *tmp;
    AL(tmp, tmp_bytes + 256);
    OV2_HIP(c, hipMemsetAsync(hdr, 0, sizeof(u64) * BH_N, st));
    OV2_HIP(c, hipMemsetAsync(used_lm, 0, sizeof(int) * (L + 1), st));
    OV2_HIP(c, hipMemsetAsync(used_pose, 0, sizeof(int) * (NP + 1), st));
    if (n > 0) BA_LAUNCH(S, K_MISC, bb_mark_kernel, dim3((n + 255) / 256), dim3(256), 0, st, R, S.W, used_lm, used_pose, hdr);
    OV2_HIP(c, hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, used_lm, escan, L + 1, st));     // also for an empty batch:
    OV2_HIP(c, hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, used_pose, fscan, NP + 1, st));  // the window ranges read them
    if (n > 0) {
        const dim3 gn((n + 255) / 256);
        if (L) BA_LAUNCH(S, K_MISC, bb_number_kernel, dim3((L + 255) / 256), dim3(256), 0, st, used_lm, escan, L, eidx, lm_of_e);
        if (NP) BA_LAUNCH(S, K_MISC, bb_number_kernel, dim3((NP + 255) / 256), dim3(256), 0, st, used_pose, fscan, NP, fidx, pose_of_f);
        // hdr[BH_NE] / hdr[BH_NF] = the last scan entries
        OV2_HIP(c, hipMemcpyAsync(hdr + BH_NE, escan + L, sizeof(int), hipMemcpyDeviceToDevice, st));
        OV2_HIP(c, hipMemcpyAsync(hdr + BH_NF, fscan + NP, sizeof(int), hipMemcpyDeviceToDevice, st));
        BA_LAUNCH(S, K_MISC, bb_keys_kernel, gn, dim3(256), 0, st, R, S.W, eidx, fidx, keys, hdr, nbits, fb, dead_bit);
        BA_LAUNCH(S, K_MISC, bb_rows_kernel, dim3(1), dim3(64), 0, st, hdr);
        OV2_HIP(c, hipcub::DeviceRadixSort::SortKeys(tmp, tmp_bytes, keys, keys2, n, nbits, dead_bit + 1, st));
        if (S.meas_pending) {   // the sorted rows take their measurements along: the second half of the upload must be in
            OV2_HIP(c, hipStreamWaitEvent(st, c->ba_copy_ev[1], 0));
            S.meas_pending = false;
        }
        BA_LAUNCH(S, K_MISC, bb_fill_kernel, gn, dim3(256), 0, st, R, keys2, eidx, fidx, hdr, O, nbits);
        BA_LAUNCH(S, K_MISC, bb_rowptr_kernel, gn, dim3(256), 0, st, O.eb, hdr, row_ptr);
        BA_LAUNCH(S, K_MISC, bb_posekeys_kernel, gn, dim3(256), 0, st, O.fk, O.fa, hdr, n, pk, fb);
        OV2_HIP(c, hipcub::DeviceRadixSort::SortKeys(tmp, tmp_bytes, pk, pk2, 2 * n, 32, 32 + fb + 1, st));
        BA_LAUNCH(S, K_MISC, bb_poseptr_kernel, dim3((2 * n + 255) / 256), dim3(256), 0, st, pk2, 2 * n, hdr, pose_ptr, pose_ent, fb, O.eb, ent_lm);
        BA_LAUNCH(S, K_MISC, bb_winfill_kernel, dim3((std::max(L, NP) + 255) / 256), dim3(256), 0, st, R, hdr, lm_of_e, pose_of_f,
                  win_of_e, win_of_f);
    }
    BA_LAUNCH(S, K_MISC, bb_winranges_kernel, dim3(1), dim3(256), 0, st, R, escan, fscan, O.row_win, hdr, S.W, vb_start);
    // sizes + the window ranges come back through the pinned mirror (its upload was enqueued before these copies)
    u64 *h_hdr = (u64 *)c->ba_host;
    ba_win *h_W = (ba_win *)(h_hdr + 128);   // behind the BH_N header words
    OV2_HIP(c, hipMemcpyAsync(h_hdr, hdr, sizeof(u64) * BH_N, hipMemcpyDeviceToHost, st));
    OV2_HIP(c, hipMemcpyAsync(h_W, S.W, sizeof(ba_win) * B, hipMemcpyDeviceToHost, st));
    OV2_HIP(c, hipStreamSynchronize(st));
    if (h_hdr[BH_ERR]) {
        static const char *what[] = {"", "landmark out of range", "unknown type", "type does not match inv_depth",
                                     "pose out of range", "anchor pose out of range"};
        const int code = (int)(h_hdr[BH_ERR] >> 56);
        return ov2_set_err(c, OV2_ERR_INVALID, "window %d, residual %d: %s", (int)((h_hdr[BH_ERR] >> 32) & 0xffffff),
                           (int)(h_hdr[BH_ERR] & 0xffffffffu), what[code > 5 ? 0 : code]);
    }
    memcpy(S.hW.data(), h_W, sizeof(ba_win) * B);
    d.n_rows = (int)h_hdr[BH_ROWS]; d.n_e = (int)h_hdr[BH_NE]; d.n_f = (int)h_hdr[BH_NF];
    d.m = 6 * d.n_f; d.nc = d.n_e * e + d.m;
    S.vblocks = (int)h_hdr[BH_VB]; S.mmax = (int)h_hdr[BH_MMAX];
    const size_t s_tot = (size_t)h_hdr[BH_STOT];
    const long long pair_cap_bound = (long long)h_hdr[BH_PAIRCAP];
    S.cover_max = 1;
    for (int w = 0; w < B; ++w) {
        const ba_win &X = S.hW[w];
        const long long cov = std::max<long long>((long long)(X.e1 - X.e0) * e + X.m, (long long)X.m * X.m);
        if (cov > S.cover_max) S.cover_max = (int)std::min<long long>(cov, 0x7fffffff);
    }
    d.type = O.type; d.pose = O.pose; d.lm = O.lm; d.anch = O.anch; d.eb = O.eb; d.fk = O.fk; d.fa = O.fa;
    d.rowrec = O.rowrec;
    d.uv = O.uv; d.inv_sigma = O.isg; d.lm_auv = R.lm_auv; d.row_ptr = row_ptr; d.lm_of_e = lm_of_e; d.pose_of_f = pose_of_f;
    d.row_win = O.row_win; d.win_of_e = win_of_e; d.win_of_f = win_of_f; d.vb_start = vb_start;
    S.rows = rows; S.pose_ptr = pose_ptr; S.pose_ent = pose_ent;
    d.ent_lm = ent_lm;
    const int nr = d.n_rows;
#undef AL
#define AL(field, count) if ((s = dalloc(c, S.arena_off, &d.field, (size_t)(count))) != OV2_OK) return s
    AL(res, 2 * (size_t)nr); AL(Je, 2 * (size_t)e * nr); AL(G, 6 * (size_t)nr); AL(wpt, WPT_S * (size_t)d.n_e + WPT_S);
    OV2_HIP(c, hipMemsetAsync(d.wpt, 0, (WPT_S * (size_t)d.n_e + WPT_S) * sizeof(double), st));
    AL(scale, d.nc); AL(sqn, d.nc); AL(grad, d.nc); AL(diag, d.nc); AL(lmd, d.nc); AL(step, d.nc);
    AL(Spool, s_tot); AL(rhs, d.m + 1); AL(iete, (size_t)d.n_e * e * e); AL(ieg, (size_t)d.n_e * e); AL(FFp, (size_t)d.n_f * 21);
    AL(part, 3 * ((size_t)d.n_e + d.n_f) + (size_t)S.vblocks + 16);
    AL(n_active, 64);   // |step|^2, |x+|^2, model change per block | cost partials per virtual block
    AL(dead, (size_t)nr + 1); AL(e_dead, (size_t)d.n_e + 1); AL(f_live, (size_t)d.n_f + 1);
    OV2_HIP(c, hipMemsetAsync(d.dead, 0, (size_t)nr + 1, st));
    d.masked = 0;
#undef AL
    S.chold_stride = (size_t)(S.mmax / CHOL_NB + 1) * CHOL_NB * CHOL_NB;
    if ((s = dalloc(c, S.arena_off, &S.chold, S.chold_stride * B)) != OV2_OK) return s;
    S.pair_cap = 0;
    if (d.n_rows == 0 || d.n_e == 0) return OV2_OK;
    // ---- structure of the atomic-free Schur complement (bs_* kernels)
    {
        int *ncell, *npair, *cell_ptr, *pair_off;
#define AL2(ptr, count) if ((s = dalloc(c, S.arena_off, &ptr, (size_t)(count))) != OV2_OK) return s
#define AL3(ptr, count) if ((s = dalloc2(c, off2, &ptr, (size_t)(count))) != OV2_OK) return s
        AL2(ncell, d.n_e + 1); AL2(npair, d.n_e + 1); AL2(cell_ptr, d.n_e + 1); AL2(pair_off, d.n_e + 1);
        size_t tb = 0;
        OV2_HIP(c, hipcub::DeviceScan::ExclusiveSum(nullptr, tb, ncell, cell_ptr, d.n_e + 1, st));
        unsigned char *tscan;
        AL2(tscan, tb + 256);
        BA_LAUNCH(S, K_MISC, bs_count_kernel, dim3((d.n_e + 256) / 256), dim3(256), 0, st, d, ncell, npair);
        OV2_HIP(c, hipcub::DeviceScan::ExclusiveSum(tscan, tb, ncell, cell_ptr, d.n_e + 1, st));
        OV2_HIP(c, hipcub::DeviceScan::ExclusiveSum(tscan, tb, npair, pair_off, d.n_e + 1, st));
        int *h_tot = (int *)c->ba_host;
        OV2_HIP(c, hipMemcpyAsync(h_tot, cell_ptr + d.n_e, sizeof(int), hipMemcpyDeviceToHost, st));
        OV2_HIP(c, hipMemcpyAsync(h_tot + 1, pair_off + d.n_e, sizeof(int), hipMemcpyDeviceToHost, st));
        OV2_HIP(c, hipStreamSynchronize(st));
        const int C_tot = h_tot[0], P_tot = h_tot[1];
        if (C_tot < 0 || P_tot < 0)
            return ov2_set_err(c, OV2_ERR_UNSUPPORTED, "batch too large: more than 2^31 Schur cells or cell pairs");
        int *cell_f, *cell_row, *cell_lm, *cell_rank, *pcell_ptr, *pcell_ent, *seg_start, *head, *rank, *pent;
        int2 *pent_sorted;
        u64 *ukey, *pkey = nullptr, *pkey2 = nullptr, *ckey = nullptr, *ckey2 = nullptr;
        ba_cells &Cc = S.cells;
        const long long pair_cap = std::min<long long>((long long)P_tot, pair_cap_bound);
        const int fb = bits_for((size_t)d.n_f), pb = bits_for((size_t)P_tot);   // pair keys: [pose hi : fb] [pose lo : fb] [entry : pb]
        S.fb = fb; S.pb = pb;
        if (2 * fb + pb > 64)
            return ov2_set_err(c, OV2_ERR_UNSUPPORTED, "batch too large for the packed pair keys (%d free poses, %d cell pairs): split it",
                               d.n_f, P_tot);
        size_t t1 = 0, t2 = 0, t4 = 0;
        OV2_HIP(c, hipcub::DeviceRadixSort::SortKeys(nullptr, t1, pkey, pkey2, std::max(P_tot, 1), 0, 64, st));
        OV2_HIP(c, hipcub::DeviceRadixSort::SortKeys(nullptr, t2, ckey, ckey2, std::max(C_tot, 1), 0, 64, st));
        OV2_HIP(c, hipcub::DeviceScan::ExclusiveSum(nullptr, t4, (int *)nullptr, (int *)nullptr, P_tot + 1, st));
        size_t tbytes = std::max(std::max(t1, t2), t4);
        {   // the structure lives in its own block, sized now that the counts are known and kept across solves
            const size_t need2 = (size_t)C_tot * (6 * 4 + 2 * 8 + 16 + (size_t)(e == 1 ? cell_rec<1>::STRIDE : cell_rec<3>::STRIDE) * 8) + (size_t)P_tot * (2 * 8 + 6 * 4) +
                                 (size_t)pair_cap * 12 + (size_t)d.n_f * 4 + tbytes + 64 * 256;
            if (need2 > c->ba_arena2_cap) {
                OV2_HIP(c, hipStreamSynchronize(st));
                if (c->ba_arena2) OV2_HIP(c, hipFree(c->ba_arena2));
                c->ba_arena2 = nullptr; c->ba_arena2_cap = 0;
                const size_t want = need2 + need2 / 4;
                hipError_t he = hipMalloc(&c->ba_arena2, want);
                if (he != hipSuccess) return ov2_set_err(c, OV2_ERR_NOMEM, "BA structure arena hipMalloc(%zu): %s", want, hipGetErrorString(he));
                c->ba_arena2_cap = want;
            }
        }
        size_t off2 = 0;
        AL3(cell_f, C_tot); AL3(cell_row, C_tot); AL3(cell_lm, C_tot); AL3(cell_rank, C_tot);
        AL3(Cc.V, (size_t)C_tot * (e == 1 ? cell_rec<1>::STRIDE : cell_rec<3>::STRIDE));
        int4 *qrow;
        AL3(qrow, C_tot);
        AL3(pcell_ptr, d.n_f + 1); AL3(pcell_ent, C_tot); AL3(ckey, C_tot); AL3(ckey2, C_tot);
        AL3(pkey, P_tot); AL3(pkey2, P_tot); AL3(pent, 2 * (size_t)P_tot); AL3(pent_sorted, P_tot); AL3(head, P_tot + 1); AL3(rank, P_tot + 1);
        AL3(ukey, pair_cap + 1); AL3(seg_start, pair_cap + 2);
        unsigned char *tsort;
        AL3(tsort, tbytes + 256);
#undef AL3
#undef AL2
        Cc.cell_ptr = cell_ptr; Cc.cell_f = cell_f; Cc.cell_row = cell_row; Cc.cell_lm = cell_lm; Cc.cell_rank = cell_rank; Cc.qrow = qrow;
        BA_LAUNCH(S, K_MISC, bs_cells_kernel, dim3((d.n_e + 31) / 32), dim3(256), 0, st, d, cell_ptr, cell_f, cell_row, cell_lm, ckey);
        if (P_tot > 0)
            BA_LAUNCH(S, K_MISC, bs_pairs_kernel, dim3((d.n_e + 31) / 32), dim3(256), 0, st, d, cell_ptr, pair_off, cell_f, pkey, pent, fb, pb);
        if (C_tot > 0) {
            OV2_HIP(c, hipcub::DeviceRadixSort::SortKeys(tsort, tbytes, ckey, ckey2, C_tot, 32, 32 + fb, st));
            BA_LAUNCH(S, K_MISC, bs_posecells_kernel, dim3((C_tot + 255) / 256), dim3(256), 0, st, ckey2, C_tot, d.n_f, pcell_ptr, pcell_ent);
            BA_LAUNCH(S, K_MISC, bs_rank_kernel, dim3((C_tot + 255) / 256), dim3(256), 0, st, pcell_ent, C_tot, cell_rank);
            BA_LAUNCH(S, K_MISC, bs_qrow_kernel, dim3((C_tot + 255) / 256), dim3(256), 0, st, d, cell_f, cell_row, cell_lm, cell_rank, C_tot, qrow);
        }
        if (P_tot > 0) {
            BA_LAUNCH(S, K_MISC, bs_pent_kernel, dim3((unsigned)((2 * (size_t)P_tot + 255) / 256)), dim3(256), 0, st, pent, 2 * (size_t)P_tot,
                      cell_rank, cell_row);
            OV2_HIP(c, hipcub::DeviceRadixSort::SortKeys(tsort, tbytes, pkey, pkey2, P_tot, pb, pb + 2 * fb, st));
            BA_LAUNCH(S, K_MISC, bs_pent_sorted_kernel, dim3((P_tot + 255) / 256), dim3(256), 0, st, pkey2, P_tot, pb, pent, pent_sorted);
            BA_LAUNCH(S, K_MISC, bs_heads_kernel, dim3((P_tot + 256) / 256), dim3(256), 0, st, pkey2, P_tot, head, pb);
            OV2_HIP(c, hipcub::DeviceScan::ExclusiveSum(tsort, tbytes, head, rank, P_tot + 1, st));
            BA_LAUNCH(S, K_MISC, bs_segs_kernel, dim3((P_tot + 255) / 256), dim3(256), 0, st, pkey2, P_tot, head, rank, seg_start, ukey, pb);
        }
        S.pcell_ptr = pcell_ptr; S.pcell_ent = pcell_ent; S.n_pairs = rank + P_tot;   // exclusive rank behind the last entry = number of pose pairs
        S.seg_start = seg_start; S.pair_key = ukey; S.pair_val = pkey2; S.pair_ent = pent_sorted; S.pair_cap = P_tot > 0 ? pair_cap : 0;
    }
    return OV2_OK;
}

ba_lmopt make_lmopt(const ov2_ba_options *o)
{
    ba_lmopt L;
    L.min_d = o->min_lm_diagonal; L.max_d = o->max_lm_diagonal; L.max_radius = o->max_radius; L.min_radius = o->min_radius;
    L.min_rel = o->min_relative_decrease; L.ptol = o->parameter_tolerance; L.gtol = o->gradient_tolerance;
    L.ftol = o->function_tolerance; L.max_invalid = o->max_consecutive_invalid_steps; L.jacobi = o->jacobi_scaling ? 1 : 0;
    return L;
}

__global__ void ba_fill_kernel(double *__restrict__ p, size_t n, double v)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// a window of the program that has nothing to minimise ends before iteration zero, as TrustRegionMinimizer does not run
__global__ void ba_skip_empty_kernel(ba_dev d)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= d.B) return;
    ba_win &W = d.W[w];
    if (W.skip) return;
    W.done = 0; W.active = 0; W.valid = 0; W.accepted = 0; W.eval_at_cand = 0; W.chol_fail = 0;
    W.nbad = W.n_left = W.n_right = 0;
    if (W.row1 == W.row0 || (W.e1 - W.e0) * d.e + W.m == 0) {
        W.skip = 2;   // in the program, but empty
        W.done = 1; W.termination = OV2_BA_TERM_SKIPPED;
        W.initial_cost = W.minimum_cost = W.x_cost = 0.0;
    }
}

// Enqueues TrustRegionMinimizer::Minimize for every window of the program: iteration zero + max_rounds LM rounds, a fixed
// chain of launches with no host synchronisation (every kernel skips the windows that are done).
ov2_status enqueue_minimize(ba_solver &S, int max_rounds)
{
    ov2_ctx *c = S.c;
    const ov2_ba_options *o = S.o;
    ba_dev &d = S.d;
    hipStream_t st = c->stream;
    const int e = d.e, B = S.B;
    const ba_lmopt lo = make_lmopt(o);
    BA_LAUNCH(S, K_MISC, ba_skip_empty_kernel, dim3((B + 63) / 64), dim3(64), 0, st, d);
    if (d.n_rows == 0 || d.nc == 0) return OV2_OK;
    if (max_rounds > 62) max_rounds = 62;
    OV2_HIP(c, hipMemsetAsync(d.n_active, 0, 64 * sizeof(int), st));
    // candidate buffers start as copies of x: blocks outside this program are never written by Plus, and x <- candidate
    // copies whole windows (after a first solve the buffers still hold its last, possibly rejected, candidate)
    if (S.n_pose) OV2_HIP(c, hipMemcpyAsync(S.cp, S.xp, 7 * (size_t)S.n_pose * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (S.n_lm) OV2_HIP(c, hipMemcpyAsync(S.cl, S.xl, (size_t)e * S.n_lm * sizeof(double), hipMemcpyDeviceToDevice, st));
    const dim3 g_rows(S.vblocks), g_lm((d.n_e + BS_LM_PER_WG - 1) / BS_LM_PER_WG), g_win(B);
    auto eval = [&](bool jac, const double *xp, const double *xl, int mode) {
        double *pc = d.part + 3 * (size_t)(d.n_e + d.n_f);
        if (jac) {
            if (e == 1) BA_LAUNCH(S, K_EVAL, (ba_eval_kernel<true, 1>), g_rows, dim3(256), 0, st, d, xp, xl, mode, o->huber_delta, pc, S.raw.pose_off);
            else BA_LAUNCH(S, K_EVAL, (ba_eval_kernel<true, 3>), g_rows, dim3(256), 0, st, d, xp, xl, mode, o->huber_delta, pc, S.raw.pose_off);
        } else {
            if (e == 1) BA_LAUNCH(S, K_EVAL, (ba_eval_kernel<false, 1>), g_rows, dim3(256), 0, st, d, xp, xl, mode, o->huber_delta, pc, S.raw.pose_off);
            else BA_LAUNCH(S, K_EVAL, (ba_eval_kernel<false, 3>), g_rows, dim3(256), 0, st, d, xp, xl, mode, o->huber_delta, pc, S.raw.pose_off);
        }
    };
    auto colnorm = [&](int mode) {
        BA_LAUNCH(S, K_COLNORM, ba_colnorm16_kernel<BS_LM_GW>, g_lm, dim3(256), 0, st, d, mode);
        if (d.n_f > 0) BA_LAUNCH(S, K_COLNORM, ba_pose_normal_kernel, dim3(d.n_f), dim3(256), 0, st, d, S.pose_ptr, S.pose_ent, mode);
    };
    const int nb = d.n_e + d.n_f;
    double *part_step = d.part, *part_norm = d.part + nb, *part_model = d.part + 2 * (size_t)nb;
    double *part_cost = d.part + 3 * (size_t)nb;
    // ---- IterationZero: EvaluateGradientAndJacobian at x.  The rows hold the unscaled jacobian; the Jacobi scaling
    // (trust_region_minimizer.cc:185-199: 1 / (1 + column norm) of the FIRST jacobian) is a vector the consumers apply.
    BA_LAUNCH(S, K_SCALE, ba_fill_kernel, dim3((d.nc + 255) / 256), dim3(256), 0, st, d.scale, (size_t)d.nc, 1.0);
    eval(true, S.xp, S.xl, EV_ZERO);
    colnorm(EV_ZERO);
    if (lo.jacobi) {
        // the LM diagonal is taken from the SCALED jacobian (levenberg_marquardt_strategy.cc:82): the kernel also turns the
        // norms, the gradient and F'F into those of the scaled jacobian; the tolerance test unscales the gradient on the fly
        BA_LAUNCH(S, K_SCALE, ba_make_scale_kernel, dim3((d.nc + 255) / 256), dim3(256), 0, st, d);
    }
    BA_LAUNCH(S, K_REDUCE, ba_winreduce_kernel<WR_JAC>, g_win, dim3(256), 0, st, d, lo, S.xp, part_cost, part_step, part_norm, part_model,
              1, o->initial_radius, 0);
    static const int dbg_gather = getenv("OV2_BA_GATHER_ONLY") ? atoi(getenv("OV2_BA_GATHER_ONLY")) : 0;
    static const int chol_multi_min = getenv("OV2_CHOL_MULTI_MIN") ? atoi(getenv("OV2_CHOL_MULTI_MIN")) : CHOL_MULTI_MIN;
    for (int round = 0; round < max_rounds; ++round) {
        // ---- ComputeTrustRegionStep
        BA_LAUNCH(S, K_LMDIAG, ba_lmdiag_sinit_kernel, dim3((unsigned)((S.cover_max + 255) / 256), B), dim3(256), 0, st, d, lo.min_d, lo.max_d);
        {
            const long long gblocks = (long long)d.n_f + (((S.pair_cap + 3) / 4 + 7) / 8) * 8;   // pair blocks: a multiple of 8 (XCD renumbering)
            if (e == 1) {
                BA_LAUNCH(S, K_SCHUR, (bs_landmark_kernel<1, BS_LM_GW>), g_lm, dim3(256), 0, st, d, S.cells);
                if (gblocks > 0)
                    BA_LAUNCH(S, K_SCHUR, bs_gather_kernel<1>, dim3((unsigned)gblocks), dim3(256), 0, st, d, S.cells, S.pcell_ptr,
                              S.n_pairs, S.pair_key, S.seg_start, S.pair_ent, S.fb, dbg_gather);
            } else {
                BA_LAUNCH(S, K_SCHUR, (bs_landmark_kernel<3, BS_LM_GW>), g_lm, dim3(256), 0, st, d, S.cells);
                if (gblocks > 0)
                    BA_LAUNCH(S, K_SCHUR, bs_gather_kernel<3>, dim3((unsigned)gblocks), dim3(256), 0, st, d, S.cells, S.pcell_ptr,
                              S.n_pairs, S.pair_key, S.seg_start, S.pair_ent, S.fb, dbg_gather);
            }
        }
        if (S.mmax > 0) {
            const int m = S.mmax;
            if (m >= chol_multi_min) {
                // right-looking, two launches per panel (trailing update on the matrix cores), then the backward pass;
                // window = blockIdx.y, grids sized for the largest window
                for (int k0 = 0; k0 < m; k0 += CHOL_NB) {
                    const int nbk = std::min(CHOL_NB, m - k0), below = m - (k0 + nbk) + 1;
                    BA_LAUNCH(S, K_CHOL, ba_chol_panel_kernel, dim3((below + 63) / 64, B), dim3(64), 0, st, d, S.chold, S.chold_stride, k0);
                    if (k0 + nbk < m) {
                        const int t = m - (k0 + nbk), nt = (t + 15) / 16;
                        BA_LAUNCH(S, K_CHOL, ba_chol_syrk_kernel, dim3(nt * (nt + 1) / 2 + (t + 63) / 64, B), dim3(64), 0, st, d, k0);
                    }
                }
                BA_LAUNCH(S, K_CHOL, ba_chol_backward_kernel, dim3(B), dim3(256), (size_t)(m + CHOL_NB + 2) * 8, st, d, S.chold, S.chold_stride);
            } else {
                static const int chol_mfma = getenv("OV2_CHOL_MFMA") ? atoi(getenv("OV2_CHOL_MFMA")) : 1;
                const size_t ldsm = ((size_t)(m + 1) * 33 + 32 + 8) * 8;
                if (chol_mfma && ldsm <= 158 * 1024) {
                    BA_LAUNCH(S, K_CHOL, ba_chol_mfma_kernel, g_win, dim3(CHOL_THREADS), ldsm, st, d);
                } else {
                // one workgroup per window; panel width by LDS budget: (m+1) x (NB+1) + NB x 64 doubles <= 158 KiB
                const size_t lds32 = ((size_t)(m + 1) * 33 + 32 * 64 + 4) * 8, lds16 = ((size_t)(m + 1) * 17 + 16 * 64 + 4) * 8,
                             lds8 = ((size_t)(m + 1) * 9 + 8 * 64 + 4) * 8;
                if (lds32 <= 158 * 1024) BA_LAUNCH(S, K_CHOL, ba_chol_kernel<32>, g_win, dim3(CHOL_THREADS), lds32, st, d);
                else if (lds16 <= 158 * 1024) BA_LAUNCH(S, K_CHOL, ba_chol_kernel<16>, g_win, dim3(CHOL_THREADS), lds16, st, d);
                else if (lds8 <= 158 * 1024) BA_LAUNCH(S, K_CHOL, ba_chol_kernel<8>, g_win, dim3(CHOL_THREADS), lds8, st, d);
                else return ov2_set_err(c, OV2_ERR_UNSUPPORTED, "reduced camera system of %d unknowns exceeds the one-workgroup Cholesky", m);
                }
            }
        }
        {
            const int bgrid = std::max((d.n_e + BS_LM_PER_WG - 1) / BS_LM_PER_WG, (d.m + 255) / 256);
            if (e == 1) BA_LAUNCH(S, K_BACKSUB, (ba_backsub16_kernel<1, BS_LM_GW>), dim3(bgrid), dim3(256), 0, st, d, part_model);
            else BA_LAUNCH(S, K_BACKSUB, (ba_backsub16_kernel<3, BS_LM_GW>), dim3(bgrid), dim3(256), 0, st, d, part_model);
        }
        BA_LAUNCH(S, K_REDUCE, ba_winreduce_kernel<WR_MODEL>, g_win, dim3(256), 0, st, d, lo, S.xp, part_cost, part_step, part_norm,
                  part_model, 0, 0.0, 0);
        // ---- ComputeCandidatePointAndEvaluateCost (windows with a valid step)
        BA_LAUNCH(S, K_PLUS, ba_plus_kernel, dim3((nb + 63) / 64), dim3(64), 0, st, d, S.xp, S.xl, S.cp, S.cl, d.step, lo.jacobi,
                  part_step, part_norm);
        eval(false, S.cp, S.cl, EV_CAND);
        BA_LAUNCH(S, K_REDUCE, ba_winreduce_kernel<WR_CAND>, g_win, dim3(256), 0, st, d, lo, S.xp, part_cost, part_step, part_norm,
                  part_model, 0, 0.0, 0);
        // ---- HandleSuccessfulStep (windows whose step was accepted): x <- candidate, jacobian at the new x
        BA_LAUNCH(S, K_MISC, ba_accept_kernel, dim3((std::max(S.n_pose, S.n_lm) + 255) / 256), dim3(256), 0, st, d, S.raw.pose_off,
                  S.raw.lm_off, S.xp, S.cp, S.xl, S.cl);
        eval(true, S.xp, S.xl, EV_ACC);
        colnorm(EV_ACC);
        BA_LAUNCH(S, K_REDUCE, ba_winreduce_kernel<WR_JAC>, g_win, dim3(256), 0, st, d, lo, S.xp, part_cost, part_step, part_norm,
                  part_model, 0, 0.0, round + 1);
        // Rounds are enqueued blind (no host round trip) until the first one a typical solve can end in; from then on the
        // host looks at how many windows entered the NEXT round and stops at zero (an idle round is ~14 launches whose
        // grids still scale with the batch).
        if (round + 1 >= 2 && round + 1 < max_rounds) {
            int *h_na = (int *)c->ba_host;
            OV2_HIP(c, hipMemcpyAsync(h_na, d.n_active + round + 1, sizeof(int), hipMemcpyDeviceToHost, st));
            OV2_HIP(c, hipStreamSynchronize(st));
            if (*h_na == 0) break;
        }
    }
    OV2_HIP(c, hipGetLastError());
    return OV2_OK;
}

// chi2 / depth flags of the rows of the program that was just minimised (src/optimizer.cpp:500-592, 637-735)
ov2_status enqueue_flags(ba_solver &S, int pass)
{
    ba_dev &d = S.d;
    if (d.n_rows == 0) return OV2_OK;
    BA_LAUNCH(S, K_FLAG, ba_flag_kernel, dim3((d.n_rows + 255) / 256), dim3(256), 0, S.c->stream, d, S.xp, S.xl, S.cp, S.cl, S.rows,
              S.o->chi2_th, pass, S.want_chi2 ? S.chi2_dev : nullptr, S.want_depth ? S.depth_dev : nullptr, S.raw.active,
              S.want_out ? S.outlier_dev : nullptr);
    OV2_HIP(S.c, hipGetLastError());
    return OV2_OK;
}

ov2_status fetch_windows(ba_solver &S)
{
    ov2_ctx *c = S.c;
    ba_win *h_W = (ba_win *)c->ba_host;
    OV2_HIP(c, hipMemcpyAsync(h_W, S.W, sizeof(ba_win) * S.B, hipMemcpyDeviceToHost, c->stream));
    OV2_HIP(c, hipStreamSynchronize(c->stream));
    memcpy(S.hW.data(), h_W, sizeof(ba_win) * S.B);
    return OV2_OK;
}

}  // namespace

#ifdef OV2_CHOL_PROF
extern "C" int ov2_debug_chol_prof(unsigned long long *out, int reset)
{
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chol_prof), sizeof(z)) != hipSuccess) return 1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_chol_prof), z, sizeof(z)) != hipSuccess) return 1;
    return 0;
}
#endif

extern "C" void ov2_ba_default_options(ov2_ba_options *o, float robust_mono_th)
{
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->huber_delta = (double)sqrtf(robust_mono_th);  // HuberLoss(std::sqrt(mono_th)), float (src/optimizer.cpp:48-49)
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

extern "C" ov2_status ov2_ba_solve(ov2_ctx *c, const ov2_ba_problem *P, const ov2_ba_options *o, ov2_ba_result *R)
{
    return ov2_ba_solve_batch(c, 1, P, o, R);
}

static ov2_status ba_solve_batch_impl(ov2_ctx *c, int B, const ov2_ba_problem *P, const ov2_ba_options *o, ov2_ba_result *R, bool on_device);

extern "C" ov2_status ov2_ba_solve_batch(ov2_ctx *c, int B, const ov2_ba_problem *P, const ov2_ba_options *o, ov2_ba_result *R)
{
    return ba_solve_batch_impl(c, B, P, o, R, false);
}

extern "C" ov2_status ov2_ba_solve_batch_dev(ov2_ctx *c, int B, const ov2_ba_problem *P, const ov2_ba_options *o, ov2_ba_result *R)
{
    return ba_solve_batch_impl(c, B, P, o, R, true);
}

static ov2_status ba_solve_batch_impl(ov2_ctx *c, int B, const ov2_ba_problem *P, const ov2_ba_options *o, ov2_ba_result *R, bool on_device)
{
    if (!c) return OV2_ERR_INVALID;
    if (B == 0) return OV2_OK;
    if (B < 0 || !P || !o || !R) return ov2_set_err(c, OV2_ERR_INVALID, "null problem/options/result");
    ba_solver S;
    S.c = c; S.B = B; S.P = P; S.o = o; S.on_device = on_device; S.Rdev = R;
    for (int w = 0; w < B; ++w) { S.want_chi2 |= R[w].chi2 != nullptr; S.want_depth |= R[w].depth_positive != nullptr; S.want_out |= R[w].outlier != nullptr; }
    S.e = P[0].inv_depth ? 1 : 3;
    S.res_off.assign(B + 1, 0); S.lm_off.assign(B + 1, 0); S.pose_off.assign(B + 1, 0);
    long long tn = 0, tl = 0, tp = 0;
    size_t s_bound = 0, chold_bound = 0;
    for (int w = 0; w < B; ++w) {
        const ov2_ba_problem &Q = P[w];
        if (Q.n_pose < 0 || Q.n_lm < 0 || Q.n_res < 0 || (Q.n_pose && (!Q.pose || !Q.pose_const)) || (Q.n_lm && !Q.lm) ||
            (Q.n_res && (!Q.res_type || !Q.res_pose || !Q.res_lm || !Q.res_uv)) ||
            (Q.inv_depth && Q.n_lm && (!Q.lm_anchor_pose || !Q.lm_anchor_uv)))
            return ov2_set_err(c, OV2_ERR_INVALID, "inconsistent ov2_ba_problem (window %d)", w);
        if ((Q.inv_depth ? 1 : 3) != S.e)
            return ov2_set_err(c, OV2_ERR_INVALID, "the windows of a batch must share one landmark parametrisation (window %d)", w);
        S.res_off[w] = (int)tn; S.lm_off[w] = (int)tl; S.pose_off[w] = (int)tp;
        tn += Q.n_res; tl += Q.n_lm; tp += Q.n_pose;
        if (Q.res_sigma) S.any_sigma = true;
        const size_t m6 = 6 * (size_t)Q.n_pose;
        s_bound += m6 * m6;
        chold_bound += (m6 / CHOL_NB + 1) * CHOL_NB * CHOL_NB;
    }
    if (tn >= (1ll << 31) - 1024 || tl >= (1ll << 31) - 1024 || tp >= (1ll << 31) - 1024)
        return ov2_set_err(c, OV2_ERR_UNSUPPORTED, "batch too large: split it (more than 2^31 residual blocks / landmarks / poses)");
    S.res_off[B] = (int)tn; S.lm_off[B] = (int)tl; S.pose_off[B] = (int)tp;
    S.n_res = (int)tn; S.n_lm = (int)tl; S.n_pose = (int)tp;
    OV2_HIP(c, hipSetDevice(c->device));
    const int use_loss = o->huber_delta > 0.0;
    S.hW.assign(B, ba_win());
    for (int w = 0; w < B; ++w) {
        ba_win &X = S.hW[w];
        memset(&X, 0, sizeof(X));
        X.use_loss = use_loss; X.max_iters = o->max_iters;
        ov2_ba_result &Rw = R[w];
        Rw.n_log = 0; Rw.n_log_robust = 0; Rw.l2_done = 0; Rw.n_outliers_pass1 = Rw.n_outliers_pass2 = 0;
        Rw.initial_cost = Rw.final_cost = Rw.l2_initial_cost = Rw.l2_final_cost = 0.0;
        Rw.termination = Rw.l2_termination = OV2_BA_TERM_SKIPPED;
        if (Rw.outlier && !on_device) memset(Rw.outlier, 0, (size_t)P[w].n_res);
    }
    // arena: upper bound of everything upload_batch + build_program carve for the full batch
    {
        const size_t n = (size_t)tn, L = (size_t)tl, NP = (size_t)tp;
        // per residual block: raw copy 33 B + flags 3 + chi2 8 + sort keys / values 80 + program records 85 + radix-sort
        // scratch (~ keys + values) + jacobian rows 256 -> 600 with slack
        const size_t need = n * 632 + L * 720 + NP * 1280 + s_bound * 8 + chold_bound * 8 + (size_t)B * (sizeof(ba_win) + 1024) + (4u << 20);
        if (need > c->ba_arena_cap) {
            OV2_HIP(c, hipStreamSynchronize(c->stream));
            if (c->ba_arena) OV2_HIP(c, hipFree(c->ba_arena));
            c->ba_arena = nullptr; c->ba_arena_cap = 0;
            const size_t want = need + need / 4;
            hipError_t he = hipMalloc(&c->ba_arena, want);
            if (he != hipSuccess) return ov2_set_err(c, OV2_ERR_NOMEM, "BA arena hipMalloc(%zu): %s", want, hipGetErrorString(he));
            c->ba_arena_cap = want;
        }
        // pinned mirror of the uploaded head (raw problems + initial states + window records) and of what comes back
        // (window records, states, chi2 / depth / outlier flags)
        const size_t hneed = n * 64 + L * 96 + NP * 192 + (size_t)B * (2 * sizeof(ba_win) + sizeof(ba_wconst) + 256) + (1u << 20);
        if (hneed > c->ba_host_cap) {
            OV2_HIP(c, hipStreamSynchronize(c->stream));
            if (c->ba_host) OV2_HIP(c, hipHostFree(c->ba_host));
            c->ba_host = nullptr; c->ba_host_cap = 0;
            const size_t want = hneed + hneed / 4;
            hipError_t he = hipHostMalloc(&c->ba_host, want, hipHostMallocDefault);
            if (he != hipSuccess) return ov2_set_err(c, OV2_ERR_NOMEM, "BA pinned mirror hipHostMalloc(%zu): %s", want, hipGetErrorString(he));
            c->ba_host_cap = want;
        }
    }
    // OV2_BA_TRACE=1: host-side wall time of the phases on stderr (diagnostics)
    static const bool trace = getenv("OV2_BA_TRACE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t0 = now();
    ov2_status s = upload_batch(S);
    if (s != OV2_OK) return s;
    const auto t0b = now();
    if ((s = build_program(S)) != OV2_OK) return s;
    const auto t1 = now();
    // ---- robust solve + flags, no synchronisation in between
    if ((s = enqueue_minimize(S, o->max_iters)) != OV2_OK) return s;
    if ((s = enqueue_flags(S, 1)) != OV2_OK) return s;
    const auto t1b = now();
    if ((s = fetch_windows(S)) != OV2_OK) return s;
    const auto t2 = now();
    int n_l2 = 0;
    std::vector<uint8_t> did_l2((size_t)B, 0);
    for (int w = 0; w < B; ++w) {
        ba_win &X = S.hW[w];
        ov2_ba_result &Rw = R[w];
        Rw.initial_cost = X.initial_cost; Rw.final_cost = X.minimum_cost; Rw.termination = X.termination;
        Rw.n_log_robust = X.n_log;
        Rw.n_outliers_pass1 = X.nbad;
        const bool l2 = o->l2_refine && use_loss && X.nbad > 0;
        X.skip = l2 ? 0 : 1;
        if (l2) {
            ++n_l2;
            did_l2[w] = 1;
            X.use_loss = !(X.n_left > 0 && X.n_right > 0);   // src/optimizer.cpp:606-608
            X.max_iters = o->l2_max_iters;
            if (X.nbad >= X.row1 - X.row0) {   // every residual block was flagged: nothing left to minimise
                X.skip = 2; X.done = 1; X.termination = OV2_BA_TERM_SKIPPED;
                X.initial_cost = X.minimum_cost = X.x_cost = 0.0;
                X.active = X.valid = X.accepted = 0; X.eval_at_cand = 0;
            }
            X.nbad = X.n_left = X.n_right = 0;
        }
    }
    auto t3 = t2, t4 = t2, t5 = t2;
    if (n_l2 > 0) {
        // ---- second solve of the windows that lost residual blocks: program rebuilt from the device-side active flags
        ba_win *h_W = (ba_win *)c->ba_host;
        memcpy(h_W, S.hW.data(), sizeof(ba_win) * B);
        OV2_HIP(c, hipMemcpyAsync(S.W, h_W, sizeof(ba_win) * B, hipMemcpyHostToDevice, c->stream));
        // the program (row order, Schur structure, work arrays) is the robust pass's: nothing is rebuilt or re-sorted
        if (S.d.n_rows > 0 && S.d.n_e > 0) {
            S.d.masked = 1;
            OV2_HIP(c, hipMemsetAsync(S.d.f_live, 0, (size_t)S.d.n_f + 1, c->stream));
            BA_LAUNCH(S, K_MISC, ba_live_kernel, dim3((S.d.n_e + 255) / 256), dim3(256), 0, c->stream, S.d);
        }
        t3 = now();
        if ((s = enqueue_minimize(S, o->l2_max_iters)) != OV2_OK) return s;
        if ((s = enqueue_flags(S, 2)) != OV2_OK) return s;
        t4 = now();
    }
    // ---- results: window records, states, per-residual outputs
    {
        const size_t n = (size_t)tn, L = (size_t)tl, NP = (size_t)tp, e = (size_t)S.e;
        char *hb = (char *)c->ba_host;
        size_t off = 0;
        auto carve = [&](size_t bytes) { char *p = hb + off; off += (bytes + 255) / 256 * 256; return p; };
        ba_win *h_W = (ba_win *)carve(sizeof(ba_win) * B);
        double *h_xp = (double *)carve(7 * NP * 8), *h_xl = (double *)carve(e * L * 8);
        const bool want_chi2 = S.want_chi2, want_depth = S.want_depth, want_out = S.want_out;
        double *h_chi2 = want_chi2 ? (double *)carve(n * 8) : nullptr;
        unsigned char *h_depth = want_depth ? (unsigned char *)carve(n) : nullptr, *h_out = want_out ? (unsigned char *)carve(n) : nullptr;
        if (off > c->ba_host_cap) return ov2_set_err(c, OV2_ERR_NOMEM, "BA pinned mirror too small for the results");
        if (S.meas_pending) { OV2_HIP(c, hipStreamWaitEvent(c->stream, c->ba_copy_ev[1], 0)); S.meas_pending = false; }   // a batch without residual blocks
        if (n_l2 > 0) OV2_HIP(c, hipMemcpyAsync(h_W, S.W, sizeof(ba_win) * B, hipMemcpyDeviceToHost, c->stream));
        if (S.on_device) {
            // states and per-residual outputs go back to the windows' own device arrays; only the window records travel
            BA_LAUNCH(S, K_MISC, bb_scatter_kernel, dim3(32, B), dim3(256), 0, c->stream, S.gtab, S.xp, S.xl, S.e, S.chi2_dev, S.depth_dev,
                      S.outlier_dev);
        } else {
            if (NP) OV2_HIP(c, hipMemcpyAsync(h_xp, S.xp, 7 * NP * 8, hipMemcpyDeviceToHost, c->stream));
            if (L) OV2_HIP(c, hipMemcpyAsync(h_xl, S.xl, e * L * 8, hipMemcpyDeviceToHost, c->stream));
            if (h_chi2 && n) OV2_HIP(c, hipMemcpyAsync(h_chi2, S.chi2_dev, n * 8, hipMemcpyDeviceToHost, c->stream));
            if (h_depth && n) OV2_HIP(c, hipMemcpyAsync(h_depth, S.depth_dev, n, hipMemcpyDeviceToHost, c->stream));
            if (h_out && n) OV2_HIP(c, hipMemcpyAsync(h_out, S.outlier_dev, n, hipMemcpyDeviceToHost, c->stream));
        }
        OV2_HIP(c, hipStreamSynchronize(c->stream));
        t5 = now();
        for (int w = 0; w < B; ++w) {
            const ov2_ba_problem &Q = P[w];
            ov2_ba_result &Rw = R[w];
            const bool l2 = did_l2[w] != 0;
            const ba_win &X = n_l2 > 0 ? h_W[w] : S.hW[w];
            if (l2) {
                Rw.l2_done = 1;
                Rw.l2_initial_cost = X.initial_cost; Rw.l2_final_cost = X.minimum_cost; Rw.l2_termination = X.termination;
                Rw.n_outliers_pass2 = X.nbad;
            }
            Rw.n_log = std::min<int>(X.n_log, OV2_BA_MAX_LOG);
            memcpy(Rw.log, X.log, sizeof(ov2_ba_iter) * (size_t)Rw.n_log);
            if (S.on_device) continue;
            const size_t r0 = (size_t)S.res_off[w], l0 = (size_t)S.lm_off[w], p0 = (size_t)S.pose_off[w];
            // write back the non-constant blocks ("parameters_"; constant ones come back unchanged)
            if (Q.n_pose) memcpy(Q.pose, h_xp + 7 * p0, sizeof(double) * 7 * (size_t)Q.n_pose);
            if (Q.n_lm) memcpy(Q.lm, h_xl + e * l0, sizeof(double) * e * (size_t)Q.n_lm);
            if (Rw.chi2 && Q.n_res) memcpy(Rw.chi2, h_chi2 + r0, sizeof(double) * (size_t)Q.n_res);
            if (Rw.depth_positive && Q.n_res) memcpy(Rw.depth_positive, h_depth + r0, (size_t)Q.n_res);
            if (Rw.outlier && Q.n_res) memcpy(Rw.outlier, h_out + r0, (size_t)Q.n_res);
        }
    }
    if (trace)
        fprintf(stderr, "[ov2_ba_solve_batch] B %d | upload %.2f | build %.2f | enqueue robust %.2f | sync %.2f | rebuild %.2f (%d windows) | "
                        "enqueue L2 %.2f | results %.2f ms\n", B, ms(t0, t0b), ms(t0b, t1), ms(t1, t1b), ms(t1b, t2), ms(t2, t3), n_l2,
                ms(t3, t4), ms(t4, t5));
    return OV2_OK;
}
