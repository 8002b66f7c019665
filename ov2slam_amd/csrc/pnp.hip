// pnp.hip -- motion-only bundle adjustment (one 6-dof pose, fixed 3D points) on gfx950, whole LM loop on device.
//
// Replaces (reference, /root/reference): MultiViewGeometry::ceresPnP src/multi_view_geometry.cpp:492-586, called once
// per frame from VisualFrontEnd::computePose (src/visual_front_end.cpp:791) -- SURVEY.md section 8f, "next" row 1.
// Cost functor ReprojectionErrorSE3 (src/ceres_parametrization.cpp:301-358), Huber(sqrt(chi2th)) + Ceres corrector,
// Jacobi scaling, Levenberg-Marquardt with the Ceres radius rules (restated in oracle/ov2_oracle_pnp.c), chi2 flags,
// optional L2 re-solve without the flagged points.
//
// One workgroup (256 threads) per frame, any number of frames per launch: every thread keeps the pose and the 6x6
// system in registers (wave-uniform), residuals are strided over the threads, the 28 sums (cost, J'r, upper J'J) are
// reduced in a fixed order (DPP row sums -> readlane -> LDS across the 4 waves), and the scalar LM logic is executed
// redundantly by all threads, so there is no host round trip and no divergence.  The jacobian is never stored.
#include "ov2_internal.h"

namespace {

struct pnp_params {
    int max_iters, use_robust, l2_after_robust, jacobi_scaling, max_invalid;
    double huber_a, chi2_th, ftol, initial_radius, max_radius, min_radius, min_diag, max_diag, min_rel, ptol;
};

__device__ __forceinline__ double readlane_d(double v, int lane)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | lo);
}

template <int CTRL>
__device__ __forceinline__ double dpp_d(double v)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | lo);
}

__device__ __forceinline__ double wave_sum_d(double v)
{
    v += dpp_d<0xB1>(v);
    v += dpp_d<0x4E>(v);
    v += dpp_d<0x141>(v);
    v += dpp_d<0x140>(v);
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}

// acc[K] of every thread -> block totals in every thread
template <int K>
__device__ __forceinline__ void block_sum(double *acc, double (*sh)[28])
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double w = wave_sum_d(acc[k]);
        if ((tid & 63) == 0) sh[tid >> 6][k] = w;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = (sh[0][k] + sh[1][k]) + (sh[2][k] + sh[3][k]);
    __syncthreads();
}

__device__ __forceinline__ void quat_R(const double *p, double R[9])
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

__device__ inline void se3_plus_d(const double *x, const double *d, double *out)
{
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
    const double a[7] = {0, 0, 0, imag * w[0], imag * w[1], imag * w[2], real};
    double Ra[9], V[9];
    {
        const double q[7] = {0, 0, 0, a[3], a[4], a[5], a[6]};
        // exp's quaternion is unit up to rounding; quat_R normalises, the oracle's quat_to_R does not: build it raw
        const double X = q[3], Y = q[4], Z = q[5], W = q[6];
        const double tx = 2 * X, ty = 2 * Y, tz = 2 * Z;
        const double twx = tx * W, twy = ty * W, twz = tz * W, txx = tx * X, txy = ty * X, txz = tz * X;
        const double tyy = ty * Y, tyz = tz * Y, tzz = tz * Z;
        Ra[0] = 1 - (tyy + tzz); Ra[1] = txy - twz;       Ra[2] = txz + twy;
        Ra[3] = txy + twz;       Ra[4] = 1 - (txx + tzz); Ra[5] = tyz - twx;
        Ra[6] = txz - twy;       Ra[7] = tyz + twx;       Ra[8] = 1 - (txx + tyy);
    }
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
    q[3] = a[6] * b[3] - a[3] * b[0] - a[4] * b[1] - a[5] * b[2];
    q[0] = a[6] * b[0] + a[3] * b[3] + a[4] * b[2] - a[5] * b[1];
    q[1] = a[6] * b[1] + a[4] * b[3] + a[5] * b[0] - a[3] * b[2];
    q[2] = a[6] * b[2] + a[5] * b[3] + a[3] * b[1] - a[4] * b[0];
    const double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int r = 0; r < 3; ++r)
        out[r] = (V[3 * r] * u[0] + V[3 * r + 1] * u[1] + V[3 * r + 2] * u[2]) +
                 (Ra[3 * r] * x[0] + Ra[3 * r + 1] * x[1] + Ra[3 * r + 2] * x[2]);
    out[3] = q[0] / nq; out[4] = q[1] / nq; out[5] = q[2] / nq; out[6] = q[3] / nq;
}

struct pnp_frame {
    int i0, i1;
    const double *unpx, *wpts;
    const int *scales;
    const unsigned char *removed;   // 1 = residual block removed (L2 pass)
    double K[4];
};

// acc = {cost, g[6], H upper triangle [21]} at pose x (JAC) or {cost} only
template <bool JAC>
__device__ inline void pnp_accumulate(const pnp_frame &F, const double *x, bool use_removed, int use_loss, double a,
                                      double *acc)
{
    double R[9];
    quat_R(x, R);
#pragma unroll
    for (int k = 0; k < (JAC ? 28 : 1); ++k) acc[k] = 0.0;
    for (int i = F.i0 + (int)threadIdx.x; i < F.i1; i += 256) {
        if (use_removed && F.removed[i]) continue;
        const double inv_sigma = 1.0 / (F.scales ? exp2((double)F.scales[i]) : 1.0);
        const double *wp = F.wpts + 3 * (size_t)i;
        const double d[3] = {wp[0] - x[0], wp[1] - x[1], wp[2] - x[2]};
        const double cam[3] = {R[0] * d[0] + R[3] * d[1] + R[6] * d[2], R[1] * d[0] + R[4] * d[1] + R[7] * d[2],
                               R[2] * d[0] + R[5] * d[1] + R[8] * d[2]};
        const double invz = 1.0 / cam[2];
        double r0 = inv_sigma * ((F.K[0] * cam[0] * invz + F.K[2]) - F.unpx[2 * (size_t)i]);
        double r1 = inv_sigma * ((F.K[1] * cam[1] * invz + F.K[3]) - F.unpx[2 * (size_t)i + 1]);
        const double chi2 = r0 * r0 + r1 * r1;
        double rho0 = chi2, rho1 = 1.0;
        if (use_loss) {
            const double bb = a * a;
            if (chi2 > bb) {
                const double rr = sqrt(chi2);
                rho0 = 2.0 * a * rr - bb;
                rho1 = fmax(2.2250738585072014e-308, a / rr);
            }
        }
        acc[0] += 0.5 * rho0;
        if (!JAC) continue;
        const double invz2 = invz * invz;
        const double Jc[6] = {invz * F.K[0], 0.0, -cam[0] * invz2 * F.K[0], 0.0, invz * F.K[1], -cam[1] * invz2 * F.K[1]};
        double J[12];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            double t[3];
            for (int c = 0; c < 3; ++c) t[c] = Jc[3 * q] * R[3 * c] + Jc[3 * q + 1] * R[3 * c + 1] + Jc[3 * q + 2] * R[3 * c + 2];
            J[6 * q + 0] = -inv_sigma * t[0]; J[6 * q + 1] = -inv_sigma * t[1]; J[6 * q + 2] = -inv_sigma * t[2];
            J[6 * q + 3] = inv_sigma * (t[1] * wp[2] - t[2] * wp[1]);
            J[6 * q + 4] = inv_sigma * (t[2] * wp[0] - t[0] * wp[2]);
            J[6 * q + 5] = inv_sigma * (t[0] * wp[1] - t[1] * wp[0]);
        }
        if (use_loss) {   // Huber: rho'' <= 0 -> corrector scales by sqrt(rho')
            const double s = sqrt(rho1);
#pragma unroll
            for (int k = 0; k < 12; ++k) J[k] *= s;
            r0 *= s; r1 *= s;
        }
        int t = 7;
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            acc[1 + p] += J[p] * r0 + J[6 + p] * r1;
#pragma unroll
            for (int q = p; q < 6; ++q) acc[t++] += J[p] * J[q] + J[6 + p] * J[6 + q];
        }
    }
}

__device__ inline bool chol6(const double *A, const double *b, double *x)
{
    double L[36];
    for (int i = 0; i < 36; ++i) L[i] = A[i];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = L[j * 6 + j];
        for (int k = 0; k < j; ++k) d -= L[j * 6 + k] * L[j * 6 + k];
        if (!(d > 0.0)) ok = false;
        d = sqrt(d > 0.0 ? d : 1.0);
        L[j * 6 + j] = d;
        for (int i = j + 1; i < 6; ++i) {
            double s = L[i * 6 + j];
            for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * L[j * 6 + k];
            L[i * 6 + j] = s / d;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * x[k];
        x[i] = s / L[i * 6 + i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = x[i];
        for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * x[k];
        x[i] = s / L[i * 6 + i];
    }
    return ok;
}

// TrustRegionMinimizer on one pose; T = best pose on return; returns termination code
// Te receives the pose of the LAST residual evaluation (the reference flags outliers from the cost functors' cached
// chi2err_ / isdepthpositive_, src/multi_view_geometry.cpp:559-571: x after an accepted last step, the candidate otherwise)
__device__ inline int pnp_minimize(const pnp_frame &F, double *T, const pnp_params &P, bool use_removed, int use_loss,
                                   double (*sh)[28], int *n_iter, double *Te)
{
    double x[7], acc[28], H[36], g[6], scale[6], diag[6];
    for (int c = 0; c < 7; ++c) { x[c] = T[c]; Te[c] = T[c]; }
    auto unpack = [&]() {
        int t = 7;
        for (int p = 0; p < 6; ++p) {
            g[p] = acc[1 + p];
            for (int q = p; q < 6; ++q) { H[p * 6 + q] = acc[t]; H[q * 6 + p] = acc[t]; ++t; }
        }
    };
    pnp_accumulate<true>(F, x, use_removed, use_loss, P.huber_a, acc);
    block_sum<28>(acc, sh);
    double x_cost = acc[0];
    unpack();
    for (int c = 0; c < 6; ++c) scale[c] = P.jacobi_scaling ? 1.0 / (1.0 + sqrt(H[c * 6 + c])) : 1.0;
    double minimum_cost = x_cost, x_norm = -1.0, radius = P.initial_radius, dec = 2.0;
    int reuse = 0, invalid = 0, iteration = 0, term = OV2_BA_TERM_MAX_ITER;
    *n_iter = 0;
    for (;;) {
        if (iteration >= P.max_iters) { term = OV2_BA_TERM_MAX_ITER; break; }
        if (radius <= P.min_radius) { term = OV2_BA_TERM_MIN_RADIUS; break; }
        ++iteration;
        *n_iter = iteration;
        double Hs[36], gs[6], A[36], y[6], step[6];
        for (int p = 0; p < 6; ++p) {
            gs[p] = g[p] * scale[p];
            for (int q = 0; q < 6; ++q) Hs[p * 6 + q] = H[p * 6 + q] * scale[p] * scale[q];
        }
        if (!reuse) for (int c = 0; c < 6; ++c) diag[c] = fmin(fmax(Hs[c * 6 + c], P.min_diag), P.max_diag);
        reuse = 1;
        for (int k = 0; k < 36; ++k) A[k] = Hs[k];
        for (int c = 0; c < 6; ++c) A[c * 6 + c] += diag[c] / radius;
        bool ok = chol6(A, gs, y);
        double model_change = 0.0;
        if (ok) {
            double sg = 0, shs = 0;
            for (int p = 0; p < 6; ++p) { step[p] = -y[p]; if (!isfinite(step[p])) ok = false; }
            for (int p = 0; p < 6; ++p) {
                sg += step[p] * gs[p];
                for (int q = 0; q < 6; ++q) shs += step[p] * Hs[p * 6 + q] * step[q];
            }
            model_change = -(sg + 0.5 * shs);
        }
        if (!ok || !(model_change > 0.0)) {
            if (++invalid >= P.max_invalid) { term = OV2_BA_TERM_FAILURE; break; }
            radius /= dec; dec *= 2.0;
            continue;
        }
        invalid = 0;
        double delta[6], cand[7];
        for (int c = 0; c < 6; ++c) delta[c] = step[c] * scale[c];
        se3_plus_d(x, delta, cand);
        double cacc[1];
        pnp_accumulate<false>(F, cand, use_removed, use_loss, P.huber_a, cacc);
        block_sum<1>(cacc, sh);
        const double cand_cost = cacc[0];
        for (int c = 0; c < 7; ++c) Te[c] = cand[c];
        double sn = 0;
        for (int c = 0; c < 7; ++c) sn += (x[c] - cand[c]) * (x[c] - cand[c]);
        if (sqrt(sn) <= P.ptol * (x_norm + P.ptol)) { term = OV2_BA_TERM_PTOL; break; }
        const double cost_change = x_cost - cand_cost;
        if (fabs(cost_change) <= P.ftol * x_cost) { term = OV2_BA_TERM_FTOL; break; }
        const double rel = cost_change / model_change;
        if (rel > P.min_rel) {
            x_norm = 0;
            for (int c = 0; c < 7; ++c) { x[c] = cand[c]; x_norm += x[c] * x[c]; }
            x_norm = sqrt(x_norm);
            pnp_accumulate<true>(F, x, use_removed, use_loss, P.huber_a, acc);
            block_sum<28>(acc, sh);
            x_cost = acc[0];
            unpack();
            radius = fmin(P.max_radius, radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3.0)));
            dec = 2.0; reuse = 0;
            if (x_cost < minimum_cost) { minimum_cost = x_cost; for (int c = 0; c < 7; ++c) T[c] = x[c]; }
        } else {
            radius /= dec; dec *= 2.0;
        }
    }
    return term;
}

__global__ __launch_bounds__(256) void pnp_kernel(int B, const int *__restrict__ off, const double *__restrict__ unpx,
                                                  const double *__restrict__ wpts, const int *__restrict__ scales,
                                                  const double *__restrict__ Kall, double *__restrict__ Twc,
                                                  unsigned char *__restrict__ outlier, unsigned char *__restrict__ removed,
                                                  int *__restrict__ success, int *__restrict__ iters, pnp_params P)
{
    __shared__ double sh[4][28];
    __shared__ int sh_nbad[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b >= B) return;
    pnp_frame F;
    F.i0 = off[b]; F.i1 = off[b + 1];
    F.unpx = unpx; F.wpts = wpts; F.scales = scales; F.removed = removed;
    for (int k = 0; k < 4; ++k) F.K[k] = Kall[4 * b + k];
    const int n = F.i1 - F.i0;
    double T[7];
    for (int c = 0; c < 7; ++c) T[c] = Twc[7 * b + c];
    int it1 = 0, it2 = 0;
    double Te[7];
    int term = pnp_minimize(F, T, P, false, P.use_robust, sh, &it1, Te);
    // chi2 / depth flags as the functors cached them at their last evaluation (:551-565)
    double R[9];
    quat_R(Te, R);
    int nbad = 0;
    for (int i = F.i0 + tid; i < F.i1; i += 256) {
        const double inv_sigma = 1.0 / (scales ? exp2((double)scales[i]) : 1.0);
        const double *wp = wpts + 3 * (size_t)i;
        const double d[3] = {wp[0] - Te[0], wp[1] - Te[1], wp[2] - Te[2]};
        const double cam[3] = {R[0] * d[0] + R[3] * d[1] + R[6] * d[2], R[1] * d[0] + R[4] * d[1] + R[7] * d[2],
                               R[2] * d[0] + R[5] * d[1] + R[8] * d[2]};
        const double invz = 1.0 / cam[2];
        const double r0 = inv_sigma * ((F.K[0] * cam[0] * invz + F.K[2]) - unpx[2 * (size_t)i]);
        const double r1 = inv_sigma * ((F.K[1] * cam[1] * invz + F.K[3]) - unpx[2 * (size_t)i + 1]);
        const bool bad = (r0 * r0 + r1 * r1 > P.chi2_th) || !(cam[2] > 0.0);
        outlier[i] = bad ? 1 : 0;
        removed[i] = (bad && P.l2_after_robust) ? 1 : 0;
        nbad += bad ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) nbad += __shfl_xor(nbad, o);
    if ((tid & 63) == 0) sh_nbad[tid >> 6] = nbad;
    __threadfence_block();
    __syncthreads();
    nbad = sh_nbad[0] + sh_nbad[1] + sh_nbad[2] + sh_nbad[3];
    __syncthreads();
    int ok = 1;
    if (nbad == n) ok = 0;   // every point flagged: return false, pose untouched (:567-569)
    else {
        if (P.l2_after_robust && nbad > 0) term = pnp_minimize(F, T, P, true, 0, sh, &it2, Te);
        if (tid < 7) Twc[7 * b + tid] = T[tid];
        ok = term != OV2_BA_TERM_FAILURE;
    }
    if (tid == 0) {
        success[b] = ok;
        if (iters) { iters[2 * b] = it1; iters[2 * b + 1] = it2; }
    }
}

}  // namespace

// device-resident, asynchronous form: every array already in HBM, nothing synchronised.
// d_off: B + 1 prefix offsets of the frames' points; d_removed: scratch of sum(n) bytes (may alias nothing else).
extern "C" ov2_status ov2_pnp_solve_batch_dev(ov2_ctx *c, int B, const int32_t *d_off, const double *d_unpx,
                                              const double *d_wpts, const int32_t *d_scales, const double *d_K,
                                              double *d_Twc, int max_iters, float chi2th, int use_robust,
                                              int l2_after_robust, uint8_t *d_outlier, uint8_t *d_removed,
                                              int32_t *d_success, int32_t *d_iters)
{
    if (!c) return OV2_ERR_INVALID;
    if (B < 0 || (B && (!d_off || !d_K || !d_Twc || !d_success || !d_outlier || !d_removed || !d_unpx || !d_wpts)))
        return ov2_set_err(c, OV2_ERR_INVALID, "null argument");
    if (B == 0) return OV2_OK;
    OV2_HIP(c, hipSetDevice(c->device));
    pnp_params P;
    ov2_ba_options o;
    ov2_ba_default_options(&o, chi2th);
    P.max_iters = max_iters; P.use_robust = use_robust ? 1 : 0; P.l2_after_robust = l2_after_robust ? 1 : 0;
    P.jacobi_scaling = o.jacobi_scaling; P.max_invalid = o.max_consecutive_invalid_steps;
    P.huber_a = o.huber_delta; P.chi2_th = o.chi2_th; P.ftol = o.function_tolerance;
    P.initial_radius = o.initial_radius; P.max_radius = o.max_radius; P.min_radius = o.min_radius;
    P.min_diag = o.min_lm_diagonal; P.max_diag = o.max_lm_diagonal; P.min_rel = o.min_relative_decrease;
    P.ptol = o.parameter_tolerance;
    OV2_LAUNCH(c, OV2_K_DETECT + 3, pnp_kernel, dim3(B), dim3(256), 0, c->stream, B, d_off, d_unpx, d_wpts, d_scales, d_K,
               d_Twc, d_outlier, d_removed, d_success, d_iters, P);
    OV2_HIP(c, hipGetLastError());
    return OV2_OK;
}

// host-pointer form: arguments staged through pinned memory (one copy in, one copy out, one synchronisation)
extern "C" ov2_status ov2_pnp_solve_batch(ov2_ctx *c, int B, const int *n_pts, const double *unpx, const double *wpts,
                                          const int *scales, const double *K, double *Twc, int max_iters, float chi2th,
                                          int use_robust, int l2_after_robust, uint8_t *outlier, int *success, int *iters)
{
    if (!c) return OV2_ERR_INVALID;
    if (B < 0 || (B && (!n_pts || !K || !Twc || !success))) return ov2_set_err(c, OV2_ERR_INVALID, "null argument");
    if (B == 0) return OV2_OK;
    size_t n = 0;
    for (int b = 0; b < B; ++b) {
        if (n_pts[b] < 0) return ov2_set_err(c, OV2_ERR_INVALID, "negative point count");
        n += (size_t)n_pts[b];
    }
    if (n && (!unpx || !wpts || !outlier)) return ov2_set_err(c, OV2_ERR_INVALID, "null point arrays");
    OV2_HIP(c, hipSetDevice(c->device));
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    // staging block: inputs first (one H2D), then the outputs (one D2H): [off | K | unpx | wpts | scales || T | ok | it | outlier | removed]
    const size_t o_off = 0, o_K = up(sizeof(int) * (B + 1)), o_un = o_K + up(sizeof(double) * 4 * B);
    const size_t o_wp = o_un + up(sizeof(double) * 2 * n), o_sc = o_wp + up(sizeof(double) * 3 * n);
    const size_t o_T = o_sc + up(sizeof(int) * n), o_ok = o_T + up(sizeof(double) * 7 * B), o_it = o_ok + up(sizeof(int) * B);
    const size_t o_out = o_it + up(sizeof(int) * 2 * B), o_rem = o_out + up(n), total = o_rem + up(n);
    char *hp = nullptr, *dp = nullptr;
    ov2_status s = ov2_staging(c, total, (void **)&hp, (void **)&dp);
    if (s != OV2_OK) return s;
    {
        int *off = (int *)(hp + o_off);
        off[0] = 0;
        for (int b = 0; b < B; ++b) off[b + 1] = off[b] + n_pts[b];
    }
    memcpy(hp + o_K, K, sizeof(double) * 4 * B);
    if (n) {
        memcpy(hp + o_un, unpx, sizeof(double) * 2 * n);
        memcpy(hp + o_wp, wpts, sizeof(double) * 3 * n);
        if (scales) memcpy(hp + o_sc, scales, sizeof(int) * n);
    }
    memcpy(hp + o_T, Twc, sizeof(double) * 7 * B);
    hipStream_t st = c->stream;
    OV2_HIP(c, hipMemcpyAsync(dp, hp, o_ok, hipMemcpyHostToDevice, st));   // inputs + initial poses
    s = ov2_pnp_solve_batch_dev(c, B, (const int32_t *)(dp + o_off), (const double *)(dp + o_un), (const double *)(dp + o_wp),
                                scales ? (const int32_t *)(dp + o_sc) : nullptr, (const double *)(dp + o_K),
                                (double *)(dp + o_T), max_iters, chi2th, use_robust, l2_after_robust,
                                (uint8_t *)(dp + o_out), (uint8_t *)(dp + o_rem), (int32_t *)(dp + o_ok),
                                iters ? (int32_t *)(dp + o_it) : nullptr);
    if (s != OV2_OK) return s;
    OV2_HIP(c, hipMemcpyAsync(hp + o_T, dp + o_T, o_rem - o_T, hipMemcpyDeviceToHost, st));   // poses, flags, counts, outliers
    OV2_HIP(c, hipStreamSynchronize(st));
    memcpy(Twc, hp + o_T, sizeof(double) * 7 * B);
    memcpy(success, hp + o_ok, sizeof(int) * B);
    if (iters) memcpy(iters, hp + o_it, sizeof(int) * 2 * B);
    if (n) memcpy(outlier, hp + o_out, n);
    return OV2_OK;
}
