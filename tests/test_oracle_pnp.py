"""CPU tests of the ceresPnP restatement (oracle/ov2_oracle_pnp.c; reference src/multi_view_geometry.cpp:492-586).
No reference fixture covers this function ("parity unpinned" by golden data); it is pinned here against (1) an
independent numpy Levenberg-Marquardt written from the Ceres rules with NUMERIC jacobians, (2) scipy's least-squares
optimum on the inlier set, (3) the ground truth of the synthetic scene."""
import numpy as np
import pytest

from ov2slam_amd import synth_ba as S
from oracle import oracle_py as O


def _residuals(T, p, active, use_loss, a):
    R = S.quat_to_rot(T[3:])
    cam = (p["wpts"] - T[:3]) @ R
    sig = np.ones(len(cam)) if p["scales"] is None else 2.0 ** p["scales"]
    K = p["K"]
    r = np.stack([(K[0] * cam[:, 0] / cam[:, 2] + K[2] - p["unpx"][:, 0]) / sig,
                  (K[1] * cam[:, 1] / cam[:, 2] + K[3] - p["unpx"][:, 1]) / sig], 1)
    chi2 = (r ** 2).sum(1)
    w = np.ones(len(r))
    cost = 0.5 * chi2
    if use_loss:
        big = chi2 > a * a
        rr = np.sqrt(np.where(big, chi2, 1.0))
        cost = np.where(big, 0.5 * (2 * a * rr - a * a), cost)
        w = np.where(big, np.sqrt(a / rr), 1.0)
    return (r * w[:, None])[active].ravel(), cost[active].sum()


def _plus(T, d):
    dR, dt = S.se3_exp(d)
    return S.pose7(dR @ S.quat_to_rot(T[3:]), dR @ T[:3] + dt)


def _numpy_lm(p, T0, active, use_loss, a, max_iters=5, ftol=1e-3):
    """Ceres TrustRegionMinimizer + LevenbergMarquardtStrategy, numeric jacobian (central differences)."""
    # NB: the corrector scales r and J by sqrt(rho'), and d(sqrt(rho') r)/dx != sqrt(rho') dr/dx, so the numeric
    # jacobian must be of the UNWEIGHTED residual, weighted afterwards (Ceres linearises r, not rho).
    def lin(T):
        rw, cost = _residuals(T, p, active, use_loss, a)
        r0, _ = _residuals(T, p, active, False, a)
        w = np.where(np.abs(r0) > 0, rw / np.where(r0 == 0, 1, r0), 1.0)
        J = []
        for k in range(6):
            d = np.zeros(6)
            d[k] = 1e-6
            J.append((_residuals(_plus(T, d), p, active, False, a)[0] -
                      _residuals(_plus(T, -d), p, active, False, a)[0]) / 2e-6)
        return rw, np.stack(J, 1) * w[:, None], cost
    x = T0.copy()
    r, J, x_cost = lin(x)
    scale = 1.0 / (1.0 + np.sqrt((J ** 2).sum(0)))
    radius, dec, reuse, x_norm, it = 1e4, 2.0, False, -1.0, 0
    best, best_cost = x.copy(), x_cost
    while it < max_iters:
        it += 1
        Js = J * scale
        H, g = Js.T @ Js, Js.T @ r
        if not reuse:
            diag = np.clip(np.diag(H), 1e-6, 1e32)
        reuse = True
        step = -np.linalg.solve(H + np.diag(diag / radius), g)
        model = -(step @ g + 0.5 * step @ H @ step)
        cand = _plus(x, step * scale)
        cand_cost = _residuals(cand, p, active, use_loss, a)[1]
        if np.linalg.norm(x - cand) <= 1e-8 * (x_norm + 1e-8):
            break
        change = x_cost - cand_cost
        if abs(change) <= ftol * x_cost:
            break
        rel = change / model
        if rel > 1e-3:
            x = cand
            x_norm = np.linalg.norm(x)
            r, J, x_cost = lin(x)
            radius = min(1e16, radius / max(1 / 3, 1 - (2 * rel - 1) ** 3))
            dec, reuse = 2.0, False
            if x_cost < best_cost:
                best, best_cost = x.copy(), x_cost
        else:
            radius /= dec
            dec *= 2
    return best, it


@pytest.mark.parametrize("seed,scales", [(0, False), (1, True), (4, False)])
def test_matches_independent_numpy_lm(seed, scales):
    p = S.make_pnp(250, seed, with_scales=scales)
    a = float(np.sqrt(np.float32(5.9915)))
    ok, T, out, it = O.pnp_solve(p["unpx"], p["wpts"], p["K"], p["Twc0"], p["scales"], l2_after_robust=False)
    Tn, itn = _numpy_lm(p, p["Twc0"], np.ones(250, bool), True, a)
    assert ok and it[0] == itn and it[1] == 0
    assert np.abs(T - Tn).max() < 1e-6
    # second stage: L2 on the points that survive the flags
    ok2, T2, out2, it2 = O.pnp_solve(p["unpx"], p["wpts"], p["K"], p["Twc0"], p["scales"], l2_after_robust=True)
    assert np.array_equal(out, out2) and it2[0] == it[0]
    Tn2, itn2 = _numpy_lm(p, Tn, ~out, False, a)
    assert it2[1] == itn2 and np.abs(T2 - Tn2).max() < 1e-6


def test_converged_solution_is_the_least_squares_optimum():
    from scipy.optimize import least_squares
    p = S.make_pnp(400, 7, outlier_frac=0.15)
    ok, T, out, _ = O.pnp_solve(p["unpx"], p["wpts"], p["K"], p["Twc0"], max_iters=50)
    assert ok and np.array_equal(out, p["gt_outlier"])
    keep = ~out
    sol = least_squares(lambda d: _residuals(_plus(T, d), p, keep, False, 0.0)[0], np.zeros(6), xtol=1e-14, ftol=1e-14,
                        gtol=1e-14)
    # ftol 1e-3 stops the reference's solver early: the remaining step to the exact optimum is tiny
    assert np.abs(sol.x).max() < 2e-4
    assert np.abs(T[:3] - p["Twc_gt"][:3]).max() < 5e-3


def test_flags_depth_and_returns_false_when_everything_is_flagged():
    p = S.make_pnp(120, 3, behind=6)
    ok, T, out, _ = O.pnp_solve(p["unpx"], p["wpts"], p["K"], p["Twc0"])
    assert ok and out[:6].all() and np.array_equal(out, p["gt_outlier"])
    q = S.make_pnp(40, 5, outlier_frac=0.0)
    q["unpx"] = q["unpx"] + 200.0          # every observation off by 200 px
    ok, T, out, _ = O.pnp_solve(q["unpx"], q["wpts"], q["K"], q["Twc0"], max_iters=2)
    assert not ok and out.all() and np.array_equal(T, q["Twc0"])


def test_non_robust_and_empty():
    p = S.make_pnp(100, 9, outlier_frac=0.0)
    ok, T, out, it = O.pnp_solve(p["unpx"], p["wpts"], p["K"], p["Twc0"], use_robust=False)
    assert ok and not out.any() and it[1] == 0
    assert np.abs(T - p["Twc_gt"]).max() < 5e-3
