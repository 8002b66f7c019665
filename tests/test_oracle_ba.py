"""CPU tests of the BA oracle (no GPU): pinned against the known answers the vendored Ceres 2.0.0 tests hold as
text (tests/golden/ceres_known_answers.json), finite differences, an independent numpy/scipy restatement, and
self-consistency (zero-noise windows converge to ground truth)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from ov2slam_amd import ba_types as T, synth_ba

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ceres_known_answers.json")))


def _dense_to_bs(A, n_e):
    """scalar-block (R=E=F=1) structure of a dense matrix whose first n_e columns are e blocks"""
    A = np.asarray(A, float)
    nr, nc = A.shape
    n_f = nc - n_e
    row_e = -np.ones(nr, np.int32)
    row_f = -np.ones((nr, n_f), np.int32)
    Je = np.zeros((nr, 1))
    Jf = np.zeros((nr, n_f, 1))
    for r in range(nr):
        es = [c for c in range(n_e) if A[r, c] != 0]
        assert len(es) <= 1
        if es:
            row_e[r], Je[r, 0] = es[0], A[r, es[0]]
        k = 0
        for c in range(n_e, nc):
            if A[r, c] != 0:
                row_f[r, k], Jf[r, k, 0] = c - n_e, A[r, c]
                k += 1
    return row_e, row_f, Je, Jf, n_f


def test_ceres_lls_problem1_schur_known_answer(oracle):
    g = GOLD["lls_problem1"]
    A, b = np.array(g["A"], float), np.array(g["b"], float)
    assert np.allclose(A.T @ A, g["AtA"])          # the printed A'A belongs to the printed A
    row_e, row_f, Je, Jf, n_f = _dense_to_bs(A, g["num_eliminate_blocks"])
    rc, S, rhs, x = oracle.schur_solve(1, 1, 1, row_e, row_f, Je, Jf, b, 2, n_f, D=None)
    assert rc == 0
    assert np.allclose(S, g["S"], atol=6e-5)       # printed to 4 decimals
    assert np.allclose(rhs, g["r"], atol=6e-5)
    assert np.allclose(x[2:], g["S_solve_r"], atol=6e-5)
    assert np.allclose(x, g["A_solve_b"], atol=6e-5)
    assert np.allclose(x, np.linalg.lstsq(A, b, rcond=None)[0], atol=1e-12)


def test_ceres_lls_problem0_with_D(oracle):
    g = GOLD["lls_problem0"]
    A, b, D = np.array(g["A"], float), np.array(g["b"], float), np.array(g["D"], float)
    row_e, row_f, Je, Jf, n_f = _dense_to_bs(A, 0)
    rc, _, _, x = oracle.schur_solve(1, 1, 1, row_e, row_f, Je, Jf, b, 0, n_f)
    assert rc == 0 and np.allclose(x, g["x"], atol=1e-12)
    rc, _, _, xd = oracle.schur_solve(1, 1, 1, row_e, row_f, Je, Jf, b, 0, n_f, D=D)
    assert rc == 0 and np.allclose(xd, g["x_D"], atol=1e-8)
    # first column eliminated instead (e block) must give the same regularised solution
    row_e, row_f, Je, Jf, n_f = _dense_to_bs(A, 1)
    rc, _, _, xe = oracle.schur_solve(1, 1, 1, row_e, row_f, Je, Jf, b, 1, n_f, D=D)
    assert rc == 0 and np.allclose(xe, g["x_D"], atol=1e-8)


def test_schur_matches_dense_random_blocks(oracle):
    rng = np.random.default_rng(3)
    R, E, F, n_e, n_f, maxf = 2, 3, 6, 7, 4, 2
    rows = []
    for e in range(n_e):
        for _ in range(int(rng.integers(2, 6))):
            fs = rng.choice(n_f, size=int(rng.integers(0, 3)), replace=False)
            rows.append((e, list(fs)))
    rows.append((-1, [0, 2]))
    nr = len(rows)
    row_e = np.array([r[0] for r in rows], np.int32)
    row_f = -np.ones((nr, maxf), np.int32)
    Je, Jf, b = rng.normal(size=(nr, R, E)), rng.normal(size=(nr, maxf, R, F)), rng.normal(size=(nr, R))
    A = np.zeros((nr * R, n_e * E + n_f * F))
    for i, (e, fs) in enumerate(rows):
        if e >= 0:
            A[i * R:(i + 1) * R, e * E:(e + 1) * E] = Je[i]
        for k, f in enumerate(fs):
            row_f[i, k] = f
            A[i * R:(i + 1) * R, n_e * E + f * F:n_e * E + (f + 1) * F] = Jf[i, k]
    D = rng.uniform(0.1, 1.0, A.shape[1])
    rc, S, rhs, x = oracle.schur_solve(R, E, F, row_e, row_f, Je, Jf, b, n_e, n_f, D=D)
    assert rc == 0
    H = A.T @ A + np.diag(D ** 2)
    g = A.T @ b.ravel()
    ne = n_e * E
    S_ref = H[ne:, ne:] - H[ne:, :ne] @ np.linalg.solve(H[:ne, :ne], H[:ne, ne:])
    assert np.allclose(S, S_ref, atol=1e-10)
    assert np.allclose(rhs, g[ne:] - H[ne:, :ne] @ np.linalg.solve(H[:ne, :ne], g[:ne]), atol=1e-10)
    assert np.allclose(x, np.linalg.solve(H, g), atol=1e-10)


def test_corrector_and_huber_known_answers(oracle):
    import ctypes as C
    L = oracle._ba_lib()
    L.ov2o_corrector.argtypes = [C.c_double, oracle.f64p, C.c_int, oracle.f64p, C.c_int, C.POINTER(oracle.f64p), oracle.i32p]
    for c in GOLD["corrector_scalar"]["cases"]:
        res, jac = np.array([c["residual"]]), np.array([c["jacobian"]])
        rho = np.array(c["rho"], float)
        jp = (oracle.f64p * 1)(jac.ctypes.data_as(oracle.f64p))
        nc = np.array([1], np.int32)
        L.ov2o_corrector(res[0] ** 2, rho.ctypes.data_as(oracle.f64p), 1, res.ctypes.data_as(oracle.f64p), 1, jp,
                         nc.ctypes.data_as(oracle.i32p))
        assert abs(res[0] - c["expected_residual"]) < 1e-6 and abs(jac[0] - c["expected_jacobian"]) < 1e-6
    # HuberLoss (loss_function.cc:48-62): finite-difference recipe of loss_function_test.cc
    a = 1.3
    for s in (0.2, 1.0, 1.69, 1.7, 5.0, 40.0):
        rho = oracle.huber(a, s)
        h = 1e-6 * max(s, 1)
        fd1 = (oracle.huber(a, s + h)[0] - oracle.huber(a, s - h)[0]) / (2 * h)
        assert abs(rho[1] - fd1) < 1e-5
        if abs(s - a * a) > 0.1:
            fd2 = (oracle.huber(a, s + h)[1] - oracle.huber(a, s - h)[1]) / (2 * h)
            assert abs(rho[2] - fd2) < 1e-5


def test_lm_radius_and_diagonal_known_answers(oracle):
    """levenberg_marquardt_strategy_test.cc:81-150 replayed on the C functions the oracle's minimize() calls for its
    radius / LM-diagonal updates (ov2o_lm_step_accepted / _rejected / ov2o_lm_diagonal)."""
    L = oracle.lib()
    dp = C.POINTER(C.c_double)
    L.ov2o_lm_step_accepted.argtypes = [dp, dp, C.c_double, C.c_double]
    L.ov2o_lm_step_rejected.argtypes = [dp, dp]
    L.ov2o_lm_diagonal.argtypes = [C.c_int, dp, C.c_double, C.c_double, C.c_double, dp, dp]
    g = GOLD["lm_radius_sequence"]
    radius, dec = C.c_double(g["initial_radius"]), C.c_double(2.0)
    for kind, q, expect in g["events"]:
        if kind == "reject":
            L.ov2o_lm_step_rejected(C.byref(radius), C.byref(dec))
        else:
            L.ov2o_lm_step_accepted(C.byref(radius), C.byref(dec), q, g["max_radius"])
            assert dec.value == 2.0
        assert radius.value == pytest.approx(expect, rel=1e-15)
    d = GOLD["lm_diagonal"]
    J = np.array(d["jacobian"], float)
    cn = np.ascontiguousarray((J ** 2).sum(0))
    diag, D = np.zeros(len(cn)), np.zeros(len(cn))
    L.ov2o_lm_diagonal(len(cn), cn.ctypes.data_as(dp), d["min_lm_diagonal"], d["max_lm_diagonal"], d["radius"], diag.ctypes.data_as(dp),
                       D.ctypes.data_as(dp))
    assert np.allclose(D, d["expected_D"])
    # reuse_diagonal: the stored diagonal with a new radius
    L.ov2o_lm_diagonal(len(cn), None, d["min_lm_diagonal"], d["max_lm_diagonal"], 4 * d["radius"], diag.ctypes.data_as(dp), D.ctypes.data_as(dp))
    assert np.allclose(D, np.array(d["expected_D"]) / 2)


def test_se3_exp_against_matrix_exponential(oracle):
    from scipy.linalg import expm
    rng = np.random.default_rng(0)
    for d in list(rng.normal(0, 0.7, size=(20, 6))) + [np.zeros(6), np.array([1, 2, 3, 1e-12, 0, 0.0]),
                                                         np.array([0.1, 0, 0, 0, 0, 3.1])]:
        out = oracle.se3_exp(d)
        w = d[3:]
        M = np.zeros((4, 4))
        M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
        M[:3, 3] = d[:3]
        E = expm(M)
        assert np.allclose(synth_ba.quat_to_rot(out[3:]), E[:3, :3], atol=1e-12)
        assert np.allclose(out[:3], E[:3, 3], atol=1e-12)
    # left update and group law: Plus(Plus(x, a), -a) for commuting (parallel) tangents is the identity
    x = np.concatenate([[1.0, -2.0, 0.5], synth_ba.rot_to_quat(synth_ba.se3_exp(np.array([0, 0, 0, 0.3, -0.2, 0.9]))[0])])
    a = np.array([0.02, -0.01, 0.03, 0.01, 0.02, -0.015])
    back = oracle.se3_plus(oracle.se3_plus(x, a), -a)
    assert np.allclose(back, x, atol=1e-14)


@pytest.mark.parametrize("inv_depth", [True, False])
def test_jacobians_match_finite_differences(oracle, inv_depth):
    P = synth_ba.make_window(8, 120, inv_depth=inv_depth, seed=5)
    h = 1e-6
    seen = set()
    for i in range(P.n_res):
        t = int(P.res_type[i])
        if t in seen and i > 40:
            continue
        seen.add(t)
        ev = oracle.ba_eval_residual(P, i)
        k, l = int(P.res_pose[i]), int(P.res_lm[i])

        def fd_pose(idx):
            J = np.zeros((2, 6))
            for c in range(6):
                d = np.zeros(6)
                d[c] = h
                pp, pm = P.pose.copy(), P.pose.copy()
                pp[idx], pm[idx] = oracle.se3_plus(P.pose[idx], d), oracle.se3_plus(P.pose[idx], -d)
                J[:, c] = (oracle.ba_eval_residual(P, i, pp, None, False)["r"] -
                           oracle.ba_eval_residual(P, i, pm, None, False)["r"]) / (2 * h)
            return J
        if t != T.RANCH_INV:
            assert np.allclose(ev["Jk"], fd_pose(k), rtol=1e-6, atol=1e-5)
        if t in (T.L_INV, T.R_INV):
            assert np.allclose(ev["Ja"], fd_pose(int(P.lm_anchor_pose[l])), rtol=1e-6, atol=1e-5)
        e = P.lm.shape[1]
        J = np.zeros((2, e))
        for c in range(e):
            hh = h * max(abs(P.lm[l, c]), 1e-3)
            lp, lm = P.lm.copy(), P.lm.copy()
            lp[l, c] += hh
            lm[l, c] -= hh
            J[:, c] = (oracle.ba_eval_residual(P, i, None, lp, False)["r"] -
                       oracle.ba_eval_residual(P, i, None, lm, False)["r"]) / (2 * hh)
        assert np.allclose(ev["Jl"], J, rtol=1e-5, atol=1e-4)
    assert len(seen) == (3 if inv_depth else 2)


def _numpy_residuals(P, poses, lms):
    """independent numpy restatement of the five cost functors (src/ceres_parametrization.cpp) for cross-checking"""
    out = np.zeros((P.n_res, 2))
    Rrl, trl = synth_ba.quat_to_rot(P.T_rl[3:]), P.T_rl[:3]
    for i in range(P.n_res):
        t, k, l = int(P.res_type[i]), int(P.res_pose[i]), int(P.res_lm[i])
        if P.inv_depth:
            z = 1.0 / lms[l, 0]
            u, v = P.lm_anchor_uv[l]
            anch = z * np.array([(u - P.calib_l[2]) / P.calib_l[0], (v - P.calib_l[3]) / P.calib_l[1], 1.0])
            a = int(P.lm_anchor_pose[l])
            Xw = synth_ba.quat_to_rot(poses[a, 3:]) @ anch + poses[a, :3]
        else:
            Xw = lms[l]
        if t == T.RANCH_INV:
            Xc = Rrl @ anch + trl
            K = P.calib_r
        else:
            Xc = synth_ba.quat_to_rot(poses[k, 3:]).T @ (Xw - poses[k, :3])
            K = P.calib_l
            if t in (T.R_XYZ, T.R_INV):
                Xc = Rrl @ Xc + trl
                K = P.calib_r
        out[i] = [K[0] * Xc[0] / Xc[2] + K[2] - P.res_uv[i, 0], K[1] * Xc[1] / Xc[2] + K[3] - P.res_uv[i, 1]]
    return out


@pytest.mark.parametrize("inv_depth", [True, False])
def test_residuals_match_numpy_restatement(oracle, inv_depth):
    P = synth_ba.make_window(10, 200, inv_depth=inv_depth, seed=11)
    ref = _numpy_residuals(P, P.pose, P.lm)
    got = np.array([oracle.ba_eval_residual(P, i, want_jac=False)["r"] for i in range(P.n_res)])
    assert np.allclose(got, ref, rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("inv_depth", [True, False])
def test_zero_noise_window_converges_to_ground_truth(oracle, inv_depth):
    P, gt = synth_ba.make_window(12, 400, inv_depth=inv_depth, seed=3, px_noise=0.0, outlier_frac=0.0, return_gt=True)
    o = oracle.ba_default_options()
    o.max_iters, o.function_tolerance = 30, 1e-12
    R = oracle.ba_solve(P, o)
    assert R.c.final_cost < 1e-12 * max(R.c.initial_cost, 1.0) + 1e-10
    free = P.pose_const == 0
    assert np.abs(P.pose[free, :3] - gt["poses"][free, :3]).max() < 1e-6
    assert np.median(np.abs(P.lm - gt["lm"]) / np.abs(gt["lm"])) < 1e-6
    assert R.c.n_outliers_pass1 == 0 and not R.c.l2_done
    assert np.array_equal(P.pose[~free], gt["poses"][~free])       # constant blocks untouched


@pytest.mark.parametrize("inv_depth", [True, False])
def test_lm_solution_matches_scipy_least_squares(oracle, inv_depth):
    """converged L2 optimum of a small noisy window == scipy's trust-region solution of the same cost"""
    from scipy.optimize import least_squares
    P = synth_ba.make_window(6, 60, inv_depth=inv_depth, seed=21, outlier_frac=0.0)
    P0 = P.copy()
    o = oracle.ba_default_options()
    o.huber_delta, o.max_iters, o.function_tolerance, o.l2_refine = 0.0, 60, 1e-15, 0
    o.chi2_th = 1e30
    R = oracle.ba_solve(P, o)
    free = np.nonzero(P0.pose_const == 0)[0]
    e = P0.lm.shape[1]

    def fun(x):
        poses = P0.pose.copy()
        for j, k in enumerate(free):
            poses[k] = oracle.se3_plus(P0.pose[k], x[6 * j:6 * j + 6])
        lms = P0.lm + x[6 * len(free):].reshape(-1, e)
        return _numpy_residuals(P0, poses, lms).ravel()
    sol = least_squares(fun, np.zeros(6 * len(free) + P0.lm.size), method="trf", xtol=1e-14, ftol=1e-14, gtol=1e-12,
                        x_scale="jac")
    assert R.c.final_cost == pytest.approx(0.5 * (sol.fun ** 2).sum(), rel=1e-7)


def test_local_ba_flow_robust_then_l2(oracle):
    P, gt = synth_ba.make_window(20, 1500, inv_depth=True, seed=synth_ba.SEED_BA, return_gt=True)
    P0 = P.copy()
    R = oracle.ba_solve(P)
    s = R.summary()
    assert s["l2_done"] and s["outliers"][0] > 0.05 * P.n_res          # 5 % gross outliers per image
    assert R.c.final_cost < 0.5 * R.c.initial_cost
    assert set(np.unique(R.outlier)) <= {0, 1, 2}
    assert (R.chi2[R.outlier == 0] <= 5.9915 + 1e-12).all() and R.depth_positive[R.outlier == 0].all()
    free = P.pose_const == 0
    assert np.abs(P.pose[free, :3] - gt["poses"][free, :3]).max() < 0.25 * np.abs(P0.pose[free, :3] - gt["poses"][free, :3]).max()
    # the iteration log follows the Ceres accept/reject radius rule
    log = R.log[:R.c.n_log_robust]
    assert log[0]["radius"] == 1e4
    for a, b in zip(log[:-1], log[1:]):
        if b["ok"]:
            q = b["relative_decrease"]
            assert b["radius"] == pytest.approx(min(1e16, a["radius"] / max(1 / 3, 1 - (2 * q - 1) ** 3)))
