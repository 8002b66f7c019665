"""Pose-graph half of the oracle (oracle/ov2_oracle_pg.c): LeftSE3RelativePoseError (src/ceres_parametrization.cpp:30-102)
and the LM solve, against closed forms, finite differences and an independent scipy solve.  The reference holds no fixture
for these functions: parity unpinned beyond these checks."""
import numpy as np
import pytest
from scipy.linalg import expm, logm
from scipy.optimize import least_squares

from ov2slam_amd import ba_types as T, synth_ba


def mat(p7):
    M = np.eye(4)
    M[:3, :3] = synth_ba.quat_to_rot(np.asarray(p7[3:]) / np.linalg.norm(p7[3:]))
    M[:3, 3] = p7[:3]
    return M


def pose7(M):
    return synth_ba.pose7(M[:3, :3], M[:3, 3])


def hat6(v):
    H = np.zeros((4, 4))
    H[:3, :3] = [[0, -v[5], v[4]], [v[5], 0, -v[3]], [-v[4], v[3], 0]]
    H[:3, 3] = v[:3]
    return H


def se3_log(M):
    L = np.real(logm(M))
    return np.array([L[0, 3], L[1, 3], L[2, 3], L[2, 1], L[0, 2], L[1, 0]])


def rand_pose(rng, scale=1.0):
    return expm(hat6(rng.normal(0, scale, 6) * np.array([1, 1, 1, 0.5, 0.5, 0.5])))


def test_edge_residual_is_the_se3_log(oracle):
    rng = np.random.default_rng(0)
    for _ in range(20):
        A, B, Mij = rand_pose(rng), rand_pose(rng), rand_pose(rng, 0.5)
        r, _, _ = oracle.pg_eval_edge(pose7(A), pose7(B), pose7(Mij), want_jac=False)
        want = se3_log(np.linalg.inv(B) @ A @ Mij)
        assert np.allclose(r, want, atol=1e-9)
    # consistent measurement: zero residual
    A, B = rand_pose(rng), rand_pose(rng)
    r, _, _ = oracle.pg_eval_edge(pose7(A), pose7(B), pose7(np.linalg.inv(A) @ B), want_jac=False)
    assert np.abs(r).max() < 1e-12


def test_edge_jacobians_match_finite_differences_at_small_error(oracle):
    """(I + J_c / 2) Adj is the first-order form of the exact jacobian: at a residual of 1e-3 the two agree to ~1e-6"""
    rng = np.random.default_rng(1)
    A, B = rand_pose(rng), rand_pose(rng)
    Mij = np.linalg.inv(A) @ B @ expm(hat6(rng.normal(0, 1e-3, 6)))
    r0, Ji, Jj = oracle.pg_eval_edge(pose7(A), pose7(B), pose7(Mij))
    h = 1e-6
    for side, J in ((0, Ji), (1, Jj)):
        for c in range(6):
            d = np.zeros(6)
            d[c] = h
            Ap = oracle.se3_plus(pose7(A), d) if side == 0 else pose7(A)      # SE3LeftParameterization::Plus
            Bp = oracle.se3_plus(pose7(B), d) if side == 1 else pose7(B)
            rp, _, _ = oracle.pg_eval_edge(Ap, Bp, pose7(Mij), want_jac=False)
            assert np.allclose((rp - r0) / h, J[:, c], atol=2e-5), (side, c)


def chain(rng, n, drift=0.01, loop=True):
    """n keyframes on an arc; odometry edges carry drift; pose 0 is constant; the loop edge ties the last to the first"""
    gt = [np.eye(4)]
    for k in range(1, n):
        gt.append(gt[-1] @ expm(hat6(np.array([0.5, 0.02, 0.0, 0.0, 0.1, 0.02]))))
    meas = [np.linalg.inv(gt[k - 1]) @ gt[k] @ expm(hat6(rng.normal(0, drift, 6))) for k in range(1, n)]
    est = [np.eye(4)]
    for M in meas:
        est.append(est[-1] @ M)          # dead reckoning
    ei, ej, Tij = list(range(n - 1)), list(range(1, n)), [pose7(M) for M in meas]
    if loop:
        ei.append(0); ej.append(n - 1); Tij.append(pose7(np.linalg.inv(gt[0]) @ gt[n - 1]))
    const = np.zeros(n, np.uint8)
    const[0] = 1
    return T.PgProblem(np.stack([pose7(M) for M in est]), const, ei, ej, np.stack(Tij)), gt


def test_consistent_chain_does_not_move(oracle):
    P, _ = chain(np.random.default_rng(2), 12, drift=0.01, loop=False)   # dead reckoning satisfies every odometry edge
    x0 = P.pose.copy()
    R = oracle.pose_graph_solve(P)
    assert R.initial_cost < 1e-20 and R.final_cost < 1e-20
    assert np.allclose(P.pose, x0, atol=1e-12)


def test_loop_closure_against_scipy(oracle):
    P, gt = chain(np.random.default_rng(3), 12, drift=0.01)
    Q = P.copy()
    o = oracle.pg_default_options(max_iters=50, function_tolerance=1e-10)
    R = oracle.pose_graph_solve(P, o)
    assert R.final_cost < 0.2 * R.initial_cost
    loop0 = np.linalg.norm(mat(Q.pose[-1])[:3, 3] - gt[-1][:3, 3])
    loop1 = np.linalg.norm(mat(P.pose[-1])[:3, 3] - gt[-1][:3, 3])
    assert loop1 < 0.3 * loop0
    assert np.array_equal(P.pose[0], Q.pose[0])

    free = [k for k in range(len(Q.pose)) if not Q.pose_const[k]]

    def fun(x):   # the same residuals, parameters = left perturbations of the initial poses
        poses = [mat(p) for p in Q.pose]
        for n, k in enumerate(free):
            poses[k] = expm(hat6(x[6 * n:6 * n + 6])) @ poses[k]
        out = []
        for i, j, t in zip(Q.edge_i, Q.edge_j, Q.T_ij):
            out.append(se3_log(np.linalg.inv(poses[j]) @ poses[i] @ mat(t)))
        return np.concatenate(out)
    S = least_squares(fun, np.zeros(6 * len(free)), method="lm", xtol=1e-12, ftol=1e-12)
    assert R.final_cost == pytest.approx(0.5 * float(S.fun @ S.fun), rel=2e-3)   # the reference's jacobian is approximate: same valley
