"""GPU parity tests (through the C ABI) of the local-BA solve against the CPU oracle.
Bar (north_star): pose / landmark states within 1e-4 relative of the CPU path at the same LM iteration count;
here also: identical iteration log (accept/reject, termination) and costs to 1e-9 relative, identical outlier flags."""
import numpy as np
import pytest

from ov2slam_amd import local_ba, synth_ba

pytestmark = pytest.mark.gpu
REL = 1e-4


def _compare(P_gpu, R_gpu, P_cpu, R_cpu, flags_exact=True):
    sg, sc = R_gpu.summary(), R_cpu.summary()
    assert sg["termination"] == sc["termination"] and sg["l2_done"] == sc["l2_done"]
    assert sg["iterations"] == sc["iterations"]
    assert R_gpu.c.n_log == R_cpu.c.n_log
    for a, b in zip(R_gpu.log, R_cpu.log):
        assert a["ok"] == b["ok"] and a["valid"] == b["valid"]
        assert a["cost"] == pytest.approx(b["cost"], rel=1e-9)
        assert a["radius"] == pytest.approx(b["radius"], rel=1e-6)
    assert R_gpu.c.final_cost == pytest.approx(R_cpu.c.final_cost, rel=1e-9)
    free = P_cpu.pose_const == 0
    # 1e-4 relative: translation vs its own magnitude floor of 1 m, quaternion components vs 1, landmarks vs value
    assert np.abs(P_gpu.pose[:, :3] - P_cpu.pose[:, :3]).max() <= REL * max(1.0, np.abs(P_cpu.pose[:, :3]).max())
    assert np.abs(P_gpu.pose[:, 3:] - P_cpu.pose[:, 3:]).max() <= REL
    assert np.array_equal(P_gpu.pose[~free], P_cpu.pose[~free])
    assert (np.abs(P_gpu.lm - P_cpu.lm) <= REL * np.maximum(np.abs(P_cpu.lm), 1e-3)).all()
    if flags_exact:
        assert np.array_equal(R_gpu.outlier, R_cpu.outlier)
        assert np.array_equal(R_gpu.depth_positive, R_cpu.depth_positive)
    else:   # residuals sitting within 1e-9 of the chi2 threshold may flip
        assert (R_gpu.outlier != R_cpu.outlier).mean() < 1e-3
    assert np.allclose(R_gpu.chi2, R_cpu.chi2, rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("inv_depth", [True, False])
@pytest.mark.parametrize("n_kf,n_lm", [(6, 80), (20, 2000)])
def test_local_ba_matches_oracle(ctx, oracle, inv_depth, n_kf, n_lm):
    P = synth_ba.make_window(n_kf, n_lm, inv_depth=inv_depth, seed=17 + n_kf)
    Pc = P.copy()
    Rg = local_ba.Optimizer(ctx).localBA(P)
    Rc = oracle.ba_solve(Pc)
    _compare(P, Rg, Pc, Rc)


@pytest.mark.parametrize("inv_depth", [True, False])
def test_local_ba_l2_only_and_mono(ctx, oracle, inv_depth):
    """buse_robust_cost = false (no loss, no refinement) and a mono window (no right-camera residuals)."""
    P = synth_ba.make_window(10, 500, inv_depth=inv_depth, seed=5, stereo=False, outlier_frac=0.02)
    Pc = P.copy()
    Rg = local_ba.Optimizer(ctx).localBA(P, buse_robust_cost=False)
    o = oracle.ba_default_options()
    o.huber_delta = 0.0
    Rc = oracle.ba_solve(Pc, o)
    _compare(P, Rg, Pc, Rc)


def test_local_ba_zero_noise_converges(ctx):
    P, gt = synth_ba.make_window(12, 400, inv_depth=True, seed=3, px_noise=0.0, outlier_frac=0.0, return_gt=True)
    o = local_ba.default_options()
    o.max_iters, o.function_tolerance = 30, 1e-12
    R = local_ba.Optimizer(ctx).localBA(P, options=o)
    free = P.pose_const == 0
    assert R.c.final_cost < 1e-10
    assert np.abs(P.pose[free, :3] - gt["poses"][free, :3]).max() < 1e-6
    assert np.median(np.abs(P.lm - gt["lm"]) / np.abs(gt["lm"])) < 1e-6


def test_local_ba_edge_cases(ctx, oracle):
    # every pose constant: structure-only problem (reduced camera system is empty)
    P = synth_ba.make_window(6, 100, inv_depth=True, seed=9)
    P.pose_const[:] = 1
    Pc = P.copy()
    Rg = local_ba.Optimizer(ctx).localBA(P)
    Rc = oracle.ba_solve(Pc)
    _compare(P, Rg, Pc, Rc)
    # empty problem
    E = synth_ba.make_window(4, 20, inv_depth=False, seed=1)
    E.res_type, E.res_pose, E.res_lm, E.res_uv = E.res_type[:0], E.res_pose[:0], E.res_lm[:0], E.res_uv[:0]
    R = local_ba.Optimizer(ctx).localBA(E)
    assert R.summary()["termination"] == "skipped"
    # invalid input is reported, not crashed on
    B = synth_ba.make_window(4, 20, inv_depth=False, seed=1)
    B.res_lm[0] = 10 ** 6
    with pytest.raises(Exception):
        local_ba.Optimizer(ctx).localBA(B)


def test_local_ba_multi_workgroup_cholesky_ragged(ctx, oracle):
    """68 keyframes: the reduced camera system (m = 6 x free poses >= 320) takes the right-looking multi-workgroup
    Cholesky (MFMA trailing updates) with a ragged last panel and ragged 16 x 16 tiles."""
    P = synth_ba.make_window(68, 5000, inv_depth=True, seed=61, max_obs=7)
    m = 6 * int((P.pose_const == 0).sum())
    assert m >= 320 and m % 32 != 0 and m % 16 != 0
    Pc = P.copy()
    Rg = local_ba.Optimizer(ctx).localBA(P)
    Rc = oracle.ba_solve(Pc)
    _compare(P, Rg, Pc, Rc, flags_exact=False)


@pytest.mark.parametrize("inv_depth", [True, False])
def test_local_ba_config4_window(ctx, oracle, inv_depth):
    """BASELINE config 4 size: 100 KFs / 20 k landmarks (~240 k residual blocks), parity at full size."""
    P = synth_ba.make_window(100, 20000, inv_depth=inv_depth, seed=synth_ba.SEED_BA, max_obs=7)
    Pc = P.copy()
    Rg = local_ba.Optimizer(ctx).localBA(P)
    Rc = oracle.ba_solve(Pc)
    _compare(P, Rg, Pc, Rc, flags_exact=False)


def test_local_ba_batch_matches_oracle_and_single(ctx, oracle):
    """ov2_ba_solve_batch on ragged windows (different sizes, one structure-only, one empty, one that needs no L2
    refinement): every window matches the oracle like a lone solve does, and is BITWISE what the same window gives when it
    is solved alone (a window's arithmetic does not depend on the batch it shares)."""
    Ps = [synth_ba.make_window(6, 80, inv_depth=True, seed=23),
          synth_ba.make_window(20, 2000, inv_depth=True, seed=37),
          synth_ba.make_window(12, 600, inv_depth=True, seed=3, outlier_frac=0.0, px_noise=0.05),   # no outliers: no L2 pass
          synth_ba.make_window(9, 300, inv_depth=True, seed=41, stereo=False),
          synth_ba.make_window(31, 3000, inv_depth=True, seed=43, max_obs=9)]
    allc = synth_ba.make_window(6, 100, inv_depth=True, seed=9)
    allc.pose_const[:] = 1                                      # structure-only window
    Ps.append(allc)
    E = synth_ba.make_window(4, 20, inv_depth=True, seed=1)     # empty window
    E.res_type, E.res_pose, E.res_lm, E.res_uv = E.res_type[:0], E.res_pose[:0], E.res_lm[:0], E.res_uv[:0]
    Ps.insert(2, E)
    singles = [p.copy() for p in Ps]
    oracles = [p.copy() for p in Ps]
    opt = local_ba.Optimizer(ctx)
    Rb = opt.localBA_batch(Ps)
    assert Rb[2].summary()["termination"] == "skipped" and Rb[2].c.n_log == 0
    for k, (P, R) in enumerate(zip(Ps, Rb)):
        Rs = opt.localBA(singles[k])
        assert np.array_equal(P.pose.view(np.uint64), singles[k].pose.view(np.uint64)), k
        assert np.array_equal(P.lm.view(np.uint64), singles[k].lm.view(np.uint64)), k
        assert R.c.n_log == Rs.c.n_log and R.c.final_cost == Rs.c.final_cost and R.c.l2_final_cost == Rs.c.l2_final_cost
        assert np.array_equal(R.outlier, Rs.outlier) and np.array_equal(R.chi2.view(np.uint64), Rs.chi2.view(np.uint64))
        if k == 2:
            continue
        Rc = oracle.ba_solve(oracles[k])
        _compare(P, R, oracles[k], Rc)
    assert not Rb[3].summary()["l2_done"] or Rb[3].c.n_outliers_pass1 > 0


def test_local_ba_batch_xyz_and_mixed_calibration(ctx, oracle):
    """XYZ parametrisation, per-window calibrations; mixing parametrisations in one batch is refused"""
    Ps = [synth_ba.make_window(8, 300, inv_depth=False, seed=51), synth_ba.make_window(15, 900, inv_depth=False, seed=52)]
    Ps[1].calib_l = Ps[1].calib_l * np.array([1.02, 0.99, 1.0, 1.0])
    Ps[1].calib_r = Ps[1].calib_r * np.array([0.98, 1.01, 1.0, 1.0])
    Pc = [p.copy() for p in Ps]
    Rb = local_ba.Optimizer(ctx).localBA_batch(Ps)
    for k in range(2):
        _compare(Ps[k], Rb[k], Pc[k], oracle.ba_solve(Pc[k]))
    with pytest.raises(Exception):
        local_ba.Optimizer(ctx).localBA_batch([synth_ba.make_window(6, 80, inv_depth=True, seed=1),
                                               synth_ba.make_window(6, 80, inv_depth=False, seed=1)])


def test_local_ba_long_tracks(ctx, oracle):
    """a landmark seen by every keyframe of a 100-keyframe window (round 1 refused more than 40 observing keyframes)"""
    P = synth_ba.make_window(100, 3000, inv_depth=True, seed=77, max_obs=100)
    nobs = np.bincount(P.res_lm[P.res_type != 4], minlength=len(P.lm))
    assert nobs.max() > 80
    Pc = P.copy()
    Rg = local_ba.Optimizer(ctx).localBA(P)
    Rc = oracle.ba_solve(Pc)
    _compare(P, Rg, Pc, Rc, flags_exact=False)


@pytest.mark.parametrize("inv_depth", [True, False])
def test_local_ba_batch_device_resident_is_bitwise_the_host_form(ctx, inv_depth):
    """ov2_ba_solve_batch_dev (arrays + per-residual outputs in device memory, gathered / scattered by kernels) against
    ov2_ba_solve_batch on the same ragged windows (one empty, one without outliers, one with per-residual sigmas):
    states, costs, logs, chi2 and flags bitwise equal -- the host form is what the oracle tests pin."""
    Ps = [synth_ba.make_window(6, 80, inv_depth=inv_depth, seed=23),
          synth_ba.make_window(20, 2000, inv_depth=inv_depth, seed=37),
          synth_ba.make_window(12, 600, inv_depth=inv_depth, seed=3, outlier_frac=0.0, px_noise=0.05),
          synth_ba.make_window(31, 3000, inv_depth=inv_depth, seed=43, max_obs=9)]
    Ps[1].res_sigma = (2.0 ** np.random.default_rng(5).integers(0, 3, Ps[1].n_res)).astype(np.float64)
    E = synth_ba.make_window(4, 20, inv_depth=inv_depth, seed=1)     # empty window
    E.res_type, E.res_pose, E.res_lm, E.res_uv = E.res_type[:0], E.res_pose[:0], E.res_lm[:0], E.res_uv[:0]
    Ps.insert(1, E)
    opt = local_ba.Optimizer(ctx)
    Ds = [local_ba.DeviceBaProblem(ctx, p) for p in Ps]
    Rd = opt.localBA_batch_dev(Ds)
    Rh = opt.localBA_batch(Ps)          # solves Ps in place (the device copies were taken before)
    for k, (D, P) in enumerate(zip(Ds, Ps)):
        pose, lm, chi2, depth, outl = D.download()
        assert np.array_equal(pose.view(np.uint64), P.pose.view(np.uint64)), k
        assert np.array_equal(lm.view(np.uint64), P.lm.view(np.uint64)), k
        a, b = Rd[k], Rh[k].c
        assert (a.n_log, a.n_log_robust, a.termination, a.l2_termination, a.l2_done) == \
               (b.n_log, b.n_log_robust, b.termination, b.l2_termination, b.l2_done), k
        assert (a.initial_cost, a.final_cost, a.l2_initial_cost, a.l2_final_cost) == \
               (b.initial_cost, b.final_cost, b.l2_initial_cost, b.l2_final_cost), k
        assert (a.n_outliers_pass1, a.n_outliers_pass2) == (b.n_outliers_pass1, b.n_outliers_pass2), k
        for i in range(a.n_log):
            assert a.log[i].cost == b.log[i].cost and a.log[i].radius == b.log[i].radius
        if P.n_res:
            assert np.array_equal(chi2.view(np.uint64), Rh[k].chi2.view(np.uint64)), k
            assert np.array_equal(depth, Rh[k].depth_positive) and np.array_equal(outl, Rh[k].outlier), k
    # a second solve from the re-set state gives the same answer (nothing of the first call survives in the arena)
    first = [D.download()[0] for D in Ds]
    for D in Ds:
        D.reset()
    opt.localBA_batch_dev(Ds)
    for k, D in enumerate(Ds):
        assert np.array_equal(D.download()[0].view(np.uint64), first[k].view(np.uint64)), k


def test_local_ba_l2_on_a_nearly_empty_program(ctx, oracle):
    """the L2 re-solve runs the robust pass's program with the flagged rows masked: (a) 566 of 573 residual blocks flagged --
    most landmark and pose blocks are left without a row, Ceres' reduced program would not contain them and the solve ends
    on the PARAMETER tolerance, whose norms must not see them; (b) every block flagged -- nothing to minimise, skipped."""
    P = synth_ba.make_window(6, 60, inv_depth=True, seed=5, outlier_frac=1.0)
    Pc = P.copy()
    Rg, Rc = local_ba.Optimizer(ctx).localBA(P), oracle.ba_solve(Pc)
    assert Rc.summary()["l2_termination"] == "parameter_tolerance" and Rc.c.n_outliers_pass1 > 0.9 * P.n_res
    _compare(P, Rg, Pc, Rc)
    assert Rg.summary()["l2_termination"] == "parameter_tolerance"
    assert Rg.c.l2_initial_cost == pytest.approx(Rc.c.l2_initial_cost, rel=1e-9)
    Q = synth_ba.make_window(6, 60, inv_depth=True, seed=5)
    Q.res_uv = Q.res_uv + 500.0
    Qc = Q.copy()
    Rg, Rc = local_ba.Optimizer(ctx).localBA(Q), oracle.ba_solve(Qc)
    assert Rc.c.n_outliers_pass1 == Q.n_res and Rc.summary()["l2_termination"] == "skipped"
    # (measurements 500 px off: the robust pass takes wild, rejected steps whose costs agree to 1e-8 only -- not the subject here)
    assert Rg.summary()["iterations"] == Rc.summary()["iterations"] and Rg.c.n_log == Rc.c.n_log
    assert Rg.c.final_cost == pytest.approx(Rc.c.final_cost, rel=1e-6) and Rg.c.n_outliers_pass1 == Q.n_res
    assert np.array_equal(Rg.outlier, Rc.outlier) and np.abs(Q.pose - Qc.pose).max() < 1e-4
    assert Rg.summary()["l2_termination"] == "skipped" and Rg.c.l2_done == Rc.c.l2_done
    assert (Rg.c.l2_initial_cost, Rg.c.l2_final_cost) == (0.0, 0.0)
    # and inside a batch, beside an ordinary window
    Ps = [synth_ba.make_window(6, 60, inv_depth=True, seed=5, outlier_frac=1.0), synth_ba.make_window(9, 300, inv_depth=True, seed=41)]
    Pcs = [p.copy() for p in Ps]
    Rb = local_ba.Optimizer(ctx).localBA_batch(Ps)
    for k in range(2):
        _compare(Ps[k], Rb[k], Pcs[k], oracle.ba_solve(Pcs[k]))
