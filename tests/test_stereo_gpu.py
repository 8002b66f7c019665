"""GPU parity tests (through the C ABI) of the left -> right stereo matching path (SURVEY.md 8a row a4,
MapManager::stereoMatching src/map_manager.cpp:367-611): fbKltTracking / the two-stage batching on (left pyramid,
right pyramid) with 8-40 px disparities, the SAD line search, the epipolar gate with the row snap / the Sampson distance.
Bar: status identical, positions bit-identical (float32 bit patterns), SAD priors bit-identical."""
import numpy as np
import pytest

from ov2slam_amd import frontend as fe, synth

pytestmark = pytest.mark.gpu

K = np.array([[458.654, 0, 367.215], [0, 457.296, 248.375], [0, 0, 1]])
TX = np.array([[0, 0, 0], [0, 0, 0.110074], [0, -0.110074, 0]])
F_RL = np.linalg.inv(K).T @ TX @ np.linalg.inv(K)      # pure x baseline (the synthetic pair is rectified-like)


@pytest.fixture(scope="module")
def pair(ctx, oracle, stream):
    L, R = stream.left(2), stream.right(2)
    gl, gr = fe.preprocess_image(ctx, L), fe.preprocess_image(ctx, R)
    ol, orr = oracle.Pyramid(oracle.clahe(L)), oracle.Pyramid(oracle.clahe(R))
    return gl, gr, ol, orr


@pytest.mark.parametrize("n", [2048, 4000])
def test_fb_klt_left_to_right(ctx, oracle, stream, pair, n):
    """FeatureTracker::fbKltTracking on (left, right): priors = true disparity + N(0, 1 px), 2 levels and full pyramid"""
    gl, gr, ol, orr = pair
    kps = synth.grid_keypoints(n, seed=31)
    gt = stream.stereo_gt(kps).astype(np.float32)
    pri = (gt + np.random.default_rng(5).normal(0, 1.0, gt.shape)).astype(np.float32)
    trk = fe.FeatureTracker(ctx, 30, 0.01)
    for nl, p in ((1, pri), (3, pri), (3, kps)):
        out, st = trk.fbKltTracking(gl, gr, 9, nl, 30.0, 0.5, kps, p)
        eo, es, _ = oracle.fb_klt_tracking(ol, orr, kps, p, 9, nl, 30.0, 0.5, 30, 0.01)
        assert np.array_equal(st, es.astype(bool))
        assert np.array_equal(out.view(np.uint32), eo.view(np.uint32))
    out, st = trk.fbKltTracking(gl, gr, 9, 1, 30.0, 0.5, kps, pri)
    assert st.mean() > 0.9 and np.median(np.abs(out[st] - gt[st])) < 0.25


@pytest.mark.parametrize("n", [2048, 4000])
@pytest.mark.parametrize("rectified", [True, False])
def test_stereo_matching_flat(ctx, oracle, stream, pair, n, rectified):
    """ov2_stereo_matching: 2-level call on the keypoints with a prior, failures re-queued with the updated prior, the rest
    on the full pyramid, then the gate (row check + snap | Sampson)"""
    gl, gr, ol, orr = pair
    kps = synth.grid_keypoints(n, seed=33)
    gt = stream.stereo_gt(kps).astype(np.float32)
    pri, has = synth.make_priors(kps, gt, seed=8)
    # some hopeless priors so that the re-queue branch (:533-537) and the gate both have work
    rng = np.random.default_rng(2)
    bad = rng.uniform(size=n) < 0.08
    pri[bad & (has > 0)] += rng.normal(0, 25.0, (int((bad & (has > 0)).sum()), 2)).astype(np.float32)
    # keypoints WITHOUT a 2-level prior still carry one for the full pyramid (the SAD prior of rectified rigs, :431-435)
    sadlike = (has == 0) & (rng.uniform(size=n) < 0.6)
    pri[sadlike, 0] = (gt[sadlike, 0] + rng.normal(0, 3.0, int(sadlike.sum()))).astype(np.float32)
    lunpx = kps.copy()
    lunpx[::7, 1] += np.float32(1.9)      # undistorted left pixels that differ from the raw ones (gate input only)
    lunpx[::11, 1] += np.float32(2.3)
    trk = fe.FeatureTracker(ctx, 30, 0.01)
    out, st = trk.stereoMatching(gl, gr, 9, 3, 30.0, 0.5, kps, pri, has, lunpx=lunpx, rectified=rectified, F_rl=F_RL)
    eo, es = oracle.stereo_matching(ol, orr, kps, pri, has, 9, 3, 30.0, 0.5, 30, 0.01, lunpx=lunpx, rectified=rectified,
                                    F_rl=F_RL)
    assert np.array_equal(st, es)
    assert np.array_equal(out.view(np.uint32), eo.view(np.uint32))
    assert 0.5 < st.mean() < 1.0          # the gate removes some
    if rectified:
        assert np.array_equal(out[st][:, 1], lunpx[st][:, 1])
    # lunpx = NULL path
    out2, st2 = trk.stereoMatching(gl, gr, 9, 3, 30.0, 0.5, kps, pri, has, rectified=rectified, F_rl=F_RL)
    eo2, es2 = oracle.stereo_matching(ol, orr, kps, pri, has, 9, 3, 30.0, 0.5, 30, 0.01, rectified=rectified, F_rl=F_RL)
    assert np.array_equal(st2, es2) and np.array_equal(out2.view(np.uint32), eo2.view(np.uint32))


def test_stereo_matching_batched_dev(ctx, oracle, stream):
    """device-resident form on a batch of 3 stereo pairs with an image index per keypoint"""
    B, n = 3, 700
    il, ir = fe.Images(ctx, B, synth.IMG_W, synth.IMG_H), fe.Images(ctx, B, synth.IMG_W, synth.IMG_H)
    Ls, Rs = [stream.left(3 * b) for b in range(B)], [stream.right(3 * b) for b in range(B)]
    for b in range(B):
        il.upload(b, Ls[b]); ir.upload(b, Rs[b])
    gl, gr = fe.preprocess_images(ctx, il), fe.preprocess_images(ctx, ir)
    kps = [synth.grid_keypoints(n, seed=40 + b) for b in range(B)]
    pr = [synth.make_priors(k, stream.stereo_gt(k).astype(np.float32), seed=50 + b) for b, k in enumerate(kps)]
    d_k, d_p = ctx.to_device(np.concatenate(kps)), ctx.to_device(np.concatenate([p[0] for p in pr]))
    d_h = ctx.to_device(np.concatenate([p[1] for p in pr]))
    d_i = ctx.to_device(np.repeat(np.arange(B, dtype=np.int32), n))
    d_o, d_s = ctx.empty((B * n, 2), np.float32), ctx.empty((B * n,), np.uint8)
    trk = fe.FeatureTracker(ctx, 30, 0.01)
    trk.stereoMatching_dev(gl, gr, 9, 3, 30.0, 0.5, d_k, d_p, d_h, d_o, d_s, B * n, d_i, None, True, None)
    ctx.synchronize()
    out, st = d_o.get(), d_s.get().astype(bool)
    for b in range(B):
        ol, orr = oracle.Pyramid(oracle.clahe(Ls[b])), oracle.Pyramid(oracle.clahe(Rs[b]))
        eo, es = oracle.stereo_matching(ol, orr, kps[b], pr[b][0], pr[b][1])
        assert np.array_equal(st[b * n:(b + 1) * n], es)
        assert np.array_equal(out[b * n:(b + 1) * n].view(np.uint32), eo.view(np.uint32))


@pytest.mark.parametrize("level", [3, 2])
def test_line_min_sad_bit_exact(ctx, oracle, stream, pair, level):
    """FeatureTracker::getLineMinSAD as stereoMatching calls it: coarsest level, kp.px_ * 2^-level, window 7, going left;
    border points exercise the shrinking / growing half window and cv::getRectSubPix's replicated border"""
    gl, gr, ol, orr = pair
    kps = synth.grid_keypoints(2048, seed=35)
    pts = (kps * np.float32(1.0 / (1 << level))).astype(np.float32)
    w, h = gl.level_size(level)[:2]
    edge = np.float32([[1.5, 30.2], [2.9, 2.9], [w - 1.3, 30.0], [w - 0.1, h - 0.1], [50.0, 0.4], [50.3, h - 0.4], [0.2, 0.2],
                       [3.0, 3.0], [w - 3.01, h - 3.0], [w - 4.0, 10.0], [40.5, h - 2.2], [-1.0, 5.0], [w + 2.0, 5.0]])
    pts = np.concatenate([pts, edge])
    trk = fe.FeatureTracker(ctx, 30, 0.01)
    for go_left in (True, False):
        xp, er = trk.getLineMinSAD(gl, gr, level, pts, 7, go_left)
        inside = (pts[:, 0] >= 0) & (pts[:, 0] < w) & (pts[:, 1] >= 0) & (pts[:, 1] < h)
        ex, ee = oracle.line_min_sad(ol, orr, level, pts[inside], 7, go_left)
        assert np.array_equal(xp[inside].view(np.uint32), ex.view(np.uint32))
        found = ex >= 0
        assert np.array_equal(er[inside][found].view(np.uint32), ee[found].view(np.uint32))
        assert (xp[~inside] == -1).all()           # documented: points outside the level image give -1
    # the prior it yields is the disparity (rectified pair), as src/map_manager.cpp:431-435 uses it
    xp, _ = trk.getLineMinSAD(gl, gr, level, pts[:2048], 7, True)
    gt = stream.stereo_gt(kps)[:, 0] / (1 << level)
    ok = (xp >= 0) & (pts[:2048, 0] > 40.0 / (1 << level) + 4)
    assert np.median(np.abs(xp[ok] - gt[ok])) < 1.0


# EuRoC cam1 (parameters_files/accurate/euroc/euroc_stereo.yaml): radial-tangential coefficients of the right camera; the
# reference runs these files with bdo_stereo_rect 0 / bdo_undist 0, i.e. through the Sampson gate on undistorted pixels
EUROC_D1 = [-0.28368365, 0.07451284, -0.00010473, -3.55590700e-05]
K_R4 = [457.587, 456.134, 379.999, 255.238]


@pytest.mark.parametrize("model,coeffs", [("radtan", EUROC_D1), ("fisheye", [-0.012, 0.009, -0.004, 0.001])])
@pytest.mark.parametrize("rectified", [False, True])
def test_stereo_gate_undistorts_the_right_pixel(ctx, oracle, stream, pair, model, coeffs, rectified):
    """the epipolar gate with a distorted right camera (MapManager::stereoMatching :586: runpx = undistortImagePoint(r)):
    same statuses and bit-identical stored right pixels as the oracle, and NOT the result of gating the raw pixels"""
    from ov2slam_amd.ba_types import CamModelC
    gl, gr, ol, orr = pair
    n = 2048
    kps = synth.grid_keypoints(n, seed=35)
    gt = stream.stereo_gt(kps).astype(np.float32)
    pri, has = synth.make_priors(kps, gt, seed=9)
    cam = CamModelC.make(K_R4, model, coeffs)
    # left undistorted pixels consistent with the lens: the right track undistorted, moved back by the disparity, so that the
    # undistorted pair passes the gate while the raw pair (several px of distortion off the axis) often does not
    trk = fe.FeatureTracker(ctx, 30, 0.01)
    raw, st0 = trk.stereoMatching(gl, gr, 9, 3, 30.0, 0.5, kps, pri, has, lunpx=kps, rectified=True)
    und = oracle.cam_undistort(cam, raw)
    lunpx = kps.copy()
    lunpx[:, 1] = und[:, 1] + np.float32(0.3)
    rxy, st = trk.stereoMatching(gl, gr, 9, 3, 30.0, 0.5, kps, pri, has, lunpx=lunpx, rectified=rectified, F_rl=F_RL.ravel(), right_cam=cam)
    exy, est = oracle.stereo_matching(ol, orr, kps, pri, has, lunpx=lunpx, rectified=rectified, F_rl=F_RL.ravel(), right_cam=cam)
    exact = model == "radtan"   # the fisheye model goes through tan(): the last bit of libm and the device may differ
    assert (np.array_equal(st, est) if exact else (st != est).mean() < 0.002)
    same = st == est
    assert np.array_equal(rxy[same].view(np.uint32), exy[same].view(np.uint32))
    nod, _ = trk.stereoMatching(gl, gr, 9, 3, 30.0, 0.5, kps, pri, has, lunpx=lunpx, rectified=rectified, F_rl=F_RL.ravel())[1], None
    assert st.sum() > 0.5 * n
    if rectified:
        assert (nod != st).sum() > 20, "the lens model made no difference to the gate"


def test_stereo_matching_edge_cases(ctx, oracle, stream, pair):
    """an empty call; a call in which no keypoint has a 2-level prior (everything goes to the full pyramid); keypoints on
    the image border and priors far outside the image: status / positions still identical to the oracle, nothing faults"""
    gl, gr, ol, orr = pair
    trk = fe.FeatureTracker(ctx, 30, 0.01)
    out, st = trk.stereoMatching(gl, gr, 9, 3, 30.0, 0.5, np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), np.zeros(0, np.uint8))
    assert out.shape == (0, 2) and st.shape == (0,)
    W, H = 752, 480
    kps = np.array([[0, 0], [W - 1, 0], [0, H - 1], [W - 1, H - 1], [4.5, 4.5], [W - 5.5, H - 5.5], [1, 240], [750, 240],
                    [376, 1], [376, 478], [30.25, 30.75], [700.5, 20.5]], np.float32)
    kps = np.concatenate([kps, synth.grid_keypoints(300, seed=77)]).astype(np.float32)
    n = len(kps)
    gt = stream.stereo_gt(kps).astype(np.float32)
    cases = {"no_priors": (kps.copy(), np.zeros(n, np.uint8)),
             "wild_priors": (gt + np.float32(1e4) * (np.arange(n)[:, None] % 3 - 1).astype(np.float32), np.ones(n, np.uint8)),
             "true_priors": (gt, np.ones(n, np.uint8))}
    for name, (pri, has) in cases.items():
        for rect in (True, False):
            out, st = trk.stereoMatching(gl, gr, 9, 3, 30.0, 0.5, kps, pri, has, lunpx=kps, rectified=rect, F_rl=F_RL)
            eo, es = oracle.stereo_matching(ol, orr, kps, pri, has, 9, 3, 30.0, 0.5, 30, 0.01, lunpx=kps, rectified=rect, F_rl=F_RL)
            assert np.array_equal(st, es.astype(bool)), (name, rect)
            assert np.array_equal(out[st].view(np.uint32), eo[st].view(np.uint32)), (name, rect)
    out, st = trk.stereoMatching(gl, gr, 9, 3, 30.0, 0.5, kps, cases["true_priors"][0], cases["true_priors"][1], lunpx=kps, rectified=True)
    assert st[12:].mean() > 0.8 and not st[:4].any()   # the four corners cannot be tracked (window leaves the image)
