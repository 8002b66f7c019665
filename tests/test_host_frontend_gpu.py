"""The C++ host mirror's MapManager::stereoMatching and VisualFrontEnd::kltTracking (ov2slam_amd/host/, the reference
call surface of SURVEY.md 8b) driven through libov2host.so on a synthetic keyframe, against the same flow assembled from
the oracle's pieces in numpy: priors exactly as src/map_manager.cpp:385-490 / src/visual_front_end.cpp:142-184 build
them, then oracle.stereo_matching / oracle.klt_tracking_frame.  Bar: same stereo / tracked keypoint sets, identical
float32 pixels."""
import numpy as np
import pytest

from ov2slam_amd import host_map, synth

pytestmark = pytest.mark.gpu

K4 = np.array([458.654, 457.296, 367.215, 248.375])
BASE = 0.110074
W, H = synth.IMG_W, synth.IMG_H
CELL = 35


def _scene(stream, n=900, seed=3):
    """keypoints of a keyframe at identity pose: 60 % 3D (depth from the synthetic disparity, so that the right-camera
    reprojection is the true match), 40 % 2D; two 3D keypoints whose map point has been removed"""
    rng = np.random.default_rng(seed)
    kps = synth.grid_keypoints(n, seed=seed + 1)
    is3d = rng.uniform(size=n) < 0.6
    d = stream.disparity(kps[:, 1].astype(np.float64)) * rng.normal(1.0, 0.02, n)   # slightly wrong depths
    z = K4[0] * BASE / d
    xyz = np.stack([(kps[:, 0].astype(np.float64) - K4[2]) / K4[0] * z, (kps[:, 1].astype(np.float64) - K4[3]) / K4[1] * z, z], 1)
    return kps, is3d, xyz


def _build(stream, rect, **kw):
    kps, is3d, xyz = _scene(stream)
    fr = host_map.FrontEndFrame(K4, BASE, W, H, ncellsize=CELL, stereo_rect=rect, **kw)
    for i in range(len(kps)):
        fr.add_keypoint(i, kps[i], xyz[i] if is3d[i] else None)
    return fr, kps, is3d, xyz


def _right_proj(p_cam):
    """Frame::projCamToRightImageDist for the pinhole pair (Tcic0 = translation by -baseline)"""
    x, y, z = p_cam[0] - BASE, p_cam[1], p_cam[2]
    invz = 1.0 / z
    return np.float32(K4[0] * (x * invz) + K4[2]), np.float32(K4[1] * (y * invz) + K4[3])


@pytest.mark.parametrize("rect", [True, False])
def test_map_manager_stereo_matching(ctx, oracle, stream, rect):
    fr, kps, is3d, xyz = _build(stream, rect)
    gone = [i for i in range(len(kps)) if is3d[i]][:2]
    for i in gone:
        fr.forget_landmark(i)
    L, R = stream.left(4), stream.right(4)
    assert fr.stereo_matching(ctx, L, R) == 0
    got = fr.keypoints()
    assert all(i not in got for i in gone)                      # :414 removeMapPointObs
    # ---- the same flow from the oracle's pieces
    ol, orr = oracle.Pyramid(oracle.clahe(L)), oracle.Pyramid(oracle.clahe(R))
    ids = [i for i in range(len(kps)) if i not in gone]
    prior, has = {}, {}
    # grid of the frame (src/frame.cpp:40-45, 508-519): cell -> ids in insertion order
    nbw = int(np.ceil(np.float32(W) / np.float32(CELL)))
    cells = {}
    for i in range(len(kps)):
        r, c = int(np.floor(kps[i, 1] / np.float32(CELL))), int(np.floor(kps[i, 0] / np.float32(CELL)))
        cells.setdefault(r * nbw + c, []).append(i)
    for i in gone:                                               # removeKeypointById takes them out of the grid
        r, c = int(np.floor(kps[i, 1] / np.float32(CELL))), int(np.floor(kps[i, 0] / np.float32(CELL)))
        cells[r * nbw + c].remove(i)
    sad_ids = []
    for i in ids:
        if is3d[i]:
            u, v = _right_proj(xyz[i])
            if 0 <= u < W and 0 <= v < H:
                prior[i], has[i] = (u, v), 1
                continue
        if rect:
            sad_ids.append(i)
            prior[i], has[i] = (kps[i, 0], kps[i, 1]), 0
        else:
            r, c = int(np.floor(kps[i, 1] / np.float32(CELL))), int(np.floor(kps[i, 0] / np.float32(CELL)))
            near = []
            for rr in (r - 1, r):
                for cc in (c - 1, c):
                    if rr < 0 or cc < 0:
                        continue
                    near += [j for j in cells.get(rr * nbw + cc, []) if j != i]
            near3d = [j for j in near if is3d[j] and j not in gone]
            mean_z = weights = 0.0
            for j in near3d:
                dx, dy = np.float32(kps[j, 0] - kps[i, 0]), np.float32(kps[j, 1] - kps[i, 1])
                coef = 1.0 / np.sqrt(float(dx) * float(dx) + float(dy) * float(dy))
                weights += coef
                mean_z += coef * xyz[j, 2]
            prior[i], has[i] = (kps[i, 0], kps[i, 1]), 0
            if near3d:
                mean_z /= weights
                bv = np.array([(float(kps[i, 0]) - K4[2]) / K4[0], (float(kps[i, 1]) - K4[3]) / K4[1], 1.0])
                bv = bv / np.sqrt(bv[0] * bv[0] + bv[1] * bv[1] + bv[2] * bv[2])
                u, v = _right_proj(np.array([mean_z * (bv[0] / bv[2]), mean_z * (bv[1] / bv[2]), mean_z * (bv[2] / bv[2])]))
                if 0 <= u < W and 0 <= v < H:
                    prior[i], has[i] = (u, v), 1
    if sad_ids:
        pts = (kps[sad_ids] * np.float32(1.0 / 8.0)).astype(np.float32)
        xp, _ = oracle.line_min_sad(ol, orr, 3, pts, 7, True)
        for k, i in enumerate(sad_ids):
            x = np.float32(xp[k] * np.float32(8.0))
            if 0 <= x <= kps[i, 0]:
                prior[i] = (x, kps[i, 1])
    pk = kps[ids]
    pp = np.float32([prior[i] for i in ids])
    ph = np.uint8([has[i] for i in ids])
    assert 0.2 < ph.mean() < 1.0
    eo, es = oracle.stereo_matching(ol, orr, pk, pp, ph, 9, 3, 30.0, 0.5, 30, 0.01, rectified=rect, F_rl=fr.frl())
    assert es.mean() > 0.6
    for k, i in enumerate(ids):
        px, i3, ist, rpx = got[i]
        assert ist == bool(es[k]), (i, k)
        if es[k]:
            assert np.array_equal(rpx.view(np.uint32), eo[k].view(np.uint32)), (i, rpx, eo[k])
    # F_rl of the mirror = K^-T [t]x K^-1 for the x baseline (src/frame.cpp:53-62)
    Km = np.array([[K4[0], 0, K4[2]], [0, K4[1], K4[3]], [0, 0, 1]])
    tx = np.array([[0, 0, 0], [0, 0, BASE], [0, -BASE, 0]])    # t = Tcic0.translation() = (-b, 0, 0)
    assert np.allclose(fr.frl(), np.linalg.inv(Km).T @ tx @ np.linalg.inv(Km), rtol=1e-12, atol=1e-18)


@pytest.mark.parametrize("use_prior", [True, False])
def test_visual_front_end_klt_tracking(ctx, oracle, stream, use_prior):
    """VisualFrontEnd::kltTracking: 3D keypoints start from the reprojection of their map point (2 levels), failures and
    2D keypoints go through the full pyramid, lost keypoints leave the frame"""
    t0, t1 = 3, 9
    kps, is3d, _ = _scene(stream, seed=11)
    # world points such that the frame (identity pose) sees them at the position they have in the CURRENT image
    cur_gt = stream.flow(t0, t1, kps)
    noisy = cur_gt + np.random.default_rng(1).normal(0, 0.7, cur_gt.shape)
    z = np.full(len(kps), 5.0)
    xyz = np.stack([(noisy[:, 0] - K4[2]) / K4[0] * z, (noisy[:, 1] - K4[3]) / K4[1] * z, z], 1)
    xyz[5] = [100.0, 0.0, 1.0]                  # projects outside: falls back to the no-prior list (:166-171)
    fr = host_map.FrontEndFrame(K4, BASE, W, H, ncellsize=CELL, klt_use_prior=use_prior)
    for i in range(len(kps)):
        fr.add_keypoint(i, kps[i], xyz[i] if is3d[i] else None)
    I0, I1 = stream.left(t0), stream.left(t1)
    st, p3p = fr.klt_tracking(ctx, I0, I1)
    assert st == 0
    got = fr.keypoints()
    prior, has = kps.copy(), np.zeros(len(kps), np.uint8)
    if use_prior:
        for i in range(len(kps)):
            if is3d[i]:
                invz = 1.0 / xyz[i, 2]
                u = np.float32(K4[0] * (xyz[i, 0] * invz) + K4[2]); v = np.float32(K4[1] * (xyz[i, 1] * invz) + K4[3])
                if 0 <= u < W and 0 <= v < H:
                    prior[i], has[i] = (u, v), 1
    o0, o1 = oracle.Pyramid(oracle.clahe(I0)), oracle.Pyramid(oracle.clahe(I1))
    eo, es, ep3p = oracle.klt_tracking_frame(o0, o1, kps, prior, has, 9, 3, 30.0, 0.5, 30, 0.01)
    assert p3p == ep3p
    assert es.mean() > 0.8
    assert set(got.keys()) == {i for i in range(len(kps)) if es[i]}       # removeObsFromCurFrameById for the lost ones
    for i in got:
        assert np.array_equal(got[i][0].view(np.uint32), eo[i].view(np.uint32))
