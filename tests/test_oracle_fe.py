"""CPU tests of the front-end oracle itself (no GPU): self-consistency on the synthetic stream and a
brute-force numpy re-derivation of the integer stages.  OpenCV is not in the container and the reference
holds no fixture for this path, so the KLT oracle is 'parity unpinned' (see oracle/ov2_oracle.h); these
tests pin it against independent numpy formulas of the published algorithm instead."""
import numpy as np

from ov2slam_amd import synth


def _reflect(i, n):
    i = np.abs(i)
    return np.where(i >= n, 2 * (n - 1) - i, i)


def test_pyrdown_and_scharr_against_numpy(oracle):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(37, 53), dtype=np.uint8)
    P = oracle.Pyramid(img, 9, 1)
    i0, g0, w, h, p = P.level(0)
    assert (w, h, p) == (53, 37, 9)
    # level 0 interior and REFLECT_101 border
    assert np.array_equal(i0[p:p + h, p:p + w], img)
    ys = _reflect(np.arange(-p, h + p), h)
    xs = _reflect(np.arange(-p, w + p), w)
    assert np.array_equal(i0, img[np.ix_(ys, xs)])
    # Scharr with numpy (int32)
    a = img.astype(np.int32)
    pad = a[np.ix_(_reflect(np.arange(-1, h + 1), h), _reflect(np.arange(-1, w + 1), w))]
    t0 = 3 * (pad[:-2] + pad[2:]) + 10 * pad[1:-1]
    t1 = pad[2:] - pad[:-2]
    ix = t0[:, 2:] - t0[:, :-2]
    iy = 3 * (t1[:, :-2] + t1[:, 2:]) + 10 * t1[:, 1:-1]
    assert np.array_equal(g0[p:p + h, p:p + w, 0], ix) and np.array_equal(g0[p:p + h, p:p + w, 1], iy)
    assert not g0[:p].any() and not g0[:, :p].any() and not g0[p + h:].any() and not g0[:, p + w:].any()
    # pyrDown with numpy
    i1, _, w1, h1, _ = P.level(1)
    assert (w1, h1) == (27, 19)
    k = np.array([1, 4, 6, 4, 1])
    exp = np.zeros((h1, w1), np.int64)
    for j in range(5):
        for i in range(5):
            yy = _reflect(2 * np.arange(h1) - 2 + j, h)
            xx = _reflect(2 * np.arange(w1) - 2 + i, w)
            exp += k[j] * k[i] * a[np.ix_(yy, xx)]
    assert np.array_equal(i1[p:p + h1, p:p + w1], ((exp + 128) >> 8).astype(np.uint8))


def test_pyramid_early_stop(oracle):
    img = np.zeros((40, 30), np.uint8)
    assert oracle.Pyramid(img, 9, 3).nlevels == 2   # 15x20 ok, next 8x10 <= 9 stops (buildOpticalFlowPyramid)


def test_clahe_properties(oracle):
    rng = np.random.default_rng(2)
    img = rng.integers(60, 120, size=(480, 752), dtype=np.uint8)
    out = oracle.clahe(img, 3.0, 15, 9)
    assert out.shape == img.shape and out.std() > img.std()        # contrast is stretched
    flat = np.full((480, 752), 77, np.uint8)
    f = oracle.clahe(flat, 3.0, 15, 9)
    assert len(np.unique(f)) == 1                                   # constant in, constant out
    # clip=0 disables clipping -> plain tile histogram equalisation; LUT must be monotone: order preserved per tile centre
    o2 = oracle.clahe(img, 0.0, 1, 1)
    order = np.argsort(img.ravel(), kind="stable")
    assert (np.diff(o2.ravel()[order].astype(int)) >= 0).all()


def test_lk_recovers_known_subpixel_shift(oracle, stream):
    I0, I1 = stream.left(0), stream.left(10)
    P0, P1 = oracle.Pyramid(oracle.clahe(I0)), oracle.Pyramid(oracle.clahe(I1))
    kps = synth.grid_keypoints(600)
    gt = stream.flow(0, 10, kps)
    out, st, iters = oracle.fb_klt_tracking(P0, P1, kps, kps)
    e = np.linalg.norm(out[st > 0] - gt[st > 0], axis=1)
    assert st.mean() > 0.97 and np.median(e) < 0.15 and e.max() < 1.5   # 2 % zoom over 10 frames: translational LK model error
    assert iters > 0
    # identity pair: zero motion, every point survives the forward-backward gate
    out2, st2, _ = oracle.fb_klt_tracking(P0, P0, kps, kps)
    assert st2.all() and np.abs(out2 - kps).max() < 1e-3


def test_fb_wrapper_gates(oracle, stream):
    I0 = stream.left(0).copy()
    I0[200:300, 200:400] = 90
    P0 = oracle.Pyramid(I0)
    kps = np.array([[300.0, 250.0], [0.2, 0.3], [100.0, 100.0]], np.float32)
    out, st, _ = oracle.fb_klt_tracking(P0, P0, kps, kps)
    assert list(st) == [0, 0, 1]       # flat -> minEig reject ; (0.2,0.3) fails inBorder ; textured ok
    o, s, _ = oracle.fb_klt_tracking(P0, P0, np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32))
    assert o.shape == (0, 2) and s.shape == (0,)


def test_klt_tracking_frame_batches(oracle, stream):
    P0, P1 = oracle.Pyramid(stream.left(0)), oracle.Pyramid(stream.left(10))
    kps = synth.grid_keypoints(500)
    gt = stream.flow(0, 10, kps)
    pri, has = synth.make_priors(kps, gt, sigma=1.0)
    out, st, p3p = oracle.klt_tracking_frame(P0, P1, kps, pri, has)
    assert not p3p and st.mean() > 0.95
    pri, has = synth.make_priors(kps, gt, sigma=25.0)
    out, st, p3p = oracle.klt_tracking_frame(P0, P1, kps, pri, has)
    assert p3p and st.mean() > 0.9      # priors dropped, full pyramid recovers
